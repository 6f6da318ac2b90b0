// Lat-lon sub-grids, all six fields in one launch -- C-ABI entry points; the kernel is in ogg_latlon_fused_dev.h.
#include "ogg_latlon_fused_dev.h"

extern "C" int ogg_latlon_supergrid_multi_dev(int n_bands, const ogg_latlon_band* bands, long ni1, double lon0, double lenlon,
                                              double Re, int metrics, void* stream) {
    FusedParams p;
    long points = 0;
    if (int e = plan_latlon(n_bands, bands, ni1, lon0, lenlon, Re, metrics, p, points)) return e;
    if (p.n_bands == 0) return OGG_OK;
    const long gx = latlon_gx(ni1);
    // Resident workgroups (each owns 512 columns and a contiguous block of row strips inside its XCD's eighth of the rows): the
    // write path saturates with 150-250 of them (1/8 degree, 19.8 M points: 60 -> 5.1 TB/s, 240 -> 4.9-5.5, 1024 -> 4.3; 1/16
    // degree: 240 -> 4.7 TB/s); without the per-XCD assignment the plateau was 4.3-4.6 TB/s.
    long max_wg = 240;
    if (const char* e = getenv("OGG_FUSED_MAX_WG")) max_wg = atol(e);
    long gy = p.strip0[p.n_bands];
    if (gx * gy > max_wg) gy = (max_wg + gx - 1) / gx;
    dim3 grid((unsigned)gx, (unsigned)(gy < 1 ? 1 : gy));
    latlon_fused_kernel<<<grid, LF_TX, 0, ogg::as_stream(stream)>>>(p);
    OGG_LAUNCH_CHECK();
    return OGG_OK;
}

extern "C" int ogg_latlon_supergrid_dev(long n_pt_rows, long n_cell_rows, long ni1, const double* lat1d, const double* lon1d,
                                        double Re, int metrics, double* x, double* y, double* dx, double* dy, double* area,
                                        double* angle, void* stream) {
    OGG_REQUIRE(lat1d && lon1d, OGG_EARG, "ogg_latlon_supergrid: null axis");
    // the longitude axis of every caller is lon0 + (i*len)/Ni; recover its two scalars from the device array
    double ends[2];
    OGG_HIP_CHECK(hipMemcpyAsync(&ends[0], lon1d, sizeof(double), hipMemcpyDeviceToHost, ogg::as_stream(stream)));
    OGG_HIP_CHECK(hipMemcpyAsync(&ends[1], lon1d + (ni1 - 1), sizeof(double), hipMemcpyDeviceToHost, ogg::as_stream(stream)));
    OGG_HIP_CHECK(hipStreamSynchronize(ogg::as_stream(stream)));
    ogg_latlon_band b{};
    b.axis_kind = 2, b.lat1d = lat1d, b.k0 = 0, b.n_pt_rows = n_pt_rows, b.n_cell_rows = n_cell_rows;
    b.x = x, b.y = y, b.dx = dx, b.dy = dy, b.area = area, b.angle = angle;
    return ogg_latlon_supergrid_multi_dev(1, &b, ni1, ends[0], ends[1] - ends[0], Re, metrics, stream);
}
