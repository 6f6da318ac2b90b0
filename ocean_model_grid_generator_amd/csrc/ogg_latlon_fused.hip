// Lat-lon sub-grids, all six fields in one launch -- C-ABI entry points; the kernel is in ogg_latlon_fused_dev.h.
//
// Second form, for launches that carry NOTHING but lat-lon sub-grids (ogg_latlon_supergrid_rows_ws_dev): the write path of the chip
// wants compact windows that sweep through memory (scripts/microbench/write_patterns.hip pattern h), which the
// column-tile workgroups above -- 4 KB of a row, then the next row, six arrays interleaved -- are not.  Here the unit of work is ONE ROW
// OF ONE FIELD (46 KB contiguous), units are taken in (band, field, row) order by the persistent workgroups of the launch, so at any
// moment the chip writes a few thousand consecutive rows of one array; the per-row and per-column scalars come from two small tables (a
// first launch; L2-resident afterwards).  The same operations on the same operands as latlon_rows(): the same bits.  Measured on
// one box, same process (scripts/rows_probe.py): 1/8 degree 5.9 TB/s against 5.7 for the column-tile kernel, 1/16 degree 5.5 against
// 5.6 -- the difference between two MI355X boxes (4.6 .. 5.7 TB/s for the same binary) is larger than the difference between the two
// patterns, so the column-tile kernel stays the default and this one an option (OGG_LATLON_ROWS=1 in supergrid.py).
#include "ogg_latlon_fused_dev.h"

namespace {

struct RowsParams {
    FusedParams f;
    const RowScalars* row_tab;     // per band: n_pt_rows + 1 entries, band k at row0[k]
    long row0[LF_MAX_BANDS + 1];
    const double* col_tab;         // [4][ni1]: lon_c, dlam, hdlam, xdiff
    long unit0[LF_MAX_BANDS * 6 + 1];   // prefix sum of the units (rows) of (band, field), field order x, y, angle_dx, dx, dy, area
};

__global__ __launch_bounds__(256) void latlon_tables_kernel(RowsParams p) {
    const long k = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const FusedParams& f = p.f;
    const long n_rows = p.row0[f.n_bands];
    if (k < n_rows) {
        int bi = 0;
        while (bi + 1 < f.n_bands && k >= p.row0[bi + 1]) ++bi;
        const_cast<RowScalars*>(p.row_tab)[k] = latlon_row_scalars(f, f.band[bi], k - p.row0[bi]);
    } else if (k < n_rows + f.ni1) {
        const long i = k - n_rows;
        const ColScalars c = column_scalars(f, i);
        double* t = const_cast<double*>(p.col_tab);
        t[i] = c.lon_c, t[f.ni1 + i] = c.dlam, t[2 * f.ni1 + i] = c.hdlam, t[3 * f.ni1 + i] = c.xdiff;
    }
}

// one element of field `field` (0..5: x, y, angle_dx, dx, dy, area) at column e of a row with scalars rs (ds = sl_{j+1} - sl_j)
template <int FIELD>
OGG_DEV double rows_value(const RowsParams& p, const RowScalars& rs, double ds, const double* __restrict__ tab, long e) {
    if (FIELD == 0) return tab[e];                                                  // x = lon_c
    if (FIELD == 1) return rs.lat;
    if (FIELD == 2) return latlon_angle(tab[e] * rs.cl);                            // xdiff
    if (FIELD == 3) return p.f.Re * fabs(tab[e] * rs.cl);                           // dlam
    if (FIELD == 4) return rs.dy;
    return p.f.Re2 * (tab[e] * ds);                                                 // hdlam
}

template <int FIELD>
OGG_DEV void rows_write(const RowsParams& p, const RowScalars& rs, double ds, double* __restrict__ out, long len) {
    constexpr int T[6] = {0, 0, 3, 1, 0, 2};    // column table of the field
    const double* __restrict__ tab = p.col_tab + (long)T[FIELD] * p.f.ni1;
    const int tid = threadIdx.x;
    // 16-byte stores need a 16-byte aligned address: rows of odd length start on an odd element every other row -- peel one element
    const long peel = (reinterpret_cast<unsigned long long>(out) >> 3) & 1ull;
    if (peel && tid == 0) out[0] = rows_value<FIELD>(p, rs, ds, tab, 0);
    const long n_pairs = (len - peel) / 2;
    for (long q = tid; q < n_pairs; q += 256) {
        const long e = peel + 2 * q;
        dbl2 v;
        v.x = rows_value<FIELD>(p, rs, ds, tab, e), v.y = rows_value<FIELD>(p, rs, ds, tab, e + 1);
        *reinterpret_cast<dbl2*>(out + e) = v;
    }
    if (((len - peel) & 1) && tid == 255) out[len - 1] = rows_value<FIELD>(p, rs, ds, tab, len - 1);
}

__global__ __launch_bounds__(256) void latlon_rows_kernel(RowsParams p) {
    const FusedParams& f = p.f;
    const long n_units = p.unit0[f.n_bands * 6];
    const long ni1 = f.ni1, ni = ni1 - 1;
    for (long u = blockIdx.x; u < n_units; u += gridDim.x) {     // workgroup-uniform
        int s = 0;
        while (s + 1 < f.n_bands * 6 && u >= p.unit0[s + 1]) ++s;
        const int bi = s / 6, field = s % 6;
        const ogg_latlon_band& b = f.band[bi];
        const long j = u - p.unit0[s];
        const RowScalars rs = p.row_tab[p.row0[bi] + j];
        const double ds = (field == 5) ? p.row_tab[p.row0[bi] + j + 1].sl - rs.sl : 0.0;
        switch (field) {
            case 0: rows_write<0>(p, rs, ds, b.x + j * ni1, ni1); break;
            case 1: rows_write<1>(p, rs, ds, b.y + j * ni1, ni1); break;
            case 2: rows_write<2>(p, rs, ds, b.angle + j * ni1, ni1); break;
            case 3: rows_write<3>(p, rs, ds, b.dx + j * ni, ni); break;
            case 4: rows_write<4>(p, rs, ds, b.dy + j * ni1, ni1); break;
            default: rows_write<5>(p, rs, ds, b.area + j * ni, ni); break;
        }
    }
}

}  // namespace

extern "C" long ogg_latlon_rows_workspace_bytes(int n_bands, const ogg_latlon_band* bands, long ni1) {
    if (n_bands < 0 || n_bands > LF_MAX_BANDS || (n_bands && !bands) || ni1 < 2) return 0;
    long rows = 0;
    for (int k = 0; k < n_bands; ++k) rows += (bands[k].n_pt_rows > 0 ? bands[k].n_pt_rows : 0) + 1;
    return rows * (long)sizeof(RowScalars) + 4 * ni1 * (long)sizeof(double);
}

extern "C" int ogg_latlon_supergrid_rows_ws_dev(int n_bands, const ogg_latlon_band* bands, long ni1, double lon0, double lenlon, double Re,
                                                int metrics, void* workspace, long workspace_bytes, void* stream) {
    RowsParams r{};
    long points = 0;
    if (int e = plan_latlon(n_bands, bands, ni1, lon0, lenlon, Re, metrics, r.f, points)) return e;
    if (r.f.n_bands == 0) return OGG_OK;
    OGG_REQUIRE(workspace && workspace_bytes >= ogg_latlon_rows_workspace_bytes(n_bands, bands, ni1), OGG_EARG,
                "ogg_latlon_supergrid_rows: workspace too small (ogg_latlon_rows_workspace_bytes)");
    r.row0[0] = 0, r.unit0[0] = 0;
    for (int k = 0; k < r.f.n_bands; ++k) {
        const ogg_latlon_band& b = r.f.band[k];
        r.row0[k + 1] = r.row0[k] + b.n_pt_rows + 1;
        const long ncell = metrics ? b.n_cell_rows : 0;
        const long rows_of[6] = {b.n_pt_rows, b.n_pt_rows, b.n_pt_rows, metrics ? b.n_pt_rows : 0, ncell, ncell};
        for (int fld = 0; fld < 6; ++fld) r.unit0[k * 6 + fld + 1] = r.unit0[k * 6 + fld] + rows_of[fld];
    }
    r.row_tab = static_cast<const RowScalars*>(workspace);
    r.col_tab = reinterpret_cast<const double*>(r.row_tab + r.row0[r.f.n_bands]);
    hipStream_t st = ogg::as_stream(stream);
    const long n_tab = r.row0[r.f.n_bands] + ni1;
    latlon_tables_kernel<<<(unsigned)((n_tab + 255) / 256), 256, 0, st>>>(r);
    OGG_LAUNCH_CHECK();
    // one short-lived workgroup per unit: measured faster than 1024 .. 16384 persistent ones (1/8 degree, same box: 5.9 TB/s against
    // 4.1 .. 5.9; the column-tile kernel 5.7; 1/16 degree 5.5 against 5.6 -- scripts/rows_probe.py)
    long max_wg = 1 << 20;
    if (const char* e = getenv("OGG_ROWS_MAX_WG")) max_wg = atol(e) > 0 ? atol(e) : max_wg;
    const long n_units = r.unit0[r.f.n_bands * 6];
    latlon_rows_kernel<<<(unsigned)(n_units < max_wg ? n_units : max_wg), 256, 0, st>>>(r);
    OGG_LAUNCH_CHECK();
    return OGG_OK;
}

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
