// K1: 1-D axis builders and the 2-D lat-lon tile (pure streaming write, 16 B per point).
//   y_mercator / y_mercator_rounded / phi_mercator  OGG:292-311, axis OGG:336
//   linear axes                                     OGG:113,115,431,834,835
//   np.tile pair                                    OGG:430-432, 840-841
#include "ogg_common.h"
#include "ogg_math.h"

namespace {

using namespace ogg;

__global__ void y_mercator_rounded_kernel(long Ni, long n, const double* __restrict__ phi, long long* __restrict__ ystar) {
    const long k = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const double R = (double)Ni / (2 * kPi);
    const double p = phi[k];
    const double yf = R * log((1.0 + sin(p)) / cos(p));
    const double sgn = (yf > 0.0) ? 1.0 : ((yf < 0.0) ? -1.0 : 0.0);
    ystar[k] = (long long)(sgn * rint(fabs(yf)));  // rint: round-half-even, like numpy.round
}

__global__ void phi_mercator_kernel(long Ni, long n, const double* __restrict__ y, long long y0, double* __restrict__ phi) {
    const long k = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const double R = (double)Ni / (2 * kPi);
    const double yy = y ? y[k] : (double)(y0 + k);
    phi[k] = atan(sinh(yy / R)) * k180Pi;
}

__global__ void linear_axis_kernel(long n, double a0, double len, double denom, double* __restrict__ out) {
    const long k = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    out[k] = a0 + ((double)k * len) / denom;
}

__global__ void fill_kernel(long n, double v, double* __restrict__ out) {
    long k = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long stride = (long)gridDim.x * blockDim.x;
    for (; k < n; k += stride) out[k] = v;
}

constexpr int TILE_TX = 256;
constexpr int TILE_ROWS = 16;

__global__ __launch_bounds__(TILE_TX) void tile_latlon_kernel(long nrows, long ni1, const double* __restrict__ lat1d,
                                                              const double* __restrict__ lon1d, double* __restrict__ x,
                                                              double* __restrict__ y) {
    const long v = xcd_contiguous((long)blockIdx.y * gridDim.x + blockIdx.x, (long)gridDim.x * gridDim.y);  // one row range per XCD
    const long i = (v % gridDim.x) * TILE_TX + threadIdx.x;
    if (i >= ni1) return;
    const double lon = lon1d[i];
    const long j0 = (v / gridDim.x) * TILE_ROWS;
    const long j1 = (j0 + TILE_ROWS < nrows) ? j0 + TILE_ROWS : nrows;
    for (long j = j0; j < j1; ++j) {
        const double lat = lat1d[j];  // wave-uniform: scalar load
        x[j * ni1 + i] = lon;
        y[j * ni1 + i] = lat;
    }
}

inline unsigned blocks_for(long n, int bs) { return (unsigned)((n + bs - 1) / bs); }

}  // namespace

extern "C" {

int ogg_y_mercator_rounded_dev(long Ni, long n, const double* phi_rad, long long* ystar, void* stream) {
    OGG_REQUIRE(Ni > 0 && n >= 0, OGG_EARG, "ogg_y_mercator_rounded: bad Ni/n");
    OGG_REQUIRE(phi_rad && ystar, OGG_EARG, "ogg_y_mercator_rounded: null pointer");
    if (n == 0) return OGG_OK;
    y_mercator_rounded_kernel<<<blocks_for(n, 64), 64, 0, ogg::as_stream(stream)>>>(Ni, n, phi_rad, ystar);
    OGG_LAUNCH_CHECK();
    return OGG_OK;
}

int ogg_phi_mercator_dev(long Ni, long n, const double* y, double* phi_deg, void* stream) {
    OGG_REQUIRE(Ni > 0 && n >= 0, OGG_EARG, "ogg_phi_mercator: bad Ni/n");
    OGG_REQUIRE(y && phi_deg, OGG_EARG, "ogg_phi_mercator: null pointer");
    if (n == 0) return OGG_OK;
    phi_mercator_kernel<<<blocks_for(n, 256), 256, 0, ogg::as_stream(stream)>>>(Ni, n, y, 0, phi_deg);
    OGG_LAUNCH_CHECK();
    return OGG_OK;
}

int ogg_mercator_axis_dev(long Ni, long long y0, long n, double* phi_deg, void* stream) {
    OGG_REQUIRE(Ni > 0 && n >= 0, OGG_EARG, "ogg_mercator_axis: bad Ni/n");
    OGG_REQUIRE(phi_deg, OGG_EARG, "ogg_mercator_axis: null pointer");
    if (n == 0) return OGG_OK;
    phi_mercator_kernel<<<blocks_for(n, 256), 256, 0, ogg::as_stream(stream)>>>(Ni, n, nullptr, y0, phi_deg);
    OGG_LAUNCH_CHECK();
    return OGG_OK;
}

int ogg_linear_axis_dev(long n, double a0, double len, double denom, double* out, void* stream) {
    OGG_REQUIRE(n >= 0 && out, OGG_EARG, "ogg_linear_axis: bad argument");
    if (n == 0) return OGG_OK;
    linear_axis_kernel<<<blocks_for(n, 256), 256, 0, ogg::as_stream(stream)>>>(n, a0, len, denom, out);
    OGG_LAUNCH_CHECK();
    return OGG_OK;
}

int ogg_fill_dev(long n, double value, double* out, void* stream) {
    OGG_REQUIRE(n >= 0 && out, OGG_EARG, "ogg_fill: bad argument");
    if (n == 0) return OGG_OK;
    unsigned nb = blocks_for(n, 256);
    if (nb > 4096) nb = 4096;
    fill_kernel<<<nb, 256, 0, ogg::as_stream(stream)>>>(n, value, out);
    OGG_LAUNCH_CHECK();
    return OGG_OK;
}

int ogg_tile_latlon_dev(long nrows, long ni1, const double* lat1d, const double* lon1d, double* x, double* y, void* stream) {
    OGG_REQUIRE(nrows >= 0 && ni1 > 0, OGG_ESHAPE, "ogg_tile_latlon: bad shape %ld x %ld", nrows, ni1);
    OGG_REQUIRE(lat1d && lon1d && x && y, OGG_EARG, "ogg_tile_latlon: null pointer");
    if (nrows == 0) return OGG_OK;
    dim3 grid(blocks_for(ni1, TILE_TX), blocks_for(nrows, TILE_ROWS));
    tile_latlon_kernel<<<grid, TILE_TX, 0, ogg::as_stream(stream)>>>(nrows, ni1, lat1d, lon1d, x, y);
    OGG_LAUNCH_CHECK();
    return OGG_OK;
}

}  // extern "C"
