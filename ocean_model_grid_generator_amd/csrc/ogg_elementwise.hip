// Small element-wise kernels that complete the reference's function surface at the C ABI:
//   mdist OGG:682-684, y_mercator OGG:292-295, the index->angle maps of OGG:126-127 / 479-482,
//   great_arc_distance's haversine OGG:527-532, bipolar_cap_ij_array OGG:125-133.
#include "ogg_common.h"
#include "ogg_math.h"

namespace {
using namespace ogg;

__global__ void mdist_kernel(long n, const double* __restrict__ x1, const double* __restrict__ x2, double* __restrict__ out) {
    const long k = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) out[k] = mdist(x1[k], x2[k]);
}

__global__ void y_mercator_kernel(long Ni, long n, const double* __restrict__ phi, double* __restrict__ y) {
    const long k = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const double R = (double)Ni / (2 * kPi);
    const double p = phi[k];
    y[k] = R * log((1.0 + sin(p)) / cos(p));
}

__global__ void affine_index_kernel(long n, const double* __restrict__ idx, double a0, double len, double denom,
                                    double* __restrict__ out) {
    const long k = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) out[k] = a0 + (idx[k] * len) / denom;
}

__global__ void haversine_kernel(long n, const double* __restrict__ lam0d, const double* __restrict__ phi0d,
                                 const double* __restrict__ lam1d, const double* __restrict__ phi1d, double* __restrict__ out) {
    const long k = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const double lam0 = lam0d[k] * kPi180, phi0 = phi0d[k] * kPi180;
    const double lam1 = lam1d[k] * kPi180, phi1 = phi1d[k] * kPi180;
    const double dphi = phi1 - phi0, dlam = lam1 - lam0;
    const double sp = sin(0.5 * dphi), sl = sin(0.5 * dlam);
    const double d = sp * sp + sl * sl * cos(phi0) * cos(phi1);
    out[k] = 2.0 * asin(sqrt(d));
}

// write_nc's output is NetCDF classic (OGG:773-829): big-endian fp64.  Byte-swap-on-copy: 8-byte reversal of n values from device
// memory into dst, which may be device memory or PINNED HOST memory (hipHostMalloc: mapped into the device's address space, so the
// kernel's stores go straight over PCIe into the buffer the file is written from).  Grid-stride, one value per thread per step
// (src and dst rows are only 8-byte aligned); 16 B of traffic per value.
__global__ __launch_bounds__(256) void bswap64_kernel(long n, const unsigned long long* __restrict__ src, unsigned long long* __restrict__ dst) {
    const long stride = (long)gridDim.x * blockDim.x;
    for (long k = (long)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += stride) dst[k] = __builtin_bswap64(src[k]);
}

inline unsigned nblk(long n) { return (unsigned)((n + 255) / 256); }
}  // namespace

extern "C" {

int ogg_bswap64_dev(long n, const void* src, void* dst, void* stream) {
    OGG_REQUIRE(n >= 0 && src && dst, OGG_EARG, "ogg_bswap64: bad argument");
    if (n == 0) return OGG_OK;
    const long blocks = (n + 255) / 256;
    bswap64_kernel<<<(unsigned)(blocks < 4096 ? blocks : 4096), 256, 0, ogg::as_stream(stream)>>>(
        n, static_cast<const unsigned long long*>(src), static_cast<unsigned long long*>(dst));
    OGG_LAUNCH_CHECK();
    return OGG_OK;
}

int ogg_mdist_dev(long n, const double* x1, const double* x2, double* out, void* stream) {
    OGG_REQUIRE(n >= 0 && x1 && x2 && out, OGG_EARG, "ogg_mdist: bad argument");
    if (n == 0) return OGG_OK;
    mdist_kernel<<<nblk(n), 256, 0, ogg::as_stream(stream)>>>(n, x1, x2, out);
    OGG_LAUNCH_CHECK();
    return OGG_OK;
}

int ogg_y_mercator_dev(long Ni, long n, const double* phi_rad, double* y, void* stream) {
    OGG_REQUIRE(Ni > 0 && n >= 0 && phi_rad && y, OGG_EARG, "ogg_y_mercator: bad argument");
    if (n == 0) return OGG_OK;
    y_mercator_kernel<<<nblk(n), 256, 0, ogg::as_stream(stream)>>>(Ni, n, phi_rad, y);
    OGG_LAUNCH_CHECK();
    return OGG_OK;
}

int ogg_affine_index_dev(long n, const double* idx, double a0, double len, double denom, double* out, void* stream) {
    OGG_REQUIRE(n >= 0 && idx && out, OGG_EARG, "ogg_affine_index: bad argument");
    if (n == 0) return OGG_OK;
    affine_index_kernel<<<nblk(n), 256, 0, ogg::as_stream(stream)>>>(n, idx, a0, len, denom, out);
    OGG_LAUNCH_CHECK();
    return OGG_OK;
}

int ogg_haversine_dev(long n, const double* lam0, const double* phi0, const double* lam1, const double* phi1, double* out,
                      void* stream) {
    OGG_REQUIRE(n >= 0 && lam0 && phi0 && lam1 && phi1 && out, OGG_EARG, "ogg_haversine: bad argument");
    if (n == 0) return OGG_OK;
    haversine_kernel<<<nblk(n), 256, 0, ogg::as_stream(stream)>>>(n, lam0, phi0, lam1, phi1, out);
    OGG_LAUNCH_CHECK();
    return OGG_OK;
}

}  // extern "C"
