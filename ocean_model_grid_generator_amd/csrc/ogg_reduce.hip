// The sums behind metrics_error (OGG:732-770) taken on the device, so that the self-check of a band-sharded supergrid
// needs no field on the host: five doubles per sub-grid band, then (supergrid.py) one all-reduce over the ranks.
//   out[0] = sum(area)                 OGG:750      out[1] = sum(dy[:, col_a])   OGG:739 / 760
//   out[2] = sum(dy[:, col_b]) or 0    OGG:760      out[3] = sum(dx[0, :])       OGG:740
//   out[4] = sum(dx[-1, :])            OGG:742, 745
// Deterministic: fixed assignment of elements to threads, tree reductions in LDS, partial sums combined in index order.
#include "ogg_common.h"
#include "ogg_math.h"

namespace {

constexpr int RED_TX = 256;
constexpr int RED_AREA_BLOCKS = 256;

struct SumParams {
    long n_dx_rows, n_cell_rows, ni;
    const double *dx, *dy, *area;
    long col_a, col_b;
    int want_first, want_last;
    double* partial;  // [RED_AREA_BLOCKS + 4]
    double* out;      // [5]
};

__device__ double block_sum(double v, double* lds) {
    lds[threadIdx.x] = v;
    __syncthreads();
    for (int s = RED_TX / 2; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) lds[threadIdx.x] += lds[threadIdx.x + s];
        __syncthreads();
    }
    const double r = lds[0];
    __syncthreads();
    return r;
}

__global__ __launch_bounds__(RED_TX) void metrics_partial_kernel(SumParams p) {
    __shared__ double lds[RED_TX];
    const int b = blockIdx.x;
    double v = 0.0;
    if (b < RED_AREA_BLOCKS) {
        const long n = p.n_cell_rows * p.ni;
        for (long k = (long)b * RED_TX + threadIdx.x; k < n; k += (long)RED_AREA_BLOCKS * RED_TX) v += p.area[k];
    } else if (b == RED_AREA_BLOCKS || b == RED_AREA_BLOCKS + 1) {
        const long col = (b == RED_AREA_BLOCKS) ? p.col_a : p.col_b;
        if (col >= 0)
            for (long j = threadIdx.x; j < p.n_cell_rows; j += RED_TX) v += p.dy[j * (p.ni + 1) + col];
    } else {
        const bool first = (b == RED_AREA_BLOCKS + 2);
        if ((first ? p.want_first : p.want_last) && p.n_dx_rows > 0) {
            const double* row = p.dx + (first ? 0 : (p.n_dx_rows - 1) * p.ni);
            for (long i = threadIdx.x; i < p.ni; i += RED_TX) v += row[i];
        }
    }
    const double s = block_sum(v, lds);
    if (threadIdx.x == 0) p.partial[b] = s;
}

__global__ __launch_bounds__(RED_TX) void metrics_final_kernel(SumParams p) {
    __shared__ double lds[RED_TX];
    const double s = block_sum(p.partial[threadIdx.x], lds);  // RED_AREA_BLOCKS == RED_TX
    if (threadIdx.x == 0) p.out[0] = s;
    if (threadIdx.x >= 1 && threadIdx.x <= 4) p.out[threadIdx.x] = p.partial[RED_AREA_BLOCKS + threadIdx.x - 1];
}
static_assert(RED_AREA_BLOCKS == RED_TX, "the final kernel reads one partial sum per thread");

}  // namespace

extern "C" int ogg_metrics_sums_dev(long n_dx_rows, long n_cell_rows, long ni, const double* dx, const double* dy, const double* area,
                                    long col_a, long col_b, int want_first_row, int want_last_row, double* out5, void* stream) {
    OGG_REQUIRE(n_dx_rows >= 0 && n_cell_rows >= 0 && ni > 0 && out5, OGG_EARG, "ogg_metrics_sums: bad argument");
    OGG_REQUIRE((n_cell_rows == 0 || (dy && area)) && (n_dx_rows == 0 || dx), OGG_EARG, "ogg_metrics_sums: null field");
    OGG_REQUIRE(col_a >= 0 && col_a <= ni && col_b <= ni, OGG_ESHAPE, "ogg_metrics_sums: column %ld / %ld outside 0..%ld", col_a, col_b, ni);
    hipStream_t st = ogg::as_stream(stream);
    SumParams p{n_dx_rows, n_cell_rows, ni, dx, dy, area, col_a, col_b, want_first_row, want_last_row, nullptr, out5};
    ogg::AsyncScratch scratch(st);   // returned to the stream-ordered allocator on every exit path
    void* partial = nullptr;
    if (int e = scratch.alloc(&partial, (RED_AREA_BLOCKS + 4) * sizeof(double))) return e;
    p.partial = static_cast<double*>(partial);
    metrics_partial_kernel<<<RED_AREA_BLOCKS + 4, RED_TX, 0, st>>>(p);
    OGG_LAUNCH_CHECK();
    metrics_final_kernel<<<1, RED_TX, 0, st>>>(p);
    OGG_LAUNCH_CHECK();
    return OGG_OK;
}
