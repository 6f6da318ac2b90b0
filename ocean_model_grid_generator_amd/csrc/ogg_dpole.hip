// K5 / K6: displaced-pole Southern cap -- kernels and entry points.
//   displacedPoleCap_projection / _mesh      OGG:447-506   (generate_displaced_pole_grid OGG:509-518)
//   monotonic_bounding                       OGG:470-475
//   great_arc_distance, numerical_hi/hj      OGG:522-562
//   displacedPoleCap_metrics_quad            OGG:565-601
//
// Kernels:
//   dpole_mesh_kernel                    integer mesh rows: lam, phi, exact 360-degree unwrap and angle_x in ONE pass   -> K5
//   dpole_quad_tables / _quad_kernel     Lobatto quadrature of the finite-difference scale factors, nothing in HBM       -> K6
//   dpole_eval_kernel + _unwrap_kernel   the mesh at arbitrary (fractional) index vectors (drop-in displacedPoleCap_mesh)
//   dpole_direct_kernel                  displacedPoleCap_projection on explicit grids / bare monotonic_bounding
//   dpole_h_kernel<F, group>             numerical_hi / numerical_hj at arbitrary index vectors
// The device code of the first two lives in ogg_dpole_dev.h (shared with the fused pass, ogg_pass.hip).
#include <cstdlib>

#include <cstring>

#include "ogg_dpole_dev.h"

namespace ogg {
QuadNodes quad_nodes_host(int order);
}

namespace {

constexpr int SW_TX = 256;

struct SweepParams : DpGeom {
    // lattice of the mesh kernel
    long n_cols;          // columns per row
    const double* i_arr;  // column indices or NULL for 0,1,2,...
    const double* j_arr;  // row indices or NULL for j0, j0+1, ...
    long j0;
    double* out0;         // lam
    double* out1;         // phi
};

// ---- monotonic_bounding (OGG:470-475) as a workgroup-wide scan of 1-bit maps ------------------------------------
template <int NP, int TX = SW_TX>
struct UnwrapShared {
    double v[NP][TX];
    unsigned w0[TX / 64], w1[TX / 64];
};

// Packed "was lowered by 360" states of this thread's column for NP independent scans.  v: raw values of this
// column; first_col: this is column 0 of the row (compared with seed[], no dependence on a previous state);
// carry_v / carry_state_ptr: raw values and states of the column just before this chunk (LDS; the caller rewrites
// them only AFTER this function returns).  Contains two workgroup barriers: every thread of the workgroup must call it
// (inactive threads contribute the identity map).
template <int NP, int TX = SW_TX>
OGG_DEV unsigned unwrap_states(const double* v, bool active, bool first_col, const double* seed, const double* carry_v,
                               const unsigned* carry_state_ptr, UnwrapShared<NP, TX>& sh) {
    constexpr unsigned ALL = (NP >= 32) ? 0xffffffffu : ((1u << NP) - 1u);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
#pragma unroll
    for (int q = 0; q < NP; ++q) sh.v[q][tid] = v[q];
    __syncthreads();
    // The carry of the previous chunk was written after that chunk's second barrier; this read sits behind the
    // barrier above, and the next write sits behind the barrier below: no race in either direction.
    const unsigned carry_state = *carry_state_ptr;
    double cv[NP];
#pragma unroll
    for (int q = 0; q < NP; ++q) cv[q] = carry_v[q];
    unsigned f0 = 0u, f1 = ALL;  // identity map
    if (active) {
        f0 = 0u;
        f1 = 0u;
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            bool b0, b1;
            if (first_col) {
                b0 = b1 = (v[q] - seed[q] > 100);
            } else {
                const double vp = (tid > 0) ? sh.v[q][tid - 1] : cv[q];
                b0 = (v[q] - vp > 100);          // previous column was not lowered
                b1 = (v[q] - (vp - 360) > 100);  // previous column was lowered (x_im1 = vp - 360, OGG:473-474)
            }
            f0 |= (b0 ? 1u : 0u) << q;
            f1 |= (b1 ? 1u : 0u) << q;
        }
    }
    // inclusive wave64 scan of the composition (later o earlier)
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned e0 = __shfl_up(f0, off);
        const unsigned e1 = __shfl_up(f1, off);
        if (lane >= off) {
            const unsigned h0 = (e0 & f1) | (~e0 & f0);
            const unsigned h1 = (e1 & f1) | (~e1 & f0);
            f0 = h0;
            f1 = h1;
        }
    }
    if (lane == 63) {
        sh.w0[wave] = f0;
        sh.w1[wave] = f1;
    }
    __syncthreads();
    unsigned st = carry_state;
    for (int w = 0; w < wave; ++w) st = (st & sh.w1[w]) | (~st & sh.w0[w]);
    return ((st & f1) | (~st & f0)) & ALL;
}

// ---- mesh (OGG:488-518) in two launches ---------------------------------------------------------------------------------
// The unwrap of a row is a sequential chain over its chunks, so a kernel that evaluates the projection inside that chain takes
// n_cols / TX projection latencies per row whatever the number of rows -- a band of 36 rows (one rank's share on 8 GPUs) kept
// 36 CUs busy for 28 us.  Hence: (1) every point's raw longitude and latitude, fully parallel (one workgroup per 256 columns
// of a row); (2) the unwrap alone, one workgroup of 1024 threads per row reading the raw longitudes back (L2) and lowering them
// in place.  Same arithmetic per point, same scan: the same bits as the one-kernel form.
__global__ __launch_bounds__(SW_TX) void dpole_eval_kernel(SweepParams p) {
    __shared__ DpConst s_c;
    __shared__ double s_r;
    const int tid = threadIdx.x;
    const long row = blockIdx.y;
    if (tid == 0) {
        s_c = dp_const(p);
        const double jv = p.j_arr ? p.j_arr[row] : (double)(p.j0 + row);
        s_r = dp_row_radius(jv, p, s_c);                                   // row-only
    }
    __syncthreads();
    const long g = (long)blockIdx.x * SW_TX + tid;
    if (g >= p.n_cols) return;
    const DpConst c = s_c;
    const double iv = p.i_arr ? p.i_arr[g] : (double)g;
    double lam_raw, ph;
    dp_point(s_r, dp_column(iv, p, c), c, lam_raw, ph);
    p.out0[row * p.n_cols + g] = lam_raw;
    p.out1[row * p.n_cols + g] = ph;
}

// Unwrap of one row by one workgroup in ONE step: thread t owns the CPT consecutive columns t*CPT .. and composes their
// 1-bit maps sequentially (both incoming states), a wave64 shuffle scan and a scan over the waves give every thread its
// incoming state, and a second sequential walk lowers its columns.  Two barriers per row instead of two per 1024 columns.
template <int TX>
__global__ __launch_bounds__(TX) void dpole_unwrap_kernel(SweepParams p, int cpt) {
    __shared__ unsigned s_w0[TX / 64], s_w1[TX / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long row = blockIdx.x;
    const double i_first = p.i_arr ? p.i_arr[0] : 0.0;
    const double seed = p.lon0 + (i_first * 360.0) / (double)p.ni;         // lon_grid[0,0] (OGG:463)
    double* lam = p.out0 + row * p.n_cols;
    const long g0 = (long)tid * cpt;
    const long g1 = (g0 + cpt < p.n_cols) ? g0 + cpt : p.n_cols;
    // f0 / f1: state after this thread's columns when the column before them was not / was lowered (identity if it owns none)
    unsigned f0 = 0u, f1 = 1u;
    if (g0 < g1) {
        unsigned s0 = 0u, s1 = 1u;
        double prev = (g0 > 0) ? lam[g0 - 1] : 0.0;
        for (long g = g0; g < g1; ++g) {
            const double v = lam[g];
            if (g == 0) {
                s0 = s1 = (v - seed > 100) ? 1u : 0u;                      // OGG:471-472: compared with x_0, no incoming state
            } else {
                s0 = (v - (s0 ? prev - 360 : prev) > 100) ? 1u : 0u;       // OGG:473-474: x_im1 is the ADJUSTED previous column
                s1 = (v - (s1 ? prev - 360 : prev) > 100) ? 1u : 0u;
            }
            prev = v;
        }
        f0 = s0, f1 = s1;
    }
    // inclusive scan of the composition (later o earlier) over the wave, then over the waves
    unsigned c0 = f0, c1 = f1;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned e0 = __shfl_up(c0, off), e1 = __shfl_up(c1, off);
        if (lane >= off) {
            const unsigned h0 = e0 ? c1 : c0, h1 = e1 ? c1 : c0;
            c0 = h0, c1 = h1;
        }
    }
    if (lane == 63) s_w0[wave] = c0, s_w1[wave] = c1;
    __syncthreads();
    unsigned st = 0u;  // state of the column before this WAVE's first column (row start: no previous column, value unused)
    for (int w = 0; w < wave; ++w) st = st ? s_w1[w] : s_w0[w];
    // exclusive within the wave: state after the previous lane's columns
    const unsigned p0 = __shfl_up(c0, 1), p1 = __shfl_up(c1, 1);
    unsigned s_in = (lane == 0) ? st : (st ? p1 : p0);
    const double prev0 = (g0 > 0 && g0 < g1) ? lam[g0 - 1] : 0.0;
    __syncthreads();   // every thread has read its left neighbour's raw value before anybody lowers one (uniform barrier)
    if (g0 < g1) {
        double prev = prev0;
        unsigned sp = s_in;
        for (long g = g0; g < g1; ++g) {
            const double v = lam[g];
            unsigned sc;
            if (g == 0)
                sc = (v - seed > 100) ? 1u : 0u;
            else
                sc = (v - (sp ? prev - 360 : prev) > 100) ? 1u : 0u;
            if (sc) lam[g] = v - 360;                                      // OGG:473
            prev = v;
            sp = sc;
        }
    }
}

// ---- finite-difference scale factors on a lattice row (OGG:535-562), one workgroup per lattice row and probe group ----
// JGROUP = false: h_i from the 2H probes (j, i +- m eps);  JGROUP = true: h_j from the probes (j +- m eps, i).
// The two groups of a lattice row are independent scans, so they run as separate workgroups: twice the parallelism and
// half the live registers of a kernel that carries all 4H probes.
struct HParams {
    long ni, nj;
    double lon0, lat0, lam_pole, r_pole, eps;
    long n_cols, n_rows;
    const double* i_arr;  // explicit column indices, or NULL: Lobatto nodes of cells 0.. (lattice_M unique nodes per cell)
    const double* j_arr;  // explicit row indices, or NULL: Lobatto nodes of cell rows row0_cell..
    int lattice_M;
    long row0_cell;
    QuadNodes q;
    double* h_i;          // [n_rows][n_cols] or NULL
    double* h_j;
};

template <int F, bool JGROUP>
__global__ __launch_bounds__(SW_TX) void dpole_h_kernel(HParams p) {
    constexpr int H = F / 2;
    __shared__ UnwrapShared<F> s_u;
    __shared__ double s_carry_v[F];
    __shared__ unsigned s_carry_state;
    __shared__ double s_r[F + 1];
    const int tid = threadIdx.x;
    const long row = blockIdx.x;
    double* out = JGROUP ? p.h_j : p.h_i;
    const DpGeom sp{p.ni, p.nj, p.lon0, p.lat0, p.lam_pole, p.r_pole};
    const DpConst c = dp_const(sp);
    const double reps = 1.0 / p.eps;
    const int M = p.lattice_M;
    if (tid <= F) {  // s_r[0]: base row; s_r[2m-1]: j + m eps; s_r[2m]: j - m eps
        double jv = p.j_arr ? p.j_arr[row] : lattice_node(p.q, (int)(row % M), p.row0_cell + row / M);
        if (tid > 0) {
            const double off = (double)((tid + 1) / 2) * p.eps;
            jv = (tid & 1) ? jv + off : jv - off;
        }
        if (tid == 0 || JGROUP) s_r[tid] = dp_row_radius(jv, sp, c);
    }
    const double i_first = p.i_arr ? p.i_arr[0] : lattice_node(p.q, 0, 0);
    __syncthreads();
    for (long c0 = 0; c0 < p.n_cols; c0 += SW_TX) {
        const long g = c0 + tid;
        const bool active = g < p.n_cols;
        double v[F], ph[F], seed[F];
#pragma unroll
        for (int q = 0; q < F; ++q) v[q] = ph[q] = seed[q] = 0.0;
        if (active) {
            const double iv = p.i_arr ? p.i_arr[g] : lattice_node(p.q, (int)(g % M), g / M);
            if (JGROUP) {
                const cplx ep = dp_column(iv, sp, c);
#pragma unroll
                for (int q = 0; q < F; ++q) dp_point(s_r[q + 1], ep, c, v[q], ph[q]);
            } else {
#pragma unroll
                for (int m = 1; m <= H; ++m) {
                    const double off = (double)m * p.eps;
                    dp_point(s_r[0], dp_column(iv + off, sp, c), c, v[2 * (m - 1)], ph[2 * (m - 1)]);
                    dp_point(s_r[0], dp_column(iv - off, sp, c), c, v[2 * (m - 1) + 1], ph[2 * (m - 1) + 1]);
                }
            }
        }
#pragma unroll
        for (int q = 0; q < F; ++q) {  // seed lon_grid[0,0] of each probe's mesh (OGG:463)
            double di = 0.0;
            if (!JGROUP) {
                const double off = (double)(q / 2 + 1) * p.eps;
                di = (q & 1) ? -off : off;
            }
            seed[q] = p.lon0 + ((i_first + di) * 360.0) / (double)p.ni;
        }
        const unsigned st = unwrap_states<F>(v, active, g == 0, seed, s_carry_v, &s_carry_state, s_u);
        if (active && tid == SW_TX - 1) {
#pragma unroll
            for (int q = 0; q < F; ++q) s_carry_v[q] = v[q];
            s_carry_state = st;
        }
        if (active) {
            double ds[(H > 0) ? H : 1];
#pragma unroll
            for (int m = 0; m < H; ++m) {
                const double x0 = ((st >> (2 * m)) & 1u) ? v[2 * m] - 360 : v[2 * m];              // OGG:473
                const double x1 = ((st >> (2 * m + 1)) & 1u) ? v[2 * m + 1] - 360 : v[2 * m + 1];
                ds[m] = great_arc(x0, ph[2 * m], x1, ph[2 * m + 1]);
            }
            out[row * p.n_cols + g] = central_difference<F>(ds, reps);
        }
    }
}


// ---- displacedPoleCap_projection on explicit 2-D lon/lat grids (OGG:447-467) and bare monotonic_bounding ---------
struct DirectParams {
    long nj, ni;
    const double* lon;  // PROJECT: lon_grid; UNWRAP: unused
    const double* lat;
    double z0r, z0i, r_joint, x0;
    double* lam;        // PROJECT: out; UNWRAP: in/out
    double* phi;
};

template <bool PROJECT>
__global__ __launch_bounds__(SW_TX) void dpole_direct_kernel(DirectParams p) {
    __shared__ UnwrapShared<1> s_u;
    __shared__ double s_carry_v[1];
    __shared__ unsigned s_carry_state;
    const int tid = threadIdx.x;
    const long row = blockIdx.x;
    const DpConst c = {p.z0r, p.z0i, p.r_joint};
    for (long c0 = 0; c0 < p.ni; c0 += SW_TX) {
        const long g = c0 + tid;
        const bool active = g < p.ni;
        double v[1] = {0.0}, ph = 0.0;
        if (active) {
            if (PROJECT) {
                const double lon = p.lon[row * p.ni + g], lat = p.lat[row * p.ni + g];
                const double r = tan((90 + lat) * kPi180) / p.r_joint;  // OGG:448
                double s, co;
                sincos(lon * kPi180, &s, &co);
                const cplx e = {co, s};
                const cplx num = {e.re - c.z0r, e.im - c.z0i};
                const cplx cz = cmul(cplx{c.z0r, -c.z0i}, e);
                const cplx den = {1.0 - cz.re, 0.0 - cz.im};
                dp_point(r, cdiv(num, den), c, v[0], ph);
            } else {
                v[0] = p.lam[row * p.ni + g];
            }
        }
        const double seed[1] = {p.x0};
        const unsigned st = unwrap_states<1>(v, active, g == 0, seed, s_carry_v, &s_carry_state, s_u);
        if (active && tid == SW_TX - 1) {
            s_carry_v[0] = v[0];
            s_carry_state = st;
        }
        if (active) {
            p.lam[row * p.ni + g] = (st & 1u) ? v[0] - 360 : v[0];
            if (PROJECT) p.phi[row * p.ni + g] = ph;
        }
    }
}

// ---- kernels around the bodies of ogg_dpole_dev.h -------------------------------------------------------------------------
template <int N>
__global__ __launch_bounds__(256) void dpole_quad_tables_kernel(DpQuadParams p) {
    dpole_quad_tables_body<N>(p, blockIdx.x, gridDim.x);
}

template <int N, int ARC>
__global__ __launch_bounds__(64 * DQ_WAVES, 2) void dpole_quad_kernel(DpQuadParams p) {
    __shared__ unsigned s_slot;
    const long t = (ARC == DP_ARC_LITERAL) ? take_ticket(p.ticket, &s_slot) : (long)blockIdx.x;
    if (ARC == DP_ARC_LITERAL) {
        __shared__ double ring[(ARC == DP_ARC_LITERAL) ? dq_ring_doubles<N>() : 1];
        dpole_quad_literal_ring<N, DQ_RING>(p, (t % p.gx) * DQ_WAVES + (threadIdx.x >> 6), t / p.gx,
                                            ring + (threadIdx.x >> 6) * (DQ_RING * dq_ring_slot_doubles<N>()));
        return;
    }
    dpole_quad_body<N, ARC>(p, (t % p.gx) * DQ_WAVES + (threadIdx.x >> 6), t / p.gx);
}

// the register-pipelined walk of the literal form (dpole_quad_body: one row of slack in registers, the library's own atan2 / atan, the
// block-by-block look-back): an independently written second implementation, kept for OGG_DQ_WALK=regs -- the tests require the
// LDS-pipelined walk to reproduce its bits
template <int N>
__global__ __launch_bounds__(64 * DQ_WAVES, 2) void dpole_quad_regs_kernel(DpQuadParams p) {
    __shared__ unsigned s_slot;
    const long t = take_ticket(p.ticket, &s_slot);
    dpole_quad_body<N, DP_ARC_LITERAL>(p, (t % p.gx) * DQ_WAVES + (threadIdx.x >> 6), t / p.gx);
}

// error flag of a workspace the library allocated itself (it is returned to the stream-ordered allocator when the call returns, so the
// caller could never ask): read back behind the kernels; synchronises the stream
int check_own_workspace(const void* ws, hipStream_t s, const char* what) {
    unsigned v = 0u;
    OGG_HIP_CHECK(hipMemcpyAsync(&v, static_cast<const unsigned*>(ws) + 1, sizeof(v), hipMemcpyDeviceToHost, s));
    OGG_HIP_CHECK(hipStreamSynchronize(s));
    OGG_REQUIRE(v == 0u, OGG_EHIP, "%s: a look-back wait timed out (flag %u); results are invalid", what, v);
    return OGG_OK;
}

__global__ __launch_bounds__(256) void dpole_mesh_reset_kernel(DpMeshParams m) { dpole_mesh_reset_body(m, blockIdx.x, gridDim.x); }

__global__ __launch_bounds__(64 * DM_WAVES) void dpole_mesh_kernel(DpMeshParams m) {
    __shared__ DpMeshLds lds;
    __shared__ unsigned s_slot;
    const long t = take_ticket(m.ticket, &s_slot);
    dpole_mesh_body(m, lds, t % m.gx, t / m.gx);
}

int launch_mesh(const SweepParams& p, long rows, hipStream_t s) {
    if (rows <= 0) return OGG_OK;
    dpole_eval_kernel<<<dim3((unsigned)((p.n_cols + SW_TX - 1) / SW_TX), (unsigned)rows), SW_TX, 0, s>>>(p);
    OGG_LAUNCH_CHECK();
    const int cpt = (int)((p.n_cols + 1023) / 1024);
    dpole_unwrap_kernel<1024><<<(unsigned)rows, 1024, 0, s>>>(p, cpt);
    OGG_LAUNCH_CHECK();
    return OGG_OK;
}

int check_cap(long ni, long nj) {
    OGG_REQUIRE(ni > 0 && nj > 0, OGG_ESHAPE, "displaced pole cap: ni=%ld nj=%ld", ni, nj);
    return OGG_OK;
}

}  // namespace

extern "C" {

int ogg_displaced_pole_mesh_dev(long n_i, const double* i, long n_j, const double* j, long ni, long nj, double lon0,
                                double lat0, double lam_pole, double r_pole, double* lams, double* phis, void* stream) {
    if (int e = check_cap(ni, nj)) return e;
    OGG_REQUIRE(n_i > 0 && n_j >= 0 && i && j && lams && phis, OGG_EARG, "ogg_displaced_pole_mesh: bad argument");
    SweepParams p{};
    p.ni = ni, p.nj = nj, p.lon0 = lon0, p.lat0 = lat0, p.lam_pole = lam_pole, p.r_pole = r_pole;
    p.n_cols = n_i, p.i_arr = i, p.j_arr = j;
    p.out0 = lams, p.out1 = phis;
    return launch_mesh(p, n_j, ogg::as_stream(stream));
}

int ogg_displaced_pole_projection_dev(long nj, long ni, const double* lon_grid, const double* lat_grid, double z0_re,
                                      double z0_im, double r_joint, double x_0, double* lam, double* phi, void* stream) {
    OGG_REQUIRE(nj >= 0 && ni > 0 && lon_grid && lat_grid && lam && phi, OGG_EARG, "ogg_displaced_pole_projection: bad argument");
    if (nj == 0) return OGG_OK;
    DirectParams p{nj, ni, lon_grid, lat_grid, z0_re, z0_im, r_joint, x_0, lam, phi};
    dpole_direct_kernel<true><<<(unsigned)nj, SW_TX, 0, ogg::as_stream(stream)>>>(p);
    OGG_LAUNCH_CHECK();
    return OGG_OK;
}

int ogg_monotonic_bounding_dev(long nj, long ni, double* x, double x_0, void* stream) {
    OGG_REQUIRE(nj >= 0 && ni > 0 && x, OGG_EARG, "ogg_monotonic_bounding: bad argument");
    if (nj == 0) return OGG_OK;
    DirectParams p{nj, ni, nullptr, nullptr, 0.0, 0.0, 1.0, x_0, x, nullptr};
    dpole_direct_kernel<false><<<(unsigned)nj, SW_TX, 0, ogg::as_stream(stream)>>>(p);
    OGG_LAUNCH_CHECK();
    return OGG_OK;
}

long ogg_displaced_pole_grid_workspace_bytes(long Ni, long nrows) {
    if (Ni <= 0 || nrows < 0) return 0;
    return (long)dm_workspace_bytes(Ni, nrows);
}

int ogg_displaced_pole_grid_angle_ws_dev(long Ni, long Nj, double lon0, double lat0, double lon_dp, double r_dp, long j0, long nrows,
                                         double* x, double* y, double* angle_dx, void* workspace, long workspace_bytes, void* stream) {
    if (int e = check_cap(Ni, Nj)) return e;
    OGG_REQUIRE(j0 >= 0 && nrows >= 0 && j0 + nrows <= Nj + 1 && x && y, OGG_ESHAPE,
                "ogg_displaced_pole_grid: rows %ld..%ld outside 0..%ld", j0, j0 + nrows, Nj);
    if (nrows == 0) return OGG_OK;
    hipStream_t s = ogg::as_stream(stream);
    ogg::AsyncScratch own(s);
    void* ws = workspace;
    long ws_bytes = workspace_bytes;
    if (!ws) {
        ws_bytes = (long)dm_workspace_bytes(Ni, nrows);
        if (int e = own.alloc(&ws, (size_t)ws_bytes)) return e;
    }
    DpMeshParams m{};
    if (int e = plan_dmesh(DpGeom{Ni, Nj, lon0, lat0, lon_dp, r_dp}, j0, nrows, x, y, angle_dx, ws, ws_bytes, m)) return e;
    dpole_mesh_reset_kernel<<<(unsigned)dm_reset_blocks(m), 256, 0, s>>>(m);
    OGG_LAUNCH_CHECK();
    dpole_mesh_kernel<<<(unsigned)dm_blocks(m), 64 * DM_WAVES, 0, s>>>(m);
    OGG_LAUNCH_CHECK();
    if (!workspace) return check_own_workspace(ws, s, "ogg_displaced_pole_grid");
    return OGG_OK;
}

int ogg_displaced_pole_grid_dev(long Ni, long Nj, double lon0, double lat0, double lon_dp, double r_dp, long j0, long nrows,
                                double* x, double* y, void* stream) {
    return ogg_displaced_pole_grid_angle_ws_dev(Ni, Nj, lon0, lat0, lon_dp, r_dp, j0, nrows, x, y, nullptr, nullptr, 0, stream);
}

int ogg_displaced_pole_numerical_h_dev(long n_i, const double* i, long n_j, const double* j, long nx, long ny, double lon0,
                                       double lat0, double lon_dp, double r_dp, double eps, int fd_order, double* h_i,
                                       double* h_j, void* stream) {
    if (int e = check_cap(nx, ny)) return e;
    OGG_REQUIRE(fd_order == 2 || fd_order == 4 || fd_order == 6, OGG_EORDER, "order not coded");
    OGG_REQUIRE(n_i > 0 && n_j >= 0 && i && j && (h_i || h_j) && eps > 0, OGG_EARG, "ogg_displaced_pole_numerical_h: bad argument");
    if (n_j == 0) return OGG_OK;
    HParams p{};
    p.ni = nx, p.nj = ny, p.lon0 = lon0, p.lat0 = lat0, p.lam_pole = lon_dp, p.r_pole = r_dp, p.eps = eps;
    p.n_cols = n_i, p.n_rows = n_j, p.i_arr = i, p.j_arr = j, p.lattice_M = 1;
    p.h_i = h_i, p.h_j = h_j;
    hipStream_t s = ogg::as_stream(stream);
#define OGG_LAUNCH_H(F)                                                                      \
    do {                                                                                     \
        if (h_i) dpole_h_kernel<F, false><<<(unsigned)n_j, SW_TX, 0, s>>>(p);                \
        if (h_j) dpole_h_kernel<F, true><<<(unsigned)n_j, SW_TX, 0, s>>>(p);                 \
    } while (0)
    if (fd_order == 2)
        OGG_LAUNCH_H(2);
    else if (fd_order == 4)
        OGG_LAUNCH_H(4);
    else
        OGG_LAUNCH_H(6);
    OGG_LAUNCH_CHECK();
    return OGG_OK;
}

long ogg_displaced_pole_quad_workspace_bytes(int order, long nx, long n_cell_rows) {
    if ((order != 2 && order != 4) || nx <= 0 || n_cell_rows < 0) return 0;
    return (long)dq_workspace_bytes(order, nx, n_cell_rows);
}

int ogg_workspace_error_flag_dev(const void* workspace, int* flag, void* stream) {
    OGG_REQUIRE(workspace && flag, OGG_EARG, "ogg_workspace_error_flag: null pointer");
    hipStream_t s = ogg::as_stream(stream);
    unsigned v = 0u;
    OGG_HIP_CHECK(hipMemcpyAsync(&v, static_cast<const unsigned*>(workspace) + 1, sizeof(v), hipMemcpyDeviceToHost, s));
    OGG_HIP_CHECK(hipStreamSynchronize(s));
    *flag = (int)v;
    return OGG_OK;
}

int ogg_displaced_pole_metrics_quad_form_sym_ws_dev(int arc_form, int symmetry, int order, long nx, long ny, double lon0, double lat0,
                                                    double lon_dp, double r_dp, double Re, long j0, long n_dx_rows, long n_cell_rows,
                                                    double* dxq, double* dyq, double* daq, void* workspace, long workspace_bytes,
                                                    void* stream) {
    OGG_REQUIRE(order >= 2 && order <= 5, OGG_EORDER, "Uncoded order");
    // the quadrature order is forwarded as the finite-difference order (OGG:583-584): 3 and 5 are "not coded" there
    OGG_REQUIRE(order == 2 || order == 4, OGG_EORDER, "order not coded");
    OGG_REQUIRE(arc_form == OGG_DP_ARC_LITERAL || arc_form == OGG_DP_ARC_CHORD, OGG_EARG, "displaced-pole quadrature: arc_form %d", arc_form);
    if (int e = check_cap(nx, ny)) return e;
    OGG_REQUIRE(dxq && (n_cell_rows <= 0 || (dyq && daq)), OGG_EARG, "ogg_displaced_pole_metrics_quad: null output");
    OGG_REQUIRE(j0 >= 0 && n_cell_rows >= 0 && j0 + n_cell_rows <= ny &&
                    (n_dx_rows == n_cell_rows || (n_dx_rows == n_cell_rows + 1 && j0 + n_cell_rows == ny)),
                OGG_ESHAPE, "ogg_displaced_pole_metrics_quad: band j0=%ld cell rows=%ld dx rows=%ld of ny=%ld", j0,
                n_cell_rows, n_dx_rows, ny);
    if (n_dx_rows == 0) return OGG_OK;
    hipStream_t s = ogg::as_stream(stream);
    ogg::AsyncScratch own(s);
    void* ws = workspace;
    long ws_bytes = workspace_bytes;
    if (!ws) {
        ws_bytes = (long)dq_workspace_bytes(order, nx, n_cell_rows);
        if (int e = own.alloc(&ws, (size_t)ws_bytes)) return e;
    }
    DpQuadParams p{};
    if (int e = plan_dquad(arc_form, order, DpGeom{nx, ny, lon0, lat0, lon_dp, r_dp}, Re, j0, n_dx_rows, n_cell_rows, dxq, dyq, daq, ws,
                           ws_bytes, ogg::quad_nodes_host(order), p, ogg::cap_symmetry(symmetry) ? 1 : 0))
        return e;
    const unsigned nwg = (unsigned)(p.gx * p.n_chunks);
    const char* walk = getenv("OGG_DQ_WALK");
    const bool regs = walk && strcmp(walk, "regs") == 0;   // the second implementation of the literal walk (tests)
    if (order == 2) {
        dpole_quad_tables_kernel<2><<<(unsigned)dpole_quad_tables_blocks<2>(p), 256, 0, s>>>(p);
        if (arc_form == OGG_DP_ARC_CHORD)
            dpole_quad_kernel<2, DP_ARC_CHORD><<<nwg, 64 * DQ_WAVES, 0, s>>>(p);
        else if (regs)
            dpole_quad_regs_kernel<2><<<nwg, 64 * DQ_WAVES, 0, s>>>(p);
        else
            dpole_quad_kernel<2, DP_ARC_LITERAL><<<nwg, 64 * DQ_WAVES, 0, s>>>(p);
    } else {
        dpole_quad_tables_kernel<4><<<(unsigned)dpole_quad_tables_blocks<4>(p), 256, 0, s>>>(p);
        if (arc_form == OGG_DP_ARC_CHORD)
            dpole_quad_kernel<4, DP_ARC_CHORD><<<nwg, 64 * DQ_WAVES, 0, s>>>(p);
        else if (regs)
            dpole_quad_regs_kernel<4><<<nwg, 64 * DQ_WAVES, 0, s>>>(p);
        else
            dpole_quad_kernel<4, DP_ARC_LITERAL><<<nwg, 64 * DQ_WAVES, 0, s>>>(p);
    }
    OGG_LAUNCH_CHECK();
    if (!workspace && arc_form == OGG_DP_ARC_LITERAL) return check_own_workspace(ws, s, "ogg_displaced_pole_metrics_quad");
    return OGG_OK;
}

int ogg_displaced_pole_metrics_quad_form_ws_dev(int arc_form, int order, long nx, long ny, double lon0, double lat0, double lon_dp,
                                                double r_dp, double Re, long j0, long n_dx_rows, long n_cell_rows, double* dxq,
                                                double* dyq, double* daq, void* workspace, long workspace_bytes, void* stream) {
    return ogg_displaced_pole_metrics_quad_form_sym_ws_dev(arc_form, OGG_SYM_DEFAULT, order, nx, ny, lon0, lat0, lon_dp, r_dp, Re, j0,
                                                           n_dx_rows, n_cell_rows, dxq, dyq, daq, workspace, workspace_bytes, stream);
}

int ogg_displaced_pole_metrics_quad_ws_dev(int order, long nx, long ny, double lon0, double lat0, double lon_dp, double r_dp,
                                           double Re, long j0, long n_dx_rows, long n_cell_rows, double* dxq, double* dyq,
                                           double* daq, void* workspace, long workspace_bytes, void* stream) {
    return ogg_displaced_pole_metrics_quad_form_ws_dev(OGG_DP_ARC_CHORD, order, nx, ny, lon0, lat0, lon_dp, r_dp, Re, j0, n_dx_rows,
                                                       n_cell_rows, dxq, dyq, daq, workspace, workspace_bytes, stream);
}

int ogg_displaced_pole_metrics_quad_dev(int order, long nx, long ny, double lon0, double lat0, double lon_dp, double r_dp,
                                        double Re, long j0, long n_dx_rows, long n_cell_rows, double* dxq, double* dyq,
                                        double* daq, void* stream) {
    return ogg_displaced_pole_metrics_quad_form_ws_dev(OGG_DP_ARC_CHORD, order, nx, ny, lon0, lat0, lon_dp, r_dp, Re, j0, n_dx_rows,
                                                       n_cell_rows, dxq, dyq, daq, nullptr, 0, stream);
}

}  // extern "C"
