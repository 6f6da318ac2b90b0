// One rank's whole pass over a tripolar supergrid (the sub-grid loop of OGG:1100-1313) in THREE launches on one stream.
//
// The lat-lon sub-grids are HBM-write bound (48 B/cell, a few waves per CU saturate the write path), the bipolar cap is
// fp64-VALU bound (mesh: asin/atan/atan2 per point; quadrature: (N-1)^2 lattice points per cell).  Instead of putting them
// on separate streams -- which costs 10-20 us of cross-queue signalling per dependency, as much as the kernels themselves
// once the grid is split over 8 GPUs -- each launch carries workgroups of BOTH kinds, told apart by their workgroup index:
//
//   launch A:  row/column tables of the quadrature (a few microseconds)
//   launch B:  lat-lon row strips  |  cap mesh + angle  |  quadrature strips with the guard  |  quadrature strips without
//   launch C:  literal fix-up of the cells the guard handed over  |  j = ny row of the quadrature (literal)
// (without metrics there is no quadrature: launch A then carries the lat-lon strips and the mesh, and B, C do not exist)
//
// The lat-lon workgroups come first in the index space (they are resident from the start and walk their strips grid-stride
// while the compute workgroups stream through the remaining slots).  Every workgroup runs the same body function as the
// stand-alone kernels of ogg_latlon_fused.hip / ogg_bipolar.hip, so the results are bit-identical to the function-level
// entry points.
#include "ogg_bipolar_dev.h"
#include "ogg_latlon_fused_dev.h"

namespace {

constexpr int PASS_TX = 256;
static_assert(PASS_TX == LF_TX && PASS_TX == 64 * MESH_WAVES && PASS_TX == 64 * QS_WAVES, "one workgroup shape for all roles");

struct LatlonShare {
    long n_wg;      // workgroups of this launch that stream lat-lon strips (gx * gy)
    long gx, gy;
    long strip_lo, strip_hi;
};

union PassLds {
    RowScalars ll[LF_ROWS + 1];
    BpRow mesh[MESH_ROWS];
};

struct PassAParams {
    FusedParams ll;
    LatlonShare share;
    MeshParams mesh;
    long mesh_gx, n_mesh;   // mesh workgroups: mesh_gx column tiles x row tiles
    QuadParams q;
    long n_tab;             // workgroups of the tables
};

template <int N>
__global__ __launch_bounds__(PASS_TX) void pass_a_kernel(PassAParams a) {
    __shared__ PassLds lds;
    long b = blockIdx.x;
    if (b < a.share.n_wg) {
        latlon_fused_body(a.ll, lds.ll, b, a.share.gx, a.share.gy, a.share.strip_lo, a.share.strip_hi);
        return;
    }
    b -= a.share.n_wg;
    if (b < a.n_tab) {
        bipolar_tables_body<N>(a.q, b);
        return;
    }
    b -= a.n_tab;
    if (b < a.n_mesh) bipolar_mesh_body<false>(a.mesh, lds.mesh, b % a.mesh_gx, b / a.mesh_gx);
}

struct PassBParams {
    FusedParams ll;
    LatlonShare share;
    MeshParams mesh;        // the cap mesh runs here, next to the quadrature; launch A only builds the tables
    long mesh_gx, n_mesh;
    QuadParams q;
    QuadRange guard, fast;
    long gx, n_guard, n_fast;
};

template <int N>
__global__ __launch_bounds__(PASS_TX) void pass_b_kernel(PassBParams a) {
    __shared__ PassLds lds;
    long b = blockIdx.x;
    if (b < a.share.n_wg) {
        latlon_fused_body(a.ll, lds.ll, b, a.share.gx, a.share.gy, a.share.strip_lo, a.share.strip_hi);
        return;
    }
    b -= a.share.n_wg;
    if (b < a.n_mesh) {
        bipolar_mesh_body<false>(a.mesh, lds.mesh, b % a.mesh_gx, b / a.mesh_gx);
        return;
    }
    b -= a.n_mesh;
    if (b < a.n_guard) {
        bipolar_quad_body<N, QM_GUARD>(a.q, a.guard, (b % a.gx) * QS_WAVES + (threadIdx.x >> 6), b / a.gx);
        return;
    }
    b -= a.n_guard;
    if (b < a.n_fast) bipolar_quad_body<N, QM_FAST>(a.q, a.fast, (b % a.gx) * QS_WAVES + (threadIdx.x >> 6), b / a.gx);
}

long env_long(const char* name, long dflt) {
    const char* e = getenv(name);
    return e ? atol(e) : dflt;
}

LatlonShare make_share(const FusedParams& ll, long lo, long hi, long ni1, bool alone) {
    LatlonShare s{};
    s.gx = latlon_gx(ni1);
    s.strip_lo = lo, s.strip_hi = hi;
    if (hi <= lo) return s;
    // resident lat-lon workgroups: enough to keep the HBM write path busy and not more, so that the VALU-bound workgroups of
    // the same launch get the remaining wave slots; a launch without compute workgroups takes the whole chip
    const long points = (hi - lo) * ll.rows_per_block * ni1;
    // measured optima (1/8 degree, shares 1, 1/2, 1/4, 1/8 = 19.8, 9.9, 5, 2.5 M points): 60, 90, 90-120, 120 -- the smaller the share,
    // the shorter the VALU work the strips can hide behind, so they need more of the write bandwidth
    long max_wg = alone ? 2048
                        : (points >= 16000000 ? env_long("OGG_PASS_LL_WG", 60)
                                              : (points >= 8000000 ? env_long("OGG_PASS_LL_WG_MID", 90) : env_long("OGG_PASS_LL_WG_SMALL", 120)));
    long gy = hi - lo;
    if (s.gx * gy > max_wg) gy = (max_wg + s.gx - 1) / s.gx;
    s.gy = gy < 1 ? 1 : gy;
    s.n_wg = s.gx * s.gy;
    return s;
}

// bytes the lat-lon row strips [lo, hi) write: x, y, angle_dx (ni1 per row), dx (ni), and dy (ni1), area (ni) on cell rows
double latlon_strip_bytes(const FusedParams& ll, long lo, long hi) {
    const long ni1 = ll.ni1, ni = ni1 - 1;
    double bytes = 0.0;
    for (int k = 0; k < ll.n_bands; ++k) {
        const long s0 = ll.strip0[k] > lo ? ll.strip0[k] : lo, s1 = ll.strip0[k + 1] < hi ? ll.strip0[k + 1] : hi;
        if (s1 <= s0) continue;
        const ogg_latlon_band& b = ll.band[k];
        const long r0 = (s0 - ll.strip0[k]) * ll.rows_per_block;
        long r1 = (s1 - ll.strip0[k]) * ll.rows_per_block;
        if (r1 > b.n_pt_rows) r1 = b.n_pt_rows;
        const long ncell = ll.metrics ? b.n_cell_rows : 0;
        const long c1 = r1 < ncell ? r1 : ncell, c0 = r0 < ncell ? r0 : ncell;
        bytes += 8.0 * ((double)(r1 - r0) * (3 * ni1 + (ll.metrics ? ni : 0)) + (double)(c1 - c0) * (ni1 + ni));
    }
    return bytes;
}

template <int N>
int launch_pass(const FusedParams& ll, long ni1, int metrics, const ogg_bipolar_band* cap, hipEvent_t* ev, double* alg_bytes3,
                hipStream_t st) {
    // ev: NULL, or 4 events recorded before launch A and after launches A, B and C (bench.py times the launches with them)
    auto mark = [&](int k) -> int {
        if (ev) OGG_HIP_CHECK(hipEventRecord(ev[k], st));
        return OGG_OK;
    };
    const long n_strips_ll = ll.n_bands ? ll.strip0[ll.n_bands] : 0;
    const bool have_cap = cap && cap->n_pt_rows > 0;
    const bool have_quad = have_cap && metrics;
    PassAParams A{};
    PassBParams B{};
    QuadPlan qp{};
    A.ll = ll, B.ll = ll;
    if (have_cap) {
        A.mesh = MeshParams{cap->Ni, cap->Nj, cap->lat0_bp, cap->lon_bp, cap->j0, cap->n_pt_rows, cap->x, cap->y, nullptr, nullptr, cap->angle, MESH_ROWS};
        const dim3 mg = mesh_grid(A.mesh);
        A.mesh_gx = mg.x, A.n_mesh = (long)mg.x * mg.y;
    }
    if (have_quad) {
        double guard_k = BP_GUARD_K_DEFAULT;
        if (const char* e = getenv("OGG_BP_GUARD_K")) guard_k = atof(e);
        QuadParams p{};
        p.nx = cap->Ni, p.ny = cap->Nj, p.lat0_bp = cap->lat0_bp, p.lon_bp = cap->lon_bp, p.rp = cap->rp, p.Re = cap->Re, p.j0 = cap->j0;
        p.dxq = cap->dx, p.dyq = cap->dy, p.daq = cap->area, p.q = make_nodes(N);
        if (int e = plan_quad<N>(p, cap->n_pt_rows, cap->n_cell_rows, guard_k, cap->workspace, cap->workspace_bytes, qp)) return e;
        A.q = qp.p, B.q = qp.p;
        A.n_tab = tables_blocks<N>(qp.p);
        B.gx = qp.gx;
        B.guard = qp.guard, B.n_guard = qp.has_guard ? (long)qp.gx * qp.guard.gy : 0;
        B.fast = qp.fast, B.n_fast = qp.has_fast ? (long)qp.gx * qp.fast.gy : 0;
    }
    // With metrics, launch A builds only the tables and launch B carries everything else, the cap mesh included: one long launch
    // in which all three kinds of workgroup overlap (measured faster than mesh + part of the lat-lon strips in A at every share
    // from 1 to 1/8 of the 1/8 degree grid).  Without metrics there is no launch B: A carries the lat-lon strips and the mesh.
    const bool launch_b = have_quad && (B.n_guard + B.n_fast > 0);
    const long s1 = launch_b ? 0 : n_strips_ll;   // lat-lon strips [0, s1) in launch A, [s1, S) in launch B
    if (launch_b) {
        B.mesh = A.mesh, B.mesh_gx = A.mesh_gx, B.n_mesh = A.n_mesh;
        A.n_mesh = 0;
    }
    if (alg_bytes3) {  // algorithmic bytes written by each launch (bench.py prices the launches against the HBM roofline)
        const double ni = (double)(ni1 - 1);
        const double mesh_bytes = have_cap ? 8.0 * 3.0 * (double)cap->n_pt_rows * (double)ni1 : 0.0;
        alg_bytes3[0] = latlon_strip_bytes(ll, 0, s1) + (launch_b ? 0.0 : mesh_bytes);
        alg_bytes3[1] = launch_b ? latlon_strip_bytes(ll, s1, n_strips_ll) + 8.0 * (double)cap->n_cell_rows * (3.0 * ni + 1.0) +
                                       mesh_bytes
                                 : 0.0;
        alg_bytes3[2] = (have_quad && qp.has_top) ? 8.0 * ni : 0.0;
    }
    A.share = make_share(ll, 0, s1, ni1, !have_cap);
    const long na = A.share.n_wg + A.n_tab + A.n_mesh;
    if (int e = mark(0)) return e;
    if (na > 0) {
        pass_a_kernel<N><<<(unsigned)na, PASS_TX, 0, st>>>(A);
        OGG_LAUNCH_CHECK();
    }
    if (int e = mark(1)) return e;
    if (launch_b) {
        B.share = make_share(ll, s1, n_strips_ll, ni1, false);
        const unsigned nb = (unsigned)(B.share.n_wg + B.n_mesh + B.n_guard + B.n_fast);
        pass_b_kernel<N><<<nb, PASS_TX, 0, st>>>(B);
        OGG_LAUNCH_CHECK();
    }
    if (int e = mark(2)) return e;
    if (have_quad) {
        if (int e = launch_quad_tail<N>(qp, st)) return e;
    }
    return mark(3);
}

}  // namespace

extern "C" int ogg_tripolar_pass_events_dev(int n_latlon, const ogg_latlon_band* latlon, long ni1, double lon0, double lenlon, double Re,
                                            int metrics, const ogg_bipolar_band* cap, void** events4, double* alg_bytes3, void* stream) {
    FusedParams ll;
    long points = 0;
    if (int e = plan_latlon(n_latlon, latlon, ni1, lon0, lenlon, Re, metrics, ll, points)) return e;
    int order = 5;
    if (cap && cap->n_pt_rows > 0) {
        OGG_REQUIRE(cap->Ni + 1 == ni1, OGG_ESHAPE, "ogg_tripolar_pass: cap has %ld columns, the lat-lon bands %ld", cap->Ni + 1, ni1);
        OGG_REQUIRE(cap->Nj > 0 && cap->j0 >= 0 && cap->j0 + cap->n_pt_rows <= cap->Nj + 1, OGG_ESHAPE,
                    "ogg_tripolar_pass: cap rows %ld..%ld outside 0..%ld", cap->j0, cap->j0 + cap->n_pt_rows, cap->Nj);
        OGG_REQUIRE(cap->x && cap->y && cap->angle, OGG_EARG, "ogg_tripolar_pass: null cap output");
        if (metrics) {
            order = cap->order;
            OGG_REQUIRE(order >= 2 && order <= 5, OGG_EORDER, "Uncoded order");
            OGG_REQUIRE(cap->dx && (cap->n_cell_rows <= 0 || (cap->dy && cap->area)), OGG_EARG, "ogg_tripolar_pass: null cap metrics output");
            OGG_REQUIRE(cap->n_cell_rows >= 0 && cap->j0 + cap->n_cell_rows <= cap->Nj &&
                            (cap->n_pt_rows == cap->n_cell_rows ||
                             (cap->n_pt_rows == cap->n_cell_rows + 1 && cap->j0 + cap->n_cell_rows == cap->Nj)),
                        OGG_ESHAPE, "ogg_tripolar_pass: cap band j0=%ld cell rows=%ld point rows=%ld of Nj=%ld", cap->j0,
                        cap->n_cell_rows, cap->n_pt_rows, cap->Nj);
            OGG_REQUIRE(cap->workspace, OGG_EARG, "ogg_tripolar_pass: the cap needs a workspace (ogg_bipolar_quad_workspace_bytes)");
        }
    }
    hipStream_t st = ogg::as_stream(stream);
    hipEvent_t* ev = reinterpret_cast<hipEvent_t*>(events4);
    switch (order) {
        case 2: return launch_pass<2>(ll, ni1, metrics, cap, ev, alg_bytes3, st);
        case 3: return launch_pass<3>(ll, ni1, metrics, cap, ev, alg_bytes3, st);
        case 4: return launch_pass<4>(ll, ni1, metrics, cap, ev, alg_bytes3, st);
        default: return launch_pass<5>(ll, ni1, metrics, cap, ev, alg_bytes3, st);
    }
}

extern "C" int ogg_tripolar_pass_dev(int n_latlon, const ogg_latlon_band* latlon, long ni1, double lon0, double lenlon, double Re,
                                     int metrics, const ogg_bipolar_band* cap, void* stream) {
    return ogg_tripolar_pass_events_dev(n_latlon, latlon, ni1, lon0, lenlon, Re, metrics, cap, nullptr, nullptr, stream);
}
