// One rank's whole pass over a tripolar supergrid (the sub-grid loop of OGG:1100-1313) in THREE launches on one stream (four
// when a displaced-pole southern cap asks for the literal arc form, whose quadrature kernel needs more registers than the
// workgroups of launch B should be held to).
//
// The lat-lon sub-grids are HBM-write bound (48 B/cell, a few waves per CU saturate the write path), the bipolar cap is
// fp64-VALU bound (mesh: asin/atan/atan2 per point; quadrature: (N-1)^2 lattice points per cell).  Instead of putting them
// on separate streams -- which costs 10-20 us of cross-queue signalling per dependency, as much as the kernels themselves
// once the grid is split over 8 GPUs -- each launch carries workgroups of BOTH kinds, told apart by their workgroup index:
//
//   launch A:  row/column tables of the quadratures (both caps), reset of the displaced-pole look-back words (a few microseconds)
//   launch B:  lat-lon row strips  |  bipolar quadrature strips without the guard  |  ... with  |  displaced-pole quadrature strips
//              (chord form)  |  bipolar mesh + angle  |  displaced-pole mesh + angle
//   launch C:  literal fix-up of the bipolar cells the guard handed over  |  j = ny row of the bipolar quadrature (literal)
//   launch D:  displaced-pole quadrature strips, literal form (only when asked for)
// (without metrics and without a displaced-pole cap there is no quadrature: launch A then carries the lat-lon strips and the
// bipolar mesh, and B, C do not exist)
//
// The tables of the NEXT pass ride in launch B (PassPipe below).  With metrics, launch A writes nothing but the cap workspaces (tables,
// cleared look-back words and counters), and what it writes does not depend on the pass before it: a plan handle keeps TWO workspaces
// per cap (its own allocations: their contents outlive a run) and lets the last workgroups of launch B(k) do launch A's work for pass k + 1 in the other one -- they fill the slots the
// draining strips leave.  Pass k + 1 then starts with launch B: one packet less per pass on the stream (3-4.5 us: whatever follows
// a kernel on this runtime costs that much, scripts/microbench/stream_overlap.hip), no second stream, no flag, no wait -- the stream's own
// order is the dependence.  Every pass still builds one set of tables; the first pass of a plan (and a pass that times its launches)
// runs launch A itself.  The one OUTPUT launch A writes, the j = ny row of the bipolar dx, goes through the workspace and is copied by
// the tail launch (QuadParams::top_src), so that nothing of pass k + 1 reaches an output array during pass k.
//
// The lat-lon workgroups come first in the index space (they are resident from the start and walk their strips grid-stride
// while the compute workgroups stream through the remaining slots).  Every workgroup runs the same body function as the
// stand-alone kernels of ogg_latlon_fused.hip / ogg_bipolar.hip, so the results are bit-identical to the function-level
// entry points.
#include <cstring>
#include <new>
#include <vector>

#include "ogg_bipolar_dev.h"
#include "ogg_dpole_dev.h"
#include "ogg_latlon_fused_dev.h"

extern "C" long ogg_dpole_band_workspace_bytes(int order, long Ni, long n_pt_rows);

namespace {

constexpr int PASS_TX = 256;
static_assert(PASS_TX == LF_TX && PASS_TX == 64 * MESH_WAVES && PASS_TX == 64 * QS_WAVES && PASS_TX == 64 * DM_WAVES &&
                  PASS_TX == 64 * DQ_WAVES,
              "one workgroup shape for all roles");

struct LatlonShare {
    long n_wg;      // workgroups of this launch that stream lat-lon strips (gx * gy)
    long gx, gy;
    long strip_lo, strip_hi;
    int nt;         // non-temporal stores (ogg_latlon_fused_dev.h, store2): when the strips share the launch with cap workgroups
    unsigned* claims;   // claim counters of the strips (2 per resident workgroup; zeroed by launch A), or NULL: every workgroup its block
    long n_help;        // helper workgroups at the END of the launch (claims != NULL): n_wg * OGG_PASS_LL_HELPERS
    int pool;           // claims != NULL: one counter per column tile, every strip of the tile handed out from it (no helpers)
    long points;        // lat-lon points of the strips (host side: size class of the launch)
};

union PassLds {
    RowScalars ll[LF_ROWS + 1];
    BpRow mesh[MESH_ROWS_MAX];
    DpMeshLds dmesh;
    QuadOutLds quad;
};

struct PassAParams {
    FusedParams ll;
    LatlonShare share;
    MeshParams mesh;
    long mesh_gx, n_mesh;   // mesh workgroups: mesh_gx column tiles x row tiles
    QuadParams q;
    long n_tab;             // workgroups of the bipolar tables
    DpQuadParams dq;        // displaced-pole cap: tables of its quadrature ...
    long n_dq_tab;
    int dq_order;
    DpMeshParams dm;        // ... and the reset of the look-back words of its mesh
    long n_dm_reset;
    long n_ll_tab;          // workgroups of the lat-lon row table (ll.row_tab; 0: the strips of launch B evaluate their rows themselves)
};

// the roles of launch A that write nothing but the cap workspaces (tables, cleared words and counters)
template <int N>
OGG_DEV void pass_table_roles(const PassAParams& a, long b) {
    if (b < a.n_tab) {
        bipolar_tables_body<N>(a.q, b);
        return;
    }
    b -= a.n_tab;
    if (b < a.n_dq_tab) {
        if (a.dq_order == 2)
            dpole_quad_tables_body<2>(a.dq, b, a.n_dq_tab);
        else
            dpole_quad_tables_body<4>(a.dq, b, a.n_dq_tab);
        return;
    }
    b -= a.n_dq_tab;
    if (b < a.n_dm_reset) {
        dpole_mesh_reset_body(a.dm, b, a.n_dm_reset);
        return;
    }
    b -= a.n_dm_reset;
    if (b < a.n_ll_tab) latlon_row_table_body(a.ll, b, a.n_ll_tab);
}

template <int N>
__global__ __launch_bounds__(PASS_TX) void pass_a_kernel(PassAParams a) {
    __shared__ PassLds lds;
    long b = blockIdx.x;
    if (b < a.share.n_wg) {
        if (a.share.nt)
            latlon_fused_body<true>(a.ll, lds.ll, b, a.share.gx, a.share.gy, a.share.strip_lo, a.share.strip_hi);
        else
            latlon_fused_body<false>(a.ll, lds.ll, b, a.share.gx, a.share.gy, a.share.strip_lo, a.share.strip_hi);
        return;
    }
    b -= a.share.n_wg;
    if (b < a.n_tab) {
        bipolar_tables_body<N>(a.q, b);
        return;
    }
    b -= a.n_tab;
    if (b < a.n_dq_tab) {
        if (a.dq_order == 2)
            dpole_quad_tables_body<2>(a.dq, b, a.n_dq_tab);
        else
            dpole_quad_tables_body<4>(a.dq, b, a.n_dq_tab);
        return;
    }
    b -= a.n_dq_tab;
    if (b < a.n_dm_reset) {
        dpole_mesh_reset_body(a.dm, b, a.n_dm_reset);
        return;
    }
    b -= a.n_dm_reset;
    if (b < a.n_ll_tab) {
        latlon_row_table_body(a.ll, b, a.n_ll_tab);
        return;
    }
    b -= a.n_ll_tab;
    if (b < a.n_mesh) bipolar_mesh_body<false>(a.mesh, lds.mesh, b % a.mesh_gx, b / a.mesh_gx);
}

struct PassBParams {
    FusedParams ll;
    LatlonShare share;
    MeshParams mesh;        // the cap mesh runs here, next to the quadrature; launch A only builds the tables
    long mesh_gx, n_mesh;
    QuadParams q;
    QuadRange guard, fast;
    long n_guard, n_fast;
    DpMeshParams dm;        // displaced-pole cap: mesh + unwrap + angle ...
    long n_dmesh;
    DpQuadParams dq;        // ... and its quadrature in the chord form (the literal form is launch D)
    long n_dquad;
    int dq_order;
    int order[5];           // dispatch order of the roles behind the lat-lon strips (ROLE_*)
    const PassAParams* next;   // NULL, or (device memory) launch A's parameters for the NEXT pass of the plan: its table roles ride at the
    long n_next;               // end of this launch, n_next workgroups of them (PassPipe)
};

// roles of launch B's workgroups; PassBParams::order lists them in dispatch order
enum { ROLE_BP_MESH = 0, ROLE_DP_MESH, ROLE_BP_GUARD, ROLE_BP_FAST, ROLE_DP_QUAD, N_ROLES };


#ifdef OGG_TIMELINE
// experiment builds only (scripts/ab_build.sh WORK tl -DOGG_TIMELINE=1; scripts/pass_timeline.py): when do the workgroups of each role of
// launch B start and end?  Slot r: 0 resident strips, 1 helpers, 2 + ROLE_*, 7 the next pass's tables.  100 MHz constant clock.
__device__ unsigned long long g_tl_first[8], g_tl_last_start[8], g_tl_end[8], g_tl_sum_end[8], g_tl_n[8];
__device__ unsigned long long g_tl_strip_end[512];   // end of every resident strip workgroup, by workgroup index
struct Timeline {
    unsigned long long t0;
    __device__ Timeline() : t0(wall_clock64()) {}
    __device__ void done(int r) const {
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned long long t1 = wall_clock64();
            atomicMin(&g_tl_first[r], t0), atomicMax(&g_tl_last_start[r], t0), atomicMax(&g_tl_end[r], t1);
            atomicAdd(&g_tl_sum_end[r], t1), atomicAdd(&g_tl_n[r], 1ull);
            if (r == 0 && blockIdx.x < 512) g_tl_strip_end[blockIdx.x] = t1;
        }
    }
};
extern "C" int ogg_timeline(unsigned long long* out40, int reset) {
    unsigned long long h[40];
    const void* sym[5] = {g_tl_first, g_tl_last_start, g_tl_end, g_tl_sum_end, g_tl_n};
    for (int k = 0; k < 5; ++k)
        if (hipMemcpyFromSymbol(h + 8 * k, sym[k], 64) != hipSuccess) return 1;
    if (out40) memcpy(out40, h, sizeof h);
    if (reset) {
        unsigned long long z[8] = {}, big[8];
        for (int k = 0; k < 8; ++k) big[k] = ~0ull;
        if (hipMemcpyToSymbol(g_tl_first, big, 64) != hipSuccess) return 1;
        for (int k = 1; k < 5; ++k)
            if (hipMemcpyToSymbol(sym[k], z, 64) != hipSuccess) return 1;
    }
    return 0;
}
extern "C" int ogg_timeline_strips(unsigned long long* out512) {
    return hipMemcpyFromSymbol(out512, g_tl_strip_end, sizeof g_tl_strip_end) == hipSuccess ? 0 : 1;
}
#define OGG_TL_DONE(r) tl.done(r)
#else
#define OGG_TL_DONE(r)
#endif

// TAB: the lat-lon strips read their rows' scalars from the table the table launch built (a.ll.row_tab).  Without atan(sinh) and sincos in
// it the strip role needs 81 VGPRs instead of 176, the kernel what its cap roles need (<= 128: four waves per SIMD instead of two), and
// the latency-bound cap workgroups of the launch get twice the wave slots.  TAB = false is the launch of a pass that owns no memory to
// keep a table in (the one-shot entry points, a one-slot plan).
template <int N, bool TAB>
__global__ __launch_bounds__(PASS_TX, TAB ? 4 : 1) void pass_b_kernel(PassBParams a) {
    __shared__ PassLds lds;
    __shared__ unsigned s_slot;
    __shared__ int s_claim;
#ifdef OGG_TIMELINE
    const Timeline tl;
#endif
    long b = blockIdx.x;
    if (b >= (long)gridDim.x - a.n_next) {   // (n_next = 0: never)
        pass_table_roles<N>(*a.next, b - ((long)gridDim.x - a.n_next));
        OGG_TL_DONE(7);
        return;
    }
    const long first_help = (long)gridDim.x - a.n_next - a.share.n_help;
    const bool helper = b >= first_help;
    if (b < a.share.n_wg || helper) {
        if (helper) {
            // helper k of the launch takes the block of a resident workgroup on ITS XCD (workgroup b runs on XCD b % 8)
            const long k = b - first_help, n = a.share.n_wg;
            long r = (k / 8) * 8 + (b % 8);
            b = r % n;
        }
        if (a.share.nt)
            latlon_fused_body<true, TAB>(a.ll, lds.ll, b, a.share.gx, a.share.gy, a.share.strip_lo, a.share.strip_hi, a.share.claims, helper, &s_claim, a.share.pool != 0);
        else
            latlon_fused_body<false, TAB>(a.ll, lds.ll, b, a.share.gx, a.share.gy, a.share.strip_lo, a.share.strip_hi, a.share.claims, helper, &s_claim, a.share.pool != 0);
        OGG_TL_DONE(helper ? 1 : 0);
        return;
    }
    b -= a.share.n_wg;
    int role = -1;
#pragma unroll
    for (int k = 0; k < N_ROLES; ++k) {
        const int r = a.order[k];
        const long n = (r == ROLE_BP_MESH) ? a.n_mesh : (r == ROLE_DP_MESH) ? a.n_dmesh : (r == ROLE_BP_GUARD) ? a.n_guard : (r == ROLE_BP_FAST) ? a.n_fast : a.n_dquad;
        if (role < 0) {
            if (b < n)
                role = r;
            else
                b -= n;
        }
    }
    if (role == ROLE_BP_MESH) {
        bipolar_mesh_body<false>(a.mesh, lds.mesh, b % a.mesh_gx, b / a.mesh_gx);
    } else if (role == ROLE_DP_MESH) {   // work item from the ticket, not from the workgroup index: see ogg_dpole_dev.h
        const long t = take_ticket(a.dm.ticket, &s_slot);
        dpole_mesh_body(a.dm, lds.dmesh, t % a.dm.gx, t / a.dm.gx);
    } else if (role == ROLE_BP_GUARD) {
        asm volatile("; role: guarded quadrature strips" ::: "memory");
        bipolar_quad_body<N, QM_GUARD>(a.q, a.guard, lds.quad, b % a.guard.gx, b / a.guard.gx);
    } else if (role == ROLE_BP_FAST) {
        asm volatile("; role: quadrature strips" ::: "memory");
        bipolar_quad_body<N, QM_FAST>(a.q, a.fast, lds.quad, b % a.fast.gx, b / a.fast.gx);
    } else if (role == ROLE_DP_QUAD) {
        const long strip = (b % a.dq.gx) * DQ_WAVES + (threadIdx.x >> 6), chunk = b / a.dq.gx;
        if (a.dq_order == 2)
            dpole_quad_body<2, DP_ARC_CHORD>(a.dq, strip, chunk);
        else
            dpole_quad_body<4, DP_ARC_CHORD>(a.dq, strip, chunk);
    }
    OGG_TL_DONE(2 + role);
}

// launch D: the displaced-pole quadrature in the reference's literal arithmetic (256 VGPRs, 70 KB of LDS: its own launch)
template <int N>
__global__ __launch_bounds__(PASS_TX, 2) void pass_d_kernel(DpQuadParams p) {
    __shared__ unsigned s_slot;
    const long t = take_ticket(p.ticket, &s_slot);
    __shared__ double ring[dq_ring_doubles<N>()];
    dpole_quad_literal_ring<N, DQ_RING>(p, (t % p.gx) * DQ_WAVES + (threadIdx.x >> 6), t / p.gx,
                                        ring + (threadIdx.x >> 6) * (DQ_RING * dq_ring_slot_doubles<N>()));
}

long env_long(const char* name, long dflt) {
    const char* e = getenv(name);
    return e ? atol(e) : dflt;
}

// pool: 0 = every resident workgroup its own block of strips, 1 = the strips of a column tile are handed out from one counter
// (ogg_latlon_fused_dev.h), -1 = the default: pooled when the launch is `light` or carries >= 16 M lat-lon points (OGG_PASS_LL_POOL overrides)
LatlonShare make_share(const FusedParams& ll, long lo, long hi, long ni1, bool alone, bool light = false, int pool = 0, bool table = false) {
    LatlonShare s{};
    s.gx = latlon_gx(ni1);
    s.strip_lo = lo, s.strip_hi = hi;
    s.nt = alone ? 0 : (int)env_long("OGG_PASS_LL_NT", 1);
    if (hi <= lo) return s;
    // resident lat-lon workgroups: enough to keep the HBM write path busy and not more, so that the VALU-bound workgroups of
    // the same launch get the remaining wave slots; a launch without compute workgroups takes the whole chip
    const long points = (hi - lo) * ll.rows_per_block * ni1;
    s.points = points;
    // measured optima (1/8 degree, shares 1, 1/2, 1/4, 1/8 = 19.8, 9.9, 5, 2.5 M points): 60, 90, 90-120, 120 -- the smaller the share,
    // the shorter the VALU work the strips can hide behind, so they need more of the write bandwidth
    // `light`: the caps' columns are mirrored (OGG_SYM_MIRROR: a third of the cap arithmetic).  The launch is then bound by its WRITES at
    // every size (VALU busy 28 %, no power limit: profiles/r05_*), the cap workgroups drain early, and the strips want more of the chip
    // for the whole launch instead of helpers at its end: round-robin sweeps on three boxes (scripts/config_sweep.py; helpers off) --
    // whole 1/8 degree grid 144-192 resident workgroups within 1 % of each other and 4-10 % faster than 90 + helpers, 1/16 degree 161
    // (184: +1.5 %, 138: +1 %), half / quarter of the 1/8 degree grid 150-180, an eighth 120 as before
    // pooled strips (round 5, after the mirrored caps): every XCD takes what it can write, the strips end together, and FEWER resident
    // workgroups are best -- the cap workgroups get the wave slots, the strips still end with them: 1/8 degree 84-96 (0.2171 -> 0.2005 ms
    // on a fast box, 0.2397 -> 0.2158 on a slow one; 156: 0.2137), 1/16 degree 72-138 (-2 ... -3 %; 72 against 96: -1.2 % on three boxes
    // with the row table; on boxes in their fast state 138-161 are 4-5 % faster still, on the slow ones 8 % slower: 72 stays); 1/8 degree
    // with 6-row table-fed strips: 108 (nine per column tile) against 96: -3.2 % and -3.9 % on two boxes (0.1948 -> 0.1886; 120: 0.1938,
    // 132: 0.1912); the shares of that grid, 6-row table-fed strips, 108 against the 96 / 96 / 120 of the longer strips: a half 0.1028 ->
    // 0.1004, a quarter 0.0534 -> 0.0508, an eighth 0.0295 -> 0.0291 (the last rank's smaller share 0.0353 -> 0.0344 at 96); with a displaced-pole quadrature in
    // the launch 48-60 and as many helpers behind the compute roles (0.3145 -> 0.286)
    if (pool < 0) pool = (int)env_long("OGG_PASS_LL_POOL", (light || points >= 16000000) ? 1 : 0);
    s.pool = pool;
    // shares of the 1/8 degree grid, pooled (same box, against the owned blocks at their own best counts): a half 96 resident workgroups
    // (0.1012 -> 0.0962 ms), a quarter 96 (0.0572 -> 0.0561; the last rank of four 0.0663 -> 0.0597), an eighth 120 (0.0307 -> 0.0295);
    // the 1/4 degree grid with its displaced-pole quadrature is better off with owned blocks (+1 ... +9 % pooled)
    long max_wg = alone ? 2048
                        : (points >= 16000000 ? env_long("OGG_PASS_LL_WG", pool ? (light ? (points >= 48000000 ? 72 : (table ? 108 : 96)) : (table ? 96 : 60)) : (light ? 156 : 90))
                                              : (points >= 8000000 ? env_long("OGG_PASS_LL_WG_MID", pool ? (table ? 108 : 96) : (light ? 150 : 90))
                                                                   : env_long("OGG_PASS_LL_WG_SMALL", points >= 4000000 ? (pool ? (table ? 108 : 96) : (light ? 180 : 120))
                                                                                                                        : ((pool && table) ? 108 : 120))));
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

// bytes the displaced-pole band writes: mesh + angle (launch B), quadrature (launch B or D)
double dpole_mesh_bytes(const ogg_dpole_band& d) { return 8.0 * 3.0 * (double)d.n_pt_rows * (double)(d.Ni + 1); }
double dpole_quad_bytes(const ogg_dpole_band& d) {
    return 8.0 * ((double)d.n_pt_rows * (double)d.Ni + (double)d.n_cell_rows * (2.0 * (double)d.Ni + 1.0));
}

// Everything a pass needs that does not change from one call to the next: the kernel parameters of the launches, their grid sizes,
// the bytes they write.  Built once per (bands, knobs) by build_pass_plan -- the environment knobs are read THERE -- and replayed by
// run_pass_plan, whose host work is the launches themselves.
struct PassPlan {
    int order;                 // Gauss-Lobatto order the bipolar kernels are instantiated for
    PassAParams A;
    PassBParams B;
    QuadPlan qp;
    DpQuadParams dq;
    int dq_order;
    bool have_quad, dq_literal, launch_b;
    unsigned na, nb, nd;
    unsigned b_lds_pad;        // dynamic LDS bytes launch B asks for on top of its static 12 KB: nothing uses them, they cap the workgroups per CU
    double alg_bytes[4];
};

template <int N>
int build_pass_plan(const FusedParams& ll, long ni1, int metrics, const ogg_bipolar_band* cap, const ogg_dpole_band* scap, PassPlan& P) {
    P = PassPlan{};
    P.order = N;
    const long n_strips_ll = ll.n_bands ? ll.strip0[ll.n_bands] : 0;
    const bool have_cap = cap && cap->n_pt_rows > 0;
    const bool have_quad = have_cap && metrics;
    const bool have_dp = scap && scap->n_pt_rows > 0;
    const bool have_dquad = have_dp && metrics;
    const bool dq_literal = have_dquad && scap->arc_form == OGG_DP_ARC_LITERAL;
    PassAParams& A = P.A;
    PassBParams& B = P.B;
    QuadPlan& qp = P.qp;
    A.ll = ll, B.ll = ll;
    if (have_cap) {
        A.mesh = MeshParams{cap->Ni, cap->Nj, cap->lat0_bp, cap->lon_bp, cap->j0, cap->n_pt_rows, cap->x, cap->y, nullptr, nullptr, cap->angle, MESH_ROWS, {}};
        const dim3 mg = mesh_grid(A.mesh, ogg::cap_symmetry(cap->symmetry) ? 1 : 0);
        A.mesh_gx = mg.x, A.n_mesh = (long)mg.x * mg.y;
    }
    if (have_quad) {
        double guard_k = BP_GUARD_K_DEFAULT;
        if (const char* e = getenv("OGG_BP_GUARD_K")) guard_k = atof(e);
        QuadParams p{};
        p.nx = cap->Ni, p.ny = cap->Nj, p.lat0_bp = cap->lat0_bp, p.lon_bp = cap->lon_bp, p.rp = cap->rp, p.Re = cap->Re, p.j0 = cap->j0;
        p.dxq = cap->dx, p.dyq = cap->dy, p.daq = cap->area, p.q = make_nodes(N);
        if (int e = plan_quad<N>(p, cap->n_pt_rows, cap->n_cell_rows, guard_k, ogg::cap_symmetry(cap->symmetry) ? 1 : 0, cap->workspace,
                                 cap->workspace_bytes, qp))
            return e;
        A.q = qp.p, B.q = qp.p;
        A.n_tab = tables_blocks<N>(qp.p);
        B.guard = qp.guard, B.n_guard = qp.has_guard ? (long)qp.guard.gx * qp.guard.gy : 0;
        B.fast = qp.fast, B.n_fast = qp.has_fast ? (long)qp.fast.gx * qp.fast.gy : 0;
    }
    DpQuadParams& dq = P.dq;
    if (have_dp) {   // workspace of the band: [mesh words][quadrature tables and words]
        const DpGeom g{scap->Ni, scap->Nj, scap->lon0, scap->lat0, scap->lon_dp, scap->r_dp};
        const long mesh_ws = (long)((dm_workspace_bytes(scap->Ni, scap->n_pt_rows) + 255) / 256 * 256);
        OGG_REQUIRE(scap->workspace && scap->workspace_bytes >= mesh_ws, OGG_EARG, "ogg_supergrid_pass: displaced-pole workspace too small");
        if (int e = plan_dmesh(g, scap->j0, scap->n_pt_rows, scap->x, scap->y, scap->angle, scap->workspace, mesh_ws, B.dm)) return e;
        A.dm = B.dm;
        A.n_dm_reset = dm_reset_blocks(B.dm);
        B.n_dmesh = dm_blocks(B.dm);
        if (have_dquad) {
            if (int e = plan_dquad(scap->arc_form, scap->order, g, scap->Re, scap->j0, scap->n_pt_rows, scap->n_cell_rows, scap->dx, scap->dy,
                                   scap->area, static_cast<char*>(scap->workspace) + mesh_ws, scap->workspace_bytes - mesh_ws,
                                   make_nodes(scap->order), dq, ogg::cap_symmetry(scap->symmetry) ? 1 : 0))
                return e;
            A.dq = dq, A.dq_order = scap->order;
            A.n_dq_tab = scap->order == 2 ? dpole_quad_tables_blocks<2>(dq) : dpole_quad_tables_blocks<4>(dq);
            if (!dq_literal) B.dq = dq, B.dq_order = scap->order, B.n_dquad = dq.gx * dq.n_chunks;
            P.dq_order = scap->order;
        }
    }
    // With metrics, launch A builds only the tables and launch B carries everything else, the cap meshes included: one long launch
    // in which all kinds of workgroup overlap (measured faster than mesh + part of the lat-lon strips in A at every share
    // from 1 to 1/8 of the 1/8 degree grid).  Without metrics and without a displaced-pole cap there is no launch B: A carries the
    // lat-lon strips and the bipolar mesh.  (The displaced-pole mesh cannot run in A: its look-back words are reset there.)
    const bool launch_b = (have_quad && (B.n_guard + B.n_fast > 0)) || have_dp;
    const long s1 = launch_b ? 0 : n_strips_ll;   // lat-lon strips [0, s1) in launch A, [s1, S) in launch B
    if (launch_b) {
        B.mesh = A.mesh, B.mesh_gx = A.mesh_gx, B.n_mesh = A.n_mesh;
        A.n_mesh = 0;
    }
    {  // algorithmic bytes written by each launch (bench.py prices the launches against the HBM roofline)
        const double ni = (double)(ni1 - 1);
        const double mesh_bytes = have_cap ? 8.0 * 3.0 * (double)cap->n_pt_rows * (double)ni1 : 0.0;
        const double bq_bytes = have_quad ? 8.0 * (double)cap->n_cell_rows * (3.0 * ni + 1.0) : 0.0;
        P.alg_bytes[0] = latlon_strip_bytes(ll, 0, s1) + (launch_b ? 0.0 : mesh_bytes) + ((have_quad && qp.has_top) ? 8.0 * ni : 0.0);   // + dxq[ny]
        P.alg_bytes[1] = launch_b ? latlon_strip_bytes(ll, s1, n_strips_ll) + bq_bytes + mesh_bytes + (have_dp ? dpole_mesh_bytes(*scap) : 0.0) +
                                        ((have_dquad && !dq_literal) ? dpole_quad_bytes(*scap) : 0.0)
                                  : 0.0;
        P.alg_bytes[2] = 0.0;   // the tail rewrites cells launch B has written: no bytes of its own
        P.alg_bytes[3] = dq_literal ? dpole_quad_bytes(*scap) : 0.0;
    }
    A.share = make_share(ll, 0, s1, ni1, !have_cap && !have_dp);
    // the strips of launch B read their rows' scalars from ll.row_tab when the caller has memory for it (a two-slot plan handle); the
    // strips of launch A (a pass without launch B) always evaluate them themselves: nothing orders a table before them
    if (!(launch_b && ll.row_tab && n_strips_ll > 0)) A.ll.row_tab = nullptr, B.ll.row_tab = nullptr;
    A.n_ll_tab = B.ll.row_tab ? latlon_row_table_blocks(ll) : 0;
    P.na = (unsigned)(A.share.n_wg + A.n_tab + A.n_dq_tab + A.n_dm_reset + A.n_ll_tab + A.n_mesh);
    if (launch_b) {
        // dispatch order of the compute roles.  A big launch (>= ~2 M quadrature cells: a whole 1/8 degree cap or half of it) puts the
        // quadrature strips first (long, issue-bound waves) and the meshes last -- their short workgroups fill the slots the draining
        // strips leave: 2 % faster than meshes first at 1/8 degree on one GPU.  A small launch (a quarter of that cap or less) no
        // longer fills the chip; there the latency-bound meshes want to start first: 3.5 % faster at 1/4, 7 % at 1/8 of the cap
        // (scripts/order_sweep.py, scripts/rank_sweep.py).  OGG_PASS_ORDER = permutation of "01234" (ROLE_* ids) for experiments
        static const int quad_first[N_ROLES] = {ROLE_BP_FAST, ROLE_BP_GUARD, ROLE_DP_QUAD, ROLE_BP_MESH, ROLE_DP_MESH};
        static const int mesh_first[N_ROLES] = {ROLE_BP_MESH, ROLE_DP_MESH, ROLE_BP_GUARD, ROLE_BP_FAST, ROLE_DP_QUAD};
        const double quad_cells = (have_quad ? (double)cap->n_cell_rows * (double)cap->Ni : 0.0) +
                                  ((have_dquad && !dq_literal) ? (double)scap->n_cell_rows * (double)scap->Ni : 0.0);
        const int* dflt = quad_cells >= 2.0e6 ? quad_first : mesh_first;
        const char* ord = getenv("OGG_PASS_ORDER");
        for (int k = 0; k < N_ROLES; ++k)
            B.order[k] = (ord && strlen(ord) == N_ROLES && ord[k] >= '0' && ord[k] < '0' + N_ROLES) ? ord[k] - '0' : dflt[k];
        // `light`: mirrored caps and no displaced-pole quadrature in this launch (with one, the launch still carries enough arithmetic for
        // the strips to hide behind: 1/8 degree with the displaced pole 0.257 ms at 90 resident workgroups, 0.269 at 120, 0.285 at 150)
        const bool light = have_cap && ogg::cap_symmetry(cap->symmetry) && B.n_dquad == 0;
        B.share = make_share(ll, s1, n_strips_ll, ni1, false, light, have_quad ? -1 : 0);
        // Strips that are handed out from the tiles' counters AND copy their rows' scalars from the table cost next to nothing to start
        // (one 32-byte load per row, one barrier; the ticket for the next strip is already on its way), and short ones keep the rows the
        // resident workgroups write at any moment close together: 6 rows instead of the 21 / 32 that paid for a strip's own atan(sinh) and
        // sincos -- 1/8 degree 0.2216 -> 0.2035 ms, 1/16 degree 1.1464 -> 1.0672, the upper half of the 1/8 degree grid 0.1075 -> 0.1039
        // (8), an eighth +-0 (same process each; 4 rows: the ticket's answer no longer has 16 stores to hide behind).  Next to a
        // displaced-pole quadrature short strips lose with its 60 resident strip workgroups (21 rows 0.2696, 12 0.2764, 6 0.2951) and win
        // with 96 of them (0.2618 -> 0.2545, same process; 80: 0.2574, 120: 0.2644): the two go together.
        if (B.share.pool && B.ll.row_tab && !getenv("OGG_LL_ROWS_PER_STRIP")) {
            const long rpb = env_long("OGG_PASS_LL_ROWS_TABLE", 6);
            if (rpb > 0 && rpb != B.ll.rows_per_block) {
                set_rows_per_strip(B.ll, rpb), set_rows_per_strip(A.ll, rpb);
                B.share = make_share(B.ll, 0, B.ll.strip0[B.ll.n_bands], ni1, false, light, have_quad ? -1 : 0, true);
            }
        }
        // helper workgroups for the lat-lon strips at the end of the launch (ogg_latlon_fused_dev.h): the claim counters live in the
        // bipolar cap's workspace and are zeroed by launch A with its tables
        // (measured: 1/8 degree whole grid 0.246 -> 0.237 ms, 1/16 degree 1.23 -> 1.19; half, a quarter, an eighth of the 1/8 degree grid
        // +1 %, +1 %, +4 %: the claims cost more than a tail that short gives back -- whole-grid launches only)
        // (mirrored caps: the compute roles drain early, helpers would join almost at once and only add writers -- measured 2-5 % slower
        // than none on three boxes: off)
        const long helpers = env_long("OGG_PASS_LL_HELPERS", (B.share.points >= 16000000 && !light) ? 2 : 0);
        if (B.share.pool && B.share.n_wg > 0 && B.share.gx <= QUAD_LL_CLAIM_WORDS) {
            B.share.claims = qp.p.ll_claims;
            // (with the row table -- launch B at four waves per SIMD -- the compute roles no longer leave a tail for helpers to fill:
            // config 4 0.2776 -> 0.2710 ms without them, same process)
            B.share.n_help = B.share.n_wg * env_long("OGG_PASS_LL_POOL_HELPERS", (light || B.ll.row_tab) ? 0 : 1);
        } else if (B.share.pool) {
            B.share.pool = 0;
        } else if (have_quad && helpers > 0 && B.share.n_wg > 0 && 2 * B.share.n_wg <= QUAD_LL_CLAIM_WORDS) {
            B.share.claims = qp.p.ll_claims;
            B.share.n_help = B.share.n_wg * helpers;
        }
        P.nb = (unsigned)(B.share.n_wg + B.n_mesh + B.n_dmesh + B.n_guard + B.n_fast + B.n_dquad + B.share.n_help);
        // Launch B with the row table runs four workgroups per CU.  A whole-grid launch with mirrored caps is bound by its writes, and there
        // three per CU are 0.5-1.5 % faster (fewer cap workgroups writing at once: 1/8 degree 0.2359 -> 0.2343 ms, same process; the same
        // from a build held to three by its registers); the launches that carry a displaced-pole quadrature or a share of the grid want the
        // four (1/4 degree OM4 0.0650 against 0.0685).  41000 bytes of unused dynamic LDS on top of the static 12 KB: three fit in 160 KB.
        P.b_lds_pad = (unsigned)env_long("OGG_PASS_B_LDS_PAD", (B.ll.row_tab && light && B.share.points >= 16000000) ? 41000 : 0);
    }
    P.nd = dq_literal ? (unsigned)(dq.gx * dq.n_chunks) : 0u;
    P.have_quad = have_quad, P.dq_literal = dq_literal, P.launch_b = launch_b;
    return OGG_OK;
}

template <int N>
int run_pass_plan(const PassPlan& P, hipEvent_t* ev, double* alg_bytes4, hipStream_t st, bool with_a = true, const PassAParams* next = nullptr,
                  long n_next = 0) {
    // ev: NULL, or 5 events (entries may be NULL) recorded before launch A and after launches A, B, C and D (bench.py times the
    // launches with them).  with_a = false: the tables of this pass were built by the previous pass's launch B.  next / n_next: the
    // table roles launch B carries for the next pass (device copy of that pass's PassAParams, number of workgroups).
    auto mark = [&](int k) -> int {
        if (ev && ev[k]) OGG_HIP_CHECK(hipEventRecord(ev[k], st));
        return OGG_OK;
    };
    if (alg_bytes4)
        for (int k = 0; k < 4; ++k) alg_bytes4[k] = P.alg_bytes[k];
    if (int e = mark(0)) return e;
    if (P.na > 0 && with_a) {
        pass_a_kernel<N><<<P.na, PASS_TX, 0, st>>>(P.A);
        OGG_LAUNCH_CHECK();
    }
    if (int e = mark(1)) return e;
    if (P.launch_b) {
        PassBParams b = P.B;
        if (next && n_next > 0) b.next = next, b.n_next = n_next;
        if (b.ll.row_tab)
            pass_b_kernel<N, true><<<P.nb + (unsigned)b.n_next, PASS_TX, P.b_lds_pad, st>>>(b);
        else
            pass_b_kernel<N, false><<<P.nb + (unsigned)b.n_next, PASS_TX, 0, st>>>(b);
        OGG_LAUNCH_CHECK();
    }
    if (int e = mark(2)) return e;
    if (P.have_quad) {
        if (int e = launch_quad_tail<N>(P.qp, st)) return e;
    }
    if (int e = mark(3)) return e;
    if (P.dq_literal) {
        if (P.dq_order == 2)
            pass_d_kernel<2><<<P.nd, PASS_TX, 0, st>>>(P.dq);
        else
            pass_d_kernel<4><<<P.nd, PASS_TX, 0, st>>>(P.dq);
        OGG_LAUNCH_CHECK();
    }
    return mark(4);
}

int build_pass_plan_any(int n_latlon, const ogg_latlon_band* latlon, long ni1, double lon0, double lenlon, double Re, int metrics,
                        const ogg_bipolar_band* cap, const ogg_dpole_band* south_cap, PassPlan& P, const RowScalars* row_tab = nullptr) {
    FusedParams ll;
    long points = 0;
    if (int e = plan_latlon(n_latlon, latlon, ni1, lon0, lenlon, Re, metrics, ll, points)) return e;
    ll.row_tab = row_tab;   // (memory for ll.row0[n_bands] entries, or NULL)
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
    if (south_cap && south_cap->n_pt_rows > 0) {
        const ogg_dpole_band& d = *south_cap;
        OGG_REQUIRE(d.Ni + 1 == ni1, OGG_ESHAPE, "ogg_supergrid_pass: southern cap has %ld columns, the lat-lon bands %ld", d.Ni + 1, ni1);
        OGG_REQUIRE(d.Nj > 0 && d.j0 >= 0 && d.j0 + d.n_pt_rows <= d.Nj + 1, OGG_ESHAPE,
                    "ogg_supergrid_pass: southern cap rows %ld..%ld outside 0..%ld", d.j0, d.j0 + d.n_pt_rows, d.Nj);
        OGG_REQUIRE(d.x && d.y && d.angle, OGG_EARG, "ogg_supergrid_pass: null southern cap output");
        OGG_REQUIRE(d.workspace && d.workspace_bytes >= ogg_dpole_band_workspace_bytes(metrics ? d.order : 4, d.Ni, d.n_pt_rows), OGG_EARG,
                    "ogg_supergrid_pass: the southern cap needs a workspace of ogg_dpole_band_workspace_bytes bytes");
        if (metrics) {
            OGG_REQUIRE(d.order >= 2 && d.order <= 5, OGG_EORDER, "Uncoded order");
            OGG_REQUIRE(d.order == 2 || d.order == 4, OGG_EORDER, "order not coded");   // OGG:547: the quadrature order is the FD order
            OGG_REQUIRE(d.arc_form == OGG_DP_ARC_LITERAL || d.arc_form == OGG_DP_ARC_CHORD, OGG_EARG, "ogg_supergrid_pass: arc_form %d", d.arc_form);
            OGG_REQUIRE(d.dx && (d.n_cell_rows <= 0 || (d.dy && d.area)), OGG_EARG, "ogg_supergrid_pass: null southern cap metrics output");
            OGG_REQUIRE(d.n_cell_rows >= 0 && d.j0 + d.n_cell_rows <= d.Nj &&
                            (d.n_pt_rows == d.n_cell_rows || (d.n_pt_rows == d.n_cell_rows + 1 && d.j0 + d.n_cell_rows == d.Nj)),
                        OGG_ESHAPE, "ogg_supergrid_pass: southern cap band j0=%ld cell rows=%ld point rows=%ld of Nj=%ld", d.j0, d.n_cell_rows,
                        d.n_pt_rows, d.Nj);
        }
    }
    switch (order) {
        case 2: return build_pass_plan<2>(ll, ni1, metrics, cap, south_cap, P);
        case 3: return build_pass_plan<3>(ll, ni1, metrics, cap, south_cap, P);
        case 4: return build_pass_plan<4>(ll, ni1, metrics, cap, south_cap, P);
        default: return build_pass_plan<5>(ll, ni1, metrics, cap, south_cap, P);
    }
}

int run_pass_plan_any(const PassPlan& P, void** events5, double* alg_bytes4, void* stream, bool with_a = true, const PassAParams* next = nullptr,
                      long n_next = 0) {
    hipStream_t st = ogg::as_stream(stream);
    hipEvent_t* ev = reinterpret_cast<hipEvent_t*>(events5);
    switch (P.order) {
        case 2: return run_pass_plan<2>(P, ev, alg_bytes4, st, with_a, next, n_next);
        case 3: return run_pass_plan<3>(P, ev, alg_bytes4, st, with_a, next, n_next);
        case 4: return run_pass_plan<4>(P, ev, alg_bytes4, st, with_a, next, n_next);
        default: return run_pass_plan<5>(P, ev, alg_bytes4, st, with_a, next, n_next);
    }
}

// A plan handle: the plan of a pass for each of two workspace slots (both the plan's own), launch A's parameters of both in device memory (the table roles
// that ride in the other slot's launch B read them from there), and which slot's tables the last launch B has built.
struct PassPipe {
    int n_slots = 1;                        // 1: every pass runs its own launch A (OGG_PASS_SLOTS=1, or a pass without launch B)
    PassPlan slot[2];
    void* own_ws[2][2] = {};                // [slot][bipolar cap, southern cap]: with two slots BOTH are the plan's own allocations -- what is in
                                            // them outlives a run (the tables of the next pass), so nobody else may write there; the caller's
                                            // workspaces are used by a one-slot plan only
    void* own_row_tab[2] = {};              // [slot]: the lat-lon row table of the slot's passes (RowScalars per row; OGG_PASS_LL_TABLE=0: none)
    PassAParams* dev_a = nullptr;           // [2]
    unsigned long long runs = 0;            // passes issued
    int ready_slot = -1;                    // the slot whose tables the previous pass's launch B built
    unsigned long long carried_runs = 0;    // passes that started with launch B
    int device = 0;
    hipStream_t last_stream = nullptr;      // the stream of the last pass (the plan's passes are ordered by the caller's stream)
    bool ran = false;
    bool captured = false;                  // a pass of this plan went into a stream capture: its graph may replay on ANY stream

    ~PassPipe() {
        if (n_slots > 1) {
            int current = 0;
            (void)hipGetDevice(&current);
            (void)hipSetDevice(device);
            // nothing of the plan may still be in flight when its workspaces go: wait for the stream of its last pass (not for the
            // whole device: other streams are none of the plan's business) -- unless a pass was captured: the graph's replays run on
            // whatever stream the caller launches them on, which the plan never sees, so then the whole device; likewise when the
            // stream no longer exists.  (A captured graph must not be replayed after the plan is destroyed: include/ogg_hip.h.)
            if (ran && (captured || hipStreamSynchronize(last_stream) != hipSuccess)) {
                (void)hipGetLastError();
                (void)hipDeviceSynchronize();
            }
            for (auto& slot_ws : own_ws)
                for (void* w : slot_ws)
                    if (w) (void)hipFree(w);
            for (void* w : own_row_tab)
                if (w) (void)hipFree(w);
            if (dev_a) (void)hipFree(dev_a);
            (void)hipSetDevice(current);
        }
    }
};

void route_top_row_through_workspace(PassPlan& P) {
    if (!P.have_quad || !P.qp.has_top) return;
    P.A.q.dxq = P.qp.top_buf, P.A.q.top_out_row = 0;
    P.qp.p.top_src = P.qp.top_buf;
}

int build_pass_pipe(int n_latlon, const ogg_latlon_band* latlon, long ni1, double lon0, double lenlon, double Re, int metrics,
                    const ogg_bipolar_band* cap, const ogg_dpole_band* south_cap, PassPipe& H) {
    if (int e = build_pass_plan_any(n_latlon, latlon, ni1, lon0, lenlon, Re, metrics, cap, south_cap, H.slot[0])) return e;
    // without launch B, launch A writes the outputs themselves: nothing to carry
    if (env_long("OGG_PASS_SLOTS", 2) < 2 || !H.slot[0].launch_b || H.slot[0].na == 0) return OGG_OK;
    OGG_HIP_CHECK(hipGetDevice(&H.device));
    const bool have_cap = cap && cap->n_pt_rows > 0 && cap->workspace, have_dp = south_cap && south_cap->n_pt_rows > 0;
    H.n_slots = 2;   // from here on ~PassPipe frees what has been allocated
    const long n_tab_rows = H.slot[0].A.ll.n_bands ? H.slot[0].A.ll.row0[H.slot[0].A.ll.n_bands] : 0;
    const bool row_table = n_tab_rows > 0 && env_long("OGG_PASS_LL_TABLE", 1) != 0;
    for (int k = 0; k < 2; ++k) {
        if (row_table) {
            OGG_HIP_CHECK(hipMalloc(&H.own_row_tab[k], (size_t)n_tab_rows * sizeof(RowScalars)));
            OGG_HIP_CHECK(hipMemset(H.own_row_tab[k], 0, (size_t)n_tab_rows * sizeof(RowScalars)));
        }
        ogg_bipolar_band c{};
        ogg_dpole_band d{};
        if (have_cap) {
            c = *cap;
            OGG_HIP_CHECK(hipMalloc(&H.own_ws[k][0], (size_t)cap->workspace_bytes));
            OGG_HIP_CHECK(hipMemset(H.own_ws[k][0], 0, (size_t)cap->workspace_bytes));
            c.workspace = H.own_ws[k][0];
        }
        if (have_dp) {
            d = *south_cap;
            OGG_HIP_CHECK(hipMalloc(&H.own_ws[k][1], (size_t)south_cap->workspace_bytes));
            OGG_HIP_CHECK(hipMemset(H.own_ws[k][1], 0, (size_t)south_cap->workspace_bytes));
            d.workspace = H.own_ws[k][1];
        }
        if (int e = build_pass_plan_any(n_latlon, latlon, ni1, lon0, lenlon, Re, metrics, cap ? (have_cap ? &c : cap) : nullptr,
                                        south_cap ? (have_dp ? &d : south_cap) : nullptr, H.slot[k], static_cast<const RowScalars*>(H.own_row_tab[k])))
            return e;
    }
    for (PassPlan& P : H.slot) route_top_row_through_workspace(P);
    OGG_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&H.dev_a), 2 * sizeof(PassAParams)));
    for (int k = 0; k < 2; ++k) OGG_HIP_CHECK(hipMemcpy(H.dev_a + k, &H.slot[k].A, sizeof(PassAParams), hipMemcpyHostToDevice));
    OGG_HIP_CHECK(hipDeviceSynchronize());
    return OGG_OK;
}

int run_pass_pipe(PassPipe& H, void** events5, double* alg_bytes4, void* stream) {
    H.last_stream = ogg::as_stream(stream), H.ran = true;
    if (H.n_slots == 1) return run_pass_plan_any(H.slot[0], events5, alg_bytes4, stream);
    // Under stream capture the launches become graph nodes that are REPLAYED: a replay must find nothing that a previous pass left behind, so
    // the captured pass runs its own launch A (which resets the slot's counters, tickets and look-back words) and builds nobody's next
    // tables; the host state does not advance (nothing has executed), except that no slot counts as prepared afterwards.
    hipStreamCaptureStatus cap_status = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(ogg::as_stream(stream), &cap_status) == hipSuccess && cap_status != hipStreamCaptureStatusNone) {
        const int sc = (int)(H.runs & 1ull);
        H.ready_slot = -1;
        H.captured = true;
        return run_pass_plan_any(H.slot[sc], events5, alg_bytes4, stream, true, nullptr, 0);
    }
    const int s = (int)(H.runs & 1ull), o = s ^ 1;
    // launch A itself on the first pass of the plan, and on a pass that times its launches (the events then time it); otherwise the
    // previous pass's launch B has built this slot's tables
    const bool carried = H.ready_slot == s && events5 == nullptr;
    const int rc = run_pass_plan_any(H.slot[s], events5, alg_bytes4, stream, !carried, H.dev_a + o, (long)H.slot[o].na);
    H.runs += 1;
    H.carried_runs += carried ? 1 : 0;
    H.ready_slot = (rc == OGG_OK) ? o : -1;
    return rc;
}

}  // namespace

// Host-side replay of the column spaces of the mirrored kernels (the same quad_lane / mesh_lane / dq_lane the kernels call): how many
// times every cell and every node column of one row is written, and how many cells / columns are evaluated.  No GPU involved.
extern "C" int ogg_symmetry_coverage(int which, int order, long n, double lon0, double lon_dp, int symmetry, int* cell_writes, int* col_writes,
                                     long* evaluated) {
    OGG_REQUIRE(which >= 0 && which <= 2 && n > 0 && col_writes && evaluated && (which == 1 || cell_writes), OGG_EARG,
                "ogg_symmetry_coverage: bad argument");
    const int sym = ogg::cap_symmetry(symmetry) ? 1 : 0;
    for (long k = 0; k <= n; ++k) col_writes[k] = 0;
    if (cell_writes)
        for (long k = 0; k < n; ++k) cell_writes[k] = 0;
    *evaluated = 0;
    auto bump = [](int* w, long a, long b, long c, long d) {   // one lane writes its value to a, b, c, d: every DISTINCT target once
        w[a] += 1;
        if (b != a) w[b] += 1;
        if (c != a && c != b) w[c] += 1;
        if (d != a && d != b && d != c) w[d] += 1;
    };
    if (which == 0) {   // the runs the workgroups store (quad_stores: the same offsets the kernel uses) + the closing columns, stored by their lanes
        const QuadCols cs = quad_cols(n, sym);
        const long h2 = n / 2;
        for (long wg = 0; wg < cs.g_end[2]; ++wg) {
            const QuadLane q0 = quad_lane(cs, wg, 0, 0);
            if (q0.nv == 0) continue;
            const QuadStores st = quad_stores(cs, q0, n, true);
            *evaluated += q0.nv;
            for (long k = 0; k < q0.nv; ++k) cell_writes[q0.c0 + k] += 1, col_writes[q0.c0 + k] += 1;
            for (long k = 0; k < st.n_ic; ++k) cell_writes[st.cell_up + k] += 1, cell_writes[st.cell_dn2 + k] += 1, cell_writes[st.cell_dn4 + k] += 1;
            for (long k = 0; k < st.n_ip; ++k) col_writes[st.col_up + k] += 1, col_writes[st.col_dn2 + k] += 1, col_writes[st.col_dn4 + k] += 1;
            for (int wave = 0; wave < QS_WAVES; ++wave)
                for (int lane = 0; lane < 64; ++lane) {
                    const QuadLane q = quad_lane(cs, wg, wave, lane);
                    OGG_REQUIRE(q.nv == q0.nv && q.c0 == q0.c0, OGG_EARG, "ogg_symmetry_coverage: workgroup-uniform values differ");
                    OGG_REQUIRE(!q.cell_lane || (q.ci >= q.c0 && q.ci < q.c0 + q.nv && q.ci - q.c0 == wave * QS_CELLS + lane), OGG_EARG,
                                "ogg_symmetry_coverage: a cell lane outside its workgroup's row");
                    OGG_REQUIRE(q.img_cell == (q.cell_lane && st.n_ic > 0 && q.ci - q.c0 >= st.k0) &&
                                    (q.closing || q.img_col == (q.cell_lane && st.n_ip > 0 && q.ci - q.c0 >= st.k1)), OGG_EARG,
                                "ogg_symmetry_coverage: image flags of a lane and image runs of its workgroup disagree");
                    if (q.closing) {
                        *evaluated += 1;
                        if (q.img_col) bump(col_writes, q.ci, h2 - q.ci, h2 + q.ci, n - q.ci);
                        else col_writes[q.ci] += 1;
                    }
                }
        }
    } else if (which == 1) {
        const MeshCols mc = mesh_cols(n, sym);
        const long h2 = n / 2;
        for (long w = 0; w < mc.w_end[3]; ++w)
            for (int lane = 0; lane < 64; ++lane) {
                const MeshLane m = mesh_lane(mc, n, w, lane, MESH_OUT);
                if (!m.out) continue;
                *evaluated += 1;
                if (m.img) bump(col_writes, m.i, h2 - m.i, h2 + m.i, n - m.i);
                else col_writes[m.i] += 1;
            }
    } else {
        OGG_REQUIRE(order == 2 || order == 4, OGG_EORDER, "order not coded");
        DpQuadParams p{};
        const size_t ws_bytes = dq_workspace_bytes(order, n, 4);
        std::vector<char> ws(ws_bytes);   // plan_dquad carves pointers out of it; nothing is dereferenced here
        if (int e = plan_dquad(DP_ARC_CHORD, order, DpGeom{n, 8, lon0, -78.0, lon_dp, 0.2}, kReDefault, 0, 4, 4, nullptr, nullptr, nullptr, ws.data(),
                               (long)ws_bytes, make_nodes(order), p, sym))
            return e;
        const int M = order - 1;
        for (long strip = 0; strip < p.n_strips + 1; ++strip)
            for (int lane = 0; lane < 64; ++lane) {
                const DqLane q = dq_lane(p, M, strip, lane, DQ_COLS);
                if (!q.active) continue;
                if (q.out_lane) {
                    *evaluated += 1;
                    cell_writes[q.ci] += 1;
                    if (p.sym && q.cm != q.ci) cell_writes[q.cm] += 1;
                }
                if (q.out_lane || q.dy_edge) {
                    col_writes[q.ci] += 1;
                    if (p.sym) {
                        if (q.pm != q.ci) col_writes[q.pm] += 1;
                        if (q.pm == 0 && q.ci != n) col_writes[n] += 1;
                    }
                }
            }
    }
    return OGG_OK;
}

extern "C" long ogg_dpole_band_workspace_bytes(int order, long Ni, long n_pt_rows) {
    if ((order != 2 && order != 4) || Ni <= 0 || n_pt_rows < 0) return 0;
    return (long)((dm_workspace_bytes(Ni, n_pt_rows) + 255) / 256 * 256) + (long)dq_workspace_bytes(order, Ni, n_pt_rows);
}

// forward: ogg_dpole_band_workspace_bytes is used by the plan builder above
extern "C" int ogg_supergrid_pass_dev(int n_latlon, const ogg_latlon_band* latlon, long ni1, double lon0, double lenlon, double Re, int metrics,
                                      const ogg_bipolar_band* cap, const ogg_dpole_band* south_cap, void** events5, double* alg_bytes4,
                                      void* stream) {
    PassPlan P;   // plan + run: the one-shot form
    if (int e = build_pass_plan_any(n_latlon, latlon, ni1, lon0, lenlon, Re, metrics, cap, south_cap, P)) return e;
    return run_pass_plan_any(P, events5, alg_bytes4, stream);
}

extern "C" int ogg_supergrid_pass_plan_dev(int n_latlon, const ogg_latlon_band* latlon, long ni1, double lon0, double lenlon, double Re, int metrics,
                                           const ogg_bipolar_band* cap, const ogg_dpole_band* south_cap, void** plan_out) {
    OGG_REQUIRE(plan_out, OGG_EARG, "ogg_supergrid_pass_plan: null plan_out");
    *plan_out = nullptr;
    PassPipe* H = new (std::nothrow) PassPipe;
    OGG_REQUIRE(H, OGG_ENOMEM, "ogg_supergrid_pass_plan: out of host memory");
    if (int e = build_pass_pipe(n_latlon, latlon, ni1, lon0, lenlon, Re, metrics, cap, south_cap, *H)) {
        delete H;
        return e;
    }
    *plan_out = H;
    return OGG_OK;
}

extern "C" int ogg_supergrid_pass_run_dev(void* plan, void** events5, double* alg_bytes4, void* stream) {
    OGG_REQUIRE(plan, OGG_EARG, "ogg_supergrid_pass_run: null plan");
    return run_pass_pipe(*static_cast<PassPipe*>(plan), events5, alg_bytes4, stream);
}

extern "C" int ogg_supergrid_pass_plan_destroy(void* plan) {
    delete static_cast<PassPipe*>(plan);
    return OGG_OK;
}

extern "C" long ogg_supergrid_pass_plan_slots(const void* plan) { return plan ? static_cast<const PassPipe*>(plan)->n_slots : 0; }
extern "C" long ogg_supergrid_pass_plan_carried_runs(const void* plan) { return plan ? (long)static_cast<const PassPipe*>(plan)->carried_runs : 0; }

// flags of the plan's passes so far, after waiting for `stream`: bit 1 / bit 2 a look-back wait of the displaced-pole mesh / quadrature
// timed out, in either workspace slot
extern "C" int ogg_supergrid_pass_plan_flags_dev(const void* plan, int* flags, void* stream) {
    OGG_REQUIRE(plan && flags, OGG_EARG, "ogg_supergrid_pass_plan_flags: null pointer");
    const PassPipe& H = *static_cast<const PassPipe*>(plan);
    OGG_HIP_CHECK(hipStreamSynchronize(ogg::as_stream(stream)));
    int f = 0;
    for (int k = 0; k < H.n_slots; ++k) {
        const PassPlan& P = H.slot[k];
        unsigned w = 0u;
        if (P.B.n_dmesh > 0 && P.B.dm.ticket) {
            OGG_HIP_CHECK(hipMemcpy(&w, P.B.dm.ticket + 1, sizeof(w), hipMemcpyDeviceToHost));
            if (w) f |= 2;
        }
        if (P.dq_order && P.dq.ticket) {
            OGG_HIP_CHECK(hipMemcpy(&w, P.dq.ticket + 1, sizeof(w), hipMemcpyDeviceToHost));
            if (w) f |= 4;
        }
    }
    *flags = f;
    return OGG_OK;
}

extern "C" int ogg_tripolar_pass_events_dev(int n_latlon, const ogg_latlon_band* latlon, long ni1, double lon0, double lenlon, double Re,
                                            int metrics, const ogg_bipolar_band* cap, void** events4, double* alg_bytes3, void* stream) {
    void* ev5[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    double by4[4] = {0.0, 0.0, 0.0, 0.0};
    if (events4)
        for (int k = 0; k < 4; ++k) ev5[k] = events4[k];
    const int rc = ogg_supergrid_pass_dev(n_latlon, latlon, ni1, lon0, lenlon, Re, metrics, cap, nullptr, events4 ? ev5 : nullptr,
                                          alg_bytes3 ? by4 : nullptr, stream);
    if (alg_bytes3)
        for (int k = 0; k < 3; ++k) alg_bytes3[k] = by4[k];
    return rc;
}

extern "C" int ogg_tripolar_pass_dev(int n_latlon, const ogg_latlon_band* latlon, long ni1, double lon0, double lenlon, double Re,
                                     int metrics, const ogg_bipolar_band* cap, void* stream) {
    return ogg_supergrid_pass_dev(n_latlon, latlon, ni1, lon0, lenlon, Re, metrics, cap, nullptr, nullptr, nullptr, stream);
}
