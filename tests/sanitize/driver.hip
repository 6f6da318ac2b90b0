// Sanitizer driver for the library's HOST code (built and run by tests/test_sanitize_host.py with -fsanitize=address,undefined, host
// compilation only, against tests/sanitize/fake_hip_runtime.cpp).  Test infrastructure: nothing here is part of the product.
//
// It includes ogg_pass.hip as a unity translation unit to reach the plan builders in its anonymous namespace (plan_quad, plan_dquad,
// plan_dmesh, plan_latlon, build_pass_plan_any); the other translation units of the library are linked as objects.
#include <cinttypes>
#include <cstdint>
#include <random>
#include <thread>
#include <vector>

#include "../../ocean_model_grid_generator_amd/csrc/ogg_pass.hip"

extern "C" long fake_hip_launches(void);
extern "C" long fake_hip_copies(void);

namespace {

int g_fail = 0;
#define CHECK(cond, ...)                                         \
    do {                                                         \
        if (!(cond)) {                                           \
            fprintf(stderr, "FAIL %s:%d: ", __FILE__, __LINE__); \
            fprintf(stderr, __VA_ARGS__);                        \
            fprintf(stderr, "\n");                               \
            g_fail += 1;                                         \
        }                                                        \
    } while (0)

bool inside(const void* p, size_t bytes, const void* ws, size_t ws_bytes) {
    const uintptr_t a = (uintptr_t)p, b = (uintptr_t)ws;
    return a >= b && a + bytes <= b + ws_bytes;
}

// ---- 1. the staging layer: upload / download of awkward sizes through the two pinned buffers ------------------------------------------
void staging_round_trips() {
    const size_t MiB = 1u << 20;
    const size_t sizes[] = {8, 16, 4096, 8 * MiB - 8, 8 * MiB, 8 * MiB + 8, 16 * MiB - 8, 16 * MiB, 16 * MiB + 8, 20 * MiB, 24 * MiB + 8, 8};
    std::mt19937_64 rng(7);
    for (size_t bytes : sizes) {
        const long n = (long)(bytes / 8);
        std::vector<double> x((size_t)n), want;
        for (double& v : x) v = (double)(rng() >> 11);
        want = x;
        // in place through the host-pointer layer: upload, a launch (a no-op here), download -- the same values must come back, whatever the
        // order in which the deferred copies of the fake runtime ran
        const int rc = ogg_monotonic_bounding(1, n, x.data(), 0.0);
        CHECK(rc == OGG_OK, "ogg_monotonic_bounding(%ld) -> %d (%s)", n, rc, ogg_last_error());
        CHECK(x == want, "round trip of %zu bytes changed the data", bytes);
    }
    // zero rows: nothing to move
    double one = 1.0;
    CHECK(ogg_monotonic_bounding(0, 5, &one, 0.0) == OGG_OK && one == 1.0, "zero-row call");
    // several buffers in one call, uneven sizes (the arena grows by blocks within a call and shrinks to one block of the call's size)
    for (int rep = 0; rep < 3; ++rep) {
        const long n = 1000003 + 17 * rep;
        std::vector<double> a((size_t)n, 1.0), b((size_t)n, 2.0), c((size_t)n);
        CHECK(ogg_mdist(n, a.data(), b.data(), c.data()) == OGG_OK, "ogg_mdist(%ld): %s", n, ogg_last_error());
    }
    // a call above the arena's keep limit (64 MiB of device scratch): released at the end of the call, the next small call starts afresh
    {
        const long n = (40u << 20) / 8;
        std::vector<double> a((size_t)n, 1.0), b((size_t)n, 2.0), c((size_t)n);
        CHECK(ogg_mdist(n, a.data(), b.data(), c.data()) == OGG_OK, "ogg_mdist(40 MiB x 3): %s", ogg_last_error());
        double s1 = 3.0, s2 = 4.0, s3 = 0.0;
        CHECK(ogg_mdist(1, &s1, &s2, &s3) == OGG_OK, "small call after a large one");
    }
    // the per-device staging state: the same calls on "device" 1
    CHECK(ogg_set_device(1) == OGG_OK, "ogg_set_device(1)");
    {
        std::vector<double> x(300000, 5.0), want = x;
        CHECK(ogg_monotonic_bounding(3, 100000, x.data(), 0.0) == OGG_OK && x == want, "round trip on device 1");
    }
    CHECK(ogg_set_device(0) == OGG_OK, "ogg_set_device(0)");
    // errors: messages, no leak (ASan's leak check at exit), no crash
    CHECK(ogg_monotonic_bounding(1, 0, &one, 0.0) == OGG_EARG, "bad argument accepted");
    CHECK(ogg_bipolar_cap_metrics_quad(7, 8, 4, 64.0, -300.0, 0.2, 1.0, &one, &one, &one) == OGG_EORDER && strstr(ogg_last_error(), "Uncoded order"),
          "order 7: %s", ogg_last_error());
}

// ---- 2. the host-pointer layer from two threads (one process-wide mutex per device serialises them) ---------------------------------------
void two_threads() {
    auto work = [](int id) {
        std::vector<double> x(200000 + 1000 * id), want;
        for (size_t k = 0; k < x.size(); ++k) x[k] = (double)(k * (id + 1));
        want = x;
        for (int rep = 0; rep < 25; ++rep) {
            if (ogg_monotonic_bounding(2, (long)x.size() / 2, x.data(), 0.0) != OGG_OK || x != want) {
                fprintf(stderr, "FAIL thread %d rep %d\n", id, rep);
                __atomic_add_fetch(&g_fail, 1, __ATOMIC_SEQ_CST);
                return;
            }
            std::vector<double> dx(9 * 8), dy(8 * 9), da(8 * 8);
            if (ogg_bipolar_cap_metrics_quad(5, 8, 8, 64.0, -300.0, 0.23, 6371e3, dx.data(), dy.data(), da.data()) != OGG_OK) {
                __atomic_add_fetch(&g_fail, 1, __ATOMIC_SEQ_CST);
                return;
            }
        }
    };
    std::thread a(work, 0), b(work, 1);
    a.join();
    b.join();
}

// ---- 3. workspace carving of the plan builders: every table pointer inside [ws, ws + bytes) ------------------------------------------
template <int N>
void check_quad_plan(long nx, long ny, long j0, long n_cell_rows, bool top, int symmetry) {
    constexpr int M = N - 1;
    const size_t bytes = quad_workspace_bytes<N>(nx, ny, n_cell_rows);
    std::vector<char> ws(bytes);
    QuadParams p{};
    p.nx = nx, p.ny = ny, p.lat0_bp = 64.0, p.lon_bp = -300.0, p.rp = 0.23, p.Re = 6371e3, p.j0 = j0, p.q = make_nodes(N);
    QuadPlan q{};
    const int rc = plan_quad<N>(p, n_cell_rows + (top ? 1 : 0), n_cell_rows, 4000.0, symmetry, ws.data(), (long)bytes, q);
    CHECK(rc == OGG_OK, "plan_quad<%d>(%ld, %ld): %s", N, nx, ny, ogg_last_error());
    if (rc != OGG_OK) return;
    CHECK(inside(q.p.row_tab, (size_t)(M * ny + 2) * sizeof(BpRow), ws.data(), bytes), "row_tab outside the workspace (%ld x %ld)", nx, ny);
    CHECK(inside(q.p.col_tab, (size_t)(M * nx + 1) * sizeof(BpCol), ws.data(), bytes), "col_tab outside the workspace");
    CHECK(inside(q.p.fix_count, 16, ws.data(), bytes) && inside(q.p.ll_claims, QUAD_LL_CLAIM_WORDS * sizeof(unsigned), ws.data(), bytes),
          "counters outside the workspace");
    CHECK(inside(q.p.fix_list, (size_t)(n_cell_rows > 0 ? n_cell_rows : 0) * nx * sizeof(unsigned), ws.data(), bytes), "fix-up list outside the workspace");
    CHECK(inside(q.top_buf, (size_t)nx * sizeof(double), ws.data(), bytes), "top row buffer outside the workspace");
    // no two of them overlap
    const uintptr_t b[6] = {(uintptr_t)q.p.row_tab, (uintptr_t)q.p.col_tab, (uintptr_t)q.p.fix_count, (uintptr_t)q.p.ll_claims, (uintptr_t)q.p.fix_list,
                            (uintptr_t)q.top_buf};
    const size_t sz[6] = {(size_t)(M * ny + 2) * sizeof(BpRow), (size_t)(M * nx + 1) * sizeof(BpCol), 16, QUAD_LL_CLAIM_WORDS * sizeof(unsigned),
                          (size_t)(n_cell_rows > 0 ? n_cell_rows : 0) * nx * sizeof(unsigned), (size_t)nx * sizeof(double)};
    for (int i = 0; i < 6; ++i)
        for (int k = i + 1; k < 6; ++k) CHECK(b[i] + sz[i] <= b[k] || b[k] + sz[k] <= b[i], "workspace parts %d and %d overlap (%ld x %ld)", i, k, nx, ny);
    // the ranges of the two strip grids cover the band's cell rows once
    long covered = 0;
    if (q.has_fast) {
        covered += q.fast.row_end - q.fast.row_begin;
        CHECK(q.fast.gx > 0 && q.fast.gy > 0 && q.fast.cols.g_end[2] == q.fast.gx, "empty fast grid");
    }
    if (q.has_guard) {
        covered += q.guard.row_end - q.guard.row_begin;
        CHECK(q.guard.cols.sym == 0, "the guarded rows must not be mirrored");
    }
    CHECK(covered == n_cell_rows, "cell rows covered %ld of %ld", covered, n_cell_rows);
}

void check_dpole_plans(long ni, long nj, long j0, long n_pt_rows, int order, int arc, int symmetry) {
    const long n_cell = (j0 + n_pt_rows == nj + 1) ? n_pt_rows - 1 : n_pt_rows;
    const long M = order - 1, NV = order + 1;
    const DpGeom g{ni, nj, -300.0, -78.0, 80.0, 0.2};
    const size_t mesh_bytes = (dm_workspace_bytes(ni, n_pt_rows) + 255) / 256 * 256, quad_bytes = dq_workspace_bytes(order, ni, n_cell);
    CHECK((long)(mesh_bytes + quad_bytes) <= ogg_dpole_band_workspace_bytes(order, ni, n_pt_rows), "band workspace smaller than its parts");
    std::vector<char> ws(mesh_bytes + quad_bytes);
    DpMeshParams dm{};
    CHECK(plan_dmesh(g, j0, n_pt_rows, nullptr, nullptr, nullptr, ws.data(), (long)mesh_bytes, dm) == OGG_OK, "plan_dmesh: %s", ogg_last_error());
    CHECK(inside(dm.ticket, 16, ws.data(), mesh_bytes) && inside(dm.words, (size_t)(n_pt_rows * dm.n_strips) * 8, ws.data(), mesh_bytes),
          "mesh look-back words outside their part of the workspace (%ld x %ld rows)", ni, n_pt_rows);
    DpQuadParams dq{};
    const int rc = plan_dquad(arc, order, g, 6371e3, j0, n_pt_rows, n_cell, nullptr, nullptr, nullptr, ws.data() + mesh_bytes, (long)quad_bytes,
                              make_nodes(order), dq, symmetry);
    CHECK(rc == OGG_OK, "plan_dquad: %s", ogg_last_error());
    if (rc != OGG_OK) return;
    const char* w = ws.data() + mesh_bytes;
    CHECK(inside(dq.ticket, 16, w, quad_bytes), "ticket");
    CHECK(inside(dq.row_tab, (size_t)(NV * dq.n_rows) * 8, w, quad_bytes) && inside(dq.col_tab, (size_t)(NV * 2 * dq.n_cols) * 8, w, quad_bytes),
          "displaced-pole tables outside the workspace (%ld x %ld, order %d)", ni, n_cell, order);
    if (arc == DP_ARC_LITERAL)
        CHECK(dq.words && inside(dq.words, (size_t)(dq.n_chunks * (M * dq.rows_per_chunk + 1) * dq.n_strips) * 8, w, quad_bytes),
              "look-back words of the literal quadrature outside the workspace (%ld cell rows, %ld per chunk)", n_cell, dq.rows_per_chunk);
    else
        CHECK(dq.words == nullptr, "chord form with look-back words");
    CHECK(dq.n_src_cols >= 2 && dq.u_first >= 0 && dq.u_first + dq.n_src_cols <= dq.n_cols, "strip columns %ld + %ld of %ld", dq.u_first, dq.n_src_cols,
          dq.n_cols);
    CHECK(!(dq.sym && arc == DP_ARC_LITERAL), "the literal form must not be mirrored");
}

void check_pass_plan(double r, bool displaced, int world, int rank, int symmetry) {
    // a rank's bands of a tripolar grid of inverse resolution r, sizes as SupergridPlan derives them (without the Mercator parity fixes)
    const long Ni = (long)(r * 720), ni1 = Ni + 1;
    const long n_merc = (long)(r * 350) + 1, n_so = (long)(r * 55) + 1, Nj_bp = (long)(r * 120), n_sc = displaced ? (long)(r * 40) * 7 / 4 : (long)(r * 24);
    auto share = [&](long n, long& lo, long& hi) { lo = n * rank / world, hi = n * (rank + 1) / world; };
    std::vector<std::vector<double>> keep;
    auto buf = [&](long rows, long cols) {
        keep.emplace_back((size_t)(rows > 0 ? rows : 0) * cols + 1);
        return keep.back().data();
    };
    ogg_latlon_band ll[3];
    int n_ll = 0;
    long lo, hi;
    auto latlon = [&](int kind, long n_axis, long long y0) {
        share(n_axis, lo, hi);
        if (hi <= lo) return;
        ogg_latlon_band b{};
        b.axis_kind = kind, b.a0 = -78.0, b.len = 11.0, b.denom = (double)(n_axis - 1), b.y0 = y0, b.k0 = lo, b.n_pt_rows = hi - lo;
        b.n_cell_rows = (hi < n_axis ? hi : n_axis - 1) - lo;
        b.x = buf(hi - lo, ni1), b.y = buf(hi - lo, ni1), b.angle = buf(hi - lo, ni1), b.dx = buf(hi - lo, Ni);
        b.dy = buf(b.n_cell_rows, ni1), b.area = buf(b.n_cell_rows, Ni);
        ll[n_ll++] = b;
    };
    if (!displaced) latlon(0, n_sc + 1, 0);
    latlon(0, n_so, 0);
    latlon(1, n_merc, -(long long)(r * 182));
    ogg_bipolar_band cap{};
    share(Nj_bp + 1, lo, hi);
    std::vector<char> cap_ws((size_t)ogg_bipolar_quad_workspace_bytes(5, Ni, Nj_bp));
    cap.Ni = Ni, cap.Nj = Nj_bp, cap.lat0_bp = 64.0, cap.lon_bp = -300.0, cap.rp = 0.23, cap.Re = 6371e3, cap.order = 5, cap.symmetry = symmetry;
    cap.j0 = lo, cap.n_pt_rows = hi - lo, cap.n_cell_rows = (hi < Nj_bp + 1 ? hi : Nj_bp) - lo;
    cap.x = buf(hi - lo, ni1), cap.y = buf(hi - lo, ni1), cap.angle = buf(hi - lo, ni1), cap.dx = buf(hi - lo, Ni);
    cap.dy = buf(cap.n_cell_rows, ni1), cap.area = buf(cap.n_cell_rows, Ni), cap.workspace = cap_ws.data(), cap.workspace_bytes = (long)cap_ws.size();
    ogg_dpole_band sc{};
    std::vector<char> sc_ws;
    if (displaced) {
        const long row0 = (long)(0.49 * n_sc) + 1, kept = n_sc + 1 - row0;
        share(kept, lo, hi);
        sc.Ni = Ni, sc.Nj = n_sc, sc.lon0 = -300.0, sc.lat0 = -78.0, sc.lon_dp = 80.0, sc.r_dp = 0.2, sc.Re = 6371e3, sc.order = 4;
        sc.arc_form = OGG_DP_ARC_CHORD, sc.symmetry = symmetry;
        sc.j0 = row0 + lo, sc.n_pt_rows = hi - lo, sc.n_cell_rows = (hi < kept ? hi : kept - 1) - lo;
        sc_ws.resize((size_t)ogg_dpole_band_workspace_bytes(4, Ni, sc.n_pt_rows) + 16);
        sc.x = buf(hi - lo, ni1), sc.y = buf(hi - lo, ni1), sc.angle = buf(hi - lo, ni1), sc.dx = buf(hi - lo, Ni);
        sc.dy = buf(sc.n_cell_rows, ni1), sc.area = buf(sc.n_cell_rows, Ni), sc.workspace = sc_ws.data(), sc.workspace_bytes = (long)sc_ws.size();
    }
    PassPlan P;
    const int rc = build_pass_plan_any(n_ll, ll, ni1, -300.0, 360.0, 6371e3, 1, cap.n_pt_rows > 0 ? &cap : nullptr,
                                       (displaced && sc.n_pt_rows > 0) ? &sc : nullptr, P);
    CHECK(rc == OGG_OK, "build_pass_plan_any(r = %g, rank %d of %d): %s", r, rank, world, ogg_last_error());
    if (rc != OGG_OK) return;
    if (cap.n_pt_rows > 0 && P.have_quad) {
        CHECK(inside(P.B.q.row_tab, 8, cap_ws.data(), cap_ws.size()) && inside(P.B.q.col_tab, 8, cap_ws.data(), cap_ws.size()) &&
                  inside(P.B.q.fix_list, (size_t)cap.n_cell_rows * Ni * 4, cap_ws.data(), cap_ws.size()),
              "the pass carved the cap's tables outside its workspace");
        CHECK(P.B.n_fast == (P.qp.has_fast ? (long)P.qp.fast.gx * P.qp.fast.gy : 0) && P.B.n_guard == (P.qp.has_guard ? (long)P.qp.guard.gx * P.qp.guard.gy : 0),
              "strip workgroup counts");
    }
    if (displaced && sc.n_pt_rows > 0) {
        CHECK(inside(P.B.dm.words, 8, sc_ws.data(), sc_ws.size()) && inside(P.dq.row_tab, 8, sc_ws.data(), sc_ws.size()) &&
                  inside(P.dq.col_tab + 2 * 5 * P.dq.n_cols - 1, 8, sc_ws.data(), sc_ws.size()),
              "the pass carved the southern cap's tables outside its workspace");
    }
    // the launch sizes add up to the roles' workgroups, and running the plan issues exactly its launches
    CHECK(P.nb == (unsigned)(P.B.share.n_wg + P.B.n_mesh + P.B.n_dmesh + P.B.n_guard + P.B.n_fast + P.B.n_dquad + P.B.share.n_help) || !P.launch_b,
          "launch B: %u workgroups", P.nb);
    const long before = fake_hip_launches();
    CHECK(run_pass_plan_any(P, nullptr, nullptr, nullptr) == OGG_OK, "run_pass_plan_any: %s", ogg_last_error());
    const long launched = fake_hip_launches() - before;
    CHECK(launched == (P.na > 0) + (P.launch_b ? 1 : 0) + ((P.have_quad && (P.qp.has_guard || (P.qp.p.top_src && P.qp.has_top))) ? 1 : 0) + (P.dq_literal ? 1 : 0),
          "a pass issued %ld launches", launched);
}

// the plan HANDLE (two workspace slots of its own): build, run a few passes, destroy -- ASan sees the slots' allocation and release
void plan_handle_life_cycle() {
    const long Ni = 1440, ni1 = Ni + 1, Nj = 238;
    std::vector<double> f((size_t)(Nj + 1) * ni1 * 6 + 64);
    std::vector<char> ws((size_t)ogg_bipolar_quad_workspace_bytes(5, Ni, Nj));
    ogg_bipolar_band cap{};
    cap.Ni = Ni, cap.Nj = Nj, cap.lat0_bp = 64.97, cap.lon_bp = -300.0, cap.rp = 0.22, cap.Re = 6371e3, cap.order = 5;
    cap.j0 = 0, cap.n_pt_rows = Nj + 1, cap.n_cell_rows = Nj;
    double* b = f.data();
    cap.x = b, cap.y = b + (Nj + 1) * ni1, cap.angle = b + 2 * (Nj + 1) * ni1, cap.dx = b + 3 * (Nj + 1) * ni1, cap.dy = b + 4 * (Nj + 1) * ni1,
    cap.area = b + 5 * (Nj + 1) * ni1;
    cap.workspace = ws.data(), cap.workspace_bytes = (long)ws.size();
    ogg_latlon_band ll{};
    std::vector<double> g((size_t)100 * ni1 * 6);
    ll.axis_kind = 0, ll.a0 = -78.0, ll.len = 11.0, ll.denom = 110.0, ll.k0 = 0, ll.n_pt_rows = 100, ll.n_cell_rows = 100;
    ll.x = g.data(), ll.y = ll.x + 100 * ni1, ll.dx = ll.y + 100 * ni1, ll.dy = ll.dx + 100 * ni1, ll.area = ll.dy + 100 * ni1, ll.angle = ll.area + 100 * ni1;
    for (int rep = 0; rep < 3; ++rep) {
        void* h = nullptr;
        if (rep == 1) setenv("OGG_PASS_LL_TABLE", "0", 1);   // the strips evaluate their rows' scalars themselves: no table, no table role
        else unsetenv("OGG_PASS_LL_TABLE");
        CHECK(ogg_supergrid_pass_plan_dev(1, &ll, ni1, -300.0, 360.0, 6371e3, 1, &cap, nullptr, &h) == OGG_OK && h, "plan: %s", ogg_last_error());
        if (!h) return;
        CHECK(ogg_supergrid_pass_plan_slots(h) == 2, "slots");
        {   // the handle's row tables: one per slot, the plan's own, read by the strips of that slot's launch B and filled by its table roles
            const PassPipe& H = *static_cast<const PassPipe*>(h);
            const bool want = rep != 1;   // (rep 1 is built with OGG_PASS_LL_TABLE=0)
            for (int k = 0; k < 2; ++k) {
                const PassPlan& P = H.slot[k];
                CHECK((H.own_row_tab[k] != nullptr) == want, "slot %d: row table allocated %d, wanted %d", k, H.own_row_tab[k] != nullptr, (int)want);
                CHECK(P.B.ll.row_tab == H.own_row_tab[k] && P.A.ll.row_tab == H.own_row_tab[k], "slot %d: the launches read another table than the slot's", k);
                CHECK(P.A.n_ll_tab == (want ? latlon_row_table_blocks(P.A.ll) : 0), "slot %d: %ld table workgroups", k, P.A.n_ll_tab);
                CHECK(P.A.ll.row0[P.A.ll.n_bands] == ll.n_pt_rows + 1, "slot %d: %ld table rows for %ld point rows", k, P.A.ll.row0[P.A.ll.n_bands], ll.n_pt_rows);
                CHECK(P.B.ll.rows_per_block == (want ? 6 : P.B.ll.rows_per_block) && P.A.ll.rows_per_block == P.B.ll.rows_per_block &&
                          P.A.ll.strip0[1] == P.B.ll.strip0[1] && P.B.ll.strip0[1] == (ll.n_pt_rows + P.B.ll.rows_per_block - 1) / P.B.ll.rows_per_block,
                      "slot %d: strips of %d rows, %ld / %ld of them", k, P.B.ll.rows_per_block, P.A.ll.strip0[1], P.B.ll.strip0[1]);
                CHECK(P.na == (unsigned)(P.A.share.n_wg + P.A.n_tab + P.A.n_dq_tab + P.A.n_dm_reset + P.A.n_ll_tab + P.A.n_mesh), "slot %d: launch A's grid", k);
            }
            CHECK(H.slot[0].B.ll.row_tab != H.slot[1].B.ll.row_tab || !want, "the two slots share a row table");
        }
        for (int k = 0; k < 5; ++k) CHECK(ogg_supergrid_pass_run_dev(h, nullptr, nullptr, nullptr) == OGG_OK, "run %d", k);
        CHECK(ogg_supergrid_pass_plan_carried_runs(h) == 4, "carried runs %ld", ogg_supergrid_pass_plan_carried_runs(h));
        int flags = -1;
        CHECK(ogg_supergrid_pass_plan_flags_dev(h, &flags, nullptr) == OGG_OK && flags == 0, "flags");
        CHECK(ogg_supergrid_pass_plan_destroy(h) == OGG_OK, "destroy");
    }
    // a workspace that is too small is refused with a message, not overrun
    cap.workspace_bytes = 1000;
    void* h = nullptr;
    CHECK(ogg_supergrid_pass_plan_dev(1, &ll, ni1, -300.0, 360.0, 6371e3, 1, &cap, nullptr, &h) == OGG_EARG && !h && strstr(ogg_last_error(), "workspace too small"),
          "small workspace: %s", ogg_last_error());
}

}  // namespace

int main() {
    if (getenv("OGG_SANITIZE_SELFTEST")) {   // the test harness checks that the sanitizer is live: this overrun must end the run
        volatile char* p = new char[8];
        p[8] = 1;
        delete[] p;
    }
    staging_round_trips();
    two_threads();
    // the five BASELINE sizes (Ni x Nj of the bipolar cap), whole caps and bands, every order, mirrored and not ...
    const long caps[][2] = {{1440, 238}, {2880, 480}, {5760, 960}, {5760, 948}, {11520, 1920}};
    for (auto& c : caps)
        for (int sym = 0; sym < 2; ++sym) {
            check_quad_plan<5>(c[0], c[1], 0, c[1], true, sym);
            check_quad_plan<5>(c[0], c[1], c[1] / 2, c[1] - c[1] / 2, true, sym);
            check_quad_plan<4>(c[0], c[1], c[1] / 8, c[1] / 8, false, sym);
        }
    // ... and twenty random ones (Ni need not be a multiple of anything)
    std::mt19937 rng(5);
    for (int k = 0; k < 20; ++k) {
        const long nx = 4 + rng() % 3000, ny = 2 + rng() % 400, j0 = rng() % ny, n = 1 + rng() % (ny - j0);
        const bool top = (j0 + n == ny);
        switch (rng() % 4) {
            case 0: check_quad_plan<2>(nx, ny, j0, n, top, k & 1); break;
            case 1: check_quad_plan<3>(nx, ny, j0, n, top, k & 1); break;
            case 2: check_quad_plan<4>(nx, ny, j0, n, top, k & 1); break;
            default: check_quad_plan<5>(nx, ny, j0, n, top, k & 1);
        }
    }
    // displaced-pole caps: config 4 (5760 x 560, rows 276-560), OM4 (2880 x 280, rows 220-280), small and odd ones; both arc forms
    for (int sym = 0; sym < 2; ++sym)
        for (int arc = 0; arc < 2; ++arc) {
            check_dpole_plans(5760, 560, 276, 285, 4, arc, sym);
            check_dpole_plans(2880, 280, 220, 61, 4, arc, sym);
            check_dpole_plans(360, 70, 36, 35, 2, arc, sym);
            check_dpole_plans(361, 70, 0, 5, 4, arc, sym);
            check_dpole_plans(1000, 33, 7, 1, 4, arc, sym);
        }
    for (int k = 0; k < 20; ++k) {
        const long ni = 4 + rng() % 4000, nj = 2 + rng() % 300, j0 = rng() % nj, n = 1 + rng() % (nj + 1 - j0);
        check_dpole_plans(ni, nj, j0, n, (rng() & 1) ? 4 : 2, (int)(rng() & 1), (int)(rng() & 1));
    }
    // whole passes: the BASELINE resolutions, one rank and shares of 2, 3, 8 ranks, with and without the displaced pole
    const double res[] = {2.0, 4.0, 8.0, 16.0, 0.5};
    for (double r : res)
        for (int displaced = 0; displaced < 2; ++displaced) {
            check_pass_plan(r, displaced, 1, 0, OGG_SYM_MIRROR);
            check_pass_plan(r, displaced, 8, 7, OGG_SYM_DEFAULT);
            check_pass_plan(r, displaced, 3, 1, OGG_SYM_NONE);
            check_pass_plan(r, displaced, 2, 0, OGG_SYM_MIRROR);
        }
    plan_handle_life_cycle();
    if (g_fail) {
        fprintf(stderr, "%d check(s) failed\n", g_fail);
        return 1;
    }
    printf("sanitize driver ok: %ld launches, %ld deferred copies\n", fake_hip_launches(), fake_hip_copies());
    return 0;
}
