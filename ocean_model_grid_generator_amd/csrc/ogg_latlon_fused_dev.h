// Device code of ogg_latlon_fused.hip (also included by the fused pass, ogg_pass.hip).
// K1+K2 fused for sub-grids that are lat-lon BY CONSTRUCTION (Mercator, Southern Ocean, regular southern cap):
// x[j][i] = lon[i], y[j][i] = lat[j] (OGG:430-432, 840-841), so every term of the MIDAS metrics (OGG:695-713) and of
// angle_x (OGG:725-728) factors into a per-row scalar times a per-column scalar.  The kernel evaluates the 1-D axes
// itself (OGG:336, 431, 834-835: one atan(sinh) or one multiply-divide-add per row / column), the transcendental functions
// once per ROW in LDS, and streams out all six fields of up to four sub-grid bands in ONE launch: 48 B written per cell,
// nothing read from HBM -- the algorithmic minimum of SURVEY 8(d).
//
// It is bit-identical to tile_latlon_kernel followed by midas_angle_kernel, because it performs the same operations on
// the same operands and only hoists those that do not depend on i (or on j):
//   dy_i = (lat_j - lat_j) PI/180 = 0, so dx = Re sqrt(0 + (dlam_i cos(lv_j))^2);  lv_j = (0.5 (lat_j+lat_j)) PI/180 = lat_j PI/180
//   dx_j = mdist(lon_i, lon_i) PI/180 = 0, so dy = Re sqrt(dphi_j^2 + 0) is a per-row constant
//   area = Re^2 ((0.5 (dlam_i + dlam_i)) (sin lv_{j+1} - sin lv_j))
//   angle = atan2(+0, (lon_{i+1} - lon_{i-1}) cos(lat_j PI/180)) / (PI/180): the IEEE value of atan2(+0, p) is +0 for
//           p > 0 or p = +0 and pi for p < 0 or p = -0.
// tests/test_gpu_pipeline.py checks the bit-identity against the generic stencil kernel.
#pragma once
#include <cmath>
#include <cstdlib>

#include "../../include/ogg_hip.h"
#include "ogg_common.h"
#include "ogg_math.h"

namespace {
using namespace ogg;

constexpr int LF_TX = 256;
constexpr int LF_ROWS = 32;  // maximum rows per strip (LDS row table); small bands use fewer (>= 8) so that there are enough strips
constexpr int LF_MAX_BANDS = 4;

struct RowScalars {
    double lat, sl, cl, dy;
};

struct FusedParams {
    int n_bands;
    ogg_latlon_band band[LF_MAX_BANDS];
    long strip0[LF_MAX_BANDS + 1];  // prefix sum of the row strips of the bands
    long ni1;
    double lon0, lenlon, Ni;        // lon[i] = lon0 + (i*lenlon)/Ni   (OGG:431, 834)
    int rows_per_block;
    double Re, Re2;
    int metrics;
    // the per-row scalars of every band as a table in device memory (NULL: the strips compute their rows' scalars themselves):
    // band k's rows 0 .. n_pt_rows at row0[k] (latlon_row_table_body fills it; latlon_fused_body<NT, true> reads it)
    const RowScalars* row_tab;
    long row0[LF_MAX_BANDS + 1];
};

// latitude of axis index k (OGG:336 / OGG:835 / explicit)
OGG_DEV double axis_lat(const ogg_latlon_band& b, long k, double Ni) {
    if (b.axis_kind == 0) return b.a0 + ((double)k * b.len) / b.denom;
    if (b.axis_kind == 1) {
        const double R = Ni / (2 * kPi);
        return atan(sinh((double)(b.y0 + k) / R)) * k180Pi;
    }
    return b.lat1d[k];
}

// The scalars of row j of band b: latitude, sin / cos of it, and dy of cell row j (which needs lat_{j+1}) -- the operations the strips
// perform when they have no table (latlon_fused_body<NT, false>), on the same operands: the same bits.  Rows that nothing reads are zero.
OGG_DEV RowScalars latlon_row_scalars(const FusedParams& f, const ogg_latlon_band& b, long j) {
    const long n_cell_rows = f.metrics ? b.n_cell_rows : 0;
    RowScalars r = {0.0, 0.0, 0.0, 0.0};
    if (j < b.n_pt_rows || j - 1 < n_cell_rows) {      // the same rows latlon_fused_body gives scalars to
        r.lat = axis_lat(b, b.k0 + j, f.Ni);
        const double lv = (0.5 * (r.lat + r.lat)) * kPi180;
        sincos(lv, &r.sl, &r.cl);
        if (j < n_cell_rows) {                          // dy of cell row j needs lat_{j+1}
            const double dyj = (axis_lat(b, b.k0 + j + 1, f.Ni) - r.lat) * kPi180;
            r.dy = f.Re * sqrt(dyj * dyj + 0.0);
        }
    }
    return r;
}

// workgroup b of n_wg: the row table of all bands (f.row_tab), one thread per row.  A role of the fused pass's table launch (ogg_pass.hip):
// atan(sinh) and sincos want 80 VGPRs more than the strips' row loop, and a strip role without them lets launch B run at twice the
// occupancy.
OGG_DEV void latlon_row_table_body(const FusedParams& f, long b, long n_wg) {
    const long n_rows = f.row0[f.n_bands];
    RowScalars* tab = const_cast<RowScalars*>(f.row_tab);
    for (long k = b * LF_TX + threadIdx.x; k < n_rows; k += n_wg * LF_TX) {
        int bi = 0;
        while (bi + 1 < f.n_bands && k >= f.row0[bi + 1]) ++bi;
        tab[k] = latlon_row_scalars(f, f.band[bi], k - f.row0[bi]);
    }
}
inline long latlon_row_table_blocks(const FusedParams& f) { return f.n_bands ? (f.row0[f.n_bands] + LF_TX - 1) / LF_TX : 0; }

// column-only quantities (OGG:696, 713, 725-727) of column i
struct ColScalars {
    double lon_c, dlam, hdlam, xdiff;
};

OGG_DEV ColScalars column_scalars(const FusedParams& p, long i) {
    const long ni1 = p.ni1, ni = ni1 - 1;
    const long ic = (i < ni1) ? i : ni;
    ColScalars c;
    c.lon_c = p.lon0 + ((double)ic * p.lenlon) / p.Ni;
    const double lon_r = p.lon0 + ((double)((ic + 1 < ni1) ? ic + 1 : ic) * p.lenlon) / p.Ni;
    const double lon_l = p.lon0 + ((double)((ic > 0) ? ic - 1 : 0) * p.lenlon) / p.Ni;
    c.dlam = mdist(lon_r, c.lon_c) * kPi180;
    c.hdlam = 0.5 * (c.dlam + c.dlam);
    if (i == 0)
        c.xdiff = lon_r - c.lon_c;
    else if (i == ni)
        c.xdiff = c.lon_c - lon_l;
    else
        c.xdiff = lon_r - lon_l;
    return c;
}

// Two adjacent columns in one 16-byte store.  Rows of odd length are only 8-byte aligned; pairing the columns per row so
// that every store is 16-byte aligned was tried and is SLOWER: what limits a lat-lon workgroup is the number of
// instructions per row, not the store width.
typedef double dbl2 __attribute__((ext_vector_type(2), aligned(8)));

// FULL: both columns exist (one unconditional 16-byte store); else each one on its own flag.
// NT: non-temporal stores (`global_store_dwordx4 ... nt`).  Alone, the kernel is 5 % SLOWER with them (0.197-0.227 against 0.189-0.211 ms
// on one box, whose write path slows down as it warms up); as a role of the fused pass, next to cap workgroups that keep the
// socket at its power limit, the pass is 6-15 % FASTER with them and no longer follows that drift (0.249-0.256 against 0.264-0.299 ms,
// scripts/ab_time.py on one box): the pass sets NT, the stand-alone launch does not.
template <bool FULL, bool NT>
OGG_DEV void store2(double* q, double a, double b, bool first, bool second) {
    if (FULL) {
        dbl2 v;
        v.x = a, v.y = b;
        if (NT)
            __builtin_nontemporal_store(v, reinterpret_cast<dbl2*>(q));
        else
            *reinterpret_cast<dbl2*>(q) = v;
    } else {
        if (NT) {
            if (first) __builtin_nontemporal_store(a, q);
            if (second) __builtin_nontemporal_store(b, q + 1);
        } else {
            if (first) q[0] = a;
            if (second) q[1] = b;
        }
    }
}

// angle_x of a lat-lon mesh: atan2(+0, pa)/PI_180 with pa = (lon difference) cos(lat).  IEEE atan2(+0, pa) is +0 for pa > 0
// or pa = +0 and pi for pa < 0 or pa = -0, i.e. it follows the SIGN BIT of pa; NaN propagates.  No branch.
OGG_DEV double latlon_angle(double pa) {
    const double r = (__double2hiint(pa) < 0) ? kPi / kPi180 : 0.0 / kPi180;
    return (pa != pa) ? pa : r;
}

// The rows js .. js+nrows-1 of one band for this thread's two columns.  FULL: both columns exist and both have a right
// neighbour -- every thread but the last one or two of a row -- so the loop has no per-lane branch: what limits a lat-lon
// workgroup is the instruction count per row, not the store width.
template <bool FULL, bool NT>
OGG_DEV void latlon_rows(const FusedParams& p, const ogg_latlon_band& b, const RowScalars* s_row, long js, int nrows, long n_cell_rows,
                         long i0, const ColScalars& c0, const ColScalars& c1) {
    const long ni1 = p.ni1, ni = ni1 - 1;
    const bool pt1 = i0 + 1 < ni1, ce0 = i0 < ni;  // !FULL only: second point column exists; first column has a right neighbour
    double* __restrict__ px = b.x + (js * ni1 + i0);
    double* __restrict__ py = b.y + (js * ni1 + i0);
    double* __restrict__ pa = b.angle + (js * ni1 + i0);
    double* __restrict__ pdx = b.dx + (js * ni + i0);
    double* __restrict__ pdy = b.dy + (js * ni1 + i0);
    double* __restrict__ par = b.area + (js * ni + i0);
    for (int r = 0; r < nrows; ++r) {
        const long j = js + r;
        const RowScalars rs = s_row[r];
        store2<FULL, NT>(px, c0.lon_c, c1.lon_c, true, pt1);
        store2<FULL, NT>(py, rs.lat, rs.lat, true, pt1);
        store2<FULL, NT>(pa, latlon_angle(c0.xdiff * rs.cl), latlon_angle(c1.xdiff * rs.cl), true, pt1);
        if (p.metrics) {
            // dx = Re sqrt(0 + t^2) with t = dlam cos(lv): sqrt(RN(t^2)) == |t| in binary floating point (no over/underflow here)
            store2<FULL, NT>(pdx, p.Re * fabs(c0.dlam * rs.cl), p.Re * fabs(c1.dlam * rs.cl), ce0, false);
            if (j < n_cell_rows) {
                store2<FULL, NT>(pdy, rs.dy, rs.dy, true, pt1);
                const double ds = s_row[r + 1].sl - rs.sl;
                store2<FULL, NT>(par, p.Re2 * (c0.hdlam * ds), p.Re2 * (c1.hdlam * ds), ce0, false);
            }
        }
        px += ni1, py += ni1, pa += ni1, pdy += ni1;
        pdx += ni, par += ni;
    }
}

// A ticket from a counter in global memory whose answer is not waited for where it is asked.  hipcc's atomicAdd() waits for the returned
// value on the spot (`s_waitcnt vmcnt(0)`: every store this wave has in flight must land first).  Vector-memory operations of a wave
// complete in the order they were issued, so once at most LATER of the operations issued after the question are still outstanding the answer
// is there: ticket_answer<LATER> is safe when the wave has issued at least LATER vector-memory operations since ticket_ask.
OGG_DEV unsigned ticket_ask(unsigned* counter) {
    unsigned ret;
    const unsigned one = 1u;
    asm volatile("global_atomic_add %0, %1, %2, off sc0" : "=v"(ret) : "v"(counter), "v"(one) : "memory");
    return ret;
}
// The compiler takes the register ticket_ask returns for a value that is already there: nothing but ticket_answer may read it (no
// initialiser, no loop-carried copy of it -- a copy made before the wait would copy the register's old content).
template <int LATER>
OGG_DEV unsigned ticket_answer(unsigned pending) {
    unsigned out;
    asm volatile("s_waitcnt vmcnt(%2)\n\tv_mov_b32 %0, %1" : "=v"(out) : "v"(pending), "n"(LATER) : "memory");
    return out;
}

constexpr int LF_COLS = 2 * LF_TX;  // columns per workgroup: every thread owns two adjacent columns
inline long latlon_gx(long ni1) { return (ni1 + LF_COLS - 1) / LF_COLS; }

// workgroup bx of the column tiles; it takes the row strips strip_lo + by, + gy, ... < strip_hi.  s_row: LF_ROWS + 1 entries.
// b: index of this workgroup among the gx * gy lat-lon workgroups of the launch, in dispatch order.  Workgroups are handed to
// the 8 XCDs round-robin, so workgroup b runs on XCD b % 8; the remap below gives every XCD a CONTIGUOUS eighth of the row
// strips (all column tiles of those rows) and every workgroup a contiguous block of strips, instead of interleaving the XCDs
// strip by strip: +15 % on the write plateau in scripts/microbench/write_patterns.hip (pattern b2).
//
// claims (NULL: none): two 32-bit counters per (column tile, row block), zero at the start of the launch.  With them the block's strips
// are CLAIMED one at a time -- counter 0 counts claims, a claim beyond the block's strip count ends the walk -- by the block's resident
// workgroup, which takes its k-th claim at strip s_first + k (the same ascending walk as without claims), and by HELPER workgroups
// (helper = true; counter 1 numbers their claims), which take theirs from the far end, s_last - 1 - j: the two sets cannot meet before
// the claims run out.  Helpers sit at the END of a fused launch, so they start when the compute workgroups drain: if the strips are
// what is left by then (1/16 degree on a box with a slow write path: 300 us of a 1.2 ms launch with nothing but the 90 resident
// workgroups running) the whole chip finishes them; if not, they find nothing to claim and leave.  s_claim: one LDS word.
// TAB: the per-row scalars come from p.row_tab (built by an EARLIER launch, or by the previous pass's launch B: the stream's order is the
// dependence) instead of being evaluated here.
template <bool NT, bool TAB = false>
OGG_DEV void latlon_fused_body(const FusedParams& p, RowScalars* s_row, long b, long gx, long gy, long strip_lo, long strip_hi,
                               unsigned* claims = nullptr, bool helper = false, int* s_claim = nullptr, bool pool = false) {
    const long v = xcd_contiguous(b, gx * gy);   // virtual index: the workgroups of XCD x are consecutive
    const long bx = v % gx, by = v / gx;
    const int tid = threadIdx.x;
    const long i0 = (bx * LF_TX + tid) * 2;
    const long ni1 = p.ni1, ni = ni1 - 1;
    const ColScalars c0 = column_scalars(p, i0), c1 = column_scalars(p, i0 + 1);
    const bool full = i0 + 1 < ni;  // both columns exist and have a right neighbour
    // The launch caps the number of resident workgroups (an HBM-write-bound kernel needs only a few waves per SIMD) so that
    // VALU-bound workgroups can share the CUs; each workgroup walks its block of row strips.
    const long per = (strip_hi - strip_lo + gy - 1) / gy;
    const long s_first = strip_lo + by * per, s_last = (s_first + per < strip_hi) ? s_first + per : strip_hi;
    // `pool`: the strips of a column tile are handed out from ONE counter per tile, in row order, to whichever of the tile's workgroups
    // asks next.  The eight XCDs do not write at the same rate -- XCD 2k + 1 gets 40 % of what the pair (2k, 2k + 1) sustains while both
    // are writing (scripts/microbench/xcd_write.hip) -- and with equal shares the launch waits for the slow half of the chip.
    unsigned* cnt = claims ? (pool ? claims + bx : claims + 2 * (by * gx + bx)) : nullptr;
    const unsigned n_pool = (unsigned)(strip_hi - strip_lo);
    if (pool) {
        if (tid == 0) {
            const unsigned t = atomicAdd(cnt, 1u);
            *s_claim = t < n_pool ? (int)t : -1;
        }
        __syncthreads();
    }
    long walked = 0;   // strips this workgroup has taken (wave-uniform)
    for (long strip = s_first;; ++strip) {
        unsigned asked;   // (thread 0, pool) the ticket for the NEXT strip, asked for while this one is written
        if (pool) {
            const int pick = *s_claim;   // written before the barrier that ended the previous strip
            if (pick < 0) break;
            strip = strip_lo + pick;
            // the next ticket is asked for BEFORE this strip's stores are issued: its answer travels ahead of them, and nobody waits for it
            // until they are all on their way
            if (tid == 0) asked = ticket_ask(cnt);
        } else if (cnt) {
            if (tid == 0) {
                const unsigned n = (unsigned)(s_last > s_first ? s_last - s_first : 0);
                int pick = -1;
                if (atomicAdd(cnt, 1u) < n) pick = helper ? (int)(n - 1u - atomicAdd(cnt + 1, 1u)) : (int)walked;
                *s_claim = pick;
            }
            __syncthreads();
            const int pick = *s_claim;   // (the barriers of the strip's body separate this read from the next write)
            if (pick < 0) break;
            strip = s_first + pick;
            ++walked;
        } else if (strip >= s_last) {
            break;
        }
        int bi = 0;
        while (bi + 1 < p.n_bands && strip >= p.strip0[bi + 1]) ++bi;
        const ogg_latlon_band& b = p.band[bi];
        const long n_cell_rows = p.metrics ? b.n_cell_rows : 0;
        const long js = (strip - p.strip0[bi]) * p.rows_per_block;
        const long je = (js + p.rows_per_block < b.n_pt_rows) ? js + p.rows_per_block : b.n_pt_rows;
        const int nrows = (int)(je - js);
        // per-row scalars for rows js .. je (row je only when a cell row needs it)
        if (TAB) {
            if (tid <= nrows) s_row[tid] = p.row_tab[p.row0[bi] + js + tid];
        } else {
            if (tid <= nrows) {
                const long j = js + tid;
                const bool have = (tid < nrows) || (j - 1 < n_cell_rows);
                RowScalars r = {0.0, 0.0, 0.0, 0.0};
                if (have) {
                    r.lat = axis_lat(b, b.k0 + j, p.Ni);
                    const double lv = (0.5 * (r.lat + r.lat)) * kPi180;
                    sincos(lv, &r.sl, &r.cl);
                }
                s_row[tid] = r;
            }
            __syncthreads();
            if (p.metrics && tid < nrows) {  // dy of cell row j needs lat_{j+1}
                const long j = js + tid;
                if (j < n_cell_rows) {
                    const double dyj = (s_row[tid + 1].lat - s_row[tid].lat) * kPi180;
                    s_row[tid].dy = p.Re * sqrt(dyj * dyj + 0.0);
                }
            }
        }
        __syncthreads();
        if (full)
            latlon_rows<true, NT>(p, b, s_row, js, nrows, n_cell_rows, i0, c0, c1);
        else if (i0 < ni1)  // the last column (ni1 odd) or the last pair (no dx / area in its second column)
            latlon_rows<false, NT>(p, b, s_row, js, nrows, n_cell_rows, i0, c0, c1);
        if (pool && tid == 0) {
            // this wave has issued at least three stores per row since it asked (x, y, angle_dx of its first column)
            const unsigned t = nrows >= 6 ? ticket_answer<16>(asked) : nrows == 5 ? ticket_answer<12>(asked) : nrows == 4 ? ticket_answer<8>(asked)
                               : nrows == 3 ? ticket_answer<6>(asked) : ticket_answer<0>(asked);
            *s_claim = t < n_pool ? (int)t : -1;
        }
        __syncthreads();  // the row table is rewritten by the next strip
    }
}

__global__ __launch_bounds__(LF_TX) void latlon_fused_kernel(FusedParams p) {
    __shared__ RowScalars s_row[LF_ROWS + 1];
    latlon_fused_body<false>(p, s_row, (long)blockIdx.y * gridDim.x + blockIdx.x, gridDim.x, gridDim.y, 0, p.strip0[p.n_bands]);
}

// the row strips of the bands at `rpb` rows per strip (1 .. LF_ROWS)
inline void set_rows_per_strip(FusedParams& p, long rpb) {
    rpb = rpb < 1 ? 1 : (rpb > LF_ROWS ? LF_ROWS : rpb);
    p.rows_per_block = (int)rpb;
    p.strip0[0] = 0;
    for (int k = 0; k < p.n_bands; ++k) p.strip0[k + 1] = p.strip0[k] + (p.band[k].n_pt_rows + rpb - 1) / rpb;
}

// Validates the bands and fills the kernel parameters; returns the number of points (0: nothing to do).
inline int plan_latlon(int n_bands, const ogg_latlon_band* bands, long ni1, double lon0, double lenlon, double Re, int metrics,
                       FusedParams& p, long& points) {
    OGG_REQUIRE(n_bands >= 0 && n_bands <= LF_MAX_BANDS && (n_bands == 0 || bands) && ni1 >= 2, OGG_EARG,
                "ogg_latlon_supergrid_multi: bad argument (at most %d bands)", LF_MAX_BANDS);
    p = FusedParams{};
    long total_rows = 0;
    points = 0;
    for (int k = 0; k < n_bands; ++k) {
        const ogg_latlon_band& b = bands[k];
        OGG_REQUIRE(b.n_pt_rows >= 0 && b.n_cell_rows >= 0 && b.n_cell_rows <= b.n_pt_rows, OGG_ESHAPE,
                    "ogg_latlon_supergrid_multi: band %d rows pt=%ld cell=%ld", k, b.n_pt_rows, b.n_cell_rows);
        if (b.n_pt_rows == 0) continue;
        OGG_REQUIRE(b.x && b.y && b.angle && (b.axis_kind != 2 || b.lat1d), OGG_EARG, "ogg_latlon_supergrid_multi: null pointer in band %d", k);
        OGG_REQUIRE(!metrics || (b.dx && (b.n_cell_rows == 0 || (b.dy && b.area))), OGG_EARG,
                    "ogg_latlon_supergrid_multi: null metrics output in band %d", k);
        p.band[p.n_bands++] = b;
        total_rows += b.n_pt_rows;
        points += b.n_pt_rows * ni1;
    }
    if (p.n_bands == 0) return OGG_OK;
    const long gx = latlon_gx(ni1);
    long rpb = (total_rows * gx + 2047) / 2048;  // aim at >= 2048 row strips x column tiles
    rpb = rpb < 8 ? 8 : (rpb > LF_ROWS ? LF_ROWS : rpb);   // >= 8 rows per strip: the per-strip set-up (row scalars, two barriers) is worth ~2 rows
    if (const char* e = getenv("OGG_LL_ROWS_PER_STRIP")) rpb = atol(e) < 1 ? 1 : (atol(e) > LF_ROWS ? LF_ROWS : atol(e));   // (experiments)
    p.row0[0] = 0;
    for (int k = 0; k < p.n_bands; ++k) p.row0[k + 1] = p.row0[k] + p.band[k].n_pt_rows + 1;   // rows 0 .. n_pt_rows: the last cell row reads lat and sin of the row above it
    p.ni1 = ni1, p.lon0 = lon0, p.lenlon = lenlon, p.Ni = (double)(ni1 - 1);
    p.Re = Re, p.Re2 = pow(Re, 2.0), p.metrics = metrics;
    set_rows_per_strip(p, rpb);
    return OGG_OK;
}

}  // namespace
