// Host-pointer entry points of the C ABI (drop-in layer: numpy arrays in, numpy arrays out) and small utilities.
// Each function stages its inputs into device memory, runs the *_dev kernels on the default stream, copies the
// results back and frees its scratch.  There is no CPU compute path: without a usable HIP device every function
// returns OGG_EHIP.
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <utility>
#include <vector>

#include "ogg_common.h"

namespace ogg {

static thread_local char g_err[512] = "";

bool cap_symmetry(int requested) {
    if (requested == OGG_SYM_MIRROR) return true;
    if (requested == OGG_SYM_NONE) return false;
    const char* e = getenv("OGG_CAP_SYMMETRY");   // OGG_SYM_DEFAULT
    if (!e) return true;
    return !(e[0] == '0' || e[0] == 'n' || e[0] == 'N');
}

int set_error(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

// ---- staging of the host-pointer layer --------------------------------------------------------------------------------------
// One call at a time (a process-wide mutex: the reference is single-threaded, ctypes releases the GIL).  Device scratch comes from a
// grow-only arena that is kept between calls (no hipMalloc / hipFree per call); host <-> device transfers go through two pinned
// buffers on a private non-blocking stream -- the copy engine fills / drains one while the CPU copies the other to / from the
// caller's (pageable) array -- so nothing runs on the null stream and nothing serialises with the caller's own streams.
namespace {

constexpr size_t kStageBytes = 8u << 20;
constexpr size_t kArenaKeepBytes = 64u << 20;   // a call's device scratch above this is returned to the device when the call ends
constexpr int kMaxDevices = 64;

struct Staging {
    std::mutex mu;
    hipStream_t stream = nullptr;
    void* pinned[2] = {nullptr, nullptr};
    hipEvent_t done[2] = {nullptr, nullptr};
    std::vector<std::pair<char*, size_t>> blocks;   // device arena: blocks of the current call (one block in the steady state)
    size_t used = 0;                                // bytes taken from blocks.back()

    int init() {
        if (stream) return OGG_OK;
        OGG_HIP_CHECK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
        for (int k = 0; k < 2; ++k) {
            OGG_HIP_CHECK(hipHostMalloc(&pinned[k], kStageBytes, hipHostMallocDefault));
            OGG_HIP_CHECK(hipEventCreateWithFlags(&done[k], hipEventDisableTiming));
        }
        return OGG_OK;
    }
    size_t hint = 0;                                // what the last call took in all (capped): the size of the next first block
    int take(void** out, size_t bytes) {
        bytes = (bytes + 255) & ~size_t(255);
        if (blocks.empty() || used + bytes > blocks.back().second) {
            // a new block for THIS request (never the sum of the earlier ones: those stay where they are until the call ends);
            // the first block of a call is as large as the whole previous call, so a repeated call allocates nothing
            const size_t want = blocks.empty() && hint > bytes ? hint : bytes;
            void* p = nullptr;
            hipError_t e = hipMalloc(&p, want);
            if (e != hipSuccess) {
                const int code = (e == hipErrorOutOfMemory) ? OGG_ENOMEM : OGG_EHIP;
                return set_error(code, "hipMalloc(%zu bytes) failed: %s", want, hipGetErrorString(e));
            }
            blocks.emplace_back(static_cast<char*>(p), want);
            used = 0;
        }
        *out = blocks.back().first + used;
        used += bytes;
        return OGG_OK;
    }
    void release() {   // end of a call (the stream has been synchronised): keep ONE block of at most kArenaKeepBytes
        size_t total = 0;
        for (auto& b : blocks) total += b.second;
        if (blocks.size() > 1 || total > kArenaKeepBytes) {
            for (auto& b : blocks) (void)hipFree(b.first);
            blocks.clear();
            hint = total < kArenaKeepBytes ? total : kArenaKeepBytes;
        }
        used = 0;
    }
};

// one staging state per DEVICE: stream, pinned buffers and arena belong to the device that was current when they were created, and a
// caller may move on to another one (ogg_set_device / hipSetDevice / torch.cuda.set_device) between calls
Staging g_staging_of[kMaxDevices];

Staging& staging_of_current_device() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) dev = 0;
    return g_staging_of[dev];
}

}  // namespace

// scratch device buffers and transfers of one host-pointer call
class DevScratch {
   public:
    DevScratch() : st_(staging_of_current_device()), lock_(st_.mu) {}
    ~DevScratch() {
        if (st_.stream) (void)hipStreamSynchronize(st_.stream);
        st_.release();
    }
    void* stream() { return st_.stream; }
    int alloc(double** out, long n) {
        if (int e = st_.init()) return e;
        void* p = nullptr;
        if (int e = st_.take(&p, (size_t)(n > 0 ? n : 1) * sizeof(double))) return e;
        *out = static_cast<double*>(p);
        return OGG_OK;
    }
    int upload(double** out, const double* host, long n) {
        if (int e = alloc(out, n)) return e;
        const size_t bytes = (size_t)(n > 0 ? n : 0) * sizeof(double);
        const char* src = reinterpret_cast<const char*>(host);
        char* dst = reinterpret_cast<char*>(*out);
        int k = 0;
        for (size_t off = 0; off < bytes; off += kStageBytes, k ^= 1) {
            const size_t len = bytes - off < kStageBytes ? bytes - off : kStageBytes;
            OGG_HIP_CHECK(hipEventSynchronize(st_.done[k]));   // the copy that last read this buffer has finished
            memcpy(st_.pinned[k], src + off, len);
            OGG_HIP_CHECK(hipMemcpyAsync(dst + off, st_.pinned[k], len, hipMemcpyHostToDevice, st_.stream));
            OGG_HIP_CHECK(hipEventRecord(st_.done[k], st_.stream));
        }
        return OGG_OK;
    }
    // device -> caller's array, after everything enqueued on the stream so far
    int download(void* host, const void* dev, size_t bytes) {
        if (!host || bytes == 0) return OGG_OK;
        char* dst = reinterpret_cast<char*>(host);
        const char* src = reinterpret_cast<const char*>(dev);
        size_t pending_off[2] = {0, 0}, pending_len[2] = {0, 0};
        int k = 0;
        for (size_t off = 0; off < bytes; off += kStageBytes, k ^= 1) {
            const size_t len = bytes - off < kStageBytes ? bytes - off : kStageBytes;
            if (pending_len[k]) {   // drain the chunk that sits in this buffer before the engine overwrites it
                OGG_HIP_CHECK(hipEventSynchronize(st_.done[k]));
                memcpy(dst + pending_off[k], st_.pinned[k], pending_len[k]);
            }
            OGG_HIP_CHECK(hipMemcpyAsync(st_.pinned[k], src + off, len, hipMemcpyDeviceToHost, st_.stream));
            OGG_HIP_CHECK(hipEventRecord(st_.done[k], st_.stream));
            pending_off[k] = off, pending_len[k] = len;
        }
        for (int j = 0; j < 2; ++j, k ^= 1) {   // oldest first
            if (pending_len[k]) {
                OGG_HIP_CHECK(hipEventSynchronize(st_.done[k]));
                memcpy(dst + pending_off[k], st_.pinned[k], pending_len[k]);
                pending_len[k] = 0;
            }
        }
        return OGG_OK;
    }

   private:
    Staging& st_;                        // (declared before lock_: the lock is taken on this device's mutex)
    std::lock_guard<std::mutex> lock_;
};

}  // namespace ogg

using ogg::DevScratch;

#define OGG_DOWNLOAD(host, dev, n) s.download((host), (dev), (size_t)((n) > 0 ? (n) : 0) * sizeof(double))

#define OGG_TRY(expr)          \
    do {                       \
        int e__ = (expr);      \
        if (e__) return e__;   \
    } while (0)

extern "C" {

const char* ogg_last_error(void) { return ogg::g_err; }
#ifndef OGG_SRC_HASH
#define OGG_SRC_HASH "unknown"
#endif
// "... src <hash>": the hash of the kernel sources this library was built from (csrc/build.py source_hash()); the counter files under
// profiles/ carry the hash of the library they were taken with, and bench.py quotes them only when the two agree
// The restated transcendental functions of ogg_math.h (atan, atan2, asin, sin, cos) reproduce the bits of the device library they were read
// from: ROCm 7.2.0's ocml.  The string names that release and the HIP version this library was built with, so that a build against another
// ROCm says so where a user looks first; whether the restatements still agree with the installed library is what ogg_libm_check_dev tests
// (tests/test_gpu_parity.py, and a short form in __graft_entry__.smoke()).
#define OGG_STR2(x) #x
#define OGG_STR(x) OGG_STR2(x)
const char* ogg_version(void) {
    return "ogg_hip 0.5 (gfx950; cap columns mirrored by default, OGG_SYM_*; libm restatements read from ROCm 7.2.0 ocml, built with HIP " OGG_STR(HIP_VERSION_MAJOR) "." OGG_STR(HIP_VERSION_MINOR) "." OGG_STR(
        HIP_VERSION_PATCH) ") src " OGG_SRC_HASH;
}
// sizeof of the descriptor structs of the ABI (0: ogg_latlon_band, 1: ogg_bipolar_band), so that a binding can check its layout
long ogg_abi_sizeof(int which) {
    return which == 0 ? (long)sizeof(ogg_latlon_band)
                      : (which == 1 ? (long)sizeof(ogg_bipolar_band) : (which == 2 ? (long)sizeof(ogg_dpole_band) : -1L));
}

int ogg_device_count(int* count) {
    OGG_REQUIRE(count, OGG_EARG, "ogg_device_count: null pointer");
    *count = 0;
    OGG_HIP_CHECK(hipGetDeviceCount(count));
    return OGG_OK;
}

int ogg_set_device(int device) {
    OGG_HIP_CHECK(hipSetDevice(device));
    return OGG_OK;
}

int ogg_device_name(char* buf, int buflen) {
    OGG_REQUIRE(buf && buflen > 0, OGG_EARG, "ogg_device_name: bad buffer");
    int dev = 0;
    OGG_HIP_CHECK(hipGetDevice(&dev));
    hipDeviceProp_t prop;
    OGG_HIP_CHECK(hipGetDeviceProperties(&prop, dev));
    snprintf(buf, (size_t)buflen, "%s (%s, %d CUs)", prop.name, prop.gcnArchName, prop.multiProcessorCount);
    return OGG_OK;
}

int ogg_event_create(void** ev) {
    OGG_REQUIRE(ev, OGG_EARG, "ogg_event_create: null pointer");
    hipEvent_t e;
    OGG_HIP_CHECK(hipEventCreate(&e));
    *ev = e;
    return OGG_OK;
}
int ogg_event_destroy(void* ev) {
    OGG_HIP_CHECK(hipEventDestroy(static_cast<hipEvent_t>(ev)));
    return OGG_OK;
}
int ogg_event_record(void* ev, void* stream) {
    OGG_HIP_CHECK(hipEventRecord(static_cast<hipEvent_t>(ev), ogg::as_stream(stream)));
    return OGG_OK;
}
int ogg_event_elapsed_ms(void* ev_start, void* ev_stop, float* ms) {
    OGG_REQUIRE(ms, OGG_EARG, "ogg_event_elapsed_ms: null pointer");
    OGG_HIP_CHECK(hipEventSynchronize(static_cast<hipEvent_t>(ev_stop)));
    OGG_HIP_CHECK(hipEventElapsedTime(ms, static_cast<hipEvent_t>(ev_start), static_cast<hipEvent_t>(ev_stop)));
    return OGG_OK;
}
int ogg_stream_synchronize(void* stream) {
    OGG_HIP_CHECK(hipStreamSynchronize(ogg::as_stream(stream)));
    return OGG_OK;
}

// ---- Mercator / lat-lon ------------------------------------------------------------------------------------
int ogg_y_mercator_rounded(long Ni, long n, const double* phi_rad, long long* ystar) {
    OGG_REQUIRE(n >= 0 && phi_rad && ystar, OGG_EARG, "ogg_y_mercator_rounded: bad argument");
    DevScratch s;
    double *d_phi, *d_y;
    OGG_TRY(s.upload(&d_phi, phi_rad, n));
    OGG_TRY(s.alloc(&d_y, n));
    OGG_TRY(ogg_y_mercator_rounded_dev(Ni, n, d_phi, reinterpret_cast<long long*>(d_y), s.stream()));
    return s.download(ystar, d_y, (size_t)n * sizeof(long long));
}

int ogg_phi_mercator(long Ni, long n, const double* y, double* phi_deg) {
    OGG_REQUIRE(n >= 0 && y && phi_deg, OGG_EARG, "ogg_phi_mercator: bad argument");
    DevScratch s;
    double *d_y, *d_phi;
    OGG_TRY(s.upload(&d_y, y, n));
    OGG_TRY(s.alloc(&d_phi, n));
    OGG_TRY(ogg_phi_mercator_dev(Ni, n, d_y, d_phi, s.stream()));
    return OGG_DOWNLOAD(phi_deg, d_phi, n);
}

int ogg_tile_latlon(long nrows, long ni1, const double* lat1d, const double* lon1d, double* x, double* y) {
    OGG_REQUIRE(nrows >= 0 && ni1 > 0 && lat1d && lon1d && x && y, OGG_EARG, "ogg_tile_latlon: bad argument");
    DevScratch s;
    double *d_lat, *d_lon, *d_x, *d_y;
    OGG_TRY(s.upload(&d_lat, lat1d, nrows));
    OGG_TRY(s.upload(&d_lon, lon1d, ni1));
    OGG_TRY(s.alloc(&d_x, nrows * ni1));
    OGG_TRY(s.alloc(&d_y, nrows * ni1));
    OGG_TRY(ogg_tile_latlon_dev(nrows, ni1, d_lat, d_lon, d_x, d_y, s.stream()));
    OGG_TRY(OGG_DOWNLOAD(x, d_x, nrows * ni1));
    return OGG_DOWNLOAD(y, d_y, nrows * ni1);
}

int ogg_generate_latlon_grid(long lni, long lnj, double llon0, double llen_lon, double llat0, double llen_lat,
                             int skip_first_row, double* x, double* y) {
    OGG_REQUIRE(lni > 0 && lnj > 0 && x && y, OGG_EARG, "ogg_generate_latlon_grid: bad argument");
    const long skip = skip_first_row ? 1 : 0;
    const long nrows = lnj + 1 - skip, ni1 = lni + 1;
    DevScratch s;
    double *d_lat, *d_lon, *d_x, *d_y;
    OGG_TRY(s.alloc(&d_lat, lnj + 1));
    OGG_TRY(s.alloc(&d_lon, ni1));
    OGG_TRY(s.alloc(&d_x, nrows * ni1));
    OGG_TRY(s.alloc(&d_y, nrows * ni1));
    OGG_TRY(ogg_linear_axis_dev(ni1, llon0, llen_lon, (double)lni, d_lon, s.stream()));      // OGG:834
    OGG_TRY(ogg_linear_axis_dev(lnj + 1, llat0, llen_lat, (double)lnj, d_lat, s.stream()));  // OGG:835
    OGG_TRY(ogg_tile_latlon_dev(nrows, ni1, d_lat + skip, d_lon, d_x, d_y, s.stream()));
    OGG_TRY(OGG_DOWNLOAD(x, d_x, nrows * ni1));
    return OGG_DOWNLOAD(y, d_y, nrows * ni1);
}

// ---- MIDAS + angle -----------------------------------------------------------------------------------------
int ogg_grid_metrics_midas(long nj1, long ni1, const double* x, const double* y, double Re, int latlon_areafix, double* dx,
                           double* dy, double* area) {
    OGG_REQUIRE(x && y && dx && dy && area, OGG_EARG, "ogg_grid_metrics_midas: null pointer");
    OGG_REQUIRE(nj1 >= 2 && ni1 >= 2, OGG_ESHAPE, "ogg_grid_metrics_midas: need at least 2x2 points, got %ld x %ld", nj1, ni1);
    DevScratch s;
    double *d_x, *d_y, *d_dx, *d_dy, *d_ar;
    OGG_TRY(s.upload(&d_x, x, nj1 * ni1));
    OGG_TRY(s.upload(&d_y, y, nj1 * ni1));
    OGG_TRY(s.alloc(&d_dx, nj1 * (ni1 - 1)));
    OGG_TRY(s.alloc(&d_dy, (nj1 - 1) * ni1));
    OGG_TRY(s.alloc(&d_ar, (nj1 - 1) * (ni1 - 1)));
    OGG_TRY(ogg_grid_metrics_midas_dev(nj1, ni1, d_x, d_y, nj1, nj1 - 1, Re, latlon_areafix, d_dx, d_dy, d_ar, nullptr, s.stream()));
    OGG_TRY(OGG_DOWNLOAD(dx, d_dx, nj1 * (ni1 - 1)));
    OGG_TRY(OGG_DOWNLOAD(dy, d_dy, (nj1 - 1) * ni1));
    return OGG_DOWNLOAD(area, d_ar, (nj1 - 1) * (ni1 - 1));
}

int ogg_angle_x(long nj1, long ni1, const double* x, const double* y, double* angle_dx) {
    OGG_REQUIRE(x && y && angle_dx, OGG_EARG, "ogg_angle_x: null pointer");
    OGG_REQUIRE(nj1 >= 1 && ni1 >= 2, OGG_ESHAPE, "Input arrays do not have the same shape!");
    DevScratch s;
    double *d_x, *d_y, *d_a;
    OGG_TRY(s.upload(&d_x, x, nj1 * ni1));
    OGG_TRY(s.upload(&d_y, y, nj1 * ni1));
    OGG_TRY(s.alloc(&d_a, nj1 * ni1));
    OGG_TRY(ogg_grid_metrics_midas_dev(nj1, ni1, d_x, d_y, nj1, 0, 6371.0e3, 1, nullptr, nullptr, nullptr, d_a, s.stream()));
    return OGG_DOWNLOAD(angle_dx, d_a, nj1 * ni1);
}

// ---- bipolar cap -------------------------------------------------------------------------------------------
int ogg_bipolar_projection(long n, const double* lamg, const double* phig, double lon_bp, double rp, int metrics_only,
                           double* lams, double* phis, double* h_i_inv, double* h_j_inv) {
    OGG_REQUIRE(n >= 0 && lamg && phig && h_i_inv && h_j_inv, OGG_EARG, "ogg_bipolar_projection: bad argument");
    OGG_REQUIRE(metrics_only || (lams && phis), OGG_EARG, "ogg_bipolar_projection: lams/phis required");
    DevScratch s;
    double *d_l, *d_p, *d_ls, *d_ps, *d_hi, *d_hj;
    OGG_TRY(s.upload(&d_l, lamg, n));
    OGG_TRY(s.upload(&d_p, phig, n));
    OGG_TRY(s.alloc(&d_ls, n));
    OGG_TRY(s.alloc(&d_ps, n));
    OGG_TRY(s.alloc(&d_hi, n));
    OGG_TRY(s.alloc(&d_hj, n));
    OGG_TRY(ogg_bipolar_projection_dev(n, d_l, d_p, lon_bp, rp, metrics_only, d_ls, d_ps, d_hi, d_hj, s.stream()));
    if (!metrics_only) {
        OGG_TRY(OGG_DOWNLOAD(lams, d_ls, n));
        OGG_TRY(OGG_DOWNLOAD(phis, d_ps, n));
    }
    OGG_TRY(OGG_DOWNLOAD(h_i_inv, d_hi, n));
    return OGG_DOWNLOAD(h_j_inv, d_hj, n);
}

int ogg_bipolar_cap_mesh_sym(long Ni, long Nj, double lat0_bp, double lon_bp, int symmetry, double* lams, double* phis, double* h_i_inv,
                             double* h_j_inv) {
    OGG_REQUIRE(Ni > 0 && Nj > 0 && lams && phis, OGG_EARG, "ogg_bipolar_cap_mesh: bad argument");
    const long n = (Nj + 1) * (Ni + 1);
    DevScratch s;
    double *d_ls, *d_ps, *d_hi = nullptr, *d_hj = nullptr;
    OGG_TRY(s.alloc(&d_ls, n));
    OGG_TRY(s.alloc(&d_ps, n));
    if (h_i_inv) OGG_TRY(s.alloc(&d_hi, (Nj + 1) * Ni));
    if (h_j_inv) OGG_TRY(s.alloc(&d_hj, Nj * (Ni + 1)));
    OGG_TRY(ogg_bipolar_cap_mesh_angle_sym_dev(Ni, Nj, lat0_bp, lon_bp, 0, Nj + 1, symmetry, d_ls, d_ps, d_hi, d_hj, nullptr, s.stream()));
    OGG_TRY(OGG_DOWNLOAD(lams, d_ls, n));
    OGG_TRY(OGG_DOWNLOAD(phis, d_ps, n));
    if (h_i_inv) OGG_TRY(OGG_DOWNLOAD(h_i_inv, d_hi, (Nj + 1) * Ni));
    if (h_j_inv) OGG_TRY(OGG_DOWNLOAD(h_j_inv, d_hj, Nj * (Ni + 1)));
    return OGG_OK;
}

int ogg_bipolar_cap_mesh(long Ni, long Nj, double lat0_bp, double lon_bp, double* lams, double* phis, double* h_i_inv,
                         double* h_j_inv) {
    return ogg_bipolar_cap_mesh_sym(Ni, Nj, lat0_bp, lon_bp, OGG_SYM_DEFAULT, lams, phis, h_i_inv, h_j_inv);
}

int ogg_bipolar_cap_metrics_quad_sym(int order, long nx, long ny, double lat0_bp, double lon_bp, double rp, double Re, int symmetry,
                                     double* dxq, double* dyq, double* daq) {
    OGG_REQUIRE(order >= 2 && order <= 5, OGG_EORDER, "Uncoded order");
    OGG_REQUIRE(nx > 0 && ny > 0 && dxq && dyq && daq, OGG_EARG, "ogg_bipolar_cap_metrics_quad: bad argument");
    DevScratch s;
    double *d_dx, *d_dy, *d_da;
    OGG_TRY(s.alloc(&d_dx, (ny + 1) * nx));
    OGG_TRY(s.alloc(&d_dy, ny * (nx + 1)));
    OGG_TRY(s.alloc(&d_da, ny * nx));
    OGG_TRY(ogg_bipolar_cap_metrics_quad_sym_ws_dev(order, nx, ny, lat0_bp, lon_bp, rp, Re, 0, ny + 1, ny, symmetry, d_dx, d_dy, d_da, nullptr, 0,
                                                    s.stream()));
    OGG_TRY(OGG_DOWNLOAD(dxq, d_dx, (ny + 1) * nx));
    OGG_TRY(OGG_DOWNLOAD(dyq, d_dy, ny * (nx + 1)));
    return OGG_DOWNLOAD(daq, d_da, ny * nx);
}

int ogg_bipolar_cap_metrics_quad(int order, long nx, long ny, double lat0_bp, double lon_bp, double rp, double Re, double* dxq,
                                 double* dyq, double* daq) {
    return ogg_bipolar_cap_metrics_quad_sym(order, nx, ny, lat0_bp, lon_bp, rp, Re, OGG_SYM_DEFAULT, dxq, dyq, daq);
}

// ---- displaced pole cap ------------------------------------------------------------------------------------
int ogg_displaced_pole_mesh(long n_i, const double* i, long n_j, const double* j, long ni, long nj, double lon0, double lat0,
                            double lam_pole, double r_pole, double* lams, double* phis) {
    OGG_REQUIRE(n_i > 0 && n_j >= 0 && i && j && lams && phis, OGG_EARG, "ogg_displaced_pole_mesh: bad argument");
    DevScratch s;
    double *d_i, *d_j, *d_l, *d_p;
    OGG_TRY(s.upload(&d_i, i, n_i));
    OGG_TRY(s.upload(&d_j, j, n_j));
    OGG_TRY(s.alloc(&d_l, n_i * n_j));
    OGG_TRY(s.alloc(&d_p, n_i * n_j));
    OGG_TRY(ogg_displaced_pole_mesh_dev(n_i, d_i, n_j, d_j, ni, nj, lon0, lat0, lam_pole, r_pole, d_l, d_p, s.stream()));
    OGG_TRY(OGG_DOWNLOAD(lams, d_l, n_i * n_j));
    return OGG_DOWNLOAD(phis, d_p, n_i * n_j);
}

int ogg_displaced_pole_numerical_h(long n_i, const double* i, long n_j, const double* j, long nx, long ny, double lon0,
                                   double lat0, double lon_dp, double r_dp, double eps, int fd_order, double* h_i, double* h_j) {
    OGG_REQUIRE(n_i > 0 && n_j >= 0 && i && j && (h_i || h_j), OGG_EARG, "ogg_displaced_pole_numerical_h: bad argument");
    DevScratch s;
    double *d_i, *d_j, *d_hi = nullptr, *d_hj = nullptr;
    OGG_TRY(s.upload(&d_i, i, n_i));
    OGG_TRY(s.upload(&d_j, j, n_j));
    if (h_i) OGG_TRY(s.alloc(&d_hi, n_i * n_j));
    if (h_j) OGG_TRY(s.alloc(&d_hj, n_i * n_j));
    OGG_TRY(ogg_displaced_pole_numerical_h_dev(n_i, d_i, n_j, d_j, nx, ny, lon0, lat0, lon_dp, r_dp, eps, fd_order, d_hi, d_hj, s.stream()));
    if (h_i) OGG_TRY(OGG_DOWNLOAD(h_i, d_hi, n_i * n_j));
    if (h_j) OGG_TRY(OGG_DOWNLOAD(h_j, d_hj, n_i * n_j));
    return OGG_OK;
}

int ogg_displaced_pole_metrics_quad_form_sym(int arc_form, int symmetry, int order, long nx, long ny, double lon0, double lat0, double lon_dp,
                                             double r_dp, double Re, double* dxq, double* dyq, double* daq) {
    OGG_REQUIRE(order >= 2 && order <= 5, OGG_EORDER, "Uncoded order");
    OGG_REQUIRE(order == 2 || order == 4, OGG_EORDER, "order not coded");
    OGG_REQUIRE(nx > 0 && ny > 0 && dxq && dyq && daq, OGG_EARG, "ogg_displaced_pole_metrics_quad: bad argument");
    DevScratch s;
    double *d_dx, *d_dy, *d_da, *d_ws;
    const long ws_bytes = ogg_displaced_pole_quad_workspace_bytes(order, nx, ny);
    OGG_TRY(s.alloc(&d_dx, (ny + 1) * nx));
    OGG_TRY(s.alloc(&d_dy, ny * (nx + 1)));
    OGG_TRY(s.alloc(&d_da, ny * nx));
    OGG_TRY(s.alloc(&d_ws, (ws_bytes + 7) / 8));
    OGG_TRY(ogg_displaced_pole_metrics_quad_form_sym_ws_dev(arc_form, symmetry, order, nx, ny, lon0, lat0, lon_dp, r_dp, Re, 0, ny + 1, ny, d_dx, d_dy,
                                                        d_da, d_ws, ws_bytes, s.stream()));
    OGG_TRY(OGG_DOWNLOAD(dxq, d_dx, (ny + 1) * nx));
    OGG_TRY(OGG_DOWNLOAD(dyq, d_dy, ny * (nx + 1)));
    OGG_TRY(OGG_DOWNLOAD(daq, d_da, ny * nx));
    int flag = 0;
    OGG_TRY(ogg_workspace_error_flag_dev(d_ws, &flag, s.stream()));
    OGG_REQUIRE(flag == 0, OGG_EHIP, "ogg_displaced_pole_metrics_quad: a look-back wait timed out (flag %d)", flag);
    return OGG_OK;
}

int ogg_displaced_pole_metrics_quad_form(int arc_form, int order, long nx, long ny, double lon0, double lat0, double lon_dp, double r_dp,
                                         double Re, double* dxq, double* dyq, double* daq) {
    return ogg_displaced_pole_metrics_quad_form_sym(arc_form, OGG_SYM_DEFAULT, order, nx, ny, lon0, lat0, lon_dp, r_dp, Re, dxq, dyq, daq);
}

int ogg_displaced_pole_metrics_quad(int order, long nx, long ny, double lon0, double lat0, double lon_dp, double r_dp, double Re,
                                    double* dxq, double* dyq, double* daq) {
    return ogg_displaced_pole_metrics_quad_form(OGG_DP_ARC_CHORD, order, nx, ny, lon0, lat0, lon_dp, r_dp, Re, dxq, dyq, daq);
}

// ---- small element-wise entry points -----------------------------------------------------------------------
int ogg_y_mercator(long Ni, long n, const double* phi_rad, double* y) {
    OGG_REQUIRE(n >= 0 && phi_rad && y, OGG_EARG, "ogg_y_mercator: bad argument");
    DevScratch s;
    double *d_in, *d_out;
    OGG_TRY(s.upload(&d_in, phi_rad, n));
    OGG_TRY(s.alloc(&d_out, n));
    OGG_TRY(ogg_y_mercator_dev(Ni, n, d_in, d_out, s.stream()));
    return OGG_DOWNLOAD(y, d_out, n);
}

int ogg_affine_index(long n, const double* idx, double a0, double len, double denom, double* out) {
    OGG_REQUIRE(n >= 0 && idx && out, OGG_EARG, "ogg_affine_index: bad argument");
    DevScratch s;
    double *d_in, *d_out;
    OGG_TRY(s.upload(&d_in, idx, n));
    OGG_TRY(s.alloc(&d_out, n));
    OGG_TRY(ogg_affine_index_dev(n, d_in, a0, len, denom, d_out, s.stream()));
    return OGG_DOWNLOAD(out, d_out, n);
}

int ogg_mdist(long n, const double* x1, const double* x2, double* out) {
    OGG_REQUIRE(n >= 0 && x1 && x2 && out, OGG_EARG, "ogg_mdist: bad argument");
    DevScratch s;
    double *d_a, *d_b, *d_out;
    OGG_TRY(s.upload(&d_a, x1, n));
    OGG_TRY(s.upload(&d_b, x2, n));
    OGG_TRY(s.alloc(&d_out, n));
    OGG_TRY(ogg_mdist_dev(n, d_a, d_b, d_out, s.stream()));
    return OGG_DOWNLOAD(out, d_out, n);
}

int ogg_haversine(long n, const double* lam0, const double* phi0, const double* lam1, const double* phi1, double* out) {
    OGG_REQUIRE(n >= 0 && lam0 && phi0 && lam1 && phi1 && out, OGG_EARG, "ogg_haversine: bad argument");
    DevScratch s;
    double *d0, *d1, *d2, *d3, *d_out;
    OGG_TRY(s.upload(&d0, lam0, n));
    OGG_TRY(s.upload(&d1, phi0, n));
    OGG_TRY(s.upload(&d2, lam1, n));
    OGG_TRY(s.upload(&d3, phi1, n));
    OGG_TRY(s.alloc(&d_out, n));
    OGG_TRY(ogg_haversine_dev(n, d0, d1, d2, d3, d_out, s.stream()));
    return OGG_DOWNLOAD(out, d_out, n);
}

int ogg_bipolar_cap_ij_array(long n_i, const double* i, long n_j, const double* j, long Ni, long Nj, double lat0_bp,
                             double lon_bp, double rp, double* h_i_inv, double* h_j_inv) {
    OGG_REQUIRE(n_i > 0 && n_j >= 0 && i && j && h_i_inv && h_j_inv, OGG_EARG, "ogg_bipolar_cap_ij_array: bad argument");
    DevScratch s;
    double *d_i, *d_j, *d_hi, *d_hj;
    OGG_TRY(s.upload(&d_i, i, n_i));
    OGG_TRY(s.upload(&d_j, j, n_j));
    OGG_TRY(s.alloc(&d_hi, n_i * n_j));
    OGG_TRY(s.alloc(&d_hj, n_i * n_j));
    OGG_TRY(ogg_bipolar_cap_ij_array_dev(n_i, d_i, n_j, d_j, Ni, Nj, lat0_bp, lon_bp, rp, d_hi, d_hj, s.stream()));
    OGG_TRY(OGG_DOWNLOAD(h_i_inv, d_hi, n_i * n_j));
    return OGG_DOWNLOAD(h_j_inv, d_hj, n_i * n_j);
}

int ogg_displaced_pole_projection(long nj, long ni, const double* lon_grid, const double* lat_grid, double z0_re, double z0_im,
                                  double r_joint, double x_0, double* lam, double* phi) {
    OGG_REQUIRE(nj >= 0 && ni > 0 && lon_grid && lat_grid && lam && phi, OGG_EARG, "ogg_displaced_pole_projection: bad argument");
    DevScratch s;
    double *d_lon, *d_lat, *d_l, *d_p;
    OGG_TRY(s.upload(&d_lon, lon_grid, nj * ni));
    OGG_TRY(s.upload(&d_lat, lat_grid, nj * ni));
    OGG_TRY(s.alloc(&d_l, nj * ni));
    OGG_TRY(s.alloc(&d_p, nj * ni));
    OGG_TRY(ogg_displaced_pole_projection_dev(nj, ni, d_lon, d_lat, z0_re, z0_im, r_joint, x_0, d_l, d_p, s.stream()));
    OGG_TRY(OGG_DOWNLOAD(lam, d_l, nj * ni));
    return OGG_DOWNLOAD(phi, d_p, nj * ni);
}

int ogg_monotonic_bounding(long nj, long ni, double* x, double x_0) {
    OGG_REQUIRE(nj >= 0 && ni > 0 && x, OGG_EARG, "ogg_monotonic_bounding: bad argument");
    DevScratch s;
    double* d_x;
    OGG_TRY(s.upload(&d_x, x, nj * ni));
    OGG_TRY(ogg_monotonic_bounding_dev(nj, ni, d_x, x_0, s.stream()));
    return OGG_DOWNLOAD(x, d_x, nj * ni);
}

}  // extern "C"
