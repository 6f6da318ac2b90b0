// A fake HIP runtime for the sanitizer build of the library's HOST code (tests/test_sanitize_host.py): test infrastructure, never part
// of the product and not importable from the package.  "Device" memory is host memory (malloc), so every byte the staging layer moves and
// every table pointer the plan builders carve is checked by AddressSanitizer; kernel launches are no-ops.  Copies that the real
// runtime performs asynchronously are DEFERRED here -- queued on their stream and executed as late as the API allows (at the event /
// stream / device synchronisation that orders them) -- so that host code which reuses a staging buffer before waiting for the copy
// that reads it moves the wrong bytes, and the driver's round-trip comparison sees it.
#include <hip/hip_runtime_api.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <mutex>
#include <vector>

namespace {

struct Op {
    int kind;   // 0: memcpy, 1: event record
    void* dst;
    const void* src;
    size_t bytes;
    void* event;
};
struct Stream {
    std::deque<Op> q;
};
struct Event {
    Stream* stream = nullptr;   // where the last record sits
    bool pending = false;
};

std::mutex g_mu;
std::vector<Stream*> g_streams;
Stream g_null_stream;
int g_device = 0;
long g_launches = 0, g_copies = 0;

Stream* S(hipStream_t s) { return s ? reinterpret_cast<Stream*>(s) : &g_null_stream; }

void run_until(Stream* st, Event* until) {   // execute queued operations, up to and including the record of `until` (all if NULL)
    while (!st->q.empty()) {
        Op op = st->q.front();
        st->q.pop_front();
        if (op.kind == 0) {
            if (op.bytes) memcpy(op.dst, op.src, op.bytes);
            g_copies += 1;
        } else {
            Event* e = static_cast<Event*>(op.event);
            e->pending = false;
            if (e == until) return;
        }
    }
}

void drain_all() {
    run_until(&g_null_stream, nullptr);
    for (Stream* s : g_streams) run_until(s, nullptr);
}

}  // namespace

extern "C" {

long fake_hip_launches(void) { return g_launches; }
long fake_hip_copies(void) { return g_copies; }

hipError_t hipGetDeviceCount(int* n) {
    *n = 2;
    return hipSuccess;
}
hipError_t hipGetDevice(int* d) {
    *d = g_device;
    return hipSuccess;
}
hipError_t hipSetDevice(int d) {
    if (d < 0 || d >= 2) return hipErrorInvalidDevice;
    g_device = d;
    return hipSuccess;
}
hipError_t hipGetDevicePropertiesR0600(hipDeviceProp_tR0600* p, int) {
    memset(p, 0, sizeof(*p));
    snprintf(p->name, sizeof(p->name), "fake gfx950 (sanitizer build)");
    return hipSuccess;
}
hipError_t hipGetLastError(void) { return hipSuccess; }
const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "no error" : "fake error"; }

hipError_t hipMalloc(void** p, size_t bytes) {
    *p = malloc(bytes ? bytes : 1);
    return *p ? hipSuccess : hipErrorOutOfMemory;
}
hipError_t hipFree(void* p) {
    std::lock_guard<std::mutex> g(g_mu);
    drain_all();   // hipFree synchronises the device
    free(p);
    return hipSuccess;
}
hipError_t hipMallocAsync(void** p, size_t bytes, hipStream_t) { return hipMalloc(p, bytes); }
hipError_t hipFreeAsync(void* p, hipStream_t s) {
    std::lock_guard<std::mutex> g(g_mu);
    run_until(S(s), nullptr);
    free(p);
    return hipSuccess;
}
hipError_t hipHostMalloc(void** p, size_t bytes, unsigned) { return hipMalloc(p, bytes); }
hipError_t hipHostFree(void* p) {
    free(p);
    return hipSuccess;
}
hipError_t hipMemcpy(void* dst, const void* src, size_t bytes, hipMemcpyKind) {
    std::lock_guard<std::mutex> g(g_mu);
    drain_all();
    if (bytes) memcpy(dst, src, bytes);
    return hipSuccess;
}
hipError_t hipMemcpyAsync(void* dst, const void* src, size_t bytes, hipMemcpyKind, hipStream_t s) {
    std::lock_guard<std::mutex> g(g_mu);
    S(s)->q.push_back(Op{0, dst, src, bytes, nullptr});
    return hipSuccess;
}
hipError_t hipMemset(void* p, int v, size_t bytes) {
    std::lock_guard<std::mutex> g(g_mu);
    drain_all();
    memset(p, v, bytes);
    return hipSuccess;
}
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) {
    std::lock_guard<std::mutex> g(g_mu);
    Stream* st = new Stream;
    g_streams.push_back(st);
    *s = reinterpret_cast<hipStream_t>(st);
    return hipSuccess;
}
hipError_t hipStreamSynchronize(hipStream_t s) {
    std::lock_guard<std::mutex> g(g_mu);
    run_until(S(s), nullptr);
    return hipSuccess;
}
hipError_t hipDeviceSynchronize(void) {
    std::lock_guard<std::mutex> g(g_mu);
    drain_all();
    return hipSuccess;
}
hipError_t hipStreamIsCapturing(hipStream_t, hipStreamCaptureStatus* st) {
    *st = hipStreamCaptureStatusNone;
    return hipSuccess;
}
hipError_t hipEventCreate(hipEvent_t* e) {
    *e = reinterpret_cast<hipEvent_t>(new Event);
    return hipSuccess;
}
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { return hipEventCreate(e); }
hipError_t hipEventDestroy(hipEvent_t e) {
    std::lock_guard<std::mutex> g(g_mu);
    Event* ev = reinterpret_cast<Event*>(e);
    if (ev->pending && ev->stream) run_until(ev->stream, ev);
    delete ev;
    return hipSuccess;
}
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s) {
    std::lock_guard<std::mutex> g(g_mu);
    Event* ev = reinterpret_cast<Event*>(e);
    if (ev->pending && ev->stream) run_until(ev->stream, ev);   // a re-record: the earlier one is resolved first
    ev->stream = S(s), ev->pending = true;
    S(s)->q.push_back(Op{1, nullptr, nullptr, 0, ev});
    return hipSuccess;
}
hipError_t hipEventSynchronize(hipEvent_t e) {
    std::lock_guard<std::mutex> g(g_mu);
    Event* ev = reinterpret_cast<Event*>(e);
    if (ev->pending && ev->stream) run_until(ev->stream, ev);
    return hipSuccess;
}
hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) {
    *ms = 0.001f;
    return hipSuccess;
}

// kernel launches: nothing runs
hipError_t hipLaunchKernel(const void*, dim3, dim3, void**, size_t, hipStream_t) {
    g_launches += 1;
    return hipSuccess;
}
struct FakeConfig {
    dim3 grid, block;
    size_t shmem;
    hipStream_t stream;
};
static thread_local FakeConfig t_cfg;
hipError_t __hipPushCallConfiguration(dim3 grid, dim3 block, size_t shmem, hipStream_t stream) {
    t_cfg = FakeConfig{grid, block, shmem, stream};
    return hipSuccess;
}
hipError_t __hipPopCallConfiguration(dim3* grid, dim3* block, size_t* shmem, hipStream_t* stream) {
    *grid = t_cfg.grid, *block = t_cfg.block, *shmem = t_cfg.shmem, *stream = t_cfg.stream;
    return hipSuccess;
}
void** __hipRegisterFatBinary(const void*) {
    static void* handle = nullptr;
    return &handle;
}
void __hipUnregisterFatBinary(void**) {}
void __hipRegisterFunction(void**, const void*, char*, const char*, unsigned, void*, void*, void*, void*, int*) {}
void __hipRegisterVar(void**, void*, char*, const char*, int, size_t, int, int) {}
void __hipRegisterManagedVar(void*, void**, void*, const char*, size_t, unsigned) {}
void __hipRegisterSurface(void**, void*, char*, char*, int, int) {}
void __hipRegisterTexture(void**, void*, char*, char*, int, int, int) {}

}  // extern "C"
