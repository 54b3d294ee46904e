// CPU stand-in for the few HIP runtime entry points pyrad_amd/csrc/lbl_api.hip uses, for the sanitizer harness of the host
// shim (tests/host_shim/, tests/test_host_shim_asan_cpu.py).  TEST INFRASTRUCTURE: "device" memory is plain host memory (so that
// AddressSanitizer sees every copy the shim issues), streams and events are counters, graphs are empty.  Nothing here is ever
// linked into libpyrad_hip.so.
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <cstring>

typedef int hipError_t;
enum { hipSuccess = 0, hipErrorOutOfMemory = 2, hipErrorInvalidValue = 1 };
struct mock_stream { int id; bool capturing; };
struct mock_event { long long stamp; };
struct mock_graph { int nodes; };
typedef mock_stream* hipStream_t;
typedef mock_event* hipEvent_t;
typedef mock_graph* hipGraph_t;
typedef mock_graph* hipGraphExec_t;
struct int2 { int x, y; };
enum hipMemcpyKind { hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3, hipMemcpyDefault = 4 };
enum { hipStreamNonBlocking = 1, hipEventDisableTiming = 2, hipHostMallocDefault = 0 };
enum hipStreamCaptureMode { hipStreamCaptureModeGlobal = 0, hipStreamCaptureModeThreadLocal = 1 };
struct hipDeviceProp_t { char name[256]; int multiProcessorCount; size_t totalGlobalMem; char gcnArchName[256]; };

namespace mockhip {
extern int device_count;            // what hipGetDeviceCount reports (the driver sets it)
extern long long fail_malloc_at;    // the n-th hipMalloc from now fails with hipErrorOutOfMemory (-1: never)
extern long long clock;
extern long long live_allocs, live_streams, live_events, live_graphs, live_host;
}

inline const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "success" : e == hipErrorOutOfMemory ? "out of memory (mock)" : "error (mock)"; }
inline hipError_t hipGetLastError() { return hipSuccess; }
inline hipError_t hipGetDeviceCount(int* n) { *n = mockhip::device_count; return hipSuccess; }
inline hipError_t hipSetDevice(int d) { return d >= 0 && d < mockhip::device_count ? hipSuccess : hipErrorInvalidValue; }
inline hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int) {
    memset(p, 0, sizeof *p); strcpy(p->name, "mock gfx950"); strcpy(p->gcnArchName, "gfx950"); p->multiProcessorCount = 256;
    p->totalGlobalMem = (size_t)288 << 30; return hipSuccess;
}
inline hipError_t hipMalloc(void** p, size_t bytes) {
    if (mockhip::fail_malloc_at >= 0 && mockhip::fail_malloc_at-- == 0) { *p = nullptr; return hipErrorOutOfMemory; }
    if (posix_memalign(p, 256, bytes ? bytes : 1)) { *p = nullptr; return hipErrorOutOfMemory; }     // hipMalloc returns 256-byte aligned blocks
    memset(*p, 0xA5, bytes);           // a read of "uninitialised device memory" is at least not zeros
    ++mockhip::live_allocs; return hipSuccess;
}
inline hipError_t hipFree(void* p) { if (p) { free(p); --mockhip::live_allocs; } return hipSuccess; }
inline hipError_t hipHostMalloc(void** p, size_t bytes, unsigned = 0) { *p = malloc(bytes ? bytes : 1); if (!*p) return hipErrorOutOfMemory; ++mockhip::live_host; return hipSuccess; }
inline hipError_t hipHostFree(void* p) { if (p) { free(p); --mockhip::live_host; } return hipSuccess; }
inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { memmove(d, s, n); return hipSuccess; }
inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { memmove(d, s, n); return hipSuccess; }
inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { memset(d, v, n); return hipSuccess; }
inline hipError_t hipMemsetD32(void* d, int v, size_t count) { for (size_t i = 0; i < count; ++i) ((int*)d)[i] = v; return hipSuccess; }
inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = new mock_stream{0, false}; ++mockhip::live_streams; return hipSuccess; }
inline hipError_t hipStreamDestroy(hipStream_t s) { delete s; --mockhip::live_streams; return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
inline hipError_t hipEventCreate(hipEvent_t* e) { *e = new mock_event{0}; ++mockhip::live_events; return hipSuccess; }
inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { return hipEventCreate(e); }
inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; --mockhip::live_events; return hipSuccess; }
inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t) { e->stamp = ++mockhip::clock; return hipSuccess; }
inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) { *ms = (float)(b->stamp - a->stamp) * 1e-3f; return hipSuccess; }
inline hipError_t hipStreamBeginCapture(hipStream_t s, hipStreamCaptureMode) { s->capturing = true; return hipSuccess; }
inline hipError_t hipStreamEndCapture(hipStream_t s, hipGraph_t* g) { s->capturing = false; *g = new mock_graph{1}; ++mockhip::live_graphs; return hipSuccess; }
inline hipError_t hipGraphInstantiate(hipGraphExec_t* x, hipGraph_t, void*, void*, size_t) { *x = new mock_graph{1}; ++mockhip::live_graphs; return hipSuccess; }
inline hipError_t hipGraphLaunch(hipGraphExec_t, hipStream_t) { return hipSuccess; }
inline hipError_t hipGraphDestroy(hipGraph_t g) { delete g; --mockhip::live_graphs; return hipSuccess; }
inline hipError_t hipGraphExecDestroy(hipGraphExec_t g) { delete g; --mockhip::live_graphs; return hipSuccess; }
