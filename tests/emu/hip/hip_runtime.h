// TEST INFRASTRUCTURE ONLY -- a minimal CPU stand-in for <hip/hip_runtime.h>.
//
// tests/emu/build_emu.py compiles pycwt_amd/csrc/*.hip with g++ and
// `-I tests/emu`, so that this header is picked up instead of the ROCm one.
// Every workgroup is run as a set of ucontext fibers that yield at
// __syncthreads(); workgroups are spread over OS threads.  This lets the
// `-m "not gpu"` tests execute the *unmodified* kernel sources (index
// algebra, LDS staging, launch geometry, host orchestration) on a box without
// a GPU.  It is never built into, loaded by, or reachable from the product
// library (pycwt_amd/libcwt_hip.so); the product has no CPU path.
#pragma once
#define CWT_HIP_EMULATED 1     // lets the product sources skip what only makes sense on a device (hardware-queue probe)
#include <ucontext.h>

#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <thread>
#include <vector>

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct double2 { double x, y; };
struct float2 { float x, y; };
struct alignas(16) float4 { float x, y, z, w; };
static inline double2 make_double2(double x, double y) { return double2{x, y}; }
static inline float2 make_float2(float x, float y) { return float2{x, y}; }

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)

namespace hipemu {
struct Ctx {
  dim3 tid, bid, bdim, gdim;
  char* smem;
  ucontext_t* self;
  ucontext_t* sched;
  bool done;
};
extern thread_local Ctx* cur;
inline void barrier() { swapcontext(cur->self, cur->sched); }
void launch(dim3 grid, dim3 block, size_t shmem, const std::function<void()>& body);
}  // namespace hipemu

#define threadIdx (hipemu::cur->tid)
#define blockIdx (hipemu::cur->bid)
#define blockDim (hipemu::cur->bdim)
#define gridDim (hipemu::cur->gdim)
#define __syncthreads() hipemu::barrier()
// wavefront-level sync: every fiber of the block runs the same barrier sequence, so the block-wide
// round-robin yield is a (stronger) stand-in
#define __builtin_amdgcn_wave_barrier() hipemu::barrier()
#define __builtin_amdgcn_fence(order, scope) ((void)0)
#define __builtin_amdgcn_s_waitcnt(imm) ((void)0)
#define __builtin_amdgcn_s_getreg(imm) (0u)
static inline unsigned long long wall_clock64() { return 0ull; }
static inline void sincospi(double x, double* s, double* c) {
  const double r = std::fmod(x, 2.0);                       // exact; keeps the argument of sin/cos small
  *s = std::sin(3.14159265358979323846 * r);
  *c = std::cos(3.14159265358979323846 * r);
}
#define __builtin_nontemporal_store(v, p) (*(p) = (v))
#define __builtin_nontemporal_load(p) (*(p))
#define __expf(x) expf(x)
#define HIP_DYNAMIC_SHARED(type, var) type* var = reinterpret_cast<type*>(hipemu::cur->smem);
#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
  hipemu::launch((grid), (block), (shmem), [=]() { kernel(__VA_ARGS__); })

static inline unsigned atomicAdd(unsigned* p, unsigned v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
static inline unsigned long long atomicAdd(unsigned long long* p, unsigned long long v) {
  return __atomic_fetch_add(p, v, __ATOMIC_RELAXED);
}

// ---- runtime API subset -------------------------------------------------
typedef int hipError_t;
typedef void* hipStream_t;
typedef struct hipemuEvent* hipEvent_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorNoDevice = 100 };
enum hipMemcpyKind { hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };
enum hipDeviceAttribute_t { hipDeviceAttributeMultiprocessorCount, hipDeviceAttributeMaxSharedMemoryPerBlock };
enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize };
struct hipemuEvent { double t; };

double hipemu_now_ms();
static inline hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
static inline hipError_t hipSetDevice(int) { return hipSuccess; }
static inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
static inline hipError_t hipMalloc(void** p, size_t n) {
  *p = nullptr;
  return posix_memalign(p, 256, n ? n : 256) == 0 ? hipSuccess : hipErrorInvalidValue;
}
static inline hipError_t hipFree(void* p) { free(p); return hipSuccess; }
constexpr unsigned hipHostMallocPortable = 0x1, hipHostMallocMapped = 0x2;
static inline hipError_t hipHostMalloc(void** p, size_t n, unsigned = 0) { return hipMalloc(p, n); }
static inline hipError_t hipHostFree(void* p) { free(p); return hipSuccess; }
static inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { memset(d, v, n); return hipSuccess; }
static inline hipError_t hipMemset(void* d, int v, size_t n) { memset(d, v, n); return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
enum { hipStreamNonBlocking = 1 };
static inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = nullptr; return hipSuccess; }
static inline hipError_t hipStreamCreateWithPriority(hipStream_t* s, unsigned, int) { *s = nullptr; return hipSuccess; }
static inline hipError_t hipDeviceGetStreamPriorityRange(int* least, int* greatest) { *least = 0; *greatest = 0; return hipSuccess; }
static inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
static inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
static inline hipError_t hipMemGetInfo(size_t* f, size_t* t) { *t = size_t(8) << 30; *f = size_t(6) << 30; return hipSuccess; }   // (a small card: 8 GiB)
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "hipSuccess" : "hipemu error"; }
static inline hipError_t hipEventCreate(hipEvent_t* e) { *e = new hipemuEvent{0}; return hipSuccess; }
enum { hipEventDisableTiming = 2 };
static inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = new hipemuEvent{0}; return hipSuccess; }
static inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t) { e->t = hipemu_now_ms(); return hipSuccess; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) { *ms = float(b->t - a->t); return hipSuccess; }
static inline hipError_t hipDeviceGetAttribute(int* v, hipDeviceAttribute_t a, int) {
  *v = (a == hipDeviceAttributeMultiprocessorCount) ? 256 : 160 * 1024;
  return hipSuccess;
}
template <typename F>
static inline hipError_t hipFuncSetAttribute(F, hipFuncAttribute, int) { return hipSuccess; }
// graphs: not emulated -- capture is refused, the library then keeps launching plainly
typedef struct hipemuGraph* hipGraph_t;
typedef struct hipemuGraphExec* hipGraphExec_t;
enum hipStreamCaptureStatus { hipStreamCaptureStatusNone = 0, hipStreamCaptureStatusActive, hipStreamCaptureStatusInvalidated };
static inline hipError_t hipStreamIsCapturing(hipStream_t, hipStreamCaptureStatus* s) { *s = hipStreamCaptureStatusNone; return hipSuccess; }
enum hipStreamCaptureMode { hipStreamCaptureModeGlobal, hipStreamCaptureModeThreadLocal, hipStreamCaptureModeRelaxed };
static inline hipError_t hipStreamBeginCapture(hipStream_t, hipStreamCaptureMode) { return hipErrorInvalidValue; }
static inline hipError_t hipStreamEndCapture(hipStream_t, hipGraph_t* g) { *g = nullptr; return hipErrorInvalidValue; }
static inline hipError_t hipGraphInstantiate(hipGraphExec_t* e, hipGraph_t, void*, void*, size_t) { *e = nullptr; return hipErrorInvalidValue; }
static inline hipError_t hipGraphDestroy(hipGraph_t) { return hipSuccess; }
static inline hipError_t hipGraphExecDestroy(hipGraphExec_t) { return hipSuccess; }
static inline hipError_t hipGraphLaunch(hipGraphExec_t, hipStream_t) { return hipErrorInvalidValue; }
