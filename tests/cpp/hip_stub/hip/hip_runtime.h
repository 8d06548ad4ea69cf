// Stand-in for <hip/hip_runtime.h> used ONLY by tests/cpp/host_fuzz.cpp: lets the host side of the library
// (csrc/mtr_api.cpp, csrc/mtr_files.cpp) be compiled by g++ with AddressSanitizer / UBSan.  "Device memory" is the host
// heap, so every hipMemcpy the host code issues is bounds-checked by ASan; streams and events are inert tokens.
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <cstring>

#define __device__
#define __host__
#define __global__
#define __forceinline__ inline
using std::max;
using std::min;

typedef int hipError_t;
enum { hipSuccess = 0, hipErrorNotReady = 600 };
typedef struct hipStub_st* hipStream_t;
typedef struct hipStubEv_st* hipEvent_t;
enum { hipStreamNonBlocking = 1, hipEventDisableTiming = 2 };
enum hipMemcpyKind { hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3 };

inline const char* hipGetErrorString(hipError_t) { return "stub"; }
// tests inject a launch failure: the next hipGetLastError() of the SAME thread reports (and clears) it
inline int& hipStubInjectedError() { static thread_local int e = 0; return e; }
inline hipError_t hipGetLastError() { const int e = hipStubInjectedError(); hipStubInjectedError() = 0; return e; }
inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
inline hipError_t hipSetDevice(int) { return hipSuccess; }
inline hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
template <class T>
inline hipError_t hipMalloc(T** p, size_t n) { *p = static_cast<T*>(malloc(n ? n : 1)); return *p ? hipSuccess : 2; }
inline hipError_t hipFree(void* p) { free(p); return hipSuccess; }
inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { if (n) memcpy(d, s, n); return hipSuccess; }
inline hipError_t hipMemcpyPeerAsync(void* d, int, const void* s, int, size_t n, hipStream_t) { if (n) memcpy(d, s, n); return hipSuccess; }
inline hipError_t hipDeviceCanAccessPeer(int* can, int, int) { *can = 1; return hipSuccess; }
inline hipError_t hipDeviceEnablePeerAccess(int, unsigned) { return hipSuccess; }
inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { if (n) memset(d, v, n); return hipSuccess; }
inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = reinterpret_cast<hipStream_t>(malloc(1)); return hipSuccess; }
inline hipError_t hipStreamDestroy(hipStream_t s) { free(s); return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
inline hipError_t hipEventCreate(hipEvent_t* e) { *e = reinterpret_cast<hipEvent_t>(malloc(1)); return hipSuccess; }
inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { return hipEventCreate(e); }
inline hipError_t hipEventDestroy(hipEvent_t e) { free(e); return hipSuccess; }
inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventQuery(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.0f; return hipSuccess; }
enum { hipHostMallocMapped = 2, hipHostMallocCoherent = 0x40000000 };
inline hipError_t hipHostMalloc(void** p, size_t n, unsigned) { *p = calloc(1, n ? n : 1); return *p ? hipSuccess : 2; }
inline hipError_t hipHostFree(void* p) { free(p); return hipSuccess; }
inline hipError_t hipHostGetDevicePointer(void** dp, void* hp, unsigned) { *dp = hp; return hipSuccess; }
inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { if (n) memcpy(d, s, n); return hipSuccess; }
struct uint2 { unsigned x, y; };
inline uint2 make_uint2(unsigned a, unsigned b) { uint2 r = {a, b}; return r; }
struct uint4 { unsigned x, y, z, w; };
struct int4 { int x, y, z, w; };
inline uint4 make_uint4(unsigned a, unsigned b, unsigned c, unsigned d) { uint4 r = {a, b, c, d}; return r; }
inline int4 make_int4(int a, int b, int c, int d) { int4 r = {a, b, c, d}; return r; }
inline unsigned __float_as_uint(float f) { unsigned u; memcpy(&u, &f, 4); return u; }
inline float __uint_as_float(unsigned u) { float f; memcpy(&f, &u, 4); return f; }
