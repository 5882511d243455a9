// Host-only stand-in for <hip/hip_runtime.h>, used by `make asan` ONLY (never by the product build): lets the
// library's host-side code -- the command-list executor (runtime.hip), the contour tracer (contours.hip), crc32c --
// be compiled as plain C++ with -fsanitize=address,undefined on a machine without a GPU.  Streams and events are
// counters; every call succeeds; ordering calls are recorded so the driver can check what the executor issued.
#pragma once
#include <stdint.h>
#include <stdlib.h>
#include <vector>
typedef int hipError_t;
enum { hipSuccess = 0, hipErrorUnknown = 999 };
struct HostStubEvent { int id; int recorded_on; };
struct HostStubStream { int id; };
typedef HostStubStream* hipStream_t;
typedef HostStubEvent* hipEvent_t;
enum { hipEventDisableTiming = 2, hipStreamNonBlocking = 1 };
struct HostStubLog { std::vector<int> ops; };           // op codes: 1 record, 2 wait, 3 sync
inline HostStubLog& host_stub_log() { static HostStubLog l; return l; }
inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = new HostStubEvent{0, -1}; return hipSuccess; }
inline hipError_t hipEventCreate(hipEvent_t* e) { return hipEventCreateWithFlags(e, 0); }
inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t s) { e->recorded_on = s ? s->id : 0; host_stub_log().ops.push_back(1); return hipSuccess; }
inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t e, unsigned) { (void)e->recorded_on; host_stub_log().ops.push_back(2); return hipSuccess; }
inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.f; return hipSuccess; }
inline hipError_t hipDeviceGetStreamPriorityRange(int* least, int* greatest) { *least = 0; *greatest = -1; return hipSuccess; }
inline hipError_t hipStreamCreateWithPriority(hipStream_t* s, unsigned, int) { static int n = 1; *s = new HostStubStream{n++}; return hipSuccess; }
inline hipError_t hipStreamDestroy(hipStream_t s) { delete s; return hipSuccess; }
inline hipError_t hipDeviceSynchronize() { host_stub_log().ops.push_back(3); return hipSuccess; }
inline hipError_t hipGetLastError() { return hipSuccess; }
inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
inline const char* hipGetErrorString(hipError_t) { return "host stub"; }
