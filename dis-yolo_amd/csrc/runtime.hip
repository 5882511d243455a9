#include <vector>
#include "common.h"
#include "runtime.h"

namespace {
struct CmdList {
  std::vector<std::function<int(void*)>> cmds;
};
thread_local CmdList* g_rec = nullptr;

__global__ __launch_bounds__(256) void add_bf16_kernel(const uint4* src, uint4* dst, int64_t nvec, int accumulate) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += stride) {
    uint4 v = src[i];
    if (accumulate) {
      float a[8], b[8];
      unpack8(v, a);
      unpack8(dst[i], b);
#pragma unroll
      for (int k = 0; k < 8; ++k) a[k] += b[k];
      v = pack8(a);
    }
    dst[i] = v;
  }
}
}  // namespace

bool dy_recording() { return g_rec != nullptr; }
int dy_record(std::function<int(void*)> fn) {
  g_rec->cmds.push_back(std::move(fn));
  return DISYOLO_OK;
}

extern "C" void* disyolo_cmdlist_create(void) { return new CmdList(); }
extern "C" void disyolo_cmdlist_destroy(void* l) { delete (CmdList*)l; }
extern "C" int disyolo_cmdlist_begin(void* l) {
  DY_REQUIRE(l && !g_rec, "cmdlist_begin: null list or already recording");
  g_rec = (CmdList*)l;
  return DISYOLO_OK;
}
extern "C" int disyolo_cmdlist_end(void) {
  DY_REQUIRE(g_rec, "cmdlist_end: not recording");
  g_rec = nullptr;
  return DISYOLO_OK;
}
extern "C" int disyolo_cmdlist_size(void* l) { return l ? (int)((CmdList*)l)->cmds.size() : DISYOLO_E_ARG; }
extern "C" int disyolo_cmdlist_run(void* l, int first, int last, void* stream) {
  DY_REQUIRE(l && !g_rec, "cmdlist_run: null list or called while recording");
  CmdList* c = (CmdList*)l;
  DY_REQUIRE(first >= 0 && last <= (int)c->cmds.size() && first <= last, "cmdlist_run: bad range [%d,%d)", first, last);
  for (int i = first; i < last; ++i) {
    const int rc = c->cmds[i](stream);
    if (rc) return rc;
  }
  return DISYOLO_OK;
}

// dst (+)= src, bf16, n % 8 == 0: gradient accumulation at residual shortcuts
extern "C" int disyolo_add_bf16(const void* src, void* dst, int64_t n, int accumulate, void* stream) {
  DY_REQUIRE(src && dst && n > 0 && n % 8 == 0, "add_bf16: bad args");
  DY_RECORD_OR_RUN([=](void* s) { return disyolo_add_bf16(src, dst, n, accumulate, s); });
  int64_t g = (n / 8 + 255) / 256;
  if (g > 4096) g = 4096;
  hipLaunchKernelGGL(add_bf16_kernel, dim3((int)g), dim3(256), 0, (hipStream_t)stream, (const uint4*)src, (uint4*)dst,
                     n / 8, accumulate);
  DY_CHECK_LAUNCH();
  return DISYOLO_OK;
}
