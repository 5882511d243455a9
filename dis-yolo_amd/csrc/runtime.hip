#include <stdlib.h>
#include <string.h>
#include <utility>
#include <vector>
#include "common.h"
#include "runtime.h"

namespace {
// A command runs on lane 0 (the caller's stream) or lane 1 (a side stream owned by the
// list).  kind 0 = launch, 1 = "lane `to` waits for everything lane `from` has enqueued".
struct Cmd {
  std::function<int(void*)> fn;
  int kind, lane, from, to;
  hipEvent_t ev;
};
constexpr int NLANES = 4;   // lane 0 = the caller's stream, lanes 1.. = side streams owned by the list (3 = the gradient exchange)
constexpr int NSLOTS = 16;  // named cross-replay marks (cmdlist_mark_slot / cmdlist_wait_slot)
struct CmdList {
  std::vector<Cmd> cmds;
  hipEvent_t slot_ev[NSLOTS] = {};
  bool slot_recorded[NSLOTS] = {};
  hipStream_t side[NLANES] = {};
  hipEvent_t ev_fork = nullptr, ev_join[NLANES] = {};
  bool uses[NLANES] = {true};
};
thread_local CmdList* g_rec = nullptr;
thread_local int g_lane = 0;

#ifndef DY_HOST_ONLY
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
#endif
}  // namespace

// CRC-32C (Castagnoli), slicing-by-8: host helper of the checkpoint writer / reader (TensorFlow tensor
// bundles carry a masked crc32c per tensor and per index block; a 247 MB stage-2 checkpoint takes ~0.2 s)
namespace {
struct Crc32cTables {
  uint32_t t[8][256];
  Crc32cTables() {
    for (uint32_t i = 0; i < 256; ++i) {
      uint32_t c = i;
      for (int k = 0; k < 8; ++k) c = (c >> 1) ^ ((c & 1) ? 0x82F63B78u : 0u);
      t[0][i] = c;
    }
    for (int k = 1; k < 8; ++k)
      for (uint32_t i = 0; i < 256; ++i) t[k][i] = (t[k - 1][i] >> 8) ^ t[0][t[k - 1][i] & 0xff];
  }
};
}  // namespace
extern "C" uint32_t disyolo_crc32c(const void* data, size_t n, uint32_t crc) {
  static const Crc32cTables T;
  const unsigned char* p = (const unsigned char*)data;
  uint32_t c = crc ^ 0xffffffffu;
  while (n >= 8) {
    uint32_t a, b;
    memcpy(&a, p, 4);
    memcpy(&b, p + 4, 4);
    a ^= c;
    c = T.t[7][a & 0xff] ^ T.t[6][(a >> 8) & 0xff] ^ T.t[5][(a >> 16) & 0xff] ^ T.t[4][a >> 24] ^ T.t[3][b & 0xff] ^
        T.t[2][(b >> 8) & 0xff] ^ T.t[1][(b >> 16) & 0xff] ^ T.t[0][b >> 24];
    p += 8;
    n -= 8;
  }
  while (n--) c = T.t[0][(c ^ *p++) & 0xff] ^ (c >> 8);
  return c ^ 0xffffffffu;
}

bool dy_recording() { return g_rec != nullptr; }
int dy_record(std::function<int(void*)> fn) {
  g_rec->cmds.push_back(Cmd{std::move(fn), 0, g_lane, 0, 0, nullptr});
  g_rec->uses[g_lane] = true;
  return DISYOLO_OK;
}

extern "C" void* disyolo_cmdlist_create(void) {
  CmdList* c = new CmdList();
  bool ok = hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming) == hipSuccess;
  // lane 1 (weight gradients, detection filter, mask loss) runs beside the critical chain on the
  // caller's stream: lowest stream priority, so the chain's kernels get the CU slots first when
  // both have blocks pending (+1.5-2 % measured).  DISYOLO_LANE1_LOW=0 keeps it at normal priority
  // (the data-parallel step does: its RCCL all-reduces are issued on this lane).
  // lane 2 carries work that must only fill the other lanes' bubbles (the next step's backbone)
  // lane 3 carries the data-parallel step's RCCL collectives and the optimizer sweeps behind them (normal priority)
  int least = 0, greatest = 0;
  const char* l1 = getenv("DISYOLO_LANE1_LOW");
  const bool lane1_low = !(l1 && l1[0] == '0');
  if (ok) ok = hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess;
  for (int i = 1; i < NLANES && ok; ++i)
    ok = hipStreamCreateWithPriority(&c->side[i], hipStreamNonBlocking,
                                     ((i == 2 && getenv("DISYOLO_LANE2_LOW")) || (i == 1 && lane1_low)) ? least : 0) == hipSuccess &&
         hipEventCreateWithFlags(&c->ev_join[i], hipEventDisableTiming) == hipSuccess;
  if (!ok)  // no device (CPU-only build check): lists can still be recorded, not run
    for (int i = 1; i < NLANES; ++i) c->side[i] = nullptr;
  return c;
}
extern "C" void disyolo_cmdlist_destroy(void* l) {
  CmdList* c = (CmdList*)l;
  if (!c) return;
  for (Cmd& k : c->cmds)
    if (k.ev) (void)hipEventDestroy(k.ev);
  if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
  for (int i = 0; i < NSLOTS; ++i)
    if (c->slot_ev[i]) (void)hipEventDestroy(c->slot_ev[i]);
  for (int i = 1; i < NLANES; ++i) {
    if (c->ev_join[i]) (void)hipEventDestroy(c->ev_join[i]);
    if (c->side[i]) (void)hipStreamDestroy(c->side[i]);
  }
  delete c;
}
extern "C" int disyolo_cmdlist_begin(void* l) {
  DY_REQUIRE(l && !g_rec, "cmdlist_begin: null list or already recording");
  g_rec = (CmdList*)l;
  g_lane = 0;
  return DISYOLO_OK;
}
// lane for the following launches of the recording thread (no-op when not recording: the
// per-call path runs everything on the caller's stream, which is trivially ordered)
extern "C" int disyolo_cmdlist_set_lane(int lane) {
  DY_REQUIRE(lane >= 0 && lane < NLANES, "cmdlist_set_lane: lane must be 0..%d", NLANES - 1);
  if (g_rec) g_lane = lane;
  return DISYOLO_OK;
}
// lane `to` waits for everything recorded so far on lane `from` (no-op when not recording)
extern "C" int disyolo_cmdlist_sync(int from, int to) {
  DY_REQUIRE(from >= 0 && from < NLANES && to >= 0 && to < NLANES && from != to, "cmdlist_sync: bad lanes");
  if (!g_rec) return DISYOLO_OK;
  hipEvent_t ev;
  if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) {
    disyolo_set_error("cmdlist_sync: hipEventCreate failed");
    return DISYOLO_E_HIP;
  }
  g_rec->cmds.push_back(Cmd{nullptr, 1, 0, from, to, ev});
  g_rec->uses[from] = g_rec->uses[to] = true;
  return DISYOLO_OK;
}
// mark: remember this point of lane `lane`; wait: lane `lane` waits for a mark made EARLIER in the list -- unlike
// cmdlist_sync the waiting lane does not wait for what the marked lane recorded after the mark.  Outside a
// recording both are no-ops (mark returns -1).
extern "C" int disyolo_cmdlist_mark(int lane) {
  if (lane < 0 || lane >= NLANES) return DISYOLO_E_ARG;
  if (!g_rec) return -1;
  hipEvent_t ev;
  if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) {
    disyolo_set_error("cmdlist_mark: hipEventCreate failed");
    return DISYOLO_E_HIP;
  }
  g_rec->cmds.push_back(Cmd{nullptr, 2, 0, lane, 0, ev});
  g_rec->uses[lane] = true;
  return (int)g_rec->cmds.size() - 1;
}
extern "C" int disyolo_cmdlist_wait(int mark, int lane) {
  if (!g_rec) return DISYOLO_OK;
  DY_REQUIRE(lane >= 0 && lane < NLANES, "cmdlist_wait: bad lane");
  DY_REQUIRE(mark >= 0 && mark < (int)g_rec->cmds.size() && g_rec->cmds[mark].kind == 2, "cmdlist_wait: %d is not a mark of this list", mark);
  g_rec->cmds.push_back(Cmd{nullptr, 3, 0, mark, lane, nullptr});
  g_rec->uses[lane] = true;
  return DISYOLO_OK;
}
// Named marks that outlive a replay.  mark_slot: remember this point of `lane` under `slot`; wait_slot: `lane` waits for
// the point the slot was LAST marked at -- by a mark earlier in this replay or, where the wait comes first in the list,
// by the PREVIOUS replay of the list (nothing to wait for on the first one).  This is how a step's tail on the side
// lane (optimizer sweeps, the last weight gradients) overlaps the next step's locked-backbone forward on the main
// lane: the next replay waits, tensor by tensor, only for the side-lane work that still reads or writes it.
extern "C" int disyolo_cmdlist_mark_slot(int lane, int slot) {
  DY_REQUIRE(lane >= 0 && lane < NLANES && slot >= 0 && slot < NSLOTS, "cmdlist_mark_slot: lane 0..%d, slot 0..%d", NLANES - 1, NSLOTS - 1);
  if (!g_rec) return DISYOLO_OK;
  if (!g_rec->slot_ev[slot] && hipEventCreateWithFlags(&g_rec->slot_ev[slot], hipEventDisableTiming) != hipSuccess) {
    disyolo_set_error("cmdlist_mark_slot: hipEventCreate failed");
    return DISYOLO_E_HIP;
  }
  g_rec->cmds.push_back(Cmd{nullptr, 4, 0, lane, slot, nullptr});
  g_rec->uses[lane] = true;
  return DISYOLO_OK;
}
extern "C" int disyolo_cmdlist_wait_slot(int slot, int lane) {
  DY_REQUIRE(lane >= 0 && lane < NLANES && slot >= 0 && slot < NSLOTS, "cmdlist_wait_slot: lane 0..%d, slot 0..%d", NLANES - 1, NSLOTS - 1);
  if (!g_rec) return DISYOLO_OK;
  g_rec->cmds.push_back(Cmd{nullptr, 5, 0, slot, lane, nullptr});
  g_rec->uses[lane] = true;
  return DISYOLO_OK;
}
extern "C" int disyolo_cmdlist_end(void) {
  DY_REQUIRE(g_rec, "cmdlist_end: not recording");
  g_rec = nullptr;
  return DISYOLO_OK;
}
extern "C" int disyolo_cmdlist_size(void* l) { return l ? (int)((CmdList*)l)->cmds.size() : DISYOLO_E_ARG; }
extern "C" void* disyolo_cmdlist_side_stream(void* l) { return l ? (void*)((CmdList*)l)->side[1] : nullptr; }
extern "C" void* disyolo_cmdlist_lane_stream(void* l, int lane) {
  return (l && lane >= 1 && lane < NLANES) ? (void*)((CmdList*)l)->side[lane] : nullptr;
}

extern "C" int disyolo_cmdlist_run(void* l, int first, int last, void* stream) {
  return disyolo_cmdlist_run_ex(l, first, last, stream, 3);
}

// flags: bit 0 = fork (side lane first waits for the caller's stream), bit 1 = join (the
// caller's stream finally waits for the side lane).  A step replayed in several ranges forks
// in the first and joins in the last, so the lanes keep running across the cuts.
extern "C" int disyolo_cmdlist_run_ex(void* l, int first, int last, void* stream, int flags) {
  DY_REQUIRE(l && !g_rec, "cmdlist_run: null list or called while recording");
  CmdList* c = (CmdList*)l;
  DY_REQUIRE(first >= 0 && last <= (int)c->cmds.size() && first <= last, "cmdlist_run: bad range [%d,%d)", first, last);
  hipStream_t lanes[NLANES];
  lanes[0] = (hipStream_t)stream;
  for (int i = 1; i < NLANES; ++i) {
    lanes[i] = c->side[i];
    if (c->uses[i]) DY_REQUIRE(c->side[i], "cmdlist_run: side stream unavailable");
  }
  if (flags & 1) {
    bool any = false;
    for (int i = 1; i < NLANES; ++i) any = any || c->uses[i];
    if (any) {
      if (hipEventRecord(c->ev_fork, lanes[0]) != hipSuccess) {
        disyolo_set_error("cmdlist_run: fork failed");
        return DISYOLO_E_HIP;
      }
      for (int i = 1; i < NLANES; ++i)
        if (c->uses[i] && hipStreamWaitEvent(lanes[i], c->ev_fork, 0) != hipSuccess) {
          disyolo_set_error("cmdlist_run: fork failed");
          return DISYOLO_E_HIP;
        }
    }
  }
  // DISYOLO_LANE_TIMING=1 (diagnostic): time how long the caller's stream sits in each wait for a side lane --
  // a timed event before and after the wait; reported (and the device synchronised) at the end of the range
  static const bool lane_timing = getenv("DISYOLO_LANE_TIMING") && getenv("DISYOLO_LANE_TIMING")[0] == '1';
  std::vector<std::pair<int, std::pair<hipEvent_t, hipEvent_t>>> waits;
  for (int i = first; i < last; ++i) {
    Cmd& k = c->cmds[i];
    if (k.kind == 0) {
      const int rc = k.fn(lanes[k.lane]);
      if (rc) return rc;
    } else if (k.kind == 2) {
      if (hipEventRecord(k.ev, lanes[k.from]) != hipSuccess) {
        disyolo_set_error("cmdlist_run: mark failed");
        return DISYOLO_E_HIP;
      }
    } else if (k.kind == 4) {
      if (hipEventRecord(c->slot_ev[k.to], lanes[k.from]) != hipSuccess) {
        disyolo_set_error("cmdlist_run: slot mark failed");
        return DISYOLO_E_HIP;
      }
      c->slot_recorded[k.to] = true;
    } else if (k.kind == 5) {
      if (c->slot_recorded[k.from] && hipStreamWaitEvent(lanes[k.to], c->slot_ev[k.from], 0) != hipSuccess) {
        disyolo_set_error("cmdlist_run: slot wait failed");
        return DISYOLO_E_HIP;
      }
    } else if (k.kind == 3) {
      // (a range that starts behind the mark waits on the event of the previous replay: replay whole steps)
      hipEvent_t e0 = nullptr, e1 = nullptr;
      if (lane_timing && k.to == 0) {
        (void)hipEventCreate(&e0);
        (void)hipEventCreate(&e1);
        (void)hipEventRecord(e0, lanes[0]);
      }
      if (hipStreamWaitEvent(lanes[k.to], c->cmds[k.from].ev, 0) != hipSuccess) {
        disyolo_set_error("cmdlist_run: wait failed");
        return DISYOLO_E_HIP;
      }
      if (e0) {
        (void)hipEventRecord(e1, lanes[0]);
        waits.push_back({i, {e0, e1}});
      }
    } else {
      hipEvent_t e0 = nullptr, e1 = nullptr;
      if (lane_timing && k.to == 0) {
        (void)hipEventCreate(&e0);
        (void)hipEventCreate(&e1);
        (void)hipEventRecord(e0, lanes[0]);
      }
      if (hipEventRecord(k.ev, lanes[k.from]) != hipSuccess || hipStreamWaitEvent(lanes[k.to], k.ev, 0) != hipSuccess) {
        disyolo_set_error("cmdlist_run: lane sync failed");
        return DISYOLO_E_HIP;
      }
      if (e0) {
        (void)hipEventRecord(e1, lanes[0]);
        waits.push_back({i, {e0, e1}});
      }
    }
  }
  if (!waits.empty()) {
    (void)hipDeviceSynchronize();
    float tot = 0.f;
    for (auto& w : waits) {
      float ms = 0.f;
      (void)hipEventElapsedTime(&ms, w.second.first, w.second.second);
      tot += ms;
      if (ms > 0.02f) fprintf(stderr, "[lane timing] main lane waited %.1f us at command %d\n", ms * 1e3f, w.first);
      (void)hipEventDestroy(w.second.first);
      (void)hipEventDestroy(w.second.second);
    }
    fprintf(stderr, "[lane timing] total main-lane wait in this range: %.1f us\n", tot * 1e3f);
  }
  if (flags & 2) {
    for (int i = 1; i < NLANES; ++i)
      if (c->uses[i] && (hipEventRecord(c->ev_join[i], lanes[i]) != hipSuccess ||
                         hipStreamWaitEvent(lanes[0], c->ev_join[i], 0) != hipSuccess)) {
        disyolo_set_error("cmdlist_run: join failed");
        return DISYOLO_E_HIP;
      }
  }
  return DISYOLO_OK;
}

// dst (+)= src, bf16, n % 8 == 0: gradient accumulation at residual shortcuts
extern "C" int disyolo_add_bf16(const void* src, void* dst, int64_t n, int accumulate, void* stream) {
  DY_REQUIRE(src && dst && n > 0 && n % 8 == 0, "add_bf16: bad args");
  DY_RECORD_OR_RUN([=](void* s) { return disyolo_add_bf16(src, dst, n, accumulate, s); });
#ifdef DY_HOST_ONLY      // (`make asan`: no device code in the host-only build)
  (void)stream;
  return DISYOLO_OK;
#else
  int64_t g = (n / 8 + 255) / 256;
  if (g > 4096) g = 4096;
  hipLaunchKernelGGL(add_bf16_kernel, dim3((int)g), dim3(256), 0, (hipStream_t)stream, (const uint4*)src, (uint4*)dst,
                     n / 8, accumulate);
  DY_CHECK_LAUNCH();
  return DISYOLO_OK;
#endif
}
