#include <stdlib.h>
#include <string.h>
#include <chrono>
#include <functional>
#include <mutex>
#include <string>
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
static bool ensure_lane(CmdList* c, int i);
static void use_lane(CmdList* c, int lane) {
  c->uses[lane] = true;
  (void)ensure_lane(c, lane);       // (without a device the list records and cmdlist_run refuses)
}
int dy_record(std::function<int(void*)> fn) {
  g_rec->cmds.push_back(Cmd{std::move(fn), 0, g_lane, 0, 0, nullptr});
  use_lane(g_rec, g_lane);
  return DISYOLO_OK;
}

// Side lanes are streams of ONE process-wide pool, created on first use (or ahead of time: disyolo_lanes_reserve) and
// shared by every list.  Round 5 found the step's speed hostage to how the runtime maps streams to hardware queues: a
// third, never used side stream per list made the hipGraph-captured B = 32 inference 23 % slower (6.09 k -> 4.66 k img/s
// same box); with a process group initialised first (torch's stream pool takes the hardware queues: later streams SHARE
// one, least-used first) a lane created afterwards can land on the caller's stream's queue and the step runs 2.5x
// slower.  So: no stream exists that no list uses, and a process that will initialise RCCL reserves its lanes BEFORE.
// lane 1 (weight gradients, detection filter, mask loss) runs beside the critical chain on the caller's stream: lowest
// stream priority, so the chain's kernels get the CU slots first when both have blocks pending (+1.5-2 % measured).
// DISYOLO_LANE1_LOW=0 keeps it at normal priority (the cut-list data-parallel step does: its RCCL all-reduces are issued on
// this lane).  lane 2 carries work that must only fill the other lanes' bubbles (the next step's backbone); lane 3 the
// data-parallel step's RCCL collectives and the optimizer sweeps behind them (lowest priority like lane 1: the HBM-bound sweeps
// must not compete with the main lane -- one RCCL rank: 4.22 -> 4.17 ms, the plain step's time; DISYOLO_LANE3_LOW=0: normal).
constexpr int MAXDEV = 16;
static hipStream_t g_pool_dev[MAXDEV][NLANES] = {};      // one pool per device (the product runs one process per GPU; tests may not)
static hipStream_t* cur_pool() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAXDEV) dev = 0;
  return g_pool_dev[dev];
}
#ifndef DY_HOST_ONLY
__global__ void lane_probe_spin_kernel(long long ticks) {     // wall_clock64: 100 MHz
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) {
  }
}
static double host_now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
// Two measurements of a stream pair (tools/micro/queue_probe.hip, profiles/r05_hw_queues.txt).  Streams map to hardware
// queues, queues to the command processor's pipes, and three kinds of neighbour exist for a given stream: one that runs
// beside it (two 100-us kernels: 110 us; a chain of 20 x 10-us kernels on `a` with `b` parked on an event behind it: 1.0x
// the chain alone), one that SHARES its queue (kernels serialise: 200 us) and one that runs beside it but stalls it while
// blocked on an event (the chain takes 1.8-2x: the pathological case -- a side lane spends most of its life parked on
// the main lane's events).  Which stream index is which changes with GPU_MAX_HW_QUEUES, priorities and how many streams
// the process already has, so the lanes are CHOSEN by measurement instead of by creation order.
// Every threshold is RELATIVE to a time measured in the same call, and every measurement is the minimum of several repeats
// (round 5 compared host wall-clock times with absolute bounds: under a profiler, a loaded host or another thread's GPU
// work every candidate failed them and the fallback could be exactly the queue-sharing stream that costs 2.5x):
//   single = one 100-us spin kernel on `a`, launch to completion;  both = one on `a` and one on `b` at once
//     -> beside each other: both ~ single; one hardware queue: both ~ 2 single.  Pass: both < 1.5 single.
//   chain_alone = 21 x 10-us kernels on `a`;  chain = 20 on `a`, `b` parked on an event behind them, 1 on `b`
//     -> pass: chain < 1.45 chain_alone (a neighbour that stalls the stream it is parked on takes 1.8-2x).
struct LaneProbe {
  double single_us = 0, both_us = 0, chain_us = 0, chain_alone_us = 0;
  bool ok = false;
};
static double timed_min(int repeats, const std::function<void()>& enqueue) {
  double best = 1e30;
  for (int k = 0; k < repeats; ++k) {
    (void)hipDeviceSynchronize();
    const double t0 = host_now();
    enqueue();
    (void)hipDeviceSynchronize();
    const double t = (host_now() - t0) * 1e6;
    if (t < best) best = t;
  }
  return best;
}
static double lane_chain_alone(hipStream_t a) {
  return timed_min(3, [&] {
    for (int r = 0; r < 21; ++r) hipLaunchKernelGGL(lane_probe_spin_kernel, dim3(1), dim3(64), 0, a, 10 * 100LL);
  });
}
static LaneProbe lane_pair_probe(hipStream_t a, hipStream_t b, double chain_alone_us) {
  LaneProbe r;
  r.chain_alone_us = chain_alone_us;
  hipEvent_t ev;
  if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) {
    r.ok = true;
    return r;
  }
  r.single_us = timed_min(3, [&] { hipLaunchKernelGGL(lane_probe_spin_kernel, dim3(1), dim3(64), 0, a, 100 * 100LL); });
  r.both_us = timed_min(3, [&] {
    hipLaunchKernelGGL(lane_probe_spin_kernel, dim3(1), dim3(64), 0, a, 100 * 100LL);
    hipLaunchKernelGGL(lane_probe_spin_kernel, dim3(1), dim3(64), 0, b, 100 * 100LL);
  });
  r.chain_us = timed_min(3, [&] {
    for (int k = 0; k < 20; ++k) hipLaunchKernelGGL(lane_probe_spin_kernel, dim3(1), dim3(64), 0, a, 10 * 100LL);
    (void)hipEventRecord(ev, a);
    (void)hipStreamWaitEvent(b, ev, 0);
    hipLaunchKernelGGL(lane_probe_spin_kernel, dim3(1), dim3(64), 0, b, 10 * 100LL);
  });
  (void)hipEventDestroy(ev);
  r.ok = r.both_us < 1.5 * r.single_us && r.chain_us < 1.45 * r.chain_alone_us;
  static const bool verbose = getenv("DISYOLO_LANE_PROBE") && getenv("DISYOLO_LANE_PROBE")[0] == '2';
  if (verbose)
    fprintf(stderr, "[lane probe] %p beside %p: single %.0f us, two at once %.0f us, chain %.0f us (alone %.0f): %s\n", (void*)b, (void*)a,
            r.single_us, r.both_us, r.chain_us, r.chain_alone_us, r.ok ? "ok" : "rejected");
  return r;
}
#endif
// what the pool did, per device and lane, for the bench line (disyolo_lanes_report): a slow run can be attributed
struct LaneChoice {
  int candidates = 0;        // streams created until one passed
  bool probed = false, fallback = false;
  double single_us = 0, both_us = 0, chain_us = 0, chain_alone_us = 0;     // the accepted (or last) candidate against the null stream
  int priority = 0;
};
static LaneChoice g_choice[MAXDEV][NLANES];
static std::mutex g_pool_mu;      // (the recorder state is thread-local, the pool is not)
static bool pool_lane(int i, hipStream_t caller = nullptr) {
  if (i <= 0 || i >= NLANES) return false;
  std::lock_guard<std::mutex> lock(g_pool_mu);
  hipStream_t* g_pool = cur_pool();
  if (g_pool[i]) return true;
  int devi = 0;
  if (hipGetDevice(&devi) != hipSuccess || devi < 0 || devi >= MAXDEV) devi = 0;
  LaneChoice& rep = g_choice[devi][i];
  int least = 0, greatest = 0;
  const char* l1 = getenv("DISYOLO_LANE1_LOW");
  const bool lane1_low = !(l1 && l1[0] == '0');
  if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) return false;   // no device (CPU-only build check)
  // lane 2 (the pipelined step's next-batch backbone): lowest priority as well since round 6 (+1.2 ... +1.5 % on the pipelined step;
  // DISYOLO_LANE2_LOW=0: normal)
  const bool lane2_low = !(getenv("DISYOLO_LANE2_LOW") && getenv("DISYOLO_LANE2_LOW")[0] == '0');
  const bool low = (i == 2 && lane2_low) || (i == 1 && lane1_low) || (i == 3 && !(getenv("DISYOLO_LANE3_LOW") && getenv("DISYOLO_LANE3_LOW")[0] == '0'));
  const int prio = low ? least : 0;
  rep.priority = prio;
#ifndef DY_HOST_ONLY
  const char* pe = getenv("DISYOLO_LANE_PROBE");
  bool probe = !(pe && pe[0] == '0');
  if (probe) {
    // the probe synchronises the device and launches on the null stream: illegal while a stream capture is active
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (caller && hipStreamIsCapturing(caller, &st) == hipSuccess && st != hipStreamCaptureStatusNone) {
      fprintf(stderr, "disyolo: lane %d is created while a stream capture is active: NOT probed (first stream taken; create the lanes before "
                      "capturing: disyolo_lanes_reserve)\n", i);
      probe = false;
    }
    (void)hipGetLastError();
    // (no stream in hand -- a lane first used while a list is being recorded: a device synchronise fails under any active
    //  capture, and then nothing below may run)
    if (probe && hipDeviceSynchronize() != hipSuccess) {
      (void)hipGetLastError();
      fprintf(stderr, "disyolo: lane %d: the device cannot be synchronised here (a stream capture is active?): NOT probed, first stream taken\n", i);
      probe = false;
    }
  }
  if (probe) {
    // candidates in creation order until one runs beside the caller's (null) stream and beside the lanes that exist, in both
    // roles; the rejected ones are released afterwards.  DISYOLO_LANE_PROBE=0: the first stream, unmeasured; =2: verbose
    hipStream_t rejected[16];
    int nrej = 0;
    hipStream_t pick = nullptr;
    const double alone0 = lane_chain_alone(nullptr);
    LaneProbe last;
    while (nrej < 16) {
      hipStream_t c = nullptr;
      if (hipStreamCreateWithPriority(&c, hipStreamNonBlocking, prio) != hipSuccess) break;
      ++rep.candidates;
      last = lane_pair_probe(nullptr, c, alone0);
      bool ok = last.ok && lane_pair_probe(c, nullptr, lane_chain_alone(c)).ok;
      for (int j = 1; j < NLANES && ok; ++j)
        if (g_pool[j]) ok = lane_pair_probe(g_pool[j], c, lane_chain_alone(g_pool[j])).ok && lane_pair_probe(c, g_pool[j], lane_chain_alone(c)).ok;
      if (ok) {
        pick = c;
        break;
      }
      rejected[nrej++] = c;
    }
    rep.probed = true;
    rep.single_us = last.single_us; rep.both_us = last.both_us; rep.chain_us = last.chain_us; rep.chain_alone_us = last.chain_alone_us;
    if (!pick && nrej > 0) {
      pick = rejected[--nrej];      // nothing passed: the last candidate (the step still runs) -- and say so
      rep.fallback = true;
      fprintf(stderr, "disyolo: lane %d: none of %d candidate streams ran cleanly beside the caller's stream (last: two 100-us kernels at once "
                      "%.0f us against %.0f alone, chain %.0f us against %.0f alone); keeping the last one -- the step may be serialised or "
                      "stalled by this lane (DISYOLO_LANE_PROBE=2 prints every candidate)\n", i, rep.candidates, last.both_us, last.single_us,
              last.chain_us, last.chain_alone_us);
    }
    for (int k = 0; k < nrej; ++k) (void)hipStreamDestroy(rejected[k]);
    (void)hipGetLastError();
    if (pick) {
      g_pool[i] = pick;
      return true;
    }
    return false;
  }
#endif
  if (hipStreamCreateWithPriority(&g_pool[i], hipStreamNonBlocking, prio) != hipSuccess) {
    g_pool[i] = nullptr;
    return false;
  }
  rep.candidates = 1;
  return true;
}
// one line per lane of the current device that exists: how it was chosen (bench.py puts it into config.box)
extern "C" int disyolo_lanes_report(char* buf, int size) {
  if (!buf || size <= 0) return DISYOLO_E_ARG;
  int devi = 0;
  if (hipGetDevice(&devi) != hipSuccess || devi < 0 || devi >= MAXDEV) devi = 0;
  (void)hipGetLastError();
  std::string out;
  std::lock_guard<std::mutex> lock(g_pool_mu);
  for (int i = 1; i < NLANES; ++i) {
    if (!g_pool_dev[devi][i]) continue;
    const LaneChoice& r = g_choice[devi][i];
    char line[256];
    snprintf(line, sizeof line, "lane %d: priority %d, %s, candidates %d%s, single %.0f us, two at once %.0f us, chain %.0f us (alone %.0f)\n", i,
             r.priority, r.probed ? "probed" : "unprobed", r.candidates, r.fallback ? ", FALLBACK (none passed)" : "", r.single_us, r.both_us,
             r.chain_us, r.chain_alone_us);
    out += line;
  }
  snprintf(buf, (size_t)size, "%s", out.c_str());
  return DISYOLO_OK;
}
static bool ensure_lane(CmdList* c, int i) {
  if (i <= 0 || i >= NLANES) return i == 0;
  if (c->side[i]) return true;
  if (!pool_lane(i)) return false;
  if (hipEventCreateWithFlags(&c->ev_join[i], hipEventDisableTiming) != hipSuccess) return false;
  c->side[i] = cur_pool()[i];
  return true;
}
// create the side streams of the lanes in `mask` (bit i = lane i) now, on the current device: call before anything else
// that creates many streams (torch.distributed's NCCL backend) so the lanes get hardware queues of their own
extern "C" int disyolo_lanes_reserve(int mask) {
  for (int i = 1; i < NLANES; ++i)
    if ((mask >> i) & 1) DY_REQUIRE(pool_lane(i), "lanes_reserve: cannot create the stream of lane %d", i);
  return DISYOLO_OK;
}

extern "C" void* disyolo_cmdlist_create(void) {
  CmdList* c = new CmdList();
  (void)hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming);   // (fails without a device: lists can still be recorded, not run)
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
  for (int i = 1; i < NLANES; ++i)
    if (c->ev_join[i]) (void)hipEventDestroy(c->ev_join[i]);      // (the streams belong to the process-wide pool)
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
  use_lane(g_rec, from);
  use_lane(g_rec, to);
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
  use_lane(g_rec, lane);
  return (int)g_rec->cmds.size() - 1;
}
extern "C" int disyolo_cmdlist_wait(int mark, int lane) {
  if (!g_rec) return DISYOLO_OK;
  DY_REQUIRE(lane >= 0 && lane < NLANES, "cmdlist_wait: bad lane");
  DY_REQUIRE(mark >= 0 && mark < (int)g_rec->cmds.size() && g_rec->cmds[mark].kind == 2, "cmdlist_wait: %d is not a mark of this list", mark);
  g_rec->cmds.push_back(Cmd{nullptr, 3, 0, mark, lane, nullptr});
  use_lane(g_rec, lane);
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
  use_lane(g_rec, lane);
  return DISYOLO_OK;
}
extern "C" int disyolo_cmdlist_wait_slot(int slot, int lane) {
  DY_REQUIRE(lane >= 0 && lane < NLANES && slot >= 0 && slot < NSLOTS, "cmdlist_wait_slot: lane 0..%d, slot 0..%d", NLANES - 1, NSLOTS - 1);
  if (!g_rec) return DISYOLO_OK;
  g_rec->cmds.push_back(Cmd{nullptr, 5, 0, slot, lane, nullptr});
  use_lane(g_rec, lane);
  return DISYOLO_OK;
}
extern "C" int disyolo_cmdlist_end(void) {
  DY_REQUIRE(g_rec, "cmdlist_end: not recording");
  g_rec = nullptr;
  return DISYOLO_OK;
}
extern "C" int disyolo_cmdlist_size(void* l) { return l ? (int)((CmdList*)l)->cmds.size() : DISYOLO_E_ARG; }
// how many queue packets of a kind a replay puts on `lane`: what = 0 launches, 1 event records (marks, slot marks, the record
// half of a sync), 2 waits (waits, slot waits, the wait half of a sync).  An event packet costs its stream ~3 us
// (tools/micro/event_cost.hip): a step is designed against these counts, and tests pin them.
extern "C" int disyolo_cmdlist_count(void* l, int what, int lane) {
  if (!l || what < 0 || what > 2 || lane < 0 || lane >= NLANES) return DISYOLO_E_ARG;
  int n = 0;
  for (const Cmd& k : ((CmdList*)l)->cmds) {
    if (k.kind == 0) n += what == 0 && k.lane == lane;
    else if (k.kind == 1) n += (what == 1 && k.from == lane) + (what == 2 && k.to == lane);
    else if (k.kind == 2 || k.kind == 4) n += what == 1 && k.from == lane;
    else n += what == 2 && k.to == lane;      // kinds 3, 5
  }
  return n;
}
extern "C" void* disyolo_cmdlist_lane_stream(void* l, int lane) {
  if (!l || lane < 1 || lane >= NLANES || !ensure_lane((CmdList*)l, lane)) return nullptr;
  return (void*)((CmdList*)l)->side[lane];
}
extern "C" void* disyolo_cmdlist_side_stream(void* l) { return disyolo_cmdlist_lane_stream(l, 1); }

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
