// How do streams created one after the other behave against the null stream?  For each of N side streams (created in order,
// normal priority unless argv says otherwise): (a) overlap: a 200-us one-block spin kernel on each of the two streams at
// once -- 200 us if they run concurrently, 400 if serialised; (b) ping-pong: 50 rounds of { main: short kernel, record;
// side: wait, short kernel, record; main: wait } -- microseconds per round trip.
// hipcc -O2 --offload-arch=gfx950 tools/micro/queue_probe.hip -o tools/bin/queue_probe ; GPU_MAX_HW_QUEUES=8 tools/bin/queue_probe 12
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
__global__ void spin(long long cycles) {
  long long t0 = wall_clock64();
  while (wall_clock64() - t0 < cycles) {}
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 12;
  const int lowfirst = argc > 2 ? atoi(argv[2]) : 0;   // 1: stream 0 is created with the lowest priority (like lane 1)
  int least = 0, greatest = 0;
  hipDeviceGetStreamPriorityRange(&least, &greatest);
  std::vector<hipStream_t> st(n);
  for (int i = 0; i < n; ++i) hipStreamCreateWithPriority(&st[i], hipStreamNonBlocking, (lowfirst && i == 0) ? least : 0);
  const long long us200 = 200 * 100;   // wall_clock64 runs at 100 MHz
  hipEvent_t ea, eb;
  hipEventCreateWithFlags(&ea, hipEventDisableTiming);
  hipEventCreateWithFlags(&eb, hipEventDisableTiming);
  spin<<<1, 64, 0, 0>>>(100);
  hipDeviceSynchronize();
  for (int i = 0; i < n; ++i) {
    spin<<<1, 64, 0, st[i]>>>(100);
    hipDeviceSynchronize();
    double t0 = now();
    spin<<<1, 64, 0, 0>>>(us200);
    spin<<<1, 64, 0, st[i]>>>(us200);
    hipDeviceSynchronize();
    const double overlap = (now() - t0) * 1e6;
    t0 = now();
    for (int r = 0; r < 50; ++r) {
      spin<<<1, 64, 0, 0>>>(100);
      hipEventRecord(ea, 0);
      hipStreamWaitEvent(st[i], ea, 0);
      spin<<<1, 64, 0, st[i]>>>(100);
      hipEventRecord(eb, st[i]);
      hipStreamWaitEvent(0, eb, 0);
    }
    hipDeviceSynchronize();
    const double pp = (now() - t0) * 1e6 / 50;
    // side stream blocked on an event while the main stream has a backlog: main enqueues 20 kernels, the side stream waits for
    // the LAST one's event first, then runs one kernel; total time against the 20 kernels alone
    t0 = now();
    for (int r = 0; r < 20; ++r) spin<<<1, 64, 0, 0>>>(1000);
    hipEventRecord(ea, 0);
    hipStreamWaitEvent(st[i], ea, 0);
    spin<<<1, 64, 0, st[i]>>>(1000);
    hipDeviceSynchronize();
    const double blocked = (now() - t0) * 1e6;
    printf("stream %2d: two 200-us kernels %.0f us | ping-pong %.1f us per round | 21 x 10-us chain with a blocked side stream %.0f us\n", i, overlap, pp, blocked);
  }
  // all side streams busy at once with the main stream
  double t0 = now();
  spin<<<1, 64, 0, 0>>>(us200);
  for (int i = 0; i < n; ++i) spin<<<1, 64, 0, st[i]>>>(us200);
  hipDeviceSynchronize();
  printf("main + %d side streams, one 200-us kernel each: %.0f us\n", n, (now() - t0) * 1e6);
  return 0;
}
