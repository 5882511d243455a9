// What does an event edge cost the stream that RECORDS it?  A chain of 100 x 10-us kernels on the null stream: alone; with a
// hipEventRecord (timing disabled) after every kernel; with a side stream waiting on every one of those events and running a
// 10-us kernel behind it (the step's weight-gradient edge); the same with one event per 3 kernels.
// hipcc -O2 --offload-arch=gfx950 tools/micro/event_cost.hip -o tools/bin/event_cost
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
__global__ void spin(long long ticks) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) {
  }
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  const int N = 100;
  hipStream_t side;
  int least = 0, greatest = 0;
  (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
  (void)hipStreamCreateWithPriority(&side, hipStreamNonBlocking, least);
  std::vector<hipEvent_t> ev(N);
  for (auto& e : ev) (void)hipEventCreateWithFlags(&e, hipEventDisableTiming);
  hipEvent_t done;
  (void)hipEventCreateWithFlags(&done, hipEventDisableTiming);
  spin<<<1, 64, 0, side>>>(100);
  (void)hipEventRecord(done, side);
  (void)hipDeviceSynchronize();
  hipEvent_t late;
  (void)hipEventCreateWithFlags(&late, hipEventDisableTiming);
  for (int mode = 0; mode < 9; ++mode) {
    double best = 1e30;
    for (int rep = 0; rep < 5; ++rep) {
      (void)hipDeviceSynchronize();
      if (mode >= 7) {      // an event that is NOT complete when the waits are enqueued (behind a 2-ms kernel on the side stream)
        spin<<<1, 64, 0, side>>>(2000 * 100);
        (void)hipEventRecord(late, side);
      }
      const double t0 = now();
      for (int i = 0; i < N; ++i) {
        if (mode == 7) (void)hipStreamWaitEvent(0, late, 0);
        if (mode == 8 && (i == 0 || i % 3 == 2)) (void)hipStreamWaitEvent(0, late, 0);
        spin<<<1, 64, 0, 0>>>(10 * 100);
        if (mode == 5) (void)hipStreamWaitEvent(0, done, 0);          // the main stream waits for an event that completed long ago
        if (mode == 6 && i % 3 == 2) (void)hipStreamWaitEvent(0, done, 0);
        const bool edge = mode == 1 || mode == 2 || ((mode == 3 || mode == 4) && i % 3 == 2);
        if (edge) {
          (void)hipEventRecord(ev[i], 0);
          if (mode == 2 || mode == 4) {
            (void)hipStreamWaitEvent(side, ev[i], 0);
            spin<<<1, 64, 0, side>>>(10 * 100);
          }
        }
      }
      (void)hipStreamSynchronize(0);
      const double t = (now() - t0) * 1e6;
      (void)hipDeviceSynchronize();
      if (t < best) best = t;
    }
    const char* what[] = {"chain alone", "+ an event record after every kernel", "+ a side stream waiting on each event, one kernel behind it",
                          "+ an event record after every 3rd kernel", "+ a side stream waiting on every 3rd",
                          "+ a wait for a long-completed event before every kernel", "+ such a wait before every 3rd kernel",
                          "+ a wait, ENQUEUED EARLY, for a 2-ms-late event before every kernel (minus 2000)", "+ such a wait before every 3rd (minus 2000)"};
    if (mode >= 7) best -= 2000.0;
    printf("%-86s main stream done after %.0f us = %.2f us per kernel\n", what[mode], best, best / N);
  }
  return 0;
}
