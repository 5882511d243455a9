// Where the fused 144^2 residual-block launch spends its cycles: the product kernel compiled with B64_PROBE (s_memtime
// stamps at the phase boundaries, summed per wave).
//   hipcc -O3 -std=c++20 --offload-arch=gfx950 -DB64_PROBE tools/probe_block64.hip -Ldis-yolo_amd -ldisyolo_hip \
//         -Wl,-rpath,'$ORIGIN/../../dis-yolo_amd' -o tools/bin/probe_block64 && tools/bin/probe_block64 [B] [S]
#include <vector>
#include <cstdlib>
#include <cstdio>
long long* g_b64_probe = nullptr;
#include "../dis-yolo_amd/csrc/conv_block64.hip"

int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 8, S = argc > 2 ? atoi(argv[2]) : 144;
  const size_t n = (size_t)B * S * S * 128;
  std::vector<unsigned short> hx(n), hwa(64 * 128), hwb(128 * 576);
  for (auto& v : hx) v = 0x3c00 + (rand() & 0x1ff);
  for (auto& v : hwa) v = 0x3a00 + (rand() & 0xff);
  for (auto& v : hwb) v = 0x3900 + (rand() & 0xff);
  std::vector<float> hs(128, 1.f), hh(128, 0.f);
  void *x, *wa, *wb, *y; float *sc, *sh;
  hipMalloc(&x, n * 2); hipMalloc(&y, n * 2); hipMalloc(&wa, 64 * 128 * 2); hipMalloc(&wb, 128 * 576 * 2);
  hipMalloc(&sc, 512); hipMalloc(&sh, 512);
  hipMemcpy(x, hx.data(), n * 2, hipMemcpyHostToDevice);
  hipMemcpy(wa, hwa.data(), 64 * 128 * 2, hipMemcpyHostToDevice);
  hipMemcpy(wb, hwb.data(), 128 * 576 * 2, hipMemcpyHostToDevice);
  hipMemcpy(sc, hs.data(), 512, hipMemcpyHostToDevice);
  hipMemcpy(sh, hh.data(), 512, hipMemcpyHostToDevice);
  const int nblk = 256;
  hipMalloc(&g_b64_probe, nblk * 8 * 8 * 8);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    hipMemset(g_b64_probe, 0, nblk * 8 * 8 * 8);
    hipEventRecord(e0, 0);
    int rc = disyolo_block64_fused_fwd(x, wa, sc, sh, wb, sc, sh, y, B, S, S, 128, 0.1f, nullptr);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    printf("rc %d  %.1f us\n", rc, ms * 1e3);
  }
  std::vector<long long> h(nblk * 8 * 8);
  hipMemcpy(h.data(), g_b64_probe, h.size() * 8, hipMemcpyDeviceToHost);
  const char* names[8] = {"fetch0 issue", "phase A", "barrier 1", "park0+fetch1", "phase B", "park1+fetch2+barrier 2", "epilogue+stores", "park2+barrier 3"};
  const int tiles = B * (S / 8) * (S / 16);
  for (int blk : {0, 100, 255}) {
    const int its = (tiles - blk + 255) / 256;
    for (int w : {0, 3, 4, 7}) {
      printf("block %3d wave %d (%d patches):", blk, w, its);
      long long tot = 0;
      for (int k = 0; k < 8; ++k) {
        printf("  %s %lld", names[k], h[(blk * 8 + w) * 8 + k] / its);
        tot += h[(blk * 8 + w) * 8 + k];
      }
      printf("  | per patch %lld ticks\n", tot / its);
    }
  }
  return 0;
}
