// Where the fused conv1 + conv2 launch spends its cycles: the product kernel compiled with F2_PROBE (s_memtime stamps at the
// phase boundaries, summed per wave).  Build + run on the GPU box:
//   hipcc -O3 -std=c++20 --offload-arch=gfx950 -DF2_PROBE tools/probe_conv12.hip -Ldis-yolo_amd -ldisyolo_hip \
//         -Wl,-rpath,$PWD/dis-yolo_amd -o gpurun_out/probe_conv12 && gpurun_out/probe_conv12 [B] [S]
#include <vector>
#include <cstdlib>
#include <cstdio>
long long* g_f2_probe = nullptr;
#include "../dis-yolo_amd/csrc/conv_first2.hip"

int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 8, S = argc > 2 ? atoi(argv[2]) : 576;
  const size_t nimg = (size_t)B * S * S * 3, ny = (size_t)B * (S / 2) * (S / 2) * 64;
  std::vector<float> himg(nimg), hw1(27 * 32), hs(64, 1.f), hh(64, 0.f);
  for (auto& v : himg) v = (float)rand() / RAND_MAX;
  for (auto& v : hw1) v = ((float)rand() / RAND_MAX - 0.5f) * 0.5f;
  std::vector<unsigned short> hw2(64 * 288);
  for (auto& v : hw2) v = 0x3c00 + (rand() & 0xff);
  float *img, *w1, *sc, *sh;
  void *w2, *y;
  hipMalloc(&img, nimg * 4); hipMalloc(&w1, 27 * 32 * 4); hipMalloc(&sc, 256); hipMalloc(&sh, 256);
  hipMalloc(&w2, 64 * 288 * 2); hipMalloc(&y, ny * 2);
  hipMemcpy(img, himg.data(), nimg * 4, hipMemcpyHostToDevice);
  hipMemcpy(w1, hw1.data(), 27 * 32 * 4, hipMemcpyHostToDevice);
  hipMemcpy(sc, hs.data(), 256, hipMemcpyHostToDevice);
  hipMemcpy(sh, hh.data(), 256, hipMemcpyHostToDevice);
  hipMemcpy(w2, hw2.data(), 64 * 288 * 2, hipMemcpyHostToDevice);
  const int nblk = 1024;
  hipMalloc(&g_f2_probe, nblk * 4 * 8 * 8);
  hipMemset(g_f2_probe, 0, nblk * 4 * 8 * 8);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0, 0);
    int rc = disyolo_conv12_fused_fwd(img, w1, sc, sh, w2, sc, sh, y, B, S, S, 0.1f, nullptr);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    printf("rc %d  %.1f us\n", rc, ms * 1e3);
  }
  std::vector<long long> h(nblk * 4 * 8);
  hipMemcpy(h.data(), g_f2_probe, h.size() * 8, hipMemcpyDeviceToHost);
  const char* names[8] = {"fetch issue", "phase A", "barrier 1", "park", "phase B mfma", "epilogue+stores", "barrier 2", "-"};
  const int tiles = B * (S / 2 / 8) * (S / 2 / 16);
  for (int blk : {0, 1, 255, 256, 511}) {
    const int its = (tiles - blk + 511) / 512;
    for (int w = 0; w < 4; ++w) {
      printf("block %3d wave %d (%d tiles):", blk, w, its);
      long long tot = 0;
      for (int k = 0; k < 7; ++k) {
        printf("  %s %lld", names[k], h[(blk * 4 + w) * 8 + k] / its);
        tot += h[(blk * 4 + w) * 8 + k];
      }
      printf("  | total/tile %lld ticks\n", tot / its);
    }
  }
  return 0;
}
