// Does an LDS fragment read overlap an MFMA?  Mimics the tap body of the 3x3 patch kernels on every CU: per "tap" a wave
// issues R ds_read_b128 for the NEXT tap (double-buffered registers, conflict-free addresses) and M MFMAs on the current
// fragments; 8 waves per block (2 per SIMD), one block per CU, no barriers, no global traffic.
//   hipcc -O3 --offload-arch=gfx950 tools/probe_mfma_lds.hip -o tools/bin/probe_mfma_lds
// Variants: 16x16x32 with a 3x2 / 3x4 / 4x4 register tile (reads 5/7/8, MFMAs 6/12/16) and 32x32x16 with 3x1 / 2x2 / 3x2
// (per k32 = two k16 steps: reads 8/8/10, MFMAs 6/8/12), each also with reads only and MFMAs only.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MI, int NI, bool READS, bool MFMA>
__global__ __launch_bounds__(512) void k16(long long* out, float* sink, int iters) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 32768 / 4; i += 512) reinterpret_cast<float*>(smem)[i] = (float)(i & 7) * 0.125f;
  __syncthreads();
  f32x4 acc[MI][NI];
  for (int a = 0; a < MI; ++a) for (int b = 0; b < NI; ++b) acc[a][b] = f32x4{0, 0, 0, 0};
  bf16x8 xf[2][MI], wf[2][NI];
  for (int s = 0; s < 2; ++s) { for (int a = 0; a < MI; ++a) for (int e = 0; e < 8; ++e) xf[s][a][e] = (__bf16)0.5f;
                                for (int b = 0; b < NI; ++b) for (int e = 0; e < 8; ++e) wf[s][b][e] = (__bf16)0.25f; }
  const char* base = smem + wave * 2048 + lane * 16;       // 1 KiB contiguous per wave-read: conflict-free
  const long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int tap = 0; tap < 8; ++tap) {
      const int bi = tap & 1;
      if (READS) {
#pragma unroll
        for (int a = 0; a < MI; ++a) xf[bi ^ 1][a] = *reinterpret_cast<const bf16x8*>(base + ((tap * 5 + a) & 15) * 1024);
#pragma unroll
        for (int b = 0; b < NI; ++b) wf[bi ^ 1][b] = *reinterpret_cast<const bf16x8*>(base + 16384 + ((tap * 3 + b) & 7) * 1024);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (MFMA) {
#pragma unroll
        for (int a = 0; a < MI; ++a)
#pragma unroll
          for (int b = 0; b < NI; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[bi][b], xf[bi][a], acc[a][b], 0, 0, 0);
      } else {
#pragma unroll
        for (int a = 0; a < MI; ++a) asm volatile("" ::"v"(xf[bi][a]));
#pragma unroll
        for (int b = 0; b < NI; ++b) asm volatile("" ::"v"(wf[bi][b]));
      }
    }
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int a = 0; a < MI; ++a) for (int b = 0; b < NI; ++b) s += acc[a][b][0] + acc[a][b][3];
  for (int a = 0; a < MI; ++a) s += (float)xf[0][a][0] + (float)xf[1][a][1];
  if (s == 123.456f) sink[0] = s;
  if (lane == 0) out[blockIdx.x * 8 + wave] = t1 - t0;
}

// 32x32x16: a "tap" = one k32 = two k16 steps; per step MI + NI reads, MI x NI MFMAs
template <int MI, int NI, bool READS, bool MFMA>
__global__ __launch_bounds__(512) void k32(long long* out, float* sink, int iters) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 32768 / 4; i += 512) reinterpret_cast<float*>(smem)[i] = (float)(i & 7) * 0.125f;
  __syncthreads();
  f32x16 acc[MI][NI];
  for (int a = 0; a < MI; ++a) for (int b = 0; b < NI; ++b) for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;
  bf16x8 xf[2][MI], wf[2][NI];
  for (int s = 0; s < 2; ++s) { for (int a = 0; a < MI; ++a) for (int e = 0; e < 8; ++e) xf[s][a][e] = (__bf16)0.5f;
                                for (int b = 0; b < NI; ++b) for (int e = 0; e < 8; ++e) wf[s][b][e] = (__bf16)0.25f; }
  const char* base = smem + wave * 2048 + lane * 16;
  const long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int st = 0; st < 16; ++st) {       // 8 taps x 2 k16 steps
      const int bi = st & 1;
      if (READS) {
#pragma unroll
        for (int a = 0; a < MI; ++a) xf[bi ^ 1][a] = *reinterpret_cast<const bf16x8*>(base + ((st * 5 + a) & 15) * 1024);
#pragma unroll
        for (int b = 0; b < NI; ++b) wf[bi ^ 1][b] = *reinterpret_cast<const bf16x8*>(base + 16384 + ((st * 3 + b) & 7) * 1024);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (MFMA) {
#pragma unroll
        for (int a = 0; a < MI; ++a)
#pragma unroll
          for (int b = 0; b < NI; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[bi][b], xf[bi][a], acc[a][b], 0, 0, 0);
      } else {
#pragma unroll
        for (int a = 0; a < MI; ++a) asm volatile("" ::"v"(xf[bi][a]));
#pragma unroll
        for (int b = 0; b < NI; ++b) asm volatile("" ::"v"(wf[bi][b]));
      }
    }
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int a = 0; a < MI; ++a) for (int b = 0; b < NI; ++b) s += acc[a][b][0] + acc[a][b][15];
  for (int a = 0; a < MI; ++a) s += (float)xf[0][a][0] + (float)xf[1][a][1];
  if (s == 123.456f) sink[0] = s;
  if (lane == 0) out[blockIdx.x * 8 + wave] = t1 - t0;
}

template <typename K>
static void run(const char* name, K kern, double mfma_cycles_per_tap, int reads_per_tap) {
  long long* out; float* sink;
  const int nb = 256, iters = 200;
  hipMalloc(&out, nb * 8 * 8); hipMalloc(&sink, 64);
  hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float ms = 0;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(kern, dim3(nb), dim3(512), 96 * 1024, 0, out, sink, iters);
    hipEventRecord(e1, 0); hipDeviceSynchronize();
    hipEventElapsedTime(&ms, e0, e1);
  }
  std::vector<long long> h(nb * 8);
  hipMemcpy(h.data(), out, nb * 8 * 8, hipMemcpyDeviceToHost);
  long long mx = 0; double mean = 0;
  for (auto v : h) { mx = std::max(mx, v); mean += (double)v / h.size(); }
  const double per_tap = mean / (iters * 8.0);
  const double mxt = (double)mx / (iters * 8.0);
  printf("%-34s mean %6.1f max %6.1f ticks per tap | kernel %7.1f us -> %.2f GHz if ticks are shader cycles | MFMA pipe alone %5.0f, LDS alone %4d\n",
         name, per_tap, mxt, ms * 1e3, (double)mx / (ms * 1e6), 2 * mfma_cycles_per_tap, 2 * 16 * reads_per_tap);
  hipFree(out); hipFree(sink);
}

#define RUN16(MI, NI) \
  run("16x16x32 " #MI "x" #NI " reads+mfma", k16<MI, NI, true, true>, MI * NI * 16.0, MI + NI); \
  run("16x16x32 " #MI "x" #NI " mfma only", k16<MI, NI, false, true>, MI * NI * 16.0, 0); \
  run("16x16x32 " #MI "x" #NI " reads only", k16<MI, NI, true, false>, 0, MI + NI);
#define RUN32(MI, NI) \
  run("32x32x16 " #MI "x" #NI " reads+mfma", k32<MI, NI, true, true>, 2 * MI * NI * 32.0, 2 * (MI + NI)); \
  run("32x32x16 " #MI "x" #NI " mfma only", k32<MI, NI, false, true>, 2 * MI * NI * 32.0, 0); \
  run("32x32x16 " #MI "x" #NI " reads only", k32<MI, NI, true, false>, 0, 2 * (MI + NI));

int main() {
  printf("8 waves per block (2 per SIMD), one block per CU, 256 blocks; per tap per SIMD the two resident waves need 2x the per-wave figures\n");
  RUN16(3, 2) RUN16(3, 4) RUN16(4, 4)
  RUN32(3, 1) RUN32(2, 2) RUN32(3, 2)
  return 0;
}
