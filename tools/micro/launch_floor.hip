// Microbenchmark: per-kernel cost of dependent launches on one stream (host ahead), plain vs hipGraph:
// what a kernel costs beyond its work (launch, end-of-kernel cache maintenance).  Build: hipcc --offload-arch=gfx950 -O2
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <chrono>
__global__ void empty_kernel() {}
__global__ void args_kernel(const uint4* x, uint4* y, long n) { if (n == -1) y[0] = x[0]; }
__global__ __launch_bounds__(256) void read_kernel(const uint4* x, uint4* y, long n) {
  unsigned acc = 0;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) acc ^= x[i].x;
  if (acc == 0x12345) y[0] = x[1];
}
__global__ __launch_bounds__(256) void write_kernel(const uint4* x, uint4* y, long n) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) y[i] = uint4{1, 2, 3, 4};
}
__global__ __launch_bounds__(256) void scale_kernel(const uint4* x, uint4* y, long n) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const float4 v = reinterpret_cast<const float4*>(x)[i];
    reinterpret_cast<float4*>(y)[i] = make_float4(v.x * 1.5f, v.y, v.z, v.w + 1.f);
  }
}
template <class F> void run(const char* name, F launch, int n, hipStream_t s) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int i = 0; i < 50; ++i) launch();
  (void)hipStreamSynchronize(s);
  (void)hipEventRecord(e0, s);
  auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < n; ++i) launch();
  auto t1 = std::chrono::steady_clock::now();
  (void)hipEventRecord(e1, s); (void)hipStreamSynchronize(s);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  printf("%-40s %6.2f us/kernel (host %5.2f us/launch)\n", name, ms * 1e3 / n, std::chrono::duration<double, std::micro>(t1 - t0).count() / n);
}
int main() {
  hipStream_t s; (void)hipStreamCreate(&s);
  uint4 *x, *y; long n = 64 * 1024 * 1024 / 16;
  (void)hipMalloc(&x, n * 16); (void)hipMalloc(&y, n * 16); (void)hipMemset(x, 0, n * 16);
  const long KB = 1024 / 16, MB = 1024 * KB;
  run("empty <<<1,64>>>", [&] { hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, s); }, 2000, s);
  run("args, no memory <<<256,256>>>", [&] { hipLaunchKernelGGL(args_kernel, dim3(256), dim3(256), 0, s, x, y, 5L); }, 2000, s);
  for (long sz : {64 * KB, 1 * MB, 8 * MB, 32 * MB}) {
    char nm[64];
    snprintf(nm, 64, "read %ld KB <<<1024,256>>>", sz / KB);
    run(nm, [&] { hipLaunchKernelGGL(read_kernel, dim3(1024), dim3(256), 0, s, x, y, sz); }, 1000, s);
    snprintf(nm, 64, "write %ld KB <<<1024,256>>>", sz / KB);
    run(nm, [&] { hipLaunchKernelGGL(write_kernel, dim3(1024), dim3(256), 0, s, x, y, sz); }, 1000, s);
    snprintf(nm, 64, "read+write %ld KB <<<1024,256>>>", sz / KB);
    run(nm, [&] { hipLaunchKernelGGL(scale_kernel, dim3(1024), dim3(256), 0, s, x, y, sz); }, 1000, s);
  }
  run("read+write 1 MB <<<64,256>>>", [&] { hipLaunchKernelGGL(scale_kernel, dim3(64), dim3(256), 0, s, x, y, MB); }, 1000, s);
  run("read+write 1 MB <<<256,256>>>", [&] { hipLaunchKernelGGL(scale_kernel, dim3(256), dim3(256), 0, s, x, y, MB); }, 1000, s);
  run("write 1 KB <<<1,64>>>", [&] { hipLaunchKernelGGL(write_kernel, dim3(1), dim3(64), 0, s, x, y, KB); }, 1000, s);
  // same pass with the destination displaced from the 64 MB-aligned alias of the source
  for (long off : {0L, 16L, 256L, 4096L + 256L, MB + 4096L + 256L}) {
    char nm[64];
    snprintf(nm, 64, "read+write 8 MB, dst +%ld B", off * 16);
    run(nm, [&] { hipLaunchKernelGGL(scale_kernel, dim3(1024), dim3(256), 0, s, x, y + off, 8 * MB); }, 1000, s);
  }
  run("in-place 8 MB", [&] { hipLaunchKernelGGL(scale_kernel, dim3(1024), dim3(256), 0, s, x, x, 8 * MB); }, 1000, s);
  // ping-pong: kernel k reads what kernel k-1 wrote (the dependent-chain case of the training step)
  int flip = 0;
  run("ping-pong 8 MB <<<1024,256>>>", [&] { hipLaunchKernelGGL(scale_kernel, dim3(1024), dim3(256), 0, s, flip ? y : x, flip ? x : y, 8 * MB); flip ^= 1; }, 1000, s);
  return 0;
}
