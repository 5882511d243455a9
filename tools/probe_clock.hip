// Calibration of s_memtime ticks on gfx950: ticks per dependent / independent VALU op, per MFMA, per LDS read, against
// s_memrealtime (100 MHz).  hipcc -O3 --offload-arch=gfx950 tools/probe_clock.hip -o tools/bin/probe_clock
#include <hip/hip_runtime.h>
#include <cstdio>
#include <algorithm>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__global__ void calib(long long* out, float* sink, int waves_active) {
  __shared__ float lds[4096];
  const int wave = threadIdx.x >> 6;
  lds[threadIdx.x] = threadIdx.x;
  __syncthreads();
  float a = threadIdx.x, b = 1.0001f;
  f32x4 acc = {0, 0, 0, 0};
  bf16x8 x, w;
  for (int i = 0; i < 8; ++i) { x[i] = (__bf16)(float)(threadIdx.x + i); w[i] = (__bf16)0.5f; }
  long long r[14];
  long long rt0 = __builtin_amdgcn_s_memrealtime();
  __syncthreads();
  r[0] = __builtin_amdgcn_s_memtime();
  // 1: dependent v_fma chain
#pragma unroll 1
  for (int i = 0; i < 64; ++i) {
#pragma unroll
    for (int j = 0; j < 64; ++j) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a) : "v"(b));
  }
  __syncthreads();
  r[1] = __builtin_amdgcn_s_memtime();
  // 2: 4 independent chains
  float c0 = a, c1 = a + 1, c2 = a + 2, c3 = a + 3;
#pragma unroll 1
  for (int i = 0; i < 64; ++i) {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(c0) : "v"(b));
      asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(c1) : "v"(b));
      asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(c2) : "v"(b));
      asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(c3) : "v"(b));
    }
  }
  __syncthreads();
  r[2] = __builtin_amdgcn_s_memtime();
  // 3: dependent MFMA chain
#pragma unroll 1
  for (int i = 0; i < 64; ++i) {
#pragma unroll
    for (int j = 0; j < 16; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, x, acc, 0, 0, 0);
  }
  asm volatile("s_nop 7\n s_nop 7" ::"v"(acc));
  __syncthreads();
  r[3] = __builtin_amdgcn_s_memtime();
  // 4: 4 independent MFMA chains
  f32x4 a0 = acc, a1 = acc, a2 = acc, a3 = acc;
#pragma unroll 1
  for (int i = 0; i < 64; ++i) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      a0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, x, a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, x, a1, 0, 0, 0);
      a2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, x, a2, 0, 0, 0);
      a3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, x, a3, 0, 0, 0);
    }
  }
  asm volatile("s_nop 7\n s_nop 7" ::"v"(a0), "v"(a1), "v"(a2), "v"(a3));
  __syncthreads();
  r[4] = __builtin_amdgcn_s_memtime();
  // 5: independent MFMA + VALU interleaved 1:1 (1024 each)
#pragma unroll 1
  for (int i = 0; i < 64; ++i) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      a0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, x, a0, 0, 0, 0);
      asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(c0) : "v"(b));
      a1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, x, a1, 0, 0, 0);
      asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(c1) : "v"(b));
      a2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, x, a2, 0, 0, 0);
      asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(c2) : "v"(b));
      a3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, x, a3, 0, 0, 0);
      asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(c3) : "v"(b));
    }
  }
  asm volatile("s_nop 7\n s_nop 7" ::"v"(a0), "v"(a1), "v"(a2), "v"(a3));
  __syncthreads();
  r[5] = __builtin_amdgcn_s_memtime();
  // 6: independent MFMA + 3 VALU each
#pragma unroll 1
  for (int i = 0; i < 64; ++i) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      a0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, x, a0, 0, 0, 0);
      asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(c0) : "v"(b));
      asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(c1) : "v"(b));
      asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(c2) : "v"(b));
      a1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, x, a1, 0, 0, 0);
      asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(c3) : "v"(b));
      asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(c0) : "v"(b));
      asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(c1) : "v"(b));
      a2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, x, a2, 0, 0, 0);
      asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(c2) : "v"(b));
      asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(c3) : "v"(b));
      asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(c0) : "v"(b));
      a3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, x, a3, 0, 0, 0);
      asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(c1) : "v"(b));
      asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(c2) : "v"(b));
      asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(c3) : "v"(b));
    }
  }
  asm volatile("s_nop 7\n s_nop 7" ::"v"(a0), "v"(a1), "v"(a2), "v"(a3));
  __syncthreads();
  r[6] = __builtin_amdgcn_s_memtime();
  // 7: pk_fma independent x4 (1024 x 4)
  f32x2 p0 = {a, b}, p1 = {b, a}, p2 = {a, a}, p3 = {b, b}, pb = {b, b};
#pragma unroll 1
  for (int i = 0; i < 64; ++i) {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p0) : "v"(pb));
      asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p1) : "v"(pb));
      asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p2) : "v"(pb));
      asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p3) : "v"(pb));
    }
  }
  __syncthreads();
  r[7] = __builtin_amdgcn_s_memtime();
  // 8: v_perm independent x4
  unsigned q0 = threadIdx.x, q1 = q0 + 1, q2 = q0 + 2, q3 = q0 + 3, sel = 0x07060302u;
#pragma unroll 1
  for (int i = 0; i < 64; ++i) {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(q0) : "v"(q1), "s"(sel));
      asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(q1) : "v"(q2), "s"(sel));
      asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(q2) : "v"(q3), "s"(sel));
      asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(q3) : "v"(q0), "s"(sel));
    }
  }
  __syncthreads();
  r[8] = __builtin_amdgcn_s_memtime();
  // 9: ds_read_b128 x 1024 (addresses conflict-free: lane * 16)
  f32x4 l0 = {0, 0, 0, 0};
  const unsigned la = (threadIdx.x & 63) * 16;
#pragma unroll 1
  for (int i = 0; i < 64; ++i) {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      f32x4 t;
      asm volatile("ds_read_b128 %0, %1" : "=v"(t) : "v"(la + (j & 3) * 1024));
      asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
      l0 += t;
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __syncthreads();
  r[9] = __builtin_amdgcn_s_memtime();
  // 10: half of the waves (one per SIMD) stream MFMAs, the other half VALU ops: max or sum?
  if ((wave >> 2) & 1) {
#pragma unroll 1
    for (int i = 0; i < 64; ++i) {
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(c0) : "v"(b));
        asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(c1) : "v"(b));
        asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(c2) : "v"(b));
        asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(c3) : "v"(b));
      }
    }
  } else {
#pragma unroll 1
    for (int i = 0; i < 64; ++i) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        a0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, x, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, x, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, x, a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, x, a3, 0, 0, 0);
      }
    }
    asm volatile("s_nop 7\n s_nop 7" ::"v"(a0), "v"(a1), "v"(a2), "v"(a3));
  }
  __syncthreads();
  r[10] = __builtin_amdgcn_s_memtime();
  // 11: every wave: chains of 3 dependent MFMAs, each followed by 8 VALU ops that read its result (the shape of phase A)
#pragma unroll 1
  for (int i = 0; i < 64; ++i) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      f32x4 t = {0, 0, 0, 0};
      t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, x, t, 0, 0, 0);
      t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, x, t, 0, 0, 0);
      t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, x, t, 0, 0, 0);
      c0 = c0 * t[0] + b; c1 = c1 * t[1] + b; c2 = c2 * t[2] + b; c3 = c3 * t[3] + b;
      c0 = c0 * b + b; c1 = c1 * b + b; c2 = c2 * b + b; c3 = c3 * b + b;
    }
  }
  __syncthreads();
  r[11] = __builtin_amdgcn_s_memtime();
  long long rt1 = __builtin_amdgcn_s_memrealtime();
  if ((threadIdx.x & 63) == 0) {
    for (int k = 0; k < 12; ++k) out[wave * 16 + k] = r[k];
    out[wave * 16 + 12] = rt1 - rt0;
    out[wave * 16 + 13] = r[11] - r[0];
  }
  sink[threadIdx.x] = a + c0 + c1 + c2 + c3 + acc[0] + a0[0] + a1[0] + a2[0] + a3[0] + p0[0] + p1[0] + p2[0] + p3[0] + q0 + q1 + q2 + q3 + l0[0];
}

int main() {
  long long* out; float* sink;
  hipMalloc(&out, 16 * 16 * 8); hipMalloc(&sink, 1024 * 4);
  for (int waves : {1, 4, 8, 16}) {
    hipMemset(out, 0, 16 * 16 * 8);
    hipLaunchKernelGGL(calib, dim3(1), dim3(waves * 64), 0, 0, out, sink, waves);
    hipDeviceSynchronize();
    long long h[16 * 16];
    hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
    const char* nm[11] = {"dep fma x4096", "indep fma x4096", "dep mfma x1024", "indep mfma x1024", "mfma+fma 1:1 x1024", "mfma+3fma x1024",
                         "pk_fma x4096", "perm x4096", "ds_read_b128 x1024",
                         "half mfma x1024 | half fma x4096", "3 dep mfma + 8 fma x256"};
    printf("== %d waves in one block (%d per SIMD; phases between block barriers): memtime/memrealtime = %.3f ticks per 10 ns\n", waves, (waves + 3) / 4,
           (double)h[13] / (double)h[12]);
    for (int k = 0; k < 11; ++k) {
      long long d = 0;
      for (int w = 0; w < waves; ++w) d = std::max(d, h[w * 16 + k + 1] - h[w * 16 + k]);
      const int n = (k == 0 || k == 1 || k == 6 || k == 7) ? 4096 : k == 10 ? 256 : 1024;
      printf("   %-22s %8lld ticks  %.2f per op per wave, %.2f per op per SIMD\n", nm[k], d, (double)d / n, (double)d / n / ((waves + 3) / 4));
    }
  }
  return 0;
}
