// Probe 2 (GPU box): soffset of `buffer_load_dwordx4 v, srd, s_off offen lds`: is it added to the
// address, and does the range check still zero lanes whose voffset is out of range?
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef int i32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const unsigned* a, unsigned* out, int nbytes, int soff) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  unsigned* s = (unsigned*)smem;
  for (int i = threadIdx.x; i < 1024; i += 64) s[i] = 0xdeadbeefu;
  __syncthreads();
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
  i32x4 srd;
  srd[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)a);
  srd[1] = __builtin_amdgcn_readfirstlane((int)((size_t)a >> 32)) & 0xffff;
  srd[2] = nbytes;
  srd[3] = 0x00020000;
  unsigned voff = threadIdx.x * 16;
  if (threadIdx.x & 1) voff = 0x80000000u;
  if (threadIdx.x == 62) voff = nbytes - 64;         // voff in range, voff + soff past the end
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %3 offen lds"
               : : "v"(voff), "s"(srd), "s"(lds0), "s"(soff) : "memory");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 256; i += 64) out[i] = s[i];
}
int main() {
  unsigned *a, *o, h[4096], r[256];
  for (int i = 0; i < 4096; ++i) h[i] = 0x1000 + i;
  (void)hipMalloc(&a, sizeof(h)); (void)hipMalloc(&o, sizeof(r));
  (void)hipMemcpy(a, h, sizeof(h), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 4096, 0, a, o, 2048, 128);
  (void)hipMemcpy(r, o, sizeof(r), hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; ++l) if (l < 4 || l >= 60) printf("lane %2d: %08x %08x %08x %08x\n", l, r[l*4], r[l*4+1], r[l*4+2], r[l*4+3]);
  return 0;
}
