// Shared by the convolution translation units (conv_igemm.hip, conv_flat.hip): the launch parameters, the LDS swizzles,
// the LDS-DMA primitive and small device helpers.  Everything here has internal linkage except ConvParams, which crosses
// from the dispatcher (conv_igemm.hip) to the flat-frame patch kernels (conv_flat.hip).
#pragma once
#include <utility>
#include "common.h"

namespace dyconv {
struct ConvParams {
  const bf16* x0;
  const bf16* x1;
  const bf16* w;
  const float* scale;
  const float* shift;
  const bf16* residual;
  void* y;
  float* stats;
  const bf16* bn_x;          // DISYOLO_CONV_BN_BWD_STATS (patch kernel): the target layer's pre-BN conv output ...
  const float* bn_scale;     // ... its scale / shift / batch mean / rstd ...
  const float* bn_shift;
  const float* bn_mean;
  const float* bn_rstd;
  float* bn_part;            // ... and the partial sums [tilesM][Cout][2] this launch writes
  float bn_alpha;
  int B, H, W, C0, C1, Cin;
  int Ho, Wo, Cout;
  int ks, stride, pad_t, pad_l, dmask, dshift;
  int M, K, nk;
  unsigned bytes0, bytes1, bytesw;
  int tilesM, tilesN;
  int pcls, Mc, tilesMc;     // stride-2 data gradient by output-parity classes (GEMM tiles): see conv_igemm_kernel
  int tapmask, d2s_c;        // stride-2 data gradient as a 2x2-tap conv over dy with a depth-to-space store (disyolo_dgrad_s2_quad)
  int xcd_n;                 // GEMM tiles: an XCD's run of tiles walks the pixel tiles of a few channel tiles (weights > input)
  int flags;
  float alpha;
};

// swizzle of the 16-byte chunk index inside one LDS row: 64-byte rows (BK=32) use a 4-entry
// table on (row>>2)&3, 128-byte rows (BK=64) XOR (row>>1)&7; both make the 16-lane groups of
// a ds_read_b128 fragment read hit 16 distinct 16-byte bank slots (DESIGN.md "LDS layout")
template <int BK>
static __device__ __forceinline__ int swz(int row, int chunk) {
  if (BK == 32) return chunk ^ ((0x78 >> (((row >> 2) & 3) * 2)) & 3);
  return chunk ^ ((row >> 1) & 7);
}

// 64-byte rows read at ANY row alignment (the patch kernel's tap-shifted pixel fragments): the
// 4-entry table above is conflict-free only for fragments starting at a multiple of 16 rows;
// XOR with 2*((row>>2)&1) is conflict-free for 16 consecutive rows from any start (the 8 tables
// with that property, by exhaustive search, are all of this alternating form).
static __device__ __forceinline__ int swz_any(int row, int chunk) { return chunk ^ (((row >> 2) & 1) << 1); }

// one LDS-DMA: 64 lanes x 16 B, buffer (descriptor + per-lane 32-bit byte offset) -> LDS
// (wave-uniform base in M0 + lane*16).  Lanes whose offset is past the descriptor's range get
// zeros written (hardware range check; probed on gfx950 with tools/probe_lds_dma.hip): that is
// the zero padding of the SAME conv, the ragged M/N edges and the odd taps of the stride-2
// data gradient, with no branch and no pointer select.  Issued through inline asm so hipcc
// does not see a pending LDS write and drain vmcnt(0) before every fragment read; completion
// is tracked by the counted s_waitcnt vmcnt(N) in the main loop (an LDS-DMA has no VGPR
// destination, so it is register-safe).  Nothing else in this kernel uses M0.
typedef int i32x4 __attribute__((ext_vector_type(4)));
constexpr unsigned OOB = 0x80000000u;
// address = descriptor base + voff (per lane) + soff (wave-uniform SGPR); the range check
// covers voff + soff (probed: tools/probe_lds_dma2.hip), so OOB lanes stay out of range.
template <int LDS_IMM>
static __device__ __forceinline__ void dma16(unsigned voff, i32x4 srd, unsigned soff, unsigned lds_base) {
  asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds"
               :
               : "v"(voff), "s"(srd), "s"(soff), "s"(lds_base + LDS_IMM)
               : "memory");
}
// m / d and m % d for 0 <= m < 2^24 via one float division and a +-1 fix-up
static __device__ __forceinline__ void divmod_small(int m, int d, int& q, int& r) {
  q = (int)(__fdividef((float)m, (float)d));
  r = m - q * d;
  if (r < 0) {
    --q;
    r += d;
  } else if (r >= d) {
    ++q;
    r -= d;
  }
}
static __device__ __forceinline__ i32x4 make_srd(const void* base, unsigned bytes) {
  i32x4 r;
  r[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)base);
  r[1] = __builtin_amdgcn_readfirstlane((int)((size_t)base >> 32)) & 0xffff;
  r[2] = __builtin_amdgcn_readfirstlane((int)bytes);
  r[3] = 0x00020000;
  return r;
}
template <int N>
static __device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}


// one block's per-channel partial (sum, sum of squares) of a statistics row / of the batch-norm backward sums
static __device__ __forceinline__ void stats_out(const ConvParams& p, int row, int n, float s, float s2) {
  p.stats[((size_t)row * p.Cout + n) * 2 + 0] = s;
  p.stats[((size_t)row * p.Cout + n) * 2 + 1] = s2;
}
static __device__ __forceinline__ void bnpart_out(const ConvParams& p, int row, int n, float s, float s2) {
  p.bn_part[((size_t)row * p.Cout + n) * 2 + 0] = s;
  p.bn_part[((size_t)row * p.Cout + n) * 2 + 1] = s2;
}

// batch-norm backward sums of one stored 16-byte chunk (8 channels of one pixel): g = dy*act'(z), xhat
struct BnBwdLane {
  float sc[8], sh[8], mu[8], rs[8], s1[8], s2[8];
  __device__ __forceinline__ void init(const ConvParams& p, int n, bool ok) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      sc[k] = ok ? p.bn_scale[n + k] : 0.f;
      sh[k] = ok ? p.bn_shift[n + k] : 0.f;
      mu[k] = ok ? p.bn_mean[n + k] : 0.f;
      rs[k] = ok ? p.bn_rstd[n + k] : 0.f;
      s1[k] = s2[k] = 0.f;
    }
  }
  __device__ __forceinline__ void add(const uint4& dy, const uint4& x, float alpha) {
    float g[8], vx[8];
    unpack8(dy, g);
    unpack8(x, vx);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float z = vx[k] * sc[k] + sh[k];
      const float gg = g[k] * (z > 0.f ? 1.f : alpha);
      const float xh = (vx[k] - mu[k]) * rs[k];
      s1[k] += gg;
      s2[k] += gg * xh;
    }
  }
  // sum over the lanes that hold the same chunk column: lane ids equal modulo CPR8
  template <int CPR8>
  __device__ __forceinline__ void reduce() {
#pragma unroll
    for (int o = CPR8; o < 64; o <<= 1) {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        s1[k] += __shfl_xor(s1[k], o, 64);
        s2[k] += __shfl_xor(s2[k], o, 64);
      }
    }
  }
};

// ---- flat-frame patch kernels (conv_flat.hip) ------------------------------------------------------------------------
struct FlatGeom {
  int bm, bn;        // block tile: frame positions x output channels
  int tilesM;        // M tiles = statistics rows
  int stages;        // LDS stages the launch will use (3 where three slices fit into 160 KiB)
};
bool flat_ok(const disyolo_conv_desc* d, int id, FlatGeom* g);
int launch_flat(const ConvParams& p, int id, hipStream_t s);

}  // namespace dyconv
