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
  int halo_split;            // patch kernel: even halo rows first, odd rows behind them (conflict-free wrapped fragments)
  int flags;
  float alpha;
  // DISYOLO_CONV_BN_FUSED / DISYOLO_CONV_BN_BWD_FUSED: batch norm inside the launch (cluster exchange below)
  void* y_act;
  const float* gamma;
  const float* beta;
  float* mm;
  float* mv;
  float* o_scale;
  float* o_shift;
  float* o_mean;
  float* o_rstd;
  float* dgamma;
  float* dbeta;
  unsigned* csync;
  float bn_decay, bn_eps;
  double inv_count;          // 1 / (B*Ho*Wo)
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

// ---- cluster exchange: the blocks of ONE launch that share a channel tile hand each other a row of per-channel partial
// sums (batch statistics, batch-norm backward sums) and each of them sums ALL rows in the same fixed order -- a reduction
// over the M tiles without a second launch.  Protocol (MI355X_MICROARCH.md "Valid forms", row 1; cdna_hip_programming.md
// Guideline 16 R1): the payload is stored write-through (agent-scope relaxed atomic stores = `global_store ... sc1`: the
// bytes leave this XCD's L2), every storing wave drains its stores (s_waitcnt vmcnt(0)), the workgroup meets at its barrier,
// ONE lane adds 1 to the tile's arrival counter (agent scope); a consumer polls that counter with relaxed agent-scope loads
// (`sc1`: served past the L1) from one lane, meets its workgroup at a barrier, and reads the rows with `sc1` loads ONLY
// (nothing of the payload is ever read through the L1, whose lines other CUs' stores never refresh).  No fences.  The rows'
// lines cannot sit stale in an L2: nobody reads them in this launch before the counter says they are there, and a launch
// boundary has invalidated whatever an earlier launch cached.  Counters: [tile][32 words] = one 128-byte line per channel
// tile, word 0 = arrivals, word 1 = departures; the block whose departure is the last one zeroes both, so a buffer zeroed
// ONCE stays valid launch after launch (nothing here depends on a per-launch argument: the launches are replayed from a
// recorded list with frozen arguments).  Every wait is bounded: on a time-out the error word (after the last tile's line)
// gets a code and the launch runs on with whatever it has -- wrong numbers and a loud host-side error instead of a hung GPU.
// Residency: every block of a cluster must be resident before any of them can leave the wait -- the launcher admits the
// fused epilogue only when the WHOLE grid fits on the device at once (disyolo_conv2d_bn_fused_ok).
typedef __attribute__((address_space(1))) unsigned gu32;
typedef __attribute__((address_space(1))) unsigned long long gu64;
constexpr int CL_LINE = 32;                      // words per channel tile
constexpr unsigned CL_SPIN_LIMIT = 1u << 21;     // polls of ~0.25 us: half a second
static __device__ __forceinline__ gu32* cl_word(unsigned* base, int idx) {
  return (gu32*)(size_t)(base + idx);
}
// one row entry (two floats) written through to memory
static __device__ __forceinline__ void cl_store2(float* p2, float a, float b) {
  const unsigned long long v = ((unsigned long long)__builtin_bit_cast(unsigned, b) << 32) | __builtin_bit_cast(unsigned, a);
  __hip_atomic_store((gu64*)(size_t)p2, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
static __device__ __forceinline__ void cl_load2(const float* p2, float& a, float& b) {
  const unsigned long long v = __hip_atomic_load((gu64*)(size_t)p2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  a = __builtin_bit_cast(float, (unsigned)v);
  b = __builtin_bit_cast(float, (unsigned)(v >> 32));
}
// after the block's row is stored (cl_store2 by any of its waves): publish it
static __device__ __forceinline__ void cl_arrive(unsigned* csync, int tile) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // EVERY storing wave drains its write-through stores
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_fetch_add(cl_word(csync, tile * CL_LINE), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// wait until `expect` blocks have published; all threads of the block return together, then read rows with cl_load2 only
static __device__ __forceinline__ void cl_wait(unsigned* csync, int tile, unsigned expect, int err_idx) {
  if (threadIdx.x == 0) {
    unsigned spins = 0;
    while (__hip_atomic_load(cl_word(csync, tile * CL_LINE), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < expect) {
      __builtin_amdgcn_s_sleep(8);
      if (++spins > CL_SPIN_LIMIT) {
        __hip_atomic_store(cl_word(csync, err_idx), 0xC1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        break;
      }
    }
  }
  __syncthreads();
}
// after the block has read every row: the last block to leave re-arms the counters for the next launch
static __device__ __forceinline__ void cl_depart(unsigned* csync, int tile, unsigned expect) {
  if (threadIdx.x == 0) {
    const unsigned old = __hip_atomic_fetch_add(cl_word(csync, tile * CL_LINE + 1), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (old + 1 == expect) {
      __hip_atomic_store(cl_word(csync, tile * CL_LINE), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(cl_word(csync, tile * CL_LINE + 1), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}
// Sum rows [0, rows) of part[row][C][2] for the BN channels n0 .. n0+BN of this block, f64, fixed order, by all T threads:
// thread = (channel cl = tid % BN, row group g = tid / BN); group g takes rows g, g+G, ...; the groups are combined in
// order through `scratch` (T*16 bytes of LDS).  The result of channel cl is returned to the threads with g == 0.
// Every block of the cluster runs exactly this code on exactly these bytes: all of them get the same bits.
template <int T, int BN>
static __device__ __forceinline__ bool cl_sum_rows(const float* part, int rows, int C, int n0, double* scratch, double& r0, double& r1) {
  static_assert(T % BN == 0, "threads per block must be a multiple of the channel tile");
  constexpr int G = T / BN;
  const int cl = threadIdx.x % BN, g = threadIdx.x / BN;
  const int n = n0 + cl;
  double a0 = 0.0, a1 = 0.0;
  if (n < C) {
    int r = g;
    for (; r + 7 * G < rows; r += 8 * G) {      // eight independent loads in flight
      float v0[8], v1[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) cl_load2(part + ((size_t)(r + u * G) * C + n) * 2, v0[u], v1[u]);
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        a0 += (double)v0[u];
        a1 += (double)v1[u];
      }
    }
    for (; r < rows; r += G) {
      float v0, v1;
      cl_load2(part + ((size_t)r * C + n) * 2, v0, v1);
      a0 += (double)v0;
      a1 += (double)v1;
    }
  }
  scratch[(g * BN + cl) * 2 + 0] = a0;
  scratch[(g * BN + cl) * 2 + 1] = a1;
  __syncthreads();
  if (g == 0 && n < C) {
    double s0 = 0.0, s1 = 0.0;
#pragma unroll
    for (int k = 0; k < G; ++k) {
      s0 += scratch[(k * BN + cl) * 2 + 0];
      s1 += scratch[(k * BN + cl) * 2 + 1];
    }
    r0 = s0;
    r1 = s1;
    return true;
  }
  return false;
}
// batch statistics -> the layer's (scale, shift): the arithmetic of bn_finalize_kernel (bn.hip), value for value
static __device__ __forceinline__ void cl_bn_coeffs(double s, double s2, double inv_count, float gamma, float beta, float eps,
                                                    float& sc, float& sh, float& meanf, float& varf, float& rstd) {
  const double mean = s * inv_count;
  double var = s2 * inv_count - mean * mean;   // population variance (tf.nn.moments)
  if (var < 0.0) var = 0.0;
  meanf = (float)mean;
  varf = (float)var;
  rstd = 1.0f / sqrtf(varf + eps);
  sc = gamma * rstd;
  sh = beta - meanf * sc;
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
