// 3x3 stride-1 SAME convolution on v_mfma_f32_32x32x16_bf16 over a FLAT padded frame ("flat patch" kernels,
// tile ids 24 / 25; forward and data gradient of yolo/yolo3_net_pos.py:142 conv_bn's tf.nn.conv2d for the deep layers).
//
// Why a second patch-kernel family (round 4; profiles/archive/r04_halo_ablation.txt, r04_mfma_lds_probe.txt).  Compile-time
// ablations of conv_halo_kernel<8,3,2> on the 18^2 512 -> 1024 layer: MFMAs alone 11.6 us of loop, fragment reads alone
// 13.4, both 20.5-21.3 -- the two do not overlap -- while MFMAs + DMAs (13.6) and reads + DMAs (16) do.  A 16x16x32
// MFMA holds the SIMD's vector issue for 8 of its 16 cycles and needs one 1-KiB fragment read per 1.2 MFMAs with a
// 48 x 32 register tile; a 32x32x16 MFMA does twice the work for the same 8 issue cycles and the same two operand
// fragments.  The micro-benchmark of the bare instruction mix (tools/probe_mfma_lds.hip): 80 % of the MFMA-only rate
// for the 16x16x32 3x2 tile, 90 % for 32x32x16 3x1, 93 % for 32x32x16 3x2.
//
// Frame.  Output pixels are numbered in a padded frame, Q = b*(H+1)*P + (y+1)*P + (x+1), P = W+1: one pad column per
// row (right pad of this row = left pad of the next) and one pad row per image (bottom pad of this image = top pad of
// the next).  The input pixel of tap (kh, kw) for output position Q is Q + (kh-1)*P + (kw-1) -- one uniform shift,
// pads included -- so a block owns ANY run of BM consecutive positions (no patch shape to divide H and W, 32-pixel
// fragments always full), its halo is the run extended by P+1 on both sides, and the LDS row of a fragment lane is
// lane + tap shift.  Pad positions are computed and dropped (11 % of the 18^2 frame, 5.6 % at 36^2, 2.8 % at 72^2).
//
// Block = BM frame positions x BN = 64 output channels, 8 waves = MW x NWV wave tiles of (MI x 32) x (NI x 32) times
// KG = 2 K groups: a K slice is 32 input channels = two k16 steps, group g multiplies step g of every tap of every
// slice from the SAME staged slice (the groups' f32 sums are exchanged through LDS and added in a fixed order; each
// group finishes half of the fragments).  Per slice one halo (BM + 2P + 2 rows of 64 B) and 9 x BN x 32 weights arrive
// by LDS-DMA (out-of-frame / pad rows are out-of-range lanes = hardware zero fill) into ST stages; with ST = 3 a slice
// has a whole slice time to land (the 2-stage patch kernel waits for its last DMA at every slice).  16-byte chunks
// are XOR-swizzled with (row >> 2) & 3 on the DMA source side: a 32-row fragment read is conflict-free at ANY row
// alignment for the lane groups ds_read_b128 is served in.  Fragment addresses are precomputed per (stage, tap); the
// fragment index and the weight tap are instruction immediates.
//
//   id 24 "F192": BM 192 = 2 x (3 x 32), BN 64 = 2 x (1 x 32), KG 2 -- 256 blocks on the 18^2 1024-channel layers
//   id 25 "F384": BM 384 = 4 x (3 x 32), BN 64 = 1 x (2 x 32), KG 2 -- 232 blocks on the 36^2 512-channel layers
#include <utility>
#include "common.h"
#include "conv_common.h"

using namespace dyconv;

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int N>
__device__ __forceinline__ void wait_vmcnt_rt(int n) {
  // s_waitcnt takes an immediate: the per-wave piece count (wave-uniform) selects one
  if constexpr (N > 0) {
    if (n >= N) {
      wait_vmcnt<N>();
      return;
    }
    wait_vmcnt_rt<N - 1>(n);
  } else {
    wait_vmcnt<0>();
  }
}

__device__ __forceinline__ void dma16_rt(unsigned voff, i32x4 srd, unsigned soff, unsigned lds_addr) {
  asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds"
               :
               : "v"(voff), "s"(srd), "s"(soff), "s"(lds_addr)
               : "memory");
}

#ifdef FLAT_ABL_NOMFMA
__device__ __forceinline__ void abl_keep32(const bf16x8& a, const bf16x8& b, f32x16& c) { asm volatile("" ::"v"(a), "v"(b), "v"(c)); }
#endif
constexpr int FLAT_DMAX = 10;   // DMA pieces (1 KiB) per wave and slice at most
constexpr int FLAT_DPT = 2;     // issued per tap during the first taps of the slice that is being multiplied

// frame position -> (valid, NHWC pixel index)
__device__ __forceinline__ bool frame_decode(int Q, int FR, int P, int B, int H, int W, int& m_out) {
  if (Q < 0) return false;
  int b, q, row, col;
  divmod_small(Q, FR, b, q);
  divmod_small(q, P, row, col);
  m_out = (b * H + row - 1) * W + col - 1;
  return b < B && row >= 1 && col >= 1;
}

template <int MW, int MI, int NWV, int NI, int KG, int ST>
__global__ __launch_bounds__(MW* NWV* KG * 64) void conv_flat_kernel(ConvParams p, int P, int FR, int HP) {
#ifdef HALO_PROBE   // tools/probe_halo.py: s_memtime at entry / first DMAs issued / main loop done / end, s_memrealtime at entry
  long long hp_t[5];
  hp_t[0] = (long long)__builtin_amdgcn_s_memtime();
  hp_t[4] = (long long)__builtin_amdgcn_s_memrealtime();
#endif
  constexpr int NWAVES = MW * NWV * KG;
  constexpr int BM = MW * MI * 32, BN = NWV * NI * 32;
  constexpr int WP = 9 * BN / 16;          // weight pieces per stage
  constexpr int SPW = 2 / KG;              // k16 steps per wave and tap
  constexpr int U = 9 * SPW;               // multiply units per slice and wave
  constexpr int NF = MI * NI;              // 32x32 fragments per wave
  static_assert(KG == 1 || KG == 2, "a 32-channel slice has two k16 steps");
  static_assert(BN <= 64, "the weight tap rides in the 16-bit DS offset");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kg = wave / (MW * NWV);                 // K group
  const int wt = wave - kg * (MW * NWV);            // wave tile
  const int mw = wt / NWV, nwv = wt - mw * NWV;
  const int NP = HP + WP;
  const unsigned STB = (unsigned)NP * 1024u;        // bytes per stage: halo pieces, then weight pieces

  int tile;
  {  // XCD-aware order (as the other conv kernels): consecutive tiles of one XCD's run share operands in its L2
    const int nblk = gridDim.x, bid = blockIdx.x;
    const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, loc = bid >> 3;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
  }
  int mt, nt;
  if (p.xcd_n) {
    nt = tile / p.tilesM;
    mt = tile - nt * p.tilesM;
  } else {
    mt = tile / p.tilesN;
    nt = tile - mt * p.tilesN;
  }
  const int n0 = nt * BN;
  const int Q0 = mt * BM;
  const int R = BM + 2 * P + 2;                     // halo rows in use

  const i32x4 srdx = make_srd(p.x0, p.bytes0);
  const i32x4 srdw = make_srd(p.w, p.bytesw);

  // ---- DMA pieces of this wave: piece k = i * NWAVES + wave; k < HP: halo rows 16k .. 16k+15, else weight rows.
  // The weight pieces' offsets come first (shifts only) and slice 0's weights are requested before the halo
  // positions are decoded: the decode runs under the first fetch.
  const int npw = __builtin_amdgcn_readfirstlane(NP > wave ? (NP - wave + NWAVES - 1) / NWAVES : 0);
  unsigned d_off[FLAT_DMAX];
#pragma unroll
  for (int i = 0; i < FLAT_DMAX; ++i) {
    const int k = i * NWAVES + wave;
    unsigned off = OOB;
    if (k >= HP && k < NP) {
      const int rb = (k - HP) * 16 + (lane >> 2), cp = lane & 3;
      const int c = cp ^ ((rb >> 2) & 3);
      const int tap = rb / BN, nl = rb - tap * BN;
      if (n0 + nl < p.Cout) off = ((unsigned)(n0 + nl) * (unsigned)p.K + (unsigned)(tap * p.Cin)) * 2u + c * 16;
    }
    d_off[i] = off;
  }
  auto issue_one = [&]<int I, int KIND = 0>(std::integral_constant<int, I>, unsigned cs2, unsigned stage_base,
                                            std::integral_constant<int, KIND> = {}) {
    if constexpr (I < FLAT_DMAX) {
      if (I < npw) {
        const int k = I * NWAVES + wave;
        if (KIND == 0 || (KIND == 1) == (k >= HP)) dma16_rt(d_off[I], k < HP ? srdx : srdw, cs2, stage_base + (unsigned)k * 1024u);
      }
    }
  };
  auto issue_all = [&]<int KIND>(std::integral_constant<int, KIND> kind, int c, int stage) {
    [&]<int... I>(std::integer_sequence<int, I...>) {
      (issue_one(std::integral_constant<int, I>{}, (unsigned)c * 64u, lds0 + (unsigned)stage * STB, kind), ...);
    }(std::make_integer_sequence<int, FLAT_DMAX>{});
  };
  const int nch = p.Cin >> 5;
  issue_all(std::integral_constant<int, 1>{}, 0, 0);       // slice 0: weights
  {
    // halo pieces: frame position of this lane's row in piece slot 0, then +16*NWAVES rows per slot.  Decoded once
    // (two float divisions), advanced by carries.  Positions are shifted by one image so that they are >= 0.
    const int step = 16 * NWAVES;
    int sa, sr;                                             // step = sa * P + sr
    divmod_small(step, P, sa, sr);
    sa = __builtin_amdgcn_readfirstlane(sa);
    sr = __builtin_amdgcn_readfirstlane(sr);
    const int rr0 = wave * 16 + (lane >> 2), cp = lane & 3;
    int b, q, row, col;
    divmod_small(Q0 - (P + 1) + rr0 + FR, FR, b, q);
    divmod_small(q, P, row, col);
    b -= 1;
#pragma unroll
    for (int i = 0; i < FLAT_DMAX; ++i) {
      const int k = i * NWAVES + wave;
      const int rr = rr0 + i * step;
      if (k < HP) {
        const int c = cp ^ ((rr >> 2) & 3);
        const bool ok = rr < R && b >= 0 && b < p.B && row >= 1 && col >= 1;
        if (ok) d_off[i] = (unsigned)((b * p.H + row - 1) * p.W + col - 1) * (unsigned)p.C0 * 2u + c * 16;
      }
      col += sr;
      row += sa;
      if (col >= P) { col -= P; ++row; }
      while (row > p.H) { row -= p.H + 1; ++b; }
    }
  }
  issue_all(std::integral_constant<int, 2>{}, 0, 0);       // slice 0: halo

  // ---- fragment read addresses
  const int rl = lane & 31, hl = lane >> 5;
  unsigned xa[ST][9];       // pixel fragment 0 of this wave at (stage, tap), k16 step kg (KG = 2) or 0 (KG = 1)
  {
    const int rbase = mw * MI * 32 + rl;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int row = rbase + (t / 3) * P + (t % 3);
      const int s0 = KG == 2 ? kg : 0;
      const unsigned a = (unsigned)row * 64u + (unsigned)(((2 * s0 + hl) ^ (row >> 2)) & 3) * 16u;
#pragma unroll
      for (int st = 0; st < ST; ++st) xa[st][t] = a + (unsigned)st * STB;
    }
  }
  unsigned wa[ST][SPW];     // weight fragment 0 at tap 0
  {
    const int rlw = nwv * NI * 32 + rl;
#pragma unroll
    for (int s = 0; s < SPW; ++s) {
      const int s0 = KG == 2 ? kg : s;
      const unsigned a = (unsigned)HP * 1024u + (unsigned)rlw * 64u + (unsigned)(((2 * s0 + hl) ^ (rl >> 2)) & 3) * 16u;
#pragma unroll
      for (int st = 0; st < ST; ++st) wa[st][s] = a + (unsigned)st * STB;
    }
  }

  f32x16 acc[MI][NI];
#pragma unroll
  for (int m = 0; m < MI; ++m)
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[m][j][e] = 0.f;

  // prologue: the slices up to ST-2 (slice 0 is already on its way)
#pragma unroll
  for (int c = 1; c < ST - 1; ++c)
    if (c < nch) issue_all(std::integral_constant<int, 0>{}, c, c);

#ifdef HALO_PROBE
  hp_t[1] = (long long)__builtin_amdgcn_s_memtime();
#endif
  bf16x8 xf[2][MI], wf[2][NI];
  auto load_frags = [&]<int STG, int UNIT>(std::integral_constant<int, STG>, std::integral_constant<int, UNIT>) {
    constexpr int tap = UNIT / SPW, s = UNIT % SPW, bi = UNIT & 1;
    const unsigned xb = (KG == 1 && s == 1) ? (xa[STG][tap] ^ 32u) : xa[STG][tap];
#ifdef FLAT_ABL_NOREAD
#pragma unroll
    for (int j = 0; j < NI; ++j) asm volatile("" : "+v"(wf[bi][j]));
#pragma unroll
    for (int m = 0; m < MI; ++m) asm volatile("" : "+v"(xf[bi][m]));
    (void)xb;
#else
#pragma unroll
    for (int j = 0; j < NI; ++j) wf[bi][j] = *reinterpret_cast<const bf16x8*>(smem + wa[STG][s] + (tap * BN * 64 + j * 2048));
#pragma unroll
    for (int m = 0; m < MI; ++m) xf[bi][m] = *reinterpret_cast<const bf16x8*>(smem + xb + m * 2048);
#endif
  };
  auto slice_body = [&]<int STG>(std::integral_constant<int, STG>, int c) {
    // slice c has landed for this wave (later slices may still be in flight), then for everyone; everyone is done
    // with slice c-1, whose stage the DMAs issued below overwrite
    const int ahead = nch - 1 - c < ST - 2 ? nch - 1 - c : ST - 2;
    wait_vmcnt_rt<(ST - 2) * FLAT_DMAX>(ahead * npw);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    const int cn = c + ST - 1;
    const bool more = cn < nch;
    constexpr int SN = (STG + ST - 1) % ST;
    const unsigned cs_next = (unsigned)cn * 64u;
    const unsigned sb_next = lds0 + (unsigned)SN * STB;
    load_frags(std::integral_constant<int, STG>{}, std::integral_constant<int, 0>{});
    auto unit_body = [&]<int UNIT>(std::integral_constant<int, UNIT>) {
      constexpr int bi = UNIT & 1;
      if constexpr (UNIT + 1 < U) load_frags(std::integral_constant<int, STG>{}, std::integral_constant<int, UNIT + 1>{});
      __builtin_amdgcn_sched_barrier(0);   // keep the reads of the next unit in front of this unit's MFMAs
#ifndef FLAT_ABL_NODMA
      if constexpr (UNIT * FLAT_DPT < FLAT_DMAX) {
        if (more) {
          [&]<int... D>(std::integer_sequence<int, D...>) {
            (issue_one(std::integral_constant<int, UNIT * FLAT_DPT + D>{}, cs_next, sb_next), ...);
          }(std::make_integer_sequence<int, FLAT_DPT>{});
        }
      }
#endif
#pragma unroll
      for (int m = 0; m < MI; ++m)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#ifdef FLAT_ABL_NOMFMA
          abl_keep32(xf[bi][m], wf[bi][j], acc[m][j]);
#else
          acc[m][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[bi][j], xf[bi][m], acc[m][j], 0, 0, 0);
#endif
    };
    [&]<int... T>(std::integer_sequence<int, T...>) {
      (unit_body(std::integral_constant<int, T>{}), ...);
    }(std::make_integer_sequence<int, U>{});
  };
  for (int c = 0; c < nch;) {
    slice_body(std::integral_constant<int, 0>{}, c++);
    if (c >= nch) break;
    slice_body(std::integral_constant<int, 1>{}, c++);
    if constexpr (ST == 3) {
      if (c >= nch) break;
      slice_body(std::integral_constant<int, 2>{}, c++);
    }
  }
  __syncthreads();      // every wave is done with the stages: the epilogue's scratch overlays them
#ifdef HALO_PROBE
  hp_t[2] = (long long)__builtin_amdgcn_s_memtime();
  struct ProbeEnd {
    const ConvParams& p; long long* t; int wave, lane, nw;
    __device__ ~ProbeEnd() {
      if (p.flags & 0x200000) {
        t[3] = (long long)__builtin_amdgcn_s_memtime();
        if (lane == 0) {
          long long* o = reinterpret_cast<long long*>(p.stats) + ((size_t)blockIdx.x * nw + wave) * 8;
          for (int k = 0; k < 5; ++k) o[k] = t[k];
        }
      }
    }
  } probe_end{p, hp_t, wave, lane, NWAVES};
#endif

  // ---- epilogue.  acc[m][j][e]: frame position Q0 + (mw*MI + m)*32 + (lane & 31), channel
  // n0 + (nwv*NI + j)*32 + (e >> 2)*8 + (lane >> 5)*4 + (e & 3).  Fragment f = m*NI + j is finished by K group f % KG.
  constexpr unsigned XCH_BYTES = (KG > 1 ? (unsigned)(MW * NWV) * NF * 4096u : 0u);   // exchange slots [wave tile][f][4][64 lanes] x 16 B
  constexpr int ROWP = 64 + 16;                                                       // staging row: 32 channels + pad
  constexpr unsigned STG_BYTES = 32u * ROWP;                                          // one fragment per wave at a time
  char* xch = smem;
  char* stg = smem + XCH_BYTES + (unsigned)wave * STG_BYTES;
  float* red = reinterpret_cast<float*>(smem + XCH_BYTES + (unsigned)NWAVES * STG_BYTES);   // [NWAVES][NI*32][2]
  float* scsh = red + NWAVES * NI * 32 * 2;                                                 // [2][BN]: scale, shift of this block's channels
  // everything the finishing passes wait for is requested NOW, under the exchange of the K groups' sums: this block's
  // scale / shift (-> LDS), the frame positions' output pixels, the residual values of the fragments this wave finishes
  if (tid < BN) {
    scsh[tid] = p.scale ? p.scale[n0 + tid] : 1.f;
    scsh[BN + tid] = p.shift ? p.shift[n0 + tid] : 0.f;
  }
  int m_of[MI];
#pragma unroll
  for (int m = 0; m < MI; ++m) {
    int mo;
    const bool ok = frame_decode(Q0 + (mw * MI + m) * 32 + rl, FR, P, p.B, p.H, p.W, mo);
    m_of[m] = ok ? mo : -1;
  }
  uint2 resv[MI][NI][4];
  if (p.residual) {
#pragma unroll
    for (int m = 0; m < MI; ++m)
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        if ((m * NI + j) % KG != kg) continue;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          resv[m][j][q] = uint2{0u, 0u};
          if (m_of[m] >= 0)
            resv[m][j][q] = *reinterpret_cast<const uint2*>(p.residual + (size_t)m_of[m] * p.Cout + n0 + (nwv * NI + j) * 32 + q * 8 + hl * 4);
        }
      }
  }
  if constexpr (KG > 1) {
#pragma unroll
    for (int m = 0; m < MI; ++m)
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        const int f = m * NI + j;
        if (f % KG != kg) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            f32x4 v = {acc[m][j][4 * q], acc[m][j][4 * q + 1], acc[m][j][4 * q + 2], acc[m][j][4 * q + 3]};
            *reinterpret_cast<f32x4*>(xch + (((unsigned)(wt * NF + f) * 4 + q) * 64 + lane) * 16) = v;
          }
        }
      }
    __syncthreads();
#pragma unroll
    for (int m = 0; m < MI; ++m)
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        const int f = m * NI + j;
        if (f % KG == kg) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(xch + (((unsigned)(wt * NF + f) * 4 + q) * 64 + lane) * 16);
            // group 0's sum + group 1's sum, whichever of the two finishes the fragment (a + b == b + a bit for bit)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[m][j][4 * q + r] += v[r];
          }
        }
      }
  }

  if constexpr (KG == 1) __syncthreads();   // (scale / shift staged above; with K groups the exchange's barrier covers it)

  if (p.flags & DISYOLO_CONV_STATS) {
    // per-channel sum / sum of squares of the f32 results over this block's valid positions: lanes -> waves (LDS, fixed order)
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      float s[16], s2[16];
#pragma unroll
      for (int e = 0; e < 16; ++e) s[e] = s2[e] = 0.f;
#pragma unroll
      for (int m = 0; m < MI; ++m) {
        if ((m * NI + j) % KG != kg) continue;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const float v = m_of[m] >= 0 ? acc[m][j][e] : 0.f;
          s[e] += v;
          s2[e] += v * v;
        }
      }
#pragma unroll
      for (int e = 0; e < 16; ++e) {
#pragma unroll
        for (int o = 1; o < 32; o <<= 1) {
          s[e] += __shfl_xor(s[e], o, 64);
          s2[e] += __shfl_xor(s2[e], o, 64);
        }
        if (rl == 0) {
          const int ch = j * 32 + (e >> 2) * 8 + hl * 4 + (e & 3);
          red[((unsigned)wave * (NI * 32) + ch) * 2 + 0] = s[e];
          red[((unsigned)wave * (NI * 32) + ch) * 2 + 1] = s2[e];
        }
      }
    }
    __syncthreads();
    for (int nl = tid; nl < BN; nl += NWAVES * 64) {
      const int wn = nl / (NI * 32), ch = nl - wn * (NI * 32);
      float s = 0.f, s2 = 0.f;
#pragma unroll
      for (int g = 0; g < KG; ++g)
#pragma unroll
        for (int w_ = 0; w_ < MW; ++w_) {
          const int wv = g * (MW * NWV) + w_ * NWV + wn;
          s += red[((unsigned)wv * (NI * 32) + ch) * 2 + 0];
          s2 += red[((unsigned)wv * (NI * 32) + ch) * 2 + 1];
        }
      stats_out(p, mt, n0 + nl, s, s2);
    }
    __syncthreads();
  }

  // scale / shift / leaky / residual in the accumulator layout, bf16 through a per-wave staging tile, whole 64-byte
  // channel runs per pixel out (4 lanes x 16 B); optional batch-norm backward sums from the stored values
  const bool bnb = p.flags & DISYOLO_CONV_BN_BWD_STATS;
  bf16* yo = reinterpret_cast<bf16*>(p.y);
  BnBwdLane bl;
  bool bl_init = false;
  int bl_n = 0;
#pragma unroll
  for (int m = 0; m < MI; ++m)
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      if ((m * NI + j) % KG != kg) continue;
      const int nf = n0 + (nwv * NI + j) * 32;            // first channel of the fragment
      // rows this lane will store (two rounds of 64 chunks: 32 rows x 4 chunks)
      int ms[2];
      uint4 bx[2];
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        const int idx = it * 64 + lane;
        ms[it] = __shfl(m_of[m], idx >> 2, 64);
        bx[it] = uint4{0, 0, 0, 0};
        if (bnb && ms[it] >= 0) bx[it] = *reinterpret_cast<const uint4*>(p.bn_x + (size_t)ms[it] * p.Cout + nf + (lane & 3) * 8);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int nl = (nwv * NI + j) * 32 + q * 8 + hl * 4;       // channel within the block
        const f32x4 sc = *reinterpret_cast<const f32x4*>(scsh + nl), sh = *reinterpret_cast<const f32x4*>(scsh + BN + nl);
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          v[r] = acc[m][j][4 * q + r] * sc[r] + sh[r];
          if (p.flags & DISYOLO_CONV_LEAKY) v[r] = leaky(v[r], p.alpha);
        }
        if (p.residual) {      // (positions outside the image hold zeros and are not stored)
          const uint2 rr = resv[m][j][q];
          v[0] += __builtin_bit_cast(float, rr.x << 16);
          v[1] += __builtin_bit_cast(float, rr.x & 0xffff0000u);
          v[2] += __builtin_bit_cast(float, rr.y << 16);
          v[3] += __builtin_bit_cast(float, rr.y & 0xffff0000u);
        }
        uint2 o2;
        o2.x = pack2(v[0], v[1]);
        o2.y = pack2(v[2], v[3]);
        *reinterpret_cast<uint2*>(stg + rl * ROWP + (q * 8 + hl * 4) * 2) = o2;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      if (bnb && !bl_init) {
        bl_n = nf + (lane & 3) * 8;
        bl.init(p, bl_n, true);
        bl_init = true;
      }
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        const int idx = it * 64 + lane;
        if (ms[it] >= 0) {
          const uint4 o = *reinterpret_cast<const uint4*>(stg + (idx >> 2) * ROWP + (idx & 3) * 16);
          *reinterpret_cast<uint4*>(yo + (size_t)ms[it] * p.Cout + nf + (idx & 3) * 8) = o;
          if (bnb) bl.add(o, bx[it], p.bn_alpha);
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
  if (bnb) {
    // the fragments a wave finishes share ONE channel range (NI = 1, or NI = KG = 2: group g finishes j = g): lanes with
    // the same chunk column -> waves (LDS, fixed order) -> one row of partials per M tile
    static_assert(NI == 1 || (NI == 2 && KG == 2), "bn backward sums: a wave's fragments must share their channels");
    __syncthreads();
    bl.template reduce<4>();
    const int jown = NI == 1 ? 0 : kg;
#pragma unroll
    for (int j = 0; j < NI; ++j)
      if (lane < 4) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const bool mine = (j == jown) && bl_init;
          red[((unsigned)wave * (NI * 32) + j * 32 + lane * 8 + k) * 2 + 0] = mine ? bl.s1[k] : 0.f;
          red[((unsigned)wave * (NI * 32) + j * 32 + lane * 8 + k) * 2 + 1] = mine ? bl.s2[k] : 0.f;
        }
      }
    __syncthreads();
    for (int nl = tid; nl < BN; nl += NWAVES * 64) {
      const int wn = nl / (NI * 32), ch = nl - wn * (NI * 32);
      float s = 0.f, s2 = 0.f;
#pragma unroll
      for (int g = 0; g < KG; ++g)
#pragma unroll
        for (int w_ = 0; w_ < MW; ++w_) {
          const int wv = g * (MW * NWV) + w_ * NWV + wn;
          s += red[((unsigned)wv * (NI * 32) + ch) * 2 + 0];
          s2 += red[((unsigned)wv * (NI * 32) + ch) * 2 + 1];
        }
      bnpart_out(p, mt, n0 + nl, s, s2);
    }
  }
}

struct FlatShape {
  int bm, bn, nwaves, xch_frags;   // xch_frags = wave tiles x fragments per wave (KG > 1), for the LDS size
};
bool flat_shape(int id, FlatShape* s) {
  if (id == 24) { *s = FlatShape{192, 64, 8, 4 * 3}; return true; }
  if (id == 25) { *s = FlatShape{384, 64, 8, 4 * 6}; return true; }
  return false;
}
int flat_halo_pieces(int bm, int W) { return (bm + 2 * (W + 1) + 2 + 15) / 16; }
size_t flat_epilogue_bytes(const FlatShape& s) {
  return (size_t)s.xch_frags * 4096 + (size_t)s.nwaves * 32 * 80 + (size_t)s.nwaves * 64 * 2 * sizeof(float) + 2 * 64 * sizeof(float);
}
constexpr size_t LDS_MAX = 160 * 1024;
int flat_stages_env() {      // DISYOLO_FLAT_STAGES=2: two stages everywhere (A/B of the pipeline depth)
  static const int v = [] { const char* e = getenv("DISYOLO_FLAT_STAGES"); return e ? atoi(e) : 0; }();
  return v;
}

template <int MW, int MI, int NWV, int NI, int KG, int ST>
int launch_cfg(const ConvParams& q, int P, int FR, int HP, size_t lds, hipStream_t s) {
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_flat_kernel<MW, MI, NWV, NI, KG, ST>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_MAX);
    attr_set = true;
  }
  hipLaunchKernelGGL((conv_flat_kernel<MW, MI, NWV, NI, KG, ST>), dim3(q.tilesM * q.tilesN), dim3(MW * NWV * KG * 64), lds, s, q, P,
                     FR, HP);
  DY_CHECK_LAUNCH();
  return DISYOLO_OK;
}

}  // namespace

namespace dyconv {

bool flat_ok(const disyolo_conv_desc* d, int id, FlatGeom* g) {
  FlatShape fs;
  if (!flat_shape(id, &fs)) return false;
  if (d->ksize != 3 || d->stride != 1 || d->in_div != 1 || d->C1 != 0 || d->C0 % 32 != 0 || d->C0 < 64) return false;
  if (d->Ho != d->H || d->Wo != d->W || d->pad_t != 1 || d->pad_l != 1) return false;
  if ((d->flags & DISYOLO_CONV_OUT_F32) || d->Cout % fs.bn) return false;
  const int64_t frame = (int64_t)d->B * (d->H + 1) * (d->W + 1);
  if (frame + fs.bm >= (1 << 24)) return false;                         // (float divmod of frame positions)
  const int np = flat_halo_pieces(fs.bm, d->W) + 9 * fs.bn / 16;
  if ((np + fs.nwaves - 1) / fs.nwaves > FLAT_DMAX) return false;        // DMA slots per wave
  if ((size_t)2 * np * 1024 > LDS_MAX) return false;
  const bool three = (size_t)3 * np * 1024 <= LDS_MAX && d->C0 >= 96 && flat_stages_env() != 2;
  if (g) *g = FlatGeom{fs.bm, fs.bn, (int)((frame + fs.bm - 1) / fs.bm), three ? 3 : 2};
  return true;
}

int launch_flat(const ConvParams& p, int id, hipStream_t s) {
  FlatShape fs;
  if (!flat_shape(id, &fs)) {
    disyolo_set_error("conv: unknown flat tile id %d", id);
    return DISYOLO_E_ARG;
  }
  ConvParams q = p;
  const int P = p.W + 1, FR = (p.H + 1) * P;
  const int64_t frame = (int64_t)p.B * FR;
  q.tilesM = (int)((frame + fs.bm - 1) / fs.bm);
  q.tilesN = p.Cout / fs.bn;
  {
    static const bool on = [] { const char* e = getenv("DISYOLO_XCD_N"); return !(e && e[0] == '0'); }();
    q.xcd_n = (on && q.tilesN >= 8 && (int64_t)p.bytesw > (int64_t)p.bytes0) ? 1 : 0;
  }
  const int HP = flat_halo_pieces(fs.bm, p.W);
  const size_t stage = (size_t)(HP + 9 * fs.bn / 16) * 1024;
  const size_t epi = flat_epilogue_bytes(fs);
  const bool three = 3 * stage <= LDS_MAX && p.Cin >= 96 && flat_stages_env() != 2;
  size_t lds = (three ? 3 : 2) * stage;
  if (lds < epi) lds = epi;
  if (id == 24)
    return three ? launch_cfg<2, 3, 2, 1, 2, 3>(q, P, FR, HP, lds, s) : launch_cfg<2, 3, 2, 1, 2, 2>(q, P, FR, HP, lds, s);
  return three ? launch_cfg<4, 3, 1, 2, 2, 3>(q, P, FR, HP, lds, s) : launch_cfg<4, 3, 1, 2, 2, 2>(q, P, FR, HP, lds, s);
}

}  // namespace dyconv
