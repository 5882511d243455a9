// Implicit-GEMM convolution for gfx950 (MI355X): NHWC bf16 activations, packed bf16
// weights [Cout][K], f32 accumulation on v_mfma_f32_16x16x32_bf16.
//
// Replaces tf.nn.conv2d(+bias_add / folded batch_normalization / leaky_relu / residual
// add / nearest-upsample+concat) of yolo/yolo3_net_pos.py:125-129,142-145,150,290-291.
//
// GEMM view: M = B*Ho*Wo output pixels, N = Cout, K = ks*ks*Cin with k = (kh,kw,ci).
// The MFMA is issued "swapped" (weights as the A operand, pixels as the B operand) so a
// lane ends up holding 4 consecutive output channels of one pixel: 8-byte bf16 stores,
// vector scale/shift/residual loads.
//
// Block tile BM x BN x 32, WM x WN waves, LDS double buffer (one barrier per K-step),
// register-staged global->LDS copies issued one K-step ahead.  LDS rows are 64 B (32
// bf16); the 16-byte chunk index is XOR-swizzled so ds_read_b128 fragment reads are
// bank-conflict free (see DESIGN.md "LDS layout").
#include <utility>
#include "common.h"
#include "conv_common.h"
#include "runtime.h"

using namespace dyconv;

namespace {

// KS = filter size (1 or 3) is a compile-time parameter: the 1x1 instance drops the tap cursor
// and the per-tap offset refresh altogether (and shows up as its own row in a profile).
//
// KG > 1 = intra-block split-K: the block has KG groups of WM*WN waves; group g multiplies the
// K slices g, g+KG, ... of the SAME output tile from its own LDS ring, and the groups' f32
// accumulators are summed through LDS (fixed order) before group 0 runs the epilogue.  It
// doubles the waves per CU for the layers whose M*N gives barely one block per CU (18^2, 36^2),
// where one wave per SIMD cannot overlap its DMA issue with its MFMAs.
// EPI = 1: the instance that carries the fused batch-norm BACKWARD epilogue (DISYOLO_CONV_BN_BWD_FUSED).  Its own instance
// because that epilogue holds ~90 registers live (four coefficient vectors, two sum vectors and the target's conv output per
// lane): inside the plain instances it cost the 1x1 tiles 40-60 registers and a wave per SIMD (64x128: 108 -> 152) and made
// the three-blocks-per-CU 128x128 tiles spill.  Only the tiles the data-gradient convs of the small maps use have one.
template <int BM, int BN, int WM, int WN, int BK, int ST, int KS, int KG = 1, int EPI = 0>
__global__ __launch_bounds__(WM* WN * 64 * KG, (BM == 128 && BN == 128) ? 3 : 1) void conv_igemm_kernel(ConvParams p) {
#ifdef HALO_PROBE
  // tools/probe_halo.py (probe build only): with flag 0x200000 the stats pointer receives, per wave, s_memtime at [0] entry,
  // [1] pipeline primed, [2] main loop done, [3] end of the epilogue, s_memrealtime at entry in [4]
  long long gp_t[5];
  gp_t[0] = (long long)__builtin_amdgcn_s_memtime();
  gp_t[4] = (long long)__builtin_amdgcn_s_memrealtime();
#endif
  constexpr int NW = WM * WN;         // waves per K group
  constexpr int T = NW * 64;          // threads per K group
  constexpr int WTM = BM / WM, WTN = BN / WN;
  constexpr int MI = WTM / 16, NI = WTN / 16;
  constexpr int ROWB = BK * 2;        // bytes per LDS row
  constexpr int CPR = ROWB / 16;      // 16-byte chunks per row
  constexpr int SLAB = 1024 * NW;     // bytes one round of DMAs (one per wave) covers
  constexpr int A_BYTES = (BM * ROWB + SLAB - 1) / SLAB * SLAB;
  constexpr int B_BYTES = (BN * ROWB + SLAB - 1) / SLAB * SLAB;
  constexpr int AI = A_BYTES / SLAB, BI = B_BYTES / SLAB;   // DMAs per wave per tile
  constexpr int LPT = AI + BI;
  constexpr int STB = A_BYTES + B_BYTES;
  constexpr int PRE = ST - 1;
  static_assert(WTM % 16 == 0 && WTN % 16 == 0, "wave tile must be a multiple of 16");
  static_assert(LPT * (PRE > 0 ? PRE - 1 : 0) < 64, "vmcnt immediate is 6 bits");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
#ifdef DY_PROBE
  if (p.flags & 0x100000) return;   // timing probe: launch + dispatch of this grid only
#endif

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave_all = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kg = KG == 1 ? 0 : wave_all / NW;     // K group of this wave
  const int wave = KG == 1 ? wave_all : wave_all % NW;
  const int wm = wave / WN, wn = wave % WN;
  int nkg = p.nk / KG;                            // K slices per group (the launcher guarantees divisibility)

  // XCD-aware tile order: blocks b and b+8 share an XCD (and its L2); give each XCD a
  // contiguous run of tiles, n-tile fastest, so the blocks that re-read one pixel panel
  // (and neighbouring halo rows) hit the same L2.  Bijective for any grid size.
  // Every XCD then reads 1/8 of the pixels and ALL the weights: right while the input is the larger operand; where the
  // weights are (the 18^2 layers: 9.4 MB of weights, 2.6 MB of input at B = 8) the run goes m-tile fastest instead --
  // 1/8 of the weights and all the pixels per XCD (xcd_n, set by the launcher)
  int tile;
  {
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
  // Stride-2 data gradient (in_div = 2: dx[y, x] takes tap (kh, kw) only where y + kh - pad and x + kw - pad are even):
  // with pixels in their natural order every tile multiplies all 9 taps and 3/4 of the products are structural zeros.
  // pcls: the pixel tiles are formed per output-parity class (y & 1, x & 1) instead -- 4 x tilesMc tiles over the Mc =
  // B * Ho/2 * Wo/2 pixels of a class -- so that a tap is valid or void for a WHOLE tile, and the void ones (5, 6, 6 or 8
  // of 9) are skipped: no DMA, no MFMA.  Pixel index of a class -> (b, yy, xx) -> (2 yy + py, 2 xx + px).
  const bool pcl = KS == 3 && KG == 1 && p.pcls != 0;
  int cls_py = 0, cls_px = 0;
  if (pcl) {
    const int cls = mt / p.tilesMc;
    mt -= cls * p.tilesMc;
    cls_py = cls >> 1;
    cls_px = cls & 1;
  }
  // tapmask (bit kh*3+kw): the taps that exist at all -- the 2x2-tap conv of the quad data gradient walks 4 of the 9
  const bool tmk = KS == 3 && KG == 1 && p.tapmask != 0;
  auto tap_ok = [&](int kh_, int kw_) {
    if (tmk) return ((p.tapmask >> (kh_ * 3 + kw_)) & 1) != 0;
    return !pcl || ((((cls_py + kh_ - p.pad_t) | (cls_px + kw_ - p.pad_l)) & 1) == 0);
  };
  const int Mlim = pcl ? p.Mc : p.M;
  const int Hh = p.Ho >> 1, Wh = p.Wo >> 1;
  // (class-local pixel index) -> pixel index of the output tensor
  auto opix = [&](int m) {
    if (!pcl) return m;
    int b_, rem_, yy_, xx_;
    divmod_small(m, Hh * Wh, b_, rem_);
    divmod_small(rem_, Wh, yy_, xx_);
    return (b_ * p.Ho + 2 * yy_ + cls_py) * p.Wo + 2 * xx_ + cls_px;
  };
  if (pcl || tmk) {
    int nv = 0;
    for (int t = 0; t < 9; ++t) nv += tap_ok(t / 3, t % 3) ? 1 : 0;
    nkg = nv * (p.Cin / BK);
  }
  // element offset of (pixel m, channel n) in the output tensor.  d2s_c = C > 0: the N axis is [class (py, px)][C] and the
  // store is a depth-to-space -- pixel (b, yy, xx) of the conv's grid, class (py, px) -> pixel (2 yy + py, 2 xx + px) of a
  // tensor with C channels
  auto oaddr = [&](int m, int n) -> size_t {
    if (p.d2s_c == 0) return (size_t)opix(m) * p.Cout + n;
    const int cls = n / p.d2s_c, c = n - cls * p.d2s_c;
    int b_, rem_, yy_, xx_;
    divmod_small(m, p.Ho * p.Wo, b_, rem_);
    divmod_small(rem_, p.Wo, yy_, xx_);
    return ((size_t)(b_ * 2 * p.Ho + 2 * yy_ + (cls >> 1)) * (2 * p.Wo) + 2 * xx_ + (cls & 1)) * p.d2s_c + c;
  };
  const int m0 = mt * BM, n0 = nt * BN;

  // ---- per-lane gather state.  DMA j of this wave fills LDS bytes
  //      [(j*NW + wave)*1024 + lane*16, +16) of the tile: row = chunk / CPR, physical chunk =
  //      chunk % CPR; the lane fetches the LOGICAL chunk that the swizzle maps there.
  //      a_base = byte offset of (pixel, tap, logical chunk) at channel 0 of the current
  //      source, or OOB; it only changes when the K cursor moves to the next filter tap.
  const i32x4 srd0 = make_srd(p.x0, p.bytes0);
  const i32x4 srd1 = make_srd(p.x1 ? (const void*)p.x1 : (const void*)p.x0, p.x1 ? p.bytes1 : 0u);
  const i32x4 srdw = make_srd(p.w, p.bytesw);
  int a_iy0[AI], a_ix0[AI], a_b[AI];
  unsigned a_kb[AI], a_base[AI], a_base1[AI], a_org[AI];
  bool a_ok[AI];
#pragma unroll
  for (int j = 0; j < AI; ++j) {
    const int chunk = (j * NW + wave) * 64 + lane;
    const int row = chunk / CPR, pc = chunk % CPR;
    const int m = ((p.flags & 0x1000) ? 0 : m0) + row;   // 0x1000: timing probe, every block gathers tile 0
    a_ok[j] = (row < BM) && (m < Mlim);
    const int mm = a_ok[j] ? m : 0;
    int b, rem, yo, xo;
    if (pcl) {
      divmod_small(mm, Hh * Wh, b, rem);
      divmod_small(rem, Wh, yo, xo);
      yo = 2 * yo + cls_py;
      xo = 2 * xo + cls_px;
    } else if (p.M < (1 << 24)) {
      divmod_small(mm, p.Ho * p.Wo, b, rem);
      divmod_small(rem, p.Wo, yo, xo);
    } else {
      const int hw = p.Ho * p.Wo;
      b = mm / hw;
      rem = mm - b * hw;
      yo = rem / p.Wo;
      xo = rem - yo * p.Wo;
    }
    a_b[j] = b;
    a_iy0[j] = yo * p.stride - p.pad_t;
    a_ix0[j] = xo * p.stride - p.pad_l;
    a_kb[j] = swz<BK>(row, pc) * 16;
    a_base1[j] = OOB;
    if (p.C1 > 0) {  // fused upsample+concat (1x1 conv: a single tap, computed once)
      const int H1 = p.H >> 1, W1 = p.W >> 1;
      a_base1[j] = a_ok[j] ? (unsigned)(((a_b[j] * H1 + (a_iy0[j] >> 1)) * W1 + (a_ix0[j] >> 1)) * p.C1) * 2u + a_kb[j] : OOB;
    }
    // forward gather: tap (kh,kw) sits at a fixed byte distance from tap (0,0); keep that
    // origin (it may lie outside the image: modular arithmetic) and a validity bit per tap
    a_org[j] = (unsigned)(((a_b[j] * p.H + a_iy0[j]) * p.W + a_ix0[j]) * p.C0) * 2u + a_kb[j];
  }
  unsigned b_base[BI];
#pragma unroll
  for (int j = 0; j < BI; ++j) {
    const int chunk = (j * NW + wave) * 64 + lane;
    const int row = chunk / CPR, pc = chunk % CPR;
    const int n = ((p.flags & 0x2000) ? 0 : n0) + row;   // 0x2000: timing probe, every block reads weight tile 0
    b_base[j] = ((row < BN) && (n < p.Cout)) ? (unsigned)n * (unsigned)p.K * 2u + swz<BK>(row, pc) * 16 : OOB;
  }

  int kh = 0, kw = 0, ci0 = 0, k0 = 0;  // wave-uniform K cursor of the next tile to fetch
  bool newtap = true;
  auto skip_void_taps = [&]() {         // (pcls) step over the taps this tile's parity class never sees
    while (kh < 3 && !tap_ok(kh, kw)) {
      k0 += p.Cin;
      if (++kw == 3) {
        kw = 0;
        ++kh;
      }
    }
  };
  if (pcl || tmk) skip_void_taps();
  auto advance = [&]() {  // move the K cursor by one BK-wide slice
    k0 += BK;
    ci0 += BK;
    if (KS == 3 && ci0 >= p.Cin) {
      ci0 = 0;
      newtap = true;
      if (++kw == 3) {
        kw = 0;
        ++kh;
      }
      if (pcl || tmk) skip_void_taps();
    }
  };
  if (KG > 1)
    for (int g = 0; g < kg; ++g) advance();

  auto issue_tile = [&](int stage) {
    const unsigned sbase = lds0 + (kg * ST + stage) * STB + wave * 1024;
    if (newtap) {  // new filter tap (wave-uniform): refresh the per-lane pixel offsets
      newtap = false;
      if (p.dshift == 0) {
        const unsigned tapoff = (unsigned)((kh * p.W + kw) * p.C0) * 2u;
#pragma unroll
        for (int j = 0; j < AI; ++j) {
          const bool ok = a_ok[j] && ((unsigned)(a_iy0[j] + kh) < (unsigned)p.H) && ((unsigned)(a_ix0[j] + kw) < (unsigned)p.W);
          a_base[j] = ok ? a_org[j] + tapoff : OOB;
        }
      } else {
#pragma unroll
        for (int j = 0; j < AI; ++j) {
          int iy = a_iy0[j] + kh, ix = a_ix0[j] + kw;
          // transposed gather of a stride-2 data gradient: only even taps are real
          bool ok = a_ok[j] && (((iy | ix) & p.dmask) == 0);
          iy >>= p.dshift;
          ix >>= p.dshift;
          ok = ok && ((unsigned)iy < (unsigned)p.H) && ((unsigned)ix < (unsigned)p.W);
          a_base[j] = ok ? (unsigned)(((a_b[j] * p.H + iy) * p.W + ix) * p.C0) * 2u + a_kb[j] : OOB;
        }
      }
    }
#ifdef DY_PROBE
    // traffic probes: 0x20000 = only tap (0,0) fetches pixels, 0x40000 = only every other
    // slice fetches weights (the skipped DMAs still issue, out of range: zeros, no traffic)
    const bool skipA = (p.flags & 0x20000) && (kh | kw);
    const bool skipB = (p.flags & 0x40000) && ((k0 / BK) & 1);
#else
    constexpr bool skipA = false, skipB = false;
#endif
    if (ci0 < p.C0) {
      const unsigned cs2 = ci0 * 2;
      [&]<int... J>(std::integer_sequence<int, J...>) {
        (dma16<J * SLAB>(skipA ? OOB : a_base[J], srd0, cs2, sbase), ...);
      }(std::make_integer_sequence<int, AI>{});
    } else {
      const unsigned cs2 = (ci0 - p.C0) * 2;
      [&]<int... J>(std::integer_sequence<int, J...>) {
        (dma16<J * SLAB>(a_base1[J], srd1, cs2, sbase), ...);
      }(std::make_integer_sequence<int, AI>{});
    }
    const unsigned k2 = k0 * 2;
    [&]<int... J>(std::integer_sequence<int, J...>) {
      (dma16<A_BYTES + J * SLAB>(skipB ? OOB : b_base[J], srdw, k2, sbase), ...);
    }(std::make_integer_sequence<int, BI>{});
#pragma unroll
    for (int g = 0; g < KG; ++g) advance();
  };

  // the epilogue's per-channel scale/shift: fetched now (one channel per thread, two registers held
  // through the loop) and handed over through LDS after the loop, so their load latency is not
  // part of the epilogue of these ~10 us kernels
  float pf_sc = 1.f, pf_sh = 0.f;
  if (tid < BN && n0 + tid < p.Cout) {
    if (p.scale) pf_sc = p.scale[n0 + tid];
    if (p.shift) pf_sh = p.shift[n0 + tid];
  }

  f32x4 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- software pipeline: PRE tiles in flight.  Iteration kt: wait until this wave's DMAs
  //      of tile kt have landed (counted vmcnt leaves the younger tiles in flight), barrier
  //      (everyone's have, and everyone is done reading the stage refilled next), issue tile
  //      kt+PRE, then multiply tile kt.
#pragma unroll
  for (int s = 0; s < PRE; ++s)
    if (s < nkg) issue_tile(s);
#ifdef HALO_PROBE
  gp_t[1] = (long long)__builtin_amdgcn_s_memtime();
#endif

  const int frow = lane & 15, fchunk = lane >> 4;
#ifdef DY_PROBE
  bf16x8 pxf[2][MI], pwf[2][NI];
#endif
  for (int kt = 0; kt < nkg; ++kt) {
    if (PRE >= 1 && kt + PRE - 1 < nkg)
      wait_vmcnt<LPT*(PRE >= 1 ? PRE - 1 : 0)>();
    else
      wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
#ifdef DY_PROBE
    // ablation build (tools/bin/libdisyolo_probe.so): 0x4000 = no DMA in the loop,
    // 0x8000 = fragments read once, 0x10000 = no MFMA
    if (PRE >= 1) {
      if (kt + PRE < nkg && !(p.flags & 0x4000)) issue_tile((kt + PRE) % ST);
    }
    const char* sA = smem + (kg * ST + kt % ST) * STB;
    const char* sB = sA + A_BYTES;
    static_assert(BK / 32 <= 2, "");
    bf16x8 xf[2][MI], wf[2][NI];
    if (!(p.flags & 0x8000) || kt == 0) {
#pragma unroll
      for (int kk = 0; kk < BK / 32; ++kk) {
#pragma unroll
        for (int i = 0; i < MI; ++i) {
          const int row = wm * WTM + i * 16 + frow;
          pxf[kk][i] = *reinterpret_cast<const bf16x8*>(sA + row * ROWB + swz<BK>(row, kk * 4 + fchunk) * 16);
        }
#pragma unroll
        for (int j = 0; j < NI; ++j) {
          const int row = wn * WTN + j * 16 + frow;
          pwf[kk][j] = *reinterpret_cast<const bf16x8*>(sB + row * ROWB + swz<BK>(row, kk * 4 + fchunk) * 16);
        }
      }
    }
#pragma unroll
    for (int kk = 0; kk < BK / 32; ++kk) {
#pragma unroll
      for (int i = 0; i < MI; ++i) xf[kk][i] = pxf[kk][i];
#pragma unroll
      for (int j = 0; j < NI; ++j) wf[kk][j] = pwf[kk][j];
    }
    if (!(p.flags & 0x10000)) {
#pragma unroll
      for (int kk = 0; kk < BK / 32; ++kk)
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < NI; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[kk][j], xf[kk][i], acc[i][j], 0, 0, 0);
    } else {
#pragma unroll
      for (int kk = 0; kk < BK / 32; ++kk) {
#pragma unroll
        for (int i = 0; i < MI; ++i) asm volatile("" ::"v"(xf[kk][i]));
#pragma unroll
        for (int j = 0; j < NI; ++j) asm volatile("" ::"v"(wf[kk][j]));
      }
    }
#else
    if (PRE >= 1) {
      if (kt + PRE < nkg) issue_tile((kt + PRE) % ST);
    }
    const char* sA = smem + (kg * ST + kt % ST) * STB;
    const char* sB = sA + A_BYTES;
#pragma unroll
    for (int kk = 0; kk < BK / 32; ++kk) {
      bf16x8 xf[MI], wf[NI];
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        const int row = wm * WTM + i * 16 + frow;
        xf[i] = *reinterpret_cast<const bf16x8*>(sA + row * ROWB + swz<BK>(row, kk * 4 + fchunk) * 16);
      }
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        const int row = wn * WTN + j * 16 + frow;
        wf[j] = *reinterpret_cast<const bf16x8*>(sB + row * ROWB + swz<BK>(row, kk * 4 + fchunk) * 16);
      }
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], xf[i], acc[i][j], 0, 0, 0);
    }
#endif
    if (PRE == 0) {  // single stage: refill only after everyone has consumed the tile
      __builtin_amdgcn_s_barrier();
      if (kt + 1 < nkg) issue_tile(0);
    }
  }
  __syncthreads();  // all fragment reads done before the LDS is reused by the epilogue

  if (KG > 1) {
    // sum the groups' accumulators in group order through LDS ([KG-1][MI*NI][T] float4)
    float4* xg = reinterpret_cast<float4*>(smem);
    if (kg > 0) {
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
          xg[((kg - 1) * MI * NI + i * NI + j) * T + wave * 64 + lane] =
              make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
    }
    __syncthreads();
    if (kg == 0) {
#pragma unroll
      for (int g = 1; g < KG; ++g)
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < NI; ++j) {
            const float4 v = xg[((g - 1) * MI * NI + i * NI + j) * T + wave * 64 + lane];
            acc[i][j][0] += v.x;
            acc[i][j][1] += v.y;
            acc[i][j][2] += v.z;
            acc[i][j][3] += v.w;
          }
    }
    __syncthreads();
  }
#ifdef DY_PROBE
  if ((p.flags & 0x80000) && acc[0][0][0] != 123.456f) return;   // timing probe: no epilogue
#endif
#ifdef HALO_PROBE
  gp_t[2] = (long long)__builtin_amdgcn_s_memtime();
  struct GemmProbeEnd {
    const ConvParams& p; long long* t; int wave, lane, nw;
    __device__ ~GemmProbeEnd() {
      if (p.flags & 0x200000) {
        t[3] = (long long)__builtin_amdgcn_s_memtime();
        if (lane == 0) {
          long long* o = reinterpret_cast<long long*>(p.stats) + ((size_t)blockIdx.x * nw + wave) * 8;
          for (int k = 0; k < 5; ++k) o[k] = t[k];
        }
      }
    }
  } gp_end{p, gp_t, wave_all, lane, NW * KG};
#endif
  const bool ep = (KG == 1) || (kg == 0);   // only group 0 holds the full sums
  float* scsh = reinterpret_cast<float*>(smem + WM * BN * 8);   // [2][BN] behind the stats scratch
  if (tid < BN) {
    scsh[tid] = pf_sc;
    scsh[BN + tid] = pf_sh;
  }
  if (!(p.flags & DISYOLO_CONV_STATS)) __syncthreads();   // (the stats path below has its own barrier)

  // ---- epilogue.  acc[i][j][r]: pixel m0 + wm*WTM + i*16 + (lane&15),
  //      channel n0 + wn*WTN + j*16 + 4*(lane>>4) + r ----
  const int px = lane & 15, cq = lane >> 4;

  const bool fused = KG == 1 && (p.flags & DISYOLO_CONV_BN_FUSED) != 0;
  if (p.flags & DISYOLO_CONV_STATS) {
    // per-channel sum / sum of squares of the raw f32 accumulators over this block's
    // pixels (rows past M hold exact zeros).  Deterministic: fixed shuffle tree, then a
    // fixed-order sum over the WM waves through LDS.
    float* red = reinterpret_cast<float*>(smem);  // [WM][BN][2]; tiles are dead by now
#pragma unroll
    for (int j = 0; j < NI; ++j) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float s = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < MI; ++i) {
          const float v = acc[i][j][r];
          s += v;
          s2 += v * v;
        }
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) {
          s += __shfl_xor(s, o, 64);
          s2 += __shfl_xor(s2, o, 64);
        }
        if (px == 0 && ep) {
          const int nl = wn * WTN + j * 16 + cq * 4 + r;
          red[(wm * BN + nl) * 2 + 0] = s;
          red[(wm * BN + nl) * 2 + 1] = s2;
        }
      }
    }
    __syncthreads();
    for (int nl = tid; nl < BN; nl += T * KG) {
      const int n = n0 + nl;
      if (n < p.Cout) {
        float s = 0.f, s2 = 0.f;
#pragma unroll
        for (int w_ = 0; w_ < WM; ++w_) {
          s += red[(w_ * BN + nl) * 2 + 0];
          s2 += red[(w_ * BN + nl) * 2 + 1];
        }
        if (fused)
          cl_store2(p.stats + ((size_t)mt * p.Cout + n) * 2, s, s2);     // write-through: the other blocks of the cluster read it
        else
          stats_out(p, mt, n, s, s2);
      }
    }
    if (fused) cl_arrive(p.csync, nt);       // (drains the stores, block barrier, one arrival)
  }

  if (!ep) return;
  if (!(p.flags & DISYOLO_CONV_OUT_F32) && (p.Cout & 7) == 0) {
    // bf16 output: every wave stages its WTM x WTN tile in LDS (rows padded by 16 B: the 8-byte
    // writes of a 16-lane group and the 16-byte reads then spread over the banks) and writes
    // it out as 16 B per lane, WTN*2 contiguous bytes per pixel -- whole 128-byte lines
    // instead of the accumulator layout's 32-byte pieces of 16 different lines per store.
    constexpr int ROWP = WTN * 2 + 16;
    char* sw = smem + WM * BN * 8 + BN * 8 + wave * (WTM * ROWP);   // behind the stats scratch and scale/shift
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int n = n0 + wn * WTN + j * 16 + cq * 4;
      const int nl = wn * WTN + j * 16 + cq * 4;
      float sc[4], sh[4];
      *reinterpret_cast<float4*>(sc) = *reinterpret_cast<const float4*>(scsh + nl);
      *reinterpret_cast<float4*>(sh) = *reinterpret_cast<const float4*>(scsh + BN + nl);
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        const int m = m0 + wm * WTM + i * 16 + px;
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          v[r] = acc[i][j][r] * sc[r] + sh[r];
          if (p.flags & DISYOLO_CONV_LEAKY) v[r] = leaky(v[r], p.alpha);
        }
        if (p.residual && m < Mlim && n < p.Cout) {
          const uint2 rr = *reinterpret_cast<const uint2*>(p.residual + oaddr(m, n));
          v[0] += __builtin_bit_cast(float, rr.x << 16);
          v[1] += __builtin_bit_cast(float, rr.x & 0xffff0000u);
          v[2] += __builtin_bit_cast(float, rr.y << 16);
          v[3] += __builtin_bit_cast(float, rr.y & 0xffff0000u);
        }
        uint2 o2;
        o2.x = pack2(v[0], v[1]);
        o2.y = pack2(v[2], v[3]);
        *reinterpret_cast<uint2*>(sw + (i * 16 + px) * ROWP + (j * 16 + cq * 4) * 2) = o2;
      }
    }
    // written and read by the same wave: LDS ordering only, no block barrier
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    constexpr int CPR8 = WTN / 8, CH = WTM * CPR8;
    bf16* yo = reinterpret_cast<bf16*>(p.y);
    if constexpr (KG == 1 && EPI == 1) if (p.flags & DISYOLO_CONV_BN_BWD_STATS) {
      // whole = DISYOLO_CONV_BN_BWD_FUSED; without it (DISYOLO_CONV_BN_BWD_STATS alone, as on the patch kernels) only pass 1
      // runs: one row of partial sums per pixel tile for bn_act_bwd_partials, no exchange, y receives the gradient as usual
      const bool whole = p.flags & DISYOLO_CONV_BN_BWD_FUSED;
      // ---- the target layer's whole batch-norm backward inside this data-gradient conv (DISYOLO_CONV_BN_BWD_FUSED): the
      //      staged tile is the (now final) gradient wrt the target's ACTIVATION, rounded to bf16 as the separate launches
      //      would have read it back.  Pass 1: this block's (sum g, sum g*xhat) per channel from the staged values and the
      //      target's conv output -> one row of partials, exchanged within the launch (conv_common.h "cluster exchange").
      //      Pass 2: dx = scale*g - x*A + C (bn_bwd_finalize_kernel / bn_bwd_apply_kernel, bn.hip, value for value) -> y.
      //      The gradient wrt the activation never reaches memory; colreduce + bn_bwd_finalize + bn_bwd_apply are gone.
      constexpr int ITERS = (CH + 63) / 64;
      const int chl = lane % CPR8;                   // this lane's chunk column: the same in every round (64 % CPR8 == 0)
      const int ncol = n0 + wn * WTN + chl * 8;
      BnBwdLane bl;
      bl.init(p, ncol, ncol < p.Cout);
      uint4 bx[ITERS];
#pragma unroll
      for (int it = 0; it < ITERS; ++it) {
        const int idx = it * 64 + lane;
        const int m = m0 + wm * WTM + idx / CPR8;
        bx[it] = (idx < CH && m < Mlim && ncol < p.Cout) ? *reinterpret_cast<const uint4*>(p.bn_x + (size_t)m * p.Cout + ncol)
                                                          : uint4{0, 0, 0, 0};
      }
#pragma unroll
      for (int it = 0; it < ITERS; ++it) {
        const int idx = it * 64 + lane;
        const int row = idx / CPR8;
        if (idx < CH && m0 + wm * WTM + row < Mlim && ncol < p.Cout)
          bl.add(*reinterpret_cast<const uint4*>(sw + row * ROWP + chl * 16), bx[it], p.bn_alpha);
      }
      bl.template reduce<CPR8>();
      float* red = reinterpret_cast<float*>(smem);   // [WM][BN][2]: the statistics scratch (a data-gradient conv has no STATS)
      if (lane < CPR8) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          red[(wm * BN + wn * WTN + lane * 8 + k) * 2 + 0] = bl.s1[k];
          red[(wm * BN + wn * WTN + lane * 8 + k) * 2 + 1] = bl.s2[k];
        }
      }
      __syncthreads();
      for (int nl = tid; nl < BN; nl += T) {
        const int n = n0 + nl;
        if (n < p.Cout) {
          float s = 0.f, s2 = 0.f;
#pragma unroll
          for (int w_ = 0; w_ < WM; ++w_) {
            s += red[(w_ * BN + nl) * 2 + 0];
            s2 += red[(w_ * BN + nl) * 2 + 1];
          }
          if (whole)
            cl_store2(p.bn_part + ((size_t)mt * p.Cout + n) * 2, s, s2);
          else
            bnpart_out(p, mt, n, s, s2);
        }
      }
      if (!whole) {
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
          const int idx = it * 64 + lane;
          const int row = idx / CPR8;
          const int m = m0 + wm * WTM + row;
          if (idx < CH && m < Mlim && ncol < p.Cout)
            *reinterpret_cast<uint4*>(yo + oaddr(m, ncol)) = *reinterpret_cast<const uint4*>(sw + row * ROWP + chl * 16);
        }
        return;
      }
      cl_arrive(p.csync, nt);
      double* dscr = reinterpret_cast<double*>(smem + ((WM * BN * 8 + BN * 8 + NW * (WTM * ROWP) + 15) & ~15));
      cl_wait(p.csync, nt, (unsigned)p.tilesM, ((p.Cout + 15) >> 4) * CL_LINE);
      double r0, r1;
      if (cl_sum_rows<T, BN>(p.bn_part, p.tilesM, p.Cout, n0, dscr, r0, r1)) {
        const int n = n0 + tid;
        const float sc0 = p.bn_scale[n], rs0 = p.bn_rstd[n], mean0 = p.bn_mean[n];
        const float c1 = (float)(r0 * p.inv_count), c2 = (float)(r1 * p.inv_count);
        const float A = sc0 * rs0 * c2;
        scsh[tid] = A;
        scsh[BN + tid] = mean0 * A - sc0 * c1;
        if (mt == 0) {
          p.dbeta[n] = (float)r0;
          p.dgamma[n] = (float)r1;
        }
      }
      __syncthreads();
      cl_depart(p.csync, nt, (unsigned)p.tilesM);
      float A8[8], C8[8];
      {
        const int nl = wn * WTN + chl * 8;
        *reinterpret_cast<float4*>(A8) = *reinterpret_cast<const float4*>(scsh + nl);
        *reinterpret_cast<float4*>(A8 + 4) = *reinterpret_cast<const float4*>(scsh + nl + 4);
        *reinterpret_cast<float4*>(C8) = *reinterpret_cast<const float4*>(scsh + BN + nl);
        *reinterpret_cast<float4*>(C8 + 4) = *reinterpret_cast<const float4*>(scsh + BN + nl + 4);
      }
#pragma unroll
      for (int it = 0; it < ITERS; ++it) {
        const int idx = it * 64 + lane;
        const int row = idx / CPR8;
        const int m = m0 + wm * WTM + row;
        if (idx < CH && m < Mlim && ncol < p.Cout) {
          float g[8], vx[8];
          unpack8(*reinterpret_cast<const uint4*>(sw + row * ROWP + chl * 16), g);
          unpack8(bx[it], vx);
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            const float z = vx[k] * bl.sc[k] + bl.sh[k];
            const float gg = g[k] * (z > 0.f ? 1.f : p.bn_alpha);
            g[k] = bl.sc[k] * gg - vx[k] * A8[k] + C8[k];
          }
          *reinterpret_cast<uint4*>(yo + (size_t)m * p.Cout + ncol) = pack8(g);
        }
      }
      return;
    }
#pragma unroll
    for (int it = 0; it < (CH + 63) / 64; ++it) {
      const int idx = it * 64 + lane;
      const int row = idx / CPR8, ch = idx % CPR8;
      const int m = m0 + wm * WTM + row, n = n0 + wn * WTN + ch * 8;
      if (idx < CH && m < Mlim && n < p.Cout)
        *reinterpret_cast<uint4*>(yo + oaddr(m, n)) = *reinterpret_cast<const uint4*>(sw + row * ROWP + ch * 16);
    }
    if constexpr (KG == 1 && EPI == 0) if (fused) {
      // ---- batch norm inside the launch (DISYOLO_CONV_BN_FUSED): the conv output is on its way to y (above); now the
      //      statistics rows of ALL pixel tiles of this channel tile -> scale / shift (every block of the cluster computes
      //      the same bits), then the activation from the bf16 values still staged in LDS -> y_act.  What the separate
      //      bn_finalize + bn_act_fwd launches did, without their two launch boundaries and without re-reading y.
      double* dscr = reinterpret_cast<double*>(smem + ((WM * BN * 8 + BN * 8 + NW * (WTM * ROWP) + 15) & ~15));
      cl_wait(p.csync, nt, (unsigned)p.tilesM, ((p.Cout + 15) >> 4) * CL_LINE);
      double r0, r1;
      if (cl_sum_rows<T, BN>(p.stats, p.tilesM, p.Cout, n0, dscr, r0, r1)) {
        const int n = n0 + tid;                      // (g == 0: tid = channel of the tile)
        float sc_, sh_, meanf, varf, rstd;
        cl_bn_coeffs(r0, r1, p.inv_count, p.gamma[n], p.beta[n], p.bn_eps, sc_, sh_, meanf, varf, rstd);
        scsh[tid] = sc_;
        scsh[BN + tid] = sh_;
        if (mt == 0) {                               // one block per channel tile keeps the layer's books
          p.o_scale[n] = sc_;
          p.o_shift[n] = sh_;
          p.o_mean[n] = meanf;
          p.o_rstd[n] = rstd;
          if (p.mm) p.mm[n] = p.mm[n] * p.bn_decay + meanf * (1.0f - p.bn_decay);
          if (p.mv) p.mv[n] = p.mv[n] * p.bn_decay + varf * (1.0f - p.bn_decay);
        }
      }
      __syncthreads();
      cl_depart(p.csync, nt, (unsigned)p.tilesM);
      bf16* ya = reinterpret_cast<bf16*>(p.y_act);
      {
        const int chl = lane % CPR8;                 // this lane's chunk column: the same in every round (64 % CPR8 == 0)
        const int nl = wn * WTN + chl * 8;
        float sc8[8], sh8[8];
        *reinterpret_cast<float4*>(sc8) = *reinterpret_cast<const float4*>(scsh + nl);
        *reinterpret_cast<float4*>(sc8 + 4) = *reinterpret_cast<const float4*>(scsh + nl + 4);
        *reinterpret_cast<float4*>(sh8) = *reinterpret_cast<const float4*>(scsh + BN + nl);
        *reinterpret_cast<float4*>(sh8 + 4) = *reinterpret_cast<const float4*>(scsh + BN + nl + 4);
#pragma unroll
        for (int it = 0; it < (CH + 63) / 64; ++it) {
          const int idx = it * 64 + lane;
          const int row = idx / CPR8, ch = idx % CPR8;
          const int m = m0 + wm * WTM + row, n = n0 + wn * WTN + ch * 8;
          if (idx < CH && m < Mlim && n < p.Cout) {
            float v[8];
            unpack8(*reinterpret_cast<const uint4*>(sw + row * ROWP + ch * 16), v);
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = leaky(v[k] * sc8[k] + sh8[k], p.alpha);
            *reinterpret_cast<uint4*>(ya + oaddr(m, n)) = pack8(v);
          }
        }
      }
    }
    return;
  }
  const bool vec_ok = (p.Cout & 3) == 0;
#pragma unroll
  for (int j = 0; j < NI; ++j) {
    const int n = n0 + wn * WTN + j * 16 + cq * 4;
    if (n >= p.Cout) continue;
    float sc[4] = {1.f, 1.f, 1.f, 1.f}, sh[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (n + r < p.Cout) {
        if (p.scale) sc[r] = p.scale[n + r];
        if (p.shift) sh[r] = p.shift[n + r];
      }
    }
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const int m = m0 + wm * WTM + i * 16 + px;
      if (m >= Mlim) continue;
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        v[r] = acc[i][j][r] * sc[r] + sh[r];
        if (p.flags & DISYOLO_CONV_LEAKY) v[r] = leaky(v[r], p.alpha);
      }
      const size_t off = oaddr(m, n);
      if (vec_ok) {
        if (p.residual) {
          const uint2 rr = *reinterpret_cast<const uint2*>(p.residual + off);
          v[0] += __builtin_bit_cast(float, rr.x << 16);
          v[1] += __builtin_bit_cast(float, rr.x & 0xffff0000u);
          v[2] += __builtin_bit_cast(float, rr.y << 16);
          v[3] += __builtin_bit_cast(float, rr.y & 0xffff0000u);
        }
        if (p.flags & DISYOLO_CONV_OUT_F32) {
          *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.y) + off) = make_float4(v[0], v[1], v[2], v[3]);
        } else {
          uint2 o;
          o.x = pack2(v[0], v[1]);
          o.y = pack2(v[2], v[3]);
          *reinterpret_cast<uint2*>(reinterpret_cast<bf16*>(p.y) + off) = o;
        }
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (n + r < p.Cout) {
            float o = v[r];
            if (p.residual) o += (float)p.residual[off + r];
            if (p.flags & DISYOLO_CONV_OUT_F32)
              reinterpret_cast<float*>(p.y)[off + r] = o;
            else
              reinterpret_cast<bf16*>(p.y)[off + r] = (bf16)o;
          }
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// 3x3 stride-1 convolution with halo reuse ("patch" kernel).
//
// The implicit-GEMM kernel above stages every input pixel once per filter tap (9x) and its
// main loop is bound by the bytes a CU can stage per FLOP (global -> LDS ~50 B/clk/CU, each
// 1 KiB LDS-DMA costing ~60 issue cycles; DESIGN.md "What bounds the conv kernel").  Here a
// block owns a PH x PW patch of output pixels of ONE image and BN = 16*NI output channels:
// per 32-channel slice it stages the (PH+2) x (PW+2) halo ONCE plus the 9 x BN x 32 weights,
// then runs all 9 taps from LDS -- the tap shift is just a different LDS row per lane.
// For an 18x18 patch x 64 channels that is 108 B staged per K element for 20,736 outputs,
// against 192 B for 8,192 outputs of the 64x128 GEMM tile: 4.5x fewer bytes per FLOP, one
// barrier per 32 input channels instead of one per 64 K elements, 9 DMAs per 108 MFMAs per wave.
//
// Waves: NW per block; M fragment f (16 consecutive patch pixels) belongs to wave f % NW, each
// wave holds up to FW of them x NI channel fragments.  LDS per stage: halo [pixels][32 ch] +
// weights [tap][BN][32], 64-byte rows, 16-byte chunks XOR-swizzled as in the GEMM kernel;
// 2 stages (compute slice c while slice c+1 lands).

#ifdef HALO_ABL_NOMFMA
__device__ __forceinline__ void abl_keep(const bf16x8& a, const bf16x8& b, f32x4& c) { asm volatile("" ::"v"(a), "v"(b), "v"(c)); }
#endif
template <int NW, int FW, int NI>
__global__ __launch_bounds__(NW * 64) void conv_halo_kernel(ConvParams p, int PH, int PW, int tilesY, int tilesX) {
#ifdef HALO_PROBE
  // tools/probe_halo.py: with flag 0x200000 the stats pointer receives, per wave, s_memtime at [0] kernel entry, [1] first
  // DMAs issued, [2] main loop done, [3] end of the epilogue, and s_memrealtime at entry in [4]
  long long hp_t[5];
  hp_t[0] = (long long)__builtin_amdgcn_s_memtime();
  hp_t[4] = (long long)__builtin_amdgcn_s_memrealtime();
#endif
  constexpr int BN = NI * 16;
  constexpr int SLAB = NW * 1024;
  constexpr int AIM = 4;                                        // halo DMAs per wave (<= 64*NW pixels)
  constexpr int BIM = (9 * BN * 4 + NW * 64 - 1) / (NW * 64);   // weight DMAs per wave
  constexpr int A_BYTES = AIM * SLAB;
  constexpr int STB = (AIM + BIM) * SLAB;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  int tile;
  {  // XCD-aware order, channel tile fastest (blocks sharing a halo sit on one L2)
    const int nblk = gridDim.x, bid = blockIdx.x;
    const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, loc = bid >> 3;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
  }
  // (xcd_n: where the weights outweigh the input -- the 18^2 layers -- an XCD's run goes patch fastest instead: the
  //  blocks that share a channel tile's weights sit on one L2 and every XCD reads 1/8 of the weights, not all of them)
  int mt, nt;
  if (p.xcd_n) {
    nt = tile / p.tilesM;
    mt = tile - nt * p.tilesM;
  } else {
    mt = tile / p.tilesN;
    nt = tile - mt * p.tilesN;
  }
  const int n0 = nt * BN;
  const int tpi = tilesY * tilesX;             // patches per image
  const int b = mt / tpi, pr = mt - b * tpi;
  const int ty = pr / tilesX, tx = pr - ty * tilesX;
  const int y0 = ty * PH, x0 = tx * PW;        // patch origin (output = input coordinates: SAME, stride 1)
  const int HW_ = PW + 2, NPIX = (PH + 2) * HW_, NOUT = PH * PW;
  // Where halo row hy sits in the LDS image.  Bank slot of a fragment read = (LDS row mod 8, chunk): swz_any makes 16 CONSECUTIVE
  // rows conflict-free, but a 16-pixel fragment of the PW-wide patch wraps to the next patch row in most alignments, and with the
  // halo rows stored one after the other (pitch PW + 2) its LDS rows then jump by 3 -- 23 / 27 / 32 % of the patch kernels' LDS
  // time were bank conflicts (profiles/r06_conv_sq_counters.txt).  With the EVEN halo rows stored first and the odd ones behind
  // them at an offset of 2 mod 8, row hy + 1 always starts PW mod 8 rows after row hy's start (pitch 20 or 28: 4 mod 8), so the
  // wrap is one row further mod 8, like a consecutive row.  Same 16-byte units, same swizzle, no extra LDS.  Other pitches keep
  // the linear image.
  const bool split_on = p.halo_split != 0;                                           // (launch_halo: DISYOLO_HALO_SPLIT_ROWS=0 keeps the linear image, for A/B)
  const int odd_base0 = ((((PH + 3) >> 1) * HW_ + 7) & ~7) + 2;                      // first slot of the odd rows: 2 mod 8
  const bool split_rows = split_on && (HW_ & 7) == 4 && odd_base0 + ((PH + 2) >> 1) * HW_ <= AIM * NW * 16;   // (the image must fit the AIM halo DMAs)
  const int odd_base = split_rows ? odd_base0 : 0;
  auto row_start = [&](int hy) { return split_rows ? ((hy & 1) ? odd_base + (hy >> 1) * HW_ : (hy >> 1) * HW_) : hy * HW_; };

  const i32x4 srd0 = make_srd(p.x0, p.bytes0);
  const i32x4 srdw = make_srd(p.w, p.bytesw);
  // halo gather offsets (constant over the K loop; the channel slice goes into the SGPR offset)
  unsigned a_off[AIM];
#pragma unroll
  for (int j = 0; j < AIM; ++j) {
    const int chunk = (j * NW + wave) * 64 + lane;
    const int hp = chunk >> 2, pc = chunk & 3;
    int hy, hx;
    bool slot_ok = hp < NPIX;
    if (split_rows) {      // slot -> halo pixel: even rows in [0, n_even * HW_), odd rows from odd_base
      const int n_even = (PH + 3) >> 1, n_odd = (PH + 2) >> 1;
      const bool odd = hp >= odd_base;
      const int rel = odd ? hp - odd_base : hp;
      int r2;
      divmod_small(rel, HW_, r2, hx);
      hy = 2 * r2 + (odd ? 1 : 0);
      slot_ok = odd ? (r2 < n_odd) : (hp < n_even * HW_);
    } else {
      divmod_small(hp < NPIX ? hp : 0, HW_, hy, hx);
    }
    const int iy = y0 + hy - p.pad_t, ix = x0 + hx - p.pad_l;
    const bool ok = slot_ok && ((unsigned)iy < (unsigned)p.H) && ((unsigned)ix < (unsigned)p.W);
    a_off[j] = ok ? (unsigned)(((b * p.H + iy) * p.W + ix) * p.C0) * 2u + swz_any(hp, pc) * 16 : OOB;
  }
  unsigned b_off[BIM];
#pragma unroll
  for (int j = 0; j < BIM; ++j) {
    const int chunk = (j * NW + wave) * 64 + lane;
    const int rb = chunk >> 2, pc = chunk & 3;     // row = tap*BN + n
    const int tap = rb / BN, nl = rb - tap * BN;
    const bool ok = (rb < 9 * BN) && (n0 + nl < p.Cout);
    b_off[j] = ok ? ((unsigned)(n0 + nl) * (unsigned)p.K + (unsigned)(tap * p.Cin)) * 2u + swz_any(rb, pc) * 16 : OOB;
  }
  // DMA i of a slice: i < AIM halo, else weights.  The first slice is issued in one go; the
  // following ones are spread over the 9 taps of the slice being multiplied, so a wave's DMA
  // issue (~60 cycles each) overlaps its own MFMAs instead of preceding them.
#ifndef HALO_DMA_TAPS
#define HALO_DMA_TAPS 5
#endif
  constexpr int NDMA = AIM + BIM, DPT = (NDMA + HALO_DMA_TAPS - 1) / HALO_DMA_TAPS;   // the next slice's DMAs go out during the first 5 taps: they have 4 taps left to land
  auto issue_one = [&]<int I>(std::integral_constant<int, I>, unsigned cs2 /* byte offset of the channel slice */, int stage) {
    if constexpr (I < NDMA) {
      const unsigned sbase = lds0 + stage * STB + wave * 1024;
      if constexpr (I < AIM)
        dma16<I * SLAB>(a_off[I], srd0, cs2, sbase);
      else
        dma16<A_BYTES + (I - AIM) * SLAB>(b_off[I - AIM], srdw, cs2, sbase);
    }
  };
  auto issue = [&](int c, int stage) {
    [&]<int... I>(std::integer_sequence<int, I...>) {
      (issue_one(std::integral_constant<int, I>{}, (unsigned)c * 64u, stage), ...);
    }(std::make_integer_sequence<int, NDMA>{});
  };

  // fragment read state
  const int frow = lane & 15, fchunk = lane >> 4;
  const int nfrags = (NOUT + 15) >> 4;
  int hp0[FW][3];       // LDS slot of this lane's patch pixel at tap (kh, 0), per M fragment (the row start depends on the row's parity)
  int q_of[FW];         // patch pixel index (or -1)
#pragma unroll
  for (int t = 0; t < FW; ++t) {
    const int f = wave + NW * t;
    const int q = f * 16 + frow;
    const bool ok = q < NOUT;
    int py, px;
    divmod_small(ok ? q : 0, PW, py, px);
#pragma unroll
    for (int kh_ = 0; kh_ < 3; ++kh_) hp0[t][kh_] = row_start(py + kh_) + px;
    q_of[t] = ok ? q : -1;
  }
  int nf_w = 0;  // fragments this wave owns (wave-uniform)
#pragma unroll
  for (int t = 0; t < FW; ++t) nf_w += (wave + NW * t < nfrags) ? 1 : 0;
  nf_w = __builtin_amdgcn_readfirstlane(nf_w);
  unsigned wb[NI];      // weight fragment byte address at tap 0 (the swizzle does not depend on the tap: BN % 16 == 0)
#pragma unroll
  for (int j = 0; j < NI; ++j) {
    const int rb = j * 16 + frow;
    wb[j] = A_BYTES + rb * 64 + swz_any(rb, fchunk) * 16;
  }

  f32x4 acc[FW][NI];
#pragma unroll
  for (int t = 0; t < FW; ++t)
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[t][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nch = p.Cin >> 5;
  issue(0, 0);
#ifdef HALO_PROBE
  hp_t[1] = (long long)__builtin_amdgcn_s_memtime();
#endif
  // Fragment registers are double-buffered over the taps: the LDS reads of tap T+1 are issued before the MFMAs of tap
  // T, so a group of MFMAs never waits for the round trip of its own operands.  (Round 2 read, waited lgkmcnt(0) and
  // multiplied one M fragment at a time, inside a wave-uniform branch per fragment: every 4 MFMAs paid a full LDS
  // latency and the scheduler could not move anything across the branches -- MFMA pipe 25-28 % busy.)  The slice body
  // is branch-free: a wave that owns fewer than FW fragments multiplies a dummy one (pixel 0 of the patch, never
  // stored; the block is as slow as its fullest wave anyway), and the last slice -- which has no successor to
  // prefetch -- is a second copy of the body without the DMAs instead of a branch around each of them.  A layer with
  // 32 input channels has ONE slice: it never touches the second stage, the launcher then allocates one stage only
  // and more blocks share a CU (their prologues and epilogues overlap).
  bf16x8 wfr[2][NI], xfr[2][FW];
  auto load_frags = [&]<int TAP>(std::integral_constant<int, TAP>, const char* st) {
    constexpr int kh = TAP / 3, kw = TAP % 3, bi = TAP & 1;
#pragma unroll
#ifdef HALO_ABL_NOREAD      // (compile-time ablation, probe builds only: the fragments are never read)
    for (int j = 0; j < NI; ++j) asm volatile("" : "+v"(wfr[bi][j]));
#pragma unroll
    for (int t = 0; t < FW; ++t) asm volatile("" : "+v"(xfr[bi][t]));
    (void)kh; (void)kw; (void)st;
#else
    for (int j = 0; j < NI; ++j) wfr[bi][j] = *reinterpret_cast<const bf16x8*>(st + wb[j] + TAP * BN * 64);
#pragma unroll
    for (int t = 0; t < FW; ++t) {
      const int hp = hp0[t][kh] + kw;
      xfr[bi][t] = *reinterpret_cast<const bf16x8*>(st + hp * 64 + swz_any(hp, fchunk) * 16);
    }
#endif
  };
  auto slice_body = [&]<bool DMA>(std::bool_constant<DMA>, int c) {
#ifndef HALO_ABL_NOBAR
    wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();   // slice c has landed for everyone; everyone is done with slice c-1
#endif
    asm volatile("" ::: "memory");
    const unsigned cs_next = (unsigned)(c + 1) * 64u;
    const int stage_next = (c + 1) & 1;
    const char* st = smem + (c & 1) * STB;
    load_frags(std::integral_constant<int, 0>{}, st);
    auto tap_body = [&]<int TAP>(std::integral_constant<int, TAP>) {
      constexpr int bi = TAP & 1;
      if constexpr (TAP < 8) load_frags(std::integral_constant<int, TAP + 1>{}, st);
      __builtin_amdgcn_sched_barrier(0);   // (hipcc otherwise sinks the reads back next to their uses)
#ifndef HALO_ABL_NODMA
      if constexpr (DMA) {
        [&]<int... D>(std::integer_sequence<int, D...>) {
          (issue_one(std::integral_constant<int, TAP * DPT + D>{}, cs_next, stage_next), ...);
        }(std::make_integer_sequence<int, DPT>{});
      }
#endif
#pragma unroll
      for (int t = 0; t < FW; ++t)
#pragma unroll
        for (int j = 0; j < NI; ++j) {
#ifdef HALO_ABL_NOMFMA
          abl_keep(xfr[bi][t], wfr[bi][j], acc[t][j]);
#else
          acc[t][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wfr[bi][j], xfr[bi][t], acc[t][j], 0, 0, 0);
#endif
        }
    };
    [&]<int... T>(std::integer_sequence<int, T...>) {
      (tap_body(std::integral_constant<int, T>{}), ...);
    }(std::make_integer_sequence<int, 9>{});
  };
  for (int c = 0; c + 1 < nch; ++c) slice_body(std::bool_constant<true>{}, c);
  slice_body(std::bool_constant<false>{}, nch - 1);
  __syncthreads();
#ifdef HALO_PROBE
  hp_t[2] = (long long)__builtin_amdgcn_s_memtime();
  struct HaloProbeEnd {
    const ConvParams& p; long long* t; int wave, lane, nw;
    __device__ ~HaloProbeEnd() {
      if (p.flags & 0x200000) {
        t[3] = (long long)__builtin_amdgcn_s_memtime();
        if (lane == 0) {
          long long* o = reinterpret_cast<long long*>(p.stats) + ((size_t)blockIdx.x * nw + wave) * 8;
          for (int k = 0; k < 5; ++k) o[k] = t[k];
        }
      }
    }
  } hp_end{p, hp_t, wave, lane, NW};
#endif

  // ---- epilogue.  acc[t][j][r]: patch pixel q_of[t] (lane & 15), channel n0 + j*16 + 4*(lane>>4) + r
  const int cq = lane >> 4;
#ifdef DY_PROBE
  if ((p.flags & 0x80000) && acc[0][0][0] != 123.456f) return;   // timing probe: no epilogue
#endif
  const bool fusedf = (p.flags & DISYOLO_CONV_BN_FUSED) != 0;     // training-mode batch norm inside the launch (see the GEMM kernel)
  if (p.flags & DISYOLO_CONV_STATS) {
    float* red = reinterpret_cast<float*>(smem);  // [NW][BN][2]
#pragma unroll
    for (int j = 0; j < NI; ++j) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float s = 0.f, s2 = 0.f;
#pragma unroll
        for (int t = 0; t < FW; ++t) {
          const float v = (t < nf_w && q_of[t] >= 0) ? acc[t][j][r] : 0.f;
          s += v;
          s2 += v * v;
        }
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) {
          s += __shfl_xor(s, o, 64);
          s2 += __shfl_xor(s2, o, 64);
        }
        if (frow == 0) {
          const int nl = j * 16 + cq * 4 + r;
          red[(wave * BN + nl) * 2 + 0] = s;
          red[(wave * BN + nl) * 2 + 1] = s2;
        }
      }
    }
    __syncthreads();
    for (int nl = tid; nl < BN; nl += NW * 64) {
      const int n = n0 + nl;
      if (n < p.Cout) {
        float s = 0.f, s2 = 0.f;
#pragma unroll
        for (int w_ = 0; w_ < NW; ++w_) {
          s += red[(w_ * BN + nl) * 2 + 0];
          s2 += red[(w_ * BN + nl) * 2 + 1];
        }
        if (fusedf)
          cl_store2(p.stats + ((size_t)mt * p.Cout + n) * 2, s, s2);
        else
          stats_out(p, mt, n, s, s2);
      }
    }
    if (fusedf) cl_arrive(p.csync, nt);
  }

  if (!(p.flags & DISYOLO_CONV_OUT_F32) && (p.Cout & 7) == 0) {
    // coalesced bf16 stores through a per-wave LDS staging tile (see the GEMM kernel's epilogue)
    constexpr int ROWP = BN * 2 + 16;
    char* sw = smem + NW * BN * 8 + wave * (FW * 16 * ROWP);
    int m_of[FW];
#pragma unroll
    for (int t = 0; t < FW; ++t) {
      int py, px;
      divmod_small(q_of[t] >= 0 ? q_of[t] : 0, PW, py, px);
      m_of[t] = (t < nf_w && q_of[t] >= 0) ? (b * p.Ho + y0 + py) * p.Wo + x0 + px : -1;
    }
    constexpr int CPR8 = BN / 8;   // 16-byte chunks per pixel row
    static_assert(64 % CPR8 == 0, "a lane keeps its chunk column over the write-out rounds");
    constexpr int RPF = (16 * CPR8 + 63) / 64;
    // batch-norm backward sums (DISYOLO_CONV_BN_BWD_STATS): the target layer's conv output for the chunks this
    // lane will store, requested now so that the loads fly while the tile is scaled, packed and staged
    const bool bnb = p.flags & DISYOLO_CONV_BN_BWD_STATS;
    const bool bwdf = bnb && (p.flags & DISYOLO_CONV_BN_BWD_FUSED);
    uint4 bx[FW * RPF];
    if (bnb) {
#pragma unroll
      for (int t = 0; t < FW; ++t)
#pragma unroll
        for (int it = 0; it < RPF; ++it) {
          const int idx = it * 64 + lane;
          const int m = idx < 16 * CPR8 ? __shfl(m_of[t], idx / CPR8, 64) : -1;
          const int n = n0 + (idx % CPR8) * 8;
          bx[t * RPF + it] = (m >= 0 && n < p.Cout) ? *reinterpret_cast<const uint4*>(p.bn_x + (size_t)m * p.Cout + n)
                                                     : uint4{0, 0, 0, 0};
        }
    }
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int n = n0 + j * 16 + cq * 4;
      float sc[4] = {1.f, 1.f, 1.f, 1.f}, sh[4] = {0.f, 0.f, 0.f, 0.f};
      if (n < p.Cout) {
        if (p.scale) *reinterpret_cast<float4*>(sc) = *reinterpret_cast<const float4*>(p.scale + n);
        if (p.shift) *reinterpret_cast<float4*>(sh) = *reinterpret_cast<const float4*>(p.shift + n);
      }
#pragma unroll
      for (int t = 0; t < FW; ++t) {
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          v[r] = acc[t][j][r] * sc[r] + sh[r];
          if (p.flags & DISYOLO_CONV_LEAKY) v[r] = leaky(v[r], p.alpha);
        }
        if (p.residual && m_of[t] >= 0 && n < p.Cout) {
          const uint2 rr = *reinterpret_cast<const uint2*>(p.residual + (size_t)m_of[t] * p.Cout + n);
          v[0] += __builtin_bit_cast(float, rr.x << 16);
          v[1] += __builtin_bit_cast(float, rr.x & 0xffff0000u);
          v[2] += __builtin_bit_cast(float, rr.y << 16);
          v[3] += __builtin_bit_cast(float, rr.y & 0xffff0000u);
        }
        uint2 o2;
        o2.x = pack2(v[0], v[1]);
        o2.y = pack2(v[2], v[3]);
        *reinterpret_cast<uint2*>(sw + (t * 16 + frow) * ROWP + (j * 16 + cq * 4) * 2) = o2;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    bf16* yo = reinterpret_cast<bf16*>(p.y);
    BnBwdLane bl;
    if (bnb) bl.init(p, n0 + (lane % CPR8) * 8, n0 + (lane % CPR8) * 8 < p.Cout);
#pragma unroll
    for (int t = 0; t < FW; ++t) {
#pragma unroll
      for (int it = 0; it < RPF; ++it) {
        const int idx = it * 64 + lane;
        const int r16 = idx / CPR8, ch = idx % CPR8;
        // the pixel of row r16 of this fragment lives in lane r16 (any cq) of m_of[t]
        const int m = __shfl(m_of[t], r16, 64);
        const int n = n0 + ch * 8;
        if (idx < 16 * CPR8 && m >= 0 && n < p.Cout) {      // (16 channels per block: the 32 chunks of a fragment fill half a round)
          const uint4 o = *reinterpret_cast<const uint4*>(sw + (t * 16 + r16) * ROWP + ch * 16);
          if (!bwdf) *reinterpret_cast<uint4*>(yo + (size_t)m * p.Cout + n) = o;     // (fused backward: y receives dx in pass 2)
          if (bnb) bl.add(o, bx[t * RPF + it], p.bn_alpha);
        }
      }
    }
    if (bnb) {
      // lanes -> waves (fixed order through LDS) -> one row of partials per patch
      bl.template reduce<CPR8>();
      float* red = reinterpret_cast<float*>(smem);  // [NW][BN][2]; the tiles are dead by now
      if (lane < CPR8) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          red[(wave * BN + lane * 8 + k) * 2 + 0] = bl.s1[k];
          red[(wave * BN + lane * 8 + k) * 2 + 1] = bl.s2[k];
        }
      }
      __syncthreads();
      for (int nl = tid; nl < BN; nl += NW * 64) {
        const int n = n0 + nl;
        if (n < p.Cout) {
          float s = 0.f, s2 = 0.f;
#pragma unroll
          for (int w_ = 0; w_ < NW; ++w_) {
            s += red[(w_ * BN + nl) * 2 + 0];
            s2 += red[(w_ * BN + nl) * 2 + 1];
          }
          if (bwdf)
            cl_store2(p.bn_part + ((size_t)mt * p.Cout + n) * 2, s, s2);
          else
            bnpart_out(p, mt, n, s, s2);
        }
      }
      if (bwdf) {
        // ---- DISYOLO_CONV_BN_BWD_FUSED (the GEMM kernel's epilogue has the commentary): exchange the rows, then pass 2
        cl_arrive(p.csync, nt);
        float* scsh = reinterpret_cast<float*>(smem + NW * BN * 8 + NW * (FW * 16 * ROWP));
        double* dscr = reinterpret_cast<double*>(smem + ((NW * BN * 8 + NW * (FW * 16 * ROWP) + BN * 8 + 15) & ~15));
        cl_wait(p.csync, nt, (unsigned)p.tilesM, ((p.Cout + 15) >> 4) * CL_LINE);
        double r0, r1;
        if (cl_sum_rows<NW * 64, BN>(p.bn_part, p.tilesM, p.Cout, n0, dscr, r0, r1)) {
          const int n = n0 + tid;
          const float sc0 = p.bn_scale[n], rs0 = p.bn_rstd[n], mean0 = p.bn_mean[n];
          const float c1 = (float)(r0 * p.inv_count), c2 = (float)(r1 * p.inv_count);
          const float A = sc0 * rs0 * c2;
          scsh[tid] = A;
          scsh[BN + tid] = mean0 * A - sc0 * c1;
          if (mt == 0) {
            p.dbeta[n] = (float)r0;
            p.dgamma[n] = (float)r1;
          }
        }
        __syncthreads();
        cl_depart(p.csync, nt, (unsigned)p.tilesM);
        float A8[8], C8[8];
        {
          const int nl = (lane % CPR8) * 8;
          *reinterpret_cast<float4*>(A8) = *reinterpret_cast<const float4*>(scsh + nl);
          *reinterpret_cast<float4*>(A8 + 4) = *reinterpret_cast<const float4*>(scsh + nl + 4);
          *reinterpret_cast<float4*>(C8) = *reinterpret_cast<const float4*>(scsh + BN + nl);
          *reinterpret_cast<float4*>(C8 + 4) = *reinterpret_cast<const float4*>(scsh + BN + nl + 4);
        }
#pragma unroll
        for (int t = 0; t < FW; ++t) {
#pragma unroll
          for (int it = 0; it < RPF; ++it) {
            const int idx = it * 64 + lane;
            const int r16 = idx / CPR8, ch = idx % CPR8;
            const int m = __shfl(m_of[t], r16, 64);
            const int n = n0 + ch * 8;
            if (idx < 16 * CPR8 && m >= 0 && n < p.Cout) {
              float g[8], vx[8];
              unpack8(*reinterpret_cast<const uint4*>(sw + (t * 16 + r16) * ROWP + ch * 16), g);
              unpack8(bx[t * RPF + it], vx);
#pragma unroll
              for (int k = 0; k < 8; ++k) {
                const float z = vx[k] * bl.sc[k] + bl.sh[k];
                const float gg = g[k] * (z > 0.f ? 1.f : p.bn_alpha);
                g[k] = bl.sc[k] * gg - vx[k] * A8[k] + C8[k];
              }
              *reinterpret_cast<uint4*>(yo + (size_t)m * p.Cout + n) = pack8(g);
            }
          }
        }
      }
    }
    if (fusedf) {
      // ---- batch norm inside the launch (DISYOLO_CONV_BN_FUSED; the GEMM kernel's epilogue has the commentary): the
      //      statistics rows of all patches of this channel tile -> scale / shift, the activation from the staged tile
      float* scsh = reinterpret_cast<float*>(smem + NW * BN * 8 + NW * (FW * 16 * ROWP));   // [2][BN] behind the staging tiles
      double* dscr = reinterpret_cast<double*>(smem + ((NW * BN * 8 + NW * (FW * 16 * ROWP) + BN * 8 + 15) & ~15));
      cl_wait(p.csync, nt, (unsigned)p.tilesM, ((p.Cout + 15) >> 4) * CL_LINE);
      double r0, r1;
      if (cl_sum_rows<NW * 64, BN>(p.stats, p.tilesM, p.Cout, n0, dscr, r0, r1)) {
        const int n = n0 + tid;
        float sc_, sh_, meanf, varf, rstd;
        cl_bn_coeffs(r0, r1, p.inv_count, p.gamma[n], p.beta[n], p.bn_eps, sc_, sh_, meanf, varf, rstd);
        scsh[tid] = sc_;
        scsh[BN + tid] = sh_;
        if (mt == 0) {
          p.o_scale[n] = sc_;
          p.o_shift[n] = sh_;
          p.o_mean[n] = meanf;
          p.o_rstd[n] = rstd;
          if (p.mm) p.mm[n] = p.mm[n] * p.bn_decay + meanf * (1.0f - p.bn_decay);
          if (p.mv) p.mv[n] = p.mv[n] * p.bn_decay + varf * (1.0f - p.bn_decay);
        }
      }
      __syncthreads();
      cl_depart(p.csync, nt, (unsigned)p.tilesM);
      bf16* ya = reinterpret_cast<bf16*>(p.y_act);
      const int nl = (lane % CPR8) * 8;
      float sc8[8], sh8[8];
      *reinterpret_cast<float4*>(sc8) = *reinterpret_cast<const float4*>(scsh + nl);
      *reinterpret_cast<float4*>(sc8 + 4) = *reinterpret_cast<const float4*>(scsh + nl + 4);
      *reinterpret_cast<float4*>(sh8) = *reinterpret_cast<const float4*>(scsh + BN + nl);
      *reinterpret_cast<float4*>(sh8 + 4) = *reinterpret_cast<const float4*>(scsh + BN + nl + 4);
#pragma unroll
      for (int t = 0; t < FW; ++t) {
#pragma unroll
        for (int it = 0; it < RPF; ++it) {
          const int idx = it * 64 + lane;
          const int r16 = idx / CPR8, ch = idx % CPR8;
          const int m = __shfl(m_of[t], r16, 64);
          const int n = n0 + ch * 8;
          if (idx < 16 * CPR8 && m >= 0 && n < p.Cout) {
            float v[8];
            unpack8(*reinterpret_cast<const uint4*>(sw + (t * 16 + r16) * ROWP + ch * 16), v);
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = leaky(v[k] * sc8[k] + sh8[k], p.alpha);
            *reinterpret_cast<uint4*>(ya + (size_t)m * p.Cout + n) = pack8(v);
          }
        }
      }
    }
    return;
  }
  const bool vec_ok = (p.Cout & 3) == 0;
#pragma unroll
  for (int j = 0; j < NI; ++j) {
    const int n = n0 + j * 16 + cq * 4;
    if (n >= p.Cout) continue;
    float sc[4] = {1.f, 1.f, 1.f, 1.f}, sh[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (n + r < p.Cout) {
        if (p.scale) sc[r] = p.scale[n + r];
        if (p.shift) sh[r] = p.shift[n + r];
      }
    }
#pragma unroll
    for (int t = 0; t < FW; ++t) {
      if (!(t < nf_w) || q_of[t] < 0) continue;
      int py, px;
      divmod_small(q_of[t], PW, py, px);
      const int m = (b * p.Ho + y0 + py) * p.Wo + x0 + px;
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        v[r] = acc[t][j][r] * sc[r] + sh[r];
        if (p.flags & DISYOLO_CONV_LEAKY) v[r] = leaky(v[r], p.alpha);
      }
      const size_t off = (size_t)m * p.Cout + n;
      if (vec_ok) {
        if (p.residual) {
          const uint2 rr = *reinterpret_cast<const uint2*>(p.residual + off);
          v[0] += __builtin_bit_cast(float, rr.x << 16);
          v[1] += __builtin_bit_cast(float, rr.x & 0xffff0000u);
          v[2] += __builtin_bit_cast(float, rr.y << 16);
          v[3] += __builtin_bit_cast(float, rr.y & 0xffff0000u);
        }
        if (p.flags & DISYOLO_CONV_OUT_F32) {
          *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.y) + off) = make_float4(v[0], v[1], v[2], v[3]);
        } else {
          uint2 o;
          o.x = pack2(v[0], v[1]);
          o.y = pack2(v[2], v[3]);
          *reinterpret_cast<uint2*>(reinterpret_cast<bf16*>(p.y) + off) = o;
        }
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (n + r < p.Cout) {
            float o = v[r];
            if (p.residual) o += (float)p.residual[off + r];
            if (p.flags & DISYOLO_CONV_OUT_F32)
              reinterpret_cast<float*>(p.y)[off + r] = o;
            else
              reinterpret_cast<bf16*>(p.y)[off + r] = (bf16)o;
          }
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Streaming variant of the patch kernel for 3x3 stride-1 layers with 32 input channels (ONE K slice): the wide, shallow
// ends of the network (32 -> 64 at 288^2, forward of conv4 / conv81), whose time is memory traffic, not MFMA work.
//
// Ablations of the per-patch kernel on that shape (tools/r3_call16.sh, B = 8): 71 us in all; 40 us with the MFMAs AND
// the epilogue compiled out -- i.e. staging alone (launch, 37 KB of weights and a 30 KB halo per block, one block per
// CU, 6.75 rounds of blocks) costs 5.9 us per block, the cold-start rate of a CU's memory path (~12 B/clk), against
// 7 us for reading the whole input once at HBM speed.  Nothing overlaps: load, multiply, store, next block.
//
// Here a block is persistent: it loads its 9 x BN x 32 weights ONCE, then walks over patches g, g+G, g+2G ...; the
// halo of the NEXT patch is in flight (LDS-DMA into the other of two halo stages) while the current one is multiplied
// and stored.  The stores leave as buffer stores issued from inline asm -- always FW*RPF per wave and patch, lanes
// outside the tensor get an out-of-range offset and are dropped by the range check -- so the wait at the top of the
// loop is a COUNTED vmcnt that retires the halo DMAs but leaves the previous patch's stores in flight: store drain,
// next-halo fetch and the multiply of the current patch overlap.  Per patch a CU moves 30 KB in and 49 KB out
// (+49 KB residual) for 1.8 us of MFMA work: the kernel is bound by HBM, which is the point.
typedef int i32x4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void buffer_store16(const uint4& v, unsigned voff, i32x4 srd) {
  i32x4v d;
  d[0] = (int)v.x; d[1] = (int)v.y; d[2] = (int)v.z; d[3] = (int)v.w;
  // (s_nop 1: a store of more than 64 bits reads its data registers over two cycles, and hipcc neither knows that this
  //  statement is one nor pads it -- without the nop its next instruction overwrote the first data register)
  asm volatile("buffer_store_dwordx4 %0, %1, %2, 0 offen\n\ts_nop 1" : : "v"(d), "v"(voff), "s"(srd) : "memory");
}

template <int NW, int FW, int NI>
__global__ __launch_bounds__(NW * 64) void conv_stream_kernel(ConvParams p, int PH, int PW, int tilesY, int tilesX, int G) {
  constexpr int BN = NI * 16;
  constexpr int SLAB = NW * 1024;
  constexpr int AIM = 4;                                        // halo DMAs per wave (<= 64*NW pixels)
  constexpr int BIM = (9 * BN * 4 + NW * 64 - 1) / (NW * 64);   // weight DMAs per wave
  constexpr int A_BYTES = AIM * SLAB, W_BYTES = BIM * SLAB;
  constexpr int ROWP = BN * 2 + 16;
  constexpr int STG = FW * 16 * ROWP;                           // epilogue staging per wave
  constexpr int CPR8 = BN / 8;                                  // 16-byte chunks per pixel row
  static_assert(64 % CPR8 == 0, "a lane keeps its chunk column over the write-out rounds");
  constexpr int RPF = (16 * CPR8 + 63) / 64;
  constexpr int NST = FW * RPF;                                 // stores per wave and patch (always issued)
  static_assert(NW * BN * 8 <= STG, "the statistics scratch aliases wave 0's staging tile");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
  char* const stg_base = smem + W_BYTES + 2 * A_BYTES;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nt = blockIdx.x % p.tilesN, g = blockIdx.x / p.tilesN;
  const int n0 = nt * BN;
  const int tpi = tilesY * tilesX;
  const int HW_ = PW + 2, NPIX = (PH + 2) * HW_, NOUT = PH * PW;

  const i32x4 srd0 = make_srd(p.x0, p.bytes0);
  const i32x4 srdw = make_srd(p.w, p.bytesw);
  const i32x4 srdy = make_srd(p.y, (unsigned)((size_t)p.M * p.Cout * 2));

  // halo gather: which halo pixel / chunk each of this lane's AIM DMAs fills is the same for every patch
  int hy[AIM], hx[AIM];
  unsigned hsw[AIM];
  bool hval[AIM];
#pragma unroll
  for (int j = 0; j < AIM; ++j) {
    const int chunk = (j * NW + wave) * 64 + lane;
    const int hp = chunk >> 2, pc = chunk & 3;
    divmod_small(hp < NPIX ? hp : 0, HW_, hy[j], hx[j]);
    hval[j] = hp < NPIX;
    hsw[j] = swz_any(hp, pc) * 16;
  }
  auto patch_origin = [&](int mt, int& b, int& y0, int& x0) {
    b = mt / tpi;
    const int pr = mt - b * tpi;
    const int ty = pr / tilesX, tx = pr - ty * tilesX;
    y0 = ty * PH;
    x0 = tx * PW;
  };
  auto issue_halo = [&](int mt, int stage) {
    int b, y0, x0;
    patch_origin(mt, b, y0, x0);
    const unsigned sbase = lds0 + W_BYTES + stage * A_BYTES + wave * 1024;
    [&]<int... J>(std::integer_sequence<int, J...>) {
      ((void)[&] {
        const int iy = y0 + hy[J] - p.pad_t, ix = x0 + hx[J] - p.pad_l;
        const bool ok = hval[J] && ((unsigned)iy < (unsigned)p.H) && ((unsigned)ix < (unsigned)p.W);
        const unsigned off = ok ? (unsigned)(((b * p.H + iy) * p.W + ix) * p.C0) * 2u + hsw[J] : OOB;
        dma16<J * SLAB>(off, srd0, 0u, sbase);
      }(), ...);
    }(std::make_integer_sequence<int, AIM>{});
  };
  {  // the weights of this block's channel tile: once
    const unsigned sbase = lds0 + wave * 1024;
    [&]<int... J>(std::integer_sequence<int, J...>) {
      ((void)[&] {
        const int chunk = (J * NW + wave) * 64 + lane;
        const int rb = chunk >> 2, pc = chunk & 3;     // row = tap*BN + n
        const int tap = rb / BN, nl = rb - tap * BN;
        const bool ok = (rb < 9 * BN) && (n0 + nl < p.Cout);
        const unsigned off = ok ? ((unsigned)(n0 + nl) * (unsigned)p.K + (unsigned)(tap * p.Cin)) * 2u + swz_any(rb, pc) * 16 : OOB;
        dma16<J * SLAB>(off, srdw, 0u, sbase);
      }(), ...);
    }(std::make_integer_sequence<int, BIM>{});
  }

  // fragment read state (patch geometry: the same for every patch)
  const int frow = lane & 15, fchunk = lane >> 4;
  int hp0[FW], q_of[FW], py_of[FW], px_of[FW];
#pragma unroll
  for (int t = 0; t < FW; ++t) {
    const int q = (wave + NW * t) * 16 + frow;
    const bool ok = q < NOUT;
    divmod_small(ok ? q : 0, PW, py_of[t], px_of[t]);
    hp0[t] = py_of[t] * HW_ + px_of[t];
    q_of[t] = ok ? q : -1;
  }
  unsigned wb[NI];
#pragma unroll
  for (int j = 0; j < NI; ++j) {
    const int rb = j * 16 + frow;
    wb[j] = rb * 64 + swz_any(rb, fchunk) * 16;
  }
  const int cq = lane >> 4;
  // fragment byte offsets inside a halo stage per (fragment, tap): patch geometry only, computed once (27 registers
  // instead of ~4 VALU instructions per fragment read)
  unsigned xoff[FW][9];
#pragma unroll
  for (int t = 0; t < FW; ++t)
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int hp = hp0[t] + (tap / 3) * HW_ + (tap % 3);
      xoff[t][tap] = (unsigned)(hp * 64 + swz_any(hp, fchunk) * 16);
    }
  // the epilogue's per-channel scale / shift: parked in LDS behind the staging tiles (BN x 2 floats), read per patch
  float* const ssc = reinterpret_cast<float*>(stg_base + NW * STG);
  for (int nl = tid; nl < BN; nl += NW * 64) {
    const int n = n0 + nl;
    ssc[nl] = (p.scale && n < p.Cout) ? p.scale[n] : 1.f;
    ssc[BN + nl] = (p.shift && n < p.Cout) ? p.shift[n] : 0.f;
  }
  const bool use_res = p.residual != nullptr;

  int mt = g;
  issue_halo(mt, 0);
  for (int it = 0; mt < p.tilesM; ++it, mt += G) {
    const int cur = it & 1;
    // In issue order this wave has in flight: [weights, first time] halo(it) | stores(it-1).  Retire the DMAs, leave the
    // NST stores of the previous patch in flight.
    if (it == 0)
      wait_vmcnt<0>();
    else
      wait_vmcnt<NST>();
    __builtin_amdgcn_s_barrier();   // halo(it) has landed for everyone; everyone is done with patch it-1 (halo stage, staging tiles)
    asm volatile("" ::: "memory");
    if (mt + G < p.tilesM) issue_halo(mt + G, cur ^ 1);
    const char* st = smem + W_BYTES + cur * A_BYTES;

    f32x4 acc[FW][NI];
#pragma unroll
    for (int t = 0; t < FW; ++t)
#pragma unroll
      for (int j = 0; j < NI; ++j) acc[t][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 wfr[2][NI], xfr[2][FW];
    auto load_frags = [&]<int TAP>(std::integral_constant<int, TAP>) {
      constexpr int bi = TAP & 1;
#pragma unroll
      for (int j = 0; j < NI; ++j) wfr[bi][j] = *reinterpret_cast<const bf16x8*>(smem + wb[j] + TAP * BN * 64);
#pragma unroll
      for (int t = 0; t < FW; ++t) xfr[bi][t] = *reinterpret_cast<const bf16x8*>(st + xoff[t][TAP]);
    };
    // the residual tile (accumulator layout: 4 channels of one pixel per lane and N fragment), requested NOW so that it
    // lands while the patch is multiplied; clamped addresses instead of a branch per load (hipcc serialises those)
    int b, y0, x0;
    patch_origin(mt, b, y0, x0);
    int m_of[FW];
#pragma unroll
    for (int t = 0; t < FW; ++t) m_of[t] = q_of[t] >= 0 ? (b * p.Ho + y0 + py_of[t]) * p.Wo + x0 + px_of[t] : -1;
    uint2 rres[FW][NI];
    if (use_res) {
#pragma unroll
      for (int t = 0; t < FW; ++t)
#pragma unroll
        for (int j = 0; j < NI; ++j) {
          const int n = n0 + j * 16 + cq * 4;
          const size_t off = (size_t)(m_of[t] >= 0 ? m_of[t] : 0) * p.Cout + (n < p.Cout ? n : 0);
          rres[t][j] = *reinterpret_cast<const uint2*>(p.residual + off);
        }
    }
    load_frags(std::integral_constant<int, 0>{});
    auto tap_body = [&]<int TAP>(std::integral_constant<int, TAP>) {
      constexpr int bi = TAP & 1;
      if constexpr (TAP < 8) load_frags(std::integral_constant<int, TAP + 1>{});
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int t = 0; t < FW; ++t)
#pragma unroll
        for (int j = 0; j < NI; ++j) {
#ifdef DY_PROBE
          if (p.flags & 0x10000) {
            asm volatile("" ::"v"(xfr[bi][t]), "v"(wfr[bi][j]));
            continue;
          }
#endif
          acc[t][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wfr[bi][j], xfr[bi][t], acc[t][j], 0, 0, 0);
        }
    };
    [&]<int... T>(std::integer_sequence<int, T...>) {
      (tap_body(std::integral_constant<int, T>{}), ...);
    }(std::make_integer_sequence<int, 9>{});

    // ---- epilogue of patch mt.  acc[t][j][r]: patch pixel q_of[t], channel n0 + j*16 + 4*cq + r
    if (p.flags & DISYOLO_CONV_STATS) {
      float* red = reinterpret_cast<float*>(stg_base);  // [NW][BN][2]
#pragma unroll
      for (int j = 0; j < NI; ++j) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float s1 = 0.f, s2 = 0.f;
#pragma unroll
          for (int t = 0; t < FW; ++t) {
            const float v = q_of[t] >= 0 ? acc[t][j][r] : 0.f;
            s1 += v;
            s2 += v * v;
          }
#pragma unroll
          for (int o = 1; o < 16; o <<= 1) {
            s1 += __shfl_xor(s1, o, 64);
            s2 += __shfl_xor(s2, o, 64);
          }
          if (frow == 0) {
            const int nl = j * 16 + cq * 4 + r;
            red[(wave * BN + nl) * 2 + 0] = s1;
            red[(wave * BN + nl) * 2 + 1] = s2;
          }
        }
      }
      __syncthreads();
      for (int nl = tid; nl < BN; nl += NW * 64) {
        const int n = n0 + nl;
        if (n < p.Cout) {
          float s1 = 0.f, s2 = 0.f;
#pragma unroll
          for (int w_ = 0; w_ < NW; ++w_) {
            s1 += red[(w_ * BN + nl) * 2 + 0];
            s2 += red[(w_ * BN + nl) * 2 + 1];
          }
          stats_out(p, mt, n, s1, s2);
        }
      }
      __syncthreads();   // the scratch is wave 0's staging tile
    }
    char* sw = stg_base + wave * STG;
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const float4 sc4 = *reinterpret_cast<const float4*>(ssc + j * 16 + cq * 4);
      const float4 sh4 = *reinterpret_cast<const float4*>(ssc + BN + j * 16 + cq * 4);
      const float esc[4] = {sc4.x, sc4.y, sc4.z, sc4.w}, esh[4] = {sh4.x, sh4.y, sh4.z, sh4.w};
#pragma unroll
      for (int t = 0; t < FW; ++t) {
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          v[r] = acc[t][j][r] * esc[r] + esh[r];
          if (p.flags & DISYOLO_CONV_LEAKY) v[r] = leaky(v[r], p.alpha);
        }
        if (use_res) {
          const uint2 rr = rres[t][j];
          v[0] += __builtin_bit_cast(float, rr.x << 16);
          v[1] += __builtin_bit_cast(float, rr.x & 0xffff0000u);
          v[2] += __builtin_bit_cast(float, rr.y << 16);
          v[3] += __builtin_bit_cast(float, rr.y & 0xffff0000u);
        }
        uint2 o2;
        o2.x = pack2(v[0], v[1]);
        o2.y = pack2(v[2], v[3]);
        *reinterpret_cast<uint2*>(sw + (t * 16 + frow) * ROWP + (j * 16 + cq * 4) * 2) = o2;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int t = 0; t < FW; ++t) {
#pragma unroll
      for (int r_ = 0; r_ < RPF; ++r_) {
        const int idx = r_ * 64 + lane;
        const int r16 = idx / CPR8, ch = idx % CPR8;
        const int m = __shfl(m_of[t], r16, 64);     // the pixel of row r16 of this fragment lives in lane r16
        const int n = n0 + ch * 8;
        const uint4 o = *reinterpret_cast<const uint4*>(sw + (t * 16 + r16) * ROWP + ch * 16);
        unsigned off = (m >= 0 && n < p.Cout) ? (unsigned)(((size_t)m * p.Cout + n) * 2) : OOB;
#ifdef DY_PROBE
        if (p.flags & 0x80000) off = OOB;      // timing probe: every store dropped by the range check
#endif
        buffer_store16(o, off, srdy);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Streaming 1x1 convolution for the wide, shallow layers (Cout <= 64, K = C0 + C1 <= 192, hundreds of thousands of
// pixels: conv3 / 6 / 8 / 77 / 79 / 80 / 82 and their data gradients), whose time is HBM traffic: 0.1-0.5 GFLOP per
// 30-130 MB.  The GEMM kernel spends it on per-block fixed costs (offset tables, a 2-6 step K loop that never reaches
// its pipeline's steady state, one round trip per phase); here nothing is staged through LDS and nothing is shared:
//   * every WAVE is on its own: it walks over groups of 32 pixels (two 16-pixel MFMA fragments), fetches their K
//     channels straight into B-operand fragments -- lane (pixel = lane & 15, chunk = lane >> 4) loads the 16 bytes
//     [k*32 + 8*chunk, +8) of its pixel, all loads of a group issued before the first is used -- no barrier anywhere
//     in the loop; the input is read exactly once and used by exactly one wave, so a trip through LDS would only add
//     latency;
//   * the weights (<= 24 KB) sit in LDS for the whole (persistent) block, rows padded by 16 bytes (conflict-free
//     ds_read_b128 for every K this kernel takes);
//   * fused nearest-upsample + concat: the second source's address is just another per-lane address;
//   * the epilogue (scale / shift / leaky / residual / f32 or bf16) leaves bf16 rows through a per-wave LDS tile as
//     whole 16-byte chunks; batch-norm statistics are accumulated in registers over ALL groups of the wave and leave
//     as ONE partial row per block.
// Latency is hidden by occupancy (16 waves per CU, ~10 KB of loads in flight each), not by a software pipeline.
template <int NI>
__global__ __launch_bounds__(512) void conv1x1_stream_kernel(ConvParams p, int ngroups) {
  constexpr int NW = 8, U = 2;
  constexpr int BN = NI * 16;
  constexpr int ROWP = BN * 2 + 16;
  constexpr int STG = U * 16 * ROWP;        // per-wave staging tile
  constexpr int MAXKS = 6;                  // K <= 192
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int frow = lane & 15, cq = lane >> 4;
  const int K = p.K, ks32 = K >> 5;
  const int wpitch = K * 2 + 16;
  char* const wl = smem;                                        // weights [BN][wpitch]
  char* const stg = smem + ((BN * wpitch + 15) & ~15);          // staging tiles, then the statistics scratch
  // ---- weights -> LDS (once): 16-byte chunks, rows >= Cout are zero
  for (int c = tid; c < BN * (K >> 3); c += NW * 64) {
    const int n = c / (K >> 3), kc = c - n * (K >> 3);
    uint4 v = uint4{0, 0, 0, 0};
    if (n < p.Cout) v = *reinterpret_cast<const uint4*>(p.w + (size_t)n * K + kc * 8);
    *reinterpret_cast<uint4*>(wl + n * wpitch + kc * 16) = v;
  }
  // per-channel scale / shift of this lane's 4 channels per N fragment
  float esc[NI][4], esh[NI][4];
#pragma unroll
  for (int j = 0; j < NI; ++j)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = j * 16 + cq * 4 + r;
      esc[j][r] = (p.scale && n < p.Cout) ? p.scale[n] : 1.f;
      esh[j][r] = (p.shift && n < p.Cout) ? p.shift[n] : 0.f;
    }
  __syncthreads();
  const bool f32out = p.flags & DISYOLO_CONV_OUT_F32;
  const bool do_stats = p.flags & DISYOLO_CONV_STATS;
  const bool vec_ok = (p.Cout & 7) == 0;
  const bool use_res = p.residual != nullptr;
  const int H1 = p.H >> 1, W1 = p.W >> 1;
  const int ks0 = p.C0 >> 5;                 // K steps that read source 0
  float s1[NI][4], s2[NI][4];
#pragma unroll
  for (int j = 0; j < NI; ++j)
#pragma unroll
    for (int r = 0; r < 4; ++r) s1[j][r] = s2[j][r] = 0.f;
  char* const sw = stg + wave * STG;

  for (int g = blockIdx.x * NW + wave; g < ngroups; g += gridDim.x * NW) {
    // ---- fetch: U fragments x ks32 chunks of 16 bytes per lane
    int m_of[U];
    bf16x8 xf[U][MAXKS];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int m = g * (U * 16) + u * 16 + frow;
      m_of[u] = m < p.M ? m : -1;
      const int mm = m < p.M ? m : 0;
      const bf16* a0 = p.x0 + (size_t)mm * p.C0 + cq * 8;
      const bf16* a1 = a0;
      if (p.C1 > 0) {
        int b, rem, y, x;
        divmod_small(mm, p.H * p.W, b, rem);
        divmod_small(rem, p.W, y, x);
        a1 = p.x1 + ((size_t)(b * H1 + (y >> 1)) * W1 + (x >> 1)) * p.C1 + cq * 8;
      }
#pragma unroll
      for (int k = 0; k < MAXKS; ++k)
        if (k < ks32) xf[u][k] = *reinterpret_cast<const bf16x8*>(k < ks0 ? a0 + k * 32 : a1 + (k - ks0) * 32);
    }
    uint2 rres[U][NI];
    if (use_res) {
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int j = 0; j < NI; ++j) {
          const int n = j * 16 + cq * 4;
          rres[u][j] = *reinterpret_cast<const uint2*>(p.residual + (size_t)(m_of[u] >= 0 ? m_of[u] : 0) * p.Cout + (n + 3 < p.Cout ? n : 0));
        }
    }
    // ---- multiply
    f32x4 acc[U][NI];
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int j = 0; j < NI; ++j) acc[u][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < MAXKS; ++k) {
      if (k < ks32) {
        bf16x8 wf[NI];
#pragma unroll
        for (int j = 0; j < NI; ++j) wf[j] = *reinterpret_cast<const bf16x8*>(wl + (j * 16 + frow) * wpitch + (k * 4 + cq) * 16);
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
          for (int j = 0; j < NI; ++j) acc[u][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], xf[u][k], acc[u][j], 0, 0, 0);
      }
    }
    // ---- epilogue.  acc[u][j][r]: pixel m_of[u], channel j*16 + 4*cq + r
    if (do_stats) {
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float v = m_of[u] >= 0 ? acc[u][j][r] : 0.f;
            s1[j][r] += v;
            s2[j][r] += v * v;
          }
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        const int n = j * 16 + cq * 4;
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          v[r] = acc[u][j][r] * esc[j][r] + esh[j][r];
          if (p.flags & DISYOLO_CONV_LEAKY) v[r] = leaky(v[r], p.alpha);
        }
        if (use_res && n + 3 < p.Cout) {
          const uint2 rr = rres[u][j];
          v[0] += __builtin_bit_cast(float, rr.x << 16);
          v[1] += __builtin_bit_cast(float, rr.x & 0xffff0000u);
          v[2] += __builtin_bit_cast(float, rr.y << 16);
          v[3] += __builtin_bit_cast(float, rr.y & 0xffff0000u);
        }
        if (f32out || !vec_ok) {
          if (m_of[u] >= 0) {
            const size_t off = (size_t)m_of[u] * p.Cout + n;
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (n + r < p.Cout) {
                float o = v[r];
                if (use_res && !(n + 3 < p.Cout)) o += (float)p.residual[off + r];
                if (f32out)
                  reinterpret_cast<float*>(p.y)[off + r] = o;
                else
                  reinterpret_cast<bf16*>(p.y)[off + r] = (bf16)o;
              }
          }
        } else {
          uint2 o2;
          o2.x = pack2(v[0], v[1]);
          o2.y = pack2(v[2], v[3]);
          *reinterpret_cast<uint2*>(sw + (u * 16 + frow) * ROWP + n * 2) = o2;
        }
      }
    if (!(f32out || !vec_ok)) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      constexpr int CPR8 = BN / 8;                       // 16-byte chunks per staged row
      constexpr int RPF = (U * 16 * CPR8 + 63) / 64;
      const int cpr = p.Cout >> 3;                       // chunks per real row
      bf16* yo = reinterpret_cast<bf16*>(p.y);
#pragma unroll
      for (int r_ = 0; r_ < RPF; ++r_) {
        const int idx = r_ * 64 + lane;
        const int row = idx / CPR8, ch = idx % CPR8;
        const int m = g * (U * 16) + row;
        if (idx < U * 16 * CPR8 && ch < cpr && m < p.M)
          *reinterpret_cast<uint4*>(yo + (size_t)m * p.Cout + ch * 8) = *reinterpret_cast<const uint4*>(sw + row * ROWP + ch * 16);
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();                   // the tile is free for the wave's next group
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
  }
  if (do_stats) {
    // lanes of one channel quad -> waves (fixed order through LDS) -> ONE row of partials per block
    float* red = reinterpret_cast<float*>(stg + NW * STG);       // [NW][BN][2]
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float a = s1[j][r], b2 = s2[j][r];
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) {
          a += __shfl_xor(a, o, 64);
          b2 += __shfl_xor(b2, o, 64);
        }
        if (frow == 0) {
          const int nl = j * 16 + cq * 4 + r;
          red[(wave * BN + nl) * 2 + 0] = a;
          red[(wave * BN + nl) * 2 + 1] = b2;
        }
      }
    __syncthreads();
    for (int nl = tid; nl < BN; nl += NW * 64) {
      if (nl < p.Cout) {
        float a = 0.f, b2 = 0.f;
#pragma unroll
        for (int w_ = 0; w_ < NW; ++w_) {
          a += red[(w_ * BN + nl) * 2 + 0];
          b2 += red[(w_ * BN + nl) * 2 + 1];
        }
        stats_out(p, blockIdx.x, nl, a, b2);
      }
    }
  }
}

// patch of the halo kernel for an H x W image: PH | H, PW | W, at most max_frags*16 output
// pixels and max_halo halo pixels; the largest area wins, then the squarest.  0 = none fits.
struct Patch {
  int ph, pw;
};
Patch pick_patch(int H, int W, int max_frags, int max_halo) {
  Patch best{0, 0};
  int best_area = 0, best_halo = 1 << 30;
  for (int ph = 1; ph <= H; ++ph) {
    if (H % ph) continue;
    for (int pw = 1; pw <= W; ++pw) {
      if (W % pw) continue;
      const int area = ph * pw, halo = (ph + 2) * (pw + 2);
      if ((area + 15) / 16 > max_frags || halo > max_halo) continue;
      if (area > best_area || (area == best_area && halo < best_halo)) {
        best = Patch{ph, pw};
        best_area = area;
        best_halo = halo;
      }
    }
  }
  return best;
}
// halo tile ids: 16 = 8 waves x 3 fragments (up to 384 pixels: 18x18), 17 = 4 waves x 3
// (up to 192 pixels: 9x18), both 64 output channels per block; 18 = as 16 with 32 output channels per block
// (twice the blocks: the 18x18 maps, one patch per image, then give 256 blocks at 1024 channels); 19 = 16 output channels per
// block (256 blocks at 512 channels: the data gradient of the 18^2 512 -> 1024 layers)
bool halo_cfg(int id, int& nw, int& fw) {
  if (id == 16 || id == 18 || id == 19) { nw = 8; fw = 3; return true; }
  if (id == 17) { nw = 4; fw = 3; return true; }
  return false;
}
int halo_bn(int id) { return id == 18 ? 32 : id == 19 ? 16 : 64; }
bool halo_ok(const disyolo_conv_desc* d, int id, Patch* out) {
  int nw, fw;
  if (!halo_cfg(id, nw, fw)) return false;
  if (d->ksize != 3 || d->stride != 1 || d->in_div != 1 || d->C1 != 0 || d->C0 % 32 != 0) return false;
  if (d->Ho != d->H || d->Wo != d->W) return false;
  const Patch pt = pick_patch(d->H, d->W, nw * fw, 64 * nw);
  if (pt.ph == 0 || pt.ph * pt.pw < 64) return false;
  if (out) *out = pt;
  return true;
}
// tile id 20: the streaming kernel (3x3 stride 1, exactly 32 input channels, bf16 output with whole 16-byte channel
// chunks, no fused batch-norm backward sums)
bool stream_ok(const disyolo_conv_desc* d, Patch* out) {
  if (d->ksize != 3 || d->stride != 1 || d->in_div != 1 || d->C1 != 0 || d->C0 != 32) return false;
  if (d->Ho != d->H || d->Wo != d->W || d->pad_t != 1 || d->pad_l != 1) return false;
  if ((d->flags & (DISYOLO_CONV_OUT_F32 | DISYOLO_CONV_BN_BWD_STATS)) || d->Cout % 8) return false;
  if ((int64_t)d->B * d->Ho * d->Wo * d->Cout * 2 >= (1LL << 31)) return false;      // 32-bit store offsets
  const Patch pt = pick_patch(d->H, d->W, 24, 512);
  if (pt.ph == 0 || pt.ph * pt.pw < 64) return false;
  if (out) *out = pt;
  return true;
}
int launch_stream(const ConvParams& p, Patch pt, hipStream_t s) {
  constexpr int NW = 8, FW = 3, NI = 4, BN = 64;
  ConvParams q = p;
  const int tilesY = p.H / pt.ph, tilesX = p.W / pt.pw;
  q.tilesM = p.B * tilesY * tilesX;
  q.tilesN = ceil_div(p.Cout, BN);
  constexpr int SLAB = NW * 1024, BIM = (9 * BN * 4 + NW * 64 - 1) / (NW * 64);
  const size_t lds = (size_t)BIM * SLAB + 2 * 4 * SLAB + (size_t)NW * FW * 16 * (BN * 2 + 16) + BN * 8;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_stream_kernel<NW, FW, NI>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  static const int ncu = [] {
    int dev = 0, n = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
    return n > 0 ? n : 256;
  }();
  int G = ncu / q.tilesN;          // one persistent block per CU (the LDS holds one)
  if (G < 1) G = 1;
  if (G > q.tilesM) G = q.tilesM;
  hipLaunchKernelGGL((conv_stream_kernel<NW, FW, NI>), dim3(G * q.tilesN), dim3(NW * 64), lds, s, q, pt.ph, pt.pw, tilesY,
                     tilesX, G);
  DY_CHECK_LAUNCH();
  return DISYOLO_OK;
}

// tile id 21: the streaming 1x1 kernel (stride 1, Cout <= 64, K = C0 + C1 <= 192; any epilogue but the fused BN
// backward sums).  Its grid -- and with it the number of statistics rows -- depends on the pixel count only.
bool stream1x1_ok(const disyolo_conv_desc* d) {
  if (d->ksize != 1 || d->stride != 1 || d->in_div != 1 || d->Ho != d->H || d->Wo != d->W) return false;
  if (d->Cout > 64 || d->C0 % 32 || d->C1 % 32 || d->C0 + d->C1 > 192) return false;
  if (d->flags & DISYOLO_CONV_BN_BWD_STATS) return false;
  return true;
}
int stream1x1_blocks(int M) {
  const int groups = ceil_div(M, 32);
  int blocks = ceil_div(groups, 8 * 4);       // >= 4 groups per wave
  if (blocks > 512) blocks = 512;             // two persistent blocks per CU
  if (blocks < 1) blocks = 1;
  return blocks;
}
template <int NI>
int launch_stream1x1_n(const ConvParams& p, hipStream_t s) {
  constexpr int BN = NI * 16;
  const size_t wbytes = ((size_t)BN * (p.K * 2 + 16) + 15) & ~(size_t)15;
  const size_t lds = wbytes + (size_t)8 * 2 * 16 * (BN * 2 + 16) + (size_t)8 * BN * 8;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv1x1_stream_kernel<NI>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)(64 * (192 * 2 + 16) + 8 * 2 * 16 * (64 * 2 + 16) + 8 * 64 * 8));
    attr_set = true;
  }
  hipLaunchKernelGGL((conv1x1_stream_kernel<NI>), dim3(stream1x1_blocks(p.M)), dim3(512), lds, s, p, ceil_div(p.M, 32));
  DY_CHECK_LAUNCH();
  return DISYOLO_OK;
}
int launch_stream1x1(const ConvParams& p, hipStream_t s) {
  if (p.Cout <= 16) return launch_stream1x1_n<1>(p, s);
  if (p.Cout <= 32) return launch_stream1x1_n<2>(p, s);
  return launch_stream1x1_n<4>(p, s);
}

// Residency query (disyolo_conv2d_bn_fused_ok): the launchers below fill this instead of launching -- the grid, the statistics
// rows, the channel tile and how many blocks of THIS instance the device holds at once (the cluster exchange of the fused
// batch-norm epilogues needs every block of the launch resident: a block waits for the rows of blocks that may not have
// started yet).
struct LaunchQuery {
  bool active = false, have = false, want_bwd = false, want_fused = false;
  int grid = 0, rows = 0, bn = 0, resident = 0;
};
thread_local LaunchQuery tl_query;
template <class K>
int resident_blocks(K kernel, int threads, size_t lds) {
  int dev = 0, cus = 0, occ = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 0;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, reinterpret_cast<const void*>(kernel), threads, lds) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  // the occupancy API can answer one block per CU high (MI355X_MICROARCH.md "Correctness boundaries"): cap it by what LDS
  // and the wave slots allow, whatever it says
  const int by_lds = lds ? (int)((size_t)160 * 1024 / lds) : 8;
  const int by_waves = 32 / (threads / 64);
  if (occ > by_lds) occ = by_lds;
  if (occ > by_waves) occ = by_waves;
  return occ * cus;
}

template <int NW, int FW, int NI>
int launch_halo(const ConvParams& p, Patch pt, hipStream_t s) {
  ConvParams q = p;
  constexpr int BN = NI * 16;
  const int tilesY = p.H / pt.ph, tilesX = p.W / pt.pw;
  q.tilesM = p.B * tilesY * tilesX;
  q.tilesN = ceil_div(p.Cout, BN);
  {
    static const bool on = [] { const char* e = getenv("DISYOLO_XCD_N"); return !(e && e[0] == '0'); }();
    q.xcd_n = (on && q.tilesN >= 8 && (int64_t)p.bytesw > (int64_t)p.bytes0) ? 1 : 0;
    static const bool split = [] { const char* e = getenv("DISYOLO_HALO_SPLIT_ROWS"); return !(e && e[0] == '0'); }();
    q.halo_split = split ? 1 : 0;
  }
  constexpr int SLAB = NW * 1024, BIM = (9 * BN * 4 + NW * 64 - 1) / (NW * 64);
  // two stages (compute slice c while slice c+1 lands); a one-slice layer (32 input channels) uses the first only.
  // The epilogue's scratch (stats rows + per-wave staging tiles) must fit as well.
  const size_t epi = (size_t)NW * BN * 8 + (size_t)NW * FW * 16 * (BN * 2 + 16) + (size_t)BN * 8 + 16 + (size_t)NW * 64 * 16;   // (+ the fused batch norm's coefficients and f64 scratch)
  size_t lds = (size_t)(p.Cin > 32 ? 2 : 1) * (4 + BIM) * SLAB;
  if (lds < epi) lds = epi;
  // (a launch whose blocks wait for each other takes whole CUs: see launch_ks)
  if (((p.flags & (DISYOLO_CONV_BN_FUSED | DISYOLO_CONV_BN_BWD_FUSED)) || (tl_query.active && tl_query.want_fused)) && lds < (size_t)81 * 1024)
    lds = (size_t)81 * 1024;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_halo_kernel<NW, FW, NI>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)((size_t)2 * (4 + BIM) * SLAB));
    attr_set = true;
  }
  if (tl_query.active) {
    tl_query.have = true;
    tl_query.grid = q.tilesM * q.tilesN;
    tl_query.rows = q.tilesM;
    tl_query.bn = BN;
    tl_query.resident = resident_blocks(&conv_halo_kernel<NW, FW, NI>, NW * 64, lds);
    return DISYOLO_OK;
  }
  hipLaunchKernelGGL((conv_halo_kernel<NW, FW, NI>), dim3(q.tilesM * q.tilesN), dim3(NW * 64), lds, s, q, pt.ph,
                     pt.pw, tilesY, tilesX);
  DY_CHECK_LAUNCH();
  return DISYOLO_OK;
}


struct TileCfg {
  int id, bm, bn;
};
// id -> block tile (pixels x channels); the wave grid, K depth and pipeline stages per
// variant are in dispatch()
const TileCfg kTiles[] = {
    {1, 128, 128}, {2, 128, 64}, {3, 64, 128}, {4, 128, 32}, {5, 128, 16}, {6, 64, 64}, {7, 256, 64}, {8, 256, 128},
    {9, 64, 128}, {10, 96, 128}, {11, 96, 128}, {12, 192, 128}, {13, 64, 128}, {14, 96, 128}, {15, 192, 128},
    // round 5: two-wave tiles for the deep 1x1 layers whose 64x64 grid is barely one block per CU (18^2: 328 blocks)
    {26, 32, 64}, {27, 64, 32}, {28, 32, 128},
    // 8 waves of 96x64: 0.42 fragment reads per MFMA against 0.58 for the 48x64 wave tiles (the loop is LDS-read bound)
    {29, 384, 128},
};
// ids 1..15 and 26.. are GEMM tiles of conv_igemm_kernel; 16..25 the patch / streaming / flat-frame kernels
inline bool is_gemm_tile(int id) { return id < 16 || id >= 26; }

template <int BM, int BN, int WM, int WN, int BK, int ST, int KS, int KG = 1>
int launch_ks(const ConvParams& p, hipStream_t s) {
  ConvParams q = p;
  q.tilesM = ceil_div(p.M, BM);
  q.tilesN = ceil_div(p.Cout, BN);
  q.nk = p.K / BK;
  q.pcls = 0; q.Mc = 0; q.tilesMc = 0;
  if (KS == 3 && KG == 1 && p.dshift == 1 && !(p.Ho & 1) && !(p.Wo & 1) && !(p.flags & DISYOLO_CONV_STATS)) {
    static const bool on = [] { const char* e = getenv("DISYOLO_DGRAD_PCLS"); return !(e && e[0] == '0'); }();
    if (on) {
      q.pcls = 1;
      q.Mc = p.B * (p.Ho >> 1) * (p.Wo >> 1);
      q.tilesMc = ceil_div(q.Mc, BM);
      q.tilesM = 4 * q.tilesMc;
    }
  }
  {
    static const bool on = [] { const char* e = getenv("DISYOLO_XCD_N"); return !(e && e[0] == '0'); }();
    q.xcd_n = (on && q.tilesN >= 8 && (int64_t)p.bytesw > (int64_t)p.bytes0 + (int64_t)p.bytes1) ? 1 : 0;
  }
  const int grid = q.tilesM * q.tilesN;
  constexpr int NW = WM * WN, SLAB = 1024 * NW, ROWB = BK * 2;
  constexpr int A_BYTES = (BM * ROWB + SLAB - 1) / SLAB * SLAB, B_BYTES = (BN * ROWB + SLAB - 1) / SLAB * SLAB;
  size_t lds = (size_t)KG * ST * (A_BYTES + B_BYTES);
  const size_t red = (size_t)WM * BN * 2 * sizeof(float);
  if (red > lds) lds = red;
  const size_t xg = (size_t)(KG - 1) * BM * BN * sizeof(float);
  if (xg > lds) lds = xg;
  const size_t stg = red + (size_t)BN * 8 + (size_t)NW * (BM / WM) * ((BN / WN) * 2 + 16)   // scale/shift + epilogue staging behind the stats scratch
                     + 16 + (size_t)NW * 64 * 16;                                              // + the fused batch norm's f64 scratch
  if (stg > lds) lds = stg;
  static bool attr_set = false;
  if (!attr_set && (lds > 64 * 1024 || KG == 1)) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_igemm_kernel<BM, BN, WM, WN, BK, ST, KS, KG>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)(lds > (size_t)81 * 1024 ? lds : (size_t)81 * 1024));
    attr_set = true;
  }
  // tiles with an EPI = 1 instance (the fused batch-norm backward): what the data-gradient convs of the 18^2 / 36^2 maps run
  constexpr bool HAS_BWD = KG == 1 && ((KS == 1 && ((BM == 64 && BN == 64) || (BM == 64 && BN == 128) || (BM == 128 && BN == 64) ||
                                                    (BM == 96 && BN == 128) || (BM == 128 && BN == 128))) ||
                                       (BM == 192 && BN == 128));
  // A launch whose blocks wait for each other (the fused batch-norm epilogues) takes whole CUs: its LDS request is raised
  // past half a CU's so that no two of its blocks share one, and disyolo_conv2d_bn_fused_ok admits it only with a grid of at
  // most one block per CU.  Why: with several small blocks per CU a partly resident grid sits on EVERY CU; a kernel of
  // another queue whose blocks then fit nowhere (the side lane's weight gradients: one wave per SIMD with most of its
  // registers) stalls the workgroup dispatcher, and the rest of this grid is never placed -- measured: the 64x64-tile data
  // gradients (656 blocks, three per CU) hung until the bounded wait gave up whenever replays ran back to back
  // (profiles/r06_bn_inkernel.txt).  With one block per CU a CU either holds a block of this launch or is free for the
  // other queue: the other queue always makes progress or this grid is fully resident.
  const bool cluster = (p.flags & (DISYOLO_CONV_BN_FUSED | DISYOLO_CONV_BN_BWD_FUSED)) != 0 || (tl_query.active && tl_query.want_fused);
  constexpr size_t EXCL_LDS = (size_t)81 * 1024;
  if (cluster && lds < EXCL_LDS) lds = EXCL_LDS;
  if (tl_query.active) {
    static const bool bwd_gemm_on = [] { const char* e = getenv("DISYOLO_BN_INKERNEL_BWD_GEMM"); return !(e && e[0] == '0'); }();
    static const bool fwd_gemm_on = [] { const char* e = getenv("DISYOLO_BN_INKERNEL_FWD_GEMM"); return !(e && e[0] == '0'); }();
    tl_query.have = KG == 1 && q.pcls == 0 && (tl_query.want_bwd ? (HAS_BWD && bwd_gemm_on) : fwd_gemm_on);
    tl_query.grid = grid;
    tl_query.rows = q.tilesM;
    tl_query.bn = BN;
    if constexpr (HAS_BWD) {
      if (tl_query.want_bwd) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_igemm_kernel<BM, BN, WM, WN, BK, ST, KS, KG, 1>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)(lds > EXCL_LDS ? lds : EXCL_LDS));
        tl_query.resident = resident_blocks(&conv_igemm_kernel<BM, BN, WM, WN, BK, ST, KS, KG, 1>, NW * 64 * KG, lds);
        return DISYOLO_OK;
      }
    }
    tl_query.resident = resident_blocks(&conv_igemm_kernel<BM, BN, WM, WN, BK, ST, KS, KG>, NW * 64 * KG, lds);
    return DISYOLO_OK;
  }
  if (p.flags & (DISYOLO_CONV_BN_BWD_FUSED | DISYOLO_CONV_BN_BWD_STATS)) {
    if constexpr (HAS_BWD) {
      DY_REQUIRE(q.pcls == 0 && p.d2s_c == 0, "conv: BN_BWD_STATS / BN_BWD_FUSED on a parity-class or depth-to-space data gradient");
      static bool attr1 = false;
      if (!attr1) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_igemm_kernel<BM, BN, WM, WN, BK, ST, KS, KG, 1>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)(lds > EXCL_LDS ? lds : EXCL_LDS));
        attr1 = true;
      }
      hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, WM, WN, BK, ST, KS, KG, 1>), dim3(grid), dim3(NW * 64 * KG), lds, s, q);
      DY_CHECK_LAUNCH();
      return DISYOLO_OK;
    } else {
      disyolo_set_error("conv: BN_BWD_STATS / BN_BWD_FUSED on a GEMM tile without that epilogue (%dx%d, k %d; ask disyolo_conv2d_bn_bwd_stats_ok / "
                        "disyolo_conv2d_bn_fused_ok)", BM, BN, KS);
      return DISYOLO_E_ARG;
    }
  }
  hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, WM, WN, BK, ST, KS, KG>), dim3(grid), dim3(NW * 64 * KG), lds, s, q);
  DY_CHECK_LAUNCH();
  return DISYOLO_OK;
}
template <int BM, int BN, int WM, int WN, int BK, int ST, int KG = 1>
int launch(const ConvParams& p, hipStream_t s) {
  return p.ks == 3 ? launch_ks<BM, BN, WM, WN, BK, ST, 3, KG>(p, s) : launch_ks<BM, BN, WM, WN, BK, ST, 1, KG>(p, s);
}

// BK = 64 needs every source's channel count to be a multiple of 64 (a K slice never
// straddles a filter tap or the concat boundary).  variant: 0 = default pipeline depth,
// 1 = alternative depth (tuning)
#define DY_TILE(ID, BM, BN, WM, WN, S64A, S64B, S32A, S32B)                                          \
  case ID:                                                                                          \
    if (bk64) return variant ? launch<BM, BN, WM, WN, 64, S64B>(p, s) : launch<BM, BN, WM, WN, 64, S64A>(p, s); \
    return variant ? launch<BM, BN, WM, WN, 32, S32B>(p, s) : launch<BM, BN, WM, WN, 32, S32A>(p, s);
// split-K tiles (two K groups per block) need BK = 64 and an even number of K slices;
// otherwise the plain tile of the same shape runs
int resolve_tile(int id, bool bk64, int K) {
  if (id >= 13 && id <= 15) {
    const bool ok = bk64 && (K / 64) % 2 == 0 && K >= 512;
    if (!ok) id = id == 13 ? 3 : id == 14 ? 10 : 12;
  }
  return id;
}
int dispatch(int id, bool bk64, int variant, const ConvParams& p, hipStream_t s) {
  id = resolve_tile(id, bk64, p.K);
#ifdef DY_ONLY_HALO   // kernel-probe builds (tools/build_probe.sh): only the patch kernels are instantiated
  disyolo_set_error("conv: probe build without the GEMM tiles (tile id %d)", id);
  return DISYOLO_E_ARG;
#else
  switch (id) {
    case 13: return variant ? launch<64, 128, 2, 2, 64, 3, 2>(p, s) : launch<64, 128, 2, 2, 64, 2, 2>(p, s);
    case 14: return launch<96, 128, 2, 2, 64, 2, 2>(p, s);   // 3 stages x 2 groups would need 168 KiB
    case 15: return launch<192, 128, 4, 2, 64, 2, 2>(p, s);
    DY_TILE(1, 128, 128, 2, 2, 2, 3, 3, 4)
    DY_TILE(2, 128, 64, 2, 2, 3, 2, 4, 3)
    DY_TILE(3, 64, 128, 2, 2, 2, 3, 4, 3)
    DY_TILE(4, 128, 32, 4, 1, 3, 2, 4, 3)
    DY_TILE(5, 128, 16, 4, 1, 3, 2, 4, 3)
    DY_TILE(6, 64, 64, 2, 2, 3, 2, 4, 3)
    DY_TILE(7, 256, 64, 4, 1, 3, 2, 4, 3)
    DY_TILE(8, 256, 128, 4, 2, 2, 3, 3, 4)
    DY_TILE(9, 64, 128, 2, 2, 6, 4, 6, 4)   // deep pipeline for layers with fewer blocks than CU slots
    DY_TILE(10, 96, 128, 2, 2, 2, 3, 4, 3)  // 48x64 wave tiles: fewer bytes staged per FLOP than 64x128
    DY_TILE(11, 96, 128, 2, 2, 4, 5, 6, 4)  // the same, deep pipeline (one block per CU)
    DY_TILE(12, 192, 128, 4, 2, 2, 3, 3, 4) // 8 waves of 48x64
    DY_TILE(26, 32, 64, 1, 2, 3, 4, 4, 6)
    DY_TILE(27, 64, 32, 2, 1, 3, 4, 4, 6)
    DY_TILE(28, 32, 128, 1, 2, 3, 4, 4, 6)
    DY_TILE(29, 384, 128, 4, 2, 2, 2, 3, 4)
    default: disyolo_set_error("conv: unknown tile id %d", id); return DISYOLO_E_ARG;
  }
#endif
}

// Launcher heuristic = the consensus of the in-sequence autotuner (YOLONet.autotune, which
// times every candidate where it runs: inside the step, operands as cold as they really are)
// over the layer shapes of the B = 8, 576^2 network; profiles/archive/r01h_autotune.txt.  Stand-alone
// timing loops (tools/bench_conv.py) keep a layer's operands hot in L2 and rank the tiles
// differently -- rules taken from them measured 1-3 % SLOWER end to end.  What the picks say:
// huge-M layers want the 8-wave 192x128 tile (48x64 wave tiles: fewest bytes staged per
// FLOP) as long as it still yields > 1 block per CU; everything else wants MANY small blocks
// (64x64 / 128x64) -- occupancy and short epilogues beat staging efficiency when a layer is
// only ~10 us of MFMA work.  Returns id | flags: bit 8 = force BK 32, bit 9 = alternative
// pipeline depth.  A caller that tunes for its own shapes passes the result in d->tile.
int pick_auto(const disyolo_conv_desc* d, int M) {
  const int N = d->Cout;
  const int K = d->ksize * d->ksize * (d->C0 + d->C1);
  const bool k3 = d->ksize == 3;
  // 3x3 stride-1 layers with >= 128 input channels: the patch kernel (halo staged once per 32
  // channels instead of once per tap) wins when its grid is one full round of the 256 CUs, or
  // when the layer is narrow (N <= 64) and the GEMM tiles are bound by staging
  if (k3 && d->C0 >= 128) {
    Patch pt;
    if (halo_ok(d, 16, &pt)) {
      const int blocks = d->B * (d->H / pt.ph) * (d->W / pt.pw) * ceil_div(N, 64);
      if ((blocks >= 192 && blocks <= 256) || (N <= 64 && M >= 40000)) return 16;
    }
  }
  // narrow layers (the HBM-bound ends of the network): small tiles with the shallow pipeline,
  // i.e. the smallest LDS footprint and the most blocks per CU, win by 20-80 %
  if (N <= 32) return 4 | 0x200;
  if (N <= 64) return (k3 ? 2 : 6) | 0x200;
  if (M < 4096) return (k3 && N >= 1024) ? (3 | 0x200) : 6;
  if (M >= 40000 && N % 128 == 0)
    return (K >= 128 && ceil_div(M, 192) * (N / 128) >= 300) ? 12 : (2 | 0x200);
  if (!k3 && M < 20000) return N >= 512 ? 3 : 6;
  return 3;
}

int pick_tile(const disyolo_conv_desc* d, int M) {
  if (d->tile > 0) return d->tile;
  return pick_auto(d, M);
}

int tile_bm(int id) {
  for (const TileCfg& t : kTiles)
    if (t.id == id) return t.bm;
  return 0;
}

int validate(const disyolo_conv_desc* d) {
  DY_REQUIRE(d != nullptr, "conv: null descriptor");
  DY_REQUIRE(d->ksize == 1 || d->ksize == 3, "conv: ksize %d unsupported", d->ksize);
  DY_REQUIRE(d->stride == 1 || d->stride == 2, "conv: stride %d unsupported", d->stride);
  DY_REQUIRE(d->C0 > 0 && d->C0 % 32 == 0, "conv: C0=%d must be a positive multiple of 32", d->C0);
  DY_REQUIRE(d->C1 >= 0 && d->C1 % 32 == 0, "conv: C1=%d must be a multiple of 32", d->C1);
  DY_REQUIRE(d->C1 == 0 || (d->ksize == 1 && d->stride == 1 && d->x1 != nullptr && (d->H % 2 == 0) && (d->W % 2 == 0)),
             "conv: fused upsample+concat needs a 1x1 stride-1 conv with even H,W");
  DY_REQUIRE(d->in_div == 1 || d->in_div == 2, "conv: in_div must be 1 or 2");
  DY_REQUIRE(d->B > 0 && d->H > 0 && d->W > 0 && d->Ho > 0 && d->Wo > 0 && d->Cout > 0, "conv: bad sizes");
  DY_REQUIRE((int64_t)d->B * d->Ho * d->Wo < (1LL << 31), "conv: too many output pixels");
  DY_REQUIRE(d->x0 && d->w && d->y, "conv: null tensor pointer");
  DY_REQUIRE((int64_t)d->B * d->H * d->W * d->C0 * 2 < (1LL << 31) &&
                 (int64_t)d->Cout * d->ksize * d->ksize * (d->C0 + d->C1) * 2 < (1LL << 31),
             "conv: a source tensor or the packed weights exceed the 2 GiB the 32-bit gather offsets address");
  return DISYOLO_OK;
}

}  // namespace

extern "C" size_t disyolo_conv_desc_size(void) { return sizeof(disyolo_conv_desc); }

// The tile code that will actually run: d->tile (or the heuristic) with every "this id does not cover the shape"
// fallback applied -- ONE place, so that the launcher, the statistics-row count and the reported tile cannot disagree.
// ids >= 16 return their patch in *pt.
static int resolve_sel(const disyolo_conv_desc* d, int M, Patch* pt) {
  Patch tmp;
  if (!pt) pt = &tmp;
  int sel = pick_tile(d, M);
  for (int guard = 0; guard < 3; ++guard) {
    const int id = sel & 0xff;
    if (id == 24 || id == 25) {
      if (flat_ok(d, id, nullptr)) return sel;
    } else if (id == 21) {
      if (stream1x1_ok(d)) return sel;
    } else if (id == 20) {
      if (stream_ok(d, pt)) return sel;
    } else if (!is_gemm_tile(id)) {
      if (halo_ok(d, id, pt)) return sel;
    } else {
      return sel;
    }
    sel = pick_auto(d, M);     // (returns a patch tile only where it covers the shape)
  }
  return sel;
}

extern "C" int disyolo_conv2d_bn_bwd_stats_ok(const disyolo_conv_desc* d) {
  if (!d || (d->flags & DISYOLO_CONV_OUT_F32) || d->Cout % 8) return 0;
  const int sel = resolve_sel(d, d->B * d->Ho * d->Wo, nullptr);
  int id = sel & 0xff;
  if ((id >= 16 && id < 20) || id == 24 || id == 25) return 1;
  if (!is_gemm_tile(id) || d->in_div != 1) return 0;
  // round 6: the GEMM tiles that have an EPI = 1 instance (launch_ks: HAS_BWD) can emit the rows too, one per pixel tile --
  // opt-in (DISYOLO_BN_BWD_STATS_GEMM=1): measured SLOWER in the step than the column reduction it replaces (19 -> 5 colreduce
  // launches per stage-1 step, 3.63 -> 3.69 ms; stage 2 9.49 -> 9.58 ms; profiles/r06_bn_inkernel.txt): the epilogue's extra
  // registers and its x-tile read at the end of every block cost the convs more than the HBM-bound sweeps cost beside them
  {
    const char* e = getenv("DISYOLO_BN_BWD_STATS_GEMM");
    if (!(e && e[0] == '1')) return 0;
  }
  const bool bk64 = (d->C0 % 64 == 0) && (d->C1 % 64 == 0) && !(sel & 0x100);
  id = resolve_tile(id, bk64, d->ksize * d->ksize * (d->C0 + d->C1));
  if (id == 12) return 1;                                                                  // 192x128: 1x1 and 3x3
  return d->ksize == 1 && (id == 1 || id == 2 || id == 3 || id == 6 || id == 9 || id == 10 || id == 11) ? 1 : 0;
}

extern "C" int disyolo_conv2d_stats_rows(const disyolo_conv_desc* d) {
  if (!d) return DISYOLO_E_ARG;
  const int M = d->B * d->Ho * d->Wo;
  Patch pt;
  const int id = resolve_sel(d, M, &pt) & 0xff;
  if (id == 24 || id == 25) {
    FlatGeom g;
    return flat_ok(d, id, &g) ? g.tilesM : DISYOLO_E_ARG;
  }
  if (id == 21) return stream1x1_blocks(M);
  if (!is_gemm_tile(id)) return d->B * (d->H / pt.ph) * (d->W / pt.pw);
  const int bm = tile_bm(id);
  if (bm == 0) return DISYOLO_E_ARG;
  return ceil_div(M, bm);
}

// stages per (tile id, BK, variant): keep in sync with DY_TILE in dispatch()
static int tile_stages(int id, bool bk64, int variant) {
  if (id == 29) {
    static const int big[4] = {2, 2, 3, 4};
    return big[(bk64 ? 0 : 2) + (variant ? 1 : 0)];
  }
  if (id >= 26) {
    static const int two_wave[4] = {3, 4, 4, 6};
    return two_wave[(bk64 ? 0 : 2) + (variant ? 1 : 0)];
  }
  static const int tab[16][4] = {{0, 0, 0, 0}, {2, 3, 3, 4}, {3, 2, 4, 3}, {2, 3, 4, 3}, {3, 2, 4, 3},
                                 {3, 2, 4, 3}, {3, 2, 4, 3}, {3, 2, 4, 3}, {2, 3, 3, 4}, {6, 4, 6, 4},
                                 {2, 3, 4, 3}, {4, 5, 6, 4}, {2, 3, 3, 4},
                                 {2, 3, 2, 3}, {2, 2, 2, 2}, {2, 2, 2, 2}};
  return tab[id][(bk64 ? 0 : 2) + (variant ? 1 : 0)];
}

extern "C" int disyolo_conv2d_tile(const disyolo_conv_desc* d, int* bm, int* bn, int* bk, int* stages) {
  if (!d) return DISYOLO_E_ARG;
  Patch pt;
  const int sel = resolve_sel(d, d->B * d->Ho * d->Wo, &pt);
  if ((sel & 0xff) == 24 || (sel & 0xff) == 25) {
    FlatGeom g;
    if (!flat_ok(d, sel & 0xff, &g)) return DISYOLO_E_ARG;
    if (bm) *bm = g.bm;
    if (bn) *bn = g.bn;
    if (bk) *bk = 32;
    if (stages) *stages = g.stages;
    return sel & 0xff;
  }
  if ((sel & 0xff) == 21) {
    if (bm) *bm = 32;
    if (bn) *bn = d->Cout <= 16 ? 16 : (d->Cout <= 32 ? 32 : 64);
    if (bk) *bk = 32;
    if (stages) *stages = 1;
    return 21;
  }
  if (!is_gemm_tile(sel & 0xff)) {
    if (bm) *bm = pt.ph * pt.pw;
    if (bn) *bn = halo_bn(sel & 0xff);
    if (bk) *bk = 32;
    if (stages) *stages = 2;
    return sel & 0xff;
  }
  const bool bk64 = (d->C0 % 64 == 0) && (d->C1 % 64 == 0) && !(sel & 0x100);
  const int id = resolve_tile(sel & 0xff, bk64, d->ksize * d->ksize * (d->C0 + d->C1));
  for (const TileCfg& t : kTiles)
    if (t.id == id) {
      if (bm) *bm = t.bm;
      if (bn) *bn = t.bn;
      if (bk) *bk = bk64 ? 64 : 32;
      if (stages) *stages = tile_stages(id, bk64, (sel >> 9) & 1);
      return id;
    }
  return DISYOLO_E_ARG;
}

// ---- stride-2 3x3 data gradient as ONE 2x2-tap conv over dy with a depth-to-space store ("quad") ---------------------
// dx[y, x, c] = sum over (kh, kw, ci) with y - kh and x - kw even of dy[(y - kh) / 2, (x - kw) / 2, ci] * w[kh, kw, c, ci]
// (SAME padding of an even-sized input: pad_top = pad_left = 0).  All four output-parity classes of the quad
// (2 yy + py, 2 xx + px) read dy rows yy - 1, yy and columns xx - 1, xx: a stride-1 conv over dy with the four taps
// (a, b) in {0, 1}^2 of a padded 3x3 (a = 0: row yy - 1), N = [class][c], and per class only the taps of its parity
// non-zero (9 of the 16 class x tap blocks).  Against the parity-class tiling of the generic kernel (exact FLOPs, but
// 4 x M/BM tiles of 1-4 K slices each) this issues 1.75x the useful MFMAs in M/BM tiles of 4 full K slices: the shallow
// layers (conv2: 663 k quads x 32 channels, conv5) are bound by the tiles' fixed cost, not by the matrix pipe.
namespace {
__global__ void pack_quad_kernel(const float* w, bf16* wq, int C, int Cdy) {
  // wq [4 C][9 Cdy]: row (py*2 + px)*C + c, column (a*3 + b)*Cdy + ci
  const int64_t n = (int64_t)4 * C * 9 * Cdy;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int ci = (int)(i % Cdy);
    const int tap = (int)((i / Cdy) % 9);
    const int row = (int)(i / ((int64_t)9 * Cdy));
    const int c = row % C, cls = row / C, py = cls >> 1, px = cls & 1;
    const int a = tap / 3, b = tap % 3;
    // class parity 0: kh = 0 at a = 1 (dy row yy), kh = 2 at a = 0 (row yy - 1); parity 1: kh = 1 at a = 1
    const int kh = py == 0 ? (a == 1 ? 0 : (a == 0 ? 2 : -1)) : (a == 1 ? 1 : -1);
    const int kw = px == 0 ? (b == 1 ? 0 : (b == 0 ? 2 : -1)) : (b == 1 ? 1 : -1);
    float v = 0.f;
    if (kh >= 0 && kw >= 0) v = w[((size_t)(kh * 3 + kw) * C + c) * Cdy + ci];      // HWIO [3][3][C][Cdy]
    wq[i] = (bf16)v;
  }
}
}  // namespace

extern "C" int disyolo_pack_quad(const float* w_hwio, void* wq, int C, int Cdy, void* stream) {
  DY_REQUIRE(w_hwio && wq && C > 0 && Cdy > 0, "pack_quad: bad args");
  DY_RECORD_OR_RUN([=](void* s) { return disyolo_pack_quad(w_hwio, wq, C, Cdy, s); });
  const int64_t n = (int64_t)4 * C * 9 * Cdy;
  const int blocks = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
  hipLaunchKernelGGL(pack_quad_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w_hwio, (bf16*)wq, C, Cdy);
  DY_CHECK_LAUNCH();
  return DISYOLO_OK;
}

extern "C" int disyolo_dgrad_s2_quad_ok(int B, int Hdy, int Wdy, int Cdy, int C) {
  return (B > 0 && Hdy > 0 && Wdy > 0 && Cdy % 64 == 0 && C % 8 == 0 && (4 * C) % 128 == 0 && C <= 64 &&
          (int64_t)B * Hdy * Wdy * 4 * C * 2 < (1LL << 31) && (int64_t)B * Hdy * Wdy * Cdy * 2 < (1LL << 31)) ? 1 : 0;
}

extern "C" int disyolo_dgrad_s2_quad(const void* dy, const void* wq, void* dx, const void* residual, int B, int Hdy, int Wdy,
                                     int Cdy, int C, void* stream) {
  DY_REQUIRE(dy && wq && dx, "dgrad_s2_quad: null pointer");
  DY_REQUIRE(disyolo_dgrad_s2_quad_ok(B, Hdy, Wdy, Cdy, C) == 1, "dgrad_s2_quad: needs Cdy %% 64 == 0, C in {32, 64}");
  DY_RECORD_OR_RUN([=](void* s) { return disyolo_dgrad_s2_quad(dy, wq, dx, residual, B, Hdy, Wdy, Cdy, C, s); });
  ConvParams p;
  p.x0 = (const bf16*)dy; p.x1 = nullptr; p.w = (const bf16*)wq;
  p.scale = nullptr; p.shift = nullptr; p.residual = (const bf16*)residual; p.y = dx; p.stats = nullptr;
  p.bn_x = nullptr; p.bn_scale = p.bn_shift = p.bn_mean = p.bn_rstd = nullptr; p.bn_part = nullptr; p.bn_alpha = 0.f;
  p.B = B; p.H = Hdy; p.W = Wdy; p.C0 = Cdy; p.C1 = 0; p.Cin = Cdy;
  p.Ho = Hdy; p.Wo = Wdy; p.Cout = 4 * C;
  p.ks = 3; p.stride = 1; p.pad_t = 1; p.pad_l = 1; p.dmask = 0; p.dshift = 0;
  p.M = B * Hdy * Wdy;
  p.K = 9 * Cdy;
  p.bytes0 = (unsigned)((size_t)B * Hdy * Wdy * Cdy * 2);
  p.bytes1 = 0;
  p.bytesw = (unsigned)((size_t)4 * C * 9 * Cdy * 2);
  p.nk = 0;
  p.flags = 0;
  p.alpha = 0.f;
  p.tilesM = p.tilesN = 0;
  p.xcd_n = 0; p.halo_split = 0;
  p.pcls = 0; p.Mc = 0; p.tilesMc = 0;
  p.tapmask = 0x1b;           // taps (0,0) (0,1) (1,0) (1,1)
  p.d2s_c = C;
  // 192x128 tiles (8 waves of 48x64): 122 / 79 us on conv2 / conv5 at B = 8, cold caches, against 136-172 / 95-110 us for
  // the 128x128, 64x128 and 256x128 tiles (tools/bench_quad.py)
  return launch<192, 128, 4, 2, 64, 2>(p, (hipStream_t)stream);
}

static int conv2d_fwd_core(const disyolo_conv_desc* d, void* stream);

extern "C" int disyolo_cluster_sync_words(int Cout) { return Cout > 0 ? (ceil_div(Cout, 16) + 1) * CL_LINE : 0; }

extern "C" int disyolo_cluster_sync_error(const uint32_t* cluster_sync, int Cout) {
  if (!cluster_sync || Cout <= 0) return 0;
  uint32_t w = 0;
  if (hipMemcpy(&w, cluster_sync + (size_t)ceil_div(Cout, 16) * CL_LINE, sizeof(w), hipMemcpyDeviceToHost) != hipSuccess) {
    (void)hipGetLastError();
    return -1;
  }
  return (int)w;
}

// Can this descriptor run the in-launch batch norm?  With DISYOLO_CONV_BN_BWD_FUSED in its flags the answer is for the backward
// form (not every GEMM tile has that epilogue), otherwise for the forward form (whose flag need not be set).
extern "C" int disyolo_conv2d_bn_fused_ok(const disyolo_conv_desc* d) {
  if (!d || validate(d) != DISYOLO_OK) return 0;
  if ((d->flags & DISYOLO_CONV_OUT_F32) || d->Cout % 8) return 0;     // (a residual: fine for the backward form, refused by the forward one)
  if (d->in_div != 1 && (d->flags & (DISYOLO_CONV_STATS | DISYOLO_CONV_BN_FUSED))) return 0;
  static const bool on = [] { const char* e = getenv("DISYOLO_BN_INKERNEL"); return !(e && e[0] == '0'); }();
  if (!on) return 0;
  tl_query = LaunchQuery{};
  tl_query.active = true;
  tl_query.want_bwd = (d->flags & DISYOLO_CONV_BN_BWD_FUSED) != 0;
  tl_query.want_fused = true;
  disyolo_conv_desc c = *d;
  c.flags &= ~(DISYOLO_CONV_BN_FUSED | DISYOLO_CONV_BN_BWD_FUSED | DISYOLO_CONV_BN_BWD_STATS);
  const int rc = conv2d_fwd_core(&c, nullptr);
  const LaunchQuery q = tl_query;
  tl_query = LaunchQuery{};
  if (rc != DISYOLO_OK || !q.have) return 0;          // (a kernel without the epilogue: streaming / flat-frame forms, split-K tiles)
  static const int max_row_kb = [] { const char* e = getenv("DISYOLO_BN_INKERNEL_ROW_KB"); return e ? atoi(e) : 128; }();
  if ((int64_t)q.rows * q.bn * 8 > (int64_t)max_row_kb * 1024) return 0;      // every block sums all the rows of its channel tile
  return (q.resident > 0 && q.grid <= q.resident) ? 1 : 0;
}

extern "C" int disyolo_conv2d_fwd(const disyolo_conv_desc* d, void* stream) {
  int rc = validate(d);
  if (rc) return rc;
  DY_REQUIRE(!(d->flags & DISYOLO_CONV_STATS) || d->stats, "conv: STATS flag without stats buffer");
  if (d->flags & DISYOLO_CONV_BN_FUSED) {
    DY_REQUIRE((d->flags & DISYOLO_CONV_STATS) && !(d->flags & (DISYOLO_CONV_LEAKY | DISYOLO_CONV_OUT_F32 | DISYOLO_CONV_BN_BWD_STATS)) &&
                   !d->scale && !d->shift && !d->residual && d->Cout % 8 == 0,
               "conv: BN_FUSED needs STATS, bf16 y, no scale / shift / LEAKY / residual (y is the raw conv output), Cout %% 8 == 0");
    DY_REQUIRE(d->y_act && d->bn_gamma && d->bn_beta && d->bn_out_scale && d->bn_out_shift && d->bn_out_mean && d->bn_out_rstd &&
                   d->cluster_sync, "conv: BN_FUSED with a null y_act / bn_gamma / bn_beta / bn_out_* / cluster_sync");
  }
  if (d->flags & DISYOLO_CONV_BN_BWD_FUSED) {
    DY_REQUIRE((d->flags & DISYOLO_CONV_BN_BWD_STATS) && d->bn_dgamma && d->bn_dbeta && d->cluster_sync,
               "conv: BN_BWD_FUSED needs BN_BWD_STATS (the bn_* fields) and bn_dgamma / bn_dbeta / cluster_sync");
  }
  {
    const disyolo_conv_desc c = *d;
    DY_RECORD_OR_RUN([c](void* s) { return disyolo_conv2d_fwd(&c, s); });
  }
  return conv2d_fwd_core(d, stream);
}

static int conv2d_fwd_core(const disyolo_conv_desc* d, void* stream) {
  ConvParams p;
  p.x0 = (const bf16*)d->x0;
  p.x1 = (const bf16*)d->x1;
  p.w = (const bf16*)d->w;
  p.scale = d->scale;
  p.shift = d->shift;
  p.residual = (const bf16*)d->residual;
  p.y = d->y;
  p.stats = d->stats;
  p.bn_x = nullptr; p.bn_scale = p.bn_shift = p.bn_mean = p.bn_rstd = nullptr; p.bn_part = nullptr; p.bn_alpha = 0.f;
  if (d->flags & DISYOLO_CONV_BN_BWD_STATS) {
    DY_REQUIRE(d->bn_x && d->bn_scale && d->bn_shift && d->bn_mean && d->bn_rstd && d->bn_partials,
               "conv: BN_BWD_STATS flag with a null bn_* pointer");
    DY_REQUIRE((d->flags & DISYOLO_CONV_BN_BWD_FUSED) || disyolo_conv2d_bn_bwd_stats_ok(d) == 1,
               "conv: BN_BWD_STATS needs a patch kernel (tile 16-19, 24, 25 on a shape it covers) or a GEMM tile with that epilogue "
               "(ask disyolo_conv2d_bn_bwd_stats_ok), bf16 output, Cout %% 8 == 0");
    DY_REQUIRE(!(d->flags & DISYOLO_CONV_BN_BWD_FUSED) || (!(d->flags & (DISYOLO_CONV_OUT_F32 | DISYOLO_CONV_STATS | DISYOLO_CONV_LEAKY)) &&
                                                          !d->scale && !d->shift && d->Cout % 8 == 0),
               "conv: BN_BWD_FUSED needs a plain bf16 data-gradient conv (no scale / shift / LEAKY / STATS), Cout %% 8 == 0");
    p.bn_x = (const bf16*)d->bn_x; p.bn_scale = d->bn_scale; p.bn_shift = d->bn_shift; p.bn_mean = d->bn_mean;
    p.bn_rstd = d->bn_rstd; p.bn_part = d->bn_partials; p.bn_alpha = d->bn_alpha;
  }
  p.B = d->B; p.H = d->H; p.W = d->W; p.C0 = d->C0; p.C1 = d->C1; p.Cin = d->C0 + d->C1;
  p.Ho = d->Ho; p.Wo = d->Wo; p.Cout = d->Cout;
  p.ks = d->ksize; p.stride = d->stride; p.pad_t = d->pad_t; p.pad_l = d->pad_l;
  p.dmask = d->in_div - 1; p.dshift = d->in_div == 2 ? 1 : 0;
  p.M = d->B * d->Ho * d->Wo;
  p.K = d->ksize * d->ksize * p.Cin;
  p.bytes0 = (unsigned)((size_t)d->B * d->H * d->W * d->C0 * 2);
  p.bytes1 = (unsigned)((size_t)d->B * (d->H / 2) * (d->W / 2) * d->C1 * 2);
  p.bytesw = (unsigned)((size_t)d->Cout * p.K * 2);
  p.nk = 0;
  p.flags = d->flags;
  p.alpha = d->alpha;
  p.tilesM = p.tilesN = 0;
  p.xcd_n = 0; p.halo_split = 0;
  p.pcls = 0; p.Mc = 0; p.tilesMc = 0;
  p.tapmask = 0; p.d2s_c = 0;
  p.y_act = d->y_act; p.gamma = d->bn_gamma; p.beta = d->bn_beta; p.mm = d->bn_moving_mean; p.mv = d->bn_moving_var;
  p.o_scale = d->bn_out_scale; p.o_shift = d->bn_out_shift; p.o_mean = d->bn_out_mean; p.o_rstd = d->bn_out_rstd;
  p.dgamma = d->bn_dgamma; p.dbeta = d->bn_dbeta; p.csync = d->cluster_sync;
  p.bn_decay = d->bn_decay; p.bn_eps = d->bn_eps; p.inv_count = 1.0 / (double)p.M;
  hipStream_t s = (hipStream_t)stream;
  // tile field: low byte = tile id (0 = auto); bit 8 forces BK = 32, bit 9 selects the
  // alternative pipeline depth (tuning / testing)
  Patch pt;
  const int sel = resolve_sel(d, p.M, &pt);
  {
    const int id = sel & 0xff;
    if (id == 24 || id == 25 || id == 20 || id == 21) {     // kernels without the in-launch batch-norm epilogues
      if (tl_query.active) return DISYOLO_OK;               // (have stays false)
      DY_REQUIRE(!(d->flags & (DISYOLO_CONV_BN_FUSED | DISYOLO_CONV_BN_BWD_FUSED)),
                 "conv: BN_FUSED / BN_BWD_FUSED on a tile whose kernel has no such epilogue (ask disyolo_conv2d_bn_fused_ok)");
    }
  }
  switch (sel & 0xff) {
    case 24:
    case 25: return launch_flat(p, sel & 0xff, s);
#ifndef DY_ONLY_HALO
    case 21: return launch_stream1x1(p, s);
    case 20: return launch_stream(p, pt, s);
#endif
    case 16: return launch_halo<8, 3, 4>(p, pt, s);
    case 18: return launch_halo<8, 3, 2>(p, pt, s);
    case 19: return launch_halo<8, 3, 1>(p, pt, s);   // 16 channels per block: 256 blocks where Cout = 512 on one patch per image (18^2 1024 -> 512)
    case 17: return launch_halo<4, 3, 4>(p, pt, s);
    default: break;
  }
  const bool bk64 = (d->C0 % 64 == 0) && (d->C1 % 64 == 0) && !(sel & 0x100);
  return dispatch(sel & 0xff, bk64, (sel >> 9) & 1, p, s);
}
