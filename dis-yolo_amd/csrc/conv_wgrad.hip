// Weight gradient of the implicit-GEMM convolution on gfx950.
//
//   dw[(kh,kw,ci)][co] = sum_m xcol[m][(kh,kw,ci)] * dy[m][co]       (f32, HWIO order)
//
// i.e. TF autodiff of tf.nn.conv2d wrt its filter (train_yolo3_mask.py:55 minimize()).
// GEMM view: rows = K index, cols = Cout, reduction over the M = B*Ho*Wo output pixels.
// Both operands live in memory with the *reduction* index (pixel) as the slow dimension
// (NHWC), so the LDS tiles are [pixel][k] / [pixel][co] images read back with the gfx950
// transposing load ds_read_b64_tr_b16 straight into v_mfma_f32_16x16x32_bf16 fragments.
// The pixel range is split over the grid (K tile fastest, XCD-aware order); each split writes an f32 slab and a second
// kernel adds the slabs in a fixed order (deterministic, no float atomics).
#include "common.h"
#include "runtime.h"

namespace {

struct WgradParams {
  const bf16* x0;
  const bf16* x1;
  const bf16* dy;
  float* out;  // slabs [splits][K][Cout]
  int B, H, W, C0, C1, Cin;
  int Ho, Wo, Cout, ldy;
  int ks, stride, pad_t, pad_l;
  int M, K;
  int steps_per_split, steps;
  int tilesK, tilesN;
  unsigned bytes0, bytes1, bytesy;
};

typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
typedef int i32x4 __attribute__((ext_vector_type(4)));
constexpr unsigned OOB = 0x80000000u;

// 32-byte-unit swizzle inside one LDS row of `units` units: the two 16-lane groups of a
// 32-lane half of ds_read_b64_tr_b16 (rows 8g..8g+3 and 8g+8..) land on disjoint banks
__device__ __forceinline__ int swz_u(int row, int u, int units) {
  const int h = (row & 3) | (((row >> 3) & 1) << 2);
  return u ^ (h & (units - 1));
}
// LDS-DMA through a buffer descriptor: out-of-range lanes write zeros (see conv_igemm.hip)
__device__ __forceinline__ void dma16(unsigned voff, i32x4 srd, unsigned lds_dst) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds"
               :
               : "v"(voff), "s"(srd), "s"(lds_dst)
               : "memory");
}
__device__ __forceinline__ i32x4 make_srd(const void* base, unsigned bytes) {
  i32x4 r;
  r[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)base);
  r[1] = __builtin_amdgcn_readfirstlane((int)((size_t)base >> 32)) & 0xffff;
  r[2] = __builtin_amdgcn_readfirstlane((int)bytes);
  r[3] = 0x00020000;
  return r;
}
__device__ uint4 g_zero16;
__device__ __forceinline__ void dma16_flat(const void* gsrc, unsigned lds_dst) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(gsrc), "s"(lds_dst) : "memory");
}
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// Block = 128 K-rows x BN channels, 4 waves (2x2).  Per reduction step the block stages a
// [32 pixels][128 k] image of the gathered input and a [32 pixels][BN] image of dy with LDS-DMA
// (ST stages, counted vmcnt, one barrier per step) and multiplies them with transposed LDS reads.
template <int BN, int ST>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(WgradParams p) {
  constexpr int BM = 128;
  constexpr int WM = 2, WN = 2, NW = 4;
  constexpr int WTM = BM / WM, WTN = BN / WN;
  constexpr int MI = WTM / 16, NI = WTN / 16;
  constexpr int XCH = BM / 8, YCH = BN / 8;  // 16-byte chunks per pixel row
  constexpr int XP = BM * 2, YP = BN * 2;    // row pitch in bytes
  constexpr int SLAB = 1024 * NW;
  constexpr int X_BYTES = 32 * XP;                                      // 8 KiB
  constexpr int Y_BYTES = (32 * YP + SLAB - 1) / SLAB * SLAB;
  constexpr int XI = X_BYTES / SLAB, YI = Y_BYTES / SLAB;
  constexpr int LPT = XI + YI;
  constexpr int STB = X_BYTES + Y_BYTES;
  constexpr int PRE = ST - 1;
  static_assert(WTN % 16 == 0, "BN must be a multiple of 32");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  // XCD-aware block order (blocks b and b+8 share an XCD and its L2): every XCD gets a contiguous
  // run of work items with the K tile fastest, so the blocks that read the same dy tile (same
  // channel tile and pixel split, all K tiles) sit on one L2 instead of filling all eight
  int lin;
  {
    const int nblk = gridDim.x, bid = blockIdx.x;
    const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, loc = bid >> 3;
    lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
  }
  const int bkt = lin % p.tilesK, brest = lin / p.tilesK;
  const int bnt = brest % p.tilesN, bsplit = brest / p.tilesN;
  const int kt0 = bkt * BM;  // first K index of this block
  const int n0 = bnt * BN;
  const int step0 = bsplit * p.steps_per_split;
  int step1 = step0 + p.steps_per_split;
  if (step1 > p.steps) step1 = p.steps;
  const int nsteps = step1 - step0;

  const i32x4 srd0 = make_srd(p.x0, p.bytes0);
  const i32x4 srd1 = make_srd(p.x1 ? (const void*)p.x1 : (const void*)p.x0, p.x1 ? p.bytes1 : 0u);
  const i32x4 srdy = make_srd(p.dy, p.bytesy);

  // ---- X gather: DMA j of this wave fills rows (j*NW+wave)*4 .. +3 of the step's image; the
  //      lane's (tap, channel) is fixed for the block, its pixel advances by 32 per step
  int x_kh[XI], x_kw[XI], x_ci[XI], x_b[XI], x_yo[XI], x_xo[XI];
  bool x_kok[XI], x_second[XI];
#pragma unroll
  for (int j = 0; j < XI; ++j) {
    const int chunk = (j * NW + wave) * 64 + lane;
    const int row = chunk / XCH, pc = chunk % XCH;
    const int lc = swz_u(row, pc >> 1, XCH / 2) * 2 + (pc & 1);
    const int kk = kt0 + lc * 8;
    x_kok[j] = kk < p.K;
    const int kq = x_kok[j] ? kk : 0;
    const int tap = kq / p.Cin;
    int ci = kq - tap * p.Cin;
    x_second[j] = ci >= p.C0;
    x_ci[j] = x_second[j] ? ci - p.C0 : ci;
    x_kh[j] = tap / p.ks;
    x_kw[j] = tap - x_kh[j] * p.ks;
    const int m = step0 * 32 + row;
    const int hw = p.Ho * p.Wo;
    const int b = m / hw;
    const int rem = m - b * hw;
    x_b[j] = b;
    x_yo[j] = rem / p.Wo;
    x_xo[j] = rem - x_yo[j] * p.Wo;
  }
  // ---- dy: plain rows of ldy channels
  unsigned y_off[YI];
  int y_m[YI];
  bool y_nok[YI];
#pragma unroll
  for (int j = 0; j < YI; ++j) {
    const int chunk = (j * NW + wave) * 64 + lane;
    const int row = chunk / YCH, pc = chunk % YCH;
    const int lc = swz_u(row, pc >> 1, YCH / 2) * 2 + (pc & 1);
    const int n = n0 + lc * 8;
    y_nok[j] = (row < 32) && (n < p.ldy);
    y_m[j] = step0 * 32 + row;
    y_off[j] = ((unsigned)y_m[j] * (unsigned)p.ldy + (unsigned)n) * 2u;
  }

  const bool mixed = (p.C1 > 0) && ((p.C0 & (BM - 1)) != 0);
  const bool blk_second = (p.C1 > 0) && (kt0 >= p.C0);
  auto issue_tile = [&](int stage) {
    const unsigned sbase = lds0 + stage * STB + wave * 1024;
#pragma unroll
    for (int j = 0; j < XI; ++j) {
      const int iy = x_yo[j] * p.stride - p.pad_t + x_kh[j];
      const int ix = x_xo[j] * p.stride - p.pad_l + x_kw[j];
      const bool ok = x_kok[j] && (x_b[j] < p.B) && ((unsigned)iy < (unsigned)p.H) && ((unsigned)ix < (unsigned)p.W);
      unsigned off;
      if (!x_second[j]) {
        off = (unsigned)(((x_b[j] * p.H + iy) * p.W + ix) * p.C0 + x_ci[j]) * 2u;
      } else {
        const int H1 = p.H >> 1, W1 = p.W >> 1;
        off = (unsigned)(((x_b[j] * H1 + (iy >> 1)) * W1 + (ix >> 1)) * p.C1 + x_ci[j]) * 2u;
      }
      if (!mixed) {
        // the block's 128 K rows come from one source: one descriptor per block
        if (blk_second) dma16(ok ? off : OOB, srd1, sbase + j * SLAB);
        else dma16(ok ? off : OOB, srd0, sbase + j * SLAB);
      } else {
        // concat boundary inside the block's K range (C0 not a multiple of 128): per-lane
        // flat addresses, zeros from a 16-byte zero page
        const void* src = ok ? (const void*)((const char*)(x_second[j] ? p.x1 : p.x0) + off) : (const void*)&g_zero16;
        dma16_flat(src, sbase + j * SLAB);
      }
      // advance this lane's pixel by 32
      x_xo[j] += 32;
      while (x_xo[j] >= p.Wo) {
        x_xo[j] -= p.Wo;
        if (++x_yo[j] == p.Ho) {
          x_yo[j] = 0;
          ++x_b[j];
        }
      }
    }
#pragma unroll
    for (int j = 0; j < YI; ++j) {
      dma16((y_nok[j] && y_m[j] < p.M) ? y_off[j] : OOB, srdy, sbase + X_BYTES + j * SLAB);
      y_m[j] += 32;
      y_off[j] += 32u * (unsigned)p.ldy * 2u;
    }
  };

  f32x4 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

#pragma unroll
  for (int s = 0; s < PRE; ++s)
    if (s < nsteps) issue_tile(s);

  // transposed-read addressing (cdna guide T10): inside a 16-lane group, lane 4q+p supplies
  // the address of row q, columns 4p..4p+3 of a 4x16 block and receives column (lane&15).
  const int g = lane >> 4, li = lane & 15, q = li >> 2, pc = li & 3;
  for (int st = 0; st < nsteps; ++st) {
    if (PRE >= 1 && st + PRE - 1 < nsteps)
      wait_vmcnt<LPT*(PRE >= 1 ? PRE - 1 : 0)>();
    else
      wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (st + PRE < nsteps) issue_tile((st + PRE) % ST);
    const char* sX = smem + (st % ST) * STB;
    const char* sY = sX + X_BYTES;
    bf16x8 af[MI], bfr[NI];
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    const int r0 = 8 * g + q, r1 = r0 + 4;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const int u = (wm * WTM + i * 16) >> 4;
      const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
          (lds_s16x4*)(sX + r0 * XP + swz_u(r0, u, XCH / 2) * 32 + pc * 8));
      const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
          (lds_s16x4*)(sX + r1 * XP + swz_u(r1, u, XCH / 2) * 32 + pc * 8));
      const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      af[i] = __builtin_bit_cast(bf16x8, v);
    }
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int u = (wn * WTN + j * 16) >> 4;
      const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
          (lds_s16x4*)(sY + r0 * YP + swz_u(r0, u, YCH / 2) * 32 + pc * 8));
      const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
          (lds_s16x4*)(sY + r1 * YP + swz_u(r1, u, YCH / 2) * 32 + pc * 8));
      const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      bfr[j] = __builtin_bit_cast(bf16x8, v);
    }
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
  }

  // D: row (K index) = 4*(lane>>4) + r, col (co) = lane&15.  Each wave stages its WTM x WTN f32
  // tile in LDS (row pitch + 16 B) and writes it out 16 B per lane, WTN*4 contiguous bytes per K
  // row, instead of 64-byte pieces of four rows per store.
  float* slab = p.out + (size_t)bsplit * p.K * p.Cout;
  __syncthreads();   // every wave is done with the operand stages
  constexpr int ROWP = WTN * 4 + 16;
  constexpr int HALF = MI / 2 * 16;                     // rows staged per round (two rounds)
  if constexpr (NW * HALF * ROWP <= ST * STB) {
    if ((p.Cout & 3) == 0) {
      char* sw = smem + wave * (HALF * ROWP);
      constexpr int CPR4 = WTN / 4, CH = HALF * CPR4;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int i2 = 0; i2 < MI / 2; ++i2)
#pragma unroll
          for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r)
              *reinterpret_cast<float*>(sw + (i2 * 16 + g * 4 + r) * ROWP + (j * 16 + li) * 4) = acc[h * (MI / 2) + i2][j][r];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int it = 0; it < (CH + 63) / 64; ++it) {
          const int idx = it * 64 + lane;
          const int row = idx / CPR4, ch = idx % CPR4;
          const int kk = kt0 + wm * WTM + h * HALF + row, n = n0 + wn * WTN + ch * 4;
          if (idx < CH && kk < p.K && n < p.Cout)
            *reinterpret_cast<float4*>(slab + (size_t)kk * p.Cout + n) =
                *reinterpret_cast<const float4*>(sw + row * ROWP + ch * 16);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
      }
      return;
    }
  }
#pragma unroll
  for (int i = 0; i < MI; ++i) {
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int n = n0 + wn * WTN + j * 16 + li;
      if (n >= p.Cout) continue;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int kk = kt0 + wm * WTM + i * 16 + g * 4 + r;
        if (kk < p.K) slab[(size_t)kk * p.Cout + n] = acc[i][j][r];
      }
    }
  }
}

// out[i] = sum_k slabs[k][i] in a fixed order.  256 threads = 64 outputs x 4 split lanes
// (lane g sums k = g, g+4, ... with four loads in flight), combined through LDS: short
// dependent chains even for hundreds of splits of a small matrix.
__global__ __launch_bounds__(256) void slab_reduce_kernel(const float* slabs, float* out, int64_t n, int splits) {
  __shared__ float sh[4][64];
  const int o = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int64_t i = (int64_t)blockIdx.x * 64 + o;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  if (i < n) {
    int k = g;
    for (; k + 12 < splits; k += 16) {
      a0 += slabs[(size_t)k * n + i];
      a1 += slabs[(size_t)(k + 4) * n + i];
      a2 += slabs[(size_t)(k + 8) * n + i];
      a3 += slabs[(size_t)(k + 12) * n + i];
    }
    for (; k < splits; k += 4) a0 += slabs[(size_t)k * n + i];
  }
  sh[g][o] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  if (g == 0 && i < n) out[i] = (sh[0][o] + sh[1][o]) + (sh[2][o] + sh[3][o]);
}

// first layer (Cin = 3): dw[27][Cout] by direct accumulation; each block reduces a pixel
// range for all 27*Cout outputs (thread = one (k, co) pair), partials summed afterwards.
__global__ __launch_bounds__(1024) void conv_first_wgrad_kernel(const float* img, const bf16* dy, float* slabs, int B,
                                                                int H, int W, int Cout, int px_per_block) {
  const int M = B * H * W;
  const int o = threadIdx.x;
  const int nout = 27 * Cout;
  if (o >= nout) return;
  const int k = o / Cout, co = o - k * Cout;
  const int tap = k / 3, ci = k - tap * 3;
  const int kh = tap / 3, kw = tap - kh * 3;
  const int m0 = blockIdx.x * px_per_block;
  int m1 = m0 + px_per_block;
  if (m1 > M) m1 = M;
  float acc = 0.f;
  for (int m = m0; m < m1; ++m) {
    const int b = m / (H * W);
    const int rem = m - b * H * W;
    const int y = rem / W, x = rem - y * W;
    const int iy = y + kh - 1, ix = x + kw - 1;
    if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W)
      acc += img[((size_t)(b * H + iy) * W + ix) * 3 + ci] * (float)dy[(size_t)m * Cout + co];
  }
  slabs[(size_t)blockIdx.x * nout + o] = acc;
}

void plan(const disyolo_conv_desc* d, int* bn, int* splits, int* steps_per_split, int* steps) {
  const int M = d->B * d->Ho * d->Wo;
  const int K = d->ksize * d->ksize * (d->C0 + d->C1);
  *bn = d->Cout > 64 ? 128 : (d->Cout > 32 ? 64 : 32);
  const int tiles = ceil_div(K, 128) * ceil_div(d->Cout, *bn);
  *steps = ceil_div(M, 32);
  int s = 512 / tiles;  // fill the 256 CUs about twice; a full grid needs no pixel split
  if (s < 1) s = 1;
  const int max_s = ceil_div(*steps, 8);
  if (s > max_s) s = max_s;
  if (s < 1) s = 1;
  *steps_per_split = ceil_div(*steps, s);
  *splits = ceil_div(*steps, *steps_per_split);
}

}  // namespace

extern "C" size_t disyolo_conv2d_wgrad_workspace(const disyolo_conv_desc* d) {
  if (!d) return 0;
  int bn, splits, sps, steps;
  plan(d, &bn, &splits, &sps, &steps);
  const size_t K = (size_t)d->ksize * d->ksize * (d->C0 + d->C1);
  return (size_t)splits * K * d->Cout * sizeof(float);
}

extern "C" int disyolo_conv2d_wgrad(const disyolo_conv_desc* d, const void* dy, int dy_ld, float* dw, void* workspace,
                                    size_t workspace_bytes, void* stream) {
  DY_REQUIRE(d && dy && dw, "wgrad: null pointer");
  DY_REQUIRE(d->ksize == 1 || d->ksize == 3, "wgrad: ksize");
  DY_REQUIRE(d->C0 > 0 && d->C0 % 8 == 0 && d->C1 % 8 == 0, "wgrad: channels must be multiples of 8");
  DY_REQUIRE(d->C1 == 0 || (d->ksize == 1 && d->stride == 1 && d->x1), "wgrad: bad fused-concat layer");
  DY_REQUIRE(dy_ld >= d->Cout && dy_ld % 8 == 0, "wgrad: dy_ld must be >= Cout and a multiple of 8");
  DY_REQUIRE(d->in_div == 1, "wgrad: in_div must be 1");
  DY_REQUIRE((int64_t)d->B * d->H * d->W * d->C0 * 2 < (1LL << 31) && (int64_t)d->B * d->Ho * d->Wo * dy_ld * 2 < (1LL << 31),
             "wgrad: a source tensor exceeds the 2 GiB the 32-bit gather offsets address");
  if (!workspace || workspace_bytes < disyolo_conv2d_wgrad_workspace(d)) {
    disyolo_set_error("wgrad: workspace too small");
    return DISYOLO_E_WORKSPACE;
  }
  {
    const disyolo_conv_desc c = *d;
    DY_RECORD_OR_RUN([=](void* s) { return disyolo_conv2d_wgrad(&c, dy, dy_ld, dw, workspace, workspace_bytes, s); });
  }
  WgradParams p;
  p.x0 = (const bf16*)d->x0;
  p.x1 = (const bf16*)d->x1;
  p.dy = (const bf16*)dy;
  p.B = d->B; p.H = d->H; p.W = d->W; p.C0 = d->C0; p.C1 = d->C1; p.Cin = d->C0 + d->C1;
  p.Ho = d->Ho; p.Wo = d->Wo; p.Cout = d->Cout; p.ldy = dy_ld;
  p.ks = d->ksize; p.stride = d->stride; p.pad_t = d->pad_t; p.pad_l = d->pad_l;
  p.M = d->B * d->Ho * d->Wo;
  p.K = d->ksize * d->ksize * p.Cin;
  p.bytes0 = (unsigned)((size_t)d->B * d->H * d->W * d->C0 * 2);
  p.bytes1 = (unsigned)((size_t)d->B * (d->H / 2) * (d->W / 2) * d->C1 * 2);
  p.bytesy = (unsigned)((size_t)p.M * dy_ld * 2);
  int bn, splits;
  plan(d, &bn, &splits, &p.steps_per_split, &p.steps);
  p.out = splits == 1 ? dw : (float*)workspace;
  hipStream_t s = (hipStream_t)stream;
  p.tilesK = ceil_div(p.K, 128);
  p.tilesN = ceil_div(p.Cout, bn);
  dim3 grid(p.tilesK * p.tilesN * splits);
  // pipeline depth: d->tile (1..3 -> 2..4 stages) overrides the default (tuning)
  const int st = (d->tile >= 1 && d->tile <= 3) ? d->tile + 1 : 3;
#define DY_WG(BNV, YB)                                                                                   \
  if (st == 2) hipLaunchKernelGGL((conv_wgrad_kernel<BNV, 2>), grid, dim3(256), 2 * (8192 + YB), s, p);      \
  else if (st == 3) hipLaunchKernelGGL((conv_wgrad_kernel<BNV, 3>), grid, dim3(256), 3 * (8192 + YB), s, p); \
  else hipLaunchKernelGGL((conv_wgrad_kernel<BNV, 4>), grid, dim3(256), 4 * (8192 + YB), s, p);
  if (bn == 128) { DY_WG(128, 8192) }
  else if (bn == 64) { DY_WG(64, 4096) }
  else { DY_WG(32, 4096) }
#undef DY_WG
  DY_CHECK_LAUNCH();
  if (splits > 1) {
    const int64_t n = (int64_t)p.K * p.Cout;
    hipLaunchKernelGGL(slab_reduce_kernel, dim3(ceil_div(n, 64)), dim3(256), 0, s, (const float*)workspace, dw, n,
                       splits);
    DY_CHECK_LAUNCH();
  }
  return DISYOLO_OK;
}

extern "C" size_t disyolo_conv_first_wgrad_workspace(int B, int H, int W, int Cout) {
  const int64_t M = (int64_t)B * H * W;
  const int blocks = ceil_div(M, 1024);
  return (size_t)blocks * 27 * Cout * sizeof(float);
}

extern "C" int disyolo_conv_first_wgrad(const float* images, const void* dy, float* dw, int B, int H, int W, int Cout,
                                        void* workspace, size_t workspace_bytes, void* stream) {
  DY_REQUIRE(images && dy && dw && B > 0 && H > 0 && W > 0 && Cout > 0 && 27 * Cout <= 1024, "first_wgrad: bad args");
  if (!workspace || workspace_bytes < disyolo_conv_first_wgrad_workspace(B, H, W, Cout)) {
    disyolo_set_error("first_wgrad: workspace too small");
    return DISYOLO_E_WORKSPACE;
  }
  DY_RECORD_OR_RUN([=](void* s) { return disyolo_conv_first_wgrad(images, dy, dw, B, H, W, Cout, workspace, workspace_bytes, s); });
  const int64_t M = (int64_t)B * H * W;
  const int blocks = ceil_div(M, 1024);
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(conv_first_wgrad_kernel, dim3(blocks), dim3(1024), 0, s, images, (const bf16*)dy,
                     (float*)workspace, B, H, W, Cout, 1024);
  DY_CHECK_LAUNCH();
  const int64_t n = 27 * Cout;
  hipLaunchKernelGGL(slab_reduce_kernel, dim3(ceil_div(n, 64)), dim3(256), 0, s, (const float*)workspace, dw, n,
                     blocks);
  DY_CHECK_LAUNCH();
  return DISYOLO_OK;
}
