// The two HBM-bound [1x1 -> 32 channels] -> [3x3 32 -> 64] chains of the 288^2 maps in ONE launch each, for the steps in
// which their batch norms run in inference mode (locked backbone of training stage 1 / every inference call):
//
//   residual block 1   act3 = leaky(bn3(conv1x1(act2)))           64 -> 32     yolo/yolo3_net_pos.py:169-176
//     (post 0)         act4 = leaky(bn4(conv3x3(act3))) + act2    32 -> 64
//   mask head          act80 = leaky(bn(conv1x1([act4, up2(act79)])))   96 -> 32   yolo/yolo3_net_pos.py:404-412
//     (post 1)         act81 = leaky(bn(conv3x3(act80)))                32 -> 64
//                      out   = conv1x1(act81) + bias                    64 -> 9 (k_map^2 position-sensitive score maps), f32
//
// Unfused, the 32- and 64-channel intermediates of these chains are written and read back at 288^2: at B = 32 the
// residual block moves 1.36 GB for 0.68 GB of input + output (97 + 175 us), the mask head 1.24 GB for 0.43 GB
// (113 + 130 + 107 us), all of it at the HBM bound of the separate kernels.  Here the intermediates live in LDS.
//
// Two persistent 4-wave blocks per CU walk over 8 x 16-pixel output patches (the scheme of conv_first2.hip):
//   park     the 10 x 18-pixel input tile (halo 1) of the NEXT patch, fetched into registers one patch ahead: 16-byte
//            chunks, the second source of the concat read at (y/2, x/2); pixels outside the image are zeros;
//   phase A  the 1x1 conv on the 180 tile pixels: 12 fragments of 16 pixels, K = 64 / 96 from the tile (rows of
//            K*2 + 16 bytes: 16 consecutive pixels hit 16 different bank groups), weights in registers; folded BN + leaky;
//            pixels outside the image are forced to 0 (they are the 3x3 conv's SAME padding); bf16 tile [180][32] with
//            80-byte rows;
//   phase B  the 3x3 conv from that tile.  Each wave owns 32 of the 64 channels and 4 of the 8 patch rows and keeps its
//            weights (72 VGPRs) in registers: per tap 4 LDS reads (one base register, immediate offsets) feed 8 MFMAs;
//   post 0   folded BN + leaky + the residual (the centre of the input tile, still in LDS), bf16 rows leave through a
//            per-wave staging tile as 64-byte half lines;
//   post 1   folded BN + leaky -> bf16 tile [128][64] in LDS -> 1x1 conv 64 -> 9 on the matrix cores (weights padded to 16
//            rows, 2 MFMAs per 16 pixels) + bias, f32 [pixels][9] straight from the accumulators.
#include <utility>
#include "common.h"
#include "runtime.h"

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int PH = 8, PW = 16;                      // output patch
constexpr int TH = PH + 2, TW = PW + 2;             // input / intermediate tile: 10 x 18
constexpr int NT = TH * TW;                         // 180 pixels
constexpr int NW = 4;
constexpr int NTF = (NT + 15) / 16;                 // 12 fragments = 3 per wave
constexpr int FPW = NTF / NW;
constexpr int AP = 80;                              // bytes per pixel of the intermediate tile
constexpr int SROW = 80;                            // staging: 16 pixels x (64 B + pad)
constexpr int TP = 144;                             // post 1: bytes per pixel of the 64-channel tile
constexpr int NOUT = 9;
static_assert(NTF % NW == 0, "fragments must divide over the waves");

template <int KIN>
struct Cfg {
  static constexpr int CPP = KIN / 8;               // 16-byte chunks per pixel
  static constexpr int XP = KIN * 2 + 16;           // bytes per pixel of the input tile
  static constexpr int NCH = NT * CPP;
  static constexpr int IPT = (NCH + NW * 64 - 1) / (NW * 64);
  static constexpr int X_BYTES = NTF * 16 * XP;     // (rows 180..191: scratch of the last fragment)
  static constexpr int A_BYTES = NTF * 16 * AP;
  static constexpr int DUMMY_BYTES = NW * 64 * 16;
  static constexpr int BN_BYTES = (2 * 64 + 2 * 32) * 4;   // folded scale / shift of the 3x3 and of the 1x1 layer
};

struct BParams {
  const bf16* x0;
  const bf16* x1;
  const bf16* wA;         // packed [32][KIN]
  const float* scA;
  const float* shA;
  const bf16* wB;         // packed [64][9 * 32]
  const float* scB;
  const float* shB;
  const bf16* wC;         // post 1: packed [9][64]
  const float* biasC;
  void* y;                // post 0: bf16 [B][H][W][64]; post 1: f32 [B][H][W][9]
  int B, H, W, C0, C1, tilesY, tilesX, tiles;
  int gb, gy, gx;         // gridDim.x tiles as (images, tile rows, tile columns)
  float alpha;
};

template <int KIN, int POST>
__global__ __launch_bounds__(NW * 64, 2) void block32_kernel(BParams p) {
  using C = Cfg<KIN>;
  constexpr int XP = C::XP, CPP = C::CPP, IPT = C::IPT;
  constexpr unsigned X_OFF = 0, A_OFF = C::X_BYTES, DUMMY_OFF = A_OFF + C::A_BYTES, BN_OFF = DUMMY_OFF + C::DUMMY_BYTES,
                     STG_OFF = BN_OFF + C::BN_BYTES;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const xl = smem + X_OFF;
  float* const bnB = reinterpret_cast<float*>(smem + BN_OFF);
  char* const stg = smem + STG_OFF;                 // post 0: per-wave staging; post 1: the [128][64] tile
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int frow = lane & 15, cq = lane >> 4;
  const int hh = wave & 1, ph = wave >> 1;          // phase B: channel half, patch-row half

  // ---- once per block
  if (tid < 64) {
    bnB[tid] = p.scB[tid];
    bnB[64 + tid] = p.shB[tid];
    bnB[128 + tid] = tid < 32 ? p.scA[tid] : p.shA[tid - 32];
  }
  // 3x3 weights of this wave's 32 channels: A fragments (channel hh*32 + nf*16 + frow, k = tap*32 + 8 cq .. +7)
  bf16x8 wBr[9][2];
#pragma unroll
  for (int tap = 0; tap < 9; ++tap)
#pragma unroll
    for (int nf = 0; nf < 2; ++nf)
      wBr[tap][nf] = *reinterpret_cast<const bf16x8*>(p.wB + (size_t)(hh * 32 + nf * 16 + frow) * 288 + tap * 32 + cq * 8);
  // 1x1 weights: A fragments (channel n*16 + frow, k = ks*32 + 8 cq .. +7)
  bf16x8 wAr[2][KIN / 32];
#pragma unroll
  for (int n = 0; n < 2; ++n)
#pragma unroll
    for (int ks = 0; ks < KIN / 32; ++ks)
      wAr[n][ks] = *reinterpret_cast<const bf16x8*>(p.wA + (size_t)(n * 16 + frow) * KIN + ks * 32 + cq * 8);
  // post 1: the 64 -> 9 weights, rows 9..15 zero
  bf16x8 wCr[2];
  f32x4 biasC = {0.f, 0.f, 0.f, 0.f};
  if (POST == 1) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const bf16x8 v = *reinterpret_cast<const bf16x8*>(p.wC + (size_t)(frow < NOUT ? frow : 0) * 64 + ks * 32 + cq * 8);
      const bf16x8 z = {};
      wCr[ks] = frow < NOUT ? v : z;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) biasC[r] = (cq * 4 + r < NOUT) ? p.biasC[cq * 4 + r] : 0.f;
  }
  const f32x2 alpha2 = f32x2{p.alpha, p.alpha};

  // phase A, fragment i of this wave = tile pixels (wave + NW*i)*16 + frow; bits 4i..4i+3 of aflag: the pixel lies in the
  // tile's first row / last row / first column / last column (outside the image when the patch touches that border)
  unsigned aflag = 0u;
#pragma unroll
  for (int i = 0; i < FPW; ++i) {
    const int q = (wave + NW * i) * 16 + frow;
    const int ay = q / TW, ax = q - ay * TW;
    const unsigned f = q < NT ? ((ay == 0 ? 1u : 0u) | (ay == TH - 1 ? 2u : 0u) | (ax == 0 ? 4u : 0u) | (ax == TW - 1 ? 8u : 0u)) : 0u;
    aflag |= f << (4 * i);
  }
  const unsigned ard = X_OFF + (unsigned)((wave * 16 + frow) * XP + cq * 16);    // + i * NW*16*XP + ks * 64
  const unsigned awr = A_OFF + (unsigned)((wave * 16 + frow) * AP + cq * 8);     // + i * NW*16*AP + n * 32
  // phase B: 16 lanes = 16 consecutive output columns; tile pixel (ph*4 + g + kh) * 18 + frow + kw
  const unsigned xb = A_OFF + (unsigned)((ph * 4 * TW + frow) * AP + cq * 16);

  // input tile: chunk e = i * 256 + tid of the [180][CPP] tile -> where it comes from (elements from the patch origin,
  // second source: bit 31) and where it is parked
  int srcoff[IPT];
  unsigned dst[IPT];
  const int W1 = p.W >> 1;
#pragma unroll
  for (int i = 0; i < IPT; ++i) {
    const int e = i * (NW * 64) + tid;
    const int px = e / CPP, ch = e - px * CPP;
    const int ay = px / TW, ax = px - ay * TW;
    const bool real = e < C::NCH;
    const bool second = ch * 8 >= p.C0;
    int off = (ay * p.W + ax) * p.C0 + ch * 8;
    if (second) off = (((ay + 1) / 2 - 1) * W1 + ((ax + 1) / 2 - 1)) * p.C1 + ch * 8 - p.C0;   // floor((a - 1) / 2), a >= 0
    srcoff[i] = real ? off : 0;
    // bits 0-15: LDS address, 16-19: tile row, 20-24: tile column, 30: a real chunk, 31: of the second source
    dst[i] = (real ? X_OFF + (unsigned)(px * XP + ch * 16) : DUMMY_OFF + (unsigned)tid * 16) | ((unsigned)ay << 16) |
             ((unsigned)ax << 20) | (real ? 1u << 30 : 0u) | (real && second ? 1u << 31 : 0u);
  }
  uint4 pre[IPT];
  auto fetch = [&](int b, int ty, int tx, auto lo_tag, auto hi_tag) {
    constexpr int I0 = decltype(lo_tag)::value, I1 = decltype(hi_tag)::value;
    const int y0 = ty * PH - 1, x0 = tx * PW - 1;   // tile origin (may be -1)
    const int64_t base0 = ((int64_t)(b * p.H + y0) * p.W + x0) * p.C0;
    const int64_t base1 = KIN > 64 ? ((int64_t)(b * (p.H >> 1) + (ty * PH >> 1)) * W1 + (tx * PW >> 1)) * p.C1 : 0;
    const bool border = ty == 0 || tx == 0 || ty == p.tilesY - 1 || tx == p.tilesX - 1;   // (uniform)
    if (!border) {
#pragma unroll
      for (int i = I0; i < I1; ++i) {
        const bool second = KIN > 64 && (dst[i] >> 31) != 0u;
        const bf16* src = second ? p.x1 + base1 : p.x0 + base0;
        pre[i] = *reinterpret_cast<const uint4*>(src + srcoff[i]);
      }
    } else {
#pragma unroll
      for (int i = I0; i < I1; ++i) {
        unsigned d = dst[i];
        asm volatile("" : "+v"(d));                 // (unpacked here, per tile: hoisted out of the loop it costs registers)
        const int ay = (d >> 16) & 15, ax = (d >> 20) & 31;
        const bool ok = (d & (1u << 30)) != 0u && (unsigned)(y0 + ay) < (unsigned)p.H && (unsigned)(x0 + ax) < (unsigned)p.W;
        const bool second = KIN > 64 && (d >> 31) != 0u;
        const bf16* src = second ? p.x1 + base1 : p.x0 + base0;
        const uint4 v = *reinterpret_cast<const uint4*>(ok ? src + srcoff[i] : p.x0);   // (always a load)
        pre[i] = ok ? v : uint4{0u, 0u, 0u, 0u};
      }
    }
  };
  auto park = [&](auto lo_tag, auto hi_tag) {
    constexpr int I0 = decltype(lo_tag)::value, I1 = decltype(hi_tag)::value;
#pragma unroll
    for (int i = I0; i < I1; ++i) {
      unsigned d = dst[i];
      asm volatile("" : "+v"(d));
      *reinterpret_cast<uint4*>(smem + (d & 0xffffu)) = pre[i];
    }
  };

  auto phaseA = [&](auto edge_tag, unsigned edge_mask) {
    constexpr bool EDGE = decltype(edge_tag)::value;
    constexpr int KS = KIN / 32;
    bf16x8 xf[2][KS];
    auto load = [&](int i, bf16x8* d) {
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) d[ks] = *reinterpret_cast<const bf16x8*>(smem + (ard & 0xffffu) + i * (NW * 16 * XP) + ks * 64);
    };
    load(0, xf[0]);
#pragma unroll
    for (int i = 0; i < FPW; ++i) {
      if (i + 1 < FPW) load(i + 1, xf[(i + 1) & 1]);
      __builtin_amdgcn_sched_barrier(0);
      const bool outside = EDGE && (aflag & (edge_mask << (4 * i))) != 0u;
#pragma unroll
      for (int n = 0; n < 2; ++n) {
        const f32x4 sv = *reinterpret_cast<const f32x4*>(bnB + 128 + n * 16 + cq * 4);     // channel n*16 + 4*cq + r
        const f32x4 hv = *reinterpret_cast<const f32x4*>(bnB + 160 + n * 16 + cq * 4);
        f32x4 a = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wAr[n][ks], xf[i & 1][ks], a, 0, 0, 0);
        f32x2 v0 = f32x2{a[0], a[1]} * f32x2{sv[0], sv[1]} + f32x2{hv[0], hv[1]};
        f32x2 v1 = f32x2{a[2], a[3]} * f32x2{sv[2], sv[3]} + f32x2{hv[2], hv[3]};
        const f32x2 t0 = v0 * alpha2, t1 = v1 * alpha2;
        uint2 o;
        o.x = pack2(fmaxf(t0[0], v0[0]), fmaxf(t0[1], v0[1]));
        o.y = pack2(fmaxf(t1[0], v1[0]), fmaxf(t1[1], v1[1]));
        if (EDGE && outside) o = uint2{0u, 0u};
        *reinterpret_cast<uint2*>(smem + (awr & 0xffffu) + i * (NW * 16 * AP) + n * 32) = o;
      }
    }
  };

  int b = 0, ty = 0, tx = 0;
  {
    const int t0 = blockIdx.x;
    if (t0 >= p.tiles) return;                      // (uniform: whole block)
    const int tpi = p.tilesY * p.tilesX;
    b = t0 / tpi;
    const int pr = t0 - b * tpi;
    ty = pr / p.tilesX;
    tx = pr - ty * p.tilesX;
  }
  using I_0 = std::integral_constant<int, 0>;
  using I_H = std::integral_constant<int, POST == 1 ? (IPT + 1) / 2 : IPT>;   // post 1 fetches / parks in two batches
  using I_N = std::integral_constant<int, IPT>;
  fetch(b, ty, tx, I_0{}, I_N{});
  park(I_0{}, I_N{});
  __syncthreads();
  for (int t = blockIdx.x; t < p.tiles; t += gridDim.x) {
    // the tile after this one
    int nb = b + p.gb, nty = ty + p.gy, ntx = tx + p.gx;
    if (ntx >= p.tilesX) { ntx -= p.tilesX; ++nty; }
    if (nty >= p.tilesY) { nty -= p.tilesY; ++nb; }
    const bool more = t + (int)gridDim.x < p.tiles;
    if (more) fetch(nb, nty, ntx, I_0{}, I_H{});    // in flight under phase A (post 0: under both phases)
    const unsigned edge_mask = (ty == 0 ? 1u : 0u) | (ty == p.tilesY - 1 ? 2u : 0u) | (tx == 0 ? 4u : 0u) |
                               (tx == p.tilesX - 1 ? 8u : 0u);
    if (edge_mask) phaseA(std::true_type{}, edge_mask);
    else phaseA(std::false_type{}, 0u);
    __syncthreads();                                // the intermediate tile is complete
    if (POST == 1 && more) {                        // (nothing reads the input tile after phase A)
      park(I_0{}, I_H{});
      fetch(nb, nty, ntx, I_H{}, I_N{});            // the second batch flies under phase B
    }
    // ---- phase B: 4 patch rows x 16 columns x 32 channels per wave
    f32x4 acc[4][2];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int nf = 0; nf < 2; ++nf) acc[g][nf] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int kh = tap / 3, kw = tap % 3;
      bf16x8 xf[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) xf[g] = *reinterpret_cast<const bf16x8*>(smem + (xb & 0xffffu) + ((g + kh) * TW + kw) * AP);
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int nf = 0; nf < 2; ++nf)
          acc[g][nf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wBr[tap][nf], xf[g], acc[g][nf], 0, 0, 0);
    }
    f32x2 s2[2][2], h2[2][2];
#pragma unroll
    for (int nf = 0; nf < 2; ++nf) {
      const f32x4 sv = *reinterpret_cast<const f32x4*>(bnB + hh * 32 + nf * 16 + cq * 4);
      const f32x4 hv = *reinterpret_cast<const f32x4*>(bnB + 64 + hh * 32 + nf * 16 + cq * 4);
      s2[nf][0] = f32x2{sv[0], sv[1]}; s2[nf][1] = f32x2{sv[2], sv[3]};
      h2[nf][0] = f32x2{hv[0], hv[1]}; h2[nf][1] = f32x2{hv[2], hv[3]};
    }
    if (POST == 0) {
      char* const sw0 = stg + wave * (2 * 16 * SROW);
      bf16* const yb = reinterpret_cast<bf16*>(p.y);
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        char* const sw = sw0 + (g & 1) * (16 * SROW);
        // the residual: input-tile pixel (ph*4 + g + 1, frow + 1), channels hh*32 + nf*16 + 4 cq .. +3
        const char* rsrc = xl + ((ph * 4 + g + 1) * TW + frow + 1) * XP + (hh * 32 + cq * 4) * 2;
#pragma unroll
        for (int nf = 0; nf < 2; ++nf) {
          const uint2 rr = *reinterpret_cast<const uint2*>(rsrc + nf * 32);
          f32x2 v0 = f32x2{acc[g][nf][0], acc[g][nf][1]} * s2[nf][0] + h2[nf][0];
          f32x2 v1 = f32x2{acc[g][nf][2], acc[g][nf][3]} * s2[nf][1] + h2[nf][1];
          const f32x2 t0 = v0 * alpha2, t1 = v1 * alpha2;
          uint2 o;
          o.x = pack2(fmaxf(t0[0], v0[0]) + __builtin_bit_cast(float, rr.x << 16),
                      fmaxf(t0[1], v0[1]) + __builtin_bit_cast(float, rr.x & 0xffff0000u));
          o.y = pack2(fmaxf(t1[0], v1[0]) + __builtin_bit_cast(float, rr.y << 16),
                      fmaxf(t1[1], v1[1]) + __builtin_bit_cast(float, rr.y & 0xffff0000u));
          *reinterpret_cast<uint2*>(sw + frow * SROW + nf * 32 + cq * 8) = o;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // 16 pixels x 64 bytes (this wave's channel half): one 16-byte chunk per lane
        const int px = lane >> 2, ch = lane & 3;
        const size_t m = ((size_t)b * p.H + ty * PH + ph * 4 + g) * p.W + tx * PW + px;
        *reinterpret_cast<uint4*>(yb + m * 64 + hh * 32 + ch * 8) = *reinterpret_cast<const uint4*>(sw + px * SROW + ch * 16);
      }
    } else {
      // act81 -> bf16 tile [128 pixels][64 channels]
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int nf = 0; nf < 2; ++nf) {
          f32x2 v0 = f32x2{acc[g][nf][0], acc[g][nf][1]} * s2[nf][0] + h2[nf][0];
          f32x2 v1 = f32x2{acc[g][nf][2], acc[g][nf][3]} * s2[nf][1] + h2[nf][1];
          const f32x2 t0 = v0 * alpha2, t1 = v1 * alpha2;
          uint2 o;
          o.x = pack2(fmaxf(t0[0], v0[0]), fmaxf(t0[1], v0[1]));
          o.y = pack2(fmaxf(t1[0], v1[0]), fmaxf(t1[1], v1[1]));
          *reinterpret_cast<uint2*>(stg + ((ph * 4 + g) * 16 + frow) * TP + (hh * 32 + nf * 16 + cq * 4) * 2) = o;
        }
      __syncthreads();
      // 64 -> 9: this wave's two patch rows
      float* const yf = reinterpret_cast<float*>(p.y);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int row = wave * 2 + j;
        f32x4 a = biasC;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          const bf16x8 xv = *reinterpret_cast<const bf16x8*>(stg + (row * 16 + frow) * TP + ks * 64 + cq * 16);
          a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wCr[ks], xv, a, 0, 0, 0);
        }
        const size_t m = ((size_t)b * p.H + ty * PH + row) * p.W + tx * PW + frow;
        float* dstp = yf + m * NOUT + cq * 4;
        if (cq < 2) {
          dstp[0] = a[0]; dstp[1] = a[1]; dstp[2] = a[2]; dstp[3] = a[3];
        } else if (cq == 2) {
          dstp[0] = a[0];
        }
      }
      if (more) park(I_H{}, I_N{});
    }
    if (POST == 0) {
      __syncthreads();                              // every wave has read its residuals from the input tile
      if (more) park(I_0{}, I_N{});
    }
    b = nb; ty = nty; tx = ntx;
    __syncthreads();                                // everyone is done with the tiles; the parked input is visible
  }
}

template <int KIN, int POST>
constexpr int lds_bytes() {
  using C = Cfg<KIN>;
  return C::X_BYTES + C::A_BYTES + C::DUMMY_BYTES + C::BN_BYTES + (POST == 0 ? NW * 2 * 16 * SROW : PH * PW * TP);
}

template <int KIN, int POST>
int launch_block32(BParams& p, hipStream_t s) {
  constexpr int LDS = lds_bytes<KIN, POST>();
  static_assert(2 * LDS <= 160 * 1024, "two blocks per CU");
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&block32_kernel<KIN, POST>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    attr_set = true;
  }
  static const int ncu = [] {
    int dev = 0, n = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
    return n > 0 ? n : 256;
  }();
  int grid = 2 * ncu;                               // two persistent blocks per CU
  if (grid > p.tiles) grid = p.tiles;
  const int tpi = p.tilesY * p.tilesX;
  p.gb = grid / tpi;
  p.gy = (grid % tpi) / p.tilesX;
  p.gx = grid % p.tilesX;
  hipLaunchKernelGGL((block32_kernel<KIN, POST>), dim3(grid), dim3(NW * 64), LDS, s, p);
  DY_CHECK_LAUNCH();
  return DISYOLO_OK;
}

}  // namespace

extern "C" int disyolo_block32_fused_ok(int B, int H, int W, int C0, int C1, int post) {
  const bool shape = (C0 == 64 && C1 == 0 && post == 0) || (C0 == 64 && C1 == 32 && post == 1);
  return (shape && B > 0 && H > 0 && W > 0 && H % PH == 0 && W % PW == 0 &&
          (int64_t)B * H * W * 128 < (1LL << 31)) ? 1 : 0;   // bytes of the largest tensor (64 bf16 channels in / out; the f32 score maps are 36 B / pixel)
}

extern "C" int disyolo_block32_fused_fwd(const void* x0, const void* x1, int C0, int C1, const void* wA, const float* scaleA,
                                         const float* shiftA, const void* wB, const float* scaleB, const float* shiftB, int post,
                                         const void* wC, const float* biasC, void* y, int B, int H, int W, float alpha,
                                         void* stream) {
  DY_REQUIRE(x0 && wA && scaleA && shiftA && wB && scaleB && shiftB && y, "block32_fused: null pointer");
  DY_REQUIRE(disyolo_block32_fused_ok(B, H, W, C0, C1, post) == 1,
             "block32_fused: covers [64 -> 32 -> 64 + residual] (post 0) and [64 + up(32) -> 32 -> 64 -> 9] (post 1), H %% %d == 0, W %% %d == 0",
             PH, PW);
  DY_REQUIRE(post == 0 || (x1 && wC && biasC), "block32_fused: post 1 needs x1, wC, biasC");
  DY_RECORD_OR_RUN([=](void* s) {
    return disyolo_block32_fused_fwd(x0, x1, C0, C1, wA, scaleA, shiftA, wB, scaleB, shiftB, post, wC, biasC, y, B, H, W, alpha, s);
  });
  BParams p;
  p.x0 = (const bf16*)x0; p.x1 = (const bf16*)x1;
  p.wA = (const bf16*)wA; p.scA = scaleA; p.shA = shiftA;
  p.wB = (const bf16*)wB; p.scB = scaleB; p.shB = shiftB;
  p.wC = (const bf16*)wC; p.biasC = biasC; p.y = y;
  p.B = B; p.H = H; p.W = W; p.C0 = C0; p.C1 = C1;
  p.tilesY = H / PH; p.tilesX = W / PW; p.tiles = B * p.tilesY * p.tilesX;
  p.alpha = alpha;
  return post == 0 ? launch_block32<64, 0>(p, (hipStream_t)stream) : launch_block32<96, 1>(p, (hipStream_t)stream);
}
