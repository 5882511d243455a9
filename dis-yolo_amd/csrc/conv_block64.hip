// The residual blocks of the 144^2 maps ([1x1 128 -> 64] -> [3x3 64 -> 128] + block input; conv6+7 and conv8+9,
// yolo/yolo3_net_pos.py:184-201) in ONE launch each, for the steps in which their batch norms run in inference mode (the
// locked backbone of training stage 1, every inference call).  Unfused, a block moves 0.68 GB at B = 32 (input read twice,
// the 64-channel intermediate written and read) for 0.34 GB of input + output; the 1x1 runs at the HBM bound, the 3x3 at
// 135 us against 39 us of MFMA work.
//
// The scheme of conv_block32.hip with twice the channels, which changes two things: the 3x3 weights (9 x 128 x 64 bf16 =
// 147 KB) only fit in the registers of EIGHT waves (one persistent 8-wave block per CU, 144 VGPRs of weights per wave:
// 32 output channels x 9 taps x 64 k), and the input tile is double-buffered in LDS instead of being held in registers over
// both phases.  Per 8 x 16-pixel output patch:
//   fetch    the 10 x 18 x 128 input tile of the NEXT patch by LDS-DMA into the other input buffer, requested at the top of
//            the patch (45 wave-wide 1 KB transfers, pixels outside the image out of the descriptor's range = zeros); 256-byte
//            pixel rows, 16-byte chunk c of pixel p at slot c ^ (p & 15): no padding (DMA writes linear runs), no conflicts;
//   phase A  the 1x1 conv on the 180 tile pixels (24 units of 16 pixels x 32 channels over 8 waves), K = 128, its weights
//            (64 x 128) read from LDS as A fragments; folded BN + leaky, zero outside the image (the 3x3's SAME padding),
//            bf16 tile [180][64] with 144-byte rows;
//   phase B  the 3x3 conv from that tile: a wave owns 32 of the 128 channels and 4 of the 8 patch rows; per tap and
//            32-channel slice 4 LDS reads (one base register, immediate offsets) feed 8 MFMAs;
//   epilogue folded BN + leaky + the residual (the centre of the input tile, still in LDS), bf16 rows through a per-wave
//            staging tile (in the intermediate tile's space, after a barrier) as 64-byte quarter rows.
#include <utility>
#include "common.h"
#include "runtime.h"

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int PH = 8, PW = 16;                      // output patch
constexpr int TH = PH + 2, TW = PW + 2;             // input / intermediate tile: 10 x 18
constexpr int NT = TH * TW;                         // 180 pixels
constexpr int NW = 8;
constexpr int NTF = (NT + 15) / 16;                 // 12 fragments: waves 0-3 take two, waves 4-7 one
constexpr int CI = 128, CM = 64, CO = 128;
constexpr int XP = CI * 2;                          // 256 bytes per input-tile pixel: LDS-DMA writes 1 KB runs, so no padding -- the
                                                    // 16-byte chunk c of pixel p sits at slot c ^ (p & 15) instead
constexpr int MP = CM * 2 + 16;                     // 144 bytes per intermediate pixel
constexpr int WP = CI * 2 + 16;                     // 272 bytes per row of the 1x1 weights
constexpr int SROW = 80;                            // staging: 16 pixels x (64 B + pad)
constexpr int CPP = CI / 8;                         // 16-byte chunks per pixel
constexpr int X_BYTES = NTF * 16 * XP;              // 49152 (rows 180..191: scratch of the last fragment)
constexpr int M_BYTES = NTF * 16 * MP;              // 27648; the staging tiles (8 x 2 x 1280) live here after phase B
constexpr int WA_BYTES = CM * WP;                   // 17408
constexpr int BN_BYTES = (2 * CM + 2 * CO) * 4;     // 1536
constexpr unsigned X_OFF = 0, M_OFF = 2 * X_BYTES, WA_OFF = M_OFF + M_BYTES, BN_OFF = WA_OFF + WA_BYTES;
constexpr int LDS_BYTES = BN_OFF + BN_BYTES;
constexpr int NDMA = (NT * CPP + 63) / 64;          // 45 wave-wide 1 KB transfers per input tile
constexpr unsigned OOB = 0x80000000u;
static_assert(LDS_BYTES <= 160 * 1024, "one block per CU");
// a base address ANDed with this is known to be non-negative, which is what hipcc needs to fold the compile-time part of
// an LDS address into the instruction's 16-bit offset field (otherwise: one address register per access, hoisted, spilled)
constexpr unsigned LDS_MASK = 0x3ffffu;
static_assert(NW * 2 * 16 * SROW <= M_BYTES, "the staging tiles fit in the intermediate tile");
static_assert(X_BYTES % 16 == 0 && M_BYTES % 16 == 0 && WA_BYTES % 16 == 0 && BN_BYTES % 16 == 0, "16-byte LDS regions");
static_assert(NDMA <= 6 * NW, "six transfers per wave cover the tile");

typedef int i32x4 __attribute__((ext_vector_type(4)));
// one LDS-DMA: 64 lanes x 16 bytes, buffer descriptor + per-lane byte offset -> LDS (wave-uniform base in M0 + lane * 16);
// lanes whose offset is past the descriptor's range get zeros (conv_igemm.hip has the long version of this comment).  The asm
// writes M0 behind the compiler's back (m0 is not allowed on a clobber list): nothing else in this kernel uses it -- no
// dynamic register indexing, and gfx9+ DS instructions do not read it; the 12 s_mov_b32 m0 of the build are all ours
__device__ __forceinline__ void dma16(unsigned voff, i32x4 srd, unsigned lds_base) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds" : : "v"(voff), "s"(srd), "s"(lds_base) : "memory");
}
__device__ __forceinline__ i32x4 make_srd(const void* base, unsigned bytes) {
  i32x4 r;
  r[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)base);
  r[1] = __builtin_amdgcn_readfirstlane((int)((size_t)base >> 32)) & 0xffff;
  r[2] = __builtin_amdgcn_readfirstlane((int)bytes);
  r[3] = 0x00020000;
  return r;
}


struct B64Params {
  const bf16* x;          // [B][H][W][128]
  const bf16* wA;         // packed [64][128]
  const float* scA;
  const float* shA;
  const bf16* wB;         // packed [128][9 * 64]
  const float* scB;
  const float* shB;
  bf16* y;                // [B][H][W][128]
  int B, H, W, tilesY, tilesX, tiles;
  int gb, gy, gx;         // gridDim.x tiles as (images, tile rows, tile columns)
  float alpha;
#ifdef B64_PROBE
  long long* probe;       // tools/probe_block64.hip: [block][wave][8] cycles per phase, summed over the block's patches
#endif
};

#ifdef B64_PROBE
#define B64_STAMP(k)                                                \
  do {                                                              \
    const long long now_ = (long long)__builtin_amdgcn_s_memtime(); \
    pacc[k] += now_ - plast;                                        \
    plast = now_;                                                   \
  } while (0)
#else
#define B64_STAMP(k) do { } while (0)
#endif

__global__ __launch_bounds__(NW * 64, 2) void block64_kernel(B64Params p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* const bn = reinterpret_cast<float*>(smem + BN_OFF);     // [scA 64][shA 64][scB 128][shB 128]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int frow = lane & 15, cq = lane >> 4;
  const int cqt = wave & 3, ph = wave >> 2;         // phase B: channel quarter, patch-row half

  // ---- once per block
  for (int c = tid; c < CM * CPP; c += NW * 64) {   // 1x1 weights -> LDS rows of 272 bytes
    const int n = c / CPP, ch = c - n * CPP;
    *reinterpret_cast<uint4*>(smem + WA_OFF + n * WP + ch * 16) = *reinterpret_cast<const uint4*>(p.wA + (size_t)n * CI + ch * 8);
  }
  if (tid < CM) {
    bn[tid] = p.scA[tid];
    bn[CM + tid] = p.shA[tid];
  }
  if (tid < CO) {
    bn[2 * CM + tid] = p.scB[tid];
    bn[2 * CM + CO + tid] = p.shB[tid];
  }
  // 3x3 weights of this wave's 32 channels: A fragments (channel cqt*32 + nf*16 + frow, k = tap*64 + ks*32 + 8 cq .. +7)
  bf16x8 wBr[9][2][2];
#pragma unroll
  for (int tap = 0; tap < 9; ++tap)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int nf = 0; nf < 2; ++nf)
        wBr[tap][ks][nf] = *reinterpret_cast<const bf16x8*>(p.wB + (size_t)(cqt * 32 + nf * 16 + frow) * (9 * CM) + tap * CM + ks * 32 + cq * 8);
  const f32x2 alpha2 = f32x2{p.alpha, p.alpha};

  const unsigned wrd = WA_OFF + (unsigned)(frow * WP + cq * 16);                // + nf * 16*WP + ks * 64
  // phase B: 16 lanes = 16 consecutive output columns; tile pixel (ph*4 + g + kh) * 18 + frow + kw
  const unsigned xb = M_OFF + (unsigned)((ph * 4 * TW + frow) * MP + cq * 16);

  // input tile by LDS-DMA: transfer i (wave i % 8 issues it) fills LDS bytes [i * 1024, +1024) of the buffer = slots
  // (pixel, c') = ((i * 64 + lane) / 16, lane % 16); the lane fetches the LOGICAL chunk c' ^ (pixel & 15) of that pixel, or
  // nothing (out of range: zeros) for the pixels outside the image and past the tile.  No registers, no park: the whole
  // next tile is requested at the top of a patch and has the patch's three phases to land.
  const i32x4 srdx = make_srd(p.x, (unsigned)((size_t)p.B * p.H * p.W * CI * 2));
  auto fetch = [&](int b, int ty, int tx, unsigned xbase) {
    const int y0 = ty * PH - 1, x0 = tx * PW - 1;   // tile origin (may be -1)
    int l_ = lane;
    asm volatile("" : "+v"(l_));                    // (offsets computed here, per patch: hoisted they are registers)
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const int i = wave + NW * j;                  // (uniform)
      if (i < NDMA) {
        const int px = i * 4 + (l_ >> 4), cp = l_ & 15;
        const int ay = (px * 3641) >> 16, ax = px - ay * TW;     // px / 18
        const int y = y0 + ay, x = x0 + ax;
        const bool ok = px < NT && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W;
        const unsigned off = ok ? (unsigned)((((b * p.H + y) * p.W + x) * CI + ((cp ^ (px & 15)) << 3)) * 2) : OOB;
        dma16(off, srdx, xbase + (unsigned)i * 1024u);
      }
    }
  };

  // phase A work units: (fragment f of 12, channel half h of 2) = 24 units, three per wave: u = wave + 8 j -> f = u >> 1,
  // h = u & 1 (a fragment's 4 K-slices are read by both of its units: LDS reads are cheaper than idle waves)
  auto phaseA = [&](auto edge_tag, unsigned edge_mask, unsigned xbase) {
    constexpr bool EDGE = decltype(edge_tag)::value;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int u = wave + NW * j, f = u >> 1, h = u & 1;      // (uniform)
      int q = f * 16 + frow;
      asm volatile("" : "+v"(q));                   // (addresses from q are computed here, per patch: see fetch())
      bf16x8 xf[4];
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
        xf[ks] = *reinterpret_cast<const bf16x8*>(smem + ((xbase + (unsigned)(q * XP + (((ks * 4 + cq) ^ (q & 15)) << 4))) & LDS_MASK));
      bool outside = false;
      if (EDGE) {
        const int ay = (q * 3641) >> 16, ax = q - ay * TW;
        outside = q < NT && (((edge_mask & 1u) && ay == 0) || ((edge_mask & 2u) && ay == TH - 1) || ((edge_mask & 4u) && ax == 0) ||
                             ((edge_mask & 8u) && ax == TW - 1));
      }
#pragma unroll
      for (int n2 = 0; n2 < 2; ++n2) {
        const int nf = h * 2 + n2;                  // (uniform)
        __builtin_amdgcn_sched_barrier(0);          // (one channel fragment's weights at a time: 144 VGPRs are taken)
        bf16x8 wf[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) wf[ks] = *reinterpret_cast<const bf16x8*>(smem + ((wrd + (unsigned)(nf * (16 * WP))) & LDS_MASK) + ks * 64);
        const f32x4 sv = *reinterpret_cast<const f32x4*>(bn + nf * 16 + cq * 4);     // channel nf*16 + 4*cq + r
        const f32x4 hv = *reinterpret_cast<const f32x4*>(bn + CM + nf * 16 + cq * 4);
        f32x4 a = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ks], xf[ks], a, 0, 0, 0);
        f32x2 v0 = f32x2{a[0], a[1]} * f32x2{sv[0], sv[1]} + f32x2{hv[0], hv[1]};
        f32x2 v1 = f32x2{a[2], a[3]} * f32x2{sv[2], sv[3]} + f32x2{hv[2], hv[3]};
        const f32x2 t0 = v0 * alpha2, t1 = v1 * alpha2;
        uint2 o;
        o.x = pack2(fmaxf(t0[0], v0[0]), fmaxf(t0[1], v0[1]));
        o.y = pack2(fmaxf(t1[0], v1[0]), fmaxf(t1[1], v1[1]));
        if (EDGE && outside) o = uint2{0u, 0u};
        *reinterpret_cast<uint2*>(smem + ((M_OFF + (unsigned)(q * MP + cq * 8 + nf * 32)) & LDS_MASK)) = o;
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
  fetch(b, ty, tx, X_OFF);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();                                  // weights, tables, the first input tile
  unsigned cur = 0;
#ifdef B64_PROBE
  long long pacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  long long plast = (long long)__builtin_amdgcn_s_memtime();
#endif
  for (int t = blockIdx.x; t < p.tiles; t += gridDim.x) {
    const unsigned xcur = X_OFF + cur * X_BYTES, xnxt = X_OFF + (cur ^ 1u) * X_BYTES;
    // the tile after this one
    int nb = b + p.gb, nty = ty + p.gy, ntx = tx + p.gx;
    if (ntx >= p.tilesX) { ntx -= p.tilesX; ++nty; }
    if (nty >= p.tilesY) { nty -= p.tilesY; ++nb; }
    const bool more = t + (int)gridDim.x < p.tiles;
    if (more) fetch(nb, nty, ntx, xnxt);            // the whole next tile: in flight under this patch (nobody reads that buffer)
    B64_STAMP(0);
    const unsigned edge_mask = (ty == 0 ? 1u : 0u) | (ty == p.tilesY - 1 ? 2u : 0u) | (tx == 0 ? 4u : 0u) |
                               (tx == p.tilesX - 1 ? 8u : 0u);
    if (edge_mask) phaseA(std::true_type{}, edge_mask, xcur);
    else phaseA(std::false_type{}, 0u, xcur);
    B64_STAMP(1);
    __syncthreads();                                // the intermediate tile is complete
    B64_STAMP(2);
    B64_STAMP(3);
    // ---- phase B: 4 patch rows x 16 columns x 32 channels per wave
    f32x4 acc[4][2];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int nf = 0; nf < 2; ++nf) acc[g][nf] = f32x4{0.f, 0.f, 0.f, 0.f};
    // 18 steps (tap, 32-channel slice) of 4 fragment reads + 8 MFMAs.  The registers are the weights': no second set of
    // fragments to read ahead into -- a fragment register is refilled for the NEXT step right after the two MFMAs that
    // consumed it, in exactly this order (the read then has the other three groups' MFMAs to land)
    auto xaddr = [&](int step, int g) {
      const int tap = step >> 1, ks = step & 1, kh = tap / 3, kw = tap % 3;
      return reinterpret_cast<const bf16x8*>(smem + (xb & LDS_MASK) + ((g + kh) * TW + kw) * MP + ks * 64);
    };
    bf16x8 xf[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) xf[g] = *xaddr(0, g);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int step = 0; step < 18; ++step) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
#pragma unroll
        for (int nf = 0; nf < 2; ++nf)
          acc[g][nf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wBr[step >> 1][step & 1][nf], xf[g], acc[g][nf], 0, 0, 0);
        if (step + 1 < 18) xf[g] = *xaddr(step + 1, g);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    B64_STAMP(4);
    __syncthreads();                                // everyone is done with the intermediate tile: its space is the staging now
    B64_STAMP(5);
    f32x2 s2[2][2], h2[2][2];
#pragma unroll
    for (int nf = 0; nf < 2; ++nf) {
      const f32x4 sv = *reinterpret_cast<const f32x4*>(bn + 2 * CM + cqt * 32 + nf * 16 + cq * 4);
      const f32x4 hv = *reinterpret_cast<const f32x4*>(bn + 2 * CM + CO + cqt * 32 + nf * 16 + cq * 4);
      s2[nf][0] = f32x2{sv[0], sv[1]}; s2[nf][1] = f32x2{sv[2], sv[3]};
      h2[nf][0] = f32x2{hv[0], hv[1]}; h2[nf][1] = f32x2{hv[2], hv[3]};
    }
    char* const sw0 = smem + M_OFF + wave * (2 * 16 * SROW);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      char* const sw = sw0 + (g & 1) * (16 * SROW);
      // the residual: input-tile pixel (ph*4 + g + 1, frow + 1), channels cqt*32 + nf*16 + 4 cq .. +3
      const int rp = (ph * 4 + g + 1) * TW + frow + 1;          // tile pixel of the residual
#pragma unroll
      for (int nf = 0; nf < 2; ++nf) {
        // channels cqt*32 + nf*16 + 4 cq .. +3 = 8 bytes at (cq & 1) * 8 of chunk cqt*4 + nf*2 + (cq >> 1)
        const uint2 rr = *reinterpret_cast<const uint2*>(smem + xcur + rp * XP + (((cqt * 4 + nf * 2 + (cq >> 1)) ^ (rp & 15)) << 4) + (cq & 1) * 8);
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
      // 16 pixels x 64 bytes (this wave's channel quarter): one 16-byte chunk per lane
      const int px = lane >> 2, ch = lane & 3;
      const size_t m = ((size_t)b * p.H + ty * PH + ph * 4 + g) * p.W + tx * PW + px;
      *reinterpret_cast<uint4*>(p.y + m * CO + cqt * 32 + ch * 8) = *reinterpret_cast<const uint4*>(sw + px * SROW + ch * 16);
    }
    B64_STAMP(6);
    b = nb; ty = nty; tx = ntx;
    cur ^= 1u;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's share of the next input tile has landed (and its stores left)
    __syncthreads();                                // staging reads done; everyone's share of the next input tile is in LDS
    B64_STAMP(7);
  }
#ifdef B64_PROBE
  if (lane == 0)
    for (int k = 0; k < 8; ++k) p.probe[(blockIdx.x * NW + wave) * 8 + k] = pacc[k];
#endif
}

}  // namespace

#ifdef B64_PROBE
extern long long* g_b64_probe;
#endif

extern "C" int disyolo_block64_fused_ok(int B, int H, int W, int C0) {
  return (C0 == CI && B > 0 && H > 0 && W > 0 && H % PH == 0 && W % PW == 0 && (int64_t)B * H * W * CI * 2 < (1LL << 31)) ? 1 : 0;   // (32-bit DMA offsets, bit 31 = out of range)
}

extern "C" int disyolo_block64_fused_fwd(const void* x, const void* wA, const float* scaleA, const float* shiftA, const void* wB,
                                         const float* scaleB, const float* shiftB, void* y, int B, int H, int W, int C0, float alpha,
                                         void* stream) {
  DY_REQUIRE(x && wA && scaleA && shiftA && wB && scaleB && shiftB && y, "block64_fused: null pointer");
  DY_REQUIRE(disyolo_block64_fused_ok(B, H, W, C0) == 1, "block64_fused: covers [128 -> 64 -> 128 + residual], H %% %d == 0, W %% %d == 0",
             PH, PW);
  DY_RECORD_OR_RUN([=](void* s) { return disyolo_block64_fused_fwd(x, wA, scaleA, shiftA, wB, scaleB, shiftB, y, B, H, W, C0, alpha, s); });
  B64Params p;
  p.x = (const bf16*)x; p.wA = (const bf16*)wA; p.scA = scaleA; p.shA = shiftA;
  p.wB = (const bf16*)wB; p.scB = scaleB; p.shB = shiftB; p.y = (bf16*)y;
  p.B = B; p.H = H; p.W = W;
  p.tilesY = H / PH; p.tilesX = W / PW; p.tiles = B * p.tilesY * p.tilesX;
  p.alpha = alpha;
#ifdef B64_PROBE
  p.probe = g_b64_probe;
#endif
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&block64_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    attr_set = true;
  }
  static const int ncu = [] {
    int dev = 0, n = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
    return n > 0 ? n : 256;
  }();
  int grid = ncu;                                   // one persistent block per CU
  if (grid > p.tiles) grid = p.tiles;
  const int tpi = p.tilesY * p.tilesX;
  p.gb = grid / tpi;
  p.gy = (grid % tpi) / p.tilesX;
  p.gx = grid % p.tilesX;
  hipLaunchKernelGGL(block64_kernel, dim3(grid), dim3(NW * 64), LDS_BYTES, (hipStream_t)stream, p);
  DY_CHECK_LAUNCH();
  return DISYOLO_OK;
}
