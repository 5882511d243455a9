// conv1 + conv2 of the backbone in ONE launch, for the steps in which both are in inference mode (the locked backbone
// of training stage 1; every inference call):
//
//   act1 = leaky(bn1(conv3x3_s1(image)))          yolo/yolo3_net_pos.py:159-161   (3 -> 32, 576^2 -> 576^2)
//   act2 = leaky(bn2(conv3x3_s2(act1)))           yolo/yolo3_net_pos.py:165-167   (32 -> 64, 576^2 -> 288^2)
//
// act1 ("skip1", :163) has ONE consumer in the active m = 1/2 mask subnet configuration: conv2.  Stored, it is the
// largest tensor of the network (B x 576 x 576 x 32 bf16 = 170 MB at B = 8) -- written once, read once, 0.34 GB of HBM
// traffic for 8 % of one layer's FLOPs; the two kernels cost 83 + 92 us of a 4.3 ms training step (cold caches) and
// 224 + 260 us of a 5.8 ms inference batch, both at half their HBM bound.  Here act1 only ever exists as a 17 x 33-pixel
// tile in LDS.
//
// Two persistent 4-wave blocks per CU (67 KB of LDS each) walk over 8 x 16-pixel patches of act2; while one block sits in
// a barrier or in the VALU-heavy phase A, the other has the matrix cores:
//   park     the 19 x 35 x 3 f32 image patch of the NEXT tile (fetched into registers one tile ahead) is split ONCE per
//            value into bf16 hi + bf16 lo (x = hi + lo + O(2^-16 x)) and parked in LDS as one dword (hi << 16 | lo);
//   phase A  act1 on the 17 x 33 pixels conv2 needs, on the bf16 matrix cores with split operands: acc += w_lo*x_hi +
//            w_hi*x_lo + w_hi*x_hi -- three v_mfma_f32_16x16x32_bf16 per 16 pixels x 16 channels, relative error 2^-16
//            per product against exact f32 (the result is rounded to bf16, 2^-9, next) at 1/5 of the f32 MFMA's cycles.
//            K = 27 is laid out for the LDS reads, not in (kh, kw, c) order: k-chunk cq < 3 of a lane = the first 8 of
//            the 9 contiguous (kw, c) values of image row ay + cq (two v_perm per dword pair make the hi / lo
//            fragments), chunk 3 = the ninth value (kw = 2, c = 2) of the three rows, read from a column-major copy of
//            the patch's channel 2; the remaining five k are zero weights over finite junk.  Folded BN + leaky; pixels
//            beyond the image zeroed (they are conv2's SAME padding; only the bottom / right border tiles pay for the
//            test).  The tile is stored as two column-parity planes [ay][ax / 2] of 80-byte pixel rows (64 + 16 pad):
//            conv2's stride-2 taps then read 16 CONSECUTIVE rows per wave -- conflict-free ds_read_b128;
//   phase B  conv2 from that tile.  Each wave owns 32 of the 64 output channels and 4 of the 8 patch rows and keeps its
//            weights (9 taps x 32 channels x 32 k = 72 VGPRs) in registers for the whole launch: per tap 4 LDS reads
//            (one base register, immediate offsets) feed 8 MFMAs.  Folded BN + leaky; the bf16 rows leave through a
//            per-wave staging tile as 64-byte half lines.
// Per patch a CU reads 8 KB and writes 16 KB for 504 MFMAs.
#include <utility>
#include "common.h"
#include "runtime.h"

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int PH2 = 8, PW2 = 16;                    // act2 patch
constexpr int AH = 2 * PH2 + 1, AW = 2 * PW2 + 1;   // act1 tile: 17 x 33
constexpr int EVW = PW2 + 1, ODW = PW2;             // columns of the even / odd plane
constexpr int NEV = AH * EVW;                       // 289 even-plane rows, then 272 odd-plane rows
constexpr int NA = AH * AW;                         // 561 act1 pixels
constexpr int NW = 4;
constexpr int NAF = (NA + 15) / 16;                 // 36 fragments of 16 = 9 per wave
constexpr int FPW = NAF / NW;
constexpr int AP = 80;                              // bytes per act1 pixel row
constexpr int IH = AH + 2, IW = AW + 2;             // image patch: 19 x 35
constexpr int IROW = IW * 3;                        // 105 values per patch row
constexpr int IPITCH = 108;                         // dwords per parked row
constexpr int C2P = 25;                             // dwords per column of the channel-2 copy (19 used, the rest stay 0;
                                                    // odd: 16 lanes two columns apart land in 16 different banks)
constexpr int NIMG = IH * IROW;                     // 1995
constexpr int IPT = (NIMG + NW * 64 - 1) / (NW * 64);   // 8 values per thread
constexpr int SROW = 80;                            // staging: 16 pixels x (64 B + pad)

constexpr int A1_BYTES = NAF * 16 * AP;             // 46080 (rows 561..575: scratch of the last fragment)
constexpr int IM_BYTES = IH * IPITCH * 4;           // 8208
constexpr int C2_BYTES = (IW * C2P * 4 + 15) / 16 * 16;   // 3504 (what follows stays 16-byte aligned)
constexpr int DUMMY_BYTES = NW * 64 * 4;            // a private dword per thread for the writes that have no target
constexpr int STG_BYTES = NW * 2 * 16 * SROW;       // two staging tiles per wave
constexpr int BN2_BYTES = 2 * 64 * 4;               // conv2's folded scale, shift
constexpr int LDS_BYTES = A1_BYTES + IM_BYTES + C2_BYTES + DUMMY_BYTES + STG_BYTES + BN2_BYTES;
static_assert(NAF % NW == 0, "fragments must divide over the waves");
static_assert(A1_BYTES % 16 == 0 && IM_BYTES % 16 == 0 && C2_BYTES % 16 == 0 && DUMMY_BYTES % 16 == 0 && STG_BYTES % 16 == 0, "16-byte LDS regions");
static_assert(2 * LDS_BYTES <= 160 * 1024, "two blocks per CU");

struct F2Params {
  const float* img;
  const float* w1;        // HWIO [3][3][3][32] f32
  const float* sc1;
  const float* sh1;
  const bf16* w2;         // packed [64][9 * 32]
  const float* sc2;
  const float* sh2;
  bf16* y;                // [B][H/2][W/2][64]
  int B, H, W, tilesY, tilesX, tiles;
  int gb, gy, gx;         // gridDim.x tiles as (images, tile rows, tile columns)
  float alpha;
#ifdef F2_PROBE
  long long* probe;       // tools/probe_conv12.hip: [block][wave][8] cycles per phase, summed over the block's tiles
#endif
};

#ifdef F2_PROBE
#define F2_STAMP(k)                                              \
  do {                                                           \
    const long long now_ = (long long)__builtin_amdgcn_s_memtime(); \
    pacc[k] += now_ - plast;                                     \
    plast = now_;                                                \
  } while (0)
#else
#define F2_STAMP(k) do { } while (0)
#endif

__device__ __forceinline__ unsigned split_pack(float x) {
  const bf16 h = (bf16)x;
  const bf16 l = (bf16)(x - (float)h);
  return ((unsigned)__builtin_bit_cast(unsigned short, h) << 16) | (unsigned)__builtin_bit_cast(unsigned short, l);
}

__global__ __launch_bounds__(NW * 64, 2) void conv12_fused_kernel(F2Params p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const a1l = smem;
  char* const iml = smem + A1_BYTES;                // image dwords, then the channel-2 copy, then the dummies
  constexpr unsigned IM_OFF = A1_BYTES;             // (the dynamic LDS segment starts at address 0: offsets from smem
                                                    //  are kept in 16 bits and fit the DS immediates)
  char* const stg = smem + A1_BYTES + IM_BYTES + C2_BYTES + DUMMY_BYTES;
  float* const bn2 = reinterpret_cast<float*>(stg + STG_BYTES);
  constexpr unsigned C2_OFF = IM_BYTES, DUMMY_OFF = IM_BYTES + C2_BYTES;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int frow = lane & 15, cq = lane >> 4;
  const int hh = wave & 1, ph = wave >> 1;          // phase B: channel half, patch-row half
  const int H2 = p.H >> 1, W2 = p.W >> 1;

  // ---- once per block
  for (int c = tid; c < C2_BYTES / 4; c += NW * 64) reinterpret_cast<unsigned*>(iml + C2_OFF)[c] = 0u;
  if (tid < 64) {
    bn2[tid] = p.sc2[tid];
    bn2[64 + tid] = p.sh2[tid];
  }
  // conv2's weights of this wave's 32 channels: A fragments (channel hh*32 + nf*16 + frow, k = tap*32 + 8 cq .. +7)
  bf16x8 w2r[9][2];
#pragma unroll
  for (int tap = 0; tap < 9; ++tap)
#pragma unroll
    for (int nf = 0; nf < 2; ++nf)
      w2r[tap][nf] = *reinterpret_cast<const bf16x8*>(p.w2 + (size_t)(hh * 32 + nf * 16 + frow) * 288 + tap * 32 + cq * 8);
  // conv1's weights as A fragments, hi / lo, in the k order of the header: chunk cq < 3 = (kh = cq, e = 0..7),
  // chunk 3 = (kh = 0..2, e = 8), e = kw * 3 + c; HWIO index (kh * 9 + e) * 32 + channel
  bf16x8 w1h[2], w1l[2];
#pragma unroll
  for (int n = 0; n < 2; ++n)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int kk = cq < 3 ? cq * 9 + i : i * 9 + 8;
      const bool real = cq < 3 || i < 3;
      const float w = real ? p.w1[(real ? kk : 0) * 32 + n * 16 + frow] : 0.f;
      const bf16 h = (bf16)w;
      w1h[n][i] = h;
      w1l[n][i] = (bf16)(w - (float)h);
    }
  // epilogue constants of this lane's channels (accumulator layout: channel nf*16 + 4*cq + r)
  f32x2 s1[2][2], h1[2][2];
#pragma unroll
  for (int n = 0; n < 2; ++n)
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int c1 = n * 16 + cq * 4 + 2 * r;
      s1[n][r] = f32x2{p.sc1[c1], p.sc1[c1 + 1]};
      h1[n][r] = f32x2{p.sh1[c1], p.sh1[c1 + 1]};
    }
  const f32x2 alpha2 = f32x2{p.alpha, p.alpha};

  // phase A, fragment i of this wave = plane rows (wave + NW*i)*16 + frow: where its 8 k values start in the parked
  // patch (LDS address), and whether the pixel is in the tile's last row / last column (bits 2i, 2i + 1 of aflag)
  unsigned aoff[FPW], aflag = 0u;
#pragma unroll
  for (int i = 0; i < FPW; ++i) {
    const int pr = (wave + NW * i) * 16 + frow;
    int ay = 0, ax = 0;
    if (pr < NEV) { ay = pr / EVW; ax = 2 * (pr - ay * EVW); }
    else if (pr < NA) { const int r = pr - NEV; ay = r / ODW; ax = 2 * (r - ay * ODW) + 1; }
    const unsigned off = cq < 3 ? (unsigned)(((ay + cq) * IPITCH + ax * 3) * 4) : C2_OFF + (unsigned)(((ax + 2) * C2P + ay) * 4);
    aoff[i] = IM_OFF + off;
    aflag |= ((ay == AH - 1 ? 1u : 0u) | (ax == AW - 1 ? 2u : 0u)) << (2 * i);
  }
  const unsigned awr = (unsigned)((wave * 16 + frow) * AP + cq * 8);   // + i * NW*16*AP + n * 32

  // phase B: 16 lanes = 16 consecutive output columns; even plane row (2 opy + kh) * 17 + opx (+1 for kw = 2), odd plane
  // row 289 + (2 opy + kh) * 16 + opx, opy = ph * 4 + g
  const unsigned xb_ev = (unsigned)((ph * 8 * EVW + frow) * AP + cq * 16);
  const unsigned xb_od = (unsigned)((NEV + ph * 8 * ODW + frow) * AP + cq * 16);

  // image patch: value e = i * 256 + tid of the [19][105] patch -> where it comes from and where it is parked
  int srcoff[IPT];
  unsigned dpk[IPT];                                // parked at (low 16 bits) and, channel 2, at (high 16 bits)
#pragma unroll
  for (int i = 0; i < IPT; ++i) {
    const int e = i * (NW * 64) + tid;
    const int r = e / IROW, c = e - r * IROW;
    const bool real = e < NIMG;
    srcoff[i] = real ? r * p.W * 3 + c : 0;
    const unsigned dmain = real ? (unsigned)((r * IPITCH + c) * 4) : DUMMY_OFF + tid * 4;
    const unsigned dc2 = (real && c % 3 == 2) ? C2_OFF + (unsigned)(((c / 3) * C2P + r) * 4) : DUMMY_OFF + tid * 4;
    dpk[i] = (IM_OFF + dmain) | ((IM_OFF + dc2) << 16);
  }
  float pre[IPT];
  auto fetch = [&](int b, int ty, int tx) {
    const int iy0 = 2 * ty * PH2 - 1, ix0 = 2 * tx * PW2 - 1;
    const int base = ((b * p.H + iy0) * p.W + ix0) * 3;
    const bool border = ty == 0 || tx == 0 || ty == p.tilesY - 1 || tx == p.tilesX - 1;   // (uniform)
    if (!border) {
#pragma unroll
      for (int i = 0; i < IPT; ++i) pre[i] = p.img[base + srcoff[i]];
    } else {
#pragma unroll
      for (int i = 0; i < IPT; ++i) {
        const int e = i * (NW * 64) + tid;
        const int r = e / IROW, c = e - r * IROW;
        const bool ok = e < NIMG && (unsigned)(iy0 + r) < (unsigned)p.H && (unsigned)(ix0 + c / 3) < (unsigned)p.W;
        const float v = p.img[ok ? base + srcoff[i] : 0];   // (always a load: the eight stay in flight together)
        pre[i] = ok ? v : 0.f;
      }
    }
  };
  auto park = [&]() {
#pragma unroll
    for (int i = 0; i < IPT; ++i) {
      const unsigned v = split_pack(pre[i]);
      unsigned d = dpk[i];
      asm volatile("" : "+v"(d));                   // (unpacked per tile, not hoisted into 16 more registers)
      *reinterpret_cast<unsigned*>(smem + (d & 0xffffu)) = v;
      *reinterpret_cast<unsigned*>(smem + (d >> 16)) = v;
    }
  };

  auto phaseA = [&](auto edge_tag, unsigned edge_mask) {
    constexpr bool EDGE = decltype(edge_tag)::value;
    unsigned d[2][8];
    auto load = [&](int i, unsigned* dst) {
      const unsigned* px = reinterpret_cast<const unsigned*>(smem + (aoff[i] & 0xffffu));   // (< 64 KB: immediates fold)
#pragma unroll
      for (int j = 0; j < 8; ++j) dst[j] = px[j];
    };
    load(0, d[0]);
#pragma unroll
    for (int i = 0; i < FPW; ++i) {
      if (i + 1 < FPW) load(i + 1, d[(i + 1) & 1]);   // the next fragment's reads fly under this one's MFMAs
      __builtin_amdgcn_sched_barrier(0);
      const unsigned* c = d[i & 1];
      uint4 uh, ul;
      uh.x = __builtin_amdgcn_perm(c[1], c[0], 0x07060302u); ul.x = __builtin_amdgcn_perm(c[1], c[0], 0x05040100u);
      uh.y = __builtin_amdgcn_perm(c[3], c[2], 0x07060302u); ul.y = __builtin_amdgcn_perm(c[3], c[2], 0x05040100u);
      uh.z = __builtin_amdgcn_perm(c[5], c[4], 0x07060302u); ul.z = __builtin_amdgcn_perm(c[5], c[4], 0x05040100u);
      uh.w = __builtin_amdgcn_perm(c[7], c[6], 0x07060302u); ul.w = __builtin_amdgcn_perm(c[7], c[6], 0x05040100u);
      const bf16x8 xh = __builtin_bit_cast(bf16x8, uh), xl = __builtin_bit_cast(bf16x8, ul);
      const bool outside = EDGE && (aflag & (edge_mask << (2 * i))) != 0u;
#pragma unroll
      for (int n = 0; n < 2; ++n) {
        f32x4 a = f32x4{0.f, 0.f, 0.f, 0.f};
        a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1l[n], xh, a, 0, 0, 0);
        a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1h[n], xl, a, 0, 0, 0);
        a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1h[n], xh, a, 0, 0, 0);
        f32x2 v0 = f32x2{a[0], a[1]} * s1[n][0] + h1[n][0];
        f32x2 v1 = f32x2{a[2], a[3]} * s1[n][1] + h1[n][1];
        const f32x2 t0 = v0 * alpha2, t1 = v1 * alpha2;
        uint2 o;
        o.x = pack2(fmaxf(t0[0], v0[0]), fmaxf(t0[1], v0[1]));
        o.y = pack2(fmaxf(t1[0], v1[0]), fmaxf(t1[1], v1[1]));
        if (EDGE && outside) o = uint2{0u, 0u};
        *reinterpret_cast<uint2*>(a1l + awr + i * (NW * 16 * AP) + n * 32) = o;
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
  fetch(b, ty, tx);
  __syncthreads();                                  // the cleared channel-2 copy
  park();
  __syncthreads();
#ifdef F2_PROBE
  long long pacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  long long plast = (long long)__builtin_amdgcn_s_memtime();
#endif
  for (int t = blockIdx.x; t < p.tiles; t += gridDim.x) {
    // the tile after this one
    int nb = b + p.gb, nty = ty + p.gy, ntx = tx + p.gx;
    if (ntx >= p.tilesX) { ntx -= p.tilesX; ++nty; }
    if (nty >= p.tilesY) { nty -= p.tilesY; ++nb; }
    const bool more = t + (int)gridDim.x < p.tiles;
    if (more) fetch(nb, nty, ntx);                  // in flight under phase A
    F2_STAMP(0);
    const unsigned edge_mask = (ty == p.tilesY - 1 ? 1u : 0u) | (tx == p.tilesX - 1 ? 2u : 0u);
    if (edge_mask) phaseA(std::true_type{}, edge_mask);
    else phaseA(std::false_type{}, 0u);
    F2_STAMP(1);
    __syncthreads();                                // the act1 tile is complete; everyone is done with the image patch
    F2_STAMP(2);
    if (more) park();                               // (visible after the barrier at the end of the tile)
    F2_STAMP(3);
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
      for (int g = 0; g < 4; ++g) {
        const char* src = (kw == 1) ? a1l + xb_od + ((2 * g + kh) * ODW) * AP
                                    : a1l + xb_ev + ((2 * g + kh) * EVW + (kw == 2 ? 1 : 0)) * AP;
        xf[g] = *reinterpret_cast<const bf16x8*>(src);
      }
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int nf = 0; nf < 2; ++nf)
          acc[g][nf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w2r[tap][nf], xf[g], acc[g][nf], 0, 0, 0);
    }
    F2_STAMP(4);
    char* const sw0 = stg + wave * (2 * 16 * SROW);
    f32x2 s2[2][2], h2[2][2];
#pragma unroll
    for (int nf = 0; nf < 2; ++nf) {
      const f32x4 sv = *reinterpret_cast<const f32x4*>(bn2 + hh * 32 + nf * 16 + cq * 4);
      const f32x4 hv = *reinterpret_cast<const f32x4*>(bn2 + 64 + hh * 32 + nf * 16 + cq * 4);
      s2[nf][0] = f32x2{sv[0], sv[1]}; s2[nf][1] = f32x2{sv[2], sv[3]};
      h2[nf][0] = f32x2{hv[0], hv[1]}; h2[nf][1] = f32x2{hv[2], hv[3]};
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      char* const sw = sw0 + (g & 1) * (16 * SROW);
#pragma unroll
      for (int nf = 0; nf < 2; ++nf) {
        f32x2 v0 = f32x2{acc[g][nf][0], acc[g][nf][1]} * s2[nf][0] + h2[nf][0];
        f32x2 v1 = f32x2{acc[g][nf][2], acc[g][nf][3]} * s2[nf][1] + h2[nf][1];
        const f32x2 t0 = v0 * alpha2, t1 = v1 * alpha2;
        uint2 o;
        o.x = pack2(fmaxf(t0[0], v0[0]), fmaxf(t0[1], v0[1]));
        o.y = pack2(fmaxf(t1[0], v1[0]), fmaxf(t1[1], v1[1]));
        *reinterpret_cast<uint2*>(sw + frow * SROW + nf * 32 + cq * 8) = o;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      // 16 pixels x 64 bytes (this wave's channel half): one 16-byte chunk per lane
      const int px = lane >> 2, ch = lane & 3;
      const size_t m = ((size_t)b * H2 + ty * PH2 + ph * 4 + g) * W2 + tx * PW2 + px;
      *reinterpret_cast<uint4*>(p.y + m * 64 + hh * 32 + ch * 8) = *reinterpret_cast<const uint4*>(sw + px * SROW + ch * 16);
    }
    b = nb; ty = nty; tx = ntx;
    F2_STAMP(5);
    __syncthreads();                                // everyone is done with the act1 tile; the parked image is visible
    F2_STAMP(6);
  }
#ifdef F2_PROBE
  if (lane == 0)
    for (int k = 0; k < 8; ++k) p.probe[(blockIdx.x * NW + wave) * 8 + k] = pacc[k];
#endif
}

}  // namespace

#ifdef F2_PROBE
extern long long* g_f2_probe;
#endif

extern "C" int disyolo_conv12_fused_ok(int B, int H, int W) {
  return (B > 0 && H > 0 && W > 0 && (H / 2) % PH2 == 0 && (W / 2) % PW2 == 0 && H % 2 == 0 && W % 2 == 0 &&
          // 32-bit byte offsets: the f32 image (12 B / pixel) and the bf16 output (64 channels at half resolution)
          (int64_t)B * H * W * 12 < (1LL << 31) && (int64_t)B * (H / 2) * (W / 2) * 128 < (1LL << 31)) ? 1 : 0;
}

extern "C" int disyolo_conv12_fused_fwd(const float* images, const float* w1_hwio, const float* scale1, const float* shift1,
                                        const void* w2_packed, const float* scale2, const float* shift2, void* y_bf16, int B,
                                        int H, int W, float alpha, void* stream) {
  DY_REQUIRE(images && w1_hwio && scale1 && shift1 && w2_packed && scale2 && shift2 && y_bf16, "conv12_fused: null pointer");
  DY_REQUIRE(disyolo_conv12_fused_ok(B, H, W) == 1, "conv12_fused: H/2 must be a multiple of %d, W/2 of %d", PH2, PW2);
  DY_RECORD_OR_RUN([=](void* s) {
    return disyolo_conv12_fused_fwd(images, w1_hwio, scale1, shift1, w2_packed, scale2, shift2, y_bf16, B, H, W, alpha, s);
  });
  F2Params p;
  p.img = images; p.w1 = w1_hwio; p.sc1 = scale1; p.sh1 = shift1;
  p.w2 = (const bf16*)w2_packed; p.sc2 = scale2; p.sh2 = shift2; p.y = (bf16*)y_bf16;
  p.B = B; p.H = H; p.W = W;
  p.tilesY = (H / 2) / PH2; p.tilesX = (W / 2) / PW2; p.tiles = B * p.tilesY * p.tilesX;
  p.alpha = alpha;
#ifdef F2_PROBE
  p.probe = g_f2_probe;
#endif
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv12_fused_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                              LDS_BYTES);
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
  hipLaunchKernelGGL(conv12_fused_kernel, dim3(grid), dim3(NW * 64), LDS_BYTES, (hipStream_t)stream, p);
  DY_CHECK_LAUNCH();
  return DISYOLO_OK;
}
