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
#include "common.h"
#include "runtime.h"

namespace {

struct ConvParams {
  const bf16* x0;
  const bf16* x1;
  const bf16* w;
  const float* scale;
  const float* shift;
  const bf16* residual;
  void* y;
  float* stats;
  int B, H, W, C0, C1, Cin;
  int Ho, Wo, Cout;
  int ks, stride, pad_t, pad_l, div;
  int M, K, nk;
  int tilesM, tilesN;
  int flags;
  float alpha;
};

// swizzle of the 16-byte chunk index inside a 64-byte LDS row
__device__ __forceinline__ int swz(int row, int chunk) {
  return chunk ^ ((0x78 >> (((row >> 2) & 3) * 2)) & 3);
}

template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(WM* WN * 64) void conv_igemm_kernel(ConvParams p) {
  constexpr int T = WM * WN * 64;
  constexpr int WTM = BM / WM, WTN = BN / WN;
  constexpr int MI = WTM / 16, NI = WTN / 16;
  constexpr int NA = (BM * 4 + T - 1) / T;
  constexpr int NB = (BN * 4 + T - 1) / T;
  constexpr int A_BYTES = BM * 64, B_BYTES = BN * 64;
  static_assert(WTM % 16 == 0 && WTN % 16 == 0, "wave tile must be a multiple of 16");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sA = smem;                 // [2][BM][64 B]
  char* sB = smem + 2 * A_BYTES;   // [2][BN][64 B]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;

  // XCD-aware tile order: blocks b and b+8 share an XCD (and its L2); give each XCD a
  // contiguous run of tiles, n-tile fastest, so the blocks that re-read one pixel panel
  // (and neighbouring halo rows) hit the same L2.  Bijective for any grid size.
  int tile;
  {
    const int nblk = gridDim.x, bid = blockIdx.x;
    const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, loc = bid >> 3;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
  }
  const int mt = tile / p.tilesN, nt = tile - mt * p.tilesN;
  const int m0 = mt * BM, n0 = nt * BN;

  // ---- per-thread gather state for the A (pixel) tile ----
  const int kc = tid & 3;  // 16-byte chunk (8 channels) inside the 32-wide K slice
  int a_iy0[NA], a_ix0[NA], a_b[NA];
  bool a_ok[NA];
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const int row = (tid >> 2) + i * (T / 4);
    const int m = m0 + row;
    a_ok[i] = (row < BM) && (m < p.M);
    const int mm = a_ok[i] ? m : 0;
    const int hw = p.Ho * p.Wo;
    const int b = mm / hw;
    const int rem = mm - b * hw;
    const int yo = rem / p.Wo;
    const int xo = rem - yo * p.Wo;
    a_b[i] = b;
    a_iy0[i] = yo * p.stride - p.pad_t;
    a_ix0[i] = xo * p.stride - p.pad_l;
  }
  bool b_ok[NB];
  const bf16* b_ptr[NB];
#pragma unroll
  for (int i = 0; i < NB; ++i) {
    const int row = (tid >> 2) + i * (T / 4);
    const int n = n0 + row;
    b_ok[i] = (row < BN) && (n < p.Cout);
    b_ptr[i] = p.w + (size_t)(b_ok[i] ? n : 0) * p.K + kc * 8;
  }

  uint4 ra[NA], rb[NB];
  int kh = 0, kw = 0, ci0 = 0, k0 = 0;  // wave-uniform K cursor of the tile being loaded

  auto load_tile = [&]() {
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      uint4 v = make_uint4(0, 0, 0, 0);
      if (a_ok[i]) {
        int iy = a_iy0[i] + kh, ix = a_ix0[i] + kw;
        bool ok = true;
        if (p.div > 1) {
          ok = ((iy | ix) >= 0) && (iy % p.div == 0) && (ix % p.div == 0);
          iy /= p.div;
          ix /= p.div;
        }
        ok = ok && ((unsigned)iy < (unsigned)p.H) && ((unsigned)ix < (unsigned)p.W);
        if (ok) {
          const bf16* src;
          if (ci0 < p.C0) {
            src = p.x0 + ((size_t)(a_b[i] * p.H + iy) * p.W + ix) * p.C0 + ci0 + kc * 8;
          } else {
            const int H1 = p.H >> 1, W1 = p.W >> 1;
            src = p.x1 + ((size_t)(a_b[i] * H1 + (iy >> 1)) * W1 + (ix >> 1)) * p.C1 + (ci0 - p.C0) + kc * 8;
          }
          v = *reinterpret_cast<const uint4*>(src);
        }
      }
      ra[i] = v;
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      uint4 v = make_uint4(0, 0, 0, 0);
      if (b_ok[i]) v = *reinterpret_cast<const uint4*>(b_ptr[i] + k0);
      rb[i] = v;
    }
    // advance the K cursor by one 32-wide slice
    k0 += 32;
    ci0 += 32;
    if (ci0 >= p.Cin) {
      ci0 = 0;
      if (++kw == p.ks) {
        kw = 0;
        ++kh;
      }
    }
  };
  auto store_tile = [&](int buf) {
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int row = (tid >> 2) + i * (T / 4);
      if (row < BM) *reinterpret_cast<uint4*>(sA + buf * A_BYTES + row * 64 + swz(row, kc) * 16) = ra[i];
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int row = (tid >> 2) + i * (T / 4);
      if (row < BN) *reinterpret_cast<uint4*>(sB + buf * B_BYTES + row * 64 + swz(row, kc) * 16) = rb[i];
    }
  };

  f32x4 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  load_tile();
  store_tile(0);
  __syncthreads();

  const int frow = lane & 15, fchunk = lane >> 4;
  for (int kt = 0; kt < p.nk; ++kt) {
    const int cur = kt & 1;
    const bool more = (kt + 1 < p.nk);
    if (more) load_tile();
    bf16x8 xf[MI], wf[NI];
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const int row = wm * WTM + i * 16 + frow;
      xf[i] = *reinterpret_cast<const bf16x8*>(sA + cur * A_BYTES + row * 64 + swz(row, fchunk) * 16);
    }
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int row = wn * WTN + j * 16 + frow;
      wf[j] = *reinterpret_cast<const bf16x8*>(sB + cur * B_BYTES + row * 64 + swz(row, fchunk) * 16);
    }
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], xf[i], acc[i][j], 0, 0, 0);
    if (more) store_tile(cur ^ 1);
    __syncthreads();
  }

  // ---- epilogue.  acc[i][j][r]: pixel m0 + wm*WTM + i*16 + (lane&15),
  //      channel n0 + wn*WTN + j*16 + 4*(lane>>4) + r ----
  const int px = lane & 15, cq = lane >> 4;

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
        if (px == 0) {
          const int nl = wn * WTN + j * 16 + cq * 4 + r;
          red[(wm * BN + nl) * 2 + 0] = s;
          red[(wm * BN + nl) * 2 + 1] = s2;
        }
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
        p.stats[((size_t)mt * p.Cout + n) * 2 + 0] = s;
        p.stats[((size_t)mt * p.Cout + n) * 2 + 1] = s2;
      }
    }
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
      if (m >= p.M) continue;
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        v[r] = acc[i][j][r] * sc[r] + sh[r];
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

struct TileCfg {
  int id, bm, bn, threads;
};
// id -> (BM, BN, WM, WN)
const TileCfg kTiles[] = {
    {1, 128, 128, 256},  // 2x2 waves, 64x64 per wave
    {2, 128, 64, 256},   // 2x2 waves, 64x32
    {3, 64, 128, 256},   // 2x2 waves, 32x64
    {4, 128, 32, 256},   // 4x1 waves, 32x32
    {5, 128, 16, 256},   // 4x1 waves, 32x16
    {6, 64, 64, 256},    // 2x2 waves, 32x32
    {7, 256, 64, 256},   // 4x1 waves, 64x64
};

template <int BM, int BN, int WM, int WN>
int launch(const ConvParams& p, hipStream_t s) {
  ConvParams q = p;
  q.tilesM = ceil_div(p.M, BM);
  q.tilesN = ceil_div(p.Cout, BN);
  const int grid = q.tilesM * q.tilesN;
  size_t lds = 2 * (size_t)(BM + BN) * 64;
  const size_t red = (size_t)WM * BN * 2 * sizeof(float);
  if (red > lds) lds = red;
  hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, WM, WN>), dim3(grid), dim3(WM * WN * 64), lds, s, q);
  DY_CHECK_LAUNCH();
  return DISYOLO_OK;
}

int pick_tile(const disyolo_conv_desc* d, int M) {
  if (d->tile > 0) return d->tile;
  const int N = d->Cout;
  if (N <= 16) return 5;
  if (N <= 32) return 4;
  if (N <= 64) return M >= 256 * 256 ? 7 : 2;
  // N >= 128: prefer 128x128 unless that leaves the 256 CUs under-filled
  const int t128 = ceil_div(M, 128) * ceil_div(N, 128);
  if (t128 >= 384) return 1;
  return 3;
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
  DY_REQUIRE(d->in_div >= 1, "conv: in_div must be >= 1");
  DY_REQUIRE(d->B > 0 && d->H > 0 && d->W > 0 && d->Ho > 0 && d->Wo > 0 && d->Cout > 0, "conv: bad sizes");
  DY_REQUIRE((int64_t)d->B * d->Ho * d->Wo < (1LL << 31), "conv: too many output pixels");
  DY_REQUIRE(d->x0 && d->w && d->y, "conv: null tensor pointer");
  return DISYOLO_OK;
}

}  // namespace

extern "C" int disyolo_conv2d_stats_rows(const disyolo_conv_desc* d) {
  if (!d) return DISYOLO_E_ARG;
  const int M = d->B * d->Ho * d->Wo;
  const int bm = tile_bm(pick_tile(d, M));
  if (bm == 0) return DISYOLO_E_ARG;
  return ceil_div(M, bm);
}

extern "C" int disyolo_conv2d_tile(const disyolo_conv_desc* d, int* bm, int* bn) {
  if (!d) return DISYOLO_E_ARG;
  const int id = pick_tile(d, d->B * d->Ho * d->Wo);
  for (const TileCfg& t : kTiles)
    if (t.id == id) {
      if (bm) *bm = t.bm;
      if (bn) *bn = t.bn;
      return id;
    }
  return DISYOLO_E_ARG;
}

extern "C" int disyolo_conv2d_fwd(const disyolo_conv_desc* d, void* stream) {
  int rc = validate(d);
  if (rc) return rc;
  DY_REQUIRE(!(d->flags & DISYOLO_CONV_STATS) || d->stats, "conv: STATS flag without stats buffer");
  {
    const disyolo_conv_desc c = *d;
    DY_RECORD_OR_RUN([c](void* s) { return disyolo_conv2d_fwd(&c, s); });
  }
  ConvParams p;
  p.x0 = (const bf16*)d->x0;
  p.x1 = (const bf16*)d->x1;
  p.w = (const bf16*)d->w;
  p.scale = d->scale;
  p.shift = d->shift;
  p.residual = (const bf16*)d->residual;
  p.y = d->y;
  p.stats = d->stats;
  p.B = d->B; p.H = d->H; p.W = d->W; p.C0 = d->C0; p.C1 = d->C1; p.Cin = d->C0 + d->C1;
  p.Ho = d->Ho; p.Wo = d->Wo; p.Cout = d->Cout;
  p.ks = d->ksize; p.stride = d->stride; p.pad_t = d->pad_t; p.pad_l = d->pad_l; p.div = d->in_div;
  p.M = d->B * d->Ho * d->Wo;
  p.K = d->ksize * d->ksize * p.Cin;
  p.nk = p.K / 32;
  p.flags = d->flags;
  p.alpha = d->alpha;
  p.tilesM = p.tilesN = 0;
  hipStream_t s = (hipStream_t)stream;
  switch (pick_tile(d, p.M)) {
    case 1: return launch<128, 128, 2, 2>(p, s);
    case 2: return launch<128, 64, 2, 2>(p, s);
    case 3: return launch<64, 128, 2, 2>(p, s);
    case 4: return launch<128, 32, 4, 1>(p, s);
    case 5: return launch<128, 16, 4, 1>(p, s);
    case 6: return launch<64, 64, 2, 2>(p, s);
    case 7: return launch<256, 64, 4, 1>(p, s);
    default: disyolo_set_error("conv: unknown tile id %d", d->tile); return DISYOLO_E_ARG;
  }
}
