// Weight gradient of the implicit-GEMM convolution on gfx950.
//
//   dw[(kh,kw,ci)][co] = sum_m xcol[m][(kh,kw,ci)] * dy[m][co]       (f32, HWIO order)
//
// i.e. TF autodiff of tf.nn.conv2d wrt its filter (train_yolo3_mask.py:55 minimize()).
// GEMM view: rows = K index, cols = Cout, reduction over the M = B*Ho*Wo output pixels.
// Both operands live in memory with the *reduction* index (pixel) as the slow dimension
// (NHWC), so the LDS tiles are [pixel][k] / [pixel][co] images read back with the gfx950
// transposing load ds_read_b64_tr_b16 straight into v_mfma_f32_16x16x32_bf16 fragments.
// The pixel range is split over blockIdx.z; each split writes an f32 slab and a second
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
};

typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

// 32-byte-unit swizzle inside one LDS row of `units` units
__device__ __forceinline__ int swz_u(int row, int u, int units) {
  const int h = (row & 3) | (((row >> 3) & 1) << 2);
  return u ^ (h & (units - 1));
}

template <int BN>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(WgradParams p) {
  constexpr int BM = 128;  // K-index rows per block
  constexpr int WM = 2, WN = 2;
  constexpr int WTM = BM / WM, WTN = BN / WN;
  constexpr int MI = WTM / 16, NI = WTN / 16;
  constexpr int XCH = BM / 8, YCH = BN / 8;          // 16-byte chunks per pixel row
  constexpr int NX = 32 * XCH / 256, NY = (32 * YCH + 255) / 256;
  constexpr int XP = BM * 2, YP = BN * 2;            // row pitch in bytes
  constexpr int X_BYTES = 32 * XP, Y_BYTES = 32 * YP;
  static_assert(WTN % 16 == 0, "BN must be a multiple of 32");

  __shared__ __attribute__((aligned(16))) char smem[2 * X_BYTES + 2 * Y_BYTES];
  char* sX = smem;
  char* sY = smem + 2 * X_BYTES;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int kt0 = blockIdx.x * BM;  // first K index of this block
  const int n0 = blockIdx.y * BN;
  const int step0 = blockIdx.z * p.steps_per_split;
  int step1 = step0 + p.steps_per_split;
  if (step1 > p.steps) step1 = p.steps;

  // ---- X gather state: thread owns NX (pixel-lane, k-chunk) pairs; the k part is fixed
  int x_pl[NX], x_kh[NX], x_kw[NX], x_ci[NX];
  bool x_kok[NX];
  int x_b[NX], x_yo[NX], x_xo[NX];
#pragma unroll
  for (int i = 0; i < NX; ++i) {
    const int c = tid + i * 256;
    x_pl[i] = c / XCH;
    const int kk = kt0 + (c % XCH) * 8;
    x_kok[i] = kk < p.K;
    const int kq = x_kok[i] ? kk : 0;
    const int tap = kq / p.Cin;
    x_ci[i] = kq - tap * p.Cin;
    x_kh[i] = tap / p.ks;
    x_kw[i] = tap - x_kh[i] * p.ks;
    const int m = step0 * 32 + x_pl[i];
    const int hw = p.Ho * p.Wo;
    const int b = m / hw;
    const int rem = m - b * hw;
    x_b[i] = b;
    x_yo[i] = rem / p.Wo;
    x_xo[i] = rem - x_yo[i] * p.Wo;
  }
  uint4 rx[NX], ry[NY];
  int lstep = step0;  // step being loaded

  auto load_tiles = [&]() {
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      uint4 v = make_uint4(0, 0, 0, 0);
      if (x_kok[i] && x_b[i] < p.B) {
        const int iy = x_yo[i] * p.stride - p.pad_t + x_kh[i];
        const int ix = x_xo[i] * p.stride - p.pad_l + x_kw[i];
        if ((unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W) {
          const bf16* src;
          if (x_ci[i] < p.C0) {
            src = p.x0 + ((size_t)(x_b[i] * p.H + iy) * p.W + ix) * p.C0 + x_ci[i];
          } else {
            const int H1 = p.H >> 1, W1 = p.W >> 1;
            src = p.x1 + ((size_t)(x_b[i] * H1 + (iy >> 1)) * W1 + (ix >> 1)) * p.C1 + (x_ci[i] - p.C0);
          }
          v = *reinterpret_cast<const uint4*>(src);
        }
      }
      rx[i] = v;
      // advance this chunk's pixel by 32
      x_xo[i] += 32;
      while (x_xo[i] >= p.Wo) {
        x_xo[i] -= p.Wo;
        if (++x_yo[i] == p.Ho) {
          x_yo[i] = 0;
          ++x_b[i];
        }
      }
    }
#pragma unroll
    for (int i = 0; i < NY; ++i) {
      const int c = tid + i * 256;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (c < 32 * YCH) {
        const int pl = c / YCH, ch = c % YCH;
        const int m = lstep * 32 + pl;
        const int n = n0 + ch * 8;
        if (m < p.M && n < p.ldy) v = *reinterpret_cast<const uint4*>(p.dy + (size_t)m * p.ldy + n);
      }
      ry[i] = v;
    }
    ++lstep;
  };
  auto store_tiles = [&](int buf) {
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      const int c = tid + i * 256;
      const int pl = c / XCH, ch = c % XCH;
      const int off = pl * XP + swz_u(pl, ch >> 1, XCH / 2) * 32 + (ch & 1) * 16;
      *reinterpret_cast<uint4*>(sX + buf * X_BYTES + off) = rx[i];
    }
#pragma unroll
    for (int i = 0; i < NY; ++i) {
      const int c = tid + i * 256;
      if (c < 32 * YCH) {
        const int pl = c / YCH, ch = c % YCH;
        const int off = pl * YP + swz_u(pl, ch >> 1, YCH / 2) * 32 + (ch & 1) * 16;
        *reinterpret_cast<uint4*>(sY + buf * Y_BYTES + off) = ry[i];
      }
    }
  };

  f32x4 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  if (step0 < step1) {
    load_tiles();
    store_tiles(0);
  }
  __syncthreads();

  // transposed-read addressing (cdna guide T10): inside a 16-lane group, lane 4q+p supplies
  // the address of row q, columns 4p..4p+3 of a 4x16 block and receives column (lane&15).
  const int g = lane >> 4, li = lane & 15, q = li >> 2, pc = li & 3;
  for (int st = step0; st < step1; ++st) {
    const int cur = (st - step0) & 1;
    const bool more = st + 1 < step1;
    if (more) load_tiles();
    bf16x8 af[MI], bfr[NI];
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const int u = (wm * WTM + i * 16) >> 4;
      const int r0 = 8 * g + q, r1 = r0 + 4;
      const char* base = sX + cur * X_BYTES;
      const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
          (lds_s16x4*)(base + r0 * XP + swz_u(r0, u, XCH / 2) * 32 + pc * 8));
      const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
          (lds_s16x4*)(base + r1 * XP + swz_u(r1, u, XCH / 2) * 32 + pc * 8));
      typedef short s16x8 __attribute__((ext_vector_type(8)));
      const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      af[i] = __builtin_bit_cast(bf16x8, v);
    }
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int u = (wn * WTN + j * 16) >> 4;
      const int r0 = 8 * g + q, r1 = r0 + 4;
      const char* base = sY + cur * Y_BYTES;
      const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
          (lds_s16x4*)(base + r0 * YP + swz_u(r0, u, YCH / 2) * 32 + pc * 8));
      const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
          (lds_s16x4*)(base + r1 * YP + swz_u(r1, u, YCH / 2) * 32 + pc * 8));
      typedef short s16x8 __attribute__((ext_vector_type(8)));
      const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      bfr[j] = __builtin_bit_cast(bf16x8, v);
    }
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
    if (more) store_tiles(cur ^ 1);
    __syncthreads();
  }

  // D: row (K index) = 4*(lane>>4) + r, col (co) = lane&15
  float* slab = p.out + (size_t)blockIdx.z * p.K * p.Cout;
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

__global__ __launch_bounds__(256) void slab_reduce_kernel(const float* slabs, float* out, int64_t n, int splits) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    float s = 0.f;
    for (int k = 0; k < splits; ++k) s += slabs[(size_t)k * n + i];
    out[i] = s;
  }
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
  int bn, splits;
  plan(d, &bn, &splits, &p.steps_per_split, &p.steps);
  p.out = splits == 1 ? dw : (float*)workspace;
  hipStream_t s = (hipStream_t)stream;
  dim3 grid(ceil_div(p.K, 128), ceil_div(p.Cout, bn), splits);
  if (bn == 128)
    hipLaunchKernelGGL(conv_wgrad_kernel<128>, grid, dim3(256), 0, s, p);
  else if (bn == 64)
    hipLaunchKernelGGL(conv_wgrad_kernel<64>, grid, dim3(256), 0, s, p);
  else
    hipLaunchKernelGGL(conv_wgrad_kernel<32>, grid, dim3(256), 0, s, p);
  DY_CHECK_LAUNCH();
  if (splits > 1) {
    const int64_t n = (int64_t)p.K * p.Cout;
    int g = ceil_div(n, 256);
    if (g > 2048) g = 2048;
    hipLaunchKernelGGL(slab_reduce_kernel, dim3(g), dim3(256), 0, s, (const float*)workspace, dw, n, splits);
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
  hipLaunchKernelGGL(slab_reduce_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, s, (const float*)workspace, dw, n,
                     blocks);
  DY_CHECK_LAUNCH();
  return DISYOLO_OK;
}
