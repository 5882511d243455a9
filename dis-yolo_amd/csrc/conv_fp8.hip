// FP8 (OCP e4m3) forward convolution for the inference-mode layers of the network -- BASELINE.json
// configs[4] ("fp8 conv path, CDNA4 fp8 MFMA").  The reference is f32 throughout
// (yolo/yolo3_net_pos.py:42-57, conv_bn :132-146 with lock=True: moving statistics folded into a
// per-channel scale/shift); this is a storage / operand format of the port, like bf16.
//
//   y[m][co] = quant_out( leaky( acc[m][co] * escale[co] + eshift[co] ) + res[m][co] * res_scale )
//   acc = sum_k xq[m][k] * wq[co][k]        (v_mfma_f32_16x16x32_fp8_fp8, f32 accumulate)
//
// with per-tensor scales: x = xq * s_in, w = wq * s_w, escale = s_in * s_w * bn_scale, the output
// stored as e4m3 of y / s_out (and/or as bf16 of y for consumers that stay in bf16).  The non-scaled
// fp8 MFMA runs at the bf16 rate (MI355X_MICROARCH.md, matrix cores): what fp8 buys here is half the
// HBM / LDS bytes per operand -- it is applied to the HBM-bound locked backbone.
//
// Implicit GEMM: block = BM (128 or 64) output pixels x BN channels, K = k*k*Cin in slices of BK bytes.  Both
// operands are K-contiguous in memory ([pixel][Cin] NHWC, [Cout][k*k*Cin] packed weights), so the LDS
// tiles are [row][BK] byte images filled by LDS-DMA (16 B per lane; zero padding, ragged M / Cout edges
// are lanes out of range of the buffer descriptor) and MFMA fragments are plain ds_read_b64 (lane l:
// row l&15, k bytes 8*(l>>4)..+7).  16-byte units of a row are XOR-swizzled with the row so the reads
// are conflict-free.  The weights are the MFMA A operand: a lane then owns 4 consecutive channels of
// one pixel, the epilogue stages the tile through LDS and stores whole rows.
#include "common.h"
#include "runtime.h"

namespace {

typedef int i32x4 __attribute__((ext_vector_type(4)));
constexpr unsigned OOB = 0x80000000u;

struct Fp8Params {
  const uint8_t* x;
  const uint8_t* w;
  const float* escale;
  const float* eshift;
  const uint8_t* res;
  uint8_t* y8;
  bf16* y16;
  float res_scale, inv_out_scale, alpha;
  int B, H, W, Cin, lgCin, Ho, Wo, Cout, ks, stride, pad_t, pad_l, M, K;
  unsigned bytesx, bytesw;
};

__device__ __forceinline__ void dma16(unsigned voff, i32x4 srd, unsigned lds_dst) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds" : : "v"(voff), "s"(srd), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ i32x4 make_srd(const void* base, unsigned bytes) {
  i32x4 r;
  r[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)base);
  r[1] = __builtin_amdgcn_readfirstlane((int)((size_t)base >> 32)) & 0xffff;
  r[2] = __builtin_amdgcn_readfirstlane((int)bytes);
  r[3] = 0x00020000;
  return r;
}
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// f32 -> e4m3 (OCP, saturating to +-448) / back
__device__ __forceinline__ unsigned pack4_fp8(float a, float b, float c, float d) {
  a = fminf(fmaxf(a, -448.f), 448.f);
  b = fminf(fmaxf(b, -448.f), 448.f);
  c = fminf(fmaxf(c, -448.f), 448.f);
  d = fminf(fmaxf(d, -448.f), 448.f);
  int v = 0;
  v = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, v, false);
  v = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, v, true);
  return (unsigned)v;
}
__device__ __forceinline__ void unpack4_fp8(unsigned v, float* f) {
  f[0] = __builtin_amdgcn_cvt_f32_fp8((int)v, 0);
  f[1] = __builtin_amdgcn_cvt_f32_fp8((int)v, 1);
  f[2] = __builtin_amdgcn_cvt_f32_fp8((int)v, 2);
  f[3] = __builtin_amdgcn_cvt_f32_fp8((int)v, 3);
}

// swizzle of the 16-byte units of an LDS row (U units per row) by the row index
template <int U>
__device__ __forceinline__ int swz(int row, int u) {
  return U == 8 ? (u ^ ((row >> 1) & 7)) : (U == 4 ? (u ^ ((row >> 2) & 3)) : (U == 2 ? (u ^ ((row >> 3) & 1)) : u));
}

// MX = the block-scaled form of the fp8 MFMA, v_mfma_scale_f32_16x16x128_f8f6f4 with unit scales (E8M0 127 = 2^0 in every
// scale byte): 128 k-bytes per instruction at twice the non-scaled form's rate per byte (MI355X_MICROARCH.md, matrix cores:
// "block-scaled ... with e4m3 operands: twice the cycles of the BF16 form of the same M x N at 4x the K") -- the only fp8 form
// that is faster than bf16.  Same products, same f32 accumulation: the numbers of the non-scaled path.  K slices of 128 bytes
// (Cin a multiple of 128: a slice never straddles a tap); a lane's fragment is 32 of the row's 128 bytes -- the 16-byte units
// fg and 4 + fg (fg = lane >> 4), the SAME units for both operands, so whatever k index the instruction gives register byte
// j of lane group fg, the two operands agree on it -- read with two ds_read_b128 whose lane -> (row, unit) pattern is the
// bf16 GEMM tile's (unit = kk * 4 + fg with a 128-byte row), conflict-free under the same XOR swizzle.
typedef int i32x8 __attribute__((ext_vector_type(8)));
template <int BM, int BN, int BK, int ST, bool MX = false>
__global__ __launch_bounds__(256) void conv_fp8_kernel(Fp8Params p) {
  static_assert(!MX || BK % 128 == 0, "the block-scaled MFMA multiplies 128 k-bytes");
  // BM = pixels per block (128; 64 for the small feature maps whose grid would not fill the chip)
  constexpr int WN = (BN == 128 || BM == 64) ? 2 : 1;   // waves along the channels
  constexpr int WM = 4 / WN;                    // waves along the pixels
  constexpr int WTM = BM / WM, WTN = BN / WN;   // wave tile: pixels x channels
  constexpr int PI = WTM / 16, CI = WTN / 16;   // 16-pixel / 16-channel fragments per wave
  constexpr int U = BK / 16;                    // 16-byte units per LDS row
  constexpr int RPD = 64 / U;                   // rows per wave DMA (1 KiB)
  constexpr int XD_T = BM / RPD;                // pixel DMAs per slice (whole block)
  constexpr int XD = (XD_T + 3) / 4;            // ... per wave
  constexpr int WD_T = (BN + RPD - 1) / RPD;    // weight DMAs per slice (whole block)
  constexpr int WD = (WD_T + 3) / 4;            // ... per wave (the last ones may be idle lanes)
  constexpr int XB = XD * 4 * 1024, WB = WD * 4 * 1024;
  static_assert(WTM % 16 == 0 && WTN % 16 == 0, "wave tile");
  constexpr int STB = XB + WB;
  constexpr int PRE = ST - 1;
  constexpr int LPT = XD + WD;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  int lin;
  {
    const int nblk = gridDim.x, bid = blockIdx.x;
    const int q8 = nblk >> 3, r8 = nblk & 7, xcd = bid & 7, loc = bid >> 3;
    lin = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + loc;
  }
  const int tilesN = (p.Cout + BN - 1) / BN;
  const int nt = lin % tilesN, mt = lin / tilesN;      // channel tile fastest: blocks sharing pixels share an XCD's L2
  const int m0 = mt * BM, n0 = nt * BN;

  const i32x4 srdx = make_srd(p.x, p.bytesx);
  const i32x4 srdw = make_srd(p.w, p.bytesw);

  // ---- pixel DMAs: lane -> row (pixel) and 16-byte piece; the pixel is fixed, the tap moves with the K slice
  int x_b[XD], x_y[XD], x_x[XD], x_lu[XD];
  bool x_ok[XD];
#pragma unroll
  for (int j = 0; j < XD; ++j) {
    const int row = (j * 4 + wave) * RPD + lane / U, pu = lane % U;
    x_lu[j] = swz<U>(row, pu);
    const int m = m0 + row;
    x_ok[j] = row < BM && m < p.M;
    const int hw = p.Ho * p.Wo;
    const int b = m / hw, rem = m - b * hw;
    x_b[j] = b;
    x_y[j] = (rem / p.Wo) * p.stride - p.pad_t;
    x_x[j] = (rem % p.Wo) * p.stride - p.pad_l;
  }
  unsigned w_off[WD];
  bool w_ok[WD];
#pragma unroll
  for (int j = 0; j < WD; ++j) {
    const int row = (j * 4 + wave) * RPD + lane / U, pu = lane % U;
    const int lu = swz<U>(row, pu);
    w_ok[j] = row < BN && (n0 + row) < p.Cout;
    w_off[j] = (unsigned)(n0 + row) * (unsigned)p.K + lu * 16;
  }
  const int nslices = p.K / BK;
  auto issue = [&](int slice, int stage) {
    const unsigned sbase = lds0 + stage * STB + wave * 1024;
#pragma unroll
    for (int j = 0; j < XD; ++j) {
      const int k = slice * BK + x_lu[j] * 16;
      const int tap = k >> p.lgCin, ci = k & (p.Cin - 1);
      const int kh = p.ks == 3 ? (tap * 11) >> 5 : 0, kw = tap - kh * 3;    // tap / 3 for tap < 9
      const int iy = x_y[j] + kh, ix = x_x[j] + kw;
      const bool ok = x_ok[j] && ((unsigned)iy < (unsigned)p.H) && ((unsigned)ix < (unsigned)p.W);
      const unsigned off = (unsigned)(((x_b[j] * p.H + iy) * p.W + ix) * p.Cin + ci);
      dma16(ok ? off : OOB, srdx, sbase + j * 4096);
    }
#pragma unroll
    for (int j = 0; j < WD; ++j) dma16(w_ok[j] ? w_off[j] + (unsigned)(slice * BK) : OOB, srdw, sbase + XB + j * 4096);
  };

  f32x4 acc[CI][PI];
#pragma unroll
  for (int i = 0; i < CI; ++i)
#pragma unroll
    for (int j = 0; j < PI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

#pragma unroll
  for (int s = 0; s < PRE; ++s)
    if (s < nslices) issue(s, s);

  const int fr = lane & 15, fg = lane >> 4;      // fragment row, 8-byte k group
  for (int s = 0; s < nslices; ++s) {
    if (s + PRE - 1 < nslices)
      wait_vmcnt<LPT*(PRE - 1)>();
    else
      wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (s + PRE < nslices) issue(s + PRE, (s + PRE) % ST);
    const char* sX = smem + (s % ST) * STB;
    const char* sW = sX + XB;
    if constexpr (MX) {
#pragma unroll
      for (int kk = 0; kk < BK / 128; ++kk) {
        i32x8 wf[CI], xf[PI];
#pragma unroll
        for (int i = 0; i < CI; ++i) {
          const int row = wn * WTN + i * 16 + fr;
          const int4 lo = *reinterpret_cast<const int4*>(sW + row * BK + swz<U>(row, kk * 8 + fg) * 16);
          const int4 hi = *reinterpret_cast<const int4*>(sW + row * BK + swz<U>(row, kk * 8 + 4 + fg) * 16);
          wf[i] = i32x8{lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
        }
#pragma unroll
        for (int j = 0; j < PI; ++j) {
          const int row = wm * WTM + j * 16 + fr;
          const int4 lo = *reinterpret_cast<const int4*>(sX + row * BK + swz<U>(row, kk * 8 + fg) * 16);
          const int4 hi = *reinterpret_cast<const int4*>(sX + row * BK + swz<U>(row, kk * 8 + 4 + fg) * 16);
          xf[j] = i32x8{lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
        }
#pragma unroll
        for (int i = 0; i < CI; ++i)
#pragma unroll
          for (int j = 0; j < PI; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wf[i], xf[j], acc[i][j], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
      }
    } else
#pragma unroll
    for (int kk = 0; kk < BK / 32; ++kk) {
      long wf[CI], xf[PI];
      const int chunk = kk * 4 + fg;             // 8-byte chunk of the row
#pragma unroll
      for (int i = 0; i < CI; ++i) {
        const int row = wn * WTN + i * 16 + fr;
        wf[i] = *reinterpret_cast<const long*>(sW + row * BK + swz<U>(row, chunk >> 1) * 16 + (chunk & 1) * 8);
      }
#pragma unroll
      for (int j = 0; j < PI; ++j) {
        const int row = wm * WTM + j * 16 + fr;
        xf[j] = *reinterpret_cast<const long*>(sX + row * BK + swz<U>(row, chunk >> 1) * 16 + (chunk & 1) * 8);
      }
#pragma unroll
      for (int i = 0; i < CI; ++i)
#pragma unroll
        for (int j = 0; j < PI; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(wf[i], xf[j], acc[i][j], 0, 0, 0);
    }
  }

  // ---- epilogue.  acc[i][j][r]: channel n0 + wn*WTN + i*16 + 4*fg + r, pixel m0 + wm*WTM + j*16 + fr
  wait_vmcnt<0>();
  __syncthreads();
  constexpr int ROW8 = WTN + 16, ROW16 = WTN * 2 + 16;     // staging row pitch (bytes), padded
  char* st = smem + wave * (WTM * ROW16);
  float v[CI][PI][4];
#pragma unroll
  for (int i = 0; i < CI; ++i) {
    const int c = n0 + wn * WTN + i * 16 + 4 * fg;
    float es[4], eh[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      es[r] = (c + r) < p.Cout ? p.escale[c + r] : 0.f;
      eh[r] = (c + r) < p.Cout ? p.eshift[c + r] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < PI; ++j) {
      const int m = m0 + wm * WTM + j * 16 + fr;
      float rs[4] = {0.f, 0.f, 0.f, 0.f};
      if (p.res && m < p.M && c < p.Cout) {
        unpack4_fp8(*reinterpret_cast<const unsigned*>(p.res + (size_t)m * p.Cout + c), rs);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) v[i][j][r] = leaky(acc[i][j][r] * es[r] + eh[r], p.alpha) + rs[r] * p.res_scale;
    }
  }
  if (p.y8) {
#pragma unroll
    for (int i = 0; i < CI; ++i)
#pragma unroll
      for (int j = 0; j < PI; ++j)
        *reinterpret_cast<unsigned*>(st + (j * 16 + fr) * ROW8 + i * 16 + 4 * fg) =
            pack4_fp8(v[i][j][0] * p.inv_out_scale, v[i][j][1] * p.inv_out_scale, v[i][j][2] * p.inv_out_scale,
                      v[i][j][3] * p.inv_out_scale);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    constexpr int CPR = WTN / 16;                // 16-byte chunks per pixel row
#pragma unroll
    for (int it = 0; it < (WTM * CPR + 63) / 64; ++it) {
      const int idx = it * 64 + lane;
      const int row = idx / CPR, ch = idx % CPR;
      const int m = m0 + wm * WTM + row, c = n0 + wn * WTN + ch * 16;
      if (idx < WTM * CPR && m < p.M && c < p.Cout)
        *reinterpret_cast<uint4*>(p.y8 + (size_t)m * p.Cout + c) = *reinterpret_cast<const uint4*>(st + row * ROW8 + ch * 16);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
  if (p.y16) {
#pragma unroll
    for (int i = 0; i < CI; ++i)
#pragma unroll
      for (int j = 0; j < PI; ++j) {
        uint2 u;
        u.x = pack2(v[i][j][0], v[i][j][1]);
        u.y = pack2(v[i][j][2], v[i][j][3]);
        *reinterpret_cast<uint2*>(st + (j * 16 + fr) * ROW16 + (i * 16 + 4 * fg) * 2) = u;
      }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    constexpr int CPR = WTN / 8;
#pragma unroll
    for (int it = 0; it < (WTM * CPR + 63) / 64; ++it) {
      const int idx = it * 64 + lane;
      const int row = idx / CPR, ch = idx % CPR;
      const int m = m0 + wm * WTM + row, c = n0 + wn * WTN + ch * 8;
      if (idx < WTM * CPR && m < p.M && c < p.Cout)
        *reinterpret_cast<uint4*>(p.y16 + (size_t)m * p.Cout + c) = *reinterpret_cast<const uint4*>(st + row * ROW16 + ch * 16);
    }
  }
}

// x (bf16 or f32) -> e4m3 of x * inv_scale; n % 8 == 0
template <typename T>
__global__ __launch_bounds__(256) void quant_fp8_kernel(const T* x, uint8_t* y, int64_t n, float inv_scale) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n / 8; i += stride) {
    float f[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) f[k] = (float)x[i * 8 + k] * inv_scale;
    uint2 u;
    u.x = pack4_fp8(f[0], f[1], f[2], f[3]);
    u.y = pack4_fp8(f[4], f[5], f[6], f[7]);
    *reinterpret_cast<uint2*>(y + i * 8) = u;
  }
}
__global__ __launch_bounds__(256) void dequant_fp8_kernel(const uint8_t* x, float* y, int64_t n, float scale) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n / 4; i += stride) {
    float f[4];
    unpack4_fp8(*reinterpret_cast<const unsigned*>(x + i * 4), f);
    *reinterpret_cast<float4*>(y + i * 4) = float4{f[0] * scale, f[1] * scale, f[2] * scale, f[3] * scale};
  }
}
// HWIO f32 weights -> e4m3 [Cout][k*k*Cin] of w * inv_scale
__global__ __launch_bounds__(256) void pack_fp8_kernel(const float* w, uint8_t* out, int K, int Cout, float inv_scale) {
  const int64_t total = (int64_t)K * Cout / 4;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int co = (int)(i / (K / 4)), k = (int)(i % (K / 4)) * 4;
    *reinterpret_cast<unsigned*>(out + (size_t)co * K + k) =
        pack4_fp8(w[(size_t)k * Cout + co] * inv_scale, w[(size_t)(k + 1) * Cout + co] * inv_scale,
                  w[(size_t)(k + 2) * Cout + co] * inv_scale, w[(size_t)(k + 3) * Cout + co] * inv_scale);
  }
}

template <int BM, int BN, int BK, bool MX = false>
int launch_fp8(const Fp8Params& p, hipStream_t s) {
  constexpr int ST = MX ? 2 : 3;        // (128-byte slices: two stages = 64 KB, two blocks per CU)
  constexpr int U = BK / 16, RPD = 64 / U;
  constexpr int WD = ((BN + RPD - 1) / RPD + 3) / 4;
  constexpr int XD = ((BM + RPD - 1) / RPD + 3) / 4;
  constexpr size_t lds = (size_t)ST * ((XD + WD) * 4 * 1024);
  constexpr int WN = (BN == 128 || BM == 64) ? 2 : 1;
  constexpr size_t stage = (size_t)4 * (BM / (4 / WN)) * ((BN / WN) * 2 + 16);
  const size_t bytes = lds > stage ? lds : stage;
  const int grid = ceil_div(p.M, BM) * ceil_div(p.Cout, BN);
  hipLaunchKernelGGL((conv_fp8_kernel<BM, BN, BK, ST, MX>), dim3(grid), dim3(256), bytes, s, p);
  return 0;
}

}  // namespace

extern "C" int disyolo_conv2d_fp8_fwd(const disyolo_conv_desc* d, const void* w_fp8, const float* escale, const float* eshift,
                                      const void* residual_fp8, float residual_scale, void* y_fp8, float out_scale,
                                      void* y_bf16, void* stream) {
  DY_REQUIRE(d && d->x0 && w_fp8 && escale && eshift && (y_fp8 || y_bf16), "conv_fp8: null pointer");
  DY_REQUIRE(d->C1 == 0 && d->in_div == 1 && (d->ksize == 1 || d->ksize == 3), "conv_fp8: plain 1x1 / 3x3 layers only");
  DY_REQUIRE(d->C0 >= 32 && (d->C0 & (d->C0 - 1)) == 0 && d->Cout % 16 == 0, "conv_fp8: Cin must be a power of two >= 32, Cout a multiple of 16");
  DY_REQUIRE(!y_fp8 || out_scale > 0.f, "conv_fp8: out_scale");
  DY_REQUIRE((int64_t)d->B * d->H * d->W * d->C0 < (1LL << 31) && (int64_t)d->Cout * d->ksize * d->ksize * d->C0 < (1LL << 31),
             "conv_fp8: tensor exceeds the 2 GiB of the 32-bit gather offsets");
  {
    const disyolo_conv_desc c = *d;
    DY_RECORD_OR_RUN([=](void* s) {
      return disyolo_conv2d_fp8_fwd(&c, w_fp8, escale, eshift, residual_fp8, residual_scale, y_fp8, out_scale, y_bf16, s);
    });
  }
  Fp8Params p;
  p.x = (const uint8_t*)d->x0;
  p.w = (const uint8_t*)w_fp8;
  p.escale = escale; p.eshift = eshift;
  p.res = (const uint8_t*)residual_fp8;
  p.y8 = (uint8_t*)y_fp8; p.y16 = (bf16*)y_bf16;
  p.res_scale = residual_scale;
  p.inv_out_scale = y_fp8 ? 1.0f / out_scale : 0.f;
  p.alpha = (d->flags & DISYOLO_CONV_LEAKY) ? d->alpha : 1.0f;
  p.B = d->B; p.H = d->H; p.W = d->W; p.Cin = d->C0; p.Ho = d->Ho; p.Wo = d->Wo; p.Cout = d->Cout;
  p.lgCin = 0;
  while ((1 << p.lgCin) < d->C0) ++p.lgCin;
  p.ks = d->ksize; p.stride = d->stride; p.pad_t = d->pad_t; p.pad_l = d->pad_l;
  p.M = d->B * d->Ho * d->Wo;
  p.K = d->ksize * d->ksize * d->C0;
  p.bytesx = (unsigned)((size_t)d->B * d->H * d->W * d->C0);
  p.bytesw = (unsigned)((size_t)d->Cout * p.K);
  hipStream_t s = (hipStream_t)stream;
  const bool bk64 = (d->C0 % 64) == 0;
  // 64-pixel tiles when 128-pixel tiles would leave CUs without a block (the 18x18 / 36x36 maps at batch 8)
  const bool small = bk64 && d->Cout >= 128 && ceil_div(p.M, 128) * ceil_div(d->Cout, 128) < 256;
  // the block-scaled MFMA (K slices of 128 bytes) wherever the input has >= 128 channels: conv10-52 of the backbone
  static const bool mx_on = [] { const char* e = getenv("DISYOLO_FP8_MX"); return !(e && e[0] == '0'); }();
  const bool mx = mx_on && (d->C0 % 128) == 0 && d->Cout >= 128;
  if (small) {
    if (mx) launch_fp8<64, 128, 128, true>(p, s); else launch_fp8<64, 128, 64>(p, s);
  } else if (d->Cout > 64) {
    if (mx) launch_fp8<128, 128, 128, true>(p, s);
    else if (bk64) launch_fp8<128, 128, 64>(p, s); else launch_fp8<128, 128, 32>(p, s);
  } else if (d->Cout > 32) {
    if (bk64) launch_fp8<128, 64, 64>(p, s); else launch_fp8<128, 64, 32>(p, s);
  } else {
    if (bk64) launch_fp8<128, 32, 64>(p, s); else launch_fp8<128, 32, 32>(p, s);
  }
  DY_CHECK_LAUNCH();
  return DISYOLO_OK;
}

extern "C" int disyolo_quant_fp8(const void* x, int x_is_f32, void* y_fp8, int64_t n, float scale, void* stream) {
  DY_REQUIRE(x && y_fp8 && n > 0 && n % 8 == 0 && scale > 0.f, "quant_fp8: bad args (n %% 8 == 0)");
  DY_RECORD_OR_RUN([=](void* s) { return disyolo_quant_fp8(x, x_is_f32, y_fp8, n, scale, s); });
  int64_t g = (n / 8 + 255) / 256;
  if (g > 8192) g = 8192;
  if (x_is_f32)
    hipLaunchKernelGGL(quant_fp8_kernel<float>, dim3((int)g), dim3(256), 0, (hipStream_t)stream, (const float*)x, (uint8_t*)y_fp8, n, 1.0f / scale);
  else
    hipLaunchKernelGGL(quant_fp8_kernel<bf16>, dim3((int)g), dim3(256), 0, (hipStream_t)stream, (const bf16*)x, (uint8_t*)y_fp8, n, 1.0f / scale);
  DY_CHECK_LAUNCH();
  return DISYOLO_OK;
}

extern "C" int disyolo_dequant_fp8(const void* x_fp8, float* y, int64_t n, float scale, void* stream) {
  DY_REQUIRE(x_fp8 && y && n > 0 && n % 4 == 0, "dequant_fp8: bad args (n %% 4 == 0)");
  DY_RECORD_OR_RUN([=](void* s) { return disyolo_dequant_fp8(x_fp8, y, n, scale, s); });
  int64_t g = (n / 4 + 255) / 256;
  if (g > 8192) g = 8192;
  hipLaunchKernelGGL(dequant_fp8_kernel, dim3((int)g), dim3(256), 0, (hipStream_t)stream, (const uint8_t*)x_fp8, y, n, scale);
  DY_CHECK_LAUNCH();
  return DISYOLO_OK;
}

extern "C" int disyolo_pack_weights_fp8(const float* w_hwio, void* w_fp8, int ksize, int Cin, int Cout, float scale, void* stream) {
  DY_REQUIRE(w_hwio && w_fp8 && ksize > 0 && Cin > 0 && Cout > 0 && (ksize * ksize * Cin) % 4 == 0 && scale > 0.f, "pack_weights_fp8: bad args");
  DY_RECORD_OR_RUN([=](void* s) { return disyolo_pack_weights_fp8(w_hwio, w_fp8, ksize, Cin, Cout, scale, s); });
  const int K = ksize * ksize * Cin;
  int64_t g = ((int64_t)K * Cout / 4 + 255) / 256;
  if (g > 8192) g = 8192;
  hipLaunchKernelGGL(pack_fp8_kernel, dim3((int)g), dim3(256), 0, (hipStream_t)stream, w_hwio, (uint8_t*)w_fp8, K, Cout, 1.0f / scale);
  DY_CHECK_LAUNCH();
  return DISYOLO_OK;
}
