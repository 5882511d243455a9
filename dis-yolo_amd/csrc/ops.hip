// First-layer convolution, weight packing, Adam and the l2 term.
#include <stdarg.h>
#include <stdlib.h>
#include "common.h"
#include "runtime.h"

// ---- error reporting -------------------------------------------------------------
static thread_local char g_err[512] = "";
void disyolo_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
extern "C" const char* disyolo_last_error(void) { return g_err; }
extern "C" int disyolo_version(void) { return 100; }

namespace {

// ---- conv 1: Cin = 3, k = 3, s = 1, SAME (yolo/yolo3_net_pos.py:159) ---------------
// Exact f32 FMA (27 taps) per output; 1 thread = 1 pixel x all COUT channels: the 27 input
// values are loaded once into registers, the weights are wave-uniform scalar loads (SGPR operands).
// FP8OUT: the output is stored as OCP e4m3 of y * inv_out_scale (32 contiguous bytes per lane, no staging)
template <int COUT, bool FP8OUT = false>
__global__ __launch_bounds__(256) void conv_first_kernel(const float* __restrict__ img, const float* __restrict__ sw,
                                                         const float* __restrict__ ssc,
                                                         const float* __restrict__ ssh, bf16* __restrict__ y, int B,
                                                         int H, int W, float alpha, float inv_out_scale = 0.f) {
  // each wave stages its 64 pixels x COUT bf16 in LDS so the global stores are contiguous 1 KiB rows
  __shared__ uint4 stage[4][64 * (COUT / 8)];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t total = (int64_t)B * H * W;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t rounds = (total + stride - 1) / stride;
  for (int64_t r = 0; r < rounds; ++r) {
    const int64_t m = r * stride + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t wave_m0 = m - lane;
    if (m < total) {
      const int x = (int)(m % W);
      const int64_t m2 = m / W;
      const int yy = (int)(m2 % H);
      const int b = (int)(m2 / H);
      float in[27];
#pragma unroll
      for (int kh = 0; kh < 3; ++kh) {
        const int iy = yy + kh - 1;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          const int ix = x + kw - 1;
          const bool ok = ((unsigned)iy < (unsigned)H) && ((unsigned)ix < (unsigned)W);
          const float* px = img + ((size_t)(b * H + (ok ? iy : 0)) * W + (ok ? ix : 0)) * 3;
#pragma unroll
          for (int ci = 0; ci < 3; ++ci) in[(kh * 3 + kw) * 3 + ci] = ok ? px[ci] : 0.f;
        }
      }
#pragma unroll
      for (int cg = 0; cg < COUT / 8; ++cg) {
        float acc[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] = 0.f;
#pragma unroll
        for (int t = 0; t < 27; ++t) {
          const float v = in[t];
          const float* wr = sw + t * COUT + cg * 8;
#pragma unroll
          for (int k = 0; k < 8; ++k) acc[k] = fmaf(v, wr[k], acc[k]);
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] = leaky(acc[k] * ssc[cg * 8 + k] + ssh[cg * 8 + k], alpha);
        if constexpr (FP8OUT) {
          int lo = 0, hi = 0;
#pragma unroll
          for (int k = 0; k < 8; ++k) acc[k] = fminf(fmaxf(acc[k] * inv_out_scale, -448.f), 448.f);
          lo = __builtin_amdgcn_cvt_pk_fp8_f32(acc[0], acc[1], lo, false);
          lo = __builtin_amdgcn_cvt_pk_fp8_f32(acc[2], acc[3], lo, true);
          hi = __builtin_amdgcn_cvt_pk_fp8_f32(acc[4], acc[5], hi, false);
          hi = __builtin_amdgcn_cvt_pk_fp8_f32(acc[6], acc[7], hi, true);
          *reinterpret_cast<uint2*>(reinterpret_cast<unsigned char*>(y) + (size_t)m * COUT + cg * 8) = uint2{(unsigned)lo, (unsigned)hi};
        } else {
          // chunk index XOR-swizzled by pixel so the 64 B-strided writes spread over the banks
          stage[wave][lane * (COUT / 8) + (cg ^ ((lane >> 1) & (COUT / 8 - 1)))] = pack8(acc);
        }
      }
    }
    if constexpr (FP8OUT) continue;
    // same wave wrote and reads: no block barrier needed, only LDS ordering
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (wave_m0 < total) {
      uint4* dst = reinterpret_cast<uint4*>(y + (size_t)wave_m0 * COUT);
      const int64_t lim = (total - wave_m0) * (COUT / 8);  // valid 16 B chunks of this wave's row block
#pragma unroll
      for (int i = 0; i < COUT / 8; ++i) {
        const int c = i * 64 + lane;  // chunk in the wave's contiguous output
        const int p = c / (COUT / 8), cg = c % (COUT / 8);
        if (c < lim) dst[c] = stage[wave][p * (COUT / 8) + (cg ^ ((p >> 1) & (COUT / 8 - 1)))];
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
}

// ---- conv 1 on the f32 matrix cores ----------------------------------------------------------
// The same layer as an exact-f32 GEMM: D[channel][pixel] = sum_k W[k][channel] * X[pixel][k], k = (kh,kw,ci)
// = 27 padded to 28, on v_mfma_f32_32x32x2_f32 (f32 in, f32 accumulate, no reduced-precision step): 14 MFMAs
// per 32 pixels x 32 channels.  The thread-per-pixel kernel above is bound by its wave-uniform scalar weight
// loads (90-100 us at B = 8, 576^2; packing two channels per v_pk_fma_f32 changed nothing); here the weights
// sit in 14 VGPRs per lane for the whole kernel (A operand: lane l holds W[k = 2j + (l>>5)][channel l&31])
// and the image is the B operand (lane l: pixel l&31, k = 2j + (l>>5)).  A tile is 32 consecutive pixels of
// one image row (W % 32 == 0): its 3 x 34 x 3 input floats are fetched as three contiguous 408-byte runs
// (6 coalesced loads per wave instead of 14 scattered ones -- those cost 49 of 108 us), one tile ahead,
// parked in LDS, and read back as operands with a 12-byte lane stride (3 is odd: conflict-free).
// With channels as rows the accumulator layout gives every lane ITS pixel's channels 4 at a time (rows
// (r&3) + 8(r>>2) + 4(l>>5)): scaled, activated, packed to 8-byte pieces, staged through LDS (16-byte chunk
// index XOR pixel&3) and stored as whole 64-byte pixel rows, 1 KiB per instruction.
// Measured at B = 8, 576^2: 75-82 us against 104-108 us for the VALU kernel on the same box.  The parts add
// up instead of overlapping (14 MFMAs 42 us, epilogue math 9 us, stores 12-14 us, the rest 17 us): the f32 MFMA
// runs at the vector-f32 rate and keeps the SIMD's VALU busy, so other waves' epilogues do not hide under it.
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void conv_first_mfma_kernel(const float* __restrict__ img, const float* __restrict__ sw,
                                                              const float* __restrict__ ssc, const float* __restrict__ ssh,
                                                              bf16* __restrict__ y, int B, int H, int W, float alpha) {
  constexpr int ROW = 104;                 // 34 pixels x 3 channels = 102 floats, padded
  __shared__ uint4 stage[4][32 * 4];       // per wave: 32 pixels x 64 bytes of output
  __shared__ float xin[4][2][3 * ROW];     // per wave, double-buffered: the tile's 3 input rows
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int col = lane & 31, half = lane >> 5;
  const int total = B * H * W;
  const int tiles = total >> 5;            // W % 32 == 0
  // a wave owns a contiguous range of tiles: the tile coordinates advance with scalar adds, no divisions
  const int nwaves = gridDim.x * 4, wid = blockIdx.x * 4 + wave;
  const int per = (tiles + nwaves - 1) / nwaves;
  int tile = wid * per;
  const int tend = min(tile + per, tiles);
  if (tile >= tend) return;
  float wreg[14];
#pragma unroll
  for (int j = 0; j < 14; ++j) {
    const int k = 2 * j + half;
    wreg[j] = k < 27 ? sw[k * 32 + col] : 0.f;
  }
  float sc[16], sh[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int ch = (r & 3) + 8 * (r >> 2) + 4 * half;
    sc[r] = ssc[ch];
    sh[r] = ssh[ch];
  }
  // byte offset in a tile's LDS image of this lane's operand of MFMA j (k = 2j + half -> tap (kh, kw), channel ci)
  int xoff[14];
#pragma unroll
  for (int j = 0; j < 14; ++j) {
    const int k0 = 2 * j, k1 = 2 * j + 1;
    const int o0 = ((k0 / 3) / 3) * ROW + ((k0 / 3) % 3) * 3 + k0 % 3;
    const int o1 = k1 < 27 ? ((k1 / 3) / 3) * ROW + ((k1 / 3) % 3) * 3 + k1 % 3 : 0;
    xoff[j] = ((half ? o1 : o0) + col * 3) * 4;
  }
  const bool k27 = half != 0;              // MFMA 13, upper lane half: k = 27 does not exist (its weight is 0 too)
  // fetch lanes: element e = lane + 64 t of a 102-float row run; the first pixel of the run is left of the
  // image in the first tile of a row, the last one right of it in the last tile
  const bool e1_in = lane + 64 < 102;
  const bool left0 = lane < 3, right1 = (lane + 64) / 3 == 33;
  int x0 = (tile << 5) % W, m2 = (tile << 5) / W;   // m2 = b*H + y (scalar)
  int yy = m2 % H;
  x0 = __builtin_amdgcn_readfirstlane(x0);
  m2 = __builtin_amdgcn_readfirstlane(m2);
  yy = __builtin_amdgcn_readfirstlane(yy);
  auto fetch = [&](bool live, int fx0, int fm2, int fyy, float* r) {   // input rows, global -> registers
    const bool lok = !(fx0 == 0 && left0), rok = !(fx0 + 32 == W && right1) && e1_in;
#pragma unroll
    for (int rr = 0; rr < 3; ++rr) {
      const int iy = fyy + rr - 1;
      const bool rowok = live && ((unsigned)iy < (unsigned)H);
      const float* rowp = img + ((size_t)(fm2 + rr - 1) * W + fx0 - 1) * 3 + lane;
      r[rr * 2 + 0] = (rowok && lok) ? rowp[0] : 0.f;
      r[rr * 2 + 1] = (rowok && rok) ? rowp[64] : 0.f;
    }
  };
  auto park = [&](int buf, const float* r) {
#pragma unroll
    for (int rr = 0; rr < 3; ++rr) {
      xin[wave][buf][rr * ROW + lane] = r[rr * 2 + 0];
      if (lane + 64 < ROW) xin[wave][buf][rr * ROW + lane + 64] = r[rr * 2 + 1];
    }
  };
  auto advance = [&]() {
    x0 += 32;
    if (x0 == W) {
      x0 = 0;
      ++m2;
      yy = yy + 1 == H ? 0 : yy + 1;
    }
  };
  float r[6];
  int buf = 0;
  fetch(true, x0, m2, yy, r);
  park(0, r);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  for (; tile < tend; ++tile) {
    advance();
    fetch(tile + 1 < tend, x0, m2, yy, r);   // the next tile, in flight under this tile's MFMAs
    f32x16 acc;
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] = 0.f;
    const char* xb = reinterpret_cast<const char*>(&xin[wave][buf][0]);
    float xv[14];
#pragma unroll
    for (int j = 0; j < 14; ++j) xv[j] = *reinterpret_cast<const float*>(xb + xoff[j]);
    if (k27) xv[13] = 0.f;
#pragma unroll
    for (int j = 0; j < 14; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wreg[j], xv[j], acc, 0, 0, 0);
    // this lane: pixel `col` of the tile, channels 8g + 4*half .. +3 for g = 0..3
    uint2* st2 = reinterpret_cast<uint2*>(&stage[wave][0]);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      float v[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) v[q] = leaky(acc[g * 4 + q] * sc[g * 4 + q] + sh[g * 4 + q], alpha);
      uint2 o;
      o.x = pack2(v[0], v[1]);
      o.y = pack2(v[2], v[3]);
      st2[(col * 4 + (g ^ (col & 3))) * 2 + half] = o;
    }
    park(buf ^ 1, r);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    uint4* dst = reinterpret_cast<uint4*>(y + ((size_t)tile << 5) * 32);
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int c = it * 64 + lane;        // 16-byte chunk of the tile's contiguous 2 KiB of output
      const int p = c >> 2, g = c & 3;
      dst[c] = stage[wave][p * 4 + (g ^ (p & 3))];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    buf ^= 1;
  }
}

// f32 [rows,3] image -> bf16 [rows,8] (channels 3..7 zero): the layout the MFMA weight-gradient
// kernel needs to treat the first layer like every other one
__global__ __launch_bounds__(256) void image_pad8_kernel(const float* img, uint4* out, int64_t rows) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < rows; i += stride) {
    const float v[8] = {img[i * 3], img[i * 3 + 1], img[i * 3 + 2], 0.f, 0.f, 0.f, 0.f, 0.f};
    out[i] = pack8(v);
  }
}
// dst[r][0:cols] = src[r][0:cols] with different row pitches (f32)
__global__ __launch_bounds__(256) void copy2d_kernel(const float* src, float* dst, int rows, int cols, int src_ld, int dst_ld) {
  const int n = rows * cols;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const int r = i / cols, c = i - r * cols;
    dst[(size_t)r * dst_ld + c] = src[(size_t)r * src_ld + c];
  }
}

// ---- weight packing: f32 HWIO -> bf16 [Cout][K] through a 64x64 LDS transpose ------
__global__ __launch_bounds__(256) void pack_fwd_kernel(const float* w, bf16* out, int K, int Cout) {
  __shared__ float t[64][65];
  const int k0 = blockIdx.x * 64, n0 = blockIdx.y * 64;
  for (int i = threadIdx.x; i < 64 * 64; i += 256) {
    const int kk = i / 64, nn = i % 64;
    float v = 0.f;
    if (k0 + kk < K && n0 + nn < Cout) v = w[(size_t)(k0 + kk) * Cout + n0 + nn];
    t[kk][nn] = v;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 64 * 64; i += 256) {
    const int nn = i / 64, kk = i % 64;
    if (k0 + kk < K && n0 + nn < Cout) out[(size_t)(n0 + nn) * K + k0 + kk] = (bf16)t[kk][nn];
  }
}
// dgrad operand: out[ci][tap'][co_pad] = w[taps-1-tap'][ci][co]   (co >= Cout -> 0)
__global__ __launch_bounds__(256) void pack_dgrad_kernel(const float* w, bf16* out, int taps, int Cin, int Cout,
                                                         int cout_pad) {
  const int64_t total = (int64_t)Cin * taps * cout_pad;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int co = (int)(i % cout_pad);
    const int64_t r = i / cout_pad;
    const int tp = (int)(r % taps);
    const int ci = (int)(r / taps);
    float v = 0.f;
    if (co < Cout) v = w[((size_t)(taps - 1 - tp) * Cin + ci) * Cout + co];
    out[i] = (bf16)v;
  }
}

// ---- all layers in one launch: a device table of jobs, blocks find their job by prefix
struct PackJob {
  const float* w;   // f32 HWIO master
  bf16* fwd;        // [Cout][K] or null
  bf16* dgrad;      // [Cin][taps*cout_pad] or null
  int K, Cout, Cin, taps, cout_pad;
  int blk0;         // first block of this job's forward part
  int blk1;         // first block of its dgrad part (== next job's blk0 when none)
  int tiles_k;      // forward tiles along K
};
__global__ __launch_bounds__(256) void pack_all_kernel(const PackJob* jobs, int njobs) {
  __shared__ float t[64][65];
  int lo = 0, hi = njobs - 1;
  const int bid = blockIdx.x;
  while (lo < hi) {  // last job whose blk0 <= bid
    const int mid = (lo + hi + 1) >> 1;
    if (jobs[mid].blk0 <= bid) lo = mid; else hi = mid - 1;
  }
  const PackJob j = jobs[lo];
  if (bid < j.blk1) {
    const int tb = bid - j.blk0;
    const int k0 = (tb % j.tiles_k) * 64, n0 = (tb / j.tiles_k) * 64;
    for (int i = threadIdx.x; i < 64 * 64; i += 256) {
      const int kk = i / 64, nn = i % 64;
      float v = 0.f;
      if (k0 + kk < j.K && n0 + nn < j.Cout) v = j.w[(size_t)(k0 + kk) * j.Cout + n0 + nn];
      t[kk][nn] = v;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * 64; i += 256) {
      const int nn = i / 64, kk = i % 64;
      if (k0 + kk < j.K && n0 + nn < j.Cout) j.fwd[(size_t)(n0 + nn) * j.K + k0 + kk] = (bf16)t[kk][nn];
    }
  } else {
    // dgrad part: 4096 output elements per block
    const int64_t total = (int64_t)j.Cin * j.taps * j.cout_pad;
    const int64_t base = (int64_t)(bid - j.blk1) * 4096;
    for (int e = threadIdx.x; e < 4096; e += 256) {
      const int64_t i = base + e;
      if (i >= total) break;
      const int co = (int)(i % j.cout_pad);
      const int64_t r = i / j.cout_pad;
      const int tp = (int)(r % j.taps);
      const int ci = (int)(r / j.taps);
      float v = 0.f;
      if (co < j.Cout) v = j.w[((size_t)(j.taps - 1 - tp) * j.Cin + ci) * j.Cout + co];
      j.dgrad[i] = (bf16)v;
    }
  }
}

// ---- TF-form Adam (train_yolo3_mask.py:55) ------------------------------------------
__global__ __launch_bounds__(256) void adam_kernel(float* w, const float* g, float* m, float* v, int64_t n,
                                                   int64_t n_decay, float lr_t, float b1, float b2, float eps,
                                                   float l2, float gscale) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const float wi = w[i];
    float gi = g[i] * gscale;
    if (i < n_decay) gi += l2 * wi;
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    w[i] = wi - lr_t * mi / (sqrtf(vi) + eps);
  }
}

// same, with the step count t = *counter + 1 read from device memory (graph / command-list
// replay: the host never touches the step), and the counter bumped by a trailing kernel
__global__ __launch_bounds__(256) void adam_dev_kernel(float* w, const float* g, float* m, float* v, int64_t n,
                                                       int64_t n_decay, float lr, float b1, float b2, float eps,
                                                       float l2, float gscale, const int64_t* counter) {
  const double t = (double)(*counter + 1);
  const float lr_t = (float)((double)lr * sqrt(1.0 - pow((double)b2, t)) / (1.0 - pow((double)b1, t)));
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const float wi = w[i];
    float gi = g[i] * gscale;
    if (i < n_decay) gi += l2 * wi;
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    w[i] = wi - lr_t * mi / (sqrtf(vi) + eps);
  }
}
__global__ void bump_counter_kernel(int64_t* counter) {
  if (threadIdx.x == 0 && blockIdx.x == 0) *counter += 1;
}

// Adam with the learning rate AND the step count in device memory (a recorded step can follow a
// schedule: the host only rewrites one float), which also produces the value of the l2 term,
// 0.5*l2*sum(w[0:n_decay]^2) of the PRE-update weights (what tf.losses.get_total_loss() adds,
// yolo/yolo3_net_pos.py:61), from the sweep over the weights it makes anyway: per-block partial
// sums (fixed grid, fixed order: deterministic) finished by the trailing counter kernel.
__global__ __launch_bounds__(256) void adam_fused_kernel(float* w, const float* g, float* m, float* v, int64_t n,
                                                         int64_t n_decay, const float* lr_dev, float b1, float b2,
                                                         float eps, float l2, float gscale, const int64_t* counter,
                                                         float* part) {
  __shared__ float sh[4];
  const double t = (double)(*counter + 1);
  const float lr_t = (float)((double)*lr_dev * sqrt(1.0 - pow((double)b2, t)) / (1.0 - pow((double)b1, t)));
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  float ss = 0.f;
  auto one = [&](float wi, float gi, float& mi, float& vi, bool decay) {
    gi *= gscale;
    if (decay) {
      gi += l2 * wi;
      ss += wi * wi;
    }
    mi = b1 * mi + (1.f - b1) * gi;
    vi = b2 * vi + (1.f - b2) * gi * gi;
    return wi - lr_t * mi / (sqrtf(vi) + eps);
  };
  // 16-byte accesses over the aligned body (w, g, m, v are slices at the same offset of 16-byte aligned
  // arenas: equally misaligned), scalars over the up to 3 leading and trailing elements
  const int64_t head = (int64_t)(((16u - (unsigned)((size_t)w & 15u)) & 15u) >> 2) < n
                           ? (int64_t)(((16u - (unsigned)((size_t)w & 15u)) & 15u) >> 2) : n;
  const int64_t n4 = (n - head) >> 2;
  float4* w4 = reinterpret_cast<float4*>(w + head);
  const float4* g4 = reinterpret_cast<const float4*>(g + head);
  float4* m4 = reinterpret_cast<float4*>(m + head);
  float4* v4 = reinterpret_cast<float4*>(v + head);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    float4 wi = w4[i], mi = m4[i], vi = v4[i];
    const float4 gi = g4[i];
    const int64_t e = head + (i << 2);
    wi.x = one(wi.x, gi.x, mi.x, vi.x, e < n_decay);
    wi.y = one(wi.y, gi.y, mi.y, vi.y, e + 1 < n_decay);
    wi.z = one(wi.z, gi.z, mi.z, vi.z, e + 2 < n_decay);
    wi.w = one(wi.w, gi.w, mi.w, vi.w, e + 3 < n_decay);
    m4[i] = mi;
    v4[i] = vi;
    w4[i] = wi;
  }
  if (blockIdx.x == gridDim.x - 1 && threadIdx.x < 8) {
    // thread t < 4: leading element t; 4 <= t < 8: trailing element t - 4
    const int64_t tail0 = head + (n4 << 2);
    const int64_t i = threadIdx.x < 4 ? (int64_t)threadIdx.x : tail0 + (threadIdx.x - 4);
    const bool mine = threadIdx.x < 4 ? (int64_t)threadIdx.x < head : i < n;
    if (mine) {
      float mi = m[i], vi = v[i];
      w[i] = one(w[i], g[i], mi, vi, i < n_decay);
      m[i] = mi;
      v[i] = vi;
    }
  }
  if (part) {
    ss = wave_sum(ss);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = ss;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
  }
}
// ring != NULL: the step's total loss -- (losses8[7] + mask_loss[0]) + reg, f32, the order YOLONet.total_loss() adds them in --
// goes to ring[t % ring_len], t = the counter BEFORE the increment (the 0-based index of the step that just finished)
__global__ __launch_bounds__(256) void adam_fused_tail_kernel(int64_t* counter, const float* part, int nb, float coef,
                                                              float* reg_out, const float* losses8, const float* mask_loss,
                                                              const float* reg_in, float* ring, int ring_len) {
  __shared__ double sh[4];
  float reg = reg_in ? reg_in[0] : 0.f;
  if (part && reg_out) {
    double s = 0.0;
    for (int i = threadIdx.x; i < nb; i += 256) s += (double)part[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
      reg = (float)(((sh[0] + sh[1]) + (sh[2] + sh[3])) * coef);
      reg_out[0] = reg;
    }
  }
  if (threadIdx.x == 0) {
    if (ring) ring[(int)(*counter % ring_len)] = (losses8[7] + mask_loss[0]) + reg;
    *counter += 1;
  }
}

__global__ __launch_bounds__(256) void sumsq_kernel(const float* w, int64_t n, float* part) {
  __shared__ float sh[4];
  float acc = 0.f;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) acc += w[i] * w[i];
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
}
__global__ void sumsq_final_kernel(const float* part, int nb, float coef, float* out) {
  double s = 0.0;
  for (int i = threadIdx.x; i < nb; i += 64) s += (double)part[i];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if (threadIdx.x == 0) out[0] = (float)(s * coef);
}

}  // namespace

extern "C" int disyolo_conv_first_fwd(const float* images, const float* w_hwio, const float* scale,
                                      const float* shift, void* y_bf16, int B, int H, int W, int Cout, float alpha,
                                      void* stream) {
  DY_REQUIRE(images && w_hwio && scale && shift && y_bf16 && B > 0 && H > 0 && W > 0, "conv_first: bad args");
  DY_REQUIRE(Cout == 32, "conv_first: Cout must be 32 (got %d)", Cout);
  DY_RECORD_OR_RUN([=](void* s) { return disyolo_conv_first_fwd(images, w_hwio, scale, shift, y_bf16, B, H, W, Cout, alpha, s); });
  const int64_t total = (int64_t)B * H * W;
  int grid = ceil_div(total, 256);
  if (grid > 256 * 16) grid = 256 * 16;
  // DISYOLO_FIRST_VALU=1: the thread-per-pixel VALU kernel (sequential 27-tap FMA chain per output)
  static const bool valu = getenv("DISYOLO_FIRST_VALU") && getenv("DISYOLO_FIRST_VALU")[0] == '1';
  if (!valu && W % 32 == 0 && total < (1LL << 31) / 3) {
    const int64_t tiles = total / 32;
    int g2 = (int)((tiles + 3) / 4);
    if (g2 > 256 * 4) g2 = 256 * 4;
    hipLaunchKernelGGL(conv_first_mfma_kernel, dim3(g2), dim3(256), 0, (hipStream_t)stream, images, w_hwio, scale, shift,
                       (bf16*)y_bf16, B, H, W, alpha);
    DY_CHECK_LAUNCH();
    return DISYOLO_OK;
  }
  hipLaunchKernelGGL(conv_first_kernel<32>, dim3(grid), dim3(256), 0, (hipStream_t)stream, images, w_hwio, scale,
                     shift, (bf16*)y_bf16, B, H, W, alpha, 0.f);
  DY_CHECK_LAUNCH();
  return DISYOLO_OK;
}

extern "C" int disyolo_conv_first_fwd_fp8(const float* images, const float* w_hwio, const float* scale,
                                          const float* shift, void* y_fp8, float out_scale, int B, int H, int W, int Cout,
                                          float alpha, void* stream) {
  DY_REQUIRE(images && w_hwio && scale && shift && y_fp8 && B > 0 && H > 0 && W > 0 && out_scale > 0.f, "conv_first_fp8: bad args");
  DY_REQUIRE(Cout == 32, "conv_first_fp8: Cout must be 32 (got %d)", Cout);
  DY_RECORD_OR_RUN([=](void* s) { return disyolo_conv_first_fwd_fp8(images, w_hwio, scale, shift, y_fp8, out_scale, B, H, W, Cout, alpha, s); });
  const int64_t total = (int64_t)B * H * W;
  int grid = ceil_div(total, 256);
  if (grid > 256 * 16) grid = 256 * 16;
  hipLaunchKernelGGL((conv_first_kernel<32, true>), dim3(grid), dim3(256), 0, (hipStream_t)stream, images, w_hwio, scale,
                     shift, (bf16*)y_fp8, B, H, W, alpha, 1.0f / out_scale);
  DY_CHECK_LAUNCH();
  return DISYOLO_OK;
}

extern "C" int disyolo_image_pad8(const float* images, void* out_bf16, int64_t pixels, void* stream) {
  DY_REQUIRE(images && out_bf16 && pixels > 0, "image_pad8: bad args");
  DY_RECORD_OR_RUN([=](void* s) { return disyolo_image_pad8(images, out_bf16, pixels, s); });
  int64_t g = (pixels + 255) / 256;
  if (g > 4096) g = 4096;
  hipLaunchKernelGGL(image_pad8_kernel, dim3((int)g), dim3(256), 0, (hipStream_t)stream, images, (uint4*)out_bf16, pixels);
  DY_CHECK_LAUNCH();
  return DISYOLO_OK;
}

extern "C" int disyolo_copy2d_f32(const float* src, float* dst, int rows, int cols, int src_ld, int dst_ld, void* stream) {
  DY_REQUIRE(src && dst && rows > 0 && cols > 0 && src_ld >= cols && dst_ld >= cols, "copy2d: bad args");
  DY_RECORD_OR_RUN([=](void* s) { return disyolo_copy2d_f32(src, dst, rows, cols, src_ld, dst_ld, s); });
  hipLaunchKernelGGL(copy2d_kernel, dim3(ceil_div((int64_t)rows * cols, 256)), dim3(256), 0, (hipStream_t)stream, src, dst,
                     rows, cols, src_ld, dst_ld);
  DY_CHECK_LAUNCH();
  return DISYOLO_OK;
}

extern "C" int disyolo_pack_weights(const float* w_hwio, void* w_fwd, void* w_dgrad, int ksize, int Cin, int Cout,
                                    int cout_pad, void* stream) {
  DY_REQUIRE(w_hwio && (w_fwd || w_dgrad) && ksize > 0 && Cin > 0 && Cout > 0, "pack_weights: bad args");
  DY_REQUIRE(!w_dgrad || cout_pad >= Cout, "pack_weights: cout_pad < Cout");
  DY_RECORD_OR_RUN([=](void* s) { return disyolo_pack_weights(w_hwio, w_fwd, w_dgrad, ksize, Cin, Cout, cout_pad, s); });
  hipStream_t s = (hipStream_t)stream;
  const int taps = ksize * ksize, K = taps * Cin;
  if (w_fwd) {
    hipLaunchKernelGGL(pack_fwd_kernel, dim3(ceil_div(K, 64), ceil_div(Cout, 64)), dim3(256), 0, s, w_hwio,
                       (bf16*)w_fwd, K, Cout);
    DY_CHECK_LAUNCH();
  }
  if (w_dgrad) {
    const int64_t total = (int64_t)Cin * taps * cout_pad;
    int grid = ceil_div(total, 256);
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(pack_dgrad_kernel, dim3(grid), dim3(256), 0, s, w_hwio, (bf16*)w_dgrad, taps, Cin, Cout,
                       cout_pad);
    DY_CHECK_LAUNCH();
  }
  return DISYOLO_OK;
}

// host-side job description (include/disyolo.h: disyolo_pack_job); the device table is built
// by the caller with disyolo_pack_table_build and lives in caller memory
extern "C" size_t disyolo_pack_table_bytes(int njobs) { return (size_t)njobs * sizeof(PackJob); }

extern "C" int disyolo_pack_table_build(const disyolo_pack_job* jobs, int njobs, void* host_table, int* total_blocks) {
  DY_REQUIRE(jobs && njobs > 0 && host_table && total_blocks, "pack_table_build: bad args");
  PackJob* t = (PackJob*)host_table;
  int blk = 0;
  for (int i = 0; i < njobs; ++i) {
    const disyolo_pack_job& s = jobs[i];
    DY_REQUIRE(s.w_hwio && s.w_fwd && s.ksize > 0 && s.Cin > 0 && s.Cout > 0, "pack_table_build: job %d incomplete", i);
    PackJob& d = t[i];
    d.w = s.w_hwio; d.fwd = (bf16*)s.w_fwd; d.dgrad = (bf16*)s.w_dgrad;
    d.taps = s.ksize * s.ksize; d.K = d.taps * s.Cin; d.Cout = s.Cout; d.Cin = s.Cin;
    d.cout_pad = s.cout_pad > s.Cout ? s.cout_pad : s.Cout;
    d.tiles_k = ceil_div(d.K, 64);
    d.blk0 = blk;
    blk += d.tiles_k * ceil_div(d.Cout, 64);
    d.blk1 = blk;
    if (d.dgrad) blk += ceil_div((int64_t)d.Cin * d.taps * d.cout_pad, 4096);
  }
  *total_blocks = blk;
  return DISYOLO_OK;
}

extern "C" int disyolo_pack_all(const void* device_table, int njobs, int total_blocks, void* stream) {
  DY_REQUIRE(device_table && njobs > 0 && total_blocks > 0, "pack_all: bad args");
  DY_RECORD_OR_RUN([=](void* s) { return disyolo_pack_all(device_table, njobs, total_blocks, s); });
  hipLaunchKernelGGL(pack_all_kernel, dim3(total_blocks), dim3(256), 0, (hipStream_t)stream, (const PackJob*)device_table,
                     njobs);
  DY_CHECK_LAUNCH();
  return DISYOLO_OK;
}

extern "C" int disyolo_adam_step(float* w, const float* grad, float* m, float* v, int64_t n, int64_t n_decay, float lr,
                                 float beta1, float beta2, float eps, float l2, int64_t t, float grad_scale,
                                 void* stream) {
  DY_REQUIRE(w && grad && m && v && n > 0 && t >= 1 && n_decay >= 0 && n_decay <= n, "adam: bad args");
  DY_RECORD_OR_RUN([=](void* s) { return disyolo_adam_step(w, grad, m, v, n, n_decay, lr, beta1, beta2, eps, l2, t, grad_scale, s); });
  const double lr_t = (double)lr * sqrt(1.0 - pow((double)beta2, (double)t)) / (1.0 - pow((double)beta1, (double)t));
  int grid = ceil_div(n, 256);
  if (grid > 256 * 16) grid = 256 * 16;
  hipLaunchKernelGGL(adam_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, w, grad, m, v, n, n_decay,
                     (float)lr_t, beta1, beta2, eps, l2, grad_scale);
  DY_CHECK_LAUNCH();
  return DISYOLO_OK;
}

extern "C" int disyolo_adam_step_dev(float* w, const float* grad, float* m, float* v, int64_t n, int64_t n_decay,
                                     float lr, float beta1, float beta2, float eps, float l2, int64_t* step_counter,
                                     float grad_scale, void* stream) {
  DY_REQUIRE(w && grad && m && v && step_counter && n > 0 && n_decay >= 0 && n_decay <= n, "adam_dev: bad args");
  DY_RECORD_OR_RUN([=](void* s) {
    return disyolo_adam_step_dev(w, grad, m, v, n, n_decay, lr, beta1, beta2, eps, l2, step_counter, grad_scale, s);
  });
  int grid = ceil_div(n, 256);
  if (grid > 256 * 16) grid = 256 * 16;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(adam_dev_kernel, dim3(grid), dim3(256), 0, s, w, grad, m, v, n, n_decay, lr, beta1, beta2, eps, l2,
                     grad_scale, (const int64_t*)step_counter);
  DY_CHECK_LAUNCH();
  hipLaunchKernelGGL(bump_counter_kernel, dim3(1), dim3(64), 0, s, step_counter);
  DY_CHECK_LAUNCH();
  return DISYOLO_OK;
}

namespace {
int adam_sweep_grid(int64_t n) {
  int64_t g = (n + 1023) / 1024;
  return (int)(g > 2048 ? 2048 : (g < 1 ? 1 : g));
}
}  // namespace

// One optimizer update as sweeps over slices of the arena + one finish: a recorded step runs the sweep of
// a slice as soon as the slice's gradients are final (beside the rest of the backward pass) instead of
// one sweep at the end of the critical chain.  Every sweep reads the same step counter; the finish
// sums the l2 partials of all sweeps (fixed order: deterministic) and advances the counter.
extern "C" int disyolo_adam_sweep_parts(int64_t n) { return n > 0 ? adam_sweep_grid(n) : 0; }

extern "C" int disyolo_adam_sweep(float* w, const float* grad, float* m, float* v, int64_t n, int64_t n_decay,
                                  const float* lr_dev, float beta1, float beta2, float eps, float l2,
                                  const int64_t* step_counter, float grad_scale, float* parts, void* stream) {
  DY_REQUIRE(w && grad && m && v && lr_dev && step_counter && n > 0 && n_decay >= 0 && n_decay <= n, "adam_sweep: bad args");
  DY_REQUIRE(((size_t)w & 3) == 0 && ((size_t)w & 15) == ((size_t)grad & 15) && ((size_t)w & 15) == ((size_t)m & 15) &&
                 ((size_t)w & 15) == ((size_t)v & 15),
             "adam_sweep: w, grad, m, v must sit at the same offset modulo 16 bytes (slices of aligned arenas)");
  DY_RECORD_OR_RUN([=](void* s) {
    return disyolo_adam_sweep(w, grad, m, v, n, n_decay, lr_dev, beta1, beta2, eps, l2, step_counter, grad_scale, parts, s);
  });
  hipLaunchKernelGGL(adam_fused_kernel, dim3(adam_sweep_grid(n)), dim3(256), 0, (hipStream_t)stream, w, grad, m, v, n,
                     n_decay, lr_dev, beta1, beta2, eps, l2, grad_scale, step_counter, parts);
  DY_CHECK_LAUNCH();
  return DISYOLO_OK;
}

extern "C" int disyolo_adam_finish(int64_t* step_counter, const float* parts, int nparts, float l2, float* reg_loss_out,
                                   void* stream) {
  DY_REQUIRE(step_counter && nparts >= 0 && (!reg_loss_out || (parts && nparts > 0)), "adam_finish: bad args");
  DY_RECORD_OR_RUN([=](void* s) { return disyolo_adam_finish(step_counter, parts, nparts, l2, reg_loss_out, s); });
  hipLaunchKernelGGL(adam_fused_tail_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, step_counter, parts, nparts,
                     0.5f * l2, reg_loss_out, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr,
                     (float*)nullptr, 1);
  DY_CHECK_LAUNCH();
  return DISYOLO_OK;
}

// the finish + the step's total loss into a ring the host reads LATER (a training loop that fetches the loss of every
// step -- train_yolo3_mask.py:216 does -- otherwise joins the device once per step): ring[t % ring_len] = (losses8[7] +
// mask_loss[0]) + reg, reg = the l2 term this finish just summed (reg_loss_out) or, when it sums none, reg_loss_in[0]
// (NULL: 0); t = the step counter before its increment
extern "C" int disyolo_adam_finish_record(int64_t* step_counter, const float* parts, int nparts, float l2, float* reg_loss_out,
                                          const float* losses8, const float* mask_loss, const float* reg_loss_in, float* ring,
                                          int ring_len, void* stream) {
  DY_REQUIRE(step_counter && nparts >= 0 && (!reg_loss_out || (parts && nparts > 0)), "adam_finish_record: bad args");
  DY_REQUIRE(losses8 && mask_loss && ring && ring_len > 0, "adam_finish_record: losses8, mask_loss, ring must be given");
  DY_RECORD_OR_RUN([=](void* s) {
    return disyolo_adam_finish_record(step_counter, parts, nparts, l2, reg_loss_out, losses8, mask_loss, reg_loss_in, ring, ring_len, s);
  });
  hipLaunchKernelGGL(adam_fused_tail_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, step_counter, parts, nparts,
                     0.5f * l2, reg_loss_out, losses8, mask_loss, reg_loss_in, ring, ring_len);
  DY_CHECK_LAUNCH();
  return DISYOLO_OK;
}

extern "C" size_t disyolo_adam_fused_workspace(int64_t n) { return n > 0 ? 4096 * sizeof(float) : 0; }

extern "C" int disyolo_adam_step_fused(float* w, const float* grad, float* m, float* v, int64_t n, int64_t n_decay,
                                       const float* lr_dev, float beta1, float beta2, float eps, float l2,
                                       int64_t* step_counter, float grad_scale, float* reg_loss_out, void* workspace,
                                       size_t workspace_bytes, void* stream) {
  DY_REQUIRE(w && grad && m && v && lr_dev && step_counter && n > 0 && n_decay >= 0 && n_decay <= n, "adam_fused: bad args");
  if (reg_loss_out && (!workspace || workspace_bytes < disyolo_adam_fused_workspace(n))) {
    disyolo_set_error("adam_fused: workspace too small");
    return DISYOLO_E_WORKSPACE;
  }
  float* part = reg_loss_out ? (float*)workspace : nullptr;
  int rc = disyolo_adam_sweep(w, grad, m, v, n, n_decay, lr_dev, beta1, beta2, eps, l2, step_counter, grad_scale, part, stream);
  if (rc) return rc;
  return disyolo_adam_finish(step_counter, part, part ? adam_sweep_grid(n) : 0, l2, reg_loss_out, stream);
}

extern "C" size_t disyolo_l2_workspace(int64_t n) { return n > 0 ? 1024 * sizeof(float) : 0; }

extern "C" int disyolo_l2_loss(const float* w, int64_t n, float l2, float* out, void* workspace,
                               size_t workspace_bytes, void* stream) {
  DY_REQUIRE(w && out && n > 0, "l2_loss: bad args");
  if (!workspace || workspace_bytes < disyolo_l2_workspace(n)) {
    disyolo_set_error("l2_loss: workspace too small");
    return DISYOLO_E_WORKSPACE;
  }
  DY_RECORD_OR_RUN([=](void* s) { return disyolo_l2_loss(w, n, l2, out, workspace, workspace_bytes, s); });
  hipStream_t s = (hipStream_t)stream;
  int nb = ceil_div(n, 256 * 8);
  if (nb > 1024) nb = 1024;
  hipLaunchKernelGGL(sumsq_kernel, dim3(nb), dim3(256), 0, s, w, n, (float*)workspace);
  DY_CHECK_LAUNCH();
  hipLaunchKernelGGL(sumsq_final_kernel, dim3(1), dim3(64), 0, s, (const float*)workspace, nb, 0.5f * l2, out);
  DY_CHECK_LAUNCH();
  return DISYOLO_OK;
}
