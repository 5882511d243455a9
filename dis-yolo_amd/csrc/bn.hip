// Batch-norm / activation kernels and the small column reductions of the backward pass.
// All HBM-bound: 16-byte bf16 vectors per lane, f32 math, deterministic two-stage
// reductions (per-block partials in a caller workspace, fixed-order final sum in f64).
//
// Replaces tf.nn.moments / tf.nn.batch_normalization / tf.assign(moving stats) /
// tf.maximum(alpha*x, x) and their TF-autodiff gradients, yolo/yolo3_net_pos.py:68-107.
#include <stdio.h>
#include <stdlib.h>
#include "common.h"
#include "runtime.h"

namespace {

// ---- stage 2 of every column reduction: partial[rows][C][Q] -> f64 sums ------------
// 256 threads = FIN_CPB channels x 256/FIN_CPB row lanes; each lane strides over the partial
// rows, the lane sums are combined through LDS in a fixed order (deterministic).  4 channels
// per block measured best end to end (1, 2, 8: -1.2 %, -0.7 %, -0.2 %).
constexpr int FIN_CPB = 4;
template <int Q>
__device__ __forceinline__ bool block_sum_partials(const float* part, int rows, int C, double* res /*[Q]*/, int* c_out) {
  constexpr int LANES = 256 / FIN_CPB;
  __shared__ double sh[LANES][FIN_CPB][Q];
  const int cl = threadIdx.x % FIN_CPB, g = threadIdx.x / FIN_CPB;
  const int c = blockIdx.x * FIN_CPB + cl;
  double acc[Q];
#pragma unroll
  for (int q = 0; q < Q; ++q) acc[q] = 0.0;
  if (c < C) {
    int r = g;
    for (; r + 3 * LANES < rows; r += 4 * LANES) {   // four independent loads in flight
      float v[4][Q];
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int q = 0; q < Q; ++q) v[u][q] = part[((size_t)(r + u * LANES) * C + c) * Q + q];
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int q = 0; q < Q; ++q) acc[q] += (double)v[u][q];
    }
    for (; r < rows; r += LANES) {
#pragma unroll
      for (int q = 0; q < Q; ++q) acc[q] += (double)part[((size_t)r * C + c) * Q + q];
    }
  }
#pragma unroll
  for (int q = 0; q < Q; ++q) sh[g][cl][q] = acc[q];
  __syncthreads();
  *c_out = c;
  if (g == 0 && c < C) {
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      double s = 0.0;
#pragma unroll
      for (int k = 0; k < LANES; ++k) s += sh[k][cl][q];
      res[q] = s;
    }
    return true;
  }
  return false;
}

__global__ __launch_bounds__(256) void bn_finalize_kernel(const float* stats, int rows, int C, double inv_count,
                                                          const float* gamma, const float* beta, float* mm, float* mv,
                                                          float decay, float eps, float* scale, float* shift,
                                                          float* mean_out, float* rstd_out) {
  // the per-channel operands of the writers are fetched BEFORE the reduction: the kernel is two dependent memory round
  // trips otherwise (partial rows, then gamma / beta / moving statistics), and it sits on the step's critical chain
  const int c0 = blockIdx.x * FIN_CPB + threadIdx.x;
  const bool writer = threadIdx.x < FIN_CPB && c0 < C;
  const float g0 = writer ? gamma[c0] : 0.f, b0 = writer ? beta[c0] : 0.f;
  const float mm0 = (writer && mm) ? mm[c0] : 0.f, mv0 = (writer && mv) ? mv[c0] : 0.f;
  double r[2];
  int c;
  if (!block_sum_partials<2>(stats, rows, C, r, &c)) return;
  const double mean = r[0] * inv_count;
  double var = r[1] * inv_count - mean * mean;  // population variance (tf.nn.moments)
  if (var < 0.0) var = 0.0;
  const float meanf = (float)mean, varf = (float)var;
  const float rstd = 1.0f / sqrtf(varf + eps);
  const float sc = g0 * rstd;
  scale[c] = sc;
  shift[c] = b0 - meanf * sc;
  if (mean_out) mean_out[c] = meanf;
  if (rstd_out) rstd_out[c] = rstd;
  if (mm) mm[c] = mm0 * decay + meanf * (1.0f - decay);
  if (mv) mv[c] = mv0 * decay + varf * (1.0f - decay);
}

__global__ void bn_fold_kernel(const float* gamma, const float* beta, const float* mm, const float* mv, float eps,
                               float* scale, float* shift, int C) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float sc = gamma[c] / sqrtf(mv[c] + eps);
  scale[c] = sc;
  shift[c] = beta[c] - mm[c] * sc;
}

// y = leaky(x*scale + shift) [+ residual]; one uint4 (8 channels) per thread-iteration
__global__ __launch_bounds__(256) void bn_act_fwd_kernel(const uint4* x, const float* scale, const float* shift,
                                                         const uint4* residual, uint4* y, int64_t nvec, int C,
                                                         float alpha) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  // channel of the first element of vector i, carried along instead of a 64-bit modulo per iteration
  const int chunks = C >> 3;
  const int cstep = (int)(stride % chunks);
  int ch = (int)(((int64_t)blockIdx.x * blockDim.x + threadIdx.x) % chunks);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += stride) {
    const int c0 = ch << 3;
    ch += cstep;
    if (ch >= chunks) ch -= chunks;
    float v[8];
    unpack8(x[i], v);
    const float4 s0 = *reinterpret_cast<const float4*>(scale + c0), s1 = *reinterpret_cast<const float4*>(scale + c0 + 4);
    const float4 h0 = *reinterpret_cast<const float4*>(shift + c0), h1 = *reinterpret_cast<const float4*>(shift + c0 + 4);
    const float sc[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
    const float sh[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = leaky(v[k] * sc[k] + sh[k], alpha);
    if (residual) {
      float r[8];
      unpack8(residual[i], r);
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] += r[k];
    }
    y[i] = pack8(v);
  }
}

// ---- column reductions over a bf16 [rows][C] matrix ---------------------------------
// Block = 256 threads laid out as (C/8 channel chunks) x (256/(C/8) row lanes) when
// C/8 <= 256; wider matrices are covered by gridDim.y column blocks of 2048 channels.
// MODE 0: sum(x)            (bias gradient)
// MODE 1: sum(g), sum(g*xhat) with g = dy*act'(x*scale+shift)   (BN backward)
// MODE 2: sum(x), sum(x^2)  (batch statistics of an already materialised conv output)
template <int MODE>
__global__ __launch_bounds__(256) void colreduce_kernel(const uint4* a, const uint4* b, const float* scale,
                                                        const float* shift, const float* mean, const float* rstd,
                                                        float alpha, int64_t rows, int C, int rows_per_block,
                                                        float* part) {
  constexpr int Q = MODE == 0 ? 1 : 2;
  __shared__ float sh[256 * 8 * Q];
  const int chunks = C / 8;
  const int cpb = chunks < 256 ? chunks : 256;  // chunks per block
  const int lanes = 256 / cpb;                  // row lanes
  const int cl = threadIdx.x % cpb, rl = threadIdx.x / cpb;
  const int chunk = blockIdx.y * cpb + cl;
  const int c0 = chunk * 8;
  float acc[8 * Q];
#pragma unroll
  for (int k = 0; k < 8 * Q; ++k) acc[k] = 0.f;
  const bool active = rl < lanes && chunk < chunks;
  if (active) {
    float sc[8], shf[8], mu[8], rs[8];
    if (MODE == 1) {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        sc[k] = scale[c0 + k];
        shf[k] = shift[c0 + k];
        mu[k] = mean[c0 + k];
        rs[k] = rstd[c0 + k];
      }
    }
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    int64_t r1 = r0 + rows_per_block;
    if (r1 > rows) r1 = rows;
    for (int64_t r = r0 + rl; r < r1; r += lanes) {
      float va[8];
      unpack8(a[r * chunks + chunk], va);
      if (MODE == 0) {
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] += va[k];
      } else if (MODE == 2) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          acc[k] += va[k];
          acc[8 + k] += va[k] * va[k];
        }
      } else {
        float vx[8];
        unpack8(b[r * chunks + chunk], vx);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const float z = vx[k] * sc[k] + shf[k];
          const float g = va[k] * (z > 0.f ? 1.f : alpha);
          const float xh = (vx[k] - mu[k]) * rs[k];
          acc[k] += g;
          acc[8 + k] += g * xh;
        }
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 8 * Q; ++k) sh[threadIdx.x * 8 * Q + k] = acc[k];
  __syncthreads();
  if (rl == 0 && chunk < chunks) {
    for (int l = 1; l < lanes; ++l) {
#pragma unroll
      for (int k = 0; k < 8 * Q; ++k) acc[k] += sh[(l * cpb + cl) * 8 * Q + k];
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
#pragma unroll
      for (int q = 0; q < Q; ++q) part[((size_t)blockIdx.x * C + c0 + k) * Q + q] = acc[q * 8 + k];
    }
  }
}

__global__ __launch_bounds__(256) void colsum_finalize_kernel(const float* part, int rows, int C, int out_C, float* out) {
  double r[1];
  int c;
  if (!block_sum_partials<1>(part, rows, C, r, &c)) return;
  if (c < out_C) out[c] = (float)r[0];
}

// dx = scale*(g - mean(g) - xhat*mean(g*xhat)) = scale*g - x*A + Cc with
// A = scale*rstd*mean(g*xhat), Cc = mean*A - scale*mean(g)
__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(const float* part, int rows, int C, double inv_count,
                                                              const float* scale, const float* mean, const float* rstd,
                                                              float* dgamma, float* dbeta, float* cA, float* cC) {
  const int c0 = blockIdx.x * FIN_CPB + threadIdx.x;   // (operands first, as in bn_finalize_kernel)
  const bool writer = threadIdx.x < FIN_CPB && c0 < C;
  const float sc0 = writer ? scale[c0] : 0.f, rs0 = writer ? rstd[c0] : 0.f, mean0 = writer ? mean[c0] : 0.f;
  double r[2];
  int c;
  if (!block_sum_partials<2>(part, rows, C, r, &c)) return;
  dbeta[c] = (float)r[0];
  dgamma[c] = (float)r[1];
  const float c1 = (float)(r[0] * inv_count), c2 = (float)(r[1] * inv_count);
  const float A = sc0 * rs0 * c2;
  cA[c] = A;
  cC[c] = mean0 * A - sc0 * c1;
}

// ---- cross-rank batch statistics (SyncBN option of the data-parallel step) --------------
// The two-stage reductions above are cut after their first stage: the partial rows of a column reduction are
// summed to f64 per channel (local sums), the caller adds them up over the ranks (one small all-reduce), and
// the finalize kernels below take the global sums and the global element count.
template <int Q>
__global__ __launch_bounds__(256) void partial_sums_kernel(const float* part, int rows, int C, double* sums) {
  double r[Q];
  int c;
  if (!block_sum_partials<Q>(part, rows, C, r, &c)) return;
#pragma unroll
  for (int q = 0; q < Q; ++q) sums[(size_t)c * Q + q] = r[q];
}

__global__ __launch_bounds__(256) void bn_finalize_sums_kernel(const double* sums, int C, double inv_count, const float* gamma,
                                                               const float* beta, float* mm, float* mv, float decay, float eps,
                                                               float* scale, float* shift, float* mean_out, float* rstd_out) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const double mean = sums[2 * c] * inv_count;
  double var = sums[2 * c + 1] * inv_count - mean * mean;
  if (var < 0.0) var = 0.0;
  const float meanf = (float)mean, varf = (float)var;
  const float rstd = 1.0f / sqrtf(varf + eps);
  const float sc = gamma[c] * rstd;
  scale[c] = sc;
  shift[c] = beta[c] - meanf * sc;
  if (mean_out) mean_out[c] = meanf;
  if (rstd_out) rstd_out[c] = rstd;
  if (mm) mm[c] = mm[c] * decay + meanf * (1.0f - decay);
  if (mv) mv[c] = mv[c] * decay + varf * (1.0f - decay);
}

// dgamma / dbeta are this rank's own sums (the gradient exchange adds the ranks up later); the correction terms
// of dx use the sums over ALL ranks
__global__ __launch_bounds__(256) void bn_bwd_finalize_sums_kernel(const double* local, const double* global, int C,
                                                                   double inv_count, const float* scale, const float* mean,
                                                                   const float* rstd, float* dgamma, float* dbeta, float* cA,
                                                                   float* cC) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  dbeta[c] = (float)local[2 * c];
  dgamma[c] = (float)local[2 * c + 1];
  const float c1 = (float)(global[2 * c] * inv_count), c2 = (float)(global[2 * c + 1] * inv_count);
  const float A = scale[c] * rstd[c] * c2;
  cA[c] = A;
  cC[c] = mean[c] * A - scale[c] * c1;
}

__device__ __forceinline__ void load8(const float* p, float* o) {
  const float4 a = *reinterpret_cast<const float4*>(p), c = *reinterpret_cast<const float4*>(p + 4);
  o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w; o[4] = c.x; o[5] = c.y; o[6] = c.z; o[7] = c.w;
}
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const uint4* dy, const uint4* x, const float* scale,
                                                           const float* shift, const float* cA, const float* cC,
                                                           uint4* dx, int64_t nvec, int C, float alpha, uint4* sc_grad,
                                                           int sc_accumulate) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int chunks = C >> 3;
  const int cstep = (int)(stride % chunks);
  int ch = (int)(((int64_t)blockIdx.x * blockDim.x + threadIdx.x) % chunks);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += stride) {
    const int c0 = ch << 3;
    ch += cstep;
    if (ch >= chunks) ch -= chunks;
    float g[8], vx[8], sc[8], sh[8], A[8], Cc[8];
    const uint4 dyv = dy[i];
    unpack8(dyv, g);
    if (sc_grad) {
      // the residual shortcut's gradient (+)= dy (res_conv_bn, yolo/yolo3_net_pos.py:148-151): rides on the read of dy
      // this pass makes anyway, in place of a separate add kernel; the same arithmetic as disyolo_add_bf16
      uint4 o = dyv;
      if (sc_accumulate) {
        float a[8], b[8];
        unpack8(dyv, a);
        unpack8(sc_grad[i], b);
#pragma unroll
        for (int k = 0; k < 8; ++k) a[k] += b[k];
        o = pack8(a);
      }
      sc_grad[i] = o;
    }
    unpack8(x[i], vx);
    load8(scale + c0, sc);
    load8(shift + c0, sh);
    load8(cA + c0, A);
    load8(cC + c0, Cc);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float z = vx[k] * sc[k] + sh[k];
      const float gg = g[k] * (z > 0.f ? 1.f : alpha);
      g[k] = sc[k] * gg - vx[k] * A[k] + Cc[k];
    }
    dx[i] = pack8(g);
  }
}

// dst[b,y,x,c] (+)= sum over the 2x2 block of src[b,2y+dy,2x+dx,c_off+c]
__global__ __launch_bounds__(256) void upsample2x_bwd_kernel(const bf16* src, bf16* dst, int B, int Hs, int Ws, int srcC,
                                                             int c_off, int C, int accumulate) {
  const int chunks = C / 8;
  const int Hd = Hs / 2, Wd = Ws / 2;
  const int64_t nvec = (int64_t)B * Hd * Wd * chunks;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += stride) {
    const int ch = (int)(i % chunks);
    int64_t t = i / chunks;
    const int x = (int)(t % Wd);
    t /= Wd;
    const int y = (int)(t % Hd);
    const int b = (int)(t / Hd);
    float acc[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] = 0.f;
#pragma unroll
    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
      for (int dx = 0; dx < 2; ++dx) {
        const size_t off = (((size_t)b * Hs + 2 * y + dy) * Ws + 2 * x + dx) * srcC + c_off + ch * 8;
        float v[8];
        unpack8(*reinterpret_cast<const uint4*>(src + off), v);
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] += v[k];
      }
    uint4* d = reinterpret_cast<uint4*>(dst + (size_t)i * 8);
    if (accumulate) {
      float v[8];
      unpack8(*d, v);
#pragma unroll
      for (int k = 0; k < 8; ++k) acc[k] += v[k];
    }
    *d = pack8(acc);
  }
}

// DISYOLO_EXP_BN (timing experiment, results are wrong): bit 0 skip the forward finalize, 1 the forward apply,
// 2 the backward column reduction, 3 the backward finalize, 4 the backward apply
int exp_bn() {
  static const int v = [] {
    const int e = getenv("DISYOLO_EXP_BN") ? atoi(getenv("DISYOLO_EXP_BN")) : 0;
    if (e) fprintf(stderr, "disyolo: DISYOLO_EXP_BN=%d -- batch-norm kernels are skipped / doubled, RESULTS ARE WRONG (timing experiment)\n", e);
    return e;
  }();
  return v;
}
int grid_for(int64_t n) {
  int64_t g = (n + 255) / 256;
  if (g > 256 * 8) g = 256 * 8;
  if (g < 1) g = 1;
  return (int)g;
}

// rows of partials a column reduction over `rows` produces
int colreduce_blocks(int64_t rows, int C) {
  const int chunks = C / 8;
  const int cpb = chunks < 256 ? chunks : 256;
  const int lanes = 256 / cpb;
  const int colblocks = ceil_div(chunks, cpb);
  int64_t want = 2048 / colblocks;  // target ~2048 blocks in flight
  if (want < 1) want = 1;
  int64_t rpb = (rows + want - 1) / want;
  const int64_t min_rpb = (int64_t)lanes * 4;
  if (rpb < min_rpb) rpb = min_rpb;
  return (int)((rows + rpb - 1) / rpb);
}
int colreduce_rpb(int64_t rows, int C) {
  const int nb = colreduce_blocks(rows, C);
  return (int)((rows + nb - 1) / nb);
}

}  // namespace

extern "C" int disyolo_bn_finalize(const float* stats, int rows, int C, int64_t count, const float* gamma,
                                   const float* beta, float* moving_mean, float* moving_var, float decay, float eps,
                                   float* scale, float* shift, float* mean, float* rstd, void* stream) {
  DY_REQUIRE(stats && gamma && beta && scale && shift && rows > 0 && C > 0 && count > 0, "bn_finalize: bad args");
  DY_RECORD_OR_RUN([=](void* s) { return disyolo_bn_finalize(stats, rows, C, count, gamma, beta, moving_mean, moving_var, decay, eps, scale, shift, mean, rstd, s); });
  if (exp_bn() & 1) return DISYOLO_OK;
  if (exp_bn() & 32)   // (twice: the second launch's cost is what one costs; decay applied twice -> results differ)
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(ceil_div(C, FIN_CPB)), dim3(256), 0, (hipStream_t)stream, stats, rows, C,
                       1.0 / (double)count, gamma, beta, moving_mean, moving_var, decay, eps, scale, shift, mean, rstd);
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(ceil_div(C, FIN_CPB)), dim3(256), 0, (hipStream_t)stream, stats, rows, C,
                     1.0 / (double)count, gamma, beta, moving_mean, moving_var, decay, eps, scale, shift, mean, rstd);
  DY_CHECK_LAUNCH();
  return DISYOLO_OK;
}

extern "C" int disyolo_bn_fold(const float* gamma, const float* beta, const float* moving_mean,
                               const float* moving_var, float eps, float* scale, float* shift, int C, void* stream) {
  DY_REQUIRE(gamma && beta && moving_mean && moving_var && scale && shift && C > 0, "bn_fold: bad args");
  DY_RECORD_OR_RUN([=](void* s) { return disyolo_bn_fold(gamma, beta, moving_mean, moving_var, eps, scale, shift, C, s); });
  hipLaunchKernelGGL(bn_fold_kernel, dim3(ceil_div(C, 256)), dim3(256), 0, (hipStream_t)stream, gamma, beta,
                     moving_mean, moving_var, eps, scale, shift, C);
  DY_CHECK_LAUNCH();
  return DISYOLO_OK;
}

extern "C" int disyolo_bn_act_fwd(const void* x, const float* scale, const float* shift, const void* residual, void* y,
                                  int64_t rows, int C, float alpha, void* stream) {
  DY_REQUIRE(x && scale && shift && y && rows > 0 && C > 0 && C % 8 == 0, "bn_act_fwd: bad args (C %% 8 == 0)");
  DY_RECORD_OR_RUN([=](void* s) { return disyolo_bn_act_fwd(x, scale, shift, residual, y, rows, C, alpha, s); });
  const int64_t nvec = rows * C / 8;
  if (exp_bn() & 2) return DISYOLO_OK;
  hipLaunchKernelGGL(bn_act_fwd_kernel, dim3(grid_for(nvec)), dim3(256), 0, (hipStream_t)stream, (const uint4*)x,
                     scale, shift, (const uint4*)residual, (uint4*)y, nvec, C, alpha);
  DY_CHECK_LAUNCH();
  return DISYOLO_OK;
}

extern "C" size_t disyolo_bn_act_bwd_workspace(int64_t rows, int C) {
  if (rows <= 0 || C <= 0 || C % 8) return 0;
  return ((size_t)colreduce_blocks(rows, C) * C * 2 + 2 * (size_t)C) * sizeof(float);
}

extern "C" int disyolo_bn_act_bwd(const void* dy, const void* x, const float* scale, const float* shift,
                                  const float* mean, const float* rstd, void* dx, float* dgamma, float* dbeta,
                                  int64_t rows, int C, float alpha, void* shortcut_grad, int shortcut_accumulate,
                                  void* workspace, size_t workspace_bytes, void* stream) {
  DY_REQUIRE(dy && x && scale && shift && mean && rstd && dx && dgamma && dbeta, "bn_act_bwd: null pointer");
  DY_REQUIRE(rows > 0 && C > 0 && C % 8 == 0, "bn_act_bwd: bad shape");
  if (workspace_bytes < disyolo_bn_act_bwd_workspace(rows, C) || !workspace) {
    disyolo_set_error("bn_act_bwd: workspace too small");
    return DISYOLO_E_WORKSPACE;
  }
  DY_RECORD_OR_RUN([=](void* s) { return disyolo_bn_act_bwd(dy, x, scale, shift, mean, rstd, dx, dgamma, dbeta, rows, C, alpha, shortcut_grad, shortcut_accumulate, workspace, workspace_bytes, s); });
  hipStream_t s = (hipStream_t)stream;
  const int nb = colreduce_blocks(rows, C), rpb = colreduce_rpb(rows, C);
  const int chunks = C / 8, cpb = chunks < 256 ? chunks : 256;
  float* part = (float*)workspace;
  float* c1 = part + (size_t)nb * C * 2;
  float* c2 = c1 + C;
  if (!(exp_bn() & 4))
  hipLaunchKernelGGL(colreduce_kernel<1>, dim3(nb, ceil_div(chunks, cpb)), dim3(256), 0, s, (const uint4*)dy,
                     (const uint4*)x, scale, shift, mean, rstd, alpha, rows, C, rpb, part);
  DY_CHECK_LAUNCH();
  if (exp_bn() & 64)
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(ceil_div(C, FIN_CPB)), dim3(256), 0, s, part, nb, C, 1.0 / (double)rows,
                       scale, mean, rstd, dgamma, dbeta, c1, c2);
  if (!(exp_bn() & 8))
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(ceil_div(C, FIN_CPB)), dim3(256), 0, s, part, nb, C, 1.0 / (double)rows,
                     scale, mean, rstd, dgamma, dbeta, c1, c2);
  DY_CHECK_LAUNCH();
  const int64_t nvec = rows * C / 8;
  if (!(exp_bn() & 16))
  hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(grid_for(nvec)), dim3(256), 0, s, (const uint4*)dy, (const uint4*)x,
                     scale, shift, c1, c2, (uint4*)dx, nvec, C, alpha, (uint4*)shortcut_grad, shortcut_accumulate);
  DY_CHECK_LAUNCH();
  return DISYOLO_OK;
}

// bn_act_bwd with the column reduction already done: `partials` f32 [part_rows][C][2] holds (sum g, sum g*xhat)
// per channel over disjoint row sets (what a data-gradient conv with DISYOLO_CONV_BN_BWD_STATS wrote)
extern "C" size_t disyolo_bn_act_bwd_partials_workspace(int C) { return C > 0 ? 2 * (size_t)C * sizeof(float) : 0; }

extern "C" int disyolo_bn_act_bwd_partials(const void* dy, const void* x, const float* scale, const float* shift,
                                           const float* mean, const float* rstd, void* dx, float* dgamma, float* dbeta,
                                           int64_t rows, int C, float alpha, const float* partials, int part_rows,
                                           void* shortcut_grad, int shortcut_accumulate, void* workspace,
                                           size_t workspace_bytes, void* stream) {
  DY_REQUIRE(dy && x && scale && shift && mean && rstd && dx && dgamma && dbeta && partials, "bn_act_bwd_partials: null pointer");
  DY_REQUIRE(rows > 0 && C > 0 && C % 8 == 0 && part_rows > 0, "bn_act_bwd_partials: bad shape");
  if (workspace_bytes < disyolo_bn_act_bwd_partials_workspace(C) || !workspace) {
    disyolo_set_error("bn_act_bwd_partials: workspace too small");
    return DISYOLO_E_WORKSPACE;
  }
  DY_RECORD_OR_RUN([=](void* s) {
    return disyolo_bn_act_bwd_partials(dy, x, scale, shift, mean, rstd, dx, dgamma, dbeta, rows, C, alpha, partials, part_rows,
                                       shortcut_grad, shortcut_accumulate, workspace, workspace_bytes, s);
  });
  hipStream_t s = (hipStream_t)stream;
  float* c1 = (float*)workspace;
  float* c2 = c1 + C;
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(ceil_div(C, FIN_CPB)), dim3(256), 0, s, partials, part_rows, C,
                     1.0 / (double)rows, scale, mean, rstd, dgamma, dbeta, c1, c2);
  DY_CHECK_LAUNCH();
  const int64_t nvec = rows * C / 8;
  hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(grid_for(nvec)), dim3(256), 0, s, (const uint4*)dy, (const uint4*)x,
                     scale, shift, c1, c2, (uint4*)dx, nvec, C, alpha, (uint4*)shortcut_grad, shortcut_accumulate);
  DY_CHECK_LAUNCH();
  return DISYOLO_OK;
}

// ---- SyncBN building blocks: the phases of bn_finalize / bn_act_bwd as separate calls, so the caller can add
// the per-channel sums up over the data-parallel ranks between them (f64 [C][2], one all-reduce each) ----
extern "C" int disyolo_bn_partial_sums(const float* partials, int rows, int C, double* sums, void* stream) {
  DY_REQUIRE(partials && sums && rows > 0 && C > 0, "bn_partial_sums: bad args");
  DY_RECORD_OR_RUN([=](void* s) { return disyolo_bn_partial_sums(partials, rows, C, sums, s); });
  hipLaunchKernelGGL(partial_sums_kernel<2>, dim3(ceil_div(C, FIN_CPB)), dim3(256), 0, (hipStream_t)stream, partials, rows, C, sums);
  DY_CHECK_LAUNCH();
  return DISYOLO_OK;
}

extern "C" int disyolo_bn_finalize_sums(const double* sums, int C, int64_t count, const float* gamma, const float* beta,
                                        float* moving_mean, float* moving_var, float decay, float eps, float* scale,
                                        float* shift, float* mean, float* rstd, void* stream) {
  DY_REQUIRE(sums && gamma && beta && scale && shift && C > 0 && count > 0, "bn_finalize_sums: bad args");
  DY_RECORD_OR_RUN([=](void* s) { return disyolo_bn_finalize_sums(sums, C, count, gamma, beta, moving_mean, moving_var, decay, eps, scale, shift, mean, rstd, s); });
  hipLaunchKernelGGL(bn_finalize_sums_kernel, dim3(ceil_div(C, 256)), dim3(256), 0, (hipStream_t)stream, sums, C,
                     1.0 / (double)count, gamma, beta, moving_mean, moving_var, decay, eps, scale, shift, mean, rstd);
  DY_CHECK_LAUNCH();
  return DISYOLO_OK;
}

extern "C" int disyolo_bn_bwd_reduce_rows(int64_t rows, int C) { return (rows > 0 && C > 0 && C % 8 == 0) ? colreduce_blocks(rows, C) : 0; }

// first phase of bn_act_bwd: (sum g, sum g*xhat) partial rows -> this rank's f64 sums [C][2]
extern "C" int disyolo_bn_bwd_reduce(const void* dy, const void* x, const float* scale, const float* shift, const float* mean,
                                     const float* rstd, int64_t rows, int C, float alpha, double* sums, void* workspace,
                                     size_t workspace_bytes, void* stream) {
  DY_REQUIRE(dy && x && scale && shift && mean && rstd && sums && rows > 0 && C > 0 && C % 8 == 0, "bn_bwd_reduce: bad args");
  const int nb = colreduce_blocks(rows, C), rpb = colreduce_rpb(rows, C);
  if (!workspace || workspace_bytes < (size_t)nb * C * 2 * sizeof(float)) {
    disyolo_set_error("bn_bwd_reduce: workspace too small");
    return DISYOLO_E_WORKSPACE;
  }
  DY_RECORD_OR_RUN([=](void* s) { return disyolo_bn_bwd_reduce(dy, x, scale, shift, mean, rstd, rows, C, alpha, sums, workspace, workspace_bytes, s); });
  hipStream_t s = (hipStream_t)stream;
  const int chunks = C / 8, cpb = chunks < 256 ? chunks : 256;
  float* part = (float*)workspace;
  hipLaunchKernelGGL(colreduce_kernel<1>, dim3(nb, ceil_div(chunks, cpb)), dim3(256), 0, s, (const uint4*)dy, (const uint4*)x,
                     scale, shift, mean, rstd, alpha, rows, C, rpb, part);
  DY_CHECK_LAUNCH();
  hipLaunchKernelGGL(partial_sums_kernel<2>, dim3(ceil_div(C, FIN_CPB)), dim3(256), 0, s, (const float*)part, nb, C, sums);
  DY_CHECK_LAUNCH();
  return DISYOLO_OK;
}

// second phase: dgamma/dbeta from this rank's sums, dx with the correction terms of the sums over all ranks
// (`count` = rows over all ranks); cA/cC scratch: 2*C floats of workspace
extern "C" int disyolo_bn_bwd_apply_sums(const void* dy, const void* x, const float* scale, const float* shift, const float* mean,
                                         const float* rstd, const double* local_sums, const double* global_sums, int64_t count,
                                         void* dx, float* dgamma, float* dbeta, int64_t rows, int C, float alpha,
                                         void* shortcut_grad, int shortcut_accumulate, void* workspace,
                                         size_t workspace_bytes, void* stream) {
  DY_REQUIRE(dy && x && scale && shift && mean && rstd && local_sums && global_sums && dx && dgamma && dbeta,
             "bn_bwd_apply_sums: null pointer");
  DY_REQUIRE(rows > 0 && count >= rows && C > 0 && C % 8 == 0, "bn_bwd_apply_sums: bad shape");
  if (!workspace || workspace_bytes < 2 * (size_t)C * sizeof(float)) {
    disyolo_set_error("bn_bwd_apply_sums: workspace too small");
    return DISYOLO_E_WORKSPACE;
  }
  DY_RECORD_OR_RUN([=](void* s) {
    return disyolo_bn_bwd_apply_sums(dy, x, scale, shift, mean, rstd, local_sums, global_sums, count, dx, dgamma, dbeta, rows, C,
                                     alpha, shortcut_grad, shortcut_accumulate, workspace, workspace_bytes, s);
  });
  hipStream_t s = (hipStream_t)stream;
  float* c1 = (float*)workspace;
  float* c2 = c1 + C;
  hipLaunchKernelGGL(bn_bwd_finalize_sums_kernel, dim3(ceil_div(C, 256)), dim3(256), 0, s, local_sums, global_sums, C,
                     1.0 / (double)count, scale, mean, rstd, dgamma, dbeta, c1, c2);
  DY_CHECK_LAUNCH();
  const int64_t nvec = rows * C / 8;
  hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(grid_for(nvec)), dim3(256), 0, s, (const uint4*)dy, (const uint4*)x, scale, shift,
                     c1, c2, (uint4*)dx, nvec, C, alpha, (uint4*)shortcut_grad, shortcut_accumulate);
  DY_CHECK_LAUNCH();
  return DISYOLO_OK;
}

extern "C" int disyolo_upsample2x_bwd(const void* src, void* dst, int B, int Hs, int Ws, int src_C, int c_off, int C,
                                      int accumulate, void* stream) {
  DY_REQUIRE(src && dst && B > 0 && Hs > 0 && Ws > 0 && Hs % 2 == 0 && Ws % 2 == 0, "upsample2x_bwd: bad sizes");
  DY_REQUIRE(C > 0 && C % 8 == 0 && c_off % 8 == 0 && c_off + C <= src_C && src_C % 8 == 0, "upsample2x_bwd: bad channels");
  DY_RECORD_OR_RUN([=](void* s) { return disyolo_upsample2x_bwd(src, dst, B, Hs, Ws, src_C, c_off, C, accumulate, s); });
  const int64_t nvec = (int64_t)B * (Hs / 2) * (Ws / 2) * (C / 8);
  hipLaunchKernelGGL(upsample2x_bwd_kernel, dim3(grid_for(nvec)), dim3(256), 0, (hipStream_t)stream, (const bf16*)src,
                     (bf16*)dst, B, Hs, Ws, src_C, c_off, C, accumulate);
  DY_CHECK_LAUNCH();
  return DISYOLO_OK;
}

extern "C" size_t disyolo_colsum_workspace(int64_t rows, int C) {
  if (rows <= 0 || C <= 0 || C % 8) return 0;
  return (size_t)colreduce_blocks(rows, C) * C * sizeof(float);
}

extern "C" int disyolo_colsum(const void* x, float* out, int64_t rows, int C, int out_C, void* workspace,
                              size_t workspace_bytes, void* stream) {
  DY_REQUIRE(x && out && rows > 0 && C > 0 && C % 8 == 0 && out_C > 0 && out_C <= C, "colsum: bad args (C %% 8 == 0)");
  if (workspace_bytes < disyolo_colsum_workspace(rows, C) || !workspace) {
    disyolo_set_error("colsum: workspace too small");
    return DISYOLO_E_WORKSPACE;
  }
  DY_RECORD_OR_RUN([=](void* s) { return disyolo_colsum(x, out, rows, C, out_C, workspace, workspace_bytes, s); });
  hipStream_t s = (hipStream_t)stream;
  const int nb = colreduce_blocks(rows, C), rpb = colreduce_rpb(rows, C);
  const int chunks = C / 8, cpb = chunks < 256 ? chunks : 256;
  hipLaunchKernelGGL(colreduce_kernel<0>, dim3(nb, ceil_div(chunks, cpb)), dim3(256), 0, s, (const uint4*)x,
                     (const uint4*)nullptr, nullptr, nullptr, nullptr, nullptr, 0.f, rows, C, rpb, (float*)workspace);
  DY_CHECK_LAUNCH();
  hipLaunchKernelGGL(colsum_finalize_kernel, dim3(ceil_div(C, FIN_CPB)), dim3(256), 0, s, (const float*)workspace, nb, C,
                     out_C, out);
  DY_CHECK_LAUNCH();
  return DISYOLO_OK;
}

extern "C" int disyolo_colstats_rows(int64_t rows, int C) {
  if (rows <= 0 || C <= 0 || C % 8) return DISYOLO_E_ARG;
  return colreduce_blocks(rows, C);
}

extern "C" int disyolo_colstats(const void* x, float* stats, int64_t rows, int C, void* stream) {
  DY_REQUIRE(x && stats && rows > 0 && C > 0 && C % 8 == 0, "colstats: bad args (C %% 8 == 0)");
  DY_RECORD_OR_RUN([=](void* s) { return disyolo_colstats(x, stats, rows, C, s); });
  const int nb = colreduce_blocks(rows, C), rpb = colreduce_rpb(rows, C);
  const int chunks = C / 8, cpb = chunks < 256 ? chunks : 256;
  hipLaunchKernelGGL(colreduce_kernel<2>, dim3(nb, ceil_div(chunks, cpb)), dim3(256), 0, (hipStream_t)stream,
                     (const uint4*)x, (const uint4*)nullptr, nullptr, nullptr, nullptr, nullptr, 0.f, rows, C, rpb,
                     stats);
  DY_CHECK_LAUNCH();
  return DISYOLO_OK;
}
