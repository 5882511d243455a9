// Fused loss + gradient kernels: YOLO detection loss and the position-sensitive mask loss.
//
// Replace loss_yolo (yolo/yolo3_net_pos.py:631-747) and loss_mask / overlaps_graph /
// assemble_kmask_from_box (:750-860, 954-975) together with TF's autodiff of them.  The
// reference materialises [R,288,288,9] one-hot masks and a tiled copy of the score maps
// per image; here one thread owns one score-map pixel and walks the (<=10) RoIs.
//
// Gradients wrt head logits / score maps are written as bf16 rows padded to 32 channels
// (zeros beyond the real channels) so they feed the MFMA data/weight-gradient convs
// directly.  Scalar losses use per-block partials + a fixed-order final sum.
#include <array>
#include "common.h"
#include "runtime.h"

namespace {

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }
// tf.nn.sigmoid_cross_entropy_with_logits: max(x,0) - x*z + log1p(exp(-|x|))
__device__ __forceinline__ float sigmoid_ce(float z, float x) {
  return fmaxf(x, 0.f) - x * z + log1pf(expf(-fabsf(x)));
}

constexpr int DL_LD = 32;  // padded channel count of the dlogits / dscore rows

struct YoloLossParams {
  const float* logits;  // [B,g,g,3,D]
  const float* labels;  // [B,g,g,3,D]
  const float* true_boxes;  // [B,G,5] xc,yc,w,h,cls (normalised)
  bf16* dlogits;        // [B,g,g,DL_LD]
  float* partial;       // [gridDim.y*gridDim.x][5]
  int B, g, C, G, S;
  float aw[3], ah[3];
  float ignore_thresh, obj_scale, noobj_scale, class_scale, coord_scale;
  float inv_B;
};

// one thread = one (cell, anchor); blockIdx.y = image.  The three scales run as ONE launch (round 5: three dependent launches
// at the forward / backward boundary of the main lane were 44 us where the longest scale takes 24): block bx of scale s,
// whose grid is gx blocks wide -- the partial rows keep the layout of the three separate launches, so the sums are the same
// bit for bit.
struct YoloLoss3 {
  YoloLossParams p[3];
  int gx[3];
};
__device__ __forceinline__ void yolo_loss_body(const YoloLossParams& p, const int bx, const int gx) {
  __shared__ float s_tb[64 * 4];
  __shared__ float s_red[4][5];
  const int b = blockIdx.y;
  for (int i = threadIdx.x; i < p.G * 4; i += 256) {
    const int gidx = i >> 2, k = i & 3;
    s_tb[i] = p.true_boxes[((size_t)b * p.G + gidx) * 5 + k];
  }
  __syncthreads();
  const int D = 5 + p.C;
  const int ncell = p.g * p.g;
  const int idx = bx * 256 + threadIdx.x;
  float l_obj = 0.f, l_noobj = 0.f, l_cls = 0.f, l_xy = 0.f, l_wh = 0.f;
  if (idx < ncell * 3) {
    const int a = idx % 3, cell = idx / 3;
    const int x = cell % p.g, y = cell / p.g;
    const size_t base = (((size_t)b * ncell + cell) * 3 + a) * D;
    const float* t = p.logits + base;
    const float* lab = p.labels + base;
    const float gf = (float)p.g, net = (float)p.S;
    const float sx = sigmoidf_(t[0]), sy = sigmoidf_(t[1]);
    // decoded normalised box (interpret_output, :493-506)
    const float bxc = ((float)x + sx) / gf, byc = ((float)y + sy) / gf;
    const float bw = expf(t[2]) * p.aw[a] / net, bh = expf(t[3]) * p.ah[a] / net;
    const float pminx = bxc - bw / 2.f, pmaxx = bxc + bw / 2.f, pminy = byc - bh / 2.f, pmaxy = byc + bh / 2.f;
    const float parea = bw * bh;
    float best = 0.f;  // iou is clipped to [0,1], so the max over >=1 boxes is >= 0
    for (int k = 0; k < p.G; ++k) {
      const float txc = s_tb[k * 4 + 0], tyc = s_tb[k * 4 + 1], tw = s_tb[k * 4 + 2], th = s_tb[k * 4 + 3];
      const float iw = fmaxf(fminf(pmaxx, txc + tw / 2.f) - fmaxf(pminx, txc - tw / 2.f), 0.f);
      const float ih = fmaxf(fminf(pmaxy, tyc + th / 2.f) - fmaxf(pminy, tyc - th / 2.f), 0.f);
      const float inter = iw * ih;
      const float uni = fmaxf(parea + tw * th - inter, 1e-10f);
      const float iou = fminf(fmaxf(inter / uni, 0.f), 1.f);
      best = fmaxf(best, iou);
    }
    const float ignore = best < p.ignore_thresh ? 1.f : 0.f;
    const float obj = lab[4];
    const float conf = t[4];
    const float ce = sigmoid_ce(obj, conf);
    l_obj = obj * ce * p.obj_scale;
    l_noobj = ignore * (1.f - obj) * ce * p.noobj_scale;
    const float sconf = sigmoidf_(conf);
    float d[8 + 8];  // up to 16 logits per anchor (C <= 11)
    const float dce = sconf - obj;
    d[4] = (obj * p.obj_scale + ignore * (1.f - obj) * p.noobj_scale) * dce * p.inv_B;
    // class: sparse softmax CE against argmax(label[5:]) (first max)
    int tc = 0;
    float lm = lab[5];
    for (int c = 1; c < p.C; ++c)
      if (lab[5 + c] > lm) {
        lm = lab[5 + c];
        tc = c;
      }
    float mx = t[5];
    for (int c = 1; c < p.C; ++c) mx = fmaxf(mx, t[5 + c]);
    float den = 0.f;
    for (int c = 0; c < p.C; ++c) den += expf(t[5 + c] - mx);
    const float lse = mx + logf(den);
    l_cls = obj * (lse - t[5 + tc]) * p.class_scale;
    for (int c = 0; c < p.C; ++c) {
      const float pr = expf(t[5 + c] - lse);
      d[5 + c] = obj * p.class_scale * (pr - (c == tc ? 1.f : 0.f)) * p.inv_B;
    }
    // coordinates (:706-726)
    const float tcx = lab[0] * gf - (float)x, tcy = lab[1] * gf - (float)y;
    const float ttw = fminf(fmaxf(logf(lab[2] * net / p.aw[a]), -100.f), 100.f);
    const float tth = fminf(fmaxf(logf(lab[3] * net / p.ah[a]), -100.f), 100.f);
    const float whs = 2.f - lab[2] * lab[3];
    const float whs2 = whs * whs * p.coord_scale;
    const float dx_ = obj * (sx - tcx), dy_ = obj * (sy - tcy);
    const float dw_ = obj * (t[2] - ttw), dh_ = obj * (t[3] - tth);
    l_xy = (dx_ * dx_ + dy_ * dy_) * whs2;
    l_wh = (dw_ * dw_ + dh_ * dh_) * whs2;
    d[0] = 2.f * obj * dx_ * whs2 * sx * (1.f - sx) * p.inv_B;
    d[1] = 2.f * obj * dy_ * whs2 * sy * (1.f - sy) * p.inv_B;
    d[2] = 2.f * obj * dw_ * whs2 * p.inv_B;
    d[3] = 2.f * obj * dh_ * whs2 * p.inv_B;
    bf16* o = p.dlogits + ((size_t)b * ncell + cell) * DL_LD + a * D;
    for (int k = 0; k < D; ++k) o[k] = (bf16)d[k];
    if (a == 2)
      for (int k = 3 * D; k < DL_LD; ++k) p.dlogits[((size_t)b * ncell + cell) * DL_LD + k] = (bf16)0.f;
  }
  l_obj = wave_sum(l_obj);
  l_noobj = wave_sum(l_noobj);
  l_cls = wave_sum(l_cls);
  l_xy = wave_sum(l_xy);
  l_wh = wave_sum(l_wh);
  if ((threadIdx.x & 63) == 0) {
    float* r = s_red[threadIdx.x >> 6];
    r[0] = l_obj; r[1] = l_noobj; r[2] = l_cls; r[3] = l_xy; r[4] = l_wh;
  }
  __syncthreads();
  if (threadIdx.x < 5) {
    const float v = s_red[0][threadIdx.x] + s_red[1][threadIdx.x] + s_red[2][threadIdx.x] + s_red[3][threadIdx.x];
    p.partial[((size_t)blockIdx.y * gx + bx) * 5 + threadIdx.x] = v;
  }
}
__global__ __launch_bounds__(256) void yolo_loss_kernel(YoloLoss3 P) {
  int bx = blockIdx.x, s = 0;
  if (bx >= P.gx[0]) {
    bx -= P.gx[0];
    s = 1;
    if (bx >= P.gx[1]) {
      bx -= P.gx[1];
      s = 2;
    }
  }
  yolo_loss_body(P.p[s], bx, P.gx[s]);
}

__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// losses[0..4] = obj, noobj, class, xy, wh summed over the three scales and averaged over
// the batch; [5] = conf, [6] = coord, [7] = conf + class + coord.  5 waves, one per term.
__global__ __launch_bounds__(320) void yolo_loss_final_kernel(const float* partial, int n, float inv_B, float* losses) {
  const int q = threadIdx.x >> 6, lane = threadIdx.x & 63;
  __shared__ double s[5];
  double acc = 0.0;
  for (int i = lane; i < n; i += 64) acc += (double)partial[(size_t)i * 5 + q];
  acc = wave_sum_d(acc);
  if (lane == 0) {
    s[q] = acc * inv_B;
    losses[q] = (float)s[q];
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    losses[5] = (float)(s[0] + s[1]);
    losses[6] = (float)(s[3] + s[4]);
    losses[7] = (float)(s[0] + s[1] + s[2] + s[3] + s[4]);
  }
}

// ---- tf.random_shuffle of the proposal / GT-box order (yolo/yolo3_net_pos.py:781-782) ----
// A uniformly random permutation of 0..n-1 per image: each thread draws one counter-based
// 32-bit key (hash of seed, step, image, index), the rank of the key is its position.  The step
// count is read from device memory, so a recorded step reshuffles on every replay.
__device__ __forceinline__ unsigned hash32(unsigned x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}
__global__ __launch_bounds__(64) void shuffle_perm_kernel(int* perm_a, int na, int* perm_b, int nb, unsigned seed,
                                                          const int64_t* step) {
  __shared__ unsigned keys[64];
  const int b = blockIdx.x, t = threadIdx.x;
  const unsigned st = step ? (unsigned)*step : 0u;
  for (int which = 0; which < 2; ++which) {
    const int n = which ? nb : na;
    int* out = (which ? perm_b : perm_a) + (size_t)b * n;
    if (t < n) keys[t] = hash32(hash32(seed + 0x9e3779b9u * st) ^ hash32((unsigned)(b * 2 + which) * 0x85ebca6bu + t));
    __syncthreads();
    if (t < n) {
      int rank = 0;
      for (int j = 0; j < n; ++j) rank += (keys[j] < keys[t]) || (keys[j] == keys[t] && j < t);
      out[rank] = t;
    }
    __syncthreads();
  }
}

// ---- mask-loss RoI selection (yolo/yolo3_net_pos.py:757-796, 842) ------------------
constexpr int ROI_MAX = 16;
constexpr int ROI_W = 12;  // gy0..3, gx0..3, gt_row, area, valid, pad
__device__ __forceinline__ void bin_edges3(float lo, float hi, int e[4]) {
  const float sub = (hi - lo) / 3.f;
  e[0] = (int)lo;
  e[1] = (int)rintf(lo + sub);
  e[2] = (int)rintf(lo + 2.f * sub);
  e[3] = (int)hi;
}
// One 64-thread block per image: the image's detections, ground-truth boxes and permutations are staged in
// LDS by all lanes, then lane 0 runs the (inherently sequential, order-defining) selection on them -- the
// single-thread version spent 40 us in dependent global loads.
__global__ __launch_bounds__(64) void mask_rois_kernel(const float* det_g, int max_det, const float* true_boxes_g, int G,
                                                       const int* perm_det_g, const int* perm_gt_g, int B, int Sm, int n_det,
                                                       int n_gt, float iou_thr, int* rois, int* roi_count) {
  __shared__ float s_det[64 * 6];
  __shared__ float s_tb[64 * 5];
  __shared__ int s_pd[64], s_pg[64];
  const int b = blockIdx.x;
  if (b >= B) return;
  for (int i = threadIdx.x; i < max_det * 6 && i < 64 * 6; i += 64) s_det[i] = det_g[(size_t)b * max_det * 6 + i];
  for (int i = threadIdx.x; i < G * 5 && i < 64 * 5; i += 64) s_tb[i] = true_boxes_g[(size_t)b * G * 5 + i];
  for (int i = threadIdx.x; i < max_det && i < 64; i += 64) s_pd[i] = perm_det_g ? perm_det_g[b * max_det + i] : i;
  for (int i = threadIdx.x; i < G && i < 64; i += 64) s_pg[i] = perm_gt_g ? perm_gt_g[b * G + i] : i;
  __syncthreads();
  if (threadIdx.x != 0) return;
  const float* det = s_det;            // image b's rows, staged
  const float* true_boxes = s_tb;
  const int* perm_det = s_pd;
  const int* perm_gt = s_pg;
  // trimmed lists (rows whose |coords| sum is non-zero), in row order
  int prow[64], grow[64];
  int np = 0, ng = 0;
  for (int r = 0; r < max_det && r < 64; ++r) {
    const float* d = det + ((size_t)r) * 6;
    if (fabsf(d[0]) + fabsf(d[1]) + fabsf(d[2]) + fabsf(d[3]) != 0.f) prow[np++] = r;
  }
  for (int r = 0; r < G && r < 64; ++r) {
    const float* t = true_boxes + ((size_t)r) * 5;
    if (fabsf(t[0]) + fabsf(t[1]) + fabsf(t[2]) + fabsf(t[3]) != 0.f) grow[ng++] = r;
  }
  // shuffled selection: first n_det proposals and first n_gt GT boxes in permuted order
  float rb[ROI_MAX][4];
  int nr = 0;
  for (int q = 0, taken = 0; q < max_det && taken < n_det; ++q) {
    const int j = perm_det[q];
    if (j < 0 || j >= np) continue;
    const float* d = det + ((size_t)prow[j]) * 6;
    rb[nr][0] = d[0]; rb[nr][1] = d[1]; rb[nr][2] = d[2]; rb[nr][3] = d[3];
    ++nr; ++taken;
  }
  for (int q = 0, taken = 0; q < G && taken < n_gt; ++q) {
    const int j = perm_gt[q];
    if (j < 0 || j >= ng) continue;
    const float* t = true_boxes + ((size_t)grow[j]) * 5;
    rb[nr][0] = t[1] - t[3] / 2.f; rb[nr][1] = t[0] - t[2] / 2.f;
    rb[nr][2] = t[1] + t[3] / 2.f; rb[nr][3] = t[0] + t[2] / 2.f;
    ++nr; ++taken;
  }
  int cnt = 0;
  int* out = rois + (size_t)b * ROI_MAX * ROI_W;
  for (int r = 0; r < nr; ++r) {
    float best = -INFINITY;
    int arg = 0;
    for (int j = 0; j < ng; ++j) {
      const float* t = true_boxes + ((size_t)grow[j]) * 5;
      const float gy1 = t[1] - t[3] / 2.f, gx1 = t[0] - t[2] / 2.f, gy2 = t[1] + t[3] / 2.f, gx2 = t[0] + t[2] / 2.f;
      const float y1 = fmaxf(rb[r][0], gy1), x1 = fmaxf(rb[r][1], gx1);
      const float y2 = fminf(rb[r][2], gy2), x2 = fminf(rb[r][3], gx2);
      const float inter = fmaxf(x2 - x1, 0.f) * fmaxf(y2 - y1, 0.f);
      const float a1 = (rb[r][2] - rb[r][0]) * (rb[r][3] - rb[r][1]);
      const float a2 = (gy2 - gy1) * (gx2 - gx1);
      const float iou = inter / (a1 + a2 - inter);
      if (iou > best) {  // first maximum wins (tf.argmax); NaN never wins
        best = iou;
        arg = j;
      }
    }
    if (ng > 0 && best >= iou_thr) {
      const float sz = (float)Sm;
      const float y1 = rintf(rb[r][0] * sz), x1 = rintf(rb[r][1] * sz);
      const float y2 = rintf(rb[r][2] * sz), x2 = rintf(rb[r][3] * sz);
      int gy[4], gx[4];
      bin_edges3(y1, y2, gy);
      bin_edges3(x1, x2, gx);
      int* o = out + cnt * ROI_W;
      for (int k = 0; k < 4; ++k) {
        o[k] = gy[k];
        o[4 + k] = gx[k];
      }
      o[8] = grow[arg];
      // mask_object pixel count (:848): sum over the k*k bins, clipped to the map
      int area = 0;
      for (int by = 0; by < 3; ++by)
        for (int bx = 0; bx < 3; ++bx) {
          const int hh = min(gy[by + 1], Sm) - max(gy[by], 0), ww = min(gx[bx + 1], Sm) - max(gx[bx], 0);
          if (hh > 0 && ww > 0) area += hh * ww;
        }
      o[9] = area;
      o[10] = 1;
      o[11] = 0;
      ++cnt;
    }
  }
  for (int r = cnt; r < ROI_MAX; ++r)
    for (int k = 0; k < ROI_W; ++k) out[r * ROI_W + k] = 0;
  roi_count[b] = cnt;
}

// one thread = one score-map pixel; loops the image's positive RoIs
__global__ __launch_bounds__(256) void psroi_loss_kernel(const float* score, const uint8_t* true_masks, int G,
                                                         const int* rois, const int* roi_count, int B, int Sm,
                                                         float mask_scale, bf16* dscore, float* partial) {
  __shared__ int s_roi[ROI_MAX * ROI_W];
  __shared__ float s_red[4][ROI_MAX];
  const int b = blockIdx.y;
  const int cnt = roi_count[b];
  for (int i = threadIdx.x; i < ROI_MAX * ROI_W; i += 256) s_roi[i] = rois[(size_t)b * ROI_MAX * ROI_W + i];
  __syncthreads();
  const int npx = Sm * Sm;
  const int i = blockIdx.x * 256 + threadIdx.x;
  float g[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) g[k] = 0.f;
  float lsum[ROI_MAX];
#pragma unroll
  for (int r = 0; r < ROI_MAX; ++r) lsum[r] = 0.f;
  if (i < npx && cnt > 0) {
    const int y = i / Sm, x = i - y * Sm;
    const float* sc = score + ((size_t)b * npx + i) * 9;
    const int S = 2 * Sm;
    const float coef0 = mask_scale / ((float)B * (float)cnt);
#pragma unroll
    for (int r = 0; r < ROI_MAX; ++r) {
      if (r < cnt) {
        const int* o = s_roi + r * ROI_W;
        if (y >= o[0] && y < o[3] && x >= o[4] && x < o[7]) {
          const int by = (y >= o[1]) + (y >= o[2]);
          const int bx = (x >= o[5]) + (x >= o[6]);
          const int ch = by * 3 + bx;
          const float logit = sc[ch];
          // GT mask down-sampled by exact 2x legacy bilinear == [::2, ::2] (:773-775)
          const float gt = true_masks[(((size_t)b * G + o[8]) * S + 2 * y) * S + 2 * x] ? 1.f : 0.f;
          const float inv_area = 1.f / (float)o[9];
          lsum[r] = sigmoid_ce(gt, logit) * inv_area;
          const float dv = (sigmoidf_(logit) - gt) * inv_area * coef0;
#pragma unroll
          for (int k = 0; k < 9; ++k) g[k] += (k == ch) ? dv : 0.f;
        }
      }
    }
  }
  if (i < npx) {
    float v[32];
#pragma unroll
    for (int k = 0; k < 32; ++k) v[k] = k < 9 ? g[k] : 0.f;
    uint4* o = reinterpret_cast<uint4*>(dscore + ((size_t)b * npx + i) * DL_LD);
    o[0] = pack8(v);
    o[1] = pack8(v + 8);
    o[2] = pack8(v + 16);
    o[3] = pack8(v + 24);
  }
#pragma unroll
  for (int r = 0; r < ROI_MAX; ++r) {
    const float s = wave_sum(lsum[r]);
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6][r] = s;
  }
  __syncthreads();
  if (threadIdx.x < ROI_MAX) {
    const int r = threadIdx.x;
    partial[((size_t)b * gridDim.x + blockIdx.x) * ROI_MAX + r] = s_red[0][r] + s_red[1][r] + s_red[2][r] + s_red[3][r];
  }
}

// per image: mask_scale * mean_r(sum_r);  an RoI whose area is 0 yields 0/0 = NaN like the
// reference.  One block per image: 16 RoIs x 16 partial-row lanes.
__global__ __launch_bounds__(256) void psroi_loss_image_kernel(const float* partial, const int* rois,
                                                               const int* roi_count, int nblk, float mask_scale,
                                                               float* img_loss) {
  __shared__ double sh[16][ROI_MAX];
  const int b = blockIdx.x, r = threadIdx.x % ROI_MAX, l = threadIdx.x / ROI_MAX;
  double acc = 0.0;
  for (int k = l; k < nblk; k += 16) acc += (double)partial[((size_t)b * nblk + k) * ROI_MAX + r];
  sh[l][r] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    const int cnt = roi_count[b];
    double img = 0.0;
    for (int q = 0; q < cnt; ++q) {
      double s = 0.0;
      for (int k = 0; k < 16; ++k) s += sh[k][q];
      if (rois[((size_t)b * ROI_MAX + q) * ROI_W + 9] == 0) s = NAN;
      img += s;
    }
    img_loss[b] = cnt > 0 ? (float)((double)mask_scale * img / (double)cnt) : 0.f;
  }
}
__global__ void psroi_loss_final_kernel(const float* img_loss, int B, float* loss) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  double tot = 0.0;
  for (int b = 0; b < B; ++b) tot += (double)img_loss[b];
  loss[0] = (float)(tot / (double)B);
}

}  // namespace

extern "C" size_t disyolo_yolo_loss_workspace(int B, int S, int num_class) {
  if (B <= 0 || S <= 0 || S % 32) return 0;
  const int g1 = S / 32;
  size_t blocks = 0;
  for (int m = 1; m <= 4; m *= 2) blocks += (size_t)ceil_div((size_t)(m * g1) * (m * g1) * 3, 256) * B;
  return blocks * 5 * sizeof(float);
}

extern "C" int disyolo_yolo_loss(const float* const logits[3], const float* const labels[3], const float* true_boxes,
                                 int max_boxes, int B, int S, int num_class, const float* anchors_host,
                                 float ignore_thresh, const float scales[4], void* const dlogits[3], float* losses,
                                 void* workspace, size_t workspace_bytes, void* stream) {
  DY_REQUIRE(logits && labels && true_boxes && anchors_host && scales && dlogits && losses, "yolo_loss: null pointer");
  DY_REQUIRE(B > 0 && S > 0 && S % 32 == 0 && num_class > 0 && 3 * (5 + num_class) <= DL_LD && num_class <= 11,
             "yolo_loss: bad sizes");
  DY_REQUIRE(max_boxes > 0 && max_boxes <= 64, "yolo_loss: max_boxes must be in 1..64");
  if (!workspace || workspace_bytes < disyolo_yolo_loss_workspace(B, S, num_class)) {
    disyolo_set_error("yolo_loss: workspace too small");
    return DISYOLO_E_WORKSPACE;
  }
  {
    std::array<float, 18> anc;
    for (int i = 0; i < 18; ++i) anc[i] = anchors_host[i];
    std::array<float, 4> sc4 = {scales[0], scales[1], scales[2], scales[3]};
    std::array<const float*, 3> lg = {logits[0], logits[1], logits[2]}, lb = {labels[0], labels[1], labels[2]};
    std::array<void*, 3> dl = {dlogits[0], dlogits[1], dlogits[2]};
    DY_RECORD_OR_RUN([=](void* s) {
      return disyolo_yolo_loss(lg.data(), lb.data(), true_boxes, max_boxes, B, S, num_class, anc.data(), ignore_thresh,
                               sc4.data(), dl.data(), losses, workspace, workspace_bytes, s);
    });
  }
  hipStream_t st = (hipStream_t)stream;
  const int g1 = S / 32;
  const int gs[3] = {4 * g1, 2 * g1, g1};
  float* part = (float*)workspace;
  int nblk[3];
  YoloLoss3 P;
  for (int s = 0; s < 3; ++s) {
    DY_REQUIRE(logits[s] && labels[s] && dlogits[s], "yolo_loss: null tensor for scale %d", s);
    YoloLossParams& p = P.p[s];
    p.logits = logits[s];
    p.labels = labels[s];
    p.true_boxes = true_boxes;
    p.dlogits = (bf16*)dlogits[s];
    p.partial = part;
    p.B = B; p.g = gs[s]; p.C = num_class; p.G = max_boxes; p.S = S;
    for (int a = 0; a < 3; ++a) {
      p.aw[a] = anchors_host[(3 * s + a) * 2 + 0];
      p.ah[a] = anchors_host[(3 * s + a) * 2 + 1];
    }
    p.ignore_thresh = ignore_thresh;
    p.obj_scale = scales[0]; p.noobj_scale = scales[1]; p.class_scale = scales[2]; p.coord_scale = scales[3];
    p.inv_B = 1.f / (float)B;
    P.gx[s] = ceil_div(gs[s] * gs[s] * 3, 256);
    nblk[s] = P.gx[s] * B;
    part += (size_t)nblk[s] * 5;
  }
  hipLaunchKernelGGL(yolo_loss_kernel, dim3(P.gx[0] + P.gx[1] + P.gx[2], B), dim3(256), 0, st, P);
  DY_CHECK_LAUNCH();
  hipLaunchKernelGGL(yolo_loss_final_kernel, dim3(1), dim3(320), 0, st, (const float*)workspace,
                     nblk[0] + nblk[1] + nblk[2], 1.f / (float)B, losses);
  DY_CHECK_LAUNCH();
  return DISYOLO_OK;
}

extern "C" int disyolo_shuffle_perm(int32_t* perm_det, int n_det, int32_t* perm_gt, int n_gt, int B, uint32_t seed,
                                    const int64_t* step_counter, void* stream) {
  DY_REQUIRE(perm_det && perm_gt && B > 0 && n_det > 0 && n_det <= 64 && n_gt > 0 && n_gt <= 64, "shuffle_perm: bad args");
  DY_RECORD_OR_RUN([=](void* s) { return disyolo_shuffle_perm(perm_det, n_det, perm_gt, n_gt, B, seed, step_counter, s); });
  hipLaunchKernelGGL(shuffle_perm_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, perm_det, n_det, perm_gt, n_gt, seed,
                     step_counter);
  DY_CHECK_LAUNCH();
  return DISYOLO_OK;
}

extern "C" int disyolo_mask_rois(const float* detections, int max_det, const float* true_boxes, int G,
                                 const int32_t* perm_det, const int32_t* perm_gt, int B, int map_size, int n_det,
                                 int n_gt, float iou_thresh, int32_t* rois, int32_t* roi_count, void* stream) {
  DY_REQUIRE(detections && true_boxes && rois && roi_count, "mask_rois: null pointer");
  DY_REQUIRE(B > 0 && max_det > 0 && max_det <= 64 && G > 0 && G <= 64 && map_size > 0, "mask_rois: bad sizes");
  DY_REQUIRE(n_det >= 0 && n_gt >= 0 && n_det + n_gt <= ROI_MAX, "mask_rois: n_det + n_gt > %d", ROI_MAX);
  DY_RECORD_OR_RUN([=](void* s) {
    return disyolo_mask_rois(detections, max_det, true_boxes, G, perm_det, perm_gt, B, map_size, n_det, n_gt, iou_thresh,
                             rois, roi_count, s);
  });
  hipLaunchKernelGGL(mask_rois_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, detections, max_det,
                     true_boxes, G, perm_det, perm_gt, B, map_size, n_det, n_gt, iou_thresh, rois, roi_count);
  DY_CHECK_LAUNCH();
  return DISYOLO_OK;
}

extern "C" size_t disyolo_psroi_loss_workspace(int B, int map_size) {
  if (B <= 0 || map_size <= 0) return 0;
  return ((size_t)B * ceil_div((size_t)map_size * map_size, 256) * ROI_MAX + B) * sizeof(float);
}

extern "C" int disyolo_psroi_loss(const float* score, const uint8_t* true_masks, int G, const int32_t* rois,
                                  const int32_t* roi_count, int B, int map_size, int k, float mask_scale, void* dscore,
                                  float* loss, void* workspace, size_t workspace_bytes, void* stream) {
  DY_REQUIRE(score && true_masks && rois && roi_count && dscore && loss, "psroi_loss: null pointer");
  DY_REQUIRE(B > 0 && map_size > 0 && G > 0, "psroi_loss: bad sizes");
  DY_REQUIRE(k == 3, "psroi_loss: only k = 3 (the reference's active branch, yolo/yolo3_net_pos.py:810-813)");
  if (!workspace || workspace_bytes < disyolo_psroi_loss_workspace(B, map_size)) {
    disyolo_set_error("psroi_loss: workspace too small");
    return DISYOLO_E_WORKSPACE;
  }
  DY_RECORD_OR_RUN([=](void* s) {
    return disyolo_psroi_loss(score, true_masks, G, rois, roi_count, B, map_size, k, mask_scale, dscore, loss, workspace,
                              workspace_bytes, s);
  });
  hipStream_t st = (hipStream_t)stream;
  const int nblk = ceil_div((size_t)map_size * map_size, 256);
  hipLaunchKernelGGL(psroi_loss_kernel, dim3(nblk, B), dim3(256), 0, st, score, true_masks, G, rois, roi_count, B,
                     map_size, mask_scale, (bf16*)dscore, (float*)workspace);
  DY_CHECK_LAUNCH();
  float* img_loss = (float*)workspace + (size_t)B * nblk * ROI_MAX;
  hipLaunchKernelGGL(psroi_loss_image_kernel, dim3(B), dim3(256), 0, st, (const float*)workspace, rois, roi_count,
                     nblk, mask_scale, img_loss);
  DY_CHECK_LAUNCH();
  hipLaunchKernelGGL(psroi_loss_final_kernel, dim3(1), dim3(64), 0, st, (const float*)img_loss, B, loss);
  DY_CHECK_LAUNCH();
  return DISYOLO_OK;
}
