// Detection decode + filter and the inference-time position-sensitive mask assembly.
//
// Replaces interpret_output / filter_detections / clip_boxes_graph / val_test of
// yolo/yolo3_net_pos.py:465-628, 862-952 (the reference unrolls these per image in Python
// and runs tf.image.non_max_suppression per class through tf.map_fn).
#include <array>
#include "common.h"
#include "runtime.h"

namespace {

struct ScaleInfo {
  const float* logits;  // [B,g,g,3,5+C]
  int g;
  int cand0;  // first candidate index of this scale inside an image
  float aw[3], ah[3];
};
struct DecodeParams {
  ScaleInfo sc[3];
  int B, S, C, NC;
  const float* window;  // [B,4] y1,x1,y2,x2
  float4* boxes;        // [B][NC] (y1,x1,y2,x2)
  float* scores;        // [B][NC]
  int* classes;         // [B][NC]
};

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

// candidate order inside an image: scale (S/8, S/16, S/32 grids), then y, x, anchor
// (yolo/yolo3_net_pos.py:527-542)
__global__ __launch_bounds__(256) void decode_score_kernel(DecodeParams p) {
  const int b = blockIdx.y;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= p.NC) return;
  int s = 0;
  if (idx >= p.sc[1].cand0) s = 1;
  if (idx >= p.sc[2].cand0) s = 2;
  const ScaleInfo& si = p.sc[s];
  const int local = idx - si.cand0;
  const int a = local % 3;
  const int cell = local / 3;
  const int x = cell % si.g, y = cell / si.g;
  const int D = 5 + p.C;
  const float* t = si.logits + (((size_t)b * si.g * si.g + cell) * 3 + a) * D;
  const float conf = sigmoidf_(t[4]);
  // softmax max-probability and argmax (first maximum wins, tf.argmax)
  float mx = t[5];
  int arg = 0;
  for (int c = 1; c < p.C; ++c)
    if (t[5 + c] > mx) {
      mx = t[5 + c];
      arg = c;
    }
  float den = 0.f;
  for (int c = 0; c < p.C; ++c) den += expf(t[5 + c] - mx);
  const float score = conf * (1.f / den);
  const float g = (float)si.g, net = (float)p.S;
  const float xc = ((float)x + sigmoidf_(t[0])) / g;
  const float yc = ((float)y + sigmoidf_(t[1])) / g;
  const float w = expf(t[2]) * si.aw[a] / net;
  const float h = expf(t[3]) * si.ah[a] / net;
  const float* win = p.window + b * 4;
  float4 bx;
  bx.x = fmaxf(fminf(yc - h / 2.f, win[2]), win[0]);
  bx.y = fmaxf(fminf(xc - w / 2.f, win[3]), win[1]);
  bx.z = fmaxf(fminf(yc + h / 2.f, win[2]), win[0]);
  bx.w = fmaxf(fminf(xc + w / 2.f, win[3]), win[1]);
  const size_t o = (size_t)b * p.NC + idx;
  p.boxes[o] = bx;
  p.scores[o] = score;
  p.classes[o] = arg;
}

// IoU as tf.image.non_max_suppression computes it (corner order normalised, empty box -> 0)
__device__ __forceinline__ float nms_iou(const float4& a, const float4& b) {
  const float ya1 = fminf(a.x, a.z), xa1 = fminf(a.y, a.w), ya2 = fmaxf(a.x, a.z), xa2 = fmaxf(a.y, a.w);
  const float yb1 = fminf(b.x, b.z), xb1 = fminf(b.y, b.w), yb2 = fmaxf(b.x, b.z), xb2 = fmaxf(b.y, b.w);
  const float aa = (ya2 - ya1) * (xa2 - xa1), ab = (yb2 - yb1) * (xb2 - xb1);
  if (aa <= 0.f || ab <= 0.f) return 0.f;
  const float iy1 = fmaxf(ya1, yb1), ix1 = fmaxf(xa1, xb1), iy2 = fminf(ya2, yb2), ix2 = fminf(xa2, xb2);
  const float inter = fmaxf(iy2 - iy1, 0.f) * fmaxf(ix2 - ix1, 0.f);
  return inter / (aa + ab - inter);
}

// Detection filter.  Stage 1: one block per (image, class): ordered compaction of the
// candidates with score > thr and arg-max class == c, then greedy NMS (IoU > nms_thr
// suppresses, at most max_det kept) as a repeated block-wide arg-max over the live list --
// exact for any number of candidates, never a truncated sort.  Stage 2: one block per image
// merges the per-class survivors: top max_det by score (ties: lower candidate index), zero
// padded.
//
// The live list sits in REGISTERS while it fits (<= NMS_K candidates per thread, 8,192 per
// block: score and box of list position q are held by thread q % NMS_T).  One
// round of the greedy loop is then ONE pass over a thread's slots -- suppress against the
// last winner and find the thread's best survivor in the same sweep -- and ONE barrier: the
// wave winners (score, position, box) go to LDS, double-buffered by round parity, and
// every thread picks the block winner from the 16 entries.  An untrained network passes
// thousands of candidates per class (random logits at 576^2: ~6,800 of 20,412); with the list
// in global memory a round cost 16 us (two dependent sweeps + two barriers), 30 rounds = 0.5 ms
// on the chain that holds up the mask subnet's backward pass; in registers a round is ~1 us.
// Longer lists (one class taking > 8,192 candidates) keep the global-memory form.
constexpr int NMS_T = 1024;
constexpr int NMS_K = 8;
constexpr int NMS_W = NMS_T / 64;
constexpr int NMS_CB = 8;             // compaction rounds per barrier
constexpr int MAX_KEEP = 512;  // >= num_class * max_det

__device__ __forceinline__ bool nms_better(float v2, int p2, float v, int p) { return v2 > v || (v2 == v && p2 < p); }

__device__ __forceinline__ int greedy_nms_global(int n, const int* list, const float4* boxes, float* live, float nms_thr,
                                                 int max_det, int* kept_idx, float* kept_sc, float* s_ws, int* s_wp) {
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  int nk = 0;
  for (int it = 0; it < max_det; ++it) {
    float bs = -1.f;
    int bp = 0x7fffffff;
    for (int q = tid; q < n; q += NMS_T) {
      const float v = live[q];
      if (v > bs) {  // ascending scan: lowest position wins among equal scores
        bs = v;
        bp = q;
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float v2 = __shfl_xor(bs, o, 64);
      const int p2 = __shfl_xor(bp, o, 64);
      if (nms_better(v2, p2, bs, bp)) {
        bs = v2;
        bp = p2;
      }
    }
    if (lane == 0) {
      s_ws[wv] = bs;
      s_wp[wv] = bp;
    }
    __syncthreads();
    float wsc = s_ws[0];
    int wq = s_wp[0];
#pragma unroll
    for (int w = 1; w < NMS_W; ++w) {
      const float v2 = s_ws[w];
      const int p2 = s_wp[w];
      if (nms_better(v2, p2, wsc, wq)) {
        wsc = v2;
        wq = p2;
      }
    }
    if (!(wsc > 0.f)) break;  // nothing alive (live scores are > thr >= 0); uniform exit
    const float4 wb = boxes[list[wq]];
    if (tid == 0) {
      kept_idx[nk] = list[wq];
      kept_sc[nk] = wsc;
    }
    ++nk;
    for (int q = tid; q < n; q += NMS_T) {
      if (live[q] >= 0.f) {
        const float4 bx = boxes[list[q]];
        if (q == wq || nms_iou(bx, wb) > nms_thr) live[q] = -1.f;
      }
    }
    __syncthreads();
  }
  return nk;
}

__global__ __launch_bounds__(NMS_T) void nms_class_kernel(const float4* boxes, const float* scores, const int* classes,
                                                          int NC, int C, float thr, float nms_thr, int max_det,
                                                          int* list_g, float* live_g, int* kept_idx, float* kept_sc,
                                                          int* kept_n) {
  const int b = blockIdx.x / C, c = blockIdx.x - b * C, tid = threadIdx.x;
  const int lane = tid & 63, wv = tid >> 6;
  boxes += (size_t)b * NC;
  scores += (size_t)b * NC;
  classes += (size_t)b * NC;
  int* list = list_g + (size_t)blockIdx.x * NC;
  float* live_glob = live_g + (size_t)blockIdx.x * NC;
  kept_idx += (size_t)blockIdx.x * max_det;
  kept_sc += (size_t)blockIdx.x * max_det;
  __shared__ int s_wcnt[2][NMS_CB][NMS_W];
  __shared__ float s_ws[2][NMS_W];
  __shared__ int s_wp[2][NMS_W];
  __shared__ float4 s_wb[2][NMS_W];
  __shared__ int s_kidx[64];
  __shared__ float s_ksc[64];
  // ---- ordered compaction: wave ballots; eight rounds of 1,024 candidates per barrier, their 16 loads in flight together
  //      (a round per barrier is a load round trip + a store drain each: 60 us for the 20 rounds of a 576^2 image even when
  //      nothing passes the threshold)
  int total = 0;
  for (int base = 0, bt = 0; base < NC; base += NMS_T * NMS_CB, ++bt) {
    bool f[NMS_CB];
    unsigned long long mask[NMS_CB];
#pragma unroll
    for (int j = 0; j < NMS_CB; ++j) {
      const int i = base + j * NMS_T + tid;
      const int ii = i < NC ? i : 0;
      const float sv = scores[ii];
      const int cv = classes[ii];
      f[j] = (i < NC) && (sv > thr) && (cv == c);
    }
#pragma unroll
    for (int j = 0; j < NMS_CB; ++j) {
      mask[j] = __ballot(f[j]);
      if (lane == 0) s_wcnt[bt & 1][j][wv] = __popcll(mask[j]);
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < NMS_CB; ++j) {
      int off = total;
#pragma unroll
      for (int w = 0; w < NMS_W; ++w) {
        const int cw = s_wcnt[bt & 1][j][w];
        if (w < wv) off += cw;
        total += cw;
      }
      if (f[j]) list[off + __popcll(mask[j] & ((1ull << lane) - 1ull))] = base + j * NMS_T + tid;
    }
  }
  __syncthreads();  // list[] (global) written by this block, read below by all its threads
  const int n = total;
  int nk = 0;
  if (n <= NMS_K * NMS_T) {
    float sc[NMS_K];
    float4 bx[NMS_K];
    float bs = -1.f;                 // this thread's best survivor: score, list position, box
    int bp = 0x7fffffff;
    float4 bb = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int k = 0; k < NMS_K; ++k) {
      const int q = tid + k * NMS_T;
      sc[k] = -1.f;
      bx[k] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (q < n) {
        const int i = list[q];
        sc[k] = scores[i];
        bx[k] = boxes[i];
      }
      const bool take = sc[k] > bs;  // ascending positions: the lowest wins among equal scores
      bs = take ? sc[k] : bs;
      bp = take ? q : bp;
      bb.x = take ? bx[k].x : bb.x;
      bb.y = take ? bx[k].y : bb.y;
      bb.z = take ? bx[k].z : bb.z;
      bb.w = take ? bx[k].w : bb.w;
    }
    for (int it = 0; it < max_det; ++it) {
      const int par = it & 1;
      float ws = bs;
      int wp = bp;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        const float v2 = __shfl_xor(ws, o, 64);
        const int p2 = __shfl_xor(wp, o, 64);
        if (nms_better(v2, p2, ws, wp)) {
          ws = v2;
          wp = p2;
        }
      }
      if (wp == bp) {  // the wave winner's owner (a wave with nothing alive: every lane, the same values)
        s_ws[par][wv] = ws;
        s_wp[par][wv] = wp;
        s_wb[par][wv] = bb;
      }
      __syncthreads();
      float wsc = s_ws[par][0];
      int wq = s_wp[par][0], ww = 0;
#pragma unroll
      for (int w = 1; w < NMS_W; ++w) {
        const float v2 = s_ws[par][w];
        const int p2 = s_wp[par][w];
        if (nms_better(v2, p2, wsc, wq)) {
          wsc = v2;
          wq = p2;
          ww = w;
        }
      }
      if (!(wsc > 0.f)) break;  // nothing alive (live scores are > thr >= 0); uniform exit
      const float4 wb = s_wb[par][ww];
      if (tid == 0) {
        s_kidx[nk] = wq;          // list position; the candidate index is looked up once, after the loop
        s_ksc[nk] = wsc;
      }
      ++nk;
      // suppress against the winner and find this thread's best survivor, one sweep.  Branch-free on purpose: the nested
      // form "if (alive) { if (winner || iou > thr) kill; else if (better) take; }" came out of hipcc 7.2 never killing
      // on the IoU test (caught by the many-candidates known-answer test)
      bs = -1.f;
      bp = 0x7fffffff;
#pragma unroll
      for (int k = 0; k < NMS_K; ++k) {
        const int q = tid + k * NMS_T;
        const float iou = nms_iou(bx[k], wb);
        const bool kill = (q == wq) | (iou > nms_thr);
        sc[k] = kill ? -1.f : sc[k];            // (dead slots stay at -1)
        const bool take = sc[k] > bs;
        bs = take ? sc[k] : bs;
        bp = take ? q : bp;
        bb.x = take ? bx[k].x : bb.x;
        bb.y = take ? bx[k].y : bb.y;
        bb.z = take ? bx[k].z : bb.z;
        bb.w = take ? bx[k].w : bb.w;
      }
    }
    __syncthreads();
    if (tid < nk) {
      kept_idx[tid] = list[s_kidx[tid]];
      kept_sc[tid] = s_ksc[tid];
    }
  } else {
    for (int q = tid; q < n; q += NMS_T) live_glob[q] = scores[list[q]];
    __syncthreads();
    nk = greedy_nms_global(n, list, boxes, live_glob, nms_thr, max_det, kept_idx, kept_sc, s_ws[0], s_wp[0]);
  }
  if (tid == 0) kept_n[blockIdx.x] = nk;
}

__global__ __launch_bounds__(64) void nms_merge_kernel(const float4* boxes, const int* classes, int NC, int C,
                                                       int max_det, const int* kept_idx, const float* kept_sc,
                                                       const int* kept_n, float* det, int* det_count) {
  const int b = blockIdx.x, tid = threadIdx.x;
  boxes += (size_t)b * NC;
  classes += (size_t)b * NC;
  __shared__ int s_idx[MAX_KEEP];
  __shared__ float s_sc[MAX_KEEP];
  __shared__ int s_off[17];
  if (tid == 0) {
    int o = 0;
    for (int c = 0; c < C; ++c) {
      s_off[c] = o;
      o += kept_n[b * C + c];
    }
    s_off[C] = o;
  }
  __syncthreads();
  const int nk = s_off[C];
  for (int c = 0; c < C; ++c) {
    const int cn = s_off[c + 1] - s_off[c];
    for (int k = tid; k < cn; k += 64) {
      s_idx[s_off[c] + k] = kept_idx[((size_t)b * C + c) * max_det + k];
      s_sc[s_off[c] + k] = kept_sc[((size_t)b * C + c) * max_det + k];
    }
  }
  for (int r = tid; r < max_det * 6; r += 64) det[(size_t)b * max_det * 6 + r] = 0.f;
  __syncthreads();
  for (int k = tid; k < nk; k += 64) {
    const float sc = s_sc[k];
    const int ix = s_idx[k];
    int rank = 0;
    for (int j = 0; j < nk; ++j) {
      const float sj = s_sc[j];
      if (sj > sc || (sj == sc && s_idx[j] < ix)) ++rank;
    }
    if (rank < max_det) {
      float* o = det + ((size_t)b * max_det + rank) * 6;
      const float4 bx = boxes[ix];
      o[0] = bx.x;
      o[1] = bx.y;
      o[2] = bx.z;
      o[3] = bx.w;
      o[4] = (float)classes[ix];
      o[5] = sc;
    }
  }
  if (tid == 0) det_count[b] = nk < max_det ? nk : max_det;
}

// bin edges of assemble_kmask_from_box (yolo/yolo3_net_pos.py:804-813) for k = 3:
// [int(lo), rint(lo + sub), rint(lo + 2*sub), int(hi)], sub = (hi - lo)/3, f32 math,
// rintf = round-half-to-even like tf.round.
__device__ __forceinline__ void bin_edges3(float lo, float hi, int e[4]) {
  const float sub = (hi - lo) / 3.f;
  e[0] = (int)lo;
  e[1] = (int)rintf(lo + sub);
  e[2] = (int)rintf(lo + 2.f * sub);
  e[3] = (int)hi;
}

// masks[b,r,y,x] = sigmoid(score[b,y,x,bin(y,x)]) inside the box, 0.5 outside (:925-928)
__global__ __launch_bounds__(256) void psroi_assemble_kernel(const float* score, const float* det, int B, int max_det,
                                                             int Sm, float* masks, int* keep) {
  const int r = blockIdx.y, b = blockIdx.z;
  const float* d = det + ((size_t)b * max_det + r) * 6;
  const float sz = (float)Sm;
  const float y1 = rintf(d[0] * sz), x1 = rintf(d[1] * sz), y2 = rintf(d[2] * sz), x2 = rintf(d[3] * sz);
  const bool kp = (y2 - y1) > 0.f && (x2 - x1) > 0.f;
  if (blockIdx.x == 0 && threadIdx.x == 0) keep[b * max_det + r] = kp ? 1 : 0;
  int gy[4], gx[4];
  bin_edges3(y1, y2, gy);
  bin_edges3(x1, x2, gx);
  const int npx = Sm * Sm;
  float* out = masks + ((size_t)b * max_det + r) * npx;
  const float* sc = score + (size_t)b * npx * 9;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < npx; i += gridDim.x * blockDim.x) {
    float v = 0.f;
    if (kp) {
      const int y = i / Sm, x = i - y * Sm;
      float logit = 0.f;
      if (y >= gy[0] && y < gy[3] && x >= gx[0] && x < gx[3]) {
        const int by = (y >= gy[1]) + (y >= gy[2]);
        const int bx = (x >= gx[1]) + (x >= gx[2]);
        logit = sc[(size_t)i * 9 + by * 3 + bx];
      }
      v = 1.f / (1.f + expf(-logit));
    }
    out[i] = v;
  }
}

// ---- evaluate(): mask post-processing (calculate_test_map.py:237-262) ----------------------
// One thread per image pixel walks the image's detections in order: inside a detection's
// destination rectangle it samples the cropped mask bilinearly with cv2.resize(INTER_LINEAR)
// semantics (pixel centres aligned, border taps clamped with weight 0, horizontal then vertical
// pass, every product and sum rounded to f32 separately -- no FMA, so the > 0.5 decision is the
// same as the host restatement's), writes the per-detection bit and keeps the class of the last
// covering detection for the merged class map.
__global__ __launch_bounds__(256) void mask_paste_kernel(const float* masks, int n, int size, const int* rects,
                                                         const int* classids, int H, int W, unsigned char* full,
                                                         unsigned char* merged) {
  const int64_t total = (int64_t)H * W;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int y = (int)(i / W), x = (int)(i - (int64_t)y * W);
    unsigned char mcls = 0;
    for (int k = 0; k < n; ++k) {
      const int* r = rects + k * 8;
      const int cy1 = r[0], cx1 = r[1], cy2 = r[2], cx2 = r[3], y1 = r[4], x1 = r[5], y2 = r[6], x2 = r[7];
      const int sh = cy2 - cy1, sw = cx2 - cx1, dh = y2 - y1, dw = x2 - x1;
      unsigned char bit = 0;
      if (sh > 0 && sw > 0 && dh > 0 && dw > 0 && y >= y1 && y < y2 && x >= x1 && x < x2) {
        float fx = (float)(((double)(x - x1) + 0.5) * ((double)sw / (double)dw) - 0.5);
        float fy = (float)(((double)(y - y1) + 0.5) * ((double)sh / (double)dh) - 0.5);
        int sx = (int)floorf(fx), sy = (int)floorf(fy);
        float ax = __fsub_rn(fx, (float)sx), ay = __fsub_rn(fy, (float)sy);
        if (sx < 0) { ax = 0.f; sx = 0; }
        if (sx >= sw - 1) { ax = 0.f; sx = sw - 1; }
        if (sy < 0) { ay = 0.f; sy = 0; }
        if (sy >= sh - 1) { ay = 0.f; sy = sh - 1; }
        const int sx1 = min(sx + 1, sw - 1), sy1 = min(sy + 1, sh - 1);
        const float* m = masks + (size_t)k * size * size;
        const float* row0 = m + (size_t)(cy1 + sy) * size + cx1;
        const float* row1 = m + (size_t)(cy1 + sy1) * size + cx1;
        const float bx = __fsub_rn(1.f, ax), by = __fsub_rn(1.f, ay);
        const float h0 = __fadd_rn(__fmul_rn(row0[sx], bx), __fmul_rn(row0[sx1], ax));
        const float h1 = __fadd_rn(__fmul_rn(row1[sx], bx), __fmul_rn(row1[sx1], ax));
        const float v = __fadd_rn(__fmul_rn(h0, by), __fmul_rn(h1, ay));
        bit = v > 0.5f ? 1 : 0;
      }
      if (full) full[(size_t)k * total + i] = bit;
      if (bit) mcls = (unsigned char)(classids[k] + 1);
    }
    merged[i] = mcls;
  }
}


// image_read (calculate_test_map.py:149-176, utils/val_data.py:36-63): aspect-preserving bilinear
// resize of an RGB uint8 image (as float32, like cv2.resize(image.astype(float32), INTER_LINEAR))
// centred in a size x size letter box padded with 127, then / 255.  Same tap arithmetic as the mask
// paste above (aligned pixel centres, clamped border taps with weight 0, horizontal then vertical
// pass, float32 without FMA).  One thread per output pixel; the three channels together.
__global__ __launch_bounds__(256) void letterbox_kernel(const unsigned char* rgb, int H, int W, float* out, int size,
                                                        int new_h, int new_w, int top, int left) {
  const int64_t total = (int64_t)size * size;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int y = (int)(i / size), x = (int)(i - (int64_t)y * size);
    float v[3] = {127.f, 127.f, 127.f};
    if (y >= top && y < top + new_h && x >= left && x < left + new_w) {
      float fx = (float)(((double)(x - left) + 0.5) * ((double)W / (double)new_w) - 0.5);
      float fy = (float)(((double)(y - top) + 0.5) * ((double)H / (double)new_h) - 0.5);
      int sx = (int)floorf(fx), sy = (int)floorf(fy);
      float ax = __fsub_rn(fx, (float)sx), ay = __fsub_rn(fy, (float)sy);
      if (sx < 0) { ax = 0.f; sx = 0; }
      if (sx >= W - 1) { ax = 0.f; sx = W - 1; }
      if (sy < 0) { ay = 0.f; sy = 0; }
      if (sy >= H - 1) { ay = 0.f; sy = H - 1; }
      const int sx1 = min(sx + 1, W - 1), sy1 = min(sy + 1, H - 1);
      const unsigned char* r0 = rgb + ((size_t)sy * W) * 3;
      const unsigned char* r1 = rgb + ((size_t)sy1 * W) * 3;
      const float bx = __fsub_rn(1.f, ax), by = __fsub_rn(1.f, ay);
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float h0 = __fadd_rn(__fmul_rn((float)r0[sx * 3 + c], bx), __fmul_rn((float)r0[sx1 * 3 + c], ax));
        const float h1 = __fadd_rn(__fmul_rn((float)r1[sx * 3 + c], bx), __fmul_rn((float)r1[sx1 * 3 + c], ax));
        v[c] = __fadd_rn(__fmul_rn(h0, by), __fmul_rn(h1, ay));
      }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) out[i * 3 + c] = (float)__ddiv_rn((double)v[c], 255.0);   // new_image / 255.0 in float64 (a plain "/" is narrowed to an approximate f32 division)
  }
}

// 4x4 pixel confusion counts of two uint8 class maps (calculate_test_map.py:303-331): per-block LDS
// histogram, integer atomics into the caller's int64[16] accumulator (exact, order-independent)
__global__ __launch_bounds__(256) void confusion16_kernel(const unsigned char* t, const unsigned char* p, int64_t n,
                                                          unsigned long long* conf) {
  __shared__ unsigned int h[16];
  if (threadIdx.x < 16) h[threadIdx.x] = 0;
  __syncthreads();
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const unsigned a = t[i], b = p[i];
    if (a < 4 && b < 4) atomicAdd(&h[a * 4 + b], 1u);
  }
  __syncthreads();
  if (threadIdx.x < 16 && h[threadIdx.x]) atomicAdd(&conf[threadIdx.x], (unsigned long long)h[threadIdx.x]);
}

}  // namespace

extern "C" size_t disyolo_detect_workspace(int B, int S, int num_class) {
  if (B <= 0 || S <= 0 || S % 32) return 0;
  const int g1 = S / 32;
  const size_t NC = 3 * (size_t)(16 * g1 * g1 + 4 * g1 * g1 + g1 * g1);
  // boxes (16 B) + scores + classes, per-class list + live, per-class kept (idx, score, n)
  return (size_t)B * NC * (16 + 4 + 4) + (size_t)B * num_class * NC * 8 + (size_t)B * num_class * (64 * 8 + 4) + 256;
}

extern "C" int disyolo_detect(const float* logits3, const float* logits2, const float* logits1, int B, int S,
                              int num_class, const float* anchors_host, const float* clip_window, float obj_thresh,
                              float nms_thresh, int max_det, float* detections, int32_t* det_count, void* workspace,
                              size_t workspace_bytes, void* stream) {
  DY_REQUIRE(logits3 && logits2 && logits1 && anchors_host && clip_window && detections && det_count,
             "detect: null pointer");
  DY_REQUIRE(B > 0 && S > 0 && S % 32 == 0 && num_class > 0 && num_class <= 16, "detect: bad sizes");
  DY_REQUIRE(max_det > 0 && max_det <= 64 && num_class * max_det <= MAX_KEEP, "detect: max_det must be <= 64");
  DY_REQUIRE(obj_thresh >= 0.f, "detect: obj_thresh must be >= 0");
  if (!workspace || workspace_bytes < disyolo_detect_workspace(B, S, num_class)) {
    disyolo_set_error("detect: workspace too small");
    return DISYOLO_E_WORKSPACE;
  }
  {
    std::array<float, 18> anc;
    for (int i = 0; i < 18; ++i) anc[i] = anchors_host[i];
    DY_RECORD_OR_RUN([=](void* s) {
      return disyolo_detect(logits3, logits2, logits1, B, S, num_class, anc.data(), clip_window, obj_thresh, nms_thresh,
                            max_det, detections, det_count, workspace, workspace_bytes, s);
    });
  }
  const int g1 = S / 32;
  DecodeParams p;
  const float* lg[3] = {logits3, logits2, logits1};
  const int gs[3] = {4 * g1, 2 * g1, g1};
  int c0 = 0;
  for (int s = 0; s < 3; ++s) {
    p.sc[s].logits = lg[s];
    p.sc[s].g = gs[s];
    p.sc[s].cand0 = c0;
    c0 += gs[s] * gs[s] * 3;
    for (int a = 0; a < 3; ++a) {
      p.sc[s].aw[a] = anchors_host[(3 * s + a) * 2 + 0];
      p.sc[s].ah[a] = anchors_host[(3 * s + a) * 2 + 1];
    }
  }
  p.B = B; p.S = S; p.C = num_class; p.NC = c0;
  p.window = clip_window;
  char* ws = (char*)workspace;
  p.boxes = (float4*)ws;                  ws += (size_t)B * c0 * 16;
  p.scores = (float*)ws;                  ws += (size_t)B * c0 * 4;
  p.classes = (int*)ws;                   ws += (size_t)B * c0 * 4;
  int* list = (int*)ws;                   ws += (size_t)B * num_class * c0 * 4;
  float* live = (float*)ws;               ws += (size_t)B * num_class * c0 * 4;
  int* kept_idx = (int*)ws;               ws += (size_t)B * num_class * 64 * 4;
  float* kept_sc = (float*)ws;            ws += (size_t)B * num_class * 64 * 4;
  int* kept_n = (int*)ws;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(decode_score_kernel, dim3(ceil_div(c0, 256), B), dim3(256), 0, st, p);
  DY_CHECK_LAUNCH();
  hipLaunchKernelGGL(nms_class_kernel, dim3(B * num_class), dim3(NMS_T), 0, st, p.boxes, p.scores, p.classes, c0,
                     num_class, obj_thresh, nms_thresh, max_det, list, live, kept_idx, kept_sc, kept_n);
  DY_CHECK_LAUNCH();
  hipLaunchKernelGGL(nms_merge_kernel, dim3(B), dim3(64), 0, st, p.boxes, p.classes, c0, num_class, max_det, kept_idx,
                     kept_sc, kept_n, detections, det_count);
  DY_CHECK_LAUNCH();
  return DISYOLO_OK;
}

extern "C" int disyolo_psroi_assemble(const float* score, const float* detections, int B, int max_det, int map_size,
                                      int k, float* masks, int32_t* keep, void* stream) {
  DY_REQUIRE(score && detections && masks && keep && B > 0 && max_det > 0 && map_size > 0, "psroi_assemble: bad args");
  DY_REQUIRE(k == 3, "psroi_assemble: only k = 3 (the reference's active branch, yolo/yolo3_net_pos.py:894-897)");
  DY_RECORD_OR_RUN([=](void* s) { return disyolo_psroi_assemble(score, detections, B, max_det, map_size, k, masks, keep, s); });
  const int npx = map_size * map_size;
  int gx = ceil_div(npx, 256 * 4);
  if (gx < 1) gx = 1;
  hipLaunchKernelGGL(psroi_assemble_kernel, dim3(gx, max_det, B), dim3(256), 0, (hipStream_t)stream, score, detections,
                     B, max_det, map_size, masks, keep);
  DY_CHECK_LAUNCH();
  return DISYOLO_OK;
}

extern "C" int disyolo_mask_paste(const float* masks, int n, int size, const int32_t* rects, const int32_t* classids,
                                  int image_h, int image_w, uint8_t* full_masks, uint8_t* merged, void* stream) {
  DY_REQUIRE(merged && image_h > 0 && image_w > 0 && n >= 0 && size > 0, "mask_paste: bad args");
  DY_REQUIRE(n == 0 || (masks && rects && classids), "mask_paste: null pointer");
  DY_RECORD_OR_RUN([=](void* s) { return disyolo_mask_paste(masks, n, size, rects, classids, image_h, image_w, full_masks, merged, s); });
  const int64_t total = (int64_t)image_h * image_w;
  int grid = (int)((total + 255) / 256);
  if (grid > 256 * 16) grid = 256 * 16;
  hipLaunchKernelGGL(mask_paste_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, masks, n, size, (const int*)rects,
                     (const int*)classids, image_h, image_w, (unsigned char*)full_masks, (unsigned char*)merged);
  DY_CHECK_LAUNCH();
  return DISYOLO_OK;
}

extern "C" int disyolo_letterbox(const uint8_t* rgb, int image_h, int image_w, float* out, int size, float* window_host,
                                 void* stream) {
  DY_REQUIRE(rgb && out && image_h > 0 && image_w > 0 && size > 0, "letterbox: bad args");
  int new_h, new_w;
  if (((double)size / image_w) < ((double)size / image_h)) {   // Python float division
    new_h = (int)(((int64_t)image_h * size) / image_w);
    new_w = size;
  } else {
    new_w = (int)(((int64_t)image_w * size) / image_h);
    new_h = size;
  }
  DY_REQUIRE(new_h > 0 && new_w > 0, "letterbox: degenerate image");
  const int top = (size - new_h) / 2, left = (size - new_w) / 2;
  if (window_host) {   // [top, left, bottom, right] normalised, float32 of the float64 quotient like numpy
    window_host[0] = (float)((double)top / size);
    window_host[1] = (float)((double)left / size);
    window_host[2] = (float)((double)(new_h + top) / size);
    window_host[3] = (float)((double)(new_w + left) / size);
  }
  DY_RECORD_OR_RUN([=](void* s) { return disyolo_letterbox(rgb, image_h, image_w, out, size, nullptr, s); });
  const int64_t total = (int64_t)size * size;
  int grid = (int)((total + 255) / 256);
  if (grid > 256 * 16) grid = 256 * 16;
  hipLaunchKernelGGL(letterbox_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, rgb, image_h, image_w, out, size,
                     new_h, new_w, top, left);
  DY_CHECK_LAUNCH();
  return DISYOLO_OK;
}

extern "C" int disyolo_confusion16(const uint8_t* true_map, const uint8_t* pred_map, int64_t n, int64_t* conf,
                                   void* stream) {
  DY_REQUIRE(true_map && pred_map && conf && n > 0, "confusion16: bad args");
  DY_RECORD_OR_RUN([=](void* s) { return disyolo_confusion16(true_map, pred_map, n, conf, s); });
  int grid = (int)((n + 255) / 256);
  if (grid > 1024) grid = 1024;
  hipLaunchKernelGGL(confusion16_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, true_map, pred_map, n,
                     (unsigned long long*)conf);
  DY_CHECK_LAUNCH();
  return DISYOLO_OK;
}
