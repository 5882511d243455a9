// Detection decode + filter and the inference-time position-sensitive mask assembly.
//
// Replaces interpret_output / filter_detections / clip_boxes_graph / val_test of
// yolo/yolo3_net_pos.py:465-628, 862-952 (the reference unrolls these per image in Python
// and runs tf.image.non_max_suppression per class through tf.map_fn).
#include "common.h"

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

// One block per image: threshold + ordered compaction, per-class greedy NMS (IoU > thr
// suppresses, <= max_det kept per class), then the top max_det by score (ties: lower
// candidate index), zero padded.  Exact for any number of candidates: selection is a
// repeated block-wide arg-max over the live list, never a truncated sort.
constexpr int NMS_T = 256;
constexpr int MAX_KEEP = 512;  // >= num_class * max_det
__global__ __launch_bounds__(NMS_T) void nms_kernel(const float4* boxes, const float* scores, const int* classes,
                                                    int NC, int C, float thr, float nms_thr, int max_det, int* list,
                                                    float* live, float* det, int* det_count) {
  const int b = blockIdx.x, tid = threadIdx.x;
  boxes += (size_t)b * NC;
  scores += (size_t)b * NC;
  classes += (size_t)b * NC;
  list += (size_t)b * NC;
  live += (size_t)b * NC;
  __shared__ int s_wcnt[2][NMS_T / 64];
  __shared__ int s_n;
  __shared__ float s_bs[NMS_T];
  __shared__ int s_bp[NMS_T];
  __shared__ int s_keep_idx[MAX_KEEP];
  __shared__ float s_keep_sc[MAX_KEEP];
  __shared__ int s_nkeep;
  if (tid == 0) {
    s_n = 0;
    s_nkeep = 0;
  }
  __syncthreads();
  // ---- ordered compaction of candidates with score > thr (wave ballots + one barrier)
  int total = 0;
  for (int base = 0, it = 0; base < NC; base += NMS_T, ++it) {
    const int i = base + tid;
    const bool f = (i < NC) && (scores[i] > thr);
    const unsigned long long mask = __ballot(f);
    const int lane = tid & 63, wv = tid >> 6;
    if (lane == 0) s_wcnt[it & 1][wv] = __popcll(mask);
    __syncthreads();
    int off = total;
#pragma unroll
    for (int w = 0; w < NMS_T / 64; ++w) {
      const int cw = s_wcnt[it & 1][w];
      if (w < wv) off += cw;
      total += cw;
    }
    if (f) list[off + __popcll(mask & ((1ull << lane) - 1ull))] = i;
  }
  if (tid == 0) s_n = total;
  __syncthreads();
  const int n = s_n;
  __threadfence_block();
  // ---- per-class greedy NMS
  for (int c = 0; c < C; ++c) {
    for (int q = tid; q < n; q += NMS_T) {
      const int i = list[q];
      live[q] = classes[i] == c ? scores[i] : -1.f;
    }
    __syncthreads();
    for (int it = 0; it < max_det; ++it) {
      float bs = -1.f;
      int bp = 0x7fffffff;
      for (int q = tid; q < n; q += NMS_T) {
        const float v = live[q];
        if (v > bs) {  // strided scan keeps the lowest position among equal scores
          bs = v;
          bp = q;
        }
      }
      s_bs[tid] = bs;
      s_bp[tid] = bp;
      __syncthreads();
      for (int o = NMS_T / 2; o > 0; o >>= 1) {
        if (tid < o) {
          const float v2 = s_bs[tid + o];
          const int p2 = s_bp[tid + o];
          if (v2 > s_bs[tid] || (v2 == s_bs[tid] && p2 < s_bp[tid])) {
            s_bs[tid] = v2;
            s_bp[tid] = p2;
          }
        }
        __syncthreads();
      }
      const float wsc = s_bs[0];
      const int wq = s_bp[0];
      __syncthreads();
      if (!(wsc > 0.f)) break;  // nothing alive (scores are > thr >= 0)... uniform exit
      const int wi = list[wq];
      const float4 wb = boxes[wi];
      if (tid == 0) {
        s_keep_idx[s_nkeep] = wi;
        s_keep_sc[s_nkeep] = wsc;
        ++s_nkeep;
      }
      for (int q = tid; q < n; q += NMS_T) {
        if (live[q] >= 0.f) {
          if (q == wq || nms_iou(boxes[list[q]], wb) > nms_thr) live[q] = -1.f;
        }
      }
      __syncthreads();
    }
    __syncthreads();
  }
  // ---- top max_det of the kept set: score descending, ties by lower candidate index
  const int nk = s_nkeep;
  for (int r = tid; r < max_det * 6; r += NMS_T) det[(size_t)b * max_det * 6 + r] = 0.f;
  __syncthreads();
  for (int k = tid; k < nk; k += NMS_T) {
    const float sc = s_keep_sc[k];
    const int ix = s_keep_idx[k];
    int rank = 0;
    for (int j = 0; j < nk; ++j) {
      const float sj = s_keep_sc[j];
      if (sj > sc || (sj == sc && s_keep_idx[j] < ix)) ++rank;
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

}  // namespace

extern "C" size_t disyolo_detect_workspace(int B, int S, int num_class) {
  if (B <= 0 || S <= 0 || S % 32) return 0;
  const int g1 = S / 32;
  const size_t NC = 3 * (size_t)(16 * g1 * g1 + 4 * g1 * g1 + g1 * g1);
  // boxes (16 B) + scores + classes + list + live
  return (size_t)B * NC * (16 + 4 + 4 + 4 + 4);
}

extern "C" int disyolo_detect(const float* logits3, const float* logits2, const float* logits1, int B, int S,
                              int num_class, const float* anchors_host, const float* clip_window, float obj_thresh,
                              float nms_thresh, int max_det, float* detections, int32_t* det_count, void* workspace,
                              size_t workspace_bytes, void* stream) {
  DY_REQUIRE(logits3 && logits2 && logits1 && anchors_host && clip_window && detections && det_count,
             "detect: null pointer");
  DY_REQUIRE(B > 0 && S > 0 && S % 32 == 0 && num_class > 0 && num_class <= 16, "detect: bad sizes");
  DY_REQUIRE(max_det > 0 && num_class * max_det <= MAX_KEEP, "detect: num_class*max_det > %d", MAX_KEEP);
  DY_REQUIRE(obj_thresh >= 0.f, "detect: obj_thresh must be >= 0");
  if (!workspace || workspace_bytes < disyolo_detect_workspace(B, S, num_class)) {
    disyolo_set_error("detect: workspace too small");
    return DISYOLO_E_WORKSPACE;
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
  int* list = (int*)ws;                   ws += (size_t)B * c0 * 4;
  float* live = (float*)ws;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(decode_score_kernel, dim3(ceil_div(c0, 256), B), dim3(256), 0, st, p);
  DY_CHECK_LAUNCH();
  hipLaunchKernelGGL(nms_kernel, dim3(B), dim3(NMS_T), 0, st, p.boxes, p.scores, p.classes, c0, num_class, obj_thresh,
                     nms_thresh, max_det, list, live, detections, det_count);
  DY_CHECK_LAUNCH();
  return DISYOLO_OK;
}

extern "C" int disyolo_psroi_assemble(const float* score, const float* detections, int B, int max_det, int map_size,
                                      int k, float* masks, int32_t* keep, void* stream) {
  DY_REQUIRE(score && detections && masks && keep && B > 0 && max_det > 0 && map_size > 0, "psroi_assemble: bad args");
  DY_REQUIRE(k == 3, "psroi_assemble: only k = 3 (the reference's active branch, yolo/yolo3_net_pos.py:894-897)");
  const int npx = map_size * map_size;
  int gx = ceil_div(npx, 256 * 4);
  if (gx < 1) gx = 1;
  hipLaunchKernelGGL(psroi_assemble_kernel, dim3(gx, max_det, B), dim3(256), 0, (hipStream_t)stream, score, detections,
                     B, max_det, map_size, masks, keep);
  DY_CHECK_LAUNCH();
  return DISYOLO_OK;
}
