// Training-data pipeline kernels (reference: utils/train_data.py:44-276, 321-531).
// The reference prepares every batch synchronously on the host (cv2 + scikit-image, :145) -- at the
// step rates of this path that is three orders of magnitude short, so the pixel work runs here:
//   polygon_mask   skimage.draw.polygon + the out/in/vertex rules of load_mask (:321-338)
//   place_image    cv2.resize(INTER_LINEAR, uint8) -> shift/crop/pad 127 -> flip (:466-494, 392-397)
//   place_mask     cv2.resize(INTER_LINEAR, float32) -> shift/crop/pad 0 -> flip -> np.around -> bool (:418-444)
//   salt_pepper / change_light / motion_blur3     the photometric augmentations (:496-531)
//   to_float       uint8 -> float32 / 255 (:411-413)
// Host code (train_data.py) only draws the random decisions and transforms the handful of boxes.
// scikit-image's rasteriser is pinned by golden vectors (tests/golden/polygon.json); cv2 and pyblur are not
// installable here: their arithmetic is restated from the libraries' documented behaviour -- unpinned.
#include "common.h"
#include "runtime.h"

namespace {

// scikit-image's point_in_polygon (measure/_pnpoly): 0 outside, non-zero inside / on a vertex / on an edge.
// Coordinates are doubles like there (the annotation lists are integers).
__device__ __forceinline__ int point_in_polygon(const float* xp, const float* yp, int n, double x, double y) {
  const double eps = 1e-12;
  double x0 = (double)xp[n - 1] - x, y0 = (double)yp[n - 1] - y;
  unsigned l = 0, r = 0;
  for (int i = 0; i < n; ++i) {
    const double x1 = (double)xp[i] - x, y1 = (double)yp[i] - y;
    if (x1 > -eps && x1 < eps && y1 > -eps && y1 < eps) return 2;
    if (((y0 > 0) != (y1 > 0)) && (__ddiv_rn(x0 * y1 - x1 * y0, y1 - y0) > 0)) ++r;
    if (((y0 < 0) != (y1 < 0)) && (__ddiv_rn(x0 * y1 - x1 * y0, y1 - y0) < 0)) ++l;
    x0 = x1;
    y0 = y1;
  }
  if ((r & 1) != (l & 1)) return 3;
  return (r & 1) ? 1 : 0;
}

// one instance = npoly polygons drawn in order: 'out' polygons set their pixels, 'in' polygons clear them
// (holes), and every polygon finally sets its own vertex pixels (utils/train_data.py:325-336)
__global__ __launch_bounds__(256) void polygon_mask_kernel(const float* px, const float* py, const int* start,
                                                           const int* type_out, int npoly, int H, int W,
                                                           unsigned char* mask) {
  const int64_t total = (int64_t)H * W;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int r = (int)(i / W), c = (int)(i - (int64_t)r * W);
    unsigned char v = 0;
    for (int k = 0; k < npoly; ++k) {
      const int a = start[k], n = start[k + 1] - a;
      if (n <= 0) continue;
      // bounding box of skimage's _polygon: rows [max(0,min), ceil(max)], same for columns
      float ymin = py[a], ymax = py[a], xmin = px[a], xmax = px[a];
      for (int j = 1; j < n; ++j) {
        ymin = fminf(ymin, py[a + j]); ymax = fmaxf(ymax, py[a + j]);
        xmin = fminf(xmin, px[a + j]); xmax = fmaxf(xmax, px[a + j]);
      }
      if (r >= (int)fmaxf(0.f, ymin) && r <= (int)ceilf(ymax) && c >= (int)fmaxf(0.f, xmin) && c <= (int)ceilf(xmax) &&
          point_in_polygon(px + a, py + a, n, (double)c, (double)r))
        v = type_out[k] ? 1 : 0;
      for (int j = 0; j < n; ++j)
        if ((int)py[a + j] == r && (int)px[a + j] == c) v = 1;
    }
    mask[i] = v;
  }
}

struct Taps {
  int s0, s1;
  float a;
};
// cv2.resize INTER_LINEAR source taps of destination index d (dn destination / sn source samples)
__device__ __forceinline__ Taps taps(int d, int dn, int sn) {
  float f = (float)(((double)d + 0.5) * ((double)sn / (double)dn) - 0.5);
  int s = (int)floorf(f);
  float a = __fsub_rn(f, (float)s);
  if (s < 0) { a = 0.f; s = 0; }
  if (s >= sn - 1) { a = 0.f; s = sn - 1; }
  return Taps{s, min(s + 1, sn - 1), a};
}

// destination pixel (y, x) of the S x S training image -> position in the resized image, or outside (pad):
// the resized image (new_w x new_h) is placed with its corner at (dx, dy) (negative = cropped), then flipped
__device__ __forceinline__ bool locate(int y, int x, int S, int new_w, int new_h, int dx, int dy, int flip, int* ry, int* rx) {
  if (flip == 2) x = S - 1 - x;     // horizontal flip: image[:, ::-1]
  if (flip == 3) y = S - 1 - y;     // vertical flip
  *ry = y - dy;
  *rx = x - dx;
  return *ry >= 0 && *ry < new_h && *rx >= 0 && *rx < new_w;
}

// uint8 image: OpenCV's 8-bit bilinear path works in fixed point -- coefficients rounded to 11 bits,
// horizontal pass in int, vertical pass (b0*S0 + b1*S1 + 2^21) >> 22
__device__ __forceinline__ void place_image_px(const unsigned char* src, int H, int W, unsigned char* dst, int S, int new_w, int new_h,
                                               int dx, int dy, int flip, int64_t i) {
  const int y = (int)(i / S), x = (int)(i - (int64_t)y * S);
  int ry, rx;
  unsigned char v[3] = {127, 127, 127};
  if (locate(y, x, S, new_w, new_h, dx, dy, flip, &ry, &rx)) {
    const Taps tx = taps(rx, new_w, W), ty = taps(ry, new_h, H);
    const int ax1 = (int)rintf(tx.a * 2048.f), ax0 = 2048 - ax1;
    const int ay1 = (int)rintf(ty.a * 2048.f), ay0 = 2048 - ay1;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const int h0 = src[((size_t)ty.s0 * W + tx.s0) * 3 + c] * ax0 + src[((size_t)ty.s0 * W + tx.s1) * 3 + c] * ax1;
      const int h1 = src[((size_t)ty.s1 * W + tx.s0) * 3 + c] * ax0 + src[((size_t)ty.s1 * W + tx.s1) * 3 + c] * ax1;
      const int r = (h0 * ay0 + h1 * ay1 + (1 << 21)) >> 22;
      v[c] = (unsigned char)min(max(r, 0), 255);
    }
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) dst[i * 3 + c] = v[c];
}
__global__ __launch_bounds__(256) void place_image_kernel(const unsigned char* src, int H, int W, unsigned char* dst, int S,
                                                          int new_w, int new_h, int dx, int dy, int flip) {
  const int64_t total = (int64_t)S * S;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x)
    place_image_px(src, H, W, dst, S, new_w, new_h, dx, dy, flip, i);
}

// float mask (0/1 bytes in, treated as float32): float bilinear path, pad 0, np.around (half to even) -> bool
__device__ __forceinline__ void place_mask_px(const unsigned char* src, int H, int W, unsigned char* dst, int S, int new_w, int new_h,
                                              int dx, int dy, int flip, int64_t i) {
  const int y = (int)(i / S), x = (int)(i - (int64_t)y * S);
  int ry, rx;
  float v = 0.f;
  if (locate(y, x, S, new_w, new_h, dx, dy, flip, &ry, &rx)) {
    const Taps tx = taps(rx, new_w, W), ty = taps(ry, new_h, H);
    const float bx = __fsub_rn(1.f, tx.a), by = __fsub_rn(1.f, ty.a);
    const float h0 = __fadd_rn(__fmul_rn((float)src[(size_t)ty.s0 * W + tx.s0], bx), __fmul_rn((float)src[(size_t)ty.s0 * W + tx.s1], tx.a));
    const float h1 = __fadd_rn(__fmul_rn((float)src[(size_t)ty.s1 * W + tx.s0], bx), __fmul_rn((float)src[(size_t)ty.s1 * W + tx.s1], tx.a));
    v = __fadd_rn(__fmul_rn(h0, by), __fmul_rn(h1, ty.a));
  }
  dst[i] = rintf(v) != 0.f ? 1 : 0;
}
__global__ __launch_bounds__(256) void place_mask_kernel(const unsigned char* src, int H, int W, unsigned char* dst, int S,
                                                         int new_w, int new_h, int dx, int dy, int flip) {
  const int64_t total = (int64_t)S * S;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x)
    place_mask_px(src, H, W, dst, S, new_w, new_h, dx, dy, flip, i);
}

// a whole batch's placements in ONE launch (round 6: the loader's ~25 per-image / per-instance launches were its GPU time):
// blockIdx.y = job; the per-pixel arithmetic is the single-job kernels' own
__global__ __launch_bounds__(256) void place_batch_kernel(const disyolo_place_job* jobs, int S) {
  const disyolo_place_job j = jobs[blockIdx.y];
  const int64_t total = (int64_t)S * S;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    if (j.is_mask)
      place_mask_px(j.src, j.image_h, j.image_w, j.dst, S, j.new_w, j.new_h, j.dx, j.dy, j.flip, i);
    else
      place_image_px(j.src, j.image_h, j.image_w, j.dst, S, j.new_w, j.new_h, j.dx, j.dy, j.flip, i);
  }
}

// add_salt_pepper_noise (:511-525): im[rows, cols, :] = 1 for the salt coordinates, then 0 for the pepper ones
__global__ void salt_pepper_kernel(unsigned char* img, int S, const int* rows, const int* cols, int nsalt, int npepper) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nsalt + npepper) return;
  const unsigned char v = i < nsalt ? 1 : 0;
  unsigned char* p = img + ((size_t)rows[i] * S + cols[i]) * 3;
  p[0] = v; p[1] = v; p[2] = v;
}

// change_light (:527-535): RGB -> HLS (8-bit: H/2, 255 L, 255 S), L *= coeff (float64, clipped at 255, truncated
// to uint8), HLS -> RGB.  Formulas of OpenCV's colour-conversion documentation, float32, rounded to nearest.
__device__ __forceinline__ unsigned char sat8(float v) { return (unsigned char)min(max((int)rintf(v), 0), 255); }
__global__ __launch_bounds__(256) void change_light_kernel(unsigned char* img, int64_t npix, double coeff) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += (int64_t)gridDim.x * blockDim.x) {
    const float r = img[i * 3] * (1.f / 255.f), g = img[i * 3 + 1] * (1.f / 255.f), b = img[i * 3 + 2] * (1.f / 255.f);
    const float vmax = fmaxf(r, fmaxf(g, b)), vmin = fminf(r, fminf(g, b));
    const float diff = vmax - vmin, sum = vmax + vmin;
    float h = 0.f, s = 0.f;
    const float l = sum * 0.5f;
    if (diff > 1.1920929e-07f) {
      s = l < 0.5f ? diff / sum : diff / (2.f - sum);
      const float d60 = 60.f / diff;
      if (vmax == r) h = (g - b) * d60;
      else if (vmax == g) h = (b - r) * d60 + 120.f;
      else h = (r - g) * d60 + 240.f;
      if (h < 0.f) h += 360.f;
    }
    const unsigned char H8 = sat8(h * 0.5f), S8 = sat8(s * 255.f);
    double L = (double)sat8(l * 255.f) * coeff;
    if (L > 255.0) L = 255.0;
    const unsigned char L8 = (unsigned char)L;           // np.array(..., dtype=np.uint8): truncation
    // back: HLS -> RGB
    const float hh = (float)H8 * 2.f, ll = (float)L8 * (1.f / 255.f), ss = (float)S8 * (1.f / 255.f);
    float ro, go, bo;
    if (ss == 0.f) {
      ro = go = bo = ll;
    } else {
      const float p2 = ll <= 0.5f ? ll * (1.f + ss) : ll + ss - ll * ss;
      const float p1 = 2.f * ll - p2;
      float hq = hh * (1.f / 60.f);
      if (hq < 0.f) do hq += 6.f; while (hq < 0.f);
      else if (hq >= 6.f) do hq -= 6.f; while (hq >= 6.f);
      const int sector = (int)floorf(hq);
      const float f = hq - (float)sector;
      const float tab[4] = {p2, p1, p1 + (p2 - p1) * (1.f - f), p1 + (p2 - p1) * f};
      const int idx[6][3] = {{1, 3, 0}, {1, 0, 2}, {3, 0, 1}, {0, 2, 1}, {0, 1, 3}, {2, 1, 0}};   // (b, g, r) per sector
      bo = tab[idx[sector][0]]; go = tab[idx[sector][1]]; ro = tab[idx[sector][2]];
    }
    img[i * 3] = sat8(ro * 255.f); img[i * 3 + 1] = sat8(go * 255.f); img[i * 3 + 2] = sat8(bo * 255.f);
  }
}

// linearmotion_blur3C (:466-494) with lineLength 3: a 3x3 line kernel through the centre (angle 0 / 45 / 90 / 135,
// "full" = three taps, "right" / "left" = the centre and one neighbour), normalised; pixels outside the image count as
// 255 (pyblur's LinearMotionBlur: scipy.signal.convolve2d(img, kernel, mode='same', fillvalue=255.0)), result truncated
// to uint8
__global__ __launch_bounds__(256) void motion_blur3_kernel(const unsigned char* src, unsigned char* dst, int S, int dyA, int dxA,
                                                           int use_a, int use_b) {
  const int64_t total = (int64_t)S * S;
  const float wgt = 1.f / (float)(1 + use_a + use_b);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int y = (int)(i / S), x = (int)(i - (int64_t)y * S);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      // taps in the (flipped) kernel's row-major order: the "b" end first when it lies above / left of the centre
      float acc = 0.f;
      const int ya = y + dyA, xa = x + dxA, yb = y - dyA, xb = x - dxA;
      const float va = (ya >= 0 && ya < S && xa >= 0 && xa < S) ? (float)src[((size_t)ya * S + xa) * 3 + c] : 255.f;
      const float vb = (yb >= 0 && yb < S && xb >= 0 && xb < S) ? (float)src[((size_t)yb * S + xb) * 3 + c] : 255.f;
      if (use_a) acc += va * wgt;
      acc += (float)src[i * 3 + c] * wgt;
      if (use_b) acc += vb * wgt;
      dst[i * 3 + c] = (unsigned char)min(max((int)acc, 0), 255);
    }
  }
}

__global__ __launch_bounds__(256) void to_float_kernel(const unsigned char* src, float* dst, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    dst[i] = __fdiv_rn((float)src[i], 255.f);           // image.astype(float32) / 255.0 stays float32 in numpy
}

int grid_for(int64_t n) {
  int64_t g = (n + 255) / 256;
  if (g > 256 * 16) g = 256 * 16;
  return (int)(g < 1 ? 1 : g);
}

}  // namespace

extern "C" int disyolo_polygon_mask(const float* px, const float* py, const int32_t* poly_start, const int32_t* poly_is_out,
                                    int npoly, int image_h, int image_w, uint8_t* mask, void* stream) {
  DY_REQUIRE(px && py && poly_start && poly_is_out && mask && npoly >= 0 && image_h > 0 && image_w > 0, "polygon_mask: bad args");
  DY_RECORD_OR_RUN([=](void* s) { return disyolo_polygon_mask(px, py, poly_start, poly_is_out, npoly, image_h, image_w, mask, s); });
  hipLaunchKernelGGL(polygon_mask_kernel, dim3(grid_for((int64_t)image_h * image_w)), dim3(256), 0, (hipStream_t)stream, px, py,
                     poly_start, poly_is_out, npoly, image_h, image_w, mask);
  DY_CHECK_LAUNCH();
  return DISYOLO_OK;
}

extern "C" int disyolo_aug_place(const uint8_t* src, int is_mask, int image_h, int image_w, uint8_t* dst, int size, int new_w,
                                 int new_h, int dx, int dy, int flip, void* stream) {
  DY_REQUIRE(src && dst && image_h > 0 && image_w > 0 && size > 0 && new_w > 0 && new_h > 0 && flip >= 1 && flip <= 3,
             "aug_place: bad args");
  DY_RECORD_OR_RUN([=](void* s) { return disyolo_aug_place(src, is_mask, image_h, image_w, dst, size, new_w, new_h, dx, dy, flip, s); });
  const int g = grid_for((int64_t)size * size);
  if (is_mask)
    hipLaunchKernelGGL(place_mask_kernel, dim3(g), dim3(256), 0, (hipStream_t)stream, src, image_h, image_w, dst, size, new_w, new_h, dx, dy, flip);
  else
    hipLaunchKernelGGL(place_image_kernel, dim3(g), dim3(256), 0, (hipStream_t)stream, src, image_h, image_w, dst, size, new_w, new_h, dx, dy, flip);
  DY_CHECK_LAUNCH();
  return DISYOLO_OK;
}

extern "C" int disyolo_aug_place_batch(const disyolo_place_job* jobs, int njobs, int size, void* stream) {
  DY_REQUIRE(jobs && njobs > 0 && njobs <= 65535 && size > 0, "aug_place_batch: bad args");
  DY_RECORD_OR_RUN([=](void* s) { return disyolo_aug_place_batch(jobs, njobs, size, s); });
  int gx = grid_for((int64_t)size * size);
  if (gx > 512) gx = 512;     // (grid-stride: the jobs of a batch fill the chip between them)
  hipLaunchKernelGGL(place_batch_kernel, dim3(gx, njobs), dim3(256), 0, (hipStream_t)stream, jobs, size);
  DY_CHECK_LAUNCH();
  return DISYOLO_OK;
}

extern "C" int disyolo_aug_salt_pepper(uint8_t* image, int size, const int32_t* rows, const int32_t* cols, int nsalt, int npepper,
                                       void* stream) {
  DY_REQUIRE(image && rows && cols && size > 0 && nsalt >= 0 && npepper >= 0, "aug_salt_pepper: bad args");
  DY_RECORD_OR_RUN([=](void* s) { return disyolo_aug_salt_pepper(image, size, rows, cols, nsalt, npepper, s); });
  const int n = nsalt + npepper;
  if (n == 0) return DISYOLO_OK;
  // two launches keep the reference's order: all salt writes, then all pepper writes
  if (nsalt) hipLaunchKernelGGL(salt_pepper_kernel, dim3((nsalt + 255) / 256), dim3(256), 0, (hipStream_t)stream, image, size, rows, cols, nsalt, 0);
  if (npepper) hipLaunchKernelGGL(salt_pepper_kernel, dim3((npepper + 255) / 256), dim3(256), 0, (hipStream_t)stream, image, size, rows + nsalt, cols + nsalt, 0, npepper);
  DY_CHECK_LAUNCH();
  return DISYOLO_OK;
}

extern "C" int disyolo_aug_change_light(uint8_t* image, int size, double coeff, void* stream) {
  DY_REQUIRE(image && size > 0 && coeff >= 0.0, "aug_change_light: bad args");
  DY_RECORD_OR_RUN([=](void* s) { return disyolo_aug_change_light(image, size, coeff, s); });
  hipLaunchKernelGGL(change_light_kernel, dim3(grid_for((int64_t)size * size)), dim3(256), 0, (hipStream_t)stream, image, (int64_t)size * size, coeff);
  DY_CHECK_LAUNCH();
  return DISYOLO_OK;
}

extern "C" int disyolo_aug_motion_blur3(const uint8_t* src, uint8_t* dst, int size, int angle, int line_type, void* stream) {
  DY_REQUIRE(src && dst && src != dst && size > 0 && (angle == 0 || angle == 45 || angle == 90 || angle == 135) && line_type >= 0 && line_type <= 2,
             "aug_motion_blur3: bad args (angle 0/45/90/135, line_type 0 full / 1 right / 2 left)");
  DY_RECORD_OR_RUN([=](void* s) { return disyolo_aug_motion_blur3(src, dst, size, angle, line_type, s); });
  // the "right" end of the 3x3 line in kernel coordinates (row down): pyblur's LineDictionary anchors
  // {0: [1,0,1,2], 45: [2,0,0,2], 90: [0,1,2,1], 135: [0,0,2,2]} = (row0, col0, row1, col1); 'right' keeps the SECOND
  // anchor (the first is replaced by the centre), 'left' the first: 0: +x, 45: up-right, 90: DOWN, 135: DOWN-right.
  // (Restated from pyblur 0.2.x, which is not installable here: unpinned.  pyblur also mutates the shared anchor list in
  // place, so after one 'right' and one 'left' call an angle degenerates to the identity for the rest of the process: not
  // reproduced.)
  const int dxA = angle == 0 ? 1 : (angle == 45 ? 1 : (angle == 90 ? 0 : 1));
  const int dyA = angle == 0 ? 0 : (angle == 45 ? -1 : 1);
  const int use_a = line_type != 2, use_b = line_type != 1;
  // a convolution flips the kernel: the tap at kernel offset (+dy,+dx) reads the pixel at (-dy,-dx)
  hipLaunchKernelGGL(motion_blur3_kernel, dim3(grid_for((int64_t)size * size)), dim3(256), 0, (hipStream_t)stream, src, dst, size, -dyA,
                     -dxA, use_a, use_b);
  DY_CHECK_LAUNCH();
  return DISYOLO_OK;
}

extern "C" int disyolo_aug_to_float(const uint8_t* image, float* out, int64_t n, void* stream) {
  DY_REQUIRE(image && out && n > 0, "aug_to_float: bad args");
  DY_RECORD_OR_RUN([=](void* s) { return disyolo_aug_to_float(image, out, n, s); });
  hipLaunchKernelGGL(to_float_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, image, out, n);
  DY_CHECK_LAUNCH();
  return DISYOLO_OK;
}
