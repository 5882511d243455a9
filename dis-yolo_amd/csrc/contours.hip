// Host-side helper of the dataset pre-processing (pre_process.py:66-86 of the reference): the border following
// behind cv2.findContours(img, cv2.RETR_TREE, cv2.CHAIN_APPROX_NONE) -- Suzuki & Abe 1985, Algorithm 1, with the
// search orders, point order and contour numbering of OpenCV's legacy implementation (see the conventions listed
// in oracle/disyolo_oracle.py:find_contours_tree; cv2 itself is not available here: parity unpinned).  Plain C++,
// no device code: one mask image is a few hundred KB and is visited once.
#include <stdint.h>
#include <stdlib.h>
#include <vector>
#include "common.h"

namespace {
const int kDx[8] = {1, 1, 0, -1, -1, -1, 0, 1};
const int kDy[8] = {0, -1, -1, -1, 0, 1, 1, 1};
}  // namespace

// binary: h x w bytes (non-zero = foreground).  Outputs: points_xy int32 [n_points][2] (x, y), contour_start int32
// [n_contours + 1] (prefix offsets into points), hierarchy int32 [n_contours][4] = next, previous, first child,
// parent.  Returns DISYOLO_E_WORKSPACE when max_points / max_contours are too small; *n_points / *n_contours then
// hold the sizes needed.
extern "C" int disyolo_find_contours(const uint8_t* binary, int h, int w, int32_t* points_xy, int64_t max_points,
                                     int32_t* contour_start, int32_t* hierarchy, int max_contours, int* n_contours,
                                     int64_t* n_points) {
  DY_REQUIRE(binary && h > 0 && w > 0 && n_contours && n_points, "find_contours: bad args");
  const int W = w + 2;
  std::vector<int32_t> f((size_t)(h + 2) * W, 0);
  for (int y = 0; y < h; ++y)
    for (int x = 0; x < w; ++x) f[(size_t)(y + 1) * W + x + 1] = binary[(size_t)y * w + x] ? 1 : 0;
  std::vector<char> is_hole(2, 1);          // index = NBD; 1 = the frame
  std::vector<int32_t> parent(2, 0);
  std::vector<std::vector<int32_t>> pts(2);
  int nbd = 1;
  for (int i = 1; i <= h; ++i) {
    int lnbd = 1;
    for (int j = 1; j <= w; ++j) {
      int32_t* p = &f[(size_t)i * W + j];
      int start = -1;                        // 0 outer, 1 hole
      if (*p == 1 && p[-1] == 0) {
        start = 0;
      } else if (*p >= 1 && p[1] == 0) {
        start = 1;
        if (*p > 1) lnbd = *p;
      }
      if (start >= 0) {
        ++nbd;
        is_hole.push_back((char)start);
        parent.push_back(start ? (!is_hole[lnbd] ? lnbd : parent[lnbd]) : (!is_hole[lnbd] ? parent[lnbd] : lnbd));
        pts.emplace_back();
        std::vector<int32_t>& out = pts.back();
        int s = start ? 0 : 4;
        const int s_first = s;
        bool found = false;
        do {
          s = (s - 1) & 7;
          if (p[kDy[s] * W + kDx[s]] != 0) { found = true; break; }
        } while (s != s_first);
        if (!found) {
          *p = -nbd;
          out.push_back(j - 1);
          out.push_back(i - 1);
        } else {
          int32_t* const p0 = p;
          int32_t* const p1 = p + kDy[s] * W + kDx[s];
          int32_t* p3 = p;
          for (;;) {
            const int s_end = s;
            for (;;) {
              s = (s + 1) & 7;
              if (p3[kDy[s] * W + kDx[s]] != 0) break;
            }
            int32_t* p4 = p3 + kDy[s] * W + kDx[s];
            const bool passed_east = ((0 - (s_end + 1)) & 7) < ((s - (s_end + 1)) & 7) && p3[1] == 0;
            if (passed_east)
              *p3 = -nbd;
            else if (*p3 == 1)
              *p3 = nbd;
            const int64_t off = p3 - f.data();
            out.push_back((int32_t)(off % W) - 1);
            out.push_back((int32_t)(off / W) - 1);
            if (p4 == p0 && p3 == p1) break;
            p3 = p4;
            s = (s + 4) & 7;
          }
        }
      }
      if (*p != 0 && *p != 1) lnbd = *p < 0 ? -*p : *p;
    }
  }
  // numbering: pre-order of the tree, siblings newest first
  const int nb = nbd + 1;
  std::vector<std::vector<int>> kids(nb);
  for (int b = 2; b <= nbd; ++b) kids[parent[b]].push_back(b);
  std::vector<int> seq, index(nb, -1), stack;
  for (int k = 0; k < (int)kids[1].size(); ++k) stack.push_back(kids[1][k]);   // popped newest first
  while (!stack.empty()) {
    const int b = stack.back();
    stack.pop_back();
    index[b] = (int)seq.size();
    seq.push_back(b);
    for (int k = 0; k < (int)kids[b].size(); ++k) stack.push_back(kids[b][k]);
  }
  int64_t total = 0;
  for (int b : seq) total += (int64_t)pts[b].size() / 2;
  *n_contours = (int)seq.size();
  *n_points = total;
  if ((int)seq.size() > max_contours || total > max_points || !points_xy || !contour_start || !hierarchy) {
    disyolo_set_error("find_contours: %d contours / %lld points do not fit the output buffers", (int)seq.size(), (long long)total);
    return DISYOLO_E_WORKSPACE;
  }
  int64_t at = 0;
  for (size_t k = 0; k < seq.size(); ++k) {
    contour_start[k] = (int32_t)at;
    const std::vector<int32_t>& v = pts[seq[k]];
    for (size_t q = 0; q < v.size(); ++q) points_xy[at * 2 + q] = v[q];
    at += (int64_t)v.size() / 2;
    for (int q = 0; q < 4; ++q) hierarchy[k * 4 + q] = -1;
  }
  contour_start[seq.size()] = (int32_t)at;
  for (int pnode = 1; pnode <= nbd; ++pnode) {
    const std::vector<int>& ks = kids[pnode];
    const int n = (int)ks.size();
    for (int k = n - 1; k >= 0; --k) {       // sibling order: newest first
      const int me = index[ks[k]];
      if (k > 0) hierarchy[me * 4 + 0] = index[ks[k - 1]];
      if (k < n - 1) hierarchy[me * 4 + 1] = index[ks[k + 1]];
      if (pnode != 1) hierarchy[me * 4 + 3] = index[pnode];
    }
    if (pnode != 1 && n > 0) hierarchy[index[pnode] * 4 + 2] = index[ks[n - 1]];
  }
  return DISYOLO_OK;
}
