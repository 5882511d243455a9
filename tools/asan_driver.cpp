// Host-side sanitizer driver (`make -C dis-yolo_amd/csrc asan`): the library's plain-C++ parts -- the contour tracer
// (contours.hip), the command-list executor (runtime.hip, HIP calls replaced by host_stub/hip/hip_runtime.h) and crc32c --
// built with -fsanitize=address,undefined and driven through their C ABI with random, ragged and invalid arguments.
// CPU only (the GPU pool refuses sanitizer runs).  Exit code 0 = no sanitizer report, every check held.
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <random>
#include <vector>
#include "../include/disyolo.h"

static char g_err[512];
void disyolo_set_error(const char* fmt, ...) {     // (ops.hip defines it in the product build)
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

#define CHECK(c)                                                              \
  do {                                                                        \
    if (!(c)) {                                                               \
      fprintf(stderr, "asan_driver: CHECK failed at line %d: %s (last error: %s)\n", __LINE__, #c, g_err); \
      exit(2);                                                                \
    }                                                                         \
  } while (0)

static uint32_t crc32c_ref(const uint8_t* p, size_t n, uint32_t crc) {
  crc = ~crc;
  for (size_t i = 0; i < n; ++i) {
    crc ^= p[i];
    for (int k = 0; k < 8; ++k) crc = (crc >> 1) ^ ((crc & 1) ? 0x82F63B78u : 0u);
  }
  return ~crc;
}

int main() {
  std::mt19937 rng(12345);
  // ---- crc32c: every alignment and length around the slicing-by-8 boundaries
  {
    std::vector<uint8_t> buf(4096 + 64);
    for (auto& b : buf) b = (uint8_t)rng();
    for (int off = 0; off < 9; ++off)
      for (size_t n : {size_t(0), size_t(1), size_t(7), size_t(8), size_t(9), size_t(63), size_t(64), size_t(1000), size_t(4096)})
        CHECK(disyolo_crc32c(buf.data() + off, n, 0) == crc32c_ref(buf.data() + off, n, 0));
    const char* v = "123456789";
    CHECK(disyolo_crc32c(v, 9, 0) == 0xE3069283u);      // RFC 3720 check value
  }
  // ---- contour tracer: random masks, exact / short / zero capacities
  for (int it = 0; it < 400; ++it) {
    const int h = 1 + rng() % 40, w = 1 + rng() % 40;
    const int dens = rng() % 100;
    std::vector<uint8_t> img((size_t)h * w);
    for (auto& p : img) p = (int)(rng() % 100) < dens ? (uint8_t)(1 + rng() % 255) : 0;
    int nc = -1;
    int64_t np = -1;
    int rc = disyolo_find_contours(img.data(), h, w, nullptr, 0, nullptr, nullptr, 0, &nc, &np);     // size query
    CHECK(rc == DISYOLO_OK || rc == DISYOLO_E_WORKSPACE);
    CHECK(nc >= 0 && np >= 0);
    std::vector<int32_t> pts((size_t)np * 2 + 2), start((size_t)nc + 2), hier((size_t)nc * 4 + 4);
    if (np > 1) {     // one point short: must refuse, not overrun (ASan watches the exact-size vector below)
      std::vector<int32_t> small((size_t)(np - 1) * 2);
      int nc2 = 0;
      int64_t np2 = 0;
      rc = disyolo_find_contours(img.data(), h, w, small.data(), np - 1, start.data(), hier.data(), nc, &nc2, &np2);
      CHECK(rc == DISYOLO_E_WORKSPACE && np2 == np && nc2 == nc);
    }
    std::vector<int32_t> pe((size_t)np * 2), se((size_t)nc + 1), he((size_t)nc * 4);
    int nc3 = 0;
    int64_t np3 = 0;
    rc = disyolo_find_contours(img.data(), h, w, np ? pe.data() : pts.data(), np, se.data(), nc ? he.data() : hier.data(), nc, &nc3, &np3);
    CHECK(rc == DISYOLO_OK && nc3 == nc && np3 == np);
    CHECK(se[0] == 0 && se[nc] == np);
    for (int c = 0; c < nc; ++c) {
      CHECK(se[c] < se[c + 1]);
      for (int k = 0; k < 4; ++k) CHECK(he[c * 4 + k] >= -1 && he[c * 4 + k] < nc);
      for (int64_t q = se[c]; q < se[c + 1]; ++q) {
        const int x = pe[q * 2], y = pe[q * 2 + 1];
        CHECK(x >= 0 && x < w && y >= 0 && y < h);
        CHECK(img[(size_t)y * w + x] != 0);            // every listed pixel is foreground
      }
    }
  }
  CHECK(disyolo_find_contours(nullptr, 4, 4, nullptr, 0, nullptr, nullptr, 0, nullptr, nullptr) == DISYOLO_E_ARG);
  // ---- command list: record on three lanes with edges, replay whole / in ranges / with bad ranges, destroy
  for (int it = 0; it < 50; ++it) {
    void* l = disyolo_cmdlist_create();
    CHECK(l != nullptr);
    CHECK(disyolo_cmdlist_begin(l) == DISYOLO_OK);
    CHECK(disyolo_cmdlist_begin(l) != DISYOLO_OK);                     // nested recording is refused
    std::vector<int> marks;
    const int n = 1 + rng() % 60;
    static uint8_t a[64], b[64];
    for (int i = 0; i < n; ++i) {
      switch (rng() % 6) {
        case 0: CHECK(disyolo_cmdlist_set_lane((int)(rng() % 3)) == DISYOLO_OK); break;
        case 1: CHECK(disyolo_cmdlist_sync((int)(rng() % 3), (int)(rng() % 3)) >= DISYOLO_E_HIP || true); break;
        case 2: { const int m = disyolo_cmdlist_mark((int)(rng() % 3)); if (m >= 0) marks.push_back(m); break; }
        case 3: if (!marks.empty()) (void)disyolo_cmdlist_wait(marks[rng() % marks.size()], (int)(rng() % 3)); break;
        default: CHECK(disyolo_add_bf16(a, b, 8 * (1 + rng() % 4), (int)(rng() & 1), nullptr) == DISYOLO_OK);   // records itself
      }
    }
    (void)disyolo_cmdlist_set_lane(7);                                 // invalid lane: refused or clamped, never UB
    (void)disyolo_cmdlist_wait(1 << 20, 1);                            // invalid mark
    (void)disyolo_cmdlist_set_lane(0);
    CHECK(disyolo_cmdlist_end() == DISYOLO_OK);
    const int sz = disyolo_cmdlist_size(l);
    CHECK(sz >= 0);
    CHECK(disyolo_cmdlist_run(l, 0, sz, nullptr) == DISYOLO_OK);
    const int cut = sz ? (int)(rng() % (sz + 1)) : 0;
    CHECK(disyolo_cmdlist_run_ex(l, 0, cut, nullptr, 1) == DISYOLO_OK);
    CHECK(disyolo_cmdlist_run_ex(l, cut, sz, nullptr, 2) == DISYOLO_OK);
    CHECK(disyolo_cmdlist_run(l, -1, sz, nullptr) != DISYOLO_OK);
    CHECK(disyolo_cmdlist_run(l, 0, sz + 1, nullptr) != DISYOLO_OK);
    CHECK(disyolo_cmdlist_run(l, sz, 0, nullptr) != DISYOLO_OK || sz == 0);
    CHECK(disyolo_cmdlist_run(nullptr, 0, 0, nullptr) != DISYOLO_OK);
    disyolo_cmdlist_destroy(l);
  }
  disyolo_cmdlist_destroy(nullptr);
  printf("asan_driver: ok (crc32c, 400 contour images, 50 command lists)\n");
  return 0;
}
