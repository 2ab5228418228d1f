// tests/hostcopy/hostcopy_check.cpp — every byte of every size must arrive, nothing beyond may be touched
// (edgegraph3d_amd/csrc/eg3d_host_copy.h). Sizes sweep the pattern that exposed the round-2 bug:
// bytes / threads a multiple of 64 with a non-zero remainder.
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "eg3d_host_copy.h"

int main() {
  const size_t kMin = 1024;  // small per-thread minimum so that small blocks take the threaded path
  std::vector<unsigned char> src(1 << 20), dst(src.size() + 64);
  for (size_t i = 0; i < src.size(); i++) src[i] = (unsigned char)(i * 2654435761u >> 24) | 1;
  size_t cases = 0;
  for (int nt : {2, 3, 7, 8, 16, 31, 64})
    for (size_t base : {(size_t)4096, (size_t)65536, (size_t)200000, (size_t)(1 << 20) - 200}) {
      for (size_t extra = 0; extra < 130; extra++) {
        const size_t bytes = base + extra;
        if (bytes > src.size()) continue;
        std::fill(dst.begin(), dst.end(), 0);
        eg3d::copy_mt(dst.data(), src.data(), bytes, nt, kMin);
        for (size_t i = 0; i < bytes; i++)
          if (dst[i] != src[i]) {
            std::printf("FAIL: %zu bytes on %d threads: byte %zu not copied\n", bytes, nt, i);
            return 1;
          }
        for (size_t i = bytes; i < bytes + 64; i++)
          if (dst[i] != 0) {
            std::printf("FAIL: %zu bytes on %d threads: byte %zu beyond the block written\n", bytes, nt, i);
            return 1;
          }
        cases++;
      }
      // the exact shape of the bug: bytes = nt * 64 * k + r, 0 < r < nt
      for (size_t r = 1; r < (size_t)nt && r < 20; r++) {
        const size_t bytes = (size_t)nt * 64 * 37 + r;
        std::fill(dst.begin(), dst.end(), 0);
        eg3d::copy_mt(dst.data(), src.data(), bytes, nt, kMin);
        for (size_t i = 0; i < bytes; i++)
          if (dst[i] != src[i]) {
            std::printf("FAIL: %zu bytes on %d threads: byte %zu not copied\n", bytes, nt, i);
            return 1;
          }
        cases++;
      }
    }
  std::printf("HOSTCOPY-OK %zu cases\n", cases);
  return 0;
}
