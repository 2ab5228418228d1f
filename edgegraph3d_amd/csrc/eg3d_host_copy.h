// eg3d_host_copy.h — host memcpy of a large block on a few threads (the D2H copy of a cloud lands in pinned
// staging at PCIe speed; one core copying it on to the caller's arrays would be the slow part). Host only; its
// own header so that tests/hostcopy can exercise it without a GPU: a round-2 version divided the block with a
// truncating division and left up to (threads - 1) trailing bytes uncopied whenever bytes / threads happened to be
// a multiple of 64 — found by comparing a whole 8192-seed C4 step with the oracle (three observations of zeros).
#ifndef EG3D_HOST_COPY_H_
#define EG3D_HOST_COPY_H_
#include <algorithm>
#include <cstddef>
#include <cstring>
#include <thread>
#include <vector>

namespace eg3d {

#ifndef EG3D_COPY_THREADS
#define EG3D_COPY_THREADS 16
#endif

inline void copy_mt(void* dst, const void* src, size_t bytes, int max_threads = EG3D_COPY_THREADS,
                    size_t min_per_thread = (size_t)4 << 20) {
  if (bytes < 2 * min_per_thread || max_threads < 2) {
    memcpy(dst, src, bytes);
    return;
  }
  const int nt = (int)std::min<size_t>((size_t)max_threads, bytes / min_per_thread);
  const size_t per = ((bytes + (size_t)nt - 1) / (size_t)nt + 63) & ~(size_t)63;  // ceil: nt * per covers every byte
  // a thread that cannot be created (std::system_error: resource limits) leaves its share, and everything after
  // it, to the calling thread — the copy completes either way and no exception leaves the C ABI above
  std::vector<std::thread> th;
  size_t done = 0;
  try {
    th.reserve((size_t)nt);
    for (int t = 0; t < nt; t++) {
      const size_t a = (size_t)t * per, b = std::min(bytes, a + per);
      if (a >= b) break;
      th.emplace_back([=] { memcpy((char*)dst + a, (const char*)src + a, b - a); });
      done = b;
    }
  } catch (...) {
  }
  if (done < bytes) memcpy((char*)dst + done, (const char*)src + done, bytes - done);
  for (auto& t : th) t.join();
}

}  // namespace eg3d
#endif  // EG3D_HOST_COPY_H_
