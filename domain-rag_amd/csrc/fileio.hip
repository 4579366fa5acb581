// fileio.hip — host-side batch file reader for the GPU JPEG route (no device code).
//
// The GPU decodes ~20-60 k corpus files per second (csrc/jpeg.hip); feeding it from Python costs ~80 us of interpreter time per
// file under the GIL (open / readinto / close), i.e. ~8 k files/s whatever the thread count.  These two entry points do the
// per-file system calls from native threads: sizes first (the caller lays the blob out), then every file straight into its
// slot of the caller's pinned staging buffer.  Replaces the `Image.open(path)` file access of the reference's corpus loop
// (retrieval/clip100_resnet_style_all_shots.py:270-281) for the route that decodes on the device.
#include "drag_common.h"
#include <errno.h>
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>
#include <atomic>
#include <thread>
#include <vector>

namespace {

template <typename F>
void parallel_for(int64_t n, int threads, F f) {
  if (threads < 1) threads = 1;
  if (threads > 256) threads = 256;
  if (n < threads) threads = (int)(n > 0 ? n : 1);
  std::atomic<int64_t> next{0};
  auto work = [&]() {
    for (;;) {
      const int64_t i = next.fetch_add(1, std::memory_order_relaxed);
      if (i >= n) break;
      f(i);
    }
  };
  std::vector<std::thread> pool;
  for (int t = 1; t < threads; ++t) pool.emplace_back(work);
  work();
  for (auto& t : pool) t.join();
}

}  // namespace

// sizes[i] = size of paths[i] in bytes, or -errno when it cannot be stat'ed / is not a regular file
extern "C" int drag_file_sizes(const char* const* paths, int64_t n, int64_t* sizes, int32_t threads) {
  DRAG_CHECK(paths && sizes && n >= 0, "drag_file_sizes: bad arguments");
  parallel_for(n, threads, [&](int64_t i) {
    struct stat st;
    if (paths[i] == nullptr || stat(paths[i], &st) != 0) sizes[i] = -(int64_t)(paths[i] ? errno : EINVAL);
    else if (!S_ISREG(st.st_mode)) sizes[i] = -(int64_t)EISDIR;
    else sizes[i] = (int64_t)st.st_size;
  });
  return 0;
}

// reads paths[i] into dst[offsets[i] .. offsets[i + 1]) (exactly that many bytes: the size the caller saw).  status[i] = 0, or
// errno / -1 for a short file; an empty slot (offsets[i] == offsets[i + 1]) is skipped.  `dst` is plain host memory.
extern "C" int drag_read_files(const char* const* paths, int64_t n, void* dst, const int64_t* offsets, int32_t* status,
                               int32_t threads) {
  DRAG_CHECK(paths && dst && offsets && status && n >= 0, "drag_read_files: bad arguments");
  uint8_t* base = (uint8_t*)dst;
  parallel_for(n, threads, [&](int64_t i) {
    const int64_t want = offsets[i + 1] - offsets[i];
    status[i] = 0;
    if (want <= 0) return;
    const int fd = open(paths[i], O_RDONLY | O_CLOEXEC);
    if (fd < 0) { status[i] = errno; return; }
    int64_t got = 0;
    while (got < want) {
      const ssize_t k = read(fd, base + offsets[i] + got, (size_t)(want - got));
      if (k < 0) { if (errno == EINTR) continue; status[i] = errno; break; }
      if (k == 0) { status[i] = -1; break; }          // the file shrank since it was stat'ed
      got += k;
    }
    close(fd);
  });
  return 0;
}
