// Host side of the streaming extractor (sidekit_amd/pipeline.py): wav files -> rows of a pinned int16 staging buffer, by a
// pool of host threads that never hold the Python interpreter lock.  Replaces the per-file `torchaudio.load` of the
// reference driver (sidekit/bin/extract_xvectors.py:57-70) for the canonical case -- RIFF/WAVE, PCM 16-bit, mono; anything
// else is reported as "not fast-path" and decoded by the Python side.  The GPU converts int16 -> float32 / 32768 (exact).
#include <atomic>
#include <fcntl.h>
#include <stdint.h>
#include <string.h>
#include <sys/stat.h>
#include <thread>
#include <unistd.h>
#include <vector>

#include "../../include/sidekit_amd.h"

namespace {

inline uint32_t rd32(const unsigned char* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
inline uint16_t rd16(const unsigned char* p) { return (uint16_t)(p[0] | (p[1] << 8)); }

// header walk of one file: 1 = PCM16 mono (nsamples / rate / data_offset filled), 0 = some other wav or unreadable header
int probe_one(const char* path, int32_t* nsamples, int32_t* rate, int64_t* data_offset) {
  *nsamples = 0; *rate = 0; *data_offset = 0;
  const int fd = open(path, O_RDONLY);
  if (fd < 0) return -1;
  struct stat st;
  if (fstat(fd, &st) != 0) { close(fd); return -1; }
  unsigned char buf[4096];
  const ssize_t got = pread(fd, buf, sizeof(buf), 0);
  close(fd);
  if (got < 44 || memcmp(buf, "RIFF", 4) != 0 || memcmp(buf + 8, "WAVE", 4) != 0) return 0;
  int64_t pos = 12;
  bool have_fmt = false;
  uint16_t tag = 0, channels = 0, bits = 0;
  uint32_t sr = 0;
  while (pos + 8 <= got) {
    const uint32_t size = rd32(buf + pos + 4);
    const int64_t body = pos + 8;
    if (memcmp(buf + pos, "fmt ", 4) == 0 && body + 16 <= got) {
      tag = rd16(buf + body); channels = rd16(buf + body + 2); sr = rd32(buf + body + 4); bits = rd16(buf + body + 14);
      have_fmt = true;
    } else if (memcmp(buf + pos, "data", 4) == 0) {
      if (!have_fmt || tag != 1 || channels != 1 || bits != 16) return 0;
      int64_t bytes = (int64_t)size;
      if (body + bytes > (int64_t)st.st_size) bytes = (int64_t)st.st_size - body;   // streamed headers carry 0xFFFFFFFF
      if (bytes < 0 || bytes / 2 > INT32_MAX) return 0;
      *nsamples = (int32_t)(bytes / 2); *rate = (int32_t)sr; *data_offset = body;
      return 1;
    }
    pos = body + (int64_t)size + (size & 1);
  }
  return 0;   // data chunk beyond the first 4 KB: leave it to the generic reader
}

template <class F>
void parallel_for(int32_t n, int32_t threads, F&& fn) {
  if (threads < 1) threads = 1;
  if (threads > n) threads = n;
  if (threads <= 1) { for (int32_t i = 0; i < n; ++i) fn(i); return; }
  std::atomic<int32_t> next(0);
  std::vector<std::thread> pool;
  pool.reserve((size_t)threads);
  for (int32_t t = 0; t < threads; ++t)
    pool.emplace_back([&]() { for (int32_t i = next.fetch_add(1); i < n; i = next.fetch_add(1)) fn(i); });
  for (auto& th : pool) th.join();
}

}  // namespace

extern "C" {

int sk_wav_probe(const char* const* paths, int32_t n, int32_t threads, int32_t* nsamples, int32_t* rate, int64_t* data_offset,
                 int32_t* kind) {
  if (!paths || n < 0 || !nsamples || !rate || !data_offset || !kind) return SK_EARG;
  parallel_for(n, threads, [&](int32_t i) { kind[i] = probe_one(paths[i], nsamples + i, rate + i, data_offset + i); });
  return SK_OK;
}

int sk_wav_read_pcm16(const char* const* paths, const int64_t* data_offset, const int32_t* nsamples, const int32_t* row, int32_t n,
                      int32_t threads, int16_t* dst, int64_t ld, int32_t n_rows, int32_t* status) {
  if (!paths || !data_offset || !nsamples || !row || n < 0 || !dst || ld <= 0 || n_rows <= 0 || !status) return SK_EARG;
  parallel_for(n, threads, [&](int32_t i) {
    status[i] = -1;
    if (nsamples[i] < 0 || nsamples[i] > ld || row[i] < 0 || row[i] >= n_rows) return;
    const int fd = open(paths[i], O_RDONLY);
    if (fd < 0) return;
    unsigned char* out = reinterpret_cast<unsigned char*>(dst + (int64_t)row[i] * ld);
    int64_t want = (int64_t)nsamples[i] * 2, done = 0;
    while (done < want) {
      const ssize_t r = pread(fd, out + done, (size_t)(want - done), data_offset[i] + done);
      if (r <= 0) break;
      done += r;
    }
    close(fd);
    status[i] = done == want ? 0 : -1;   // little-endian host: PCM16 bytes are the int16 values
  });
  return SK_OK;
}

}  // extern "C"
