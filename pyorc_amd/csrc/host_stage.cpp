// Staging threads of the host-pointer entry points (see host_stage.h).  Plain C++; the AVX2 paths are selected at run time.
#include "host_stage.h"

#include <immintrin.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <cmath>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace lspiv_host {

int stage_threads() {
  static const int n = [] {
    if (const char* e = getenv("LSPIV_STAGE_THREADS")) return std::max(1, atoi(e));
    const unsigned hw = std::thread::hardware_concurrency();
    return (int)std::max(1u, std::min(16u, hw ? hw : 4u));
  }();
  return n;
}

namespace {

// persistent workers: a batch is a few hundred microseconds of copying, starting 15 threads for each costs as much
class Pool {
 public:
  static Pool& get() {
    static Pool* p = new Pool(stage_threads() - 1);   // never destroyed: its workers sleep on the condition variable until the process ends
    return *p;
  }
  // fn(part) for part = 0 .. n_parts - 1, the caller takes its share; returns when all parts are done.  One batch at a time.
  void run(int n_parts, const std::function<void(int)>& fn) {
    if (n_parts <= 1 || workers_.empty()) {
      for (int i = 0; i < n_parts; ++i) fn(i);
      return;
    }
    std::unique_lock<std::mutex> batch(batch_mu_);
    {
      std::lock_guard<std::mutex> lk(mu_);
      fn_ = &fn; next_ = 0; n_parts_ = n_parts; pending_ = n_parts; ++generation_;
    }
    cv_.notify_all();
    work();
    std::unique_lock<std::mutex> lk(mu_);
    done_cv_.wait(lk, [&] { return pending_ == 0; });
    fn_ = nullptr;
  }

 private:
  explicit Pool(int n) {
    for (int i = 0; i < n; ++i) workers_.emplace_back([this] { loop(); });
    for (auto& t : workers_) t.detach();   // they sleep on the condition variable for the life of the process
  }
  void work() {
    for (;;) {
      int part;
      const std::function<void(int)>* fn;
      {
        std::lock_guard<std::mutex> lk(mu_);
        if (!fn_ || next_ >= n_parts_) return;
        part = next_++;
        fn = fn_;
      }
      (*fn)(part);
      {
        std::lock_guard<std::mutex> lk(mu_);
        if (--pending_ == 0) done_cv_.notify_all();
      }
    }
  }
  void loop() {
    uint64_t seen = 0;
    for (;;) {
      {
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [&] { return generation_ != seen; });
        seen = generation_;
      }
      work();
    }
  }
  std::vector<std::thread> workers_;
  std::mutex mu_, batch_mu_;
  std::condition_variable cv_, done_cv_;
  const std::function<void(int)>* fn_ = nullptr;
  int next_ = 0, n_parts_ = 0, pending_ = 0;
  uint64_t generation_ = 0;
};

const bool kAvx2 = __builtin_cpu_supports("avx2");

__attribute__((target("avx2"))) void copy_stream_avx2(char* dst, const char* src, size_t n) {
  size_t i = 0;
  for (; i < n && ((uintptr_t)(dst + i) & 31); ++i) dst[i] = src[i];
  for (; i + 128 <= n; i += 128) {
    const __m256i a = _mm256_loadu_si256((const __m256i*)(src + i)), b = _mm256_loadu_si256((const __m256i*)(src + i + 32));
    const __m256i c = _mm256_loadu_si256((const __m256i*)(src + i + 64)), d = _mm256_loadu_si256((const __m256i*)(src + i + 96));
    _mm256_stream_si256((__m256i*)(dst + i), a); _mm256_stream_si256((__m256i*)(dst + i + 32), b);
    _mm256_stream_si256((__m256i*)(dst + i + 64), c); _mm256_stream_si256((__m256i*)(dst + i + 96), d);
  }
  for (; i < n; ++i) dst[i] = src[i];
  _mm_sfence();
}

__attribute__((target("avx2"))) void narrow_avx2(float* dst, const double* src, size_t n, double c) {
  const __m256d vc = _mm256_set1_pd(c);
  size_t i = 0;
  for (; i < n && ((uintptr_t)(dst + i) & 31); ++i) dst[i] = (float)(src[i] - c);
  for (; i + 8 <= n; i += 8) {
    const __m128 lo = _mm256_cvtpd_ps(_mm256_sub_pd(_mm256_loadu_pd(src + i), vc));
    const __m128 hi = _mm256_cvtpd_ps(_mm256_sub_pd(_mm256_loadu_pd(src + i + 4), vc));
    _mm256_stream_ps(dst + i, _mm256_set_m128(hi, lo));
  }
  for (; i < n; ++i) dst[i] = (float)(src[i] - c);
  _mm_sfence();
}

void narrow_plain(float* dst, const double* src, size_t n, double c) {
  for (size_t i = 0; i < n; ++i) dst[i] = (float)(src[i] - c);
}

}  // namespace

void staged_copy(void* dst, const void* src, size_t bytes) {
  const int nthreads = stage_threads();
  if (nthreads <= 1 || bytes < ((size_t)4 << 20)) {
    memcpy(dst, src, bytes);
    return;
  }
  const size_t part = ((bytes / nthreads) + 4095) & ~(size_t)4095;
  const int n_parts = (int)((bytes + part - 1) / part);
  Pool::get().run(n_parts, [&](int i) {
    const size_t off = part * (size_t)i, len = std::min(bytes, off + part) - off;
    if (kAvx2) copy_stream_avx2((char*)dst + off, (const char*)src + off, len);
    else memcpy((char*)dst + off, (const char*)src + off, len);
  });
}

void staged_narrow(float* dst, const double* src, size_t frame_elems, size_t n_frames, const double* offsets) {
  const size_t n = frame_elems * n_frames;
  const int nthreads = stage_threads();
  // parts never straddle a frame (each frame has its own offset): a frame is cut into `per` pieces of whole 1024-sample runs
  const int per = (int)std::max<size_t>(1, std::min<size_t>((size_t)nthreads, n < ((size_t)1 << 19) ? 1 : ((size_t)nthreads + n_frames - 1) / n_frames));
  const size_t piece = ((frame_elems + per - 1) / per + 1023) & ~(size_t)1023;
  Pool::get().run((int)n_frames * per, [&](int i) {
    const size_t f = (size_t)i / per, k = (size_t)i % per;
    const size_t a = std::min(frame_elems, piece * k), b = std::min(frame_elems, piece * (k + 1));
    if (b <= a) return;
    const double c = offsets ? offsets[f] : 0.0;
    if (kAvx2) narrow_avx2(dst + f * frame_elems + a, src + f * frame_elems + a, b - a, c);
    else narrow_plain(dst + f * frame_elems + a, src + f * frame_elems + a, b - a, c);
  });
}

double frame_offset(const double* frame, size_t frame_elems, double min_abs) {
  if (frame_elems == 0 || !(min_abs >= 0.0)) return 0.0;
  const size_t n = std::min<size_t>(4096, frame_elems), step = frame_elems / n;
  double s = 0.0;
  for (size_t i = 0; i < n; ++i) s += frame[i * step];
  const double c = std::nearbyint(s / (double)n);
  return (std::isfinite(c) && std::fabs(c) >= min_abs) ? c : 0.0;
}

}  // namespace lspiv_host
