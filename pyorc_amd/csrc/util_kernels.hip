// Small helper kernels: ensemble mean plane, synthetic particle-image generator.
#include "common.h"

namespace lspiv {

// count filter + mean plane of the ensemble branch (pyorc/velocimetry/ffpiv.py:280-282):
// windows with corr_count < count_min * n_frames become NaN, otherwise corr_sum / corr_count
// (0/0 = NaN exactly like numpy's divide).
__global__ void ensemble_mean_kernel(const float* sum, const float* count, float min_count, int plane_elems,
                                     float* mean) {
  const uint32_t w = blockIdx.x;
  const float c = count[w];
  const bool low = c < min_count;
  for (int o = threadIdx.x; o < plane_elems; o += blockDim.x) {
    const size_t i = (size_t)w * plane_elems + o;
    mean[i] = low ? __builtin_nanf("") : sum[i] / c;
  }
}

// walking ensemble kernels: corr_sum += part[0] + part[1] + ... in segment (= time) order, counts likewise
// LANE_MAJOR (64 x 64): the slots hold element (row y, column x) at (x / 4) * 4 N + y * 4 + x % 4 (piv_fft_impl.h, slot_accumulate);
// a thread owns one slot element (coalesced reads of every segment's slot) and writes it to the fft-shifted row-major position
// of corr_sum once
template <bool LANE_MAJOR>
__global__ __launch_bounds__(256) void ensemble_merge_kernel(const float* __restrict__ part_sum, const float* __restrict__ part_cnt,
                                                             uint32_t n_seg, int64_t n_elems, uint32_t n_win, int n,
                                                             float* __restrict__ corr_sum, float* __restrict__ corr_count, int split) {
  // four consecutive elements per thread (the planes of the even window sizes hold a multiple of four samples): 16-byte loads of
  // every segment's slot
  const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i < n_elems) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    if (LANE_MAJOR) {
      // slot element e = q * 4 n + y * 4 + (x % 4), x = 4 q + (e % 4): the four elements of a thread are columns x .. x + 3 of row y
      const int nn = n * n, h = n / 2;
      const int64_t win = i / nn;
      const int e = (int)(i - win * nn), q = e / (4 * n), y = (e - q * 4 * n) >> 2, x = 4 * q;
      const int ip = y + h >= n ? y + h - n : y + h, jp = x + h >= n ? x + h - n : x + h;   // (x + 3 stays in the same half: h is a multiple of 4)
      float* dst = corr_sum + win * nn + ip * n + jp;
      f4 acc = *reinterpret_cast<const f4*>(dst);
      if (split) {
        // slot (segment, window) = two entries of nn / 2 floats: the cold array (elements below nn / 2), then the hot array
        const int half = nn / 2, eh = e >= half ? e - half : e;
        const float* arr = part_sum + (e >= half ? (int64_t)n_seg * n_win * half : 0) + win * half + eh;
        for (uint32_t sg = 0; sg < n_seg; ++sg) acc += *reinterpret_cast<const f4*>(arr + (int64_t)sg * n_win * half);
      } else {
        for (uint32_t sg = 0; sg < n_seg; ++sg) acc += *reinterpret_cast<const f4*>(part_sum + (int64_t)sg * n_elems + i);
      }
      *reinterpret_cast<f4*>(dst) = acc;
    } else {
      f4 acc = *reinterpret_cast<const f4*>(corr_sum + i);
      for (uint32_t sg = 0; sg < n_seg; ++sg) acc += *reinterpret_cast<const f4*>(part_sum + (int64_t)sg * n_elems + i);
      *reinterpret_cast<f4*>(corr_sum + i) = acc;
    }
  }
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t < n_win) {
    float c = corr_count[t];
    for (uint32_t sg = 0; sg < n_seg; ++sg) c += part_cnt[(int64_t)sg * n_win + t];
    corr_count[t] = c;
  }
}

hipError_t launch_ensemble_merge(const float* part_sum, const float* part_cnt, uint32_t n_seg, uint32_t n_win, int plane_elems,
                                 float* corr_sum, float* corr_count, hipStream_t s, int lane_major_n, bool split_halves) {
  const int64_t n_elems = (int64_t)n_win * plane_elems;
  if (n_elems == 0) return hipSuccess;
  if (plane_elems % 4 != 0) return hipErrorInvalidValue;   // (walking kernels: even window sizes only)
  const dim3 grid((unsigned)std::max<int64_t>((n_elems / 4 + 255) / 256, ((int64_t)n_win + 255) / 256));
  if (lane_major_n)
    hipLaunchKernelGGL(ensemble_merge_kernel<true>, grid, dim3(256), 0, s, part_sum, part_cnt, n_seg, n_elems, n_win, lane_major_n, corr_sum, corr_count, split_halves ? 1 : 0);
  else
    hipLaunchKernelGGL(ensemble_merge_kernel<false>, grid, dim3(256), 0, s, part_sum, part_cnt, n_seg, n_elems, n_win, 0, corr_sum, corr_count, 0);
  return hipGetLastError();
}

hipError_t launch_ensemble_mean(const float* sum, const float* count, float min_count, uint32_t n_win,
                                int plane_elems, float* mean, hipStream_t s) {
  if (n_win == 0) return hipSuccess;
  hipLaunchKernelGGL(ensemble_mean_kernel, dim3(n_win), dim3(256), 0, s, sum, count, min_count, plane_elems, mean);
  return hipGetLastError();
}

// ---- synthetic particle images ----------------------------------------------------------------
// Same model as pyorc_amd/synth.py (different random stream): Gaussian blobs sigma = 1.2 px,
// peak U(120,255), flow u = 3 + 2 sin(2 pi y/H), v = 1.5 cos(2 pi x/W) px/frame, re-seeded when a
// particle leaves.  Rendering accumulates 24.8 fixed point with integer atomics, so a stack is
// bit-reproducible for a given seed.
struct Particle {
  float x, y, amp;
  uint32_t gen;
};

__device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
__device__ __forceinline__ float u01(uint64_t h) { return (float)(h >> 40) * (1.0f / 16777216.0f); }

__device__ __forceinline__ void seed_particle(Particle& p, uint64_t seed, uint32_t i, int H, int W) {
  const uint64_t h = splitmix64(seed ^ ((uint64_t)i << 32 | p.gen));
  p.x = u01(h) * W;
  p.y = u01(splitmix64(h)) * H;
  p.amp = 120.0f + 135.0f * u01(splitmix64(h ^ 0x5bd1e995u));
}

__global__ void synth_init_kernel(Particle* ps, uint32_t n, uint64_t seed, int H, int W) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Particle p;
  p.gen = 0;
  seed_particle(p, seed, i, H, W);
  ps[i] = p;
}

__global__ void synth_render_kernel(Particle* ps, uint32_t n, uint32_t* acc, uint64_t seed, int H, int W) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Particle p = ps[i];
  const int iy = (int)rintf(p.y), ix = (int)rintf(p.x);
  const float k = -1.0f / (2.0f * 1.2f * 1.2f);
  for (int dy = -3; dy <= 3; ++dy) {
    const int yy = iy + dy;
    if (yy < 0 || yy >= H) continue;
    for (int dx = -3; dx <= 3; ++dx) {
      const int xx = ix + dx;
      if (xx < 0 || xx >= W) continue;
      const float fy = (float)yy - p.y, fx = (float)xx - p.x;
      const float w = p.amp * __expf((fy * fy + fx * fx) * k);
      atomicAdd(&acc[(size_t)yy * W + xx], (uint32_t)(w * 256.0f + 0.5f));
    }
  }
  // advect for the next frame, re-seed when the blob has left the frame
  const float u = 3.0f + 2.0f * __sinf(6.283185307f * p.y / (float)H);
  const float v = 1.5f * __cosf(6.283185307f * p.x / (float)W);
  p.x += u;
  p.y += v;
  if (p.x < -3.0f || p.x >= W + 3.0f || p.y < -3.0f || p.y >= H + 3.0f) {
    p.gen += 1;
    seed_particle(p, seed, i, H, W);
  }
  ps[i] = p;
}

__global__ void synth_finalize_kernel(uint32_t* acc, uint8_t* frame, uint32_t n) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t v = (acc[i] + 128u) >> 8;
  frame[i] = (uint8_t)(v > 255u ? 255u : v);
  acc[i] = 0;
}

hipError_t launch_synth_particles(uint8_t* d_frames, int64_t T, int H, int W, uint64_t seed, float density,
                                  hipStream_t s) {
  const uint32_t n_px = (uint32_t)H * (uint32_t)W;
  uint32_t n_p = (uint32_t)(density * (float)H * (float)W);
  if (n_p < 1) n_p = 1;
  Particle* ps = nullptr;
  uint32_t* acc = nullptr;
  hipError_t e = hipMalloc((void**)&ps, (size_t)n_p * sizeof(Particle));
  if (e != hipSuccess) return e;
  e = hipMalloc((void**)&acc, (size_t)n_px * sizeof(uint32_t));
  if (e != hipSuccess) { hipFree(ps); return e; }
  hipMemsetAsync(acc, 0, (size_t)n_px * sizeof(uint32_t), s);
  hipLaunchKernelGGL(synth_init_kernel, dim3((n_p + 255) / 256), dim3(256), 0, s, ps, n_p, seed, H, W);
  for (int64_t t = 0; t < T; ++t) {
    hipLaunchKernelGGL(synth_render_kernel, dim3((n_p + 255) / 256), dim3(256), 0, s, ps, n_p, acc, seed, H, W);
    hipLaunchKernelGGL(synth_finalize_kernel, dim3((n_px + 255) / 256), dim3(256), 0, s, acc,
                       d_frames + (size_t)t * n_px, n_px);
  }
  e = hipStreamSynchronize(s);
  hipFree(ps);
  hipFree(acc);
  return e != hipSuccess ? e : hipGetLastError();
}

__global__ __launch_bounds__(256) void negate_kernel(float* x, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) x[i] = -x[i];
}
hipError_t launch_negate(float* x, int64_t n, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(negate_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, x, n);
  return hipGetLastError();
}

}  // namespace lspiv
