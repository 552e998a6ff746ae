// Element-wise pre-processing filters on frame stacks in HBM: the ones the reference
// implements with plain numpy/xarray arithmetic, so they can be reproduced bit for bit --
//   Frames.time_diff  pyorc/api/frames.py:409-436    Frames.minmax  :344-362    Frames.normalize  :279-306
// (edge_detect / smooth are cv2.GaussianBlur calls and are not covered).  All are HBM-bound streaming kernels:
// 16-byte accesses per lane, grid-stride over the stack.
#include "common.h"

namespace lspiv {

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <typename T>
__global__ __launch_bounds__(256) void time_diff_kernel(const T* __restrict__ f, int64_t frame_elems, int64_t n_out,
                                                        float thres, int use_abs, float* __restrict__ out) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_out; i += stride) {
    float d = to_f32(f[i + frame_elems]) - to_f32(f[i]);  // float32 difference of consecutive frames
    d = (d > thres) ? d : 0.0f;                           // .where(d > thres) then .fillna(0.0): NaN compares false
    out[i] = use_abs ? fabsf(d) : d;
  }
}

__global__ __launch_bounds__(256) void minmax_kernel(const float* __restrict__ in, int64_t n, float lo, float hi,
                                                     float* __restrict__ out) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const float x = in[i];
    // np.maximum(np.minimum(x, hi), lo): NaN propagates (numpy), unlike fminf / fmaxf
    const float a = (x != x) ? x : (x < hi ? x : hi);
    out[i] = (a != a) ? a : (a > lo ? a : lo);
  }
}

// normalize, pass 1: float32 mean over the sampled frames (exact integer sums for uint8)
__global__ __launch_bounds__(256) void sample_mean_kernel(const uint8_t* __restrict__ f, int64_t frame_elems, int n_frames,
                                                          int interval, float* __restrict__ mean) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= frame_elems) return;
  uint32_t s = 0, n = 0;
  for (int t = 0; t < n_frames; t += interval) { s += f[(int64_t)t * frame_elems + i]; ++n; }
  mean[i] = (float)((double)s / (double)n);  // numpy: float64 mean, then astype(float32)
}

// pass 2: per-frame min / max of (x - mean); one block per (frame, slice), combined with float atomics on the
// order-preserving integer image of the floats
__device__ __forceinline__ int f2ord(float x) { int i = __builtin_bit_cast(int, x); return i >= 0 ? i : i ^ 0x7fffffff; }
__device__ __forceinline__ float ord2f(int i) { return __builtin_bit_cast(float, i >= 0 ? i : i ^ 0x7fffffff); }

__global__ __launch_bounds__(256) void frame_minmax_kernel(const uint8_t* __restrict__ f, const float* __restrict__ mean,
                                                           int64_t frame_elems, int* __restrict__ mn, int* __restrict__ mx) {
  const int t = blockIdx.y;
  const uint8_t* img = f + (int64_t)t * frame_elems;
  float lo = 3.0e38f, hi = -3.0e38f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < frame_elems; i += (int64_t)gridDim.x * blockDim.x) {
    const float d = (float)img[i] - mean[i];
    lo = fminf(lo, d);
    hi = fmaxf(hi, d);
  }
  lo = -half_max(-lo); hi = half_max(hi);
  lo = fminf(lo, __shfl_xor(lo, 32, 64)); hi = fmaxf(hi, __shfl_xor(hi, 32, 64));
  if ((threadIdx.x & 63) == 0) { atomicMin(&mn[t], f2ord(lo)); atomicMax(&mx[t], f2ord(hi)); }
}

// pass 3: ((x - mean) - min) / (max - min) * 255 -> uint8 (truncation, NaN -> 0), all float32 like numpy
__global__ __launch_bounds__(256) void normalize_kernel(const uint8_t* __restrict__ f, const float* __restrict__ mean,
                                                        int64_t frame_elems, const int* __restrict__ mn,
                                                        const int* __restrict__ mx, uint8_t* __restrict__ out) {
  const int t = blockIdx.y;
  const float lo = ord2f(mn[t]), span = ord2f(mx[t]) - lo;
  const uint8_t* img = f + (int64_t)t * frame_elems;
  uint8_t* dst = out + (int64_t)t * frame_elems;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < frame_elems; i += (int64_t)gridDim.x * blockDim.x) {
    const float q = (((float)img[i] - mean[i]) - lo) / span * 255.0f;
    dst[i] = (q != q) ? (uint8_t)0 : (uint8_t)(int)q;
  }
}

hipError_t launch_time_diff(const void* frames, int dtype, int64_t frame_elems, int64_t n_frames, float thres, int use_abs,
                            float* out, hipStream_t s) {
  const int64_t n_out = (n_frames - 1) * frame_elems;
  if (n_out <= 0) return hipSuccess;
  const unsigned blocks = (unsigned)std::min<int64_t>((n_out + 255) / 256, 256 * 16);
  switch (dtype) {
    case 0: hipLaunchKernelGGL(time_diff_kernel<uint8_t>, dim3(blocks), dim3(256), 0, s, (const uint8_t*)frames, frame_elems, n_out, thres, use_abs, out); break;
    case 1: hipLaunchKernelGGL(time_diff_kernel<float>, dim3(blocks), dim3(256), 0, s, (const float*)frames, frame_elems, n_out, thres, use_abs, out); break;
    case 2: hipLaunchKernelGGL(time_diff_kernel<double>, dim3(blocks), dim3(256), 0, s, (const double*)frames, frame_elems, n_out, thres, use_abs, out); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

hipError_t launch_minmax(const float* in, int64_t n, float lo, float hi, float* out, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  const unsigned blocks = (unsigned)std::min<int64_t>((n + 255) / 256, 256 * 16);
  hipLaunchKernelGGL(minmax_kernel, dim3(blocks), dim3(256), 0, s, in, n, lo, hi, out);
  return hipGetLastError();
}

hipError_t launch_normalize(const uint8_t* frames, int64_t frame_elems, int n_frames, int interval, float* d_mean,
                            int* d_mn, int* d_mx, uint8_t* out, hipStream_t s) {
  if (n_frames <= 0 || frame_elems <= 0) return hipSuccess;
  hipLaunchKernelGGL(sample_mean_kernel, dim3((unsigned)((frame_elems + 255) / 256)), dim3(256), 0, s, frames, frame_elems,
                     n_frames, interval, d_mean);
  hipError_t e = hipMemsetAsync(d_mn, 0x7f, (size_t)n_frames * sizeof(int), s);   // 0x7f7f7f7f: a huge positive float
  if (e != hipSuccess) return e;
  e = hipMemsetAsync(d_mx, 0x80, (size_t)n_frames * sizeof(int), s);               // 0x80808080: a negative ordinal
  if (e != hipSuccess) return e;
  const unsigned bx = (unsigned)std::min<int64_t>((frame_elems + 255) / 256, 64);
  hipLaunchKernelGGL(frame_minmax_kernel, dim3(bx, n_frames), dim3(256), 0, s, frames, d_mean, frame_elems, d_mn, d_mx);
  hipLaunchKernelGGL(normalize_kernel, dim3(bx * 4, n_frames), dim3(256), 0, s, frames, d_mean, frame_elems, d_mn, d_mx, out);
  return hipGetLastError();
}

}  // namespace lspiv
