// Which WRITE PATTERN of a (T, n) float32 stack reaches the streaming ceiling (round 6)?  4 bytes in, 16 out per thread and frame.
//   linear   one launch over the flat stack (tools/ubench/stream.hip: 5.4 TB/s)
//   strided  thread = quad x F frames, grid (quads / 256, T / F): how time_diff / project / blur are written
//   loop     thread = quad, loops over ALL frames (grid quads / 256)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 ex(uint32_t w) { return f32x4{(float)(w & 255), (float)((w >> 8) & 255), (float)((w >> 16) & 255), (float)(w >> 24)}; }
template <int F> __global__ __launch_bounds__(256) void strided(const uint32_t* __restrict__ a, f32x4* __restrict__ o, int nq, int T) {
  const int q = blockIdx.x * 256 + threadIdx.x; if (q >= nq) return;
  const int t0 = blockIdx.y * F; uint32_t w[F];
#pragma unroll
  for (int t = 0; t < F; ++t) if (t0 + t < T) w[t] = a[(size_t)(t0 + t) * nq + q];
#pragma unroll
  for (int t = 0; t < F; ++t) if (t0 + t < T) o[(size_t)(t0 + t) * nq + q] = ex(w[t]);
}
__global__ __launch_bounds__(256) void loopk(const uint32_t* __restrict__ a, f32x4* __restrict__ o, int nq, int T) {
  const int q = blockIdx.x * 256 + threadIdx.x; if (q >= nq) return;
  for (int t = 0; t < T; ++t) o[(size_t)t * nq + q] = ex(a[(size_t)t * nq + q]);
}
template <int F> __global__ __launch_bounds__(256) void frame_major(const uint32_t* __restrict__ a, f32x4* __restrict__ o, int nq, int T, int bx) {
  // blockIdx.x = y-major: consecutive hardware blocks walk the frames of one quad range first
  const int x = blockIdx.x / ((T + F - 1) / F), y = blockIdx.x % ((T + F - 1) / F);
  const int q = x * 256 + threadIdx.x; if (q >= nq) return;
  const int t0 = y * F; uint32_t w[F];
#pragma unroll
  for (int t = 0; t < F; ++t) if (t0 + t < T) w[t] = a[(size_t)(t0 + t) * nq + q];
#pragma unroll
  for (int t = 0; t < F; ++t) if (t0 + t < T) o[(size_t)(t0 + t) * nq + q] = ex(w[t]);
}
template <typename Fn> float timeit(Fn f, int reps) {
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  f(); f(); (void)hipDeviceSynchronize();
  (void)hipEventRecord(a); for (int i = 0; i < reps; ++i) f(); (void)hipEventRecord(b); (void)hipEventSynchronize(b);
  float ms; (void)hipEventElapsedTime(&ms, a, b); return ms / reps;
}
int main() {
  const int T = 200, nq = 1080 * 1920 / 4;
  const size_t n4 = (size_t)T * nq;
  void *din, *dout; (void)hipMalloc(&din, n4 * 4); (void)hipMalloc(&dout, n4 * 16); (void)hipMemset(din, 1, n4 * 4);
  const int bx = (nq + 255) / 256;
  float t;
#define RUN(name, ...) t = timeit([&] { __VA_ARGS__; }, 10); printf("%-28s %.3f ms  %.0f GB/s\n", name, t, n4 * 20 / t / 1e6);
  RUN("strided F=1", hipLaunchKernelGGL(strided<1>, dim3(bx, T), dim3(256), 0, 0, (const uint32_t*)din, (f32x4*)dout, nq, T));
  RUN("strided F=2", hipLaunchKernelGGL(strided<2>, dim3(bx, T / 2), dim3(256), 0, 0, (const uint32_t*)din, (f32x4*)dout, nq, T));
  RUN("strided F=8", hipLaunchKernelGGL(strided<8>, dim3(bx, T / 8), dim3(256), 0, 0, (const uint32_t*)din, (f32x4*)dout, nq, T));
  RUN("strided F=16", hipLaunchKernelGGL(strided<16>, dim3(bx, (T + 15) / 16), dim3(256), 0, 0, (const uint32_t*)din, (f32x4*)dout, nq, T));
  RUN("loop over all frames", hipLaunchKernelGGL(loopk, dim3(bx), dim3(256), 0, 0, (const uint32_t*)din, (f32x4*)dout, nq, T));
  RUN("frame-major blocks F=8", hipLaunchKernelGGL(frame_major<8>, dim3(bx * (T / 8)), dim3(256), 0, 0, (const uint32_t*)din, (f32x4*)dout, nq, T, bx));
  RUN("frame-major blocks F=1", hipLaunchKernelGGL(frame_major<1>, dim3(bx * T), dim3(256), 0, 0, (const uint32_t*)din, (f32x4*)dout, nq, T, bx));
  return 0;
}
