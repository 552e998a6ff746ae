// Streaming ceilings of the MI355X for the HBM-bound rows (round 6): write-only, read-only and u8 -> f32 expansion (the traffic mix of
// Frames.time_diff / project), with plain and non-temporal stores.  hipcc --offload-arch=gfx950 -O3 stream.hip -o stream && ./stream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
template <bool NT> __global__ __launch_bounds__(256) void wr(f32x4* __restrict__ o, size_t n) {
  for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += gridDim.x * 256ull) {
    f32x4 v = {(float)i, 1.f, 2.f, 3.f};
    if (NT) __builtin_nontemporal_store(v, o + i); else o[i] = v;
  }
}
__global__ __launch_bounds__(256) void rd(const u32x4* __restrict__ a, size_t n, uint32_t* __restrict__ sink) {
  uint32_t acc = 0;
  for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += gridDim.x * 256ull) { u32x4 v = a[i]; acc += v[0] ^ v[1] ^ v[2] ^ v[3]; }
  if (acc == 0x12345678u) *sink = acc;
}
template <bool NT> __global__ __launch_bounds__(256) void expand(const uint32_t* __restrict__ a, f32x4* __restrict__ o, size_t n) {   // 4 bytes in, 16 out
  for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += gridDim.x * 256ull) {
    const uint32_t w = a[i];
    f32x4 v = {(float)(w & 255), (float)((w >> 8) & 255), (float)((w >> 16) & 255), (float)(w >> 24)};
    if (NT) __builtin_nontemporal_store(v, o + i); else o[i] = v;
  }
}
template <typename F> float timeit(F f, int reps) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  f(); f(); hipDeviceSynchronize();
  hipEventRecord(a); for (int i = 0; i < reps; ++i) f(); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); return ms / reps;
}
int main() {
  const size_t n4 = (size_t)200 * 1080 * 1920 / 4;     // quads of a 200-frame 1080p float32 stack: 1.66 GB out, 0.41 GB in
  void *din, *dout; uint32_t* sink;
  hipMalloc(&din, n4 * 16); hipMalloc(&dout, n4 * 16); hipMalloc(&sink, 4);
  hipMemset(din, 1, n4 * 16);
  for (int grid : {2048, 8192, 32768}) {
    float t;
    t = timeit([&] { hipLaunchKernelGGL(wr<false>, dim3(grid), dim3(256), 0, 0, (f32x4*)dout, n4); }, 10);
    printf("grid %5d  write        %.3f ms  %.0f GB/s\n", grid, t, n4 * 16 / t / 1e6);
    t = timeit([&] { hipLaunchKernelGGL(wr<true>, dim3(grid), dim3(256), 0, 0, (f32x4*)dout, n4); }, 10);
    printf("grid %5d  write nt     %.3f ms  %.0f GB/s\n", grid, t, n4 * 16 / t / 1e6);
    t = timeit([&] { hipLaunchKernelGGL(rd, dim3(grid), dim3(256), 0, 0, (const u32x4*)din, n4, sink); }, 10);
    printf("grid %5d  read         %.3f ms  %.0f GB/s\n", grid, t, n4 * 16 / t / 1e6);
    t = timeit([&] { hipLaunchKernelGGL(expand<false>, dim3(grid), dim3(256), 0, 0, (const uint32_t*)din, (f32x4*)dout, n4); }, 10);
    printf("grid %5d  u8->f32      %.3f ms  %.0f GB/s (4 B in + 16 B out)\n", grid, t, n4 * 20 / t / 1e6);
    t = timeit([&] { hipLaunchKernelGGL(expand<true>, dim3(grid), dim3(256), 0, 0, (const uint32_t*)din, (f32x4*)dout, n4); }, 10);
    printf("grid %5d  u8->f32 nt   %.3f ms  %.0f GB/s\n", grid, t, n4 * 20 / t / 1e6);
  }
  return 0;
}
