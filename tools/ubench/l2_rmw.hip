// why do 27.6 GB of stores of the 64 x 64 ensemble kernel reach the fabric when the 2 MB per XCD of live partial
// sums sit in a 4 MB L2?  Read-modify-write a region of 2 MB per XCD `iters` times (16 B per lane, the slot layout of the kernel):
//   mode 0  the RMW alone;  mode 1  + a streaming read of fresh bytes between the iterations (the frame stream), plain loads;
//   mode 2  the same stream with non-temporal loads.
// rocprofv3 --pmc WRITE_SIZE (and FETCH_SIZE, TCC_EA0_WRREQ_sum in their own passes) around it tells whether the L2 writes a dirty line
// back on every store (write-through: WRITE_SIZE ~ iters x 16 MB) or on eviction only (~ 16 MB without a stream).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
//   layout 1 (argv[3]): the hot bytes are the SECOND 8 KB of every 16 KB (the kernel's slots: the lower half of a slot is written once
//   per segment, the upper half every iteration) instead of one contiguous 2 MB per XCD; argv[4] = period: one 16-byte stream load
//   every `period` iterations (the kernel's frame reads mostly hit the L2: its MISS stream is a few per cent of the slot bytes).
template <int MODE>
__global__ __launch_bounds__(256) void rmw(f32x4* __restrict__ region, const f32x4* __restrict__ stream, size_t stream_quads, int iters,
                                           int blocks_per_xcd, float* __restrict__ sink, int layout, int period, int naps) {
  // hardware block ids go round the XCDs: XCD x = blockIdx.x & 7 owns region slice x (contiguous), as the ensemble kernel's jobs do
  const int xcd = blockIdx.x & 7, b = blockIdx.x >> 3;
  size_t i = ((size_t)xcd * blocks_per_xcd + b) * 256 + threadIdx.x;
  if (layout == 1) i = (i / 512) * 1024 + 512 + (i % 512);          // quads: 8 KB = 512 quads hot after 512 quads cold
  float acc = 0.0f;
  size_t sp = ((size_t)blockIdx.x * 256 + threadIdx.x) % stream_quads;
  __shared__ f32x4 tile[256];
  for (int it = 0; it < iters; ++it) {
    f32x4 v;
    if (MODE == 5) {
      // the kernel's way of reading its slot: global_load_lds_dwordx4 (asynchronous, straight into LDS), then ds_read
      const uint32_t lds = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uintptr_t)(__attribute__((address_space(3))) f32x4*)(tile + (threadIdx.x & ~63u)));
      const uint64_t base = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(region + (i - (threadIdx.x & 63u)))) |
                            ((uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)((uintptr_t)(region + (i - (threadIdx.x & 63u))) >> 32)) << 32);
      const uint32_t voff = (threadIdx.x & 63u) * 16u;
      uint32_t m0_saved;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0\n\ts_waitcnt vmcnt(0)"
                   : "=&s"(m0_saved) : "v"(voff), "s"(base), "s"(lds) : "memory");
      v = tile[threadIdx.x];
    } else {
      v = region[i];
    }
    v += 1.0f;
    region[i] = v;
    if (MODE && MODE != 5 && (it % period) == 0) {
      // 32 B of fresh stream per 16 B of slot per iteration: the kernel reads two uint8 windows (8 KB) per 16 KB slot pass -- the
      // stream here is heavier on purpose (a cache that survives this survives the kernel's)
      for (int k = 0; k < (period > 1 ? 1 : 2); ++k) {
        const f32x4 s = MODE == 2 ? __builtin_nontemporal_load(stream + sp) : stream[sp];
        acc += s[0] + s[3];
        sp += (size_t)gridDim.x * 256;
        if (sp >= stream_quads) sp -= stream_quads;
      }
    }
    for (int z = 0; z < naps; ++z) __builtin_amdgcn_s_sleep(127);    // argv[5]: ~3.4 us each between two passes over the slots (the kernel's iteration: ~15 us)
    asm volatile("" ::: "memory");
  }
  if (acc == 12345.678f) *sink = acc;
}
int main(int argc, char** argv) {
  const int mode = argc > 1 ? atoi(argv[1]) : 0, iters = argc > 2 ? atoi(argv[2]) : 500, layout = argc > 3 ? atoi(argv[3]) : 0, period = argc > 4 ? atoi(argv[4]) : 1, naps = argc > 5 ? atoi(argv[5]) : 0;
  const size_t per_xcd = (size_t)2 << 20, quads = 8 * per_xcd / 16;          // 2 MB per XCD
  const int blocks_per_xcd = (int)(per_xcd / 16 / 256);
  const size_t stream_quads = ((size_t)4 << 30) / 16;                         // a 4 GB stream: nothing of it is read twice while cached
  f32x4 *region, *stream; float* sink;
  (void)hipMalloc(&region, 2 * quads * 16); (void)hipMalloc(&stream, stream_quads * 16); (void)hipMalloc(&sink, 4);
  (void)hipMemset(region, 0, 2 * quads * 16); (void)hipMemset(stream, 0, stream_quads * 16);
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  (void)hipDeviceSynchronize(); (void)hipEventRecord(a);
  if (mode == 0) hipLaunchKernelGGL(rmw<0>, dim3(8 * blocks_per_xcd), dim3(256), 0, 0, region, stream, stream_quads, iters, blocks_per_xcd, sink, layout, period, naps);
  if (mode == 1) hipLaunchKernelGGL(rmw<1>, dim3(8 * blocks_per_xcd), dim3(256), 0, 0, region, stream, stream_quads, iters, blocks_per_xcd, sink, layout, period, naps);
  if (mode == 5) hipLaunchKernelGGL(rmw<5>, dim3(8 * blocks_per_xcd), dim3(256), 0, 0, region, stream, stream_quads, iters, blocks_per_xcd, sink, layout, period, naps);
  if (mode == 2) hipLaunchKernelGGL(rmw<2>, dim3(8 * blocks_per_xcd), dim3(256), 0, 0, region, stream, stream_quads, iters, blocks_per_xcd, sink, layout, period, naps);
  (void)hipEventRecord(b); (void)hipEventSynchronize(b);
  float ms; (void)hipEventElapsedTime(&ms, a, b);
  f32x4 first; (void)hipMemcpy(&first, region + (layout == 1 ? 512 : 0), 16, hipMemcpyDeviceToHost);
  printf("mode %d layout %d period %d naps %d: %d iterations over %zu MB (2 MB per XCD) in %.3f ms; region[0] = %.0f (expect %d); slot bytes stored %.2f GB, stream bytes read %.2f GB\n",
         mode, layout, period, naps, iters, quads * 16 >> 20, ms, first[0], iters, (double)iters * quads * 16 / 1e9, mode ? (double)iters * quads * 32 / 1e9 : 0.0);
  return 0;
}
