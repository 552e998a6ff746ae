// Micro-benchmark: plain vs packed fp32 VALU issue rate on gfx950 at different occupancies.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f2 __attribute__((ext_vector_type(2)));
constexpr int ITERS = 4096;
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, float a, float b) {
  float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
  f2 p0 = {x0, x1}, p1 = {x2, x3}, p2 = {x4, x5}, p3 = {x6, x7}, p4 = {x1, x0}, p5 = {x3, x2}, p6 = {x5, x4}, p7 = {x7, x6};
  f2 pa = {a, a}, pb = {b, b};
  for (int i = 0; i < ITERS; ++i) {
    if (MODE == 0) {  // 8 independent v_fma_f32
      asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                   "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                   : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
    } else if (MODE == 1) {  // 8 independent v_pk_fma_f32
      asm volatile("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n"
                   "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9\n"
                   : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pa), "v"(pb));
    } else if (MODE == 2) {  // 8 independent v_add_f32
      asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n"
                   "v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8\n"
                   : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a));
    } else if (MODE == 3) {  // 8 independent v_pk_add_f32
      asm volatile("v_pk_add_f32 %0, %0, %8\n v_pk_add_f32 %1, %1, %8\n v_pk_add_f32 %2, %2, %8\n v_pk_add_f32 %3, %3, %8\n"
                   "v_pk_add_f32 %4, %4, %8\n v_pk_add_f32 %5, %5, %8\n v_pk_add_f32 %6, %6, %8\n v_pk_add_f32 %7, %7, %8\n"
                   : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pa));
    } else if (MODE == 4) {  // dependent chain v_fma_f32 (ILP 1)
      asm volatile("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                   "v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                   : "+v"(x0) : "v"(a), "v"(b));
    } else if (MODE == 5) {  // mix: 4 v_add + 4 v_mul (VOP2 with literal-free operands)
      asm volatile("v_add_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_sub_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n"
                   "v_add_f32 %4, %4, %8\n v_fmac_f32 %5, %5, %8\n v_sub_f32 %6, %6, %8\n v_fmac_f32 %7, %7, %8\n"
                   : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a));
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + p0[0] + p1[1] + p2[0] + p3[1] + p4[0] + p5[1] + p6[0] + p7[1];
}
template <int MODE>
void run(const char* name, int flop_per_inst) {
  float* d; hipMalloc(&d, 256 * 256 * 16 * 4 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int wps : {1, 2, 3, 4, 8}) {   // waves per SIMD: blocks of 4 waves, wps blocks per CU via dynamic LDS
    int lds = (wps == 8 ? 16 : wps == 4 ? 36 : wps == 3 ? 50 : wps == 2 ? 70 : 150) * 1024;
    int blocks = 256 * wps * 4;  // 4 rounds
    hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), lds, 0, d, 1.0001f, 0.5f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), lds, 0, d, 1.0001f, 0.5f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double insts = (double)blocks * 4 * ITERS * 8;  // wave-level instructions
    double per_simd_cycle = insts / (1024.0 * ms * 1e-3 * 2.4e9);
    printf("%-18s waves/SIMD %d: %.3f ms  %.2f wave-inst/SIMD/4cyc(@2.4GHz)  %.1f TFLOP/s\n", name, wps, ms, per_simd_cycle * 4,
           insts * 64 * flop_per_inst / (ms * 1e-3) / 1e12);
  }
  hipFree(d);
}
int main() {
  run<0>("v_fma_f32", 2); run<1>("v_pk_fma_f32", 4); run<2>("v_add_f32", 1); run<3>("v_pk_add_f32", 2);
  run<4>("v_fma dep-chain", 2); run<5>("add/mul/fmac mix", 1);
  return 0;
}
