// Test hook: the register FFTs of fft_regs.h, one transform per thread, so the parity suite can hold every length the
// PIV kernels instantiate (8, 16, 32, 64 and the prime-factor lengths P * 2^m) against numpy.fft directly.
#include "piv_fft_impl.h"

namespace lspiv {

template <int N, bool INV>
__global__ void fft_debug_kernel(const float* in, float* out, int count) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= count) return;
  float r[N], i[N];
#pragma unroll
  for (int j = 0; j < N; ++j) { r[j] = in[(size_t)t * 2 * N + 2 * j]; i[j] = in[(size_t)t * 2 * N + 2 * j + 1]; }
  fft_n<INV>(r, i);
#pragma unroll
  for (int j = 0; j < N; ++j) { out[(size_t)t * 2 * N + 2 * j] = r[j]; out[(size_t)t * 2 * N + 2 * j + 1] = i[j]; }
}

template <int N>
static hipError_t launch_one(bool inverse, const float* in, float* out, int count, hipStream_t s) {
  const dim3 grid((count + 63) / 64), block(64);
  if (inverse) hipLaunchKernelGGL((fft_debug_kernel<N, true>), grid, block, 0, s, in, out, count);
  else hipLaunchKernelGGL((fft_debug_kernel<N, false>), grid, block, 0, s, in, out, count);
  return hipGetLastError();
}

hipError_t launch_fft_debug(int n, bool inverse, const float* in, float* out, int count, hipStream_t s) {
  switch (n) {
    case 8: return launch_one<8>(inverse, in, out, count, s);
    case 16: return launch_one<16>(inverse, in, out, count, s);
    case 32: return launch_one<32>(inverse, in, out, count, s);
    case 64: return launch_one<64>(inverse, in, out, count, s);
#define LSPIV_PFA_CASE(m) case m: return launch_one<m>(inverse, in, out, count, s);
    LSPIV_PFA_SIZES(LSPIV_PFA_CASE)
#undef LSPIV_PFA_CASE
    default: return hipErrorInvalidValue;
  }
}

}  // namespace lspiv
