// 16x16 interrogation windows: instantiation of the fused FFT kernels (piv_fft_impl.h), four jobs per wave.
#include "piv_fft_impl.h"

namespace lspiv {
hipError_t launch_piv_fft16(const PivParams& p, int dtype, bool ensemble, hipStream_t s) {
  return launch_fft<16>(p, dtype, ensemble, s);
}
}  // namespace lspiv
