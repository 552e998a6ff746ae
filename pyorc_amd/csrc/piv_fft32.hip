// 32x32 interrogation windows: instantiation of the fused FFT kernels (piv_fft_impl.h).
#include "piv_fft_impl.h"

namespace lspiv {
hipError_t launch_piv_fft32(const PivParams& p, int dtype, bool ensemble, hipStream_t s) {
  return launch_fft<32>(p, dtype, ensemble, s);
}
}  // namespace lspiv
