// 64x64 interrogation windows: instantiation of the fused FFT kernels (piv_fft_impl.h); one window pair
// per wavefront (lane = row), BASELINE.json config 3 (64x64 @ 75 % overlap).
#include "piv_fft_impl.h"

namespace lspiv {
hipError_t launch_piv_fft64(const PivParams& p, int dtype, bool ensemble, hipStream_t s) {
  return launch_fft<64>(p, dtype, ensemble, s);
}
}  // namespace lspiv
