// Odd square windows 9..15 (and the even ones under LSPIV_NO_PFA=1) embedded in the 32-point transforms (piv_fft_impl.h,
// "embedded mode"); the even sizes have FFT kernels of their own (piv_fftNN.hip).
#include "piv_fft_impl.h"

namespace lspiv {
hipError_t launch_piv_embed32(const PivParams& p, int dtype, bool ensemble, hipStream_t s) {
  return launch_embed<32>(p, dtype, ensemble, s);
}
}  // namespace lspiv
