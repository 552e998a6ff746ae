// Odd square windows 21..31 (and the even ones under LSPIV_NO_PFA=1) embedded in the 64-point transforms (piv_fft_impl.h,
// "embedded mode"); 17 and 19 go to the direct kernel, the even sizes have FFT kernels of their own (piv_fftNN.hip).
#include "piv_fft_impl.h"

namespace lspiv {
hipError_t launch_piv_embed64(const PivParams& p, int dtype, bool ensemble, hipStream_t s) {
  return launch_embed<64>(p, dtype, ensemble, s);
}
}  // namespace lspiv
