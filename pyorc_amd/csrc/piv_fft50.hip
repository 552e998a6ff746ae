// 50x50 interrogation windows (50 = 25 x 2: prime-factor FFT, fft_regs.h): instantiation of the fused FFT kernels
// (piv_fft_impl.h); a job runs on the next power-of-two lane group, the surplus lanes idle along.
#include "piv_fft_impl.h"

namespace lspiv {
hipError_t launch_piv_fft50(const PivParams& p, int dtype, bool ensemble, hipStream_t s) {
  return launch_fft<50>(p, dtype, ensemble, s);
}
}  // namespace lspiv
