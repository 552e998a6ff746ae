// 8x8 interrogation windows: instantiation of the fused FFT kernels (piv_fft_impl.h) on quarter-wave lane groups, eight
// of the sixteen lanes of a group idle along.
#include "piv_fft_impl.h"

namespace lspiv {
hipError_t launch_piv_fft8(const PivParams& p, int dtype, bool ensemble, hipStream_t s) {
  return launch_fft<8>(p, dtype, ensemble, s);
}
}  // namespace lspiv
