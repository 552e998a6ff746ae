// Square windows 4, 5, 7 (and 6, 8 under LSPIV_NO_PFA=1) embedded in the 16-point transforms (piv_fft_impl.h, "embedded
// mode"): four jobs per wave.
#include "piv_fft_impl.h"

namespace lspiv {
hipError_t launch_piv_embed16(const PivParams& p, int dtype, bool ensemble, hipStream_t s) {
  return launch_embed<16>(p, dtype, ensemble, s);
}
}  // namespace lspiv
