// Square windows 17..31 embedded in the 64-point transforms (piv_fft_impl.h, "embedded mode").
#include "piv_fft_impl.h"

namespace lspiv {
hipError_t launch_piv_embed64(const PivParams& p, int dtype, bool ensemble, hipStream_t s) {
  return launch_embed<64>(p, dtype, ensemble, s);
}
}  // namespace lspiv
