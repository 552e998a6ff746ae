// Square windows 9..15 embedded in the 32-point transforms (4..8: piv_embed16.hip; 16 x 16 is native, piv_fft16.hip) (piv_fft_impl.h, "embedded mode").
#include "piv_fft_impl.h"

namespace lspiv {
hipError_t launch_piv_embed32(const PivParams& p, int dtype, bool ensemble, hipStream_t s) {
  return launch_embed<32>(p, dtype, ensemble, s);
}
}  // namespace lspiv
