// 64x64 interrogation-window FFT kernel -- placeholder translation unit.
// lspiv_kernel_kind() routes 64x64 to the direct kernel until this is implemented.
#include "common.h"
namespace lspiv {
hipError_t launch_piv_fft64(const PivParams&, int, bool, hipStream_t) { return hipErrorNotSupported; }
}  // namespace lspiv
