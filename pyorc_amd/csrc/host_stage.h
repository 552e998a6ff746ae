// Host side of the host-pointer entry points: pageable frames -> pinned staging slots on a few persistent threads
// (plain C++, compiled by the host compiler: host_stage.cpp).
#pragma once
#include <stddef.h>

namespace lspiv_host {

// LSPIV_STAGE_THREADS, default min(16, hardware threads): one core moves ~10 GB/s, PCIe Gen5 x16 takes ~55, and float64 stacks
// are read at twice the rate they are sent
int stage_threads();
// dst[0, bytes) = src[0, bytes) on the staging threads (non-temporal stores: the slot is read by the DMA engine next, not by a core)
void staged_copy(void* dst, const void* src, size_t bytes);
// dst[i] = (float)(src[i] - offset_of_frame) for n_frames frames of frame_elems samples each, IEEE round-to-nearest like the
// kernels' own conversion.  offsets: one double per frame (nullptr or 0.0: the plain conversion, bit for bit).
void staged_narrow(float* dst, const double* src, size_t frame_elems, size_t n_frames, const double* offsets);
// DC offset of a float64 frame worth removing before it is narrowed to float32 (0.0: none).  float32 resolves 6e-8 of a
// sample's magnitude: a frame riding on an offset far above its contrast (mean 1e4, sigma 1) would lose the texture's low bits
// in the conversion, while the reference normalises every window in float64.  The per-window normalisation
// (x - mean_w) / std_w does not see a constant added to a frame, so a constant near the frame's mean may be taken off first.
// The estimate -- the mean of 4096 samples strided over the frame, rounded to an integer -- is a function of the frame
// alone (results do not depend on the chunking); it is applied only when its magnitude reaches `min_abs` (default 1024:
// never for 8-bit-like imagery, whose narrowed copies stay bit-identical to a float32 stack).
double frame_offset(const double* frame, size_t frame_elems, double min_abs);

}  // namespace lspiv_host
