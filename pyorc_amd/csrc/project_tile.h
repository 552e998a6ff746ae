// Declarations of round 6's row kernels (the tiled orthoprojection launches of project.hip, the Gaussian filters with the clip of
// Frames.minmax in their store, filters.hip), kept out of common.h: that header is one of the four sources the committed PIV profile
// summaries are keyed to (Makefile: KERNEL_HASH), and these kernels are not on the PIV path.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace lspiv {
// the mixed plan in tiles: a wave = a block of 2^lg_bqx x 64 / 2^lg_bqx quads; the sorted 8-byte chunks its windows touch (rmax * 64 per
// wave) loaded one per lane and parked in LDS, windows = byte offsets into that tile (project.hip: project_tile_kernel)
hipError_t launch_project_tile(const uint8_t* frames, int64_t src_elems, int n_frames, int nw, int rmax, const int* wchunk, const int* twin,
                               const uint32_t* qcell, int wq, int rows, int lg_bqx, const int* slow_q, int n_slow, const int* nn_src,
                               const int* grp_of, const int* grp_off, const int* grp_src, float* out, int n_out, hipStream_t s);
// the same tiles with uint8 output: a nearest-neighbour-only plan (every cell one source byte or 0) keeps uint8 frames uint8
hipError_t launch_project_tile_u8(const uint8_t* frames, int64_t src_elems, int n_frames, int nw, int rmax, const int* wchunk, const int* twin,
                                  const uint32_t* qcell, int wq, int rows, int lg_bqx, const int* slow_q, int n_slow, const int* nn_src,
                                  uint8_t* out, int n_out, hipStream_t s);
// float32 frames in tiles: 16-byte chunks (four pixels), per cell a descriptor of dw words holding up to 3 * dw tile positions, the
// count and the group flag (project.hip: project_tile_f32_kernel)
hipError_t launch_project_tile_f32(const float* frames, int64_t src_elems, int n_frames, int dw, int rmax, const int* wchunk,
                                   const uint32_t* qdesc, int wq, int rows, int lg_bqx, const int* slow_q, int n_slow, const int* nn_src,
                                   const int* grp_of, const int* grp_off, const int* grp_src, float* out, int n_out, hipStream_t s);
// launch_blur (common.h) with np.maximum(np.minimum(x, hi), lo) applied as the result is stored; (-inf, +inf): launch_blur itself
hipError_t launch_blur_clip(const void* frames, int dtype, int n_frames, int H, int W, int ksize_a, int ksize_b, float lo, float hi,
                            float* out, hipStream_t s);
}  // namespace lspiv
