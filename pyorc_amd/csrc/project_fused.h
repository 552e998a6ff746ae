// project_cv with both remaps in one kernel (project.hip; plan: lspiv_api.hip build_remap_fused).  Declared here and not in common.h:
// common.h is one of the four sources the PIV kernels' hash is taken over (Makefile KERNEL_SRC), and the committed PIV profiles are keyed to it.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace lspiv {
// uint8 frames
hipError_t launch_remap_fused(const uint8_t* frames, int64_t src_elems, int Hs, int Ws, int n_frames, const void* tiles, int n_tiles,
                              int tiles_x, int box_cap, const uint32_t* pxd, const int* qbase, const uint64_t* qdesc, const int* mx1,
                              const int* my1, const uint16_t* mf1, uint8_t* out, int Hd, int Wd, hipStream_t s);
// float32 frames
hipError_t launch_remap_fused_f32(const float* frames, int64_t src_elems, int Hs, int Ws, int n_frames, const void* tiles, int n_tiles,
                                  int tiles_x, int box_cap, const uint32_t* pxd, const int* mx1, const int* my1, const uint16_t* mf1,
                                  float* out, int Hd, int Wd, hipStream_t s);
}  // namespace lspiv
