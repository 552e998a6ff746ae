// Orthoprojection of camera frames onto the PIV grid on the GPU.
//
// Replaces pyorc.project.img_to_ortho (pyorc/project.py:123-161) applied to every frame by project_numpy
// (:164-230), including the numba group average (:19-53) and Frames.project's fillna(0.0) (api/frames.py:265):
//   out[o] = 0
//   out[idx_ortho[k]] = img[idx_img[k]]                      nearest neighbour, undersampled cells
//   out[uidx[g]]     = mean_{i : norm_idx[i] = g} img[src_idx[i]]   oversampled cells, float32 sums IN SAMPLE ORDER
// The index maps are camera-geometry products of pyorc's CameraConfig (api/cameraconfig.py:739-860) and arrive as
// inputs; the host turns them once into a per-output-cell plan (nearest source index + CSR list of group members
// in their original order), then every frame is one gather kernel: HBM -> HBM, bit-identical to the reference
// loop because each group's float32 sum is accumulated in the same order.  Output is float32 (the reference
// returns the same float32 values widened to float64), laid out (T, Ho, Wo) -- exactly what the PIV kernels read.
#include "common.h"

namespace lspiv {

template <typename T>
__global__ __launch_bounds__(256) void project_kernel(const T* __restrict__ frames, int64_t src_elems, int n_frames,
                                                      const int* __restrict__ nn_src, const int* __restrict__ grp_of,
                                                      const int* __restrict__ grp_off, const int* __restrict__ grp_src,
                                                      float* __restrict__ out, int n_out) {
  const int o = blockIdx.x * blockDim.x + threadIdx.x;
  if (o >= n_out) return;
  const int nn = nn_src[o];
  const int g = grp_of[o];
  int k0 = 0, k1 = 0;
  if (g >= 0) { k0 = grp_off[g]; k1 = grp_off[g + 1]; }
  const float cnt = (float)(k1 - k0);
  const int t0 = blockIdx.y * 8;
  const int t1 = min(n_frames, t0 + 8);
  for (int t = t0; t < t1; ++t) {
    const T* img = frames + (int64_t)t * src_elems;
    float val = 0.0f;
    if (nn >= 0) val = to_f32(img[nn]);
    if (g >= 0) {
      float s = 0.0f;
      for (int k = k0; k < k1; ++k) s += to_f32(img[grp_src[k]]);  // sample order: same rounding as the numba loop
      val = s / cnt;                                                 // IEEE division == float64 division rounded once
    }
    out[(int64_t)t * n_out + o] = (val != val) ? 0.0f : val;         // fillna(0.0)
  }
}

hipError_t launch_project(const void* frames, int dtype, int64_t src_elems, int n_frames, const int* nn_src,
                          const int* grp_of, const int* grp_off, const int* grp_src, float* out, int n_out,
                          hipStream_t s) {
  if (n_frames <= 0 || n_out <= 0) return hipSuccess;
  const dim3 grid((n_out + 255) / 256, (n_frames + 7) / 8);
  switch (dtype) {
    case 0:
      hipLaunchKernelGGL(project_kernel<uint8_t>, grid, dim3(256), 0, s, (const uint8_t*)frames, src_elems, n_frames,
                         nn_src, grp_of, grp_off, grp_src, out, n_out);
      break;
    case 1:
      hipLaunchKernelGGL(project_kernel<float>, grid, dim3(256), 0, s, (const float*)frames, src_elems, n_frames, nn_src,
                         grp_of, grp_off, grp_src, out, n_out);
      break;
    case 2:
      hipLaunchKernelGGL(project_kernel<double>, grid, dim3(256), 0, s, (const double*)frames, src_elems, n_frames,
                         nn_src, grp_of, grp_off, grp_src, out, n_out);
      break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

// int16 packing of result variables (pyorc/const.py:80: dtype int16, scale_factor 0.01, _FillValue -9999), the
// arithmetic xarray applies on to_netcdf: float32 data / float32(scale) -> NaN -> fill -> np.around -> int16.
__global__ void pack_int16_kernel(const float* __restrict__ in, int64_t n, float scale, int fill, int16_t* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float x = in[i];
  float q = (x != x) ? (float)fill : rintf(x / scale);  // round half to even like np.around
  q = fminf(fmaxf(q, -32768.0f), 32767.0f);
  out[i] = (int16_t)q;
}

hipError_t launch_pack_int16(const float* in, int64_t n, float scale, int fill, int16_t* out, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(pack_int16_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, in, n, scale, fill, out);
  return hipGetLastError();
}

}  // namespace lspiv
