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
#include <algorithm>
#include <cstdlib>

#include "common.h"
#include "project_tile.h"
#include "project_fused.h"

namespace lspiv {

template <typename T, int FPT>
__global__ __launch_bounds__(256) void project_kernel(const T* __restrict__ frames, int64_t src_elems, int n_frames,
                                                      const int* __restrict__ nn_src, const int* __restrict__ grp_of,
                                                      const int* __restrict__ grp_off, const int* __restrict__ grp_src,
                                                      float* __restrict__ out, int n_out) {
  for (int o = blockIdx.x * blockDim.x + threadIdx.x; o < n_out; o += gridDim.x * blockDim.x) {
  const int nn = nn_src[o];
  const int g = grp_of[o];
  int k0 = 0, k1 = 0;
  if (g >= 0) { k0 = grp_off[g]; k1 = grp_off[g + 1]; }
  const float cnt = (float)(k1 - k0);
  const int t0 = blockIdx.y * FPT;
  const T* img = frames + (int64_t)t0 * src_elems;
  float* dst = out + (int64_t)t0 * n_out + o;
  if (g < 0 && t0 + FPT <= n_frames) {
    // the common cell: one nearest-neighbour sample per frame -- all FPT loads are issued before the first store
    float val[FPT];
#pragma unroll
    for (int t = 0; t < FPT; ++t) val[t] = nn >= 0 ? to_f32(img[(int64_t)t * src_elems + nn]) : 0.0f;
#pragma unroll
    for (int t = 0; t < FPT; ++t) dst[(int64_t)t * n_out] = (val[t] != val[t]) ? 0.0f : val[t];   // fillna(0.0)
    continue;
  }
  if (g >= 0 && t0 + FPT <= n_frames) {
    // group mean: samples outer, frames inner -- each sample index is read once and its FPT gathers are independent;
    // every frame still adds its samples in sample order (the numba loop's rounding)
    float acc[FPT];
#pragma unroll
    for (int t = 0; t < FPT; ++t) acc[t] = 0.0f;
    for (int k = k0; k < k1; ++k) {
      const int64_t si = grp_src[k];
#pragma unroll
      for (int t = 0; t < FPT; ++t) acc[t] += to_f32(img[(int64_t)t * src_elems + si]);
    }
#pragma unroll
    for (int t = 0; t < FPT; ++t) {
      const float val = acc[t] / cnt;                                  // IEEE division == float64 division rounded once
      dst[(int64_t)t * n_out] = (val != val) ? 0.0f : val;            // fillna(0.0)
    }
    continue;
  }
  const int nt = min(n_frames - t0, FPT);                              // ragged last block of frames
  for (int t = 0; t < nt; ++t, img += src_elems) {
    float val = 0.0f;
    if (nn >= 0) val = to_f32(img[nn]);
    if (g >= 0) {
      float s = 0.0f;
      for (int k = k0; k < k1; ++k) s += to_f32(img[grp_src[k]]);
      val = s / cnt;
    }
    dst[(int64_t)t * n_out] = (val != val) ? 0.0f : val;
  }
  }
}

template <typename T>
static void launch_project_t(const T* frames, int64_t src_elems, int n_frames, const int* nn_src, const int* grp_of,
                             const int* grp_off, const int* grp_src, float* out, int n_out, hipStream_t s) {
  static const int fpt = getenv("LSPIV_PROJECT_FPT") ? atoi(getenv("LSPIV_PROJECT_FPT")) : 8;
  static const int gx = getenv("LSPIV_PROJECT_GX") ? atoi(getenv("LSPIV_PROJECT_GX")) : 4096;   // x-persistent blocks
  const unsigned bx = gx > 0 ? (unsigned)std::min((n_out + 255) / 256, gx) : (unsigned)((n_out + 255) / 256);
#define LSPIV_PROJ(F)                                                                                              \
  hipLaunchKernelGGL((project_kernel<T, F>), dim3(bx, (n_frames + F - 1) / F), dim3(256), 0, s, frames, \
                     src_elems, n_frames, nn_src, grp_of, grp_off, grp_src, out, n_out)
  switch (fpt) {
    case 1: LSPIV_PROJ(1); break;
    case 2: LSPIV_PROJ(2); break;
    case 4: LSPIV_PROJ(4); break;
    case 16: LSPIV_PROJ(16); break;
    case 32: LSPIV_PROJ(32); break;
    default: LSPIV_PROJ(8); break;
  }
#undef LSPIV_PROJ
}

// uint8 camera frames, quads of four consecutive output cells (n_out % 4 == 0).  The one-cell kernel above moves 64 B per
// load instruction and 256 B per store instruction, and an identity plan shows that this, not HBM, holds it at 2.5 TB/s
// (a four-sample load + float4 store variant runs the identity at 5.2 TB/s).  A real homography has no four consecutive
// sources, but the sources of a quad lie within a few bytes of each other, in one camera row or -- where the row changes
// inside the quad -- in two: the host plan (lspiv_projection_create) stores per quad two 8-byte window starts and, per cell,
// which window and which byte.  A thread then loads two 8-byte windows per frame (512 B per wave-instruction), picks its
// four samples with shifts and stores one float4.  Quads that do not fit (a group mean, sources further apart) are
// listed for project_slow_kernel.
template <int F>
__global__ __launch_bounds__(256) void project_win_kernel(const uint8_t* __restrict__ frames, int64_t src_elems, int n_frames,
                                                          const int* __restrict__ qlo1, const int* __restrict__ qlo2,
                                                          const uint32_t* __restrict__ qdesc, float* __restrict__ out, int n_out) {
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  typedef uint64_t u64_u __attribute__((aligned(1)));
  const int t0 = blockIdx.y * F;
  const int nt = min(n_frames - t0, F);                    // block-uniform: the ragged last block of frames does not diverge
  const uint8_t* img = frames + (int64_t)t0 * src_elems;
  const int nq = n_out >> 2;
  for (int q = blockIdx.x * blockDim.x + threadIdx.x; q < nq; q += gridDim.x * blockDim.x) {
    const uint32_t d = qdesc[q];
    if (d >> 31) continue;                                 // project_slow_kernel's quad
    float* dst = out + (int64_t)t0 * n_out + 4 * q;
    const int a = qlo1[q], b = qlo2[q];
    uint64_t wa[F], wb[F];
#pragma unroll
    for (int t = 0; t < F; ++t)
      if (t < nt) {
        wa[t] = *reinterpret_cast<const u64_u*>(img + (int64_t)t * src_elems + a);
        wb[t] = *reinterpret_cast<const u64_u*>(img + (int64_t)t * src_elems + b);   // (predicating it on a != b was slower)
      }
#pragma unroll
    for (int t = 0; t < F; ++t)
      if (t < nt) {
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const uint32_t c = d >> (5 * e);
          const uint64_t w = (c & 8u) ? wb[t] : wa[t];
          const uint32_t byte = (uint32_t)(w >> (8 * (c & 7u))) & 0xffu;
          v[e] = (c & 16u) ? (float)byte : 0.0f;           // a cell without a source stays 0
        }
        *reinterpret_cast<f32x4*>(dst + (int64_t)t * n_out) = v;
      }
  }
}

// the quads the window plan leaves out (a group mean among the four cells, sources further apart), listed by the host: a
// thread = one cell of one such quad and F frames, the per-cell arithmetic of project_kernel (same float32 operations, same
// order).  A kernel of its own so that no wave of project_win_kernel runs both paths.
template <int F>
__global__ __launch_bounds__(256) void project_slow_kernel(const uint8_t* __restrict__ frames, int64_t src_elems, int n_frames,
                                                           const int* __restrict__ slow_q, int n_slow, const int* __restrict__ nn_src,
                                                           const int* __restrict__ grp_of, const int* __restrict__ grp_off,
                                                           const int* __restrict__ grp_src, float* __restrict__ out, int n_out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 4 * n_slow) return;
  const int o = 4 * slow_q[i >> 2] + (i & 3), nn = nn_src[o], g = grp_of[o];
  int k0 = 0, k1 = 0;
  if (g >= 0) { k0 = grp_off[g]; k1 = grp_off[g + 1]; }
  const float cnt = (float)(k1 - k0);
  const int t0 = blockIdx.y * F, nt = min(n_frames - t0, F);
  const uint8_t* im = frames + (int64_t)t0 * src_elems;
  float* dst = out + (int64_t)t0 * n_out + o;
  for (int t = 0; t < nt; ++t, im += src_elems) {
    float val = 0.0f;
    if (nn >= 0) val = (float)im[nn];
    if (g >= 0) {
      float sacc = 0.0f;
      for (int k = k0; k < k1; ++k) sacc += (float)im[grp_src[k]];
      val = sacc / cnt;
    }
    dst[(int64_t)t * n_out] = val;                         // uint8 samples: no NaN to fill
  }
}

hipError_t launch_project_win(const uint8_t* frames, int64_t src_elems, int n_frames, const int* qlo1, const int* qlo2,
                              const uint32_t* qdesc, const int* slow_q, int n_slow, const int* nn_src, const int* grp_of,
                              const int* grp_off, const int* grp_src, float* out, int n_out, hipStream_t s) {
  if (n_frames <= 0 || n_out <= 0) return hipSuccess;
  constexpr int F = 8;
  const unsigned bx = (unsigned)std::min((n_out / 4 + 255) / 256, 4096);
  hipLaunchKernelGGL((project_win_kernel<F>), dim3(bx, (n_frames + F - 1) / F), dim3(256), 0, s, frames, src_elems, n_frames, qlo1, qlo2,
                     qdesc, out, n_out);
  if (n_slow > 0)
    hipLaunchKernelGGL((project_slow_kernel<F>), dim3((unsigned)((4 * n_slow + 255) / 256), (n_frames + F - 1) / F), dim3(256), 0, s, frames,
                       src_elems, n_frames, slow_q, n_slow, nn_src, grp_of, grp_off, grp_src, out, n_out);
  return hipGetLastError();
}

// Plans WITH group means (reducer = "mean", the default: pyorc/project.py:196-199) on uint8 frames (round 6).  The quad-window kernel
// above serves nearest-neighbour cells only; with a fifth of the cells averaged (a camera at 4/3 of the grid's resolution) fewer
// than 90 % of the quads fit it and every frame went through the one-cell kernel: 64 B per load instruction, and -- because the
// lanes of a wave that hold a group take another branch than their neighbours -- every 128-byte line of the output written in
// two partial stores (WRITE_SIZE 1.8 x the output; profiles/r06_rows_project_before).  This kernel treats EVERY cell as a
// group: its samples (the group's, or the one nearest-neighbour byte, or none) lie in at most NW 8-byte windows per quad, the
// host plan gives per cell one byte MASK per window and the sample count.  uint8 samples add up to an integer below 2^24, so
// the float32 sum of the reference's loop is exact whatever the order: sum = sum over windows of dot4(window word, mask
// bytes) (v_dot4_u32_u8), value = float(sum) / float(count) -- the reference's one rounding.  Uniform code, one float4 store
// per quad and frame, 8-byte loads; quads whose samples need more windows go to project_slow_kernel as before.
// blockIdx.x -> quads: hardware block ids go round the 8 XCDs, so XCD x owns a contiguous eighth of the quads -- its share of
// the plan (24 or 48 B per quad) stays in its L2 from one frame group to the next, and neighbouring quads' camera lines are its own.
__device__ __forceinline__ uint32_t mask_bytes(uint32_t nibble) { return (nibble * 0x00204081u) & 0x01010101u; }   // bit i -> byte i

// float(sum) / float(count) without the division sequence: y = RN(1 / c) once per cell, then per frame q = RN(s y),
// r = s - q c (exact in one fma: s, c small integers), q' = RN(q + r y) -- the correctly rounded quotient for every
// 0 <= s <= 255 c, 1 <= c <= 255 (checked exhaustively against the division on the device: lspiv_debug_project_division,
// tests/test_project.py); plans with larger groups send those quads to project_slow_kernel.
__device__ __forceinline__ float quotient(float s, float c, float y) {
  const float q = s * y;
  const float r = __builtin_fmaf(-q, c, s);
  return __builtin_fmaf(r, y, q);
}
__global__ void division_check_kernel(int* __restrict__ mismatches) {
  const int c = blockIdx.x + 1;                            // 1 .. 255
  const float cf = (float)c, y = 1.0f / cf;
  int bad = 0;
  for (int s = threadIdx.x; s <= 255 * c; s += blockDim.x) {
    const float sf = (float)s;
    bad += (quotient(sf, cf, y) != sf / cf);
  }
  if (bad) atomicAdd(mismatches, bad);
}
hipError_t launch_division_check(int* d_mismatches, hipStream_t s) {
  hipLaunchKernelGGL(division_check_kernel, dim3(255), dim3(256), 0, s, d_mismatches);
  return hipGetLastError();
}

template <int F, int NW, bool DWORDS>
__global__ __launch_bounds__(256) void project_mix_kernel(const uint8_t* __restrict__ frames, int64_t src_elems, int n_frames,
                                                          const int* __restrict__ qwin, const uint32_t* __restrict__ qcell,
                                                          float* __restrict__ out, int n_out, int blocks_per_xcd) {
  constexpr int CW = NW / 2;                               // descriptor words per cell: NW = 2: masks | count << 16; NW = 4: masks, count
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  typedef int i32x2 __attribute__((ext_vector_type(2)));
  typedef uint64_t u64_u __attribute__((aligned(1)));
  typedef uint32_t u32x3 __attribute__((ext_vector_type(3), aligned(4)));
  const int t0 = blockIdx.y * F;
  const int nt = min(n_frames - t0, F);                    // block-uniform
  const uint8_t* img = frames + (int64_t)t0 * src_elems;
  const int nq = n_out >> 2;
  const int q = (((int)blockIdx.x & 7) * blocks_per_xcd + ((int)blockIdx.x >> 3)) * 256 + (int)threadIdx.x;
  if (q >= nq) return;
  int w[NW];
#pragma unroll
  for (int k = 0; k < NW; k += 2) {
    const i32x2 v = *reinterpret_cast<const i32x2*>(qwin + NW * (int64_t)q + k);
    w[k] = v[0]; w[k + 1] = v[1];
  }
  if (w[0] < 0) return;                                    // project_slow_kernel's quad
  uint32_t mlo[4][NW], mhi[4][NW];
  float cnt[4], rcp[4];
  {
    uint32_t d[4 * CW];
#pragma unroll
    for (int k = 0; k < CW; ++k) {
      const u32x4 v = *reinterpret_cast<const u32x4*>(qcell + 4 * CW * (int64_t)q + 4 * k);
      d[4 * k] = v[0]; d[4 * k + 1] = v[1]; d[4 * k + 2] = v[2]; d[4 * k + 3] = v[3];
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const uint32_t m = d[CW * e];
#pragma unroll
      for (int k = 0; k < NW; ++k) {
        mlo[e][k] = mask_bytes((m >> (8 * k)) & 15u);
        mhi[e][k] = mask_bytes((m >> (8 * k + 4)) & 15u);
      }
      cnt[e] = (float)(NW == 2 ? (m >> 16) : d[CW * e + 1]);
      rcp[e] = 1.0f / cnt[e];
    }
  }
  // The 8 bytes of a window start at any byte.  One unaligned 8-byte load per lane is what the hardware handles worst in a
  // gather (a wave's 64 windows lie ~5 bytes apart: every line is looked up for several lanes, and windows that straddle two
  // dwords cost twice): the three ALIGNED dwords around it in one 12-byte load and two v_alignbyte are 8-10 % faster (measured,
  // DESIGN.md section 3.5).
  uint32_t lo[F][NW], hi[F][NW];
  // (every frame's loads sit with their v_alignbyte under the frame's own `t < nt`: the compiler then waits for one frame after the
  // other.  Issuing all F frames' gathers back to back -- whole groups without guards -- measured 8 % SLOWER: the limit is the
  // request pattern of the gather in the L1 / TA, not latency)
#pragma unroll
  for (int t = 0; t < F; ++t)
    if (t < nt) {
#pragma unroll
      for (int k = 0; k < NW; ++k) {
        const uint8_t* p = img + (int64_t)t * src_elems + w[k];
        if (DWORDS) {
          // ONE aligned 12-byte load (global_load_dwordx3) of the three dwords the window touches (the plan keeps them inside the
          // frame: a window whose third dword would leave it sends its quad to project_slow_kernel)
          const u32x3 d = *reinterpret_cast<const u32x3*>(img + (int64_t)t * src_elems + (w[k] & ~3));
          const uint32_t sh = (uint32_t)w[k] & 3u;
          lo[t][k] = __builtin_amdgcn_alignbyte(d[1], d[0], sh);
          hi[t][k] = __builtin_amdgcn_alignbyte(d[2], d[1], sh);
        } else {
          const uint64_t v = *reinterpret_cast<const u64_u*>(p);
          lo[t][k] = (uint32_t)v; hi[t][k] = (uint32_t)(v >> 32);
        }
      }
    }
  float* dst = out + (int64_t)t0 * n_out + 4 * (int64_t)q;
#pragma unroll
  for (int t = 0; t < F; ++t)
    if (t < nt) {
      f32x4 v;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        uint32_t sum = 0;
#pragma unroll
        for (int k = 0; k < NW; ++k) {
          sum = __builtin_amdgcn_udot4(lo[t][k], mlo[e][k], sum, false);
          sum = __builtin_amdgcn_udot4(hi[t][k], mhi[e][k], sum, false);
        }
        v[e] = quotient((float)sum, cnt[e], rcp[e]);       // the reference's acc / count (a cell without samples: 0 / 1)
      }
      *reinterpret_cast<f32x4*>(dst + (int64_t)t * n_out) = v;
    }
}

hipError_t launch_project_mix(const uint8_t* frames, int64_t src_elems, int n_frames, int nw, const int* qwin, const uint32_t* qcell,
                              const int* slow_q, int n_slow, const int* nn_src, const int* grp_of, const int* grp_off,
                              const int* grp_src, float* out, int n_out, hipStream_t s) {
  if (n_frames <= 0 || n_out <= 0) return hipSuccess;
  constexpr int F = 8;
  static const bool unaligned = getenv("LSPIV_PROJECT_MIX_UNALIGNED") && atoi(getenv("LSPIV_PROJECT_MIX_UNALIGNED")) != 0;   // A/B: one unaligned 8-byte load per window
  const bool dwords = !unaligned && src_elems >= 12 && (src_elems & 3) == 0 && (reinterpret_cast<uintptr_t>(frames) & 3) == 0;
  const int nq = n_out / 4;
  const int per_xcd = ((nq + 255) / 256 + 7) / 8;
#define LSPIV_MIX(NN, DD)                                                                                                            \
  hipLaunchKernelGGL((project_mix_kernel<F, NN, DD>), dim3((unsigned)(8 * per_xcd), (unsigned)((n_frames + F - 1) / F)), dim3(256), 0, s, \
                     frames, src_elems, n_frames, qwin, qcell, out, n_out, per_xcd)
  if (nw == 2) { if (dwords) LSPIV_MIX(2, true); else LSPIV_MIX(2, false); }
  else { if (dwords) LSPIV_MIX(4, true); else LSPIV_MIX(4, false); }
#undef LSPIV_MIX
  if (n_slow > 0)
    hipLaunchKernelGGL((project_slow_kernel<F>), dim3((unsigned)((4 * n_slow + 255) / 256), (n_frames + F - 1) / F), dim3(256), 0, s, frames,
                       src_elems, n_frames, slow_q, n_slow, nn_src, grp_of, grp_off, grp_src, out, n_out);
  return hipGetLastError();
}

// project_mix_kernel with the gather taken out of the vector-memory path (round 6).  What held that kernel at half the streaming rate
// is the request pattern of its loads -- 64 lanes asking for 12 bytes every ~5 bytes: gather loads and stores did not overlap (0.153 +
// 0.171 ms = the 0.35 measured), the same byte count loaded coalesced took 0.277 ms.  Here a WAVE owns a block of 64 quads of the ortho
// grid (BQX quads wide, 64 / BQX rows high: the host picks the shape whose camera footprint is smallest) and the plan lists the 8-byte
// aligned CHUNKS of the camera frame that block's windows touch, sorted: at 4/3 oversampling 45 - 50 of them, a handful of runs along
// camera rows.  Lane l loads chunk l of the list (RMAX = 2: and chunk 64 + l) -- every camera byte the wave needs is asked for ONCE, 8
// bytes per lane, neighbouring lanes neighbouring addresses -- and parks it in the wave's own slice of LDS (no block barrier: the LDS
// operations of one wave execute in order, the wave_barrier only pins the compiler); a window is then the three dwords at its offset
// into that tile (chunks c and c + 1 of the frame are neighbours in the sorted list, so a window that straddles them reads on).  Same
// masks, same integer sums, same quotient as project_mix_kernel: same bits.  Waves whose footprint needs more chunks than RMAX * 64
// (wild geometry) give their quads to project_slow_kernel; too many of them: the plan is not built and project_mix_kernel runs.
#ifndef LSPIV_TILE_F
#define LSPIV_TILE_F 8          // frames per thread
#endif
#ifndef LSPIV_TILE_WAVES
#define LSPIV_TILE_WAVES 4      // waves per block
#endif
#ifndef LSPIV_TILE_NT
#define LSPIV_TILE_NT 0         // 1: non-temporal stores of the ortho frames
#endif
template <int F, int NW, int RMAX, typename OUT>      // OUT float: the reference's float32 cells; uint8_t: a nearest-neighbour-only plan's bytes as they are
__global__ __launch_bounds__(64 * LSPIV_TILE_WAVES) void project_tile_kernel(const uint8_t* __restrict__ frames, int64_t src_elems, int n_frames,
                                                           const int* __restrict__ wchunk, const int* __restrict__ twin,
                                                           const uint32_t* __restrict__ qcell, OUT* __restrict__ out, int n_out,
                                                           int wq, int rows, int lg_bqx, int tiles_x, int n_waves, int blocks_per_xcd) {
  constexpr int CW = NW / 2;
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  typedef int i32x2 __attribute__((ext_vector_type(2)));
  __shared__ uint64_t tile_all[LSPIV_TILE_WAVES][RMAX * 64 + 2];          // (+ 2: a window in the list's last chunk reads one dword past it)
  const int lane = threadIdx.x & 63;
  uint64_t* tile = tile_all[threadIdx.x >> 6];
  const uint32_t* tile32 = reinterpret_cast<const uint32_t*>(tile);
  const int t0 = blockIdx.y * F;
  const int nt = min(n_frames - t0, F);                    // block-uniform
  const uint8_t* img = frames + (int64_t)t0 * src_elems;
  // hardware block ids go round the 8 XCDs: XCD x owns a contiguous eighth of the waves (a band of the grid; its plan stays in its L2)
  const int gw = __builtin_amdgcn_readfirstlane((((int)blockIdx.x & 7) * blocks_per_xcd + ((int)blockIdx.x >> 3)) * LSPIV_TILE_WAVES + ((int)threadIdx.x >> 6));
  if (gw >= n_waves) return;                               // wave-uniform
  int64_t coff[RMAX];
  {
    int c0 = 0;
#pragma unroll
    for (int r = 0; r < RMAX; ++r) {
      const int c = wchunk[((int64_t)gw * RMAX + r) * 64 + lane];
      if (r == 0) c0 = c;
      coff[r] = 8 * (int64_t)c;
    }
    if (__builtin_amdgcn_readfirstlane(c0) < 0) return;    // nothing of this wave is served here (its quads are project_slow_kernel's)
  }
  // the wave's block of quads: ty, tx its position among the blocks, lane -> (row, quad column) inside it
  const int ty = gw / tiles_x, tx = gw - ty * tiles_x;
  const int row = (ty << (6 - lg_bqx)) + (lane >> lg_bqx), qx = (tx << lg_bqx) + (lane & ((1 << lg_bqx) - 1));
  const int q = row * wq + qx;
  // every lane takes part in the loads; only lanes with a quad of their own compute and store
  int w[NW];
  bool active = row < rows && qx < wq;
  if (active) {
#pragma unroll
    for (int k = 0; k < NW; k += 2) {
      const i32x2 v = *reinterpret_cast<const i32x2*>(twin + NW * (int64_t)q + k);
      w[k] = v[0]; w[k + 1] = v[1];
    }
    active = w[0] >= 0;
  }
  if (!active) {
#pragma unroll
    for (int k = 0; k < NW; ++k) w[k] = 0;
  }
  uint32_t mlo[4][NW], mhi[4][NW];
  float cnt[4], rcp[4];
  {
    uint32_t d[4 * CW];
#pragma unroll
    for (int k = 0; k < CW; ++k) {
      const u32x4 v = active ? *reinterpret_cast<const u32x4*>(qcell + 4 * CW * (int64_t)q + 4 * k) : u32x4{0u, 0u, 0u, 0u};
      d[4 * k] = v[0]; d[4 * k + 1] = v[1]; d[4 * k + 2] = v[2]; d[4 * k + 3] = v[3];
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const uint32_t m = d[CW * e];
#pragma unroll
      for (int k = 0; k < NW; ++k) {
        mlo[e][k] = mask_bytes((m >> (8 * k)) & 15u);
        mhi[e][k] = mask_bytes((m >> (8 * k + 4)) & 15u);
      }
      const uint32_t c = NW == 2 ? (m >> 16) : d[CW * e + 1];
      cnt[e] = (float)(c ? c : 1u);
      rcp[e] = 1.0f / cnt[e];
    }
  }
  // the chunks of all F frames in registers first (F independent loads in flight per lane and list row)
  uint64_t sv[F][RMAX];
#pragma unroll
  for (int t = 0; t < F; ++t)
    if (t < nt) {
#pragma unroll
      for (int r = 0; r < RMAX; ++r) sv[t][r] = *reinterpret_cast<const uint64_t*>(img + (int64_t)t * src_elems + coff[r]);
    }
  OUT* dst = out + (int64_t)t0 * n_out + 4 * (int64_t)q;
#pragma unroll
  for (int t = 0; t < F; ++t)
    if (t < nt) {
      __builtin_amdgcn_wave_barrier();                     // the previous frame's reads of the tile are issued before it is overwritten
#pragma unroll
      for (int r = 0; r < RMAX; ++r) tile[r * 64 + lane] = sv[t][r];
      __builtin_amdgcn_wave_barrier();
      uint32_t lo[NW], hi[NW];
#pragma unroll
      for (int k = 0; k < NW; ++k) {
        const int a = w[k] >> 2;
        const uint32_t d0 = tile32[a], d1 = tile32[a + 1], d2 = tile32[a + 2];
        const uint32_t sh = (uint32_t)w[k] & 3u;
        lo[k] = __builtin_amdgcn_alignbyte(d1, d0, sh);
        hi[k] = __builtin_amdgcn_alignbyte(d2, d1, sh);
      }
      if (active) {
        f32x4 v;
        uint32_t bytes = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          uint32_t sum = 0;
#pragma unroll
          for (int k = 0; k < NW; ++k) {
            sum = __builtin_amdgcn_udot4(lo[k], mlo[e][k], sum, false);
            sum = __builtin_amdgcn_udot4(hi[k], mhi[e][k], sum, false);
          }
          if (sizeof(OUT) == 1) bytes |= sum << (8 * e);   // one sample (or none) per cell: the sum is the byte
          else v[e] = quotient((float)sum, cnt[e], rcp[e]);
        }
        if (sizeof(OUT) == 1) *reinterpret_cast<uint32_t*>(dst + (int64_t)t * n_out) = bytes;
        else if (LSPIV_TILE_NT) __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(dst + (int64_t)t * n_out));
        else *reinterpret_cast<f32x4*>(dst + (int64_t)t * n_out) = v;
      }
    }
}

// the quads the tiles leave out, uint8 output: a cell is its nearest-neighbour byte or 0
template <int F>
__global__ __launch_bounds__(256) void project_slow_u8_kernel(const uint8_t* __restrict__ frames, int64_t src_elems, int n_frames,
                                                              const int* __restrict__ slow_q, int n_slow, const int* __restrict__ nn_src,
                                                              uint8_t* __restrict__ out, int n_out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 4 * n_slow) return;
  const int o = 4 * slow_q[i >> 2] + (i & 3), nn = nn_src[o];
  const int t0 = blockIdx.y * F, nt = min(n_frames - t0, F);
  const uint8_t* im = frames + (int64_t)t0 * src_elems;
  uint8_t* dst = out + (int64_t)t0 * n_out + o;
  for (int t = 0; t < nt; ++t, im += src_elems) dst[(int64_t)t * n_out] = nn >= 0 ? im[nn] : (uint8_t)0;
}

template <typename OUT>
static hipError_t launch_project_tile_t(const uint8_t* frames, int64_t src_elems, int n_frames, int nw, int rmax, const int* wchunk, const int* twin,
                                        const uint32_t* qcell, int wq, int rows, int lg_bqx, const int* slow_q, int n_slow, const int* nn_src,
                                        const int* grp_of, const int* grp_off, const int* grp_src, OUT* out, int n_out, hipStream_t s) {
  if (n_frames <= 0 || n_out <= 0) return hipSuccess;
  constexpr int F = LSPIV_TILE_F, F4 = LSPIV_TILE_F / 2;  // four list rows: half the frames per thread (the chunks of all of them wait in registers)
  const int bqx = 1 << lg_bqx, bqy = 64 >> lg_bqx;
  const int tiles_x = (wq + bqx - 1) / bqx, n_waves = tiles_x * ((rows + bqy - 1) / bqy);
  const int per_xcd = ((n_waves + LSPIV_TILE_WAVES - 1) / LSPIV_TILE_WAVES + 7) / 8;
  const int f = rmax > 2 ? F4 : F;
  const dim3 grid((unsigned)(8 * per_xcd), (unsigned)((n_frames + f - 1) / f));
#define LSPIV_TILE(FF, NN, RR)                                                                                                              \
  hipLaunchKernelGGL((project_tile_kernel<FF, NN, RR, OUT>), grid, dim3(64 * LSPIV_TILE_WAVES), 0, s, frames, src_elems, n_frames, wchunk, twin, \
                     qcell, out, n_out, wq, rows, lg_bqx, tiles_x, n_waves, per_xcd)
  if (nw == 2) { if (rmax <= 1) LSPIV_TILE(F, 2, 1); else if (rmax == 2) LSPIV_TILE(F, 2, 2); else LSPIV_TILE(F4, 2, 4); }
  else { if (rmax <= 1) LSPIV_TILE(F, 4, 1); else if (rmax == 2) LSPIV_TILE(F, 4, 2); else LSPIV_TILE(F4, 4, 4); }
#undef LSPIV_TILE
  if (n_slow > 0) {
    const dim3 sgrid((unsigned)((4 * n_slow + 255) / 256), (n_frames + F - 1) / F);
    if constexpr (sizeof(OUT) == 1)
      hipLaunchKernelGGL((project_slow_u8_kernel<F>), sgrid, dim3(256), 0, s, frames, src_elems, n_frames, slow_q, n_slow, nn_src, out, n_out);
    else
      hipLaunchKernelGGL((project_slow_kernel<F>), sgrid, dim3(256), 0, s, frames, src_elems, n_frames, slow_q, n_slow, nn_src, grp_of, grp_off,
                         grp_src, out, n_out);
  }
  return hipGetLastError();
}

hipError_t launch_project_tile(const uint8_t* frames, int64_t src_elems, int n_frames, int nw, int rmax, const int* wchunk, const int* twin,
                               const uint32_t* qcell, int wq, int rows, int lg_bqx, const int* slow_q, int n_slow, const int* nn_src,
                               const int* grp_of, const int* grp_off, const int* grp_src, float* out, int n_out, hipStream_t s) {
  return launch_project_tile_t<float>(frames, src_elems, n_frames, nw, rmax, wchunk, twin, qcell, wq, rows, lg_bqx, slow_q, n_slow, nn_src, grp_of,
                                      grp_off, grp_src, out, n_out, s);
}

hipError_t launch_project_tile_u8(const uint8_t* frames, int64_t src_elems, int n_frames, int nw, int rmax, const int* wchunk, const int* twin,
                                  const uint32_t* qcell, int wq, int rows, int lg_bqx, const int* slow_q, int n_slow, const int* nn_src,
                                  uint8_t* out, int n_out, hipStream_t s) {
  return launch_project_tile_t<uint8_t>(frames, src_elems, n_frames, nw, rmax, wchunk, twin, qcell, wq, rows, lg_bqx, slow_q, n_slow, nn_src, nullptr,
                                        nullptr, nullptr, out, n_out, s);
}

// FLOAT32 camera frames through tiles (round 6): what the reference's own recipe projects -- Frames.normalize -> edge_detect -> minmax
// come BEFORE project (examples/ngwerere/ngwerere.yml:5-11, pyorc/service/velocimetry.py:537-538), so project_numpy sees float32
// frames, and those went through the one-cell kernel (4-byte gathers, 0.37 of 8 TB/s).  Same idea as project_tile_kernel with pixels
// instead of bytes: a wave owns a block of 64 quads, the plan lists the 16-byte CHUNKS (four pixels) of the camera frame its samples lie
// in, sorted; lane l loads chunk l (and 64 + l ...: RMAX list rows) of every frame of its group -- each camera pixel is asked for once,
// 16 bytes per lane -- and parks them in the wave's slice of LDS.  A cell is then a short list of tile positions (10 bits each; up to
// 6 samples with DW = 2 descriptor words per cell, 9 with DW = 3) read from LDS and added IN THE REFERENCE'S ORDER
// (float32 sums are not associative: pyorc/project.py:19-53 adds a group's samples in index order), then the one IEEE division and
// fillna(0).  A nearest-neighbour cell is a "group" of one sample whose sum starts at -0.0f (x + -0.0f == x for every x, signed zeros
// and NaN payloads included) and is divided by 1.0f: the bits of the one-cell kernel.  Cells with more samples, waves with longer
// lists: project_slow_f32_kernel (the one-cell arithmetic).
template <int F, int DW, int RMAX>
__global__ __launch_bounds__(256) void project_tile_f32_kernel(const float* __restrict__ frames, int64_t src_elems, int n_frames,
                                                               const int* __restrict__ wchunk, const uint32_t* __restrict__ qdesc,
                                                               float* __restrict__ out, int n_out, int wq, int rows, int lg_bqx,
                                                               int tiles_x, int n_waves, int blocks_per_xcd) {
  constexpr int MAXS = 3 * DW;                             // samples per cell the descriptor holds
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  __shared__ f32x4 tile_all[4][RMAX * 64];
  const int lane = threadIdx.x & 63;
  f32x4* tile = tile_all[threadIdx.x >> 6];
  const float* tile_f = reinterpret_cast<const float*>(tile);
  const int t0 = blockIdx.y * F;
  const int nt = min(n_frames - t0, F);                    // block-uniform
  const float* img = frames + (int64_t)t0 * src_elems;
  const int gw = __builtin_amdgcn_readfirstlane((((int)blockIdx.x & 7) * blocks_per_xcd + ((int)blockIdx.x >> 3)) * 4 + ((int)threadIdx.x >> 6));
  if (gw >= n_waves) return;                               // wave-uniform
  int64_t coff[RMAX];
  {
    int c0 = 0;
#pragma unroll
    for (int r = 0; r < RMAX; ++r) {
      const int c = wchunk[((int64_t)gw * RMAX + r) * 64 + lane];
      if (r == 0) c0 = c;
      coff[r] = 4 * (int64_t)c;                            // in pixels
    }
    if (__builtin_amdgcn_readfirstlane(c0) < 0) return;    // nothing of this wave is served here
  }
  const int ty = gw / tiles_x, tx = gw - ty * tiles_x;
  const int row = (ty << (6 - lg_bqx)) + (lane >> lg_bqx), qx = (tx << lg_bqx) + (lane & ((1 << lg_bqx) - 1));
  const int q = row * wq + qx;
  bool active = row < rows && qx < wq;
  // descriptor of cell e = words d[DW e ...]: three 10-bit tile positions per word; the count in bits 30-31 of word 0 (low two bits) and
  // bit 30 (DW = 2) / bits 30-31 (DW = 3) of word 1; "is a group" (the sum starts at +0) in bit 31 of word 1 (DW = 2) / bit 30 of word 2
  uint32_t d[4 * DW];
#pragma unroll
  for (int k = 0; k < DW; ++k) {
    const u32x4 v = active ? *reinterpret_cast<const u32x4*>(qdesc + 4 * DW * (int64_t)q + 4 * k) : u32x4{0xffffffffu, 0u, 0u, 0u};
    d[4 * k] = v[0]; d[4 * k + 1] = v[1]; d[4 * k + 2] = v[2]; d[4 * k + 3] = v[3];
  }
  active = active && d[0] != 0xffffffffu;                  // (a slow quad's first word: no cell has the same position three times)
  int cnt[4];
  float fcnt[4], init[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const uint32_t w0 = active ? d[DW * e] : 0u, w1 = active ? d[DW * e + 1] : 0u;
    cnt[e] = (int)((w0 >> 30) | (DW == 2 ? ((w1 >> 30) & 1u) << 2 : (w1 >> 30) << 2));
    fcnt[e] = (float)max(cnt[e], 1);
    const uint32_t grp = DW == 2 ? w1 >> 31 : (d[DW * e + 2] >> 30) & 1u;
    // a group's sum starts at +0 (the reference's accumulator); a nearest-neighbour sample is taken as it is: -0 + x == x
    init[e] = grp ? 0.0f : -0.0f;
  }
  f32x4 sv[F][RMAX];
#pragma unroll
  for (int t = 0; t < F; ++t)
    if (t < nt) {
#pragma unroll
      for (int r = 0; r < RMAX; ++r) sv[t][r] = *reinterpret_cast<const f32x4*>(img + (int64_t)t * src_elems + coff[r]);
    }
  float* dst = out + (int64_t)t0 * n_out + 4 * (int64_t)q;
#pragma unroll
  for (int t = 0; t < F; ++t)
    if (t < nt) {
      __builtin_amdgcn_wave_barrier();                     // the previous frame's reads of the tile are issued before it is overwritten
#pragma unroll
      for (int r = 0; r < RMAX; ++r) tile[r * 64 + lane] = sv[t][r];
      __builtin_amdgcn_wave_barrier();
      f32x4 v;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float acc = init[e];
#pragma unroll
        for (int k = 0; k < MAXS; ++k)
          if (k < cnt[e]) acc += tile_f[(d[DW * e + k / 3] >> (10 * (k % 3))) & 1023u];
        const float val = cnt[e] > 1 ? acc / fcnt[e] : (cnt[e] == 1 ? acc : 0.0f);   // IEEE division == the reference's, rounded once
        v[e] = (val != val) ? 0.0f : val;                  // fillna(0.0)
      }
      if (active) *reinterpret_cast<f32x4*>(dst + (int64_t)t * n_out) = v;
    }
}

// the cells of the quads the float32 tiles leave out: project_kernel's arithmetic, one thread per cell and F frames
template <int F>
__global__ __launch_bounds__(256) void project_slow_f32_kernel(const float* __restrict__ frames, int64_t src_elems, int n_frames,
                                                               const int* __restrict__ slow_q, int n_slow, const int* __restrict__ nn_src,
                                                               const int* __restrict__ grp_of, const int* __restrict__ grp_off,
                                                               const int* __restrict__ grp_src, float* __restrict__ out, int n_out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 4 * n_slow) return;
  const int o = 4 * slow_q[i >> 2] + (i & 3), nn = nn_src[o], g = grp_of[o];
  int k0 = 0, k1 = 0;
  if (g >= 0) { k0 = grp_off[g]; k1 = grp_off[g + 1]; }
  const float cnt = (float)(k1 - k0);
  const int t0 = blockIdx.y * F, nt = min(n_frames - t0, F);
  const float* im = frames + (int64_t)t0 * src_elems;
  float* dst = out + (int64_t)t0 * n_out + o;
  for (int t = 0; t < nt; ++t, im += src_elems) {
    float val = 0.0f;
    if (nn >= 0) val = im[nn];
    if (g >= 0) {
      float sacc = 0.0f;
      for (int k = k0; k < k1; ++k) sacc += im[grp_src[k]];
      val = sacc / cnt;
    }
    dst[(int64_t)t * n_out] = (val != val) ? 0.0f : val;
  }
}

hipError_t launch_project_tile_f32(const float* frames, int64_t src_elems, int n_frames, int dw, int rmax, const int* wchunk,
                                   const uint32_t* qdesc, int wq, int rows, int lg_bqx, const int* slow_q, int n_slow, const int* nn_src,
                                   const int* grp_of, const int* grp_off, const int* grp_src, float* out, int n_out, hipStream_t s) {
  if (n_frames <= 0 || n_out <= 0) return hipSuccess;
  const int bqx = 1 << lg_bqx, bqy = 64 >> lg_bqx;
  const int tiles_x = (wq + bqx - 1) / bqx, n_waves = tiles_x * ((rows + bqy - 1) / bqy);
  const int per_xcd = ((n_waves + 3) / 4 + 7) / 8;
  // frames per thread: the chunks of all of them wait in registers (16 bytes per lane, list row and frame)
#define LSPIV_TILE(FF, DD, RR)                                                                                                               \
  hipLaunchKernelGGL((project_tile_f32_kernel<FF, DD, RR>), dim3((unsigned)(8 * per_xcd), (unsigned)((n_frames + FF - 1) / FF)), dim3(256), 0, s, \
                     frames, src_elems, n_frames, wchunk, qdesc, out, n_out, wq, rows, lg_bqx, tiles_x, n_waves, per_xcd)
  if (dw == 2) { if (rmax <= 1) LSPIV_TILE(4, 2, 1); else if (rmax == 2) LSPIV_TILE(4, 2, 2); else LSPIV_TILE(2, 2, 4); }
  else { if (rmax <= 1) LSPIV_TILE(4, 3, 1); else if (rmax == 2) LSPIV_TILE(4, 3, 2); else LSPIV_TILE(2, 3, 4); }
#undef LSPIV_TILE
  if (n_slow > 0)
    hipLaunchKernelGGL((project_slow_f32_kernel<8>), dim3((unsigned)((4 * n_slow + 255) / 256), (n_frames + 7) / 8), dim3(256), 0, s, frames,
                       src_elems, n_frames, slow_q, n_slow, nn_src, grp_of, grp_off, grp_src, out, n_out);
  return hipGetLastError();
}

// Nearest-neighbour-only plans (Frames.project with a reducer other than "mean", pyorc/project.py:196-199) on uint8 frames:
// every cell is a source byte or 0, so the stack may stay uint8 -- a quarter of the float32 bytes to write here and for the
// PIV kernels to read, and get_piv runs its uint8 kernels on the same values.  Quads of the window plan as above (one
// 32-bit store per quad and frame); the quads that plan leaves out gather their four bytes one by one.
template <int F>
__global__ __launch_bounds__(256) void project_win_u8_kernel(const uint8_t* __restrict__ frames, int64_t src_elems, int n_frames,
                                                             const int* __restrict__ qlo1, const int* __restrict__ qlo2,
                                                             const uint32_t* __restrict__ qdesc, const int* __restrict__ nn_src,
                                                             uint8_t* __restrict__ out, int n_out) {
  typedef uint64_t u64_u __attribute__((aligned(1)));
  const int t0 = blockIdx.y * F;
  const int nt = min(n_frames - t0, F);
  const uint8_t* img = frames + (int64_t)t0 * src_elems;
  const int nq = n_out >> 2;
  for (int q = blockIdx.x * blockDim.x + threadIdx.x; q < nq; q += gridDim.x * blockDim.x) {
    const uint32_t d = qdesc[q];
    uint32_t* dst = reinterpret_cast<uint32_t*>(out + (int64_t)t0 * n_out) + q;   // n_out % 4 == 0, `out` 4-byte aligned
    if (d >> 31) {
      int nn[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) nn[e] = nn_src[4 * q + e];
      for (int t = 0; t < nt; ++t) {
        uint32_t v = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (nn[e] >= 0) v |= (uint32_t)img[(int64_t)t * src_elems + nn[e]] << (8 * e);
        dst[(int64_t)t * nq] = v;
      }
      continue;
    }
    const int a = qlo1[q], b = qlo2[q];
    uint64_t wa[F], wb[F];
#pragma unroll
    for (int t = 0; t < F; ++t)
      if (t < nt) {
        wa[t] = *reinterpret_cast<const u64_u*>(img + (int64_t)t * src_elems + a);
        wb[t] = *reinterpret_cast<const u64_u*>(img + (int64_t)t * src_elems + b);
      }
#pragma unroll
    for (int t = 0; t < F; ++t)
      if (t < nt) {
        uint32_t v = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const uint32_t c = d >> (5 * e);
          const uint64_t w = (c & 8u) ? wb[t] : wa[t];
          const uint32_t byte = (uint32_t)(w >> (8 * (c & 7u))) & 0xffu;
          v |= ((c & 16u) ? byte : 0u) << (8 * e);
        }
        dst[(int64_t)t * nq] = v;
      }
  }
}

// the same without a window plan (odd grids, scattered sources): thread = cell, F frames
template <int F>
__global__ __launch_bounds__(256) void project_cell_u8_kernel(const uint8_t* __restrict__ frames, int64_t src_elems, int n_frames,
                                                              const int* __restrict__ nn_src, uint8_t* __restrict__ out, int n_out) {
  const int t0 = blockIdx.y * F;
  const int nt = min(n_frames - t0, F);
  const uint8_t* img = frames + (int64_t)t0 * src_elems;
  for (int o = blockIdx.x * blockDim.x + threadIdx.x; o < n_out; o += gridDim.x * blockDim.x) {
    const int nn = nn_src[o];
    uint8_t* dst = out + (int64_t)t0 * n_out + o;
    uint8_t val[F];
#pragma unroll
    for (int t = 0; t < F; ++t) val[t] = (t < nt && nn >= 0) ? img[(int64_t)t * src_elems + nn] : (uint8_t)0;
#pragma unroll
    for (int t = 0; t < F; ++t)
      if (t < nt) dst[(int64_t)t * n_out] = val[t];
  }
}

hipError_t launch_project_u8(const uint8_t* frames, int64_t src_elems, int n_frames, const int* qlo1, const int* qlo2,
                             const uint32_t* qdesc, const int* nn_src, uint8_t* out, int n_out, hipStream_t s) {
  if (n_frames <= 0 || n_out <= 0) return hipSuccess;
  constexpr int F = 8;
  if (qdesc && n_out % 4 == 0 && (reinterpret_cast<uintptr_t>(out) & 3) == 0) {
    const unsigned bx = (unsigned)std::min((n_out / 4 + 255) / 256, 4096);
    hipLaunchKernelGGL((project_win_u8_kernel<F>), dim3(bx, (n_frames + F - 1) / F), dim3(256), 0, s, frames, src_elems, n_frames, qlo1,
                       qlo2, qdesc, nn_src, out, n_out);
  } else {
    const unsigned bx = (unsigned)std::min((n_out + 255) / 256, 4096);
    hipLaunchKernelGGL((project_cell_u8_kernel<F>), dim3(bx, (n_frames + F - 1) / F), dim3(256), 0, s, frames, src_elems, n_frames, nn_src,
                       out, n_out);
  }
  return hipGetLastError();
}

hipError_t launch_project(const void* frames, int dtype, int64_t src_elems, int n_frames, const int* nn_src,
                          const int* grp_of, const int* grp_off, const int* grp_src, float* out, int n_out,
                          hipStream_t s) {
  if (n_frames <= 0 || n_out <= 0) return hipSuccess;
  switch (dtype) {
    case 0: launch_project_t((const uint8_t*)frames, src_elems, n_frames, nn_src, grp_of, grp_off, grp_src, out, n_out, s); break;
    case 1: launch_project_t((const float*)frames, src_elems, n_frames, nn_src, grp_of, grp_off, grp_src, out, n_out, s); break;
    case 2: launch_project_t((const double*)frames, src_elems, n_frames, nn_src, grp_of, grp_off, grp_src, out, n_out, s); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

// ---- project_cv (pyorc/project.py:56-120): cv2.undistort + cv2.warpPerspective as fixed-point bilinear remaps ----------
// OpenCV's remap with INTER_LINEAR: source coordinates quantised to 1/32 pixel (the maps are built on the host in double,
// lspiv_api.hip), 8-bit images blended with integer weights that sum to 2^15 and rounded + 2^14 >> 15, float images with the
// float32 weight table, neighbours outside the image count as 0 (BORDER_CONSTANT).  Thread = destination pixel, F frames
// per thread so that the map entry is read once per F frames.  No FMA contraction: the float path reproduces
// ((p00 w00 + p01 w01) + p10 w10) + p11 w11 in float32.
#pragma clang fp contract(off)
template <typename T, int F>
__global__ __launch_bounds__(256) void remap_kernel(const T* __restrict__ frames, int64_t src_elems, int Hs, int Ws, int n_frames,
                                                     const int* __restrict__ mx, const int* __restrict__ my,
                                                     const uint16_t* __restrict__ mf, T* __restrict__ out, int n_out,
                                                     const int* __restrict__ quads = nullptr, int n_quads = 0) {
  const int t0 = blockIdx.y * F;
  const int nt = min(n_frames - t0, F);
  // `quads`: only the pixels of the listed groups of four (the ones remap_win_kernel leaves out)
  const int n_work = quads ? 4 * n_quads : n_out;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n_work; i += gridDim.x * 256) {
    const int o = quads ? 4 * quads[i >> 2] + (i & 3) : i;
    const int ix = mx[o], iy = my[o];
    const int fr = mf[o], fx = fr & 31, fy = fr >> 5;
    const bool x0 = ix >= 0 && ix < Ws, x1 = ix + 1 >= 0 && ix + 1 < Ws, y0 = iy >= 0 && iy < Hs, y1 = iy + 1 >= 0 && iy + 1 < Hs;
    const int64_t base = (int64_t)iy * Ws + ix;
    const T* img = frames + (int64_t)t0 * src_elems;
    T* dst = out + (int64_t)t0 * n_out + o;
    if (!((x0 || x1) && (y0 || y1))) {                 // wholly outside: border value
      for (int t = 0; t < nt; ++t) dst[(int64_t)t * n_out] = (T)0;
      continue;
    }
    if (x0 && x1 && y0 && y1 && nt == F) {
      // the common pixel: all four neighbours inside, a full group of frames.  Every load of the F frames is issued before
      // the first result is formed (the per-frame loop below pays a load-use latency per frame); the two neighbours of a
      // row are one 2-pixel load (unaligned loads are fine on global memory).
      typedef T pair_t __attribute__((ext_vector_type(2), aligned(sizeof(T))));
      pair_t top[F], bot[F];
#pragma unroll
      for (int t = 0; t < F; ++t) {
        top[t] = *reinterpret_cast<const pair_t*>(img + (int64_t)t * src_elems + base);
        bot[t] = *reinterpret_cast<const pair_t*>(img + (int64_t)t * src_elems + base + Ws);
      }
      if constexpr (sizeof(T) == 1) {
        const int w00 = (32 - fx) * (32 - fy) * 32, w01 = fx * (32 - fy) * 32, w10 = (32 - fx) * fy * 32, w11 = fx * fy * 32;
#pragma unroll
        for (int t = 0; t < F; ++t) {
          const int acc = (int)top[t][0] * w00 + (int)top[t][1] * w01 + (int)bot[t][0] * w10 + (int)bot[t][1] * w11;
          dst[(int64_t)t * n_out] = (T)((acc + (1 << 14)) >> 15);
        }
      } else {
        const float a = (float)fx / 32.0f, b = (float)fy / 32.0f;
        const float w00 = (1.0f - a) * (1.0f - b), w01 = a * (1.0f - b), w10 = (1.0f - a) * b, w11 = a * b;
#pragma unroll
        for (int t = 0; t < F; ++t)
          dst[(int64_t)t * n_out] = (T)((((float)top[t][0] * w00 + (float)top[t][1] * w01) + (float)bot[t][0] * w10) + (float)bot[t][1] * w11);
      }
      continue;
    }
    for (int t = 0; t < nt; ++t, img += src_elems) {
      const T p00 = (x0 && y0) ? img[base] : (T)0, p01 = (x1 && y0) ? img[base + 1] : (T)0;
      const T p10 = (x0 && y1) ? img[base + Ws] : (T)0, p11 = (x1 && y1) ? img[base + Ws + 1] : (T)0;
      if constexpr (sizeof(T) == 1) {
        const int w00 = (32 - fx) * (32 - fy) * 32, w01 = fx * (32 - fy) * 32, w10 = (32 - fx) * fy * 32, w11 = fx * fy * 32;
        const int acc = (int)p00 * w00 + (int)p01 * w01 + (int)p10 * w10 + (int)p11 * w11;
        dst[(int64_t)t * n_out] = (T)((acc + (1 << 14)) >> 15);
      } else {
        const float a = (float)fx / 32.0f, b = (float)fy / 32.0f;
        const float w00 = (1.0f - a) * (1.0f - b), w01 = a * (1.0f - b), w10 = (1.0f - a) * b, w11 = a * b;
        dst[(int64_t)t * n_out] = (T)((((float)p00 * w00 + (float)p01 * w01) + (float)p10 * w10) + (float)p11 * w11);
      }
    }
  }
}

// uint8 frames through the quad plan of lspiv_project_cv_create: four consecutive destination pixels whose 2 x 2
// neighbourhoods are interior, lie in one pair of source rows and within 8 source bytes share TWO 8-byte loads per frame
// (the one-pixel kernel issues eight 2-byte ones), blend with the same integer weights and leave as one packed dword.
// qbase: flat source index of (iy, min ix); qdesc: per pixel 16 bits = byte offset (3) | fx (5) | fy (5); bit 62 = all four wholly outside the source
// (zeros), bit 63 = not in the plan (the one-pixel kernel runs over a list of those).
template <int F>
__global__ __launch_bounds__(256) void remap_win_kernel(const uint8_t* __restrict__ frames, int64_t src_elems, int Ws, int n_frames,
                                                        const int* __restrict__ qbase, const uint64_t* __restrict__ qdesc,
                                                        uint8_t* __restrict__ out, int n_out) {
  typedef uint64_t u64_u __attribute__((aligned(1)));
  const int t0 = blockIdx.y * F;
  const int nt = min(n_frames - t0, F);
  const uint8_t* img = frames + (int64_t)t0 * src_elems;
  const int nq = n_out >> 2;
  for (int q = blockIdx.x * blockDim.x + threadIdx.x; q < nq; q += gridDim.x * blockDim.x) {
    const uint64_t d = qdesc[q];
    if (d >> 63) continue;
    if (d >> 62) {                                           // all four pixels wholly outside the source: border value
      uint32_t* z = reinterpret_cast<uint32_t*>(out + (int64_t)t0 * n_out + 4 * q);
      for (int t = 0; t < nt; ++t) z[(int64_t)t * (n_out >> 2)] = 0u;
      continue;
    }
    const int base = qbase[q];
    uint64_t top[F], bot[F];
#pragma unroll
    for (int t = 0; t < F; ++t)
      if (t < nt) {
        top[t] = *reinterpret_cast<const u64_u*>(img + (int64_t)t * src_elems + base);
        bot[t] = *reinterpret_cast<const u64_u*>(img + (int64_t)t * src_elems + base + Ws);
      }
    int sh[4], w00[4], w01[4], w10[4], w11[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const uint32_t c = (uint32_t)(d >> (16 * e)) & 0xffffu;
      const int fx = (c >> 3) & 31, fy = (c >> 8) & 31;
      sh[e] = 8 * (int)(c & 7u);
      w00[e] = (32 - fx) * (32 - fy) * 32; w01[e] = fx * (32 - fy) * 32; w10[e] = (32 - fx) * fy * 32; w11[e] = fx * fy * 32;
    }
    uint32_t* dst = reinterpret_cast<uint32_t*>(out + (int64_t)t0 * n_out + 4 * q);
#pragma unroll
    for (int t = 0; t < F; ++t)
      if (t < nt) {
        uint32_t packed = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const uint32_t tp = (uint32_t)(top[t] >> sh[e]), bt = (uint32_t)(bot[t] >> sh[e]);   // bytes 0, 1 = the two columns
          const int acc = (int)(tp & 0xffu) * w00[e] + (int)((tp >> 8) & 0xffu) * w01[e] + (int)(bt & 0xffu) * w10[e] +
                          (int)((bt >> 8) & 0xffu) * w11[e];
          packed |= (uint32_t)(uint8_t)((acc + (1 << 14)) >> 15) << (8 * e);
        }
        dst[(int64_t)t * (n_out >> 2)] = packed;
      }
  }
}

hipError_t launch_remap_win(const uint8_t* frames, int64_t src_elems, int Hs, int Ws, int n_frames, const int* qbase, const uint64_t* qdesc,
                            const int* slow_q, int n_slow, const int* mx, const int* my, const uint16_t* mf, uint8_t* out, int n_out,
                            hipStream_t s) {
  if (n_frames <= 0 || n_out <= 0) return hipSuccess;
  constexpr int F = 8;
  const dim3 grid((unsigned)std::min((n_out / 4 + 255) / 256, 4096), (unsigned)((n_frames + F - 1) / F));
  hipLaunchKernelGGL((remap_win_kernel<F>), grid, dim3(256), 0, s, frames, src_elems, Ws, n_frames, qbase, qdesc, out, n_out);
  if (n_slow > 0)   // the pixels of the quads the plan leaves out: the one-pixel kernel over the list
    hipLaunchKernelGGL((remap_kernel<uint8_t, F>), dim3((unsigned)std::min((4 * n_slow + 255) / 256, 4096), (unsigned)((n_frames + F - 1) / F)),
                       dim3(256), 0, s, frames, src_elems, Hs, Ws, n_frames, mx, my, mf, out, n_out, slow_q, n_slow);
  return hipGetLastError();
}

hipError_t launch_remap(const void* frames, int dtype, int64_t src_elems, int Hs, int Ws, int n_frames, const int* mx, const int* my,
                        const uint16_t* mf, void* out, int n_out, hipStream_t s) {
  if (n_frames <= 0 || n_out <= 0) return hipSuccess;
  constexpr int F = 8;
  const dim3 grid((unsigned)std::min((n_out + 255) / 256, 4096), (unsigned)((n_frames + F - 1) / F));
  if (dtype == 0)
    hipLaunchKernelGGL((remap_kernel<uint8_t, F>), grid, dim3(256), 0, s, (const uint8_t*)frames, src_elems, Hs, Ws, n_frames, mx, my, mf,
                       (uint8_t*)out, n_out);
  else if (dtype == 1)
    hipLaunchKernelGGL((remap_kernel<float, F>), grid, dim3(256), 0, s, (const float*)frames, src_elems, Hs, Ws, n_frames, mx, my, mf,
                       (float*)out, n_out);
  else
    return hipErrorInvalidValue;
  return hipGetLastError();
}

// ---- project_cv, both remaps in ONE kernel (round 6) ----------------------------------------------------------------------------------
// cv2.undistort followed by cv2.warpPerspective writes and re-reads an undistorted uint8 stack the size of the camera stack: 5 x the
// algorithmic bytes (profiles/r06_rows_project_cv), and a fifth of the quads of either pass went through the per-pixel kernel.  Here a
// block owns a TILE of 64 x 16 destination pixels and a run of frames:
//   stage A  the bounding box of the undistorted pixels the tile's warp reads (plus a border of the constant 0 where it leaves the
//            image) is computed into LDS, four pixels per lane from two or THREE 8-byte windows of the camera frame (a quad of the
//            undistortion map may step into the next row pair: one `dy` bit per pixel), blended with OpenCV's integer weights and rounded
//            to uint8 exactly as the first pass stores them;
//   stage B  a lane warps four destination pixels from those bytes and stores one packed dword.
// Same integers as the two kernels in a row (tests/test_project.py); the camera frame is read once (+ the boxes' overlap, L2 hits), the
// undistorted stack never exists.  Plan (lspiv_api.hip, build_remap_fused): per tile the box {x0, y0, width, height} (x0 and width
// multiples of 4), per destination pixel `offset in the box (16) | fx (5) | fy (5) | bit 31: wholly outside`, per quad of the undistorted
// image the windows' base and `xoff (3) | fx (5) | fy (5) | dy (1)` per pixel, bit 15: three rows, bit 62: all outside, bit 63: per pixel.
// Measured on 201 1080p frames -> 810 x 1440 (tools/sessions/r06_cv_fused*.sh): the two passes 0.80 - 0.85 ms; this kernel 0.505 with the
// blend as four multiply-adds per pixel and byte reads from LDS; 0.86 with ds_read_u16 at odd addresses (~30 cycles per wave
// instruction); 0.47 with v_perm_b32 + v_dot4_u32_u8 row sums, ds_read2_b32 of the aligned dwords in stage B, 24-bit multiplies and no
// 64-bit index arithmetic (the arithmetic alone, every memory operation knocked out: 0.275 -> 0.154); 0.41 with the camera windows
// taken as dword-ALIGNED 12-byte loads + v_alignbyte_b32 (byte-aligned 8-byte loads: a third more).  What is left is the stage A loads
// (0.15 ms next to 0.15 of arithmetic, additive): not their number (without the third row: - 5 %), not occupancy (4 / 6 / 8 waves per SIMD:
// 0.49 / 0.47 / 0.52), not the lines a wave touches (8 x 8 quads per wave-round instead of 64 x 1: 0.49), frames per group 2 / 4 / 6 / 8:
// 0.52 / 0.47 / 0.475 / 0.48.
#ifndef LSPIV_RF_F
#define LSPIV_RF_F 4
#endif
constexpr int RF_TW = 64, RF_TH = 16, RF_F = LSPIV_RF_F;

__device__ __forceinline__ uint32_t rf_blend(uint32_t p00, uint32_t p01, uint32_t p10, uint32_t p11, int fx, int fy) {
  const int w00 = (32 - fx) * (32 - fy) * 32, w01 = fx * (32 - fy) * 32, w10 = (32 - fx) * fy * 32, w11 = fx * fy * 32;
  const int acc = (int)p00 * w00 + (int)p01 * w01 + (int)p10 * w10 + (int)p11 * w11;
  return (uint32_t)(uint8_t)((acc + (1 << 14)) >> 15);
}
// The same integer with a third of the instructions: the weights factor, w_rc = 32 (32 - fx | fx)_c (32 - fy | fy)_r, so with the row
// sums h_r = p_r0 (32 - fx) + p_r1 fx -- ONE v_dot4_u32_u8 on the two bytes as they lie next to each other (v_perm_b32 puts them into
// the low half of a dword, zeros above) -- (acc + 2^14) >> 15 = (h_0 (32 - fy) + h_1 fy + 2^9) >> 10; with the row weights scaled by
// 64 that byte is byte 2 of the sum (< 2^24: v_mad_u32_u24), and two more v_perm_b32 pack the four pixels of a lane.
__device__ __forceinline__ uint32_t rf_wx(uint32_t fx) { return (32u - fx) | fx << 8; }
__device__ __forceinline__ uint32_t rf_pack(uint32_t a0, uint32_t a1, uint32_t a2, uint32_t a3) {      // byte 2 of each
  return __builtin_amdgcn_perm(a1, a0, 0x0c0c0602u) | __builtin_amdgcn_perm(a3, a2, 0x06020c0cu);
}

// one group of F consecutive frames of a tile: stage A (the box of undistorted pixels into LDS), stage B (the warp from LDS)
template <int F>
__device__ __forceinline__ void rf_group(const uint8_t* __restrict__ img, int64_t src_elems, int Hs, int Ws, int bx0, int by0, int bw, int bq,
                                         int n_bq, float inv_bq, int rot, const int* __restrict__ qbase, const uint64_t* __restrict__ qdesc,
                                         const int* __restrict__ mx1, const int* __restrict__ my1, const uint16_t* __restrict__ mf1,
                                         uint8_t* __restrict__ lds, int box_cap, bool writes, const int (&b_off)[4], const uint32_t (&b_sel)[4],
                                         const uint32_t (&b_wx)[4], const uint32_t (&b_w0)[4], const uint32_t (&b_w1)[4],
                                         uint8_t* __restrict__ dst, int64_t n_out) {
  typedef uint32_t u32x2_u __attribute__((ext_vector_type(2), aligned(1)));
  // ---- stage A.  (The wave that takes the last, partial round changes from group to group: the waves of a block sit on different SIMDs.)
  for (int j = (int)((threadIdx.x + 64u * (unsigned)rot) & 255u); j < n_bq; j += 256) {
    const int r = (int)(((float)j + 0.5f) * inv_bq), c = j - __mul24(r, bq);   // (exact: bq bh <= 4096)
    const int qy = by0 + r, qx = bx0 + 4 * c;
    uint32_t res[F];
#pragma unroll
    for (int f = 0; f < F; ++f) res[f] = 0u;
    if (qy >= 0 && qy < Hs && qx >= 0 && qx < Ws) {                          // (Ws % 4 == 0: a quad is inside or outside as a whole)
      const int qi = __mul24(qy, Ws >> 2) + (qx >> 2);
      const uint64_t d = qdesc[qi];
      if (!(d >> 62)) {
        // 12 bytes per row from the dword-aligned address below the window (a byte-aligned 8-byte load costs a third more: 0.47
        // against 0.39 ms per 201 frames with the addresses rounded down), v_alignbyte_b32 shifts the window into place
        const int qb = qbase[qi];
        const uint8_t* p = img + (qb & ~3);
        const uint32_t ph = (uint32_t)qb & 3u;
        const bool three = (d >> 15) & 1u;
        typedef uint32_t u32x3_a __attribute__((ext_vector_type(3), aligned(4)));
        u32x2_u r0[F], r1[F], r2[F];
#pragma unroll
        for (int f = 0; f < F; ++f) {
          const u32x3_a a = *reinterpret_cast<const u32x3_a*>(p + f * src_elems);
          const u32x3_a b = *reinterpret_cast<const u32x3_a*>(p + f * src_elems + Ws);
          const u32x3_a c3 = three ? *reinterpret_cast<const u32x3_a*>(p + f * src_elems + 2 * Ws) : u32x3_a{0u, 0u, 0u};
          r0[f] = u32x2_u{__builtin_amdgcn_alignbyte(a[1], a[0], ph), __builtin_amdgcn_alignbyte(a[2], a[1], ph)};
          r1[f] = u32x2_u{__builtin_amdgcn_alignbyte(b[1], b[0], ph), __builtin_amdgcn_alignbyte(b[2], b[1], ph)};
          r2[f] = u32x2_u{__builtin_amdgcn_alignbyte(c3[1], c3[0], ph), __builtin_amdgcn_alignbyte(c3[2], c3[1], ph)};
        }
        uint32_t acc[F][4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const uint32_t cd = (uint32_t)(d >> (16 * e)) & 0xffffu;
          const uint32_t xo = cd & 7u, fx = (cd >> 3) & 31u, fy = (cd >> 8) & 31u;
          const bool dy = (cd >> 13) & 1u;                                   // the pixel's row pair starts at the middle row
          const uint32_t sel = 0x0c0c0000u | (xo + 1u) << 8 | xo;            // v_perm_b32: the two neighbouring bytes, zeros above them
          const uint32_t wx = rf_wx(fx);
          const uint32_t wo = dy ? 64u * fy : 64u * (32u - fy), wm = dy ? 64u * (32u - fy) : 64u * fy;   // weight of the other row, of the middle row
#pragma unroll
          for (int f = 0; f < F; ++f) {
            const uint32_t olo = dy ? r2[f][0] : r0[f][0], ohi = dy ? r2[f][1] : r0[f][1];
            const uint32_t ho = __builtin_amdgcn_udot4(__builtin_amdgcn_perm(ohi, olo, sel), wx, 0u, false);
            const uint32_t hm = __builtin_amdgcn_udot4(__builtin_amdgcn_perm(r1[f][1], r1[f][0], sel), wx, 0u, false);
            acc[f][e] = __umul24(ho, wo) + (__umul24(hm, wm) + (512u << 6));
          }
        }
#pragma unroll
        for (int f = 0; f < F; ++f) res[f] = rf_pack(acc[f][0], acc[f][1], acc[f][2], acc[f][3]);
      } else if (d >> 63) {                                                  // a quad outside the plan (the image's border, a fold of the map): per pixel
#pragma unroll 1
        for (int e = 0; e < 4; ++e) {
          const int o = __mul24(qy, Ws) + qx + e;
          const int ix = mx1[o], iy = my1[o], fr = mf1[o], fx = fr & 31, fy = fr >> 5;
          const bool x0 = ix >= 0 && ix < Ws, x1 = ix + 1 >= 0 && ix + 1 < Ws, y0 = iy >= 0 && iy < Hs, y1 = iy + 1 >= 0 && iy + 1 < Hs;
          if (!((x0 || x1) && (y0 || y1))) continue;
          const int b = __mul24(iy, Ws) + ix;
#pragma unroll 1
          for (int f = 0; f < F; ++f) {
            const uint8_t* p = img + f * src_elems;
            const uint32_t p00 = (x0 && y0) ? p[b] : 0u, p01 = (x1 && y0) ? p[b + 1] : 0u;
            const uint32_t p10 = (x0 && y1) ? p[b + Ws] : 0u, p11 = (x1 && y1) ? p[b + Ws + 1] : 0u;
            res[f] |= rf_blend(p00, p01, p10, p11, fx, fy) << (8 * e);
          }
        }
      }
    }
    uint32_t* w = reinterpret_cast<uint32_t*>(lds + __mul24(r, bw) + 4 * c);
#pragma unroll
    for (int f = 0; f < F; ++f) w[f * (box_cap >> 2)] = res[f];
  }
  __syncthreads();
  // ---- stage B: per pixel and row the two ALIGNED dwords around its two bytes (one ds_read2_b32; byte reads take twice the LDS
  // instructions, a ds_read_u16 at an odd address ~30 cycles), v_perm_b32 picks the bytes
  if (writes) {
#pragma unroll
    for (int f = 0; f < F; ++f) {
      const uint8_t* bx = lds + f * box_cap;
      uint32_t acc[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const uint32_t* qa = reinterpret_cast<const uint32_t*>(bx + b_off[e]);
        const uint32_t* qb = reinterpret_cast<const uint32_t*>(bx + b_off[e] + bw);
        const uint2 a = {qa[0], qa[1]}, b = {qb[0], qb[1]};                  // (ds_read2_b32: the pair is only 4-byte aligned)
        const uint32_t h0 = __builtin_amdgcn_udot4(__builtin_amdgcn_perm(a.y, a.x, b_sel[e]), b_wx[e], 0u, false);
        const uint32_t h1 = __builtin_amdgcn_udot4(__builtin_amdgcn_perm(b.y, b.x, b_sel[e]), b_wx[e], 0u, false);
        acc[e] = __umul24(h0, b_w0[e]) + (__umul24(h1, b_w1[e]) + (512u << 6));
      }
      *reinterpret_cast<uint32_t*>(dst + f * n_out) = rf_pack(acc[0], acc[1], acc[2], acc[3]);
    }
  }
  __syncthreads();
}

__global__ __launch_bounds__(256) void remap_fused_kernel(const uint8_t* __restrict__ frames, int64_t src_elems, int Hs, int Ws, int n_frames,
                                                          int seg_len, const int4* __restrict__ tiles, const uint32_t* __restrict__ pxd,
                                                          const int* __restrict__ qbase, const uint64_t* __restrict__ qdesc,
                                                          const int* __restrict__ mx1, const int* __restrict__ my1,
                                                          const uint16_t* __restrict__ mf1, uint8_t* __restrict__ out, int Hd, int Wd,
                                                          int tiles_x, int box_cap, int n_tiles) {
  extern __shared__ __attribute__((aligned(16))) uint8_t rf_lds[];          // RF_F boxes of box_cap bytes (+ 8: stage B reads whole dwords)
  // consecutive tiles (row-major: neighbours whose boxes overlap) to ONE XCD's L2: workgroups go round the eight XCDs
  const int per_xcd = (n_tiles + 7) >> 3;
  const int tile = (int)(blockIdx.x & 7u) * per_xcd + (int)(blockIdx.x >> 3);
  if (tile >= n_tiles) return;
  const int tx = tile % tiles_x, ty = tile / tiles_x;
  const int4 box = tiles[tile];
  const int bx0 = box.x, by0 = box.y, bw = box.z, bh = box.w, bq = bw >> 2, n_bq = bq * bh;
  const float inv_bq = 1.0f / (float)max(bq, 1);
  const int t0 = blockIdx.y * seg_len, t1 = min(t0 + seg_len, n_frames);
  // stage B: this lane's four destination pixels -- the aligned dword under the pixel's first byte, the byte selector, the weight
  // words of the columns and the (x 64) weights of the rows
  const int oy = ty * RF_TH + (threadIdx.x >> 4), ox = tx * RF_TW + 4 * (threadIdx.x & 15);
  const bool writes = oy < Hd && ox < Wd;                                    // (Wd % 4 == 0)
  const int64_t n_out = (int64_t)Hd * Wd;
  int b_off[4]; uint32_t b_sel[4], b_wx[4], b_w0[4], b_w1[4];
  {
    uint4 d = {0x80000000u, 0x80000000u, 0x80000000u, 0x80000000u};
    if (writes) d = *reinterpret_cast<const uint4*>(pxd + (int64_t)oy * Wd + ox);
    const uint32_t pd[4] = {d.x, d.y, d.z, d.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const bool zero = pd[e] >> 31;                                         // wholly outside: weights 0 on the box's first bytes
      const uint32_t off = zero ? 0u : pd[e] & 0xffffu, fx = (pd[e] >> 16) & 31u, fy = (pd[e] >> 21) & 31u, lo = off & 3u;
      b_off[e] = (int)(off & ~3u);
      b_sel[e] = 0x0c0c0000u | (lo + 1u) << 8 | lo;
      b_wx[e] = zero ? 0u : rf_wx(fx);
      b_w0[e] = 64u * (32u - fy); b_w1[e] = 64u * fy;
    }
  }
  uint8_t* dst = out + (int64_t)t0 * n_out + (int64_t)oy * Wd + ox;
  const uint8_t* img = frames + (int64_t)t0 * src_elems;
  int t = t0, rot = tile;
  for (; t + RF_F <= t1; t += RF_F, ++rot, img += RF_F * src_elems, dst += RF_F * n_out)
    rf_group<RF_F>(img, src_elems, Hs, Ws, bx0, by0, bw, bq, n_bq, inv_bq, rot, qbase, qdesc, mx1, my1, mf1, rf_lds, box_cap, writes, b_off,
                   b_sel, b_wx, b_w0, b_w1, dst, n_out);
  for (; t < t1; ++t, img += src_elems, dst += n_out)
    rf_group<1>(img, src_elems, Hs, Ws, bx0, by0, bw, bq, n_bq, inv_bq, rot, qbase, qdesc, mx1, my1, mf1, rf_lds, box_cap, writes, b_off,
                b_sel, b_wx, b_w0, b_w1, dst, n_out);
}

// The same for FLOAT32 frames (the recipe's edge-detected stack through method "cv"): the box holds floats (F = 4 / 2 / 1 frames per
// group by its size), stage A works per undistorted PIXEL -- the map entry once per group, two 2-pixel loads per frame, remap_kernel's
// float32 expression ((p00 w00 + p01 w01) + p10 w10) + p11 w11 with its weights (no contraction: the pragma above) -- and stage B blends
// four destination pixels per lane from LDS (one ds_read2_b32 per row) into one 16-byte store.  Neighbours outside the image are the
// +0 of the box's border, exactly what the per-pixel kernel multiplies by.  Two float32 passes: 1.52 ms per 201 1080p frames.
template <int F>
__global__ __launch_bounds__(256) void remap_fused_f32_kernel(const float* __restrict__ frames, int64_t src_elems, int Hs, int Ws, int n_frames,
                                                              int seg_len, const int4* __restrict__ tiles, const uint32_t* __restrict__ pxd,
                                                              const int* __restrict__ mx1, const int* __restrict__ my1,
                                                              const uint16_t* __restrict__ mf1, float* __restrict__ out, int Hd, int Wd,
                                                              int tiles_x, int box_cap, int n_tiles) {
  typedef float pair_t __attribute__((ext_vector_type(2), aligned(4)));
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  extern __shared__ __attribute__((aligned(16))) float rff_lds[];           // F boxes of box_cap floats
  const int per_xcd = (n_tiles + 7) >> 3;
  const int tile = (int)(blockIdx.x & 7u) * per_xcd + (int)(blockIdx.x >> 3);
  if (tile >= n_tiles) return;
  const int tx = tile % tiles_x, ty = tile / tiles_x;
  const int4 box = tiles[tile];
  const int bx0 = box.x, by0 = box.y, bw = box.z, n_box = box.z * box.w;
  const float inv_bw = 1.0f / (float)max(bw, 1);
  const int t0 = blockIdx.y * seg_len, t1 = min(t0 + seg_len, n_frames);
  const int oy = ty * RF_TH + (threadIdx.x >> 4), ox = tx * RF_TW + 4 * (threadIdx.x & 15);
  const bool writes = oy < Hd && ox < Wd;                                    // (Wd % 4 == 0)
  const int64_t n_out = (int64_t)Hd * Wd;
  int b_off[4]; float b00[4], b01[4], b10[4], b11[4]; bool b_zero[4];
  {
    uint4 d = {0x80000000u, 0x80000000u, 0x80000000u, 0x80000000u};
    if (writes) d = *reinterpret_cast<const uint4*>(pxd + (int64_t)oy * Wd + ox);
    const uint32_t pd[4] = {d.x, d.y, d.z, d.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      b_zero[e] = pd[e] >> 31;
      b_off[e] = b_zero[e] ? 0 : (int)(pd[e] & 0xffffu);
      const float a = (float)((pd[e] >> 16) & 31u) / 32.0f, b = (float)((pd[e] >> 21) & 31u) / 32.0f;
      b00[e] = (1.0f - a) * (1.0f - b); b01[e] = a * (1.0f - b); b10[e] = (1.0f - a) * b; b11[e] = a * b;
    }
  }
  for (int t = t0; t < t1; t += F) {
    const int nt = min(F, t1 - t);
    const float* img = frames + (int64_t)t * src_elems;
    // ---- stage A: the box of undistorted pixels
    for (int j = threadIdx.x; j < n_box; j += 256) {
      int r = (int)((float)j * inv_bw);
      r -= (r * bw > j); r += ((r + 1) * bw <= j);                           // (the float quotient may be off by one)
      const int c = j - r * bw;
      const int qy = by0 + r, qx = bx0 + c;
      float res[F];
#pragma unroll
      for (int f = 0; f < F; ++f) res[f] = 0.0f;
      if (qy >= 0 && qy < Hs && qx >= 0 && qx < Ws) {
        const int o = __mul24(qy, Ws) + qx;
        const int ix = mx1[o], iy = my1[o], fr = mf1[o];
        const bool x0 = ix >= 0 && ix < Ws, x1 = ix + 1 >= 0 && ix + 1 < Ws, y0 = iy >= 0 && iy < Hs, y1 = iy + 1 >= 0 && iy + 1 < Hs;
        const float a = (float)(fr & 31) / 32.0f, b = (float)(fr >> 5) / 32.0f;
        const float w00 = (1.0f - a) * (1.0f - b), w01 = a * (1.0f - b), w10 = (1.0f - a) * b, w11 = a * b;
        const int base = __mul24(iy, Ws) + ix;
        if (x0 && x1 && y0 && y1) {
          pair_t top[F], bot[F];
#pragma unroll
          for (int f = 0; f < F; ++f)
            if (f < nt) {
              top[f] = *reinterpret_cast<const pair_t*>(img + f * src_elems + base);
              bot[f] = *reinterpret_cast<const pair_t*>(img + f * src_elems + base + Ws);
            }
#pragma unroll
          for (int f = 0; f < F; ++f)
            if (f < nt) res[f] = ((top[f][0] * w00 + top[f][1] * w01) + bot[f][0] * w10) + bot[f][1] * w11;
        } else if ((x0 || x1) && (y0 || y1)) {
          for (int f = 0; f < nt; ++f) {
            const float* p = img + f * src_elems;
            const float p00 = (x0 && y0) ? p[base] : 0.0f, p01 = (x1 && y0) ? p[base + 1] : 0.0f;
            const float p10 = (x0 && y1) ? p[base + Ws] : 0.0f, p11 = (x1 && y1) ? p[base + Ws + 1] : 0.0f;
            res[f] = ((p00 * w00 + p01 * w01) + p10 * w10) + p11 * w11;
          }
        }
      }
#pragma unroll
      for (int f = 0; f < F; ++f)
        if (f < nt) rff_lds[f * box_cap + j] = res[f];
    }
    __syncthreads();
    // ---- stage B: the warp, from LDS
    if (writes) {
#pragma unroll
      for (int f = 0; f < F; ++f)
        if (f < nt) {
          const float* bx = rff_lds + f * box_cap;
          f32x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float p00 = bx[b_off[e]], p01 = bx[b_off[e] + 1], p10 = bx[b_off[e] + bw], p11 = bx[b_off[e] + bw + 1];
            const float v = ((p00 * b00[e] + p01 * b01[e]) + p10 * b10[e]) + p11 * b11[e];
            o[e] = b_zero[e] ? 0.0f : v;
          }
          *reinterpret_cast<f32x4*>(out + (int64_t)(t + f) * n_out + (int64_t)oy * Wd + ox) = o;
        }
    }
    __syncthreads();
  }
}

hipError_t launch_remap_fused_f32(const float* frames, int64_t src_elems, int Hs, int Ws, int n_frames, const void* tiles, int n_tiles,
                                  int tiles_x, int box_cap, const uint32_t* pxd, const int* mx1, const int* my1, const uint16_t* mf1,
                                  float* out, int Hd, int Wd, hipStream_t s) {
  if (n_frames <= 0 || n_tiles <= 0) return hipSuccess;
  const int F = 4 * 4 * box_cap <= 65000 ? 4 : 2 * 4 * box_cap <= 65000 ? 2 : 1;     // frames per group: F float boxes within 64 KB of LDS
  int n_seg = std::max(1, std::min((n_frames + F - 1) / F, (8192 + n_tiles - 1) / n_tiles));
  int seg_len = ((n_frames + n_seg - 1) / n_seg + F - 1) / F * F;
  n_seg = (n_frames + seg_len - 1) / seg_len;
  const dim3 grid((unsigned)((n_tiles + 7) / 8 * 8), (unsigned)n_seg);
  const size_t lds = (size_t)F * box_cap * sizeof(float);
#define LSPIV_RFF(FF) hipLaunchKernelGGL((remap_fused_f32_kernel<FF>), grid, dim3(256), lds, s, frames, src_elems, Hs, Ws, n_frames, seg_len, \
                                         (const int4*)tiles, pxd, mx1, my1, mf1, out, Hd, Wd, tiles_x, box_cap, n_tiles)
  if (F == 4) LSPIV_RFF(4); else if (F == 2) LSPIV_RFF(2); else LSPIV_RFF(1);
#undef LSPIV_RFF
  return hipGetLastError();
}

hipError_t launch_remap_fused(const uint8_t* frames, int64_t src_elems, int Hs, int Ws, int n_frames, const void* tiles, int n_tiles,
                              int tiles_x, int box_cap, const uint32_t* pxd, const int* qbase, const uint64_t* qdesc, const int* mx1,
                              const int* my1, const uint16_t* mf1, uint8_t* out, int Hd, int Wd, hipStream_t s) {
  if (n_frames <= 0 || n_tiles <= 0) return hipSuccess;
  // enough blocks to fill the chip a few times over, segments of whole groups of RF_F frames
  int n_seg = std::max(1, std::min((n_frames + RF_F - 1) / RF_F, (8192 + n_tiles - 1) / n_tiles));
  int seg_len = ((n_frames + n_seg - 1) / n_seg + RF_F - 1) / RF_F * RF_F;
  n_seg = (n_frames + seg_len - 1) / seg_len;
  hipLaunchKernelGGL(remap_fused_kernel, dim3((unsigned)((n_tiles + 7) / 8 * 8), (unsigned)n_seg), dim3(256), (size_t)RF_F * box_cap + 16, s, frames,
                     src_elems, Hs, Ws, n_frames, seg_len, (const int4*)tiles, pxd, qbase, qdesc, mx1, my1, mf1, out, Hd, Wd, tiles_x, box_cap,
                     n_tiles);
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
