// Element-wise pre-processing filters on frame stacks in HBM: the ones the reference
// implements with plain numpy/xarray arithmetic, so they can be reproduced bit for bit --
//   Frames.time_diff  pyorc/api/frames.py:409-436    Frames.minmax  :344-362    Frames.normalize  :279-306    Frames.range  :364-379
// (edge_detect / smooth are cv2.GaussianBlur calls and are not covered).  All are HBM-bound streaming kernels:
// 16-byte accesses per lane, grid-stride over the stack.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>

#include "common.h"
#include "project_tile.h"

namespace lspiv {

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <typename T>
__global__ __launch_bounds__(256) void time_diff_kernel(const T* __restrict__ f, int64_t frame_elems, int64_t n_out,
                                                        float thres, int use_abs, float* __restrict__ out) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_out; i += stride) {
    float d = to_f32(f[i + frame_elems]) - to_f32(f[i]);  // float32 difference of consecutive frames
    d = (d > thres) ? d : 0.0f;                           // .where(d > thres) then .fillna(0.0): NaN compares false
    out[i] = use_abs ? fabsf(d) : d;
  }
}

// uint8 frames, four pixels per thread (round 6): two 4-byte loads and one 16-byte store per lane instead of two byte loads and a 4-byte
// store -- the byte-per-lane form moved 64 B per load instruction and ran at 3.8 TB/s of the 5.4 TB/s this traffic mix streams at
// (tools/ubench/stream.hip).  Same float32 arithmetic per pixel.
__global__ __launch_bounds__(256) void time_diff_u8x4_kernel(const uint32_t* __restrict__ f, int64_t frame_quads, int64_t n_quads,
                                                             float thres, int use_abs, f32x4* __restrict__ out) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_quads; i += stride) {
    const uint32_t a = f[i], b = f[i + frame_quads];
    f32x4 v;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float d = (float)((b >> (8 * k)) & 0xffu) - (float)((a >> (8 * k)) & 0xffu);
      d = (d > thres) ? d : 0.0f;
      v[k] = use_abs ? fabsf(d) : d;
    }
    out[i] = v;
  }
}

// The same per pixel, walking through time (round 6): a lane owns four pixels and S + 1 consecutive frames -- S + 1 four-byte loads, S
// 16-byte stores.  The flat kernel above asks for every frame twice (as `a` and as `b`; the second request is an L2 / Infinity-Cache hit,
// HBM fetch = the frames once) and ran at 4.2 TB/s where one load per store streams at 5.2 - 5.5 (tools/ubench/stream.hip: expand).
template <int S>
__global__ __launch_bounds__(256) void time_diff_u8x4_walk_kernel(const uint32_t* __restrict__ f, int64_t frame_quads, int n_out_frames,
                                                                  float thres, int use_abs, f32x4* __restrict__ out) {
  const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (q >= frame_quads) return;
  const int t0 = blockIdx.y * S, nt = min(n_out_frames - t0, S);      // block-uniform
  const uint32_t* src = f + (int64_t)t0 * frame_quads + q;
  f32x4* dst = out + (int64_t)t0 * frame_quads + q;
  uint32_t w[S + 1];
#pragma unroll
  for (int t = 0; t <= S; ++t)
    if (t <= nt) w[t] = src[(int64_t)t * frame_quads];
#pragma unroll
  for (int t = 0; t < S; ++t)
    if (t < nt) {
      f32x4 v;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float d = (float)((w[t + 1] >> (8 * k)) & 0xffu) - (float)((w[t] >> (8 * k)) & 0xffu);
        d = (d > thres) ? d : 0.0f;
        v[k] = use_abs ? fabsf(d) : d;
      }
      dst[(int64_t)t * frame_quads] = v;
    }
}

// Frames.range (pyorc/api/frames.py:364-379): (max over time - min over time).astype(input dtype), one thread per pixel
// column walking the frames; xarray's max / min skip NaN for float frames (nanmax / nanmin; an all-NaN pixel stays NaN).
template <typename T>
__global__ __launch_bounds__(256) void time_range_kernel(const T* __restrict__ f, int64_t frame_elems, int64_t n_frames,
                                                         T* __restrict__ out) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < frame_elems; i += stride) {
    T mx = f[i], mn = f[i];
    bool any = mx == mx;
    for (int64_t t = 1; t < n_frames; ++t) {
      const T x = f[t * frame_elems + i];
      if (x == x) {
        mx = (!any || x > mx) ? x : mx;
        mn = (!any || x < mn) ? x : mn;
        any = true;
      }
    }
    out[i] = any ? (T)(mx - mn) : mx;   // mx is the NaN of frame 0 when nothing else was seen
  }
}
// uint8: 16 pixels per thread and frame (one 16-byte load), byte-wise max / min on the packed words
typedef uint32_t u32x4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint32_t bytes_max(uint32_t a, uint32_t b) {
  uint32_t r = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) { const uint32_t x = (a >> (8 * k)) & 0xffu, y = (b >> (8 * k)) & 0xffu; r |= (x > y ? x : y) << (8 * k); }
  return r;
}
__device__ __forceinline__ uint32_t bytes_min(uint32_t a, uint32_t b) {
  uint32_t r = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) { const uint32_t x = (a >> (8 * k)) & 0xffu, y = (b >> (8 * k)) & 0xffu; r |= (x < y ? x : y) << (8 * k); }
  return r;
}
__global__ __launch_bounds__(256) void time_range_u8x16_kernel(const uint8_t* __restrict__ f, int64_t frame_elems, int64_t n_frames,
                                                               uint8_t* __restrict__ out) {
  const int64_t n_vec = frame_elems / 16;   // the caller handles the tail with the scalar kernel
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_vec; i += stride) {
    u32x4f mx = *reinterpret_cast<const u32x4f*>(f + 16 * i), mn = mx;
#pragma unroll 8
    for (int64_t t = 1; t < n_frames; ++t) {   // 8 independent 16-byte loads in flight per lane (only ~8 waves per CU at 1080p)
      const u32x4f x = *reinterpret_cast<const u32x4f*>(f + t * frame_elems + 16 * i);
#pragma unroll
      for (int k = 0; k < 4; ++k) { mx[k] = bytes_max(mx[k], x[k]); mn[k] = bytes_min(mn[k], x[k]); }
    }
    u32x4f r;
#pragma unroll
    for (int k = 0; k < 4; ++k) {   // byte-wise mx - mn (never borrows: mx >= mn in every byte)
      r[k] = mx[k] - mn[k];
    }
    *reinterpret_cast<u32x4f*>(out + 16 * i) = r;
  }
}

__global__ __launch_bounds__(256) void minmax_kernel(const float* __restrict__ in, int64_t n, float lo, float hi,
                                                     float* __restrict__ out) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const float x = in[i];
    // np.maximum(np.minimum(x, hi), lo): NaN propagates (numpy), unlike fminf / fmaxf
    const float a = (x != x) ? x : (x < hi ? x : hi);
    out[i] = (a != a) ? a : (a > lo ? a : lo);
  }
}

// normalize, pass 1: float32 mean over the sampled frames (exact integer sums for uint8)
__global__ __launch_bounds__(256) void sample_mean_kernel(const uint8_t* __restrict__ f, int64_t frame_elems, int n_frames,
                                                          int interval, float* __restrict__ mean) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= frame_elems) return;
  uint32_t s = 0, n = 0;
  for (int t = 0; t < n_frames; t += interval) { s += f[(int64_t)t * frame_elems + i]; ++n; }
  mean[i] = (float)((double)s / (double)n);  // numpy: float64 mean, then astype(float32)
}

// pass 2: per-frame min / max of (x - mean); one block per (frame, slice), combined with float atomics on the
// order-preserving integer image of the floats
__device__ __forceinline__ int f2ord(float x) { int i = __builtin_bit_cast(int, x); return i >= 0 ? i : i ^ 0x7fffffff; }
__device__ __forceinline__ float ord2f(int i) { return __builtin_bit_cast(float, i >= 0 ? i : i ^ 0x7fffffff); }

// VEC = 4: four pixels per thread (one dword of frame bytes, one float4 of the mean plane); needs frame_elems % 4 == 0
template <int VEC>
__global__ __launch_bounds__(256) void frame_minmax_kernel(const uint8_t* __restrict__ f, const float* __restrict__ mean,
                                                           int64_t frame_elems, int t0, int* __restrict__ mn,
                                                           int* __restrict__ mx) {
  const int t = t0 + blockIdx.y;
  const uint8_t* img = f + (int64_t)t * frame_elems;
  float lo = 3.0e38f, hi = -3.0e38f;
  const int64_t n = frame_elems / VEC;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    if (VEC == 4) {
      const uint32_t w = reinterpret_cast<const uint32_t*>(img)[i];
      const f32x4 m = reinterpret_cast<const f32x4*>(mean)[i];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float d = (float)((w >> (8 * e)) & 0xffu) - m[e];
        lo = fminf(lo, d);
        hi = fmaxf(hi, d);
      }
    } else {
      const float d = (float)img[i] - mean[i];
      lo = fminf(lo, d);
      hi = fmaxf(hi, d);
    }
  }
  lo = -half_max(-lo); hi = half_max(hi);
  lo = fminf(lo, __shfl_xor(lo, 32, 64)); hi = fmaxf(hi, __shfl_xor(hi, 32, 64));
  // one atomic pair per BLOCK: every atomic of a frame hits the same two addresses, so their number (not the bytes)
  // sets the speed of this pass -- one pair per wave made the kernel 10x slower at 2048 blocks per frame
  __shared__ float red[2][4];
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = lo; red[1][threadIdx.x >> 6] = hi; }
  __syncthreads();
  if (threadIdx.x == 0) {
    lo = fminf(fminf(red[0][0], red[0][1]), fminf(red[0][2], red[0][3]));
    hi = fmaxf(fmaxf(red[1][0], red[1][1]), fmaxf(red[1][2], red[1][3]));
    atomicMin(&mn[t], f2ord(lo));
    atomicMax(&mx[t], f2ord(hi));
  }
}

// pass 3: ((x - mean) - min) / (max - min) * 255 -> uint8 (truncation, NaN -> 0), all float32 like numpy
template <int VEC>
__global__ __launch_bounds__(256) void normalize_kernel(const uint8_t* __restrict__ f, const float* __restrict__ mean,
                                                        int64_t frame_elems, int t0, const int* __restrict__ mn,
                                                        const int* __restrict__ mx, uint8_t* __restrict__ out) {
  const int t = t0 + blockIdx.y;
  const float lo = ord2f(mn[t]), span = ord2f(mx[t]) - lo;
  const uint8_t* img = f + (int64_t)t * frame_elems;
  uint8_t* dst = out + (int64_t)t * frame_elems;
  const int64_t n = frame_elems / VEC;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    if (VEC == 4) {
      const uint32_t w = reinterpret_cast<const uint32_t*>(img)[i];
      const f32x4 m = reinterpret_cast<const f32x4*>(mean)[i];
      uint32_t o = 0;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float q = (((float)((w >> (8 * e)) & 0xffu) - m[e]) - lo) / span * 255.0f;
        o |= (uint32_t)((q != q) ? (uint8_t)0 : (uint8_t)(int)q) << (8 * e);
      }
      reinterpret_cast<uint32_t*>(dst)[i] = o;
    } else {
      const float q = (((float)img[i] - mean[i]) - lo) / span * 255.0f;
      dst[i] = (q != q) ? (uint8_t)0 : (uint8_t)(int)q;
    }
  }
}

// ---- normalize, space-major passes 2 and 3 -------------------------------------------------------------------------
// The frame-major kernels above re-read the float32 mean plane (4 B / pixel, 8 MB at 1080p: more than one XCD's L2) for
// every frame in both passes -- 8 B / pixel of L2 / Infinity-Cache traffic next to the 3 B / pixel of HBM traffic.  Here a
// block owns a SLICE of 8192 pixels and a run of frames: its part of the mean plane sits in registers (32 floats per
// lane), every frame costs its 1 B / pixel (+ 1 B written in pass 3) and nothing else.  Pass 2 leaves one (min, max)
// pair per wave and frame (plain stores, no atomics); a small kernel folds them per frame.  Same float32 arithmetic,
// min / max are order-independent: bit-identical to the frame-major path.  Needs frame_elems % 16 == 0.
constexpr int NORM_PX = 32;                 // pixels per lane: two 16-byte chunks
constexpr int NORM_SLICE = 256 * NORM_PX;   // pixels per block
typedef uint32_t u32x4n __attribute__((ext_vector_type(4)));

struct NormLane {
  int64_t c[2];      // first pixel of this lane's two chunks
  bool v[2];         // chunk inside the frame
  float m[2][16];    // the mean plane under them
  __device__ __forceinline__ void init(const float* __restrict__ mean, int64_t frame_elems) {
    const int64_t base = (int64_t)blockIdx.x * NORM_SLICE;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      c[k] = base + ((int64_t)k * 256 + threadIdx.x) * 16;
      v[k] = c[k] < frame_elems;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 x = v[k] ? reinterpret_cast<const f32x4*>(mean + c[k])[q] : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        m[k][4 * q] = x[0]; m[k][4 * q + 1] = x[1]; m[k][4 * q + 2] = x[2]; m[k][4 * q + 3] = x[3];
      }
    }
  }
};

__global__ __launch_bounds__(256) void norm_minmax_space_kernel(const uint8_t* __restrict__ f, const float* __restrict__ mean,
                                                                int64_t frame_elems, int n_frames, int seg_len,
                                                                float* __restrict__ part, int n_ws, const int* __restrict__ guard) {
  if (guard && *guard == 0) return;
  NormLane L;
  L.init(mean, frame_elems);
  const int t0 = blockIdx.y * seg_len, t1 = min(t0 + seg_len, n_frames);
  const int ws = blockIdx.x * 4 + (threadIdx.x >> 6);
  // one frame: this lane's 32 differences -> wave minimum / maximum -> one pair per wave
  auto reduce_frame = [&](int t, const u32x4n (&w)[2]) {
    float lo = 3.0e38f, hi = -3.0e38f;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      if (L.v[k]) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const float d = (float)((w[k][e >> 2] >> (8 * (e & 3))) & 0xffu) - L.m[k][e];
          lo = fminf(lo, d);
          hi = fmaxf(hi, d);
        }
      }
    }
    lo = -half_max(-lo); hi = half_max(hi);
    lo = fminf(lo, __shfl_xor(lo, 32, 64)); hi = fmaxf(hi, __shfl_xor(hi, 32, 64));
    if ((threadIdx.x & 63) == 0) {
      part[((int64_t)t * n_ws + ws) * 2] = lo;
      part[((int64_t)t * n_ws + ws) * 2 + 1] = hi;
    }
  };
  auto load_frame = [&](int t, u32x4n (&w)[2]) {
    const uint8_t* img = f + (int64_t)t * frame_elems;
#pragma unroll
    for (int k = 0; k < 2; ++k) w[k] = L.v[k] ? *reinterpret_cast<const u32x4n*>(img + L.c[k]) : u32x4n{0u, 0u, 0u, 0u};
  };
  // frames are independent: the loads of four of them are issued before the first reduction (the wave reductions are
  // convergent operations, which keeps the compiler from unrolling this loop on its own)
  int t = t0;
  for (; t + 4 <= t1; t += 4) {
    u32x4n w[4][2];
#pragma unroll
    for (int j = 0; j < 4; ++j) load_frame(t + j, w[j]);
#pragma unroll
    for (int j = 0; j < 4; ++j) reduce_frame(t + j, w[j]);
  }
  for (; t < t1; ++t) {
    u32x4n w[2];
    load_frame(t, w);
    reduce_frame(t, w);
  }
}

__global__ __launch_bounds__(256) void norm_fold_kernel(const float* __restrict__ part, int n_ws, float* __restrict__ lo_out,
                                                        float* __restrict__ hi_out, const int* __restrict__ guard) {
  if (guard && *guard == 0) return;
  const int t = blockIdx.x;
  float lo = 3.0e38f, hi = -3.0e38f;
  for (int i = threadIdx.x; i < n_ws; i += 256) {
    lo = fminf(lo, part[((int64_t)t * n_ws + i) * 2]);
    hi = fmaxf(hi, part[((int64_t)t * n_ws + i) * 2 + 1]);
  }
  lo = -half_max(-lo); hi = half_max(hi);
  lo = fminf(lo, __shfl_xor(lo, 32, 64)); hi = fmaxf(hi, __shfl_xor(hi, 32, 64));
  __shared__ float red[2][4];
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = lo; red[1][threadIdx.x >> 6] = hi; }
  __syncthreads();
  if (threadIdx.x == 0) {
    lo_out[t] = fminf(fminf(red[0][0], red[0][1]), fminf(red[0][2], red[0][3]));
    hi_out[t] = fmaxf(fmaxf(red[1][0], red[1][1]), fmaxf(red[1][2], red[1][3]));
  }
}

// The stretch of one frame of a lane's 32 pixels.
// The reference's  a / span * 255  costs an IEEE division per pixel (ten instructions: this pass ran at the VALU's rate, 0.22 of its
// 0.25 ms, profiles/r06_rows_normalize) -- but only the INTEGER PART of the result is kept.  With y = RN(1 / span) once per frame,
// (a y) 255 is within 5 roundings (3e-7 relative) of RN(RN(a / span) 255), so wherever it is further than that from an integer its
// integer part is the reference's; the few pixels nearer to one (about 1 instruction in 100 has such a lane) take the division
// under a wave-uniform branch.  Frames whose span has no usable reciprocal (0: a constant frame, 0 / 0 -> 0) divide everywhere.
__device__ __forceinline__ void norm_stretch_frame(const NormLane& L, const u32x4n (&wc)[2], float lo, float span, int mode, uint8_t* __restrict__ dst) {
  const float y = 1.0f / span;
  const bool fast = mode == 0 && span > 1e-30f && span < 1e30f;     // wave-uniform
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    if (L.v[k]) {
      const u32x4n w = wc[k];
      u32x4n o = {0u, 0u, 0u, 0u};
      if (fast) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const float a = ((float)((w[e >> 2] >> (8 * (e & 3))) & 0xffu) - L.m[k][e]) - lo;
          float q = (a * y) * 255.0f;
          const bool near = !(fabsf(q - rintf(q)) > 6.0e-7f * q + 1.0e-30f);   // (NaN: near)
          if (__builtin_amdgcn_ballot_w64(near) != 0) q = near ? a / span * 255.0f : q;
          o[e >> 2] |= (uint32_t)((q != q) ? (uint8_t)0 : (uint8_t)(int)q) << (8 * (e & 3));
        }
      } else {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const float q = (((float)((w[e >> 2] >> (8 * (e & 3))) & 0xffu) - L.m[k][e]) - lo) / span * 255.0f;
          o[e >> 2] |= (uint32_t)((q != q) ? (uint8_t)0 : (uint8_t)(int)q) << (8 * (e & 3));
        }
      }
      *reinterpret_cast<u32x4n*>(dst + L.c[k]) = o;
    }
  }
}

// `guard` (all three space-major kernels): nullptr, or a flag of the one-pass kernel below -- 0: it finished, nothing to do here
__global__ __launch_bounds__(256) void norm_stretch_space_kernel(const uint8_t* __restrict__ f, const float* __restrict__ mean,
                                                                 int64_t frame_elems, int n_frames, int seg_len,
                                                                 const float* __restrict__ lo_in, const float* __restrict__ hi_in,
                                                                 uint8_t* __restrict__ out, int mode, const int* __restrict__ guard) {
  if (guard && *guard == 0) return;
  NormLane L;
  L.init(mean, frame_elems);
  const int t0 = blockIdx.y * seg_len, t1 = min(t0 + seg_len, n_frames);
  // (the ballot in the stretch is a convergent operation: the compiler does not unroll this loop, so the next frame's loads are issued by hand)
  u32x4n wn[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) wn[k] = L.v[k] && t0 < t1 ? *reinterpret_cast<const u32x4n*>(f + (int64_t)t0 * frame_elems + L.c[k]) : u32x4n{0u, 0u, 0u, 0u};
  for (int t = t0; t < t1; ++t) {
    const float lo = lo_in[t], span = hi_in[t] - lo;
    const u32x4n wc[2] = {wn[0], wn[1]};
    if (t + 1 < t1) {
#pragma unroll
      for (int k = 0; k < 2; ++k) if (L.v[k]) wn[k] = *reinterpret_cast<const u32x4n*>(f + (int64_t)(t + 1) * frame_elems + L.c[k]);
    }
    norm_stretch_frame(L, wc, lo, span, mode, out + (int64_t)t * frame_elems);
  }
}

// ---- normalize in ONE pass over the frames (round 6) -----------------------------------------------------------------------------------
// The two passes above read every frame twice: 3 bytes of HBM traffic per pixel where the row needs 2.  Here the blocks that own the
// slices of a frame EXCHANGE their minima while they hold the frame in registers: one block per slice of 8192 pixels (253 at 1080p: all
// resident at once, the launcher checks it against the device's capacity), every block walks ALL frames in groups of NORM1_D -- load
// the group, reduce it, publish the block's (min, max) per frame (device-coherent stores, then an increment of one of the frame's counters),
// issue the NEXT group's loads, wait until the counters of the group have reached the number of blocks, fold the other
// blocks' pairs, stretch from the registers, store.  No block waits for anything another block does AFTER that block's own wait, so
// resident blocks cannot deadlock; a block that does not see a counter complete within NORM1_SPIN polls (blocks not resident together:
// the device shared with another long kernel) raises `fail` and leaves, every other block follows, and the three guarded kernels
// above redo the call.  Same float32 arithmetic, minimum and maximum do not depend on the order: the same bits.
// MEASURED, and therefore opt-in (LSPIV_NORM_ONE_PASS=1): with release / acquire fences at device scope around the exchange 1.67 ms per
// 201 1080p frames (a device-scope release writes back every dirty line of the XCD's L2 -- the stretched frames: ~30 us per group); with
// device-coherent (sc1) stores and loads of the pairs, a workgroup-scope fence for the acknowledgement and 16 counters per frame 0.94 ms
// -- ~18 us per group of four frames for store -> counter -> poll -> fold of 253 pairs -> stretch, one group in flight per block -- against
// 0.32 ms of the two passes.  Hiding that latency needs ~14 frames in flight per block, and then one wave per SIMD has to issue the 450
// VALU instructions a frame costs it (min / max + stretch) in the 0.85 us the frame may take: no slack at all for 0.08 ms at best.
constexpr int NORM1_D = 4, NORM1_SLOTS = 16;   // frames per group; counters per frame (block b counts on slot b % 16: 16 same-address increments in a row instead of 253)
constexpr unsigned NORM1_SPIN = 1u << 21;

__global__ __launch_bounds__(256) void norm_onepass_kernel(const uint8_t* __restrict__ f, const float* __restrict__ mean, int64_t frame_elems,
                                                           int n_frames, float* __restrict__ part, unsigned* __restrict__ ready,
                                                           int* __restrict__ fail, float* __restrict__ lo_out, float* __restrict__ hi_out,
                                                           uint8_t* __restrict__ out, int mode) {
  constexpr int D = NORM1_D;
  if (mode & 2) {                                                      // (test hook: as if the exchange had timed out)
    if (threadIdx.x == 0) __hip_atomic_store(fail, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return;
  }
  NormLane L;
  L.init(mean, frame_elems);
  const int nb = gridDim.x, b = blockIdx.x, wv = threadIdx.x >> 6;
  __shared__ float red[2][D][4];
  __shared__ float bc[2][D];
  __shared__ int gave_up;
  if (threadIdx.x == 0) gave_up = 0;
  auto load_group = [&](int g, u32x4n (&w)[D][2]) {
#pragma unroll
    for (int j = 0; j < D; ++j) {
      const uint8_t* img = f + (int64_t)min(g + j, n_frames - 1) * frame_elems;   // (beyond the last frame: the last one again, unused)
#pragma unroll
      for (int k = 0; k < 2; ++k) w[j][k] = L.v[k] ? *reinterpret_cast<const u32x4n*>(img + L.c[k]) : u32x4n{0u, 0u, 0u, 0u};
    }
  };
  // this block's (min, max) of every frame of the group -> part, the frames' counters + 1
  auto publish_group = [&](int g, const u32x4n (&w)[D][2]) {
#pragma unroll
    for (int j = 0; j < D; ++j) {
      float lo = 3.0e38f, hi = -3.0e38f;
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        if (L.v[k]) {
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const float d = (float)((w[j][k][e >> 2] >> (8 * (e & 3))) & 0xffu) - L.m[k][e];
            lo = fminf(lo, d);
            hi = fmaxf(hi, d);
          }
        }
      }
      lo = -half_max(-lo); hi = half_max(hi);
      lo = fminf(lo, __shfl_xor(lo, 32, 64)); hi = fmaxf(hi, __shfl_xor(hi, 32, 64));
      if ((threadIdx.x & 63) == 0) { red[0][j][wv] = lo; red[1][j][wv] = hi; }
    }
    __syncthreads();
    if (threadIdx.x < D && g + (int)threadIdx.x < n_frames) {
      const int j = threadIdx.x, t = g + j;
      // device-coherent stores (they go past this XCD's L2), acknowledged before the counter moves; NO release fence at device scope:
      // that one writes back every dirty line of the L2 -- the stretched frames -- and cost 30 us per group (1.67 ms per 201 frames)
      __hip_atomic_store(&part[((int64_t)t * nb + b) * 2], fminf(fminf(red[0][j][0], red[0][j][1]), fminf(red[0][j][2], red[0][j][3])),
                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&part[((int64_t)t * nb + b) * 2 + 1], fmaxf(fmaxf(red[1][j][0], red[1][j][1]), fmaxf(red[1][j][2], red[1][j][3])),
                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");           // s_waitcnt vmcnt(0): the two stores have been acknowledged
      __hip_atomic_fetch_add(&ready[(int64_t)t * NORM1_SLOTS + (b & (NORM1_SLOTS - 1))], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  };
  // false: the exchange timed out somewhere (block-uniform)
  auto finish_group = [&](int g, const u32x4n (&w)[D][2]) -> bool {
    if (wv == 0) {                                                     // lane = (frame of the group, counter slot): D x NORM1_SLOTS = 64
      const int j = (threadIdx.x & 63) / NORM1_SLOTS, slot = threadIdx.x & (NORM1_SLOTS - 1);
      const unsigned expected = g + j < n_frames && slot < nb ? (unsigned)((nb - 1 - slot) / NORM1_SLOTS + 1) : 0u;   // blocks with b % SLOTS == slot
      const unsigned* ctr = &ready[(int64_t)min(g + j, n_frames - 1) * NORM1_SLOTS + slot];
      unsigned polls = 0;
      while (__builtin_amdgcn_ballot_w64(__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < expected) != 0) {
        if (++polls > NORM1_SPIN || __hip_atomic_load(fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {   // (wave-uniform)
          __hip_atomic_store(fail, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          gave_up = 1;
          break;
        }
        __builtin_amdgcn_s_sleep(4);
      }
    }
    __syncthreads();                                                   // (also: red[] may be written again)
    if (gave_up) return false;
    // fold the blocks' pairs: wave j takes frame j of the group
    if (wv < D && g + wv < n_frames) {
      const int t = g + wv;
      float lo = 3.0e38f, hi = -3.0e38f;
      for (int i = threadIdx.x & 63; i < nb; i += 64) {
        lo = fminf(lo, __hip_atomic_load(&part[((int64_t)t * nb + i) * 2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        hi = fmaxf(hi, __hip_atomic_load(&part[((int64_t)t * nb + i) * 2 + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
      }
      lo = -half_max(-lo); hi = half_max(hi);
      lo = fminf(lo, __shfl_xor(lo, 32, 64)); hi = fmaxf(hi, __shfl_xor(hi, 32, 64));
      if ((threadIdx.x & 63) == 0) {
        bc[0][wv] = lo; bc[1][wv] = hi;
        if (b == 0) { lo_out[t] = lo; hi_out[t] = hi; }
      }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < D; ++j)
      if (g + j < n_frames) norm_stretch_frame(L, w[j], bc[0][j], bc[1][j] - bc[0][j], mode & 1, out + (int64_t)(g + j) * frame_elems);
    return true;
  };
  static_assert(D <= 4 && D * NORM1_SLOTS == 64, "one wave folds one frame of a group; one lane polls one counter");
  u32x4n wa[D][2], wb[D][2];
  load_group(0, wa);
  for (int g = 0; g < n_frames; g += 2 * D) {                          // two groups per trip: the buffers swap without moves
    publish_group(g, wa);
    if (g + D < n_frames) load_group(g + D, wb);
    if (!finish_group(g, wa)) return;
    if (g + D >= n_frames) break;
    publish_group(g + D, wb);
    if (g + 2 * D < n_frames) load_group(g + 2 * D, wa);
    if (!finish_group(g + D, wb)) return;
  }
}

// ---- Frames.reduce_rolling (pyorc/api/frames.py:381-407) on uint8 frames ------------------------------------------
//   roll = frames.rolling(time=samples).mean()          trailing window [t - samples + 1, t], NaN for t < samples - 1
//   thres = maximum(frames - roll, 0);  out = (thres * 255 / max over the frame of thres).astype(uint8).where(roll != 0, 0)
// all in float64.  The window sum of uint8 samples is an exact integer, so roll = sum / samples has one rounding whatever
// the summation order; a lane keeps the running integer sums of its 32 pixels in registers and walks a run of frames
// (space-major like the normalize passes above).  Two passes: per-wave maxima of thres -> norm_fold-style reduction ->
// the stretch.  Frames t < samples - 1 (NaN.astype(uint8): 0 on x86) and frames whose maximum is 0 (0 / 0) come out 0.
typedef u32x4n u32x4n_u __attribute__((aligned(1)));   // frames start at t * H * W: any byte alignment
struct RollLane {
  int64_t c[2];
  int nv[2];         // valid pixels of the chunk: 16, 0 (outside the frame) or the frame's tail
  int sum[2][16];
  __device__ __forceinline__ void init(int64_t frame_elems) {
    const int64_t base = (int64_t)blockIdx.x * NORM_SLICE;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      c[k] = base + ((int64_t)k * 256 + threadIdx.x) * 16;
      const int64_t left = frame_elems - c[k];
      nv[k] = left >= 16 ? 16 : (left > 0 ? (int)left : 0);
#pragma unroll
      for (int e = 0; e < 16; ++e) sum[k][e] = 0;
    }
  }
  __device__ __forceinline__ u32x4n load(const uint8_t* __restrict__ img, int k) const {
    if (nv[k] == 16) return *reinterpret_cast<const u32x4n_u*>(img + c[k]);
    u32x4n w = {0u, 0u, 0u, 0u};
    for (int e = 0; e < nv[k]; ++e) w[e >> 2] |= (uint32_t)img[c[k] + e] << (8 * (e & 3));
    return w;
  }
  __device__ __forceinline__ void store(uint8_t* __restrict__ dst, int k, u32x4n o) const {
    if (nv[k] == 16) { *reinterpret_cast<u32x4n_u*>(dst + c[k]) = o; return; }
    for (int e = 0; e < nv[k]; ++e) dst[c[k] + e] = (uint8_t)((o[e >> 2] >> (8 * (e & 3))) & 0xffu);
  }
  // sum += sign * frame
  __device__ __forceinline__ void add(const uint8_t* __restrict__ img, int sign) {
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      if (nv[k]) {
        const u32x4n w = load(img, k);
#pragma unroll
        for (int e = 0; e < 16; ++e) sum[k][e] += sign * (int)((w[e >> 2] >> (8 * (e & 3))) & 0xffu);
      }
    }
  }
};

// sum / n without the ~12-instruction IEEE division: q0 = sum * RN(1 / n), one Newton step on the exact remainder
// (Markstein).  Whether that is the correctly rounded quotient for EVERY window sum 0 .. 255 n of this call is checked
// exhaustively on the device first (rolling_check_kernel: at most 255 n + 1 values); the kernels take the division
// path if a single one differs.
__device__ __forceinline__ double div_small(double a, double n, double y) {
  const double q0 = a * y;
  const double r = fma(-n, q0, a);
  return fma(r, y, q0);
}
__global__ void rolling_check_kernel(int samples, int* __restrict__ mismatch) {
  const double n = (double)samples, y = 1.0 / n;
  const int count = 255 * samples + 1;
  int bad = 0;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < count; i += gridDim.x * blockDim.x) {
    const double a = (double)i;
    bad |= (div_small(a, n, y) != a / n) ? 1 : 0;
  }
  if (bad) atomicOr(mismatch, 1);
}

template <bool STRETCH>
__global__ __launch_bounds__(256) void rolling_kernel(const uint8_t* __restrict__ f, int64_t frame_elems, int n_frames, int samples,
                                                      int seg_len, double* __restrict__ part, int n_ws,
                                                      const double* __restrict__ frame_max, const int* __restrict__ mismatch,
                                                      uint8_t* __restrict__ out) {
  RollLane L;
  L.init(frame_elems);
  const int first = samples - 1;                                   // first frame with a complete window
  const int t0 = max(blockIdx.y * seg_len, first), t1 = min((blockIdx.y + 1) * seg_len, n_frames);
  const int ws = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (STRETCH) {                                                   // the frames without a complete window: zeros
    for (int t = blockIdx.y * seg_len; t < min(t1, first); ++t) {
#pragma unroll
      for (int k = 0; k < 2; ++k)
        if (L.nv[k]) L.store(out + (int64_t)t * frame_elems, k, u32x4n{0u, 0u, 0u, 0u});
    }
  }
  if (t0 >= t1) return;
  for (int t = t0 - first; t < t0; ++t) L.add(f + (int64_t)t * frame_elems, 1);   // the window of frame t0 minus frame t0
  const double n = (double)samples, y = 1.0 / n;
  const bool fast = *mismatch == 0;                                 // uniform
  for (int t = t0; t < t1; ++t) {
    const uint8_t* img = f + (int64_t)t * frame_elems;
    L.add(img, 1);
    double hi = 0.0;
    const double fm = STRETCH ? frame_max[t] : 1.0;
    const double inv_fm = 1.0 / fm;                                 // one division per frame, not per pixel (inf for fm = 0)
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      if (L.nv[k]) {
        const u32x4n w = L.load(img, k);
        u32x4n o = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int e = 0; e < 16; ++e) {   // pixels past a tail chunk read as 0 with sum 0: thres 0, never the maximum, not stored
          const double x = (double)((w[e >> 2] >> (8 * (e & 3))) & 0xffu);
          const double sm = (double)L.sum[k][e];
          const double d = x - (fast ? div_small(sm, n, y) : sm / n);
          const double th = d > 0.0 ? d : 0.0;                      // np.maximum(d, 0)
          if (STRETCH) {
            // (th * 255 / fm).astype(uint8): only the integer part matters.  num * RN(1 / fm) is within 1e-13 of the
            // divided value, so its integer part is the same unless it lies within 1e-9 of an integer -- then divide.
            const double num = th * 255.0;
            const double qf = num * inv_fm;
            int iq = (int)qf;
            const double fr = qf - (double)iq;
            if (fr < 1e-9 || fr > 1.0 - 1e-9) iq = (int)(num / fm);
            const uint32_t b = (fm == 0.0 || L.sum[k][e] == 0) ? 0u : (uint32_t)iq;   // 0 / 0 -> NaN -> 0; where(roll != 0, 0)
            o[e >> 2] |= (b & 0xffu) << (8 * (e & 3));
          } else {
            hi = th > hi ? th : hi;
          }
        }
        if (STRETCH) L.store(out + (int64_t)t * frame_elems, k, o);
      }
    }
    if (!STRETCH) {
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) { const double o2 = __shfl_xor(hi, off, 64); hi = o2 > hi ? o2 : hi; }
      if ((threadIdx.x & 63) == 0) part[(int64_t)t * n_ws + ws] = hi;
    }
    L.add(f + (int64_t)(t - first) * frame_elems, -1);              // drop the oldest frame of the window
  }
}

__global__ __launch_bounds__(256) void rolling_fold_kernel(const double* __restrict__ part, int n_ws, double* __restrict__ frame_max) {
  const int t = blockIdx.x;
  double hi = 0.0;
  for (int i = threadIdx.x; i < n_ws; i += 256) { const double x = part[(int64_t)t * n_ws + i]; hi = x > hi ? x : hi; }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) { const double o2 = __shfl_xor(hi, off, 64); hi = o2 > hi ? o2 : hi; }
  __shared__ double red[4];
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = hi;
  __syncthreads();
  if (threadIdx.x == 0) frame_max[t] = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
}

size_t reduce_rolling_scratch_bytes(int64_t frame_elems, int n_frames) {
  const int64_t n_slices = (frame_elems + NORM_SLICE - 1) / NORM_SLICE;
  return ((size_t)n_frames * (size_t)(4 * n_slices) + (size_t)n_frames + 1) * sizeof(double);   // partial maxima, frame maxima, flag
}

// any frame size; scratch: reduce_rolling_scratch_bytes()
hipError_t launch_reduce_rolling(const uint8_t* frames, int64_t frame_elems, int n_frames, int samples, double* scratch,
                                 uint8_t* out, hipStream_t s) {
  if (n_frames <= 0 || frame_elems <= 0) return hipSuccess;
  const int n_slices = (int)((frame_elems + NORM_SLICE - 1) / NORM_SLICE), n_ws = 4 * n_slices;
  int n_seg = std::max(1, std::min(n_frames / std::max(1, 4 * samples) + 1, (2048 + n_slices - 1) / n_slices));
  const int seg_len = (n_frames + n_seg - 1) / n_seg;
  n_seg = (n_frames + seg_len - 1) / seg_len;
  double* part = scratch;
  double* fm = scratch + (size_t)n_frames * n_ws;
  int* flag = reinterpret_cast<int*>(fm + n_frames);
  hipError_t e = hipMemsetAsync(fm, 0, (size_t)(n_frames + 1) * sizeof(double), s);   // frame maxima and the flag
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(rolling_check_kernel, dim3(std::min(64, (255 * samples + 256) / 256)), dim3(256), 0, s, samples, flag);
  hipLaunchKernelGGL(rolling_kernel<false>, dim3(n_slices, n_seg), dim3(256), 0, s, frames, frame_elems, n_frames, samples, seg_len, part, n_ws,
                     (const double*)nullptr, (const int*)flag, (uint8_t*)nullptr);
  if (n_frames >= samples)
    hipLaunchKernelGGL(rolling_fold_kernel, dim3(n_frames - (samples - 1)), dim3(256), 0, s, part + (size_t)(samples - 1) * n_ws, n_ws, fm + (samples - 1));
  hipLaunchKernelGGL(rolling_kernel<true>, dim3(n_slices, n_seg), dim3(256), 0, s, frames, frame_elems, n_frames, samples, seg_len, part, n_ws,
                     (const double*)fm, (const int*)flag, out);
  return hipGetLastError();
}

hipError_t launch_time_diff(const void* frames, int dtype, int64_t frame_elems, int64_t n_frames, float thres, int use_abs,
                            float* out, hipStream_t s) {
  const int64_t n_out = (n_frames - 1) * frame_elems;
  if (n_out <= 0) return hipSuccess;
  const unsigned blocks = (unsigned)std::min<int64_t>((n_out + 255) / 256, 256 * 16);
  if (dtype == 0 && frame_elems % 4 == 0 && (reinterpret_cast<uintptr_t>(frames) & 3) == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0) {
    static const bool flat = getenv("LSPIV_TIME_DIFF_FLAT") != nullptr;   // A/B: the flat kernel (two loads per store)
    if (!flat) {
      constexpr int S = 8;
      const int64_t fq = frame_elems / 4;
      hipLaunchKernelGGL(time_diff_u8x4_walk_kernel<S>, dim3((unsigned)((fq + 255) / 256), (unsigned)((n_frames - 1 + S - 1) / S)), dim3(256), 0, s,
                         (const uint32_t*)frames, fq, (int)(n_frames - 1), thres, use_abs, reinterpret_cast<f32x4*>(out));
      return hipGetLastError();
    }
    const unsigned qb = (unsigned)std::min<int64_t>((n_out / 4 + 255) / 256, 256 * 32);
    hipLaunchKernelGGL(time_diff_u8x4_kernel, dim3(qb), dim3(256), 0, s, (const uint32_t*)frames, frame_elems / 4, n_out / 4, thres, use_abs,
                       reinterpret_cast<f32x4*>(out));
    return hipGetLastError();
  }
  switch (dtype) {
    case 0: hipLaunchKernelGGL(time_diff_kernel<uint8_t>, dim3(blocks), dim3(256), 0, s, (const uint8_t*)frames, frame_elems, n_out, thres, use_abs, out); break;
    case 1: hipLaunchKernelGGL(time_diff_kernel<float>, dim3(blocks), dim3(256), 0, s, (const float*)frames, frame_elems, n_out, thres, use_abs, out); break;
    case 2: hipLaunchKernelGGL(time_diff_kernel<double>, dim3(blocks), dim3(256), 0, s, (const double*)frames, frame_elems, n_out, thres, use_abs, out); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

hipError_t launch_time_range(const void* frames, int dtype, int64_t frame_elems, int64_t n_frames, void* out, hipStream_t s) {
  if (frame_elems <= 0 || n_frames <= 0) return hipSuccess;
  const unsigned blocks = (unsigned)std::min<int64_t>((frame_elems + 255) / 256, 256 * 32);
  switch (dtype) {
    case 0: {
      const uint8_t* f = (const uint8_t*)frames;
      const int64_t n_vec = (frame_elems % 16 == 0 && (reinterpret_cast<uintptr_t>(f) & 15) == 0) ? frame_elems / 16 : 0;
      if (n_vec) {
        const unsigned vb = (unsigned)std::min<int64_t>((n_vec + 255) / 256, 256 * 32);
        hipLaunchKernelGGL(time_range_u8x16_kernel, dim3(vb), dim3(256), 0, s, f, frame_elems, n_frames, (uint8_t*)out);
      } else {
        hipLaunchKernelGGL(time_range_kernel<uint8_t>, dim3(blocks), dim3(256), 0, s, f, frame_elems, n_frames, (uint8_t*)out);
      }
      break;
    }
    case 1: hipLaunchKernelGGL(time_range_kernel<float>, dim3(blocks), dim3(256), 0, s, (const float*)frames, frame_elems, n_frames, (float*)out); break;
    case 2: hipLaunchKernelGGL(time_range_kernel<double>, dim3(blocks), dim3(256), 0, s, (const double*)frames, frame_elems, n_frames, (double*)out); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

hipError_t launch_minmax(const float* in, int64_t n, float lo, float hi, float* out, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  const unsigned blocks = (unsigned)std::min<int64_t>((n + 255) / 256, 256 * 16);
  hipLaunchKernelGGL(minmax_kernel, dim3(blocks), dim3(256), 0, s, in, n, lo, hi, out);
  return hipGetLastError();
}

size_t normalize_part_bytes(int64_t frame_elems, int n_frames) {
  const int64_t n_slices = (frame_elems + NORM_SLICE - 1) / NORM_SLICE;
  // + the one-pass kernel's counters (one per frame) and its flag, behind the pairs
  return (size_t)n_frames * (size_t)(4 * n_slices) * 2 * sizeof(float) + ((size_t)n_frames * NORM1_SLOTS + 64) * sizeof(unsigned);
}

hipError_t launch_sample_mean(const uint8_t* frames, int64_t frame_elems, int n_frames, int interval, float* d_mean, hipStream_t s) {
  if (n_frames <= 0 || frame_elems <= 0) return hipSuccess;
  hipLaunchKernelGGL(sample_mean_kernel, dim3((unsigned)((frame_elems + 255) / 256)), dim3(256), 0, s, frames, frame_elems,
                     n_frames, interval, d_mean);
  return hipGetLastError();
}

hipError_t launch_normalize(const uint8_t* frames, int64_t frame_elems, int n_frames, int interval, float* d_mean,
                            int* d_mn, int* d_mx, float* d_part, uint8_t* out, hipStream_t s) {
  const hipError_t e = launch_sample_mean(frames, frame_elems, n_frames, interval, d_mean, s);
  if (e != hipSuccess) return e;
  return launch_normalize_apply(frames, frame_elems, n_frames, d_mean, d_mn, d_mx, d_part, out, s);
}

// blocks of norm_onepass_kernel the current device holds at once (its blocks wait for each other: all of them must be resident)
static int onepass_capacity() {
  static thread_local int cap_dev = -1, cap = 0;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 0;
  if (dev != cap_dev) {
    int per_cu = 0, cus = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, norm_onepass_kernel, 256, 0) != hipSuccess ||
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) { (void)hipGetLastError(); per_cu = 0; }
    cap = per_cu * cus; cap_dev = dev;
  }
  return cap;
}

// passes 2 and 3 on their own: per-frame statistics only, so a stack may be normalised in pieces against one mean plane
hipError_t launch_normalize_apply(const uint8_t* frames, int64_t frame_elems, int n_frames, const float* d_mean, int* d_mn,
                                  int* d_mx, float* d_part, uint8_t* out, hipStream_t s) {
  if (n_frames <= 0 || frame_elems <= 0) return hipSuccess;
  hipError_t e = hipMemsetAsync(d_mn, 0x7f, (size_t)n_frames * sizeof(int), s);   // 0x7f7f7f7f: a huge positive float
  if (e != hipSuccess) return e;
  e = hipMemsetAsync(d_mx, 0x80, (size_t)n_frames * sizeof(int), s);               // 0x80808080: a negative ordinal
  if (e != hipSuccess) return e;
  // 64 blocks per frame: enough to stream, few enough that the per-block atomics stay cheap.  (Running the two passes
  // chunk by chunk so that the second one finds the frames in the Infinity Cache was tried: slower, 1.05 vs 0.83 ms.)
  // space-major passes (mean plane in registers): LSPIV_NORM_FRAME_MAJOR=1 keeps the frame-major kernels for A/B
  static const bool frame_major = getenv("LSPIV_NORM_FRAME_MAJOR") != nullptr;
  if (!frame_major && d_part && frame_elems % 16 == 0 && (reinterpret_cast<uintptr_t>(frames) & 15) == 0 &&
      (reinterpret_cast<uintptr_t>(out) & 15) == 0) {
    const int n_slices = (int)((frame_elems + NORM_SLICE - 1) / NORM_SLICE), n_ws = 4 * n_slices;
    int n_seg = std::max(1, std::min(n_frames, (2048 + n_slices - 1) / n_slices));
    const int seg_len = (n_frames + n_seg - 1) / n_seg;
    n_seg = (n_frames + seg_len - 1) / seg_len;
    float* lo = reinterpret_cast<float*>(d_mn);
    float* hi = reinterpret_cast<float*>(d_mx);
    // LSPIV_NORM_ONE_PASS=1 (read per call): one pass over the frames when every slice's block is resident at once
    // (norm_onepass_kernel), the two passes behind it guarded by its flag.  NOT the default: 0.94 ms per 201 frames against 0.32.
    const bool one_pass = getenv("LSPIV_NORM_ONE_PASS") != nullptr;
    const int give_up = getenv("LSPIV_NORM_ONEPASS_FAIL") ? 2 : 0;               // test hook: the exchange "times out" at once
    static const int divide = getenv("LSPIV_NORM_DIVIDE") != nullptr;   // A/B: the division for every pixel (the round-5 pass)
    const int* guard = nullptr;
    if (one_pass && n_slices <= onepass_capacity()) {
      unsigned* state = reinterpret_cast<unsigned*>(d_part + (size_t)n_frames * n_ws * 2);
      e = hipMemsetAsync(state, 0, ((size_t)n_frames * NORM1_SLOTS + 64) * sizeof(unsigned), s);
      if (e != hipSuccess) return e;
      int* fail = reinterpret_cast<int*>(state + (size_t)n_frames * NORM1_SLOTS);
      hipLaunchKernelGGL(norm_onepass_kernel, dim3(n_slices), dim3(256), 0, s, frames, d_mean, frame_elems, n_frames, d_part, state, fail, lo, hi, out, divide | give_up);
      guard = fail;
    }
    hipLaunchKernelGGL(norm_minmax_space_kernel, dim3(n_slices, n_seg), dim3(256), 0, s, frames, d_mean, frame_elems, n_frames, seg_len, d_part, n_ws, guard);
    hipLaunchKernelGGL(norm_fold_kernel, dim3(n_frames), dim3(256), 0, s, d_part, n_ws, lo, hi, guard);
    hipLaunchKernelGGL(norm_stretch_space_kernel, dim3(n_slices, n_seg), dim3(256), 0, s, frames, d_mean, frame_elems, n_frames, seg_len, lo, hi, out, divide, guard);
    return hipGetLastError();
  }
  const bool vec = frame_elems % 4 == 0 && (reinterpret_cast<uintptr_t>(frames) & 3) == 0 && (reinterpret_cast<uintptr_t>(out) & 3) == 0;
  const int64_t per = vec ? frame_elems / 4 : frame_elems;
  const unsigned bx = (unsigned)std::min<int64_t>((per + 255) / 256, 64);
  if (vec) {
    hipLaunchKernelGGL(frame_minmax_kernel<4>, dim3(bx, n_frames), dim3(256), 0, s, frames, d_mean, frame_elems, 0, d_mn, d_mx);
    hipLaunchKernelGGL(normalize_kernel<4>, dim3(bx, n_frames), dim3(256), 0, s, frames, d_mean, frame_elems, 0, d_mn, d_mx, out);
  } else {
    hipLaunchKernelGGL(frame_minmax_kernel<1>, dim3(bx, n_frames), dim3(256), 0, s, frames, d_mean, frame_elems, 0, d_mn, d_mx);
    hipLaunchKernelGGL(normalize_kernel<1>, dim3(bx, n_frames), dim3(256), 0, s, frames, d_mean, frame_elems, 0, d_mn, d_mx, out);
  }
  return hipGetLastError();
}

}  // namespace lspiv

// ---- Gaussian blur / band-pass edge filter (Frames.smooth, Frames.edge_detect) -----------------------------------------
// pyorc calls cv2.GaussianBlur(img.astype("float32"), (k, k), 0) (pyorc/cv.py:142-183).  OpenCV is a third-party
// dependency that is absent here; its published algorithm is restated (oracle/filters_oracle.py): fixed coefficient
// tables for k <= 7, separable float32 filter, rows first, BORDER_REFLECT_101, symmetric evaluation
// k0 x0 + sum_j kj (x[-j] + x[+j]).  One fused kernel per call: a 16 x 64 output tile, its halo staged in LDS once,
// row pass into a second LDS buffer, column pass to HBM -- each frame is read once and written once.
// EDGE: out = blur(kernel B) - blur(kernel A) from the same staged tile (edge_detect's band filter).
namespace lspiv {

constexpr int BLUR_TW = 64, BLUR_MAXR = 15;   // tile width; tile height TH = 64 (unrolled radii) or 16 (run-time radii)

struct BlurTaps {
  int r;                        // radius, ksize = 2 r + 1
  float k[BLUR_MAXR + 1];       // k[0] centre, k[j] = coefficient at distance j
};

__device__ __forceinline__ int reflect101(int i, int n) {
  if (n == 1) return 0;
  // one reflection covers every halo that is narrower than the frame; the loop only runs for frames smaller than it
  while (i < 0 || i >= n) i = i < 0 ? -i : 2 * (n - 1) - i;
  return i;
}

// Frames.minmax fused into the filter's store (round 6: the recipe's edge_detect -> minmax, a pass over a float32 stack of its own
// otherwise): np.maximum(np.minimum(x, hi), lo), NaN propagates -- minmax_kernel's expression.  Without limits (-inf, +inf) nothing is done.
struct BlurClip {
  float lo, hi;
  bool on;
};
__device__ __forceinline__ float blur_clip(float x, const BlurClip& c) {
  if (!c.on) return x;                                   // (uniform)
  const float a = (x != x) ? x : (x < c.hi ? x : c.hi);
  return (a != a) ? a : (a > c.lo ? a : c.lo);
}

// symmetric taps around c with element stride S; R > 0: compile-time radius (unrolled), R == 0: run-time radius
template <int R, int S>
__device__ __forceinline__ float taps(const float* c, const BlurTaps& k) {
  float s = c[0] * k.k[0];
  if (R > 0) {
#pragma unroll
    for (int j = 1; j <= R; ++j) s += k.k[j] * (c[-j * S] + c[j * S]);
  } else {
    for (int j = 1; j <= k.r; ++j) s += k.k[j] * (c[-j * S] + c[j * S]);
  }
  return s;
}

template <typename T, bool EDGE, int RA, int RB, int BLUR_TH>
__global__ __launch_bounds__(256) void blur_kernel(const T* __restrict__ frames, int H, int W, BlurTaps ka, BlurTaps kb,
                                                   float* __restrict__ out, BlurClip clip) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int R = EDGE ? (RB > 0 ? RB : kb.r) : (RA > 0 ? RA : ka.r);   // halo = the larger radius
  const int tw = BLUR_TW + 2 * R, th = BLUR_TH + 2 * R;
  float* tile = lds;                                   // th x tw
  float* rowa = tile + th * tw;                        // th x BLUR_TW   (kernel A row pass)
  float* rowb = rowa + th * BLUR_TW;                   // th x BLUR_TW   (kernel B row pass, EDGE only)
  const int x0 = blockIdx.x * BLUR_TW, y0 = blockIdx.y * BLUR_TH;
  const T* img = frames + (int64_t)blockIdx.z * H * W;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  // stage the tile: a wave per row, lanes along x; the two source columns of a lane are resolved once
  const int cx0 = reflect101(x0 + lane - R, W);
  const int cx1 = lane + 64 < tw ? reflect101(x0 + lane + 64 - R, W) : -1;
  for (int ty = wv; ty < th; ty += 4) {
    const T* row = img + (int64_t)reflect101(y0 + ty - R, H) * W;
    tile[ty * tw + lane] = to_f32(row[cx0]);
    if (cx1 >= 0) tile[ty * tw + lane + 64] = to_f32(row[cx1]);
  }
  __syncthreads();
  for (int ty = wv; ty < th; ty += 4) {
    const float* c = tile + ty * tw + lane + R;
    rowa[ty * BLUR_TW + lane] = taps<RA, 1>(c, ka);
    if (EDGE) rowb[ty * BLUR_TW + lane] = taps<RB, 1>(c, kb);
  }
  __syncthreads();
  const int x = x0 + lane;
  if (x >= W) return;
  float* dst = out + (int64_t)blockIdx.z * H * W + x;
  // a wave owns BLUR_TH / 4 consecutive rows: unrolled, the LDS column reads are shared between neighbouring outputs
#pragma unroll
  for (int i = 0; i < BLUR_TH / 4; ++i) {
    const int ty = wv * (BLUR_TH / 4) + i;
    const int y = y0 + ty;
    if (y >= H) break;
    float res = taps<RA, BLUR_TW>(rowa + (ty + R) * BLUR_TW + lane, ka);
    if (EDGE) res = taps<RB, BLUR_TW>(rowb + (ty + R) * BLUR_TW + lane, kb) - res;
    dst[(int64_t)y * W] = blur_clip(res, clip);
  }
}

// Unrolled radii (<= 3, i.e. pyorc's wdw 1..3): one WAVE per 64-column x BLUR_TS-row strip, no block barriers.  Rows
// stream through: global row -> wave-private LDS row (for the x neighbours) -> row filter -> a register window of the
// last 2R+1 row-filtered values per lane -> column filter -> HBM.  Every pixel is loaded once (+ 2R halo rows per
// strip), touches LDS once, and the fully unrolled row loop lets the compiler issue the global loads far ahead.
constexpr int BLUR_TS = 32;

template <typename T, bool EDGE, int RA, int RB>
__global__ __launch_bounds__(256) void blur_strip_kernel(const T* __restrict__ frames, int H, int W, BlurTaps ka,
                                                         BlurTaps kb, float* __restrict__ out, BlurClip clip) {
  constexpr int R = EDGE ? RB : RA;
  constexpr int TWW = BLUR_TW + 2 * R;
  __shared__ float rowbuf[4][TWW + 2];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int x0 = blockIdx.x * BLUR_TW, y0 = (blockIdx.y * 4 + wv) * BLUR_TS;
  if (y0 >= H) return;
  const T* img = frames + (int64_t)blockIdx.z * H * W;
  const int x = x0 + lane;
  float* dst = out + (int64_t)blockIdx.z * H * W + x;
  const int cx0 = reflect101(x - R, W);
  const int cx1 = lane < 2 * R ? reflect101(x + 64 - R, W) : 0;
  float* buf = rowbuf[wv];
  float wa[2 * R + 1], wb[2 * R + 1];
  T p0[BLUR_TS + 2 * R], p1[BLUR_TS + 2 * R];   // every global load of the strip is issued before the first use
#pragma unroll
  for (int i = 0; i < BLUR_TS + 2 * R; ++i) {
    const T* row = img + (int64_t)reflect101(y0 - R + i, H) * W;
    p0[i] = row[cx0];
    p1[i] = row[cx1];
  }
#pragma unroll
  for (int i = 0; i < BLUR_TS + 2 * R; ++i) {
    __builtin_amdgcn_wave_barrier();  // same wave: LDS ops execute in order, this only pins the compiler
    buf[lane] = to_f32(p0[i]);
    if (lane < 2 * R) buf[lane + 64] = to_f32(p1[i]);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int j = 0; j < 2 * R; ++j) { wa[j] = wa[j + 1]; wb[j] = wb[j + 1]; }
    wa[2 * R] = taps<RA, 1>(buf + lane + R, ka);
    if (EDGE) wb[2 * R] = taps<RB, 1>(buf + lane + R, kb);
    if (i >= 2 * R) {
      const int y = y0 + i - 2 * R;
      float res = wa[R] * ka.k[0];
#pragma unroll
      for (int j = 1; j <= RA; ++j) res += ka.k[j] * (wa[R - j] + wa[R + j]);
      if (EDGE) {
        float sb = wb[R] * kb.k[0];
#pragma unroll
        for (int j = 1; j <= RB; ++j) sb += kb.k[j] * (wb[R - j] + wb[R + j]);
        res = sb - res;
      }
      if (y < H && x < W) dst[(int64_t)y * W] = blur_clip(res, clip);
    }
  }
}

// uint8 frames, unrolled radii, FOUR columns per lane with an INTEGER row pass (round 6).  What holds blur_strip_kernel at 0.42 of the
// roofline is neither its arithmetic (smooth and edge_detect take the same 0.61 ms) nor HBM: 14.7 M one-byte-per-lane loads and 6.5 M
// 256-byte stores per 201 frames -- the vector-memory path takes a wave instruction every ~16 cycles whatever its width.  Four columns per
// lane cut the instructions by four, but then the float arithmetic (25 instructions per pixel) paces the stores of a wave and, with 100+
// registers, too few waves are resident to keep the store stream fed (measured: every float variant of this layout is slower, docs/
// history.md).  Here the arithmetic shrinks to 11 per pixel and there is no cross-lane traffic:
//  * OpenCV's fixed kernels for k = 3, 5, 7 are (1 2 1) / 4, (1 4 6 4 1) / 16, (2 7 14 18 14 7 2) / 64, and EVERY float32 operation of
//    either pass on a uint8 frame is exact (integers below 2^24 times a power of two): the result is the exact rational, whatever the
//    order or the arithmetic.  The row pass is therefore done on the packed bytes: the lane loads the 12 bytes around its four columns
//    (its neighbours load the same cache lines: no extra HBM traffic), v_alignbyte cuts the shifted dwords out of that string,
//    v_dot4_u32_u8 applies the integer taps: 2 dot4 per output of a 5- or 7-tap kernel, 1 of the 3-tap one.
//  * the column pass runs on integer-valued floats with the taps scaled by 1 / (row divisor x column divisor) -- exact again --, two
//    columns per instruction (v_pk_*_f32 on float2 pairs).
// Same bits as blur_strip_kernel (tests/test_filters.py).  A wave covers 256 columns (eight whole 128-byte lines per stored row) and
// BLUR4_TS rows; EVERY load of the strip is issued before its first store (LSPIV_BLUR4_AHEAD >= the rows of a strip): vmcnt counts
// stores as well, a load issued after a store is only known to have arrived when the store has been acknowledged -- a rolling window
// of 6 rows of loads costs 8 % at 16 rows per strip.  The row-filtered rows sit in a ring of registers (the fully unrolled row loop
// makes every ring index a constant: no moves).  Measured on 201 1080p frames, ms per launch (one-column kernel | this one): k = 3
// 0.589 | 0.426, k = 5 0.592 | 0.421, k = 7 0.676 | 0.410, edge 3 / 5 0.593 | 0.417, 3 / 7 0.689 | 0.426, 5 / 7 0.704 | 0.434; rows
// per strip 4 / 8 / 16 / 24 / 32 / 48 / 64: 0.49 / 0.46 / 0.44 / 0.43 / 0.41 / 0.44 / 0.49 (tools/sessions/r06_ab_blur4.sh).
#ifndef LSPIV_BLUR4_AHEAD
#define LSPIV_BLUR4_AHEAD 64
#endif
#ifndef LSPIV_BLUR4_TS
#define LSPIV_BLUR4_TS 32
#endif
constexpr int BLUR4_TS = LSPIV_BLUR4_TS;                               // rows per strip of the four-column kernel
typedef float f32x2b __attribute__((ext_vector_type(2)));

// integer taps of the kernel of radius R as dot4 weight words: w0 covers the bytes [centre - R, centre - R + 3], w1 the next four
template <int R> struct IntTaps;
template <> struct IntTaps<1> { static constexpr uint32_t w0 = 0x00010201u, w1 = 0u; static constexpr float div = 4.0f; };
template <> struct IntTaps<2> { static constexpr uint32_t w0 = 0x04060401u, w1 = 0x00000001u; static constexpr float div = 16.0f; };
template <> struct IntTaps<3> { static constexpr uint32_t w0 = 0x120e0702u, w1 = 0x0002070eu; static constexpr float div = 64.0f; };

// bytes [o, o + 3] of the 12-byte string pl | pc | pr (o = 0 .. 9; bytes beyond 11 are whatever: their tap weight is 0)
template <int O>
__device__ __forceinline__ uint32_t str_dword(uint32_t pl, uint32_t pc, uint32_t pr) {
  if (O == 0) return pl;
  if (O < 4) return __builtin_amdgcn_alignbyte(pc, pl, O);
  if (O == 4) return pc;
  if (O < 8) return __builtin_amdgcn_alignbyte(pr, pc, O - 4);
  if (O == 8) return pr;
  return __builtin_amdgcn_alignbyte(pr, pr, O - 8);
}
// the row-filtered value (an integer: the taps' divisor is applied by the column pass) of output column E of the lane
template <int R, int E>
__device__ __forceinline__ uint32_t row_taps_int(uint32_t pl, uint32_t pc, uint32_t pr) {
  uint32_t s = __builtin_amdgcn_udot4(str_dword<4 + E - R>(pl, pc, pr), IntTaps<R>::w0, 0u, false);
  if (R > 1) s = __builtin_amdgcn_udot4(str_dword<8 + E - R>(pl, pc, pr), IntTaps<R>::w1, s, false);
  return s;
}

// the rows of one strip.  INTERIOR (wave-uniform): every lane's 12 bytes [xl - 4, xl + 8) lie inside the frame -- one 12-byte load per
// lane and row (neighbouring lanes overlap by 8 bytes: the same cache lines, no extra HBM traffic, and no cross-lane operation at all);
// otherwise (the first and the last strip of a row) the loads are clamped into the row and the lane at the edge mirrors the dword beyond it.
template <bool EDGE, int RA, int RB, bool INTERIOR>
__device__ __forceinline__ void blur4_strip(const uint8_t* __restrict__ img, int H, int W, int y0, int xl, float* __restrict__ dst,
                                            const BlurClip& clip) {
  constexpr int R = EDGE ? RB : RA;
  constexpr int M = 2 * R + 1;
  constexpr int NR = BLUR4_TS + 2 * R;
  constexpr int K = LSPIV_BLUR4_AHEAD < NR ? LSPIV_BLUR4_AHEAD : NR;
  typedef uint32_t u32x3b __attribute__((ext_vector_type(3), aligned(4)));
  // the first and the last strip of a row: every lane loads 12 bytes INSIDE the row (its window shifted by a dword where it reaches over
  // an edge), the lane at the edge rebuilds the dword beyond it by BORDER_REFLECT_101 from the bytes it has: left of column 0 the columns
  // 4 3 2 1, right of column W - 1 the columns W-2 W-3 W-4 W-5 (v_perm_b32) -- W % 4 == 0 and W >= 12, lanes with xl >= W do not write
  const int xc = INTERIOR ? xl - 4 : min(max(xl - 4, 0), W - 12);
  const int sh = xl - 4 - xc;                                          // -4: the lane at the left edge, +4: at the right edge
  u32x3b p[K];                                                         // a rolling window of K rows of loads in flight
  auto load_row = [&](int i, u32x3b& d) {
    const uint8_t* row = img + (int64_t)reflect101(y0 - R + i, H) * W;   // (scalar: the row is wave-uniform)
    d = *reinterpret_cast<const u32x3b*>(row + xc);
  };
  auto fix_edges = [&](uint32_t& pl, uint32_t& pc, uint32_t& pr) {
    if (INTERIOR) return;
    const uint32_t d0 = pl, d1 = pc, d2 = pr;
    pl = sh < 0 ? __builtin_amdgcn_perm(d1, d0, 0x01020304u) : (sh > 0 ? d1 : d0);
    pc = sh < 0 ? d0 : (sh > 0 ? d2 : d1);
    pr = sh < 0 ? d1 : (sh > 0 ? __builtin_amdgcn_perm(d2, d1, 0x03040506u) : d2);
  };
#pragma unroll
  for (int i = 0; i < K; ++i) load_row(i, p[i]);
  // column taps as exact floats: kernel weight / (row divisor x column divisor)
  constexpr float da = IntTaps<RA>::div * IntTaps<RA>::div, db = IntTaps<R>::div * IntTaps<R>::div;
  float ca[RA + 1], cb[R + 1];
#pragma unroll
  for (int j = 0; j <= RA; ++j) ca[j] = (float)((IntTaps<RA>::w0 >> (8 * (RA - j))) & 0xffu) / da;     // tap at distance j: byte RA - j of w0
#pragma unroll
  for (int j = 0; j <= R; ++j) cb[j] = (float)((IntTaps<R>::w0 >> (8 * (R - j))) & 0xffu) / db;
  const bool writes = xl < W;                                          // (W % 4 == 0: then all four columns are inside)
  f32x2b wa[2][M], wb[2][M];                                           // a ring of the last M row-filtered rows, columns (0, 1) and (2, 3): row i sits in slot i % M
#pragma unroll
  for (int i = 0; i < NR; ++i) {
    uint32_t pl = p[i % K][0], pc = p[i % K][1], pr = p[i % K][2];
    fix_edges(pl, pc, pr);
    if (i + K < NR) load_row(i + K, p[i % K]);                         // the slot is free: the row K ahead goes out
    wa[0][i % M] = f32x2b{(float)row_taps_int<RA, 0>(pl, pc, pr), (float)row_taps_int<RA, 1>(pl, pc, pr)};
    wa[1][i % M] = f32x2b{(float)row_taps_int<RA, 2>(pl, pc, pr), (float)row_taps_int<RA, 3>(pl, pc, pr)};
    if (EDGE) {
      wb[0][i % M] = f32x2b{(float)row_taps_int<R, 0>(pl, pc, pr), (float)row_taps_int<R, 1>(pl, pc, pr)};
      wb[1][i % M] = f32x2b{(float)row_taps_int<R, 2>(pl, pc, pr), (float)row_taps_int<R, 3>(pl, pc, pr)};
    }
    if (i >= 2 * R) {
      const int y = y0 + i - 2 * R, c = i - R;                         // the centre row of this output
      f32x4 v;
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        f32x2b res = wa[q][c % M] * ca[0];
#pragma unroll
        for (int j = 1; j <= RA; ++j) res += ca[j] * (wa[q][(c - j) % M] + wa[q][(c + j) % M]);
        if (EDGE) {
          f32x2b sb = wb[q][c % M] * cb[0];
#pragma unroll
          for (int j = 1; j <= R; ++j) sb += cb[j] * (wb[q][(c - j) % M] + wb[q][(c + j) % M]);
          res = sb - res;
        }
        v[2 * q] = res[0]; v[2 * q + 1] = res[1];
      }
      if (clip.on) {                                                   // (uniform)
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = blur_clip(v[e], clip);
      }
      if (writes && y < H) *reinterpret_cast<f32x4*>(dst + (int64_t)y * W) = v;      // (non-temporal stores: no difference)
    }
  }
}

template <bool EDGE, int RA, int RB>
__global__ __launch_bounds__(256) void blur_strip4_kernel(const uint8_t* __restrict__ frames, int H, int W, float* __restrict__ out,
                                                          BlurClip clip) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int y0 = (blockIdx.y * 4 + wv) * BLUR4_TS;
  if (y0 >= H) return;
  const uint8_t* img = frames + (int64_t)blockIdx.z * H * W;
  const int x0 = (int)blockIdx.x * 256, xl = x0 + 4 * lane;            // this lane's first column
  float* dst = out + (int64_t)blockIdx.z * H * W + xl;
  if (x0 - 4 >= 0 && x0 + 260 <= W) blur4_strip<EDGE, RA, RB, true>(img, H, W, y0, xl, dst, clip);
  else blur4_strip<EDGE, RA, RB, false>(img, H, W, y0, xl, dst, clip);
}

// Run-time radii (> 3): the same streaming structure, with the last 2R+1 row-filtered values of every column kept
// in a wave-private LDS ring instead of registers (the ring slot is wave-uniform, the column is the lane: no cross-lane
// traffic besides the staged input row).  One row of loads is kept in flight ahead of the row being filtered.
template <typename T, bool EDGE>
__global__ __launch_bounds__(256) void blur_ring_kernel(const T* __restrict__ frames, int H, int W, BlurTaps ka, BlurTaps kb,
                                                        float* __restrict__ out, BlurClip clip) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int R = EDGE ? kb.r : ka.r, M = 2 * R + 1;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int per_wave = (BLUR_TW + 2 * R) + (EDGE ? 2 : 1) * M * 64;
  float* buf = lds + wv * per_wave;              // staged input row, 64 + 2R samples
  float* ring_a = buf + BLUR_TW + 2 * R;         // M rows of 64 row-filtered values (kernel A)
  float* ring_b = ring_a + M * 64;               // (kernel B, EDGE only)
  const int x0 = blockIdx.x * BLUR_TW, y0 = (blockIdx.y * 4 + wv) * BLUR_TS;
  if (y0 >= H) return;
  const T* img = frames + (int64_t)blockIdx.z * H * W;
  const int x = x0 + lane;
  float* dst = out + (int64_t)blockIdx.z * H * W + x;
  const int cx0 = reflect101(x - R, W);
  const int cx1 = lane < 2 * R ? reflect101(x + 64 - R, W) : 0;
  const int n_rows = BLUR_TS + 2 * R;
  const T* row = img + (int64_t)reflect101(y0 - R, H) * W;
  T n0 = row[cx0], n1 = row[cx1];
  int slot = 0;
  for (int i = 0; i < n_rows; ++i) {
    const float v0 = to_f32(n0), v1 = to_f32(n1);
    if (i + 1 < n_rows) {                          // next row's loads go out before this row is filtered
      row = img + (int64_t)reflect101(y0 - R + i + 1, H) * W;
      n0 = row[cx0];
      n1 = row[cx1];
    }
    __builtin_amdgcn_wave_barrier();
    buf[lane] = v0;
    if (lane < 2 * R) buf[lane + 64] = v1;
    __builtin_amdgcn_wave_barrier();
    ring_a[slot * 64 + lane] = taps<0, 1>(buf + lane + R, ka);
    if (EDGE) ring_b[slot * 64 + lane] = taps<0, 1>(buf + lane + R, kb);
    if (i >= 2 * R) {
      const int y = y0 + i - 2 * R;
      int c = slot - R;                            // ring slot of the centre row
      c += c < 0 ? M : 0;
      auto column = [&](const float* ring, const BlurTaps& k) {
        float acc = ring[c * 64 + lane] * k.k[0];
        int lo = c, hi = c;
        for (int j = 1; j <= k.r; ++j) {
          lo = lo == 0 ? M - 1 : lo - 1;
          hi = hi == M - 1 ? 0 : hi + 1;
          acc += k.k[j] * (ring[lo * 64 + lane] + ring[hi * 64 + lane]);
        }
        return acc;
      };
      float res = column(ring_a, ka);
      if (EDGE) res = column(ring_b, kb) - res;
      if (y < H && x < W) dst[(int64_t)y * W] = blur_clip(res, clip);
    }
    slot = slot == M - 1 ? 0 : slot + 1;
  }
}

// Run-time radii, streaming with a REGISTER ring (round 6; the reference's own user guide filters with wdw 2 | 4 and 6 | 10: k = 5 | 9
// and 13 | 21): blur_strip_kernel's structure for any radius up to RMAX (template: 5 / 10) -- a wave walks down a 64-column strip, a row goes global -> wave-private LDS row -> the lane's 2R + 1 neighbours into
// registers (the row pass of kernel A and of kernel B from the same registers) -> slot i mod (2 RMAX + 1) of a ring of REGISTERS
// (the row loop is unrolled by the ring's period, so every slot is a compile-time register; outputs trail the rows by RMAX whatever
// the radius) -> column pass from registers -> HBM.  1 + 2R LDS reads per pixel where the LDS-ring kernel needs 2 (2R + 1) per filter
// kernel; no block barrier.  Taps beyond the run-time radius are skipped by wave-uniform branches.  The same expressions and order as
// taps<>: the same bits as blur_ring_kernel (tests/test_filters.py).  Per 200 1080p uint8 frames, LDS-ring kernel | this one: k = 5 | 9
// 1.90 | 1.06 ms, k = 11 1.47 | 1.03, k = 13 | 21 5.42 | 2.12, k = 5 | 15 2.93 | 2.06.  (An RMAX = 15 instance: the compiler does not unroll the
// period of 31 rows and indexes the ring dynamically -- k = 23 / 27 / 31 1.99 / 2.08 / 2.20 ms per 101 frames against 1.30 / 1.60 / 2.16 of the
// LDS-ring kernel, edge 23 | 31 4.20 against 5.71, 9 | 23 3.61 against 2.67: windows above 10 stay with the LDS ring.)  A tile-per-block kernel with four columns per lane in the row pass and
// four rows per lane in the column pass (1.5 + 5 ... 10 LDS reads per pixel, two block barriers, 40 - 55 KB of LDS) was bit-identical and
// slower than this one everywhere (2.12 / 1.53 / 3.60 / 3.27 ms): removed.
constexpr int BLURR_TS = 64;

template <int RMAX>
__device__ __forceinline__ float ring_taps(const float (&w)[2 * RMAX + 1], int c, const BlurTaps& k) {   // c: compile-time after unrolling
  constexpr int M = 2 * RMAX + 1;
  float s = w[c] * k.k[0];
#pragma unroll
  for (int j = 1; j <= RMAX; ++j)
    if (j <= k.r) s += k.k[j] * (w[(c - j + M) % M] + w[(c + j) % M]);     // (earlier row + later row, like the LDS ring)
  return s;
}

template <typename T, bool EDGE, int RMAX>
__global__ __launch_bounds__(256) void blur_stripr_kernel(const T* __restrict__ frames, int H, int W, BlurTaps ka, BlurTaps kb,
                                                          float* __restrict__ out, BlurClip clip) {
  constexpr int M = 2 * RMAX + 1;
  __shared__ float rowbuf[4][BLUR_TW + 2 * RMAX + 2];
  const int R = EDGE ? kb.r : ka.r;                                     // the columns staged to either side: the larger radius
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int x0 = blockIdx.x * BLUR_TW, y0 = (blockIdx.y * 4 + wv) * BLURR_TS;
  if (y0 >= H) return;
  const T* img = frames + (int64_t)blockIdx.z * H * W;
  const int x = x0 + lane;
  float* dst = out + (int64_t)blockIdx.z * H * W + x;
  const int cx0 = reflect101(x - R, W);
  const int cx1 = lane < 2 * R ? reflect101(x + 64 - R, W) : 0;
  float* buf = rowbuf[wv];
  const int n_rows = min(BLURR_TS, H - y0) + 2 * RMAX;                  // rows y0 - RMAX .. : an output row trails its last input row by RMAX
  const T* row = img + (int64_t)reflect101(y0 - RMAX, H) * W;
  T n0 = row[cx0], n1 = row[cx1];
  float wa[M], wb[M];
  for (int base = 0; base < n_rows; base += M) {
#pragma unroll
    for (int ii = 0; ii < M; ++ii) {
      const int i = base + ii;
      if (i < n_rows) {                                                  // (wave-uniform)
        const float v0 = to_f32(n0), v1 = to_f32(n1);
        if (i + 1 < n_rows) {                                            // next row's loads go out before this row is filtered
          row = img + (int64_t)reflect101(y0 - RMAX + i + 1, H) * W;
          n0 = row[cx0];
          n1 = row[cx1];
        }
        __builtin_amdgcn_wave_barrier();
        buf[lane] = v0;
        if (lane < 2 * R) buf[lane + 64] = v1;
        __builtin_amdgcn_wave_barrier();
        float a[M];                                                      // the lane's neighbours: a[RMAX + d] = column x + d
        a[RMAX] = buf[lane + R];
#pragma unroll
        for (int j = 1; j <= RMAX; ++j)
          if (j <= R) { a[RMAX - j] = buf[lane + R - j]; a[RMAX + j] = buf[lane + R + j]; }
        {
          float sa = a[RMAX] * ka.k[0];
#pragma unroll
          for (int j = 1; j <= RMAX; ++j)
            if (j <= ka.r) sa += ka.k[j] * (a[RMAX - j] + a[RMAX + j]);
          wa[ii] = sa;
          if (EDGE) {
            float sb = a[RMAX] * kb.k[0];
#pragma unroll
            for (int j = 1; j <= RMAX; ++j)
              if (j <= kb.r) sb += kb.k[j] * (a[RMAX - j] + a[RMAX + j]);
            wb[ii] = sb;
          }
        }
        if (i >= 2 * RMAX) {
          const int y = y0 + i - 2 * RMAX;
          const int c = (ii + RMAX + 1) % M;                             // slot of the centre row i - RMAX (compile-time)
          float res = ring_taps<RMAX>(wa, c, ka);
          if (EDGE) res = ring_taps<RMAX>(wb, c, kb) - res;
          if (y < H && x < W) dst[(int64_t)y * W] = blur_clip(res, clip);
        }
      }
    }
  }
}

// getGaussianKernel(ksize, sigma <= 0, CV_32F)
static BlurTaps make_taps(int ksize) {
  BlurTaps t;
  t.r = ksize / 2;
  static const float t3[] = {0.5f, 0.25f}, t5[] = {0.375f, 0.25f, 0.0625f}, t7[] = {0.28125f, 0.21875f, 0.109375f, 0.03125f};
  for (int j = 0; j <= BLUR_MAXR; ++j) t.k[j] = 0.0f;
  if (ksize == 1) { t.k[0] = 1.0f; return t; }
  if (ksize == 3 || ksize == 5 || ksize == 7) {
    const float* s = ksize == 3 ? t3 : ksize == 5 ? t5 : t7;
    for (int j = 0; j <= t.r; ++j) t.k[j] = s[j];
    return t;
  }
  const double sigma = 0.3 * ((ksize - 1) * 0.5 - 1.0) + 0.8;
  double sum = 0.0, v[2 * BLUR_MAXR + 1];
  for (int i = 0; i < ksize; ++i) { const double x = i - (ksize - 1) * 0.5; v[i] = std::exp(-(x * x) / (2.0 * sigma * sigma)); sum += v[i]; }
  for (int j = 0; j <= t.r; ++j) t.k[j] = (float)(v[t.r + j] / sum);
  return t;
}

hipError_t launch_blur(const void* frames, int dtype, int n_frames, int H, int W, int ksize_a, int ksize_b, float* out,
                       hipStream_t s) {
  return launch_blur_clip(frames, dtype, n_frames, H, W, ksize_a, ksize_b, -INFINITY, INFINITY, out, s);
}

hipError_t launch_blur_clip(const void* frames, int dtype, int n_frames, int H, int W, int ksize_a, int ksize_b, float lo, float hi,
                            float* out, hipStream_t s) {
  if (n_frames <= 0) return hipSuccess;
  const BlurClip clip{lo, hi, !(lo == -INFINITY && hi == INFINITY)};
  const bool edge = ksize_b > 0;
  const BlurTaps ka = make_taps(ksize_a), kb = edge ? make_taps(ksize_b) : ka;
  const int R = edge ? kb.r : ka.r;
  const bool unrolled = edge ? (kb.r <= 3 && ka.r >= 1 && ka.r < kb.r) : (ka.r >= 1 && ka.r <= 3);
  // uint8 frames under OpenCV's fixed kernels (k = 3, 5, 7): four columns per lane, integer row pass (blur_strip4_kernel);
  // LSPIV_BLUR_ONE_COLUMN=1 keeps the one-column float kernel for A/B
  if (unrolled && dtype == 0 && W % 4 == 0 && W >= 12 && (reinterpret_cast<uintptr_t>(frames) & 3) == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0 &&
      !getenv("LSPIV_BLUR_ONE_COLUMN")) {
    const int strips = (H + BLUR4_TS - 1) / BLUR4_TS;
    const dim3 grid((W + 255) / 256, (strips + 3) / 4, n_frames);
#define LSPIV_S4(E, A, B) hipLaunchKernelGGL((blur_strip4_kernel<E, A, B>), grid, dim3(256), 0, s, (const uint8_t*)frames, H, W, out, clip)
    if (!edge) {
      if (ka.r == 1) LSPIV_S4(false, 1, 1); else if (ka.r == 2) LSPIV_S4(false, 2, 2); else LSPIV_S4(false, 3, 3);
    } else {
      if (kb.r == 2) LSPIV_S4(true, 1, 2); else if (ka.r == 1) LSPIV_S4(true, 1, 3); else LSPIV_S4(true, 2, 3);
    }
#undef LSPIV_S4
    return hipGetLastError();
  }
  if (unrolled) {
    const int strips = (H + BLUR_TS - 1) / BLUR_TS;
    const dim3 grid((W + BLUR_TW - 1) / BLUR_TW, (strips + 3) / 4, n_frames);
#define LSPIV_STRIP4(T, E, A, B) hipLaunchKernelGGL((blur_strip_kernel<T, E, A, B>), grid, dim3(256), 0, s, (const T*)frames, H, W, ka, kb, out, clip)
#define LSPIV_STRIP(T)                                                                     \
  if (!edge) {                                                                             \
    if (ka.r == 1) LSPIV_STRIP4(T, false, 1, 0); else if (ka.r == 2) LSPIV_STRIP4(T, false, 2, 0); else LSPIV_STRIP4(T, false, 3, 0); \
  } else {                                                                                 \
    if (kb.r == 2) LSPIV_STRIP4(T, true, 1, 2); else if (ka.r == 1) LSPIV_STRIP4(T, true, 1, 3); else LSPIV_STRIP4(T, true, 2, 3); \
  }
    switch (dtype) {
      case 0: LSPIV_STRIP(uint8_t) break;
      case 1: LSPIV_STRIP(float) break;
      case 2: LSPIV_STRIP(double) break;
      default: return hipErrorInvalidValue;
    }
#undef LSPIV_STRIP
#undef LSPIV_STRIP4
    return hipGetLastError();
  }
  // run-time radii up to 10 (windows 4 .. 10): the streaming kernel with a register ring (LSPIV_BLUR_RING=1, read per call: the LDS-ring
  // kernel it replaces, for A/B); windows 11 .. 15 stay with the LDS-ring kernel
  if (R <= 10 && !getenv("LSPIV_BLUR_RING") && !getenv("LSPIV_BLUR_BLOCK")) {   // the streaming kernel with a register ring
    const int strips = (H + BLURR_TS - 1) / BLURR_TS;
    const dim3 grid((W + BLUR_TW - 1) / BLUR_TW, (strips + 3) / 4, n_frames);
#define LSPIV_SR(T, E, RM) hipLaunchKernelGGL((blur_stripr_kernel<T, E, RM>), grid, dim3(256), 0, s, (const T*)frames, H, W, ka, kb, out, clip)
#define LSPIV_SRR(T, E) do { if (R <= 5) LSPIV_SR(T, E, 5); else LSPIV_SR(T, E, 10); } while (0)
    switch (dtype) {
      case 0: if (edge) LSPIV_SRR(uint8_t, true); else LSPIV_SRR(uint8_t, false); break;
      case 1: if (edge) LSPIV_SRR(float, true); else LSPIV_SRR(float, false); break;
      case 2: if (edge) LSPIV_SRR(double, true); else LSPIV_SRR(double, false); break;
      default: return hipErrorInvalidValue;
    }
#undef LSPIV_SRR
#undef LSPIV_SR
    return hipGetLastError();
  }
  // larger radii: streaming ring kernel (LSPIV_BLUR_BLOCK=1: the tile-per-block kernel it replaced, for A/B)
  static const bool use_block = getenv("LSPIV_BLUR_BLOCK") != nullptr;
  if (!use_block) {
    const int strips = (H + BLUR_TS - 1) / BLUR_TS;
    const dim3 grid((W + BLUR_TW - 1) / BLUR_TW, (strips + 3) / 4, n_frames);
    const size_t lds = (size_t)4 * ((BLUR_TW + 2 * R) + (edge ? 2 : 1) * (2 * R + 1) * 64) * sizeof(float);
#define LSPIV_RING(T, E) hipLaunchKernelGGL((blur_ring_kernel<T, E>), grid, dim3(256), lds, s, (const T*)frames, H, W, ka, kb, out, clip)
    switch (dtype) {
      case 0: if (edge) LSPIV_RING(uint8_t, true); else LSPIV_RING(uint8_t, false); break;
      case 1: if (edge) LSPIV_RING(float, true); else LSPIV_RING(float, false); break;
      case 2: if (edge) LSPIV_RING(double, true); else LSPIV_RING(double, false); break;
      default: return hipErrorInvalidValue;
    }
#undef LSPIV_RING
    return hipGetLastError();
  }
  const int TH = 16;
  const size_t lds = ((size_t)(TH + 2 * R) * (BLUR_TW + 2 * R) + (size_t)(edge ? 2 : 1) * (TH + 2 * R) * BLUR_TW) * sizeof(float);
  const dim3 grid((W + BLUR_TW - 1) / BLUR_TW, (H + TH - 1) / TH, n_frames);
#define LSPIV_BLUR(T, E) hipLaunchKernelGGL((blur_kernel<T, E, 0, 0, 16>), grid, dim3(256), lds, s, (const T*)frames, H, W, ka, kb, out, clip)
  switch (dtype) {
    case 0: if (edge) LSPIV_BLUR(uint8_t, true); else LSPIV_BLUR(uint8_t, false); break;
    case 1: if (edge) LSPIV_BLUR(float, true); else LSPIV_BLUR(float, false); break;
    case 2: if (edge) LSPIV_BLUR(double, true); else LSPIV_BLUR(double, false); break;
    default: return hipErrorInvalidValue;
  }
#undef LSPIV_BLUR
  return hipGetLastError();
}

}  // namespace lspiv
