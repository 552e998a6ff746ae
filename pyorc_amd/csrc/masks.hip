// the post-PIV masks of ds.velocimetry.mask (pyorc/api/mask.py:147-403) on the
// device-resident result block [v_x | v_y | corr | s2n], each (T, R, C) float32.
//
// The reference writes the masks as xarray expressions; every kernel below restates the numpy calls those dispatch to
// (oracle/mask_oracle.py), operation by operation in float32: sums over time / over the neighbourhood run in the
// reference's order with separate multiplies and adds (FMA contraction is switched off for this file), divisions and
// sqrtf are correctly rounded (hipcc's default; NOT __fsqrt_rn, which is the 1-ulp native sqrt on this target), so
// every mask is bit-identical to the oracle except `angle` (atan2f is not correctly rounded in either library).
// All kernels are HBM streaming: one thread per output element, lanes along the contiguous (R, C) plane, time /
// neighbourhood loops inside the thread.
#include <cmath>

#include "common.h"

// numpy multiplies, then adds: hipcc's default -ffp-contract=fast would fuse them, so contraction is switched off for
// the whole file.  The __fmul_rn / __fadd_rn intrinsics must NOT be used for this: on this target they are plain
// operators defined in a header compiled with contraction on, and they carry that flag into the caller when inlined.
#pragma clang fp contract(off)

namespace {
__device__ __forceinline__ float mul_rn(float a, float b) { return a * b; }
__device__ __forceinline__ float add_rn(float a, float b) { return a + b; }
__device__ __forceinline__ float sub_rn(float a, float b) { return a - b; }
}  // namespace

namespace lspiv {

namespace {

constexpr int kBlock = 256;

__device__ __forceinline__ float speed(float vx, float vy) {
  return sqrtf(add_rn(mul_rn(vx, vx), mul_rn(vy, vy)));   // (v_x**2 + v_y**2) ** 0.5
}
__device__ __forceinline__ float nan0(float x) { return x != x ? 0.0f : x; }

// kinds 0 minmax, 1 angle, 3 corr, 4 s2n: one pass over (T, R, C)
__global__ __launch_bounds__(kBlock) void mask_pointwise_kernel(const float* __restrict__ f, int64_t N, int kind, float p0,
                                                                float p1, uint8_t* __restrict__ mask) {
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < N; i += (int64_t)gridDim.x * kBlock) {
    bool keep;
    if (kind == 0) {
      const float s = speed(f[i], f[N + i]);
      keep = s > p0 && s < p1;
    } else if (kind == 1) {
      keep = fabsf(sub_rn(atan2f(f[i], f[N + i]), p0)) < p1;
    } else {
      keep = f[(kind == 3 ? 2 : 3) * N + i] > p0;
    }
    mask[i] = keep;
  }
}

// np.nanmean / np.nanstd(ddof=0) over time.  The float32 sums must run sequentially in time order (np.add.reduce over
// the outer axis; NaN -> 0 as _replace_nan does), which leaves only one thread per (R, C) cell to do arithmetic --
// 7 854 threads for a 1080p grid, far too few loads in flight.  So a block owns 64 cells: all four waves stream
// 64-row chunks of the time axis into LDS (16 independent coalesced loads per lane and variable), then wave v
// accumulates variable v from LDS in order.  Sums start from -0.0f so that the first addition reproduces numpy's
// "start from the first element" exactly, sign of zero included.
constexpr int kCells = 64, kCh = 64;

// A grid of ONE cell (R = C = 1) is the exception: the time axis is then the array's contiguous inner axis and np.add.reduce sums it
// PAIRWISE (numpy/core/src/umath/loops_utils.h.src, pairwise_sum: below 8 elements in order from 0., up to 128 in eight strided partial
// sums combined as ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7)) plus a tail in order, above that split at n / 2 rounded down to a
// multiple of 8, left + right).  Found by tools/fuzz_rows.py (seed 1108, round 5); restated here with an explicit stack; one thread.
template <bool SQ>
__device__ float np_pairwise_time(const float* __restrict__ col, int64_t stride, int64_t T, float avg, int* cnt) {
  auto get = [&](int64_t t) -> float {
    const float x = col[t * stride];
    if (SQ) { const float d = x != x ? 0.0f : sub_rn(x, avg); return mul_rn(d, d); }
    return nan0(x);
  };
  if (!SQ) { int c = 0; for (int64_t t = 0; t < T; ++t) { const float x = col[t * stride]; c += x == x; } *cnt = c; }
  auto leaf = [&](int64_t lo, int64_t n) -> float {
    if (n < 8) { float r = 0.0f; for (int64_t i = 0; i < n; ++i) r = add_rn(r, get(lo + i)); return r; }
    float r[8];
    for (int j = 0; j < 8; ++j) r[j] = get(lo + j);
    int64_t i = 8;
    for (; i < n - (n % 8); i += 8)
      for (int j = 0; j < 8; ++j) r[j] = add_rn(r[j], get(lo + i + j));
    float res = add_rn(add_rn(add_rn(r[0], r[1]), add_rn(r[2], r[3])), add_rn(add_rn(r[4], r[5]), add_rn(r[6], r[7])));
    for (; i < n; ++i) res = add_rn(res, get(lo + i));
    return res;
  };
  struct Frame { int64_t lo, n; float left; int stage; };
  Frame st[32];   // depth: log2(T / 128) + 1
  int sp = 0;
  st[0] = {0, T, 0.0f, 0};
  float ret = 0.0f;
  while (sp >= 0) {
    Frame& f = st[sp];
    if (f.stage == 0) {
      if (f.n <= 128) { ret = leaf(f.lo, f.n); --sp; continue; }
      f.stage = 1;
      int64_t n2 = f.n / 2; n2 -= n2 % 8;
      st[sp + 1] = {f.lo, n2, 0.0f, 0};
      ++sp;
    } else if (f.stage == 1) {
      f.left = ret;
      f.stage = 2;
      int64_t n2 = f.n / 2; n2 -= n2 % 8;
      st[sp + 1] = {f.lo + n2, f.n - n2, 0.0f, 0};
      ++sp;
    } else {
      ret = add_rn(f.left, ret);
      --sp;
    }
  }
  return ret;
}

template <int NV, bool SQ>
__device__ __forceinline__ void column_pass(const float* __restrict__ f, int64_t var_stride, int64_t T, int64_t n,
                                            int64_t cell, bool valid, float (*tile)[kCh][kCells], float avg, float* acc,
                                            int* cnt) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (n == 1) {   // a single cell: numpy's pairwise order (np_pairwise_time); lane 0 of wave v does variable v
    if (w < NV && lane == 0 && valid) *acc = np_pairwise_time<SQ>(f + w * var_stride, 1, T, avg, cnt);
    return;
  }
  for (int64_t t0 = 0; t0 < T; t0 += kCh) {
    const int rows = (int)(T - t0 < kCh ? T - t0 : kCh);
#pragma unroll 4
    for (int k = w; k < rows; k += 4)
#pragma unroll
      for (int v = 0; v < NV; ++v) tile[v][k][lane] = valid ? f[v * var_stride + (t0 + k) * n + cell] : 0.0f;
    __syncthreads();
    if (w < NV) {
      float a = *acc;
      int c = *cnt;
      for (int k = 0; k < rows; ++k) {
        const float x = tile[w][k][lane];
        if (SQ) {
          const float d = x != x ? 0.0f : sub_rn(x, avg);   // arr - avg, then zero where NaN was
          a = add_rn(a, mul_rn(d, d));
        } else {
          a = add_rn(a, nan0(x));
          c += x == x;
        }
      }
      *acc = a;
      *cnt = c;
    }
    __syncthreads();
  }
}

// kinds 2 count, 6 variance -> (R, C) mask; kind 5 outliers -> per-cell stats, then the (T, R, C) mask
__global__ __launch_bounds__(256) void mask_time_kernel(const float* __restrict__ f, int64_t T, int64_t n, int kind,
                                                        float tol, double count_min, int mode_and,
                                                        uint8_t* __restrict__ mask) {
  __shared__ float tile[2][kCh][kCells];
  __shared__ float stat[4][kCells];
  __shared__ int cnts[4][kCells];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int64_t cell = (int64_t)blockIdx.x * kCells + lane, N = T * n;
  const bool valid = cell < n;
  if (kind == 2) {                                    // an integer count: order-free, split over the four waves
    int c = 0;
    if (valid)
      for (int64_t t = w; t < T; t += 4) { const float x = f[t * n + cell]; c += x == x; }
    cnts[w][lane] = c;
    __syncthreads();
    if (w == 0 && valid) mask[cell] = (double)(cnts[0][lane] + cnts[1][lane] + cnts[2][lane] + cnts[3][lane]) > count_min;
    return;
  }
  float tot = -0.0f, sq = -0.0f;
  int cnt = 0, unused = 0;
  column_pass<2, false>(f, N, T, n, cell, valid, tile, 0.0f, &tot, &cnt);
  const float avg = tot / (float)cnt;                 // 0/0 -> NaN for an all-NaN cell, as numpy
  column_pass<2, true>(f, N, T, n, cell, valid, tile, avg, &sq, &unused);
  if (w < 2) {
    stat[2 * w][lane] = avg;
    stat[2 * w + 1][lane] = sqrtf(sq / (float)cnt);
  }
  __syncthreads();
  if (!valid) return;
  const float mx = stat[0][lane], sx = stat[1][lane], my = stat[2][lane], sy = stat[3][lane];
  if (kind == 6) {
    if (w != 0) return;
    // np.maximum(mean, 1e30) (sic): NaN mean stays NaN, everything else becomes 1e30
    const float cx = mx != mx ? mx : fmaxf(mx, 1e30f), cy = my != my ? my : fmaxf(my, 1e30f);
    const bool kx = fabsf(sx / cx) < tol, ky = fabsf(sy / cy) < tol;
    mask[cell] = mode_and ? (kx && ky) : (kx || ky);
    return;
  }
  for (int64_t t = w; t < T; t += 4) {
    const bool kx = fabsf(sub_rn(f[t * n + cell], mx) / sx) < tol;
    const bool ky = fabsf(sub_rn(f[N + t * n + cell], my) / sy) < tol;
    mask[t * n + cell] = mode_and ? (kx && ky) : (kx || ky);
  }
}

// kind 7 rolling: s > tol * max_{[t - w/2, t + (w-1)/2]} fillna(s, 0); incomplete windows are NaN -> masked out
__global__ __launch_bounds__(kBlock) void mask_rolling_kernel(const float* __restrict__ f, int64_t T, int64_t n, int wdw,
                                                              float tol, uint8_t* __restrict__ mask) {
  const int64_t N = T * n;
  const int lo = wdw / 2, hi = (wdw - 1) / 2;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < N; i += (int64_t)gridDim.x * kBlock) {
    const int64_t t = i / n, cell = i - t * n;
    bool keep = false;
    if (t - lo >= 0 && t + hi < T) {
      float m = 0.0f;                                   // fillna(0) and s >= 0: the running max can start at 0
      for (int64_t k = t - lo; k <= t + hi; ++k) {
        const float s = speed(f[k * n + cell], f[N + k * n + cell]);
        m = fmaxf(m, nan0(s));
      }
      keep = speed(f[i], f[N + i]) > mul_rn(tol, m);
    }
    mask[i] = keep;
  }
}

struct Window { int x_min, x_max, y_min, y_max; };   // strides x_min..x_max and y_min..y_max-1 (helpers.py:672-679)

// nanmean over helpers.stack_window's shifted copies: x stride outer, y stride inner, value (r - sy, c - sx)
__device__ __forceinline__ float window_nanmean(const float* __restrict__ v, int R, int C, int r, int c, Window w, int* cnt) {
  float tot = 0.0f;
  int k = 0;
  bool first = true;
  for (int sx = w.x_min; sx <= w.x_max; ++sx)
    for (int sy = w.y_min; sy < w.y_max; ++sy) {
      const int rr = r - sy, cc = c - sx;
      float x = (rr >= 0 && rr < R && cc >= 0 && cc < C) ? v[(int64_t)rr * C + cc] : NAN;
      k += x == x;
      x = nan0(x);
      tot = first ? x : add_rn(tot, x);
      first = false;
    }
  *cnt = k;
  return tot / (float)k;
}

// kinds 8 window_nan, 9 window_mean: per time step, neighbourhood in (R, C)
__global__ __launch_bounds__(kBlock) void mask_window_kernel(const float* __restrict__ f, int64_t T, int R, int C, int kind,
                                                             float tol, double count_min, int mode_and, Window w,
                                                             uint8_t* __restrict__ mask) {
  const int64_t n = (int64_t)R * C, N = T * n;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < N; i += (int64_t)gridDim.x * kBlock) {
    const int64_t t = i / n, cell = i - t * n;
    const int r = (int)(cell / C), c = (int)(cell - (int64_t)r * C);
    int cnt;
    const float mx = window_nanmean(f + t * n, R, C, r, c, w, &cnt);
    if (kind == 8) { mask[i] = (double)cnt >= count_min; continue; }
    const float my = window_nanmean(f + N + t * n, R, C, r, c, w, &cnt);
    const bool kx = fabsf(sub_rn(f[i], mx)) / mx < tol, ky = fabsf(sub_rn(f[N + i], my)) / my < tol;
    mask[i] = mode_and ? (kx && ky) : (kx || ky);
  }
}

// window_replace, one iteration, all four variables: out = isnan(in) ? neighbourhood nanmean(in) : in
__global__ __launch_bounds__(kBlock) void window_replace_kernel(const float* __restrict__ in, int64_t planes, int R, int C,
                                                                Window w, float* __restrict__ out) {
  const int64_t n = (int64_t)R * C, N = planes * n;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < N; i += (int64_t)gridDim.x * kBlock) {
    const float x = in[i];
    float res = x;
    if (x != x) {
      const int64_t p = i / n, cell = i - p * n;
      const int r = (int)(cell / C), c = (int)(cell - (int64_t)r * C);
      int cnt;
      res = window_nanmean(in + p * n, R, C, r, c, w, &cnt);
    }
    out[i] = res;
  }
}

// ds[var].where(mask) on all four variables; mask (T, R, C) or, time-reduced, (R, C)
__global__ __launch_bounds__(kBlock) void mask_apply_kernel(float* __restrict__ f, int64_t N, int64_t n, int mask_has_time,
                                                            const uint8_t* __restrict__ mask) {
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < N; i += (int64_t)gridDim.x * kBlock) {
    if (mask[mask_has_time ? i : i % n]) continue;
    f[i] = NAN; f[N + i] = NAN; f[2 * N + i] = NAN; f[3 * N + i] = NAN;
  }
}

// ds.mean(dim="time") of the four variables -> (4, R, C); same block structure as mask_time_kernel
__global__ __launch_bounds__(256) void time_mean_kernel(const float* __restrict__ f, int64_t T, int64_t n,
                                                        float* __restrict__ out) {
  __shared__ float tile[1][kCh][kCells];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int64_t cell = (int64_t)blockIdx.x * kCells + lane;
  const bool valid = cell < n;
  float tot = -0.0f;
  int cnt = 0;
  column_pass<1, false>(f + (int64_t)blockIdx.y * T * n, 0, T, n, cell, valid, tile, 0.0f, &tot, &cnt);
  if (w == 0 && valid) out[(int64_t)blockIdx.y * n + cell] = tot / (float)cnt;
}

// px/frame -> m/s: (u * res / dt).astype(float32) with a python-float res (float32 product) and float64 dt
__global__ __launch_bounds__(kBlock) void scale_velocity_kernel(float* __restrict__ f, int64_t T, int64_t n, float res_x,
                                                                float res_y, const double* __restrict__ dt) {
  const int64_t N = T * n;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < N; i += (int64_t)gridDim.x * kBlock) {
    const double d = dt[i / n];
    f[i] = (float)((double)mul_rn(f[i], res_x) / d);
    f[N + i] = (float)((double)mul_rn(f[N + i], res_y) / d);
  }
}

unsigned grid_for(int64_t n) { return (unsigned)std::min<int64_t>((n + kBlock - 1) / kBlock, 256 * 32); }

}  // namespace

hipError_t launch_mask(const float* f, int64_t T, int R, int C, int kind, const double* p, uint8_t* mask, hipStream_t s) {
  const int64_t n = (int64_t)R * C, N = T * n;
  if (N == 0) return hipSuccess;
  switch (kind) {
    case 0: case 1: case 3: case 4:
      hipLaunchKernelGGL(mask_pointwise_kernel, dim3(grid_for(N)), dim3(kBlock), 0, s, f, N, kind, (float)p[0],
                         kind <= 1 ? (float)p[1] : 0.0f, mask);
      break;
    case 2:
      hipLaunchKernelGGL(mask_time_kernel, dim3((unsigned)((n + kCells - 1) / kCells)), dim3(256), 0, s, f, T, n, kind, 0.0f,
                         p[0] * (double)T, 0, mask);
      break;
    case 5: case 6:
      hipLaunchKernelGGL(mask_time_kernel, dim3((unsigned)((n + kCells - 1) / kCells)), dim3(256), 0, s, f, T, n, kind,
                         (float)p[0], 0.0, (int)p[1], mask);
      break;
    case 7:
      hipLaunchKernelGGL(mask_rolling_kernel, dim3(grid_for(N)), dim3(kBlock), 0, s, f, T, n, (int)p[0], (float)p[1], mask);
      break;
    case 8: {
      const Window w{(int)p[1], (int)p[2], (int)p[3], (int)p[4]};
      const int64_t ns = (int64_t)std::max(0, w.x_max - w.x_min + 1) * std::max(0, w.y_max - w.y_min);
      hipLaunchKernelGGL(mask_window_kernel, dim3(grid_for(N)), dim3(kBlock), 0, s, f, T, R, C, kind, 0.0f, p[0] * (double)ns, 0, w,
                         mask);
      break;
    }
    case 9: {
      const Window w{(int)p[2], (int)p[3], (int)p[4], (int)p[5]};
      hipLaunchKernelGGL(mask_window_kernel, dim3(grid_for(N)), dim3(kBlock), 0, s, f, T, R, C, kind, (float)p[0], 0.0, (int)p[1], w,
                         mask);
      break;
    }
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

hipError_t launch_mask_apply(float* f, int64_t T, int64_t n, const uint8_t* mask, int mask_has_time, hipStream_t s) {
  if (T * n == 0) return hipSuccess;
  hipLaunchKernelGGL(mask_apply_kernel, dim3(grid_for(T * n)), dim3(kBlock), 0, s, f, T * n, n, mask_has_time, mask);
  return hipGetLastError();
}

hipError_t launch_time_mean(const float* f, int64_t T, int64_t n, float* out, hipStream_t s) {
  if (T * n == 0) return hipSuccess;
  hipLaunchKernelGGL(time_mean_kernel, dim3((unsigned)((n + kCells - 1) / kCells), 4), dim3(256), 0, s, f, T, n, out);
  return hipGetLastError();
}

hipError_t launch_window_replace(const float* in, int64_t planes, int R, int C, int x_min, int x_max, int y_min, int y_max,
                                 float* out, hipStream_t s) {
  if (planes * R * C == 0) return hipSuccess;
  hipLaunchKernelGGL(window_replace_kernel, dim3(grid_for(planes * R * C)), dim3(kBlock), 0, s, in, planes, R, C,
                     Window{x_min, x_max, y_min, y_max}, out);
  return hipGetLastError();
}

hipError_t launch_scale_velocity(float* f, int64_t T, int64_t n, float res_x, float res_y, const double* d_dt, hipStream_t s) {
  if (T * n == 0) return hipSuccess;
  hipLaunchKernelGGL(scale_velocity_kernel, dim3(grid_for(T * n)), dim3(kBlock), 0, s, f, T, n, res_x, res_y, d_dt);
  return hipGetLastError();
}

}  // namespace lspiv
