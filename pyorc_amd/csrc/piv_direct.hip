// Generic-size interrogation-window kernel (any window 4..64 per side, square or not) and the
// stand-alone peak finder.
//
// The FFT kernels cover the power-of-two windows of the benchmark configurations; pyorc however
// accepts any even window (its own test uses 10, the Ngwerere camera config 25 -> rounded,
// tests/test_frames.py:139-153, examples/ngwerere/ngwerere.json).  For those sizes this kernel
// evaluates the same circular cross-correlation directly in the spatial domain,
//     plane[i'][j'] = clip( (1/N) sum_{y,x} a'[y][x] b'[(y+i'-cy) % wy][(x+j'-cx) % wx], 0, 1 )
// which is what clip(fftshift(irfft2(conj(rfft2 a') rfft2 b'))/N, 0, 1) computes (ffpiv ncc, A4),
// with a', b' the mean-offset, std-normalised, zero-clipped windows (A3).  One wavefront per
// window pair, both windows staged in LDS as float32; O(N^2) work per output instead of O(log N)
// -- a correctness path, ~10x slower than the FFT kernel at 32x32.
#include "common.h"

namespace lspiv {

constexpr int MAXW = 64;
constexpr int MAXN = MAXW * MAXW;

struct WaveArg {
  float v;
  int idx;
};

__device__ __forceinline__ void wave_argmax(float& v, int& idx) {
  half_argmax(v, idx);
  float pv = __shfl_xor(v, 32, 64);
  int pi = __shfl_xor(idx, 32, 64);
  argmax_merge(v, idx, pv, pi);
}

// stage one window into LDS as float, return the sum of (x - x0), x0 = its first sample (deterministic
// lane-strided order).  Summing offsets from x0 makes a constant window come out with an exactly-zero sum, hence
// mean == x0 and zero variance, for ANY window size -- the FFT kernels get the same from pairwise sums of 2^k terms.
template <typename T>
__device__ __forceinline__ float stage_window(const T* src, int W, int wy, int wx, float* dst, int lane, int& nonzero,
                                              float& x0) {
  const int n = wy * wx;
  x0 = to_f32(src[0]);
  float s = 0.0f;
  int nz = 0;
  for (int o = lane; o < n; o += 64) {
    const int y = o / wx, x = o - y * wx;
    const float v = to_f32(src[(int64_t)y * W + x]);
    dst[o] = v;
    s += v - x0;
    nz += (v != 0.0f) ? 1 : 0;
  }
  nonzero = half_sum_i(nz);
  nonzero += __shfl_xor(nonzero, 32, 64);
  return wave_sum(s);
}

// mean-offset / variance / clip in LDS; returns 1/std (0 if std == 0)
__device__ __forceinline__ float normalize_window(float* w, int n, float sum_off, float x0, int lane, bool& finite) {
  const float mean = x0 + sum_off / (float)n;
  float ssq = 0.0f;
  for (int o = lane; o < n; o += 64) {
    const float d = w[o] - mean;
    ssq += d * d;
    w[o] = fmaxf(d, 0.0f);
  }
  ssq = wave_sum(ssq);
  finite = finite && (fabsf(mean) <= 3.0e38f) && (ssq <= 3.0e38f);
  const float var = ssq / (float)n;
  return var > 0.0f ? 1.0f / sqrtf(var) : 0.0f;
}

// correlation plane of the staged pair into plane[] (shifted layout), clipped to [0, 1]
__device__ __forceinline__ void correlate_direct(const float* a, const float* b, float* plane, int wy, int wx,
                                                 float scale, int lane) {
  const int n = wy * wx;
  const int cy = wy / 2, cx = wx / 2;
  for (int o = lane; o < n; o += 64) {
    const int ip = o / wx, jp = o - ip * wx;
    int dy = ip - cy; dy += dy < 0 ? wy : 0;
    int dx = jp - cx; dx += dx < 0 ? wx : 0;
    float tot = 0.0f;
    for (int y = 0; y < wy; ++y) {
      int yb = y + dy; yb -= yb >= wy ? wy : 0;
      const float* ar = a + y * wx;
      const float* br = b + yb * wx;
      float rs = 0.0f;
      int x = 0;
      for (; x < wx - dx; ++x) rs = fmaf(ar[x], br[x + dx], rs);
      for (; x < wx; ++x) rs = fmaf(ar[x], br[x + dx - wx], rs);
      tot += rs;
    }
    plane[o] = fminf(fmaxf(tot * scale, 0.0f), 1.0f);
  }
}

__device__ __forceinline__ void plane_reduce(const float* plane, int n, int lane, float& vmax, int& imax, float& sum) {
  float best = -1.0f;
  int bi = 0x7fffffff;
  float s = 0.0f;
  for (int o = lane; o < n; o += 64) {
    const float v = plane[o];
    s += v;
    if (v > best) { best = v; bi = o; }
  }
  vmax = best; imax = bi;
  wave_argmax(vmax, imax);
  sum = wave_sum(s);
}

// sub-pixel peak of a plane addressed through `ld` (LDS or global), flat argmax index imax
template <typename F>
__device__ __forceinline__ void subpixel_generic(F ld, int wy, int wx, int imax, float& u, float& v) {
  const int i = imax / wx, j = imax - i * wx;
  if (i <= 0 || i >= wy - 1 || j <= 0 || j >= wx - 1) {
    u = v = __builtin_nanf("");
    return;
  }
  // same arithmetic as the fused FFT kernels (hardware log2 -- the fit is a ratio of log differences -- and a
  // 1-ulp reciprocal), so peaks found from a plane volume equal the fused results bit for bit
  const float l0 = __builtin_amdgcn_logf(ld(i * wx + j) + kEpsPeak);
  v = (float)i +
      gauss_offset_fast(__builtin_amdgcn_logf(ld((i - 1) * wx + j) + kEpsPeak), l0,
                        __builtin_amdgcn_logf(ld((i + 1) * wx + j) + kEpsPeak)) -
      (float)(wy / 2);
  u = (float)j +
      gauss_offset_fast(__builtin_amdgcn_logf(ld(i * wx + j - 1) + kEpsPeak), l0,
                        __builtin_amdgcn_logf(ld(i * wx + j + 1) + kEpsPeak)) -
      (float)(wx / 2);
}

// one window pair -> plane in LDS + (corr_max, sum).  Returns false when the plane is NaN.
template <typename T>
__device__ __forceinline__ bool direct_pair(const PivParams& p, uint32_t pair, uint32_t win, float* a, float* b,
                                            float* plane, int lane) {
  const T* frames = static_cast<const T*>(p.frames);
  const uint32_t wrow = win / (uint32_t)p.n_cols, wcol = win - wrow * (uint32_t)p.n_cols;
  const int64_t off = ((int64_t)pair * p.H + (int64_t)wrow * p.sy) * p.W + (int64_t)wcol * p.sx;
  const int n = p.wy * p.wx;
  int nza, nzb;
  float a0, b0;
  const float sa = stage_window(frames + off, p.W, p.wy, p.wx, a, lane, nza, a0);
  const float sb = stage_window(frames + off + p.frame_elems, p.W, p.wy, p.wx, b, lane, nzb, b0);
  __builtin_amdgcn_wave_barrier();
  bool finite = true;
  const float inv_a = normalize_window(a, n, sa, a0, lane, finite);
  const float inv_b = normalize_window(b, n, sb, b0, lane, finite);
  __builtin_amdgcn_wave_barrier();
  bool ok = finite;
  if (p.signal_threshold >= 0.0f) {
    const float fa = (float)nza / (float)n, fb = (float)nzb / (float)n;
    ok = ok && (fa >= p.signal_threshold) && (fb >= p.signal_threshold);
  }
  correlate_direct(a, b, plane, p.wy, p.wx, inv_a * inv_b / (float)n, lane);
  __builtin_amdgcn_wave_barrier();
  return ok;
}

template <typename T>
__global__ __launch_bounds__(64) void piv_direct_kernel(PivParams p) {
  __shared__ float a[MAXN], b[MAXN], plane[MAXN];
  const int lane = threadIdx.x;
  const uint32_t g = blockIdx.x;
  const uint32_t pair = g / p.n_win, win = g - pair * p.n_win;
  const int n = p.wy * p.wx;
  const bool ok = direct_pair<T>(p, pair, win, a, b, plane, lane);
  float vmax, sum, u, v;
  int imax;
  plane_reduce(plane, n, lane, vmax, imax, sum);
  subpixel_generic([&](int o) { return plane[o]; }, p.wy, p.wx, imax, u, v);
  float cm = vmax, sn = vmax / (sum / (float)n);
  if (!ok) u = v = cm = sn = __builtin_nanf("");
  if (lane == 0) {
    p.u[g] = u; p.v[g] = v; p.cmax[g] = cm; p.s2n[g] = sn;
  }
  if (p.planes) {
    float* dst = p.planes + (size_t)g * n;
    for (int o = lane; o < n; o += 64) dst[o] = ok ? plane[o] : __builtin_nanf("");
  }
}

// ensemble: one wave owns one window and walks the chunk's pairs in order (see piv_fft32.hip)
template <typename T>
__global__ __launch_bounds__(64) void piv_direct_ensemble_kernel(PivParams p) {
  __shared__ float a[MAXN], b[MAXN], plane[MAXN], acc[MAXN];
  const int lane = threadIdx.x;
  const uint32_t win = blockIdx.x;
  const int n = p.wy * p.wx;
  for (int o = lane; o < n; o += 64) acc[o] = 0.0f;
  float cnt = 0.0f;
  for (uint32_t pair = 0; pair < p.n_pairs; ++pair) {
    const bool ok = direct_pair<T>(p, pair, win, a, b, plane, lane);
    float vmax, sum;
    int imax;
    plane_reduce(plane, n, lane, vmax, imax, sum);
    float cm = vmax, sn = vmax / (sum / (float)n);
    const bool keep = ok && (cm >= p.corr_min) && (sn >= p.s2n_min);
    cm = keep ? cm : 0.0f;
    sn = keep ? sn : 0.0f;
    cnt += (cm > 1e-6f) ? 1.0f : 0.0f;
    if (lane == 0) {
      p.cmax[(size_t)pair * p.n_win + win] = cm;
      p.s2n[(size_t)pair * p.n_win + win] = sn;
    }
    if (keep)
      for (int o = lane; o < n; o += 64) acc[o] += plane[o];
    __builtin_amdgcn_wave_barrier();
  }
  float* dst = p.corr_sum + (size_t)win * n;
  for (int o = lane; o < n; o += 64) dst[o] += acc[o];
  if (lane == 0) p.corr_count[win] += cnt;
}

template <typename T>
static hipError_t launch_t(const PivParams& p, bool ensemble, hipStream_t s) {
  if (ensemble)
    hipLaunchKernelGGL(piv_direct_ensemble_kernel<T>, dim3(p.n_win), dim3(64), 0, s, p);
  else
    hipLaunchKernelGGL(piv_direct_kernel<T>, dim3(p.n_tiles), dim3(64), 0, s, p);
  return hipGetLastError();
}

hipError_t launch_piv_direct(const PivParams& p, int dtype, bool ensemble, hipStream_t s) {
  switch (dtype) {
    case 0: return launch_t<uint8_t>(p, ensemble, s);
    case 1: return launch_t<float>(p, ensemble, s);
    case 2: return launch_t<double>(p, ensemble, s);
    default: return hipErrorInvalidValue;
  }
}

// ---- ffpiv.u_v_displacement on a plane volume in HBM (pyorc/velocimetry/ffpiv.py:324,471) ------
// np.argmax semantics: first maximum in row-major order, NaN counts as maximum.
__global__ __launch_bounds__(64) void peaks_kernel(const float* planes, uint32_t n_planes, int wy, int wx,
                                                   float* u, float* v) {
  const uint32_t g = blockIdx.x;
  const int lane = threadIdx.x;
  const int n = wy * wx;
  const float* pl = planes + (size_t)g * n;
  float best = -__builtin_inff();
  int bi = 0x7fffffff;
  for (int o = lane; o < n; o += 64) {
    float x = pl[o];
    x = (x != x) ? __builtin_inff() : x;  // NaN ranks as the maximum, first one wins
    if (x > best || bi == 0x7fffffff) { best = x; bi = o; }
  }
  wave_argmax(best, bi);
  float uu, vv;
  subpixel_generic([&](int o) { return pl[o]; }, wy, wx, bi, uu, vv);
  if (lane == 0) { u[g] = uu; v[g] = vv; }
}

hipError_t launch_peaks_from_planes(const float* planes, uint32_t n_planes, int wy, int wx, float* u, float* v,
                                    hipStream_t s) {
  if (n_planes == 0) return hipSuccess;
  hipLaunchKernelGGL(peaks_kernel, dim3(n_planes), dim3(64), 0, s, planes, n_planes, wy, wx, u, v);
  return hipGetLastError();
}

}  // namespace lspiv
