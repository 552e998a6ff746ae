// Generic-size interrogation-window kernel (any window 4..64 per side, square or not) and the
// stand-alone peak finder.
//
// The FFT kernels cover the power-of-two windows of the benchmark configurations; pyorc however
// accepts any even window (its own test uses 10, the Ngwerere camera config 25 -> rounded,
// tests/test_frames.py:139-153, examples/ngwerere/ngwerere.json).  For those sizes this kernel
// evaluates the same circular cross-correlation directly in the spatial domain,
//     plane[i'][j'] = clip( (1/N) sum_{y,x} a'[y][x] b'[(y+i'-cy) % wy][(x+j'-cx) % wx], 0, 1 )
// which is what clip(fftshift(irfft2(conj(rfft2 a') rfft2 b'))/N, 0, 1) computes (ffpiv ncc, A4),
// with a', b' the mean-offset, std-normalised, zero-clipped windows (A3).  One 4-wave block per window pair, both
// windows staged in LDS as float32 (b with periodically doubled rows), every thread accumulating 8 neighbouring lags
// in registers: O(N^2) multiply-adds per output instead of O(log N).  Square windows up to 31 do NOT come here any
// more (they run embedded in the FFT kernels, piv_fft_impl.h); this is the path for non-square windows and 33..63.
#include <algorithm>

#include "common.h"
#include "fft_regs.h"

namespace lspiv {


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

__device__ __forceinline__ float wave_max_f(float x) {
  x = half_max(x);
  return fmaxf(x, __shfl_xor(x, 32, 64));
}

// ---- block-level pieces: 256 threads (4 waves) work on one window pair -----------------------------------------
constexpr int DBLOCK_MAX = 512;   // block = as many waves as the strips of one window pair need (2..8), one round of strips
constexpr int DXB = 8;            // lags per thread: one a-sample and one new b-sample feed 8 FMAs

struct DirectGeo {
  int n, bpitch, strips_per_row;  // samples per window; padded length of a doubled b row; ceil(wx / DXB)
  // odd pitch: lanes of a wave read different rows of b (different dy), an even pitch folds them onto few LDS banks
  __device__ DirectGeo(int wy, int wx) : n(wy * wx), bpitch((wx + ((wx + DXB - 1) / DXB) * DXB) | 1), strips_per_row((wx + DXB - 1) / DXB) {}
};

// sum over the block in a fixed order (wave reductions, then the four partials left to right): deterministic
__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  float tot = red[0];
  for (int k = 1; k < (int)(blockDim.x >> 6); ++k) tot += red[k];   // left to right: a fixed order
  return tot;
}
__device__ __forceinline__ int block_sum_i(int v, int* red) {
  v = half_sum_i(v);
  v += __shfl_xor(v, 32, 64);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  int tot = red[0];
  for (int k = 1; k < (int)(blockDim.x >> 6); ++k) tot += red[k];
  return tot;
}

// two float sums at once (one pair of barriers instead of two): same fixed order as block_sum
__device__ __forceinline__ void block_sum2(float& a, float& b, float* red) {
  a = wave_sum(a);
  b = wave_sum(b);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = a; red[8 + (threadIdx.x >> 6)] = b; }
  __syncthreads();
  a = red[0]; b = red[8];
  for (int k = 1; k < (int)(blockDim.x >> 6); ++k) { a += red[k]; b += red[8 + k]; }
}

// (row, column) of the flat window index o = threadIdx.x + i blockDim.x, stepped without a division per element
struct RowCol {
  int y, x;
  int dy, dx, wx;
  __device__ __forceinline__ RowCol(int wx_) : wx(wx_) {
    const int t = (int)threadIdx.x, step = (int)blockDim.x;
    y = t / wx; x = t - y * wx;
    dy = step / wx; dx = step - dy * wx;
  }
  __device__ __forceinline__ void next() {
    y += dy; x += dx;
    if (x >= wx) { x -= wx; ++y; }
  }
};

// Stage one window into LDS (row pitch `pitch`; with `periodic` every row is continued periodically up to the pitch,
// which turns the circular column shift into a plain offset), mean-offset / variance / clip in place.  The mean is
// x0 + mean(x - x0), x0 the first sample: a constant window has exactly zero variance for any size.
// Returns 1/std (0 if std == 0).
// `clip_sum` (optional): sum of the clipped samples, for callers that remove the mean of the clipped window afterwards.
template <typename T>
__device__ __forceinline__ float stage_window(const T* src, int W, int wy, int wx, float* dst, int pitch, bool periodic,
                                              bool nz_pos, float* red, int& nonzero, bool& finite, float* clip_sum = nullptr,
                                              bool clip = true) {
  const int n = wy * wx;
  const float x0 = to_f32(src[0]);
  const RowCol rc0(wx);
  float s = 0.0f, nz = 0.0f;                                // the count as a float: exact below 2^24
  RowCol rc = rc0;
  // eight loads in flight per thread before the first is consumed: a block is alone on its CU for the larger windows, so
  // nobody else hides the latency of a load-use chain per sample (the samples past the window re-read sample 0)
  constexpr int UB = 8;
  for (int o = threadIdx.x; o < n; o += UB * (int)blockDim.x) {
    T raw[UB];
    int ad[UB];
#pragma unroll
    for (int u = 0; u < UB; ++u) {
      const bool in = o + u * (int)blockDim.x < n;
      raw[u] = src[in ? (int64_t)rc.y * W + rc.x : 0];
      ad[u] = in ? rc.y * pitch + rc.x : -1;
      rc.next();
    }
#pragma unroll
    for (int u = 0; u < UB; ++u) {
      if (ad[u] >= 0) {
        const float v = to_f32(raw[u]);
        dst[ad[u]] = v;
        s += v - x0;
        nz += (nz_pos ? v > 0.0f : v != 0.0f) ? 1.0f : 0.0f;
      }
    }
  }
  block_sum2(s, nz, red);
  nonzero = (int)nz;
  const float mean = x0 + s / (float)n;
  float ssq = 0.0f, cs = 0.0f;
  rc = rc0;
  for (int o = threadIdx.x; o < n; o += blockDim.x, rc.next()) {
    const float d = dst[rc.y * pitch + rc.x] - mean;
    const float c = clip ? fmaxf(d, 0.0f) : d;   // clip: A3's removal of the negative lobes ("norm_clip" option)
    ssq += d * d;
    cs += c;
    dst[rc.y * pitch + rc.x] = c;
  }
  block_sum2(ssq, cs, red);
  if (clip_sum) *clip_sum = cs;
  finite = finite && (fabsf(mean) <= 3.0e38f) && (ssq <= 3.0e38f);
  if (periodic) {
    const int ext = pitch - wx;
    for (int o = threadIdx.x; o < wy * ext; o += blockDim.x) {
      const int y = o / ext, x = o - y * ext;
      dst[y * pitch + wx + x] = dst[y * pitch + x % wx];
    }
  }
  const float var = ssq / (float)n;
  return var > 0.0f ? 1.0f / sqrtf(var) : 0.0f;
}

// plane[i'][j'] (fft-shifted) = clip(scale * sum_{y,x} a[y][x] b[(y + dy) % wy][(x + dx) % wx], 0, 1), dy = i' - cy, dx = j' - cx.
// A thread owns DXB consecutive un-shifted lags dx of one dy: per (y, x) one broadcast read of a, one new b sample,
// DXB FMAs -- the doubled b rows make the window of b samples slide without a modulo.
__device__ __forceinline__ void correlate_direct(const float* a, const float* b2, float* plane, int wy, int wx,
                                                 const DirectGeo& g, float scale) {
  const int cy = wy / 2, cx = wx / 2;
  const int strips = wy * g.strips_per_row;
  for (int sidx = threadIdx.x; sidx < strips; sidx += blockDim.x) {
    const int dy = sidx / g.strips_per_row, dx0 = (sidx - dy * g.strips_per_row) * DXB;
    float acc[DXB];
#pragma unroll
    for (int e = 0; e < DXB; ++e) acc[e] = 0.0f;
    int yb = dy;
    for (int y = 0; y < wy; ++y) {
      const float* ar = a + y * wx;
      const float* br = b2 + yb * g.bpitch + dx0;
      float w[DXB];
#pragma unroll
      for (int e = 0; e < DXB - 1; ++e) w[e] = br[e];
      for (int x = 0; x < wx; ++x) {
        w[DXB - 1] = br[x + DXB - 1];
        const float av = ar[x];
#pragma unroll
        for (int e = 0; e < DXB; ++e) acc[e] = fmaf(av, w[e], acc[e]);
#pragma unroll
        for (int e = 0; e < DXB - 1; ++e) w[e] = w[e + 1];
      }
      yb = (yb + 1 == wy) ? 0 : yb + 1;
    }
    const int ip = dy + cy >= wy ? dy + cy - wy : dy + cy;
#pragma unroll
    for (int e = 0; e < DXB; ++e) {
      const int dx = dx0 + e;
      if (dx < wx) {
        const int jp = dx + cx >= wx ? dx + cx - wx : dx + cx;
        plane[ip * wx + jp] = fminf(fmaxf(acc[e] * scale, 0.0f), 1.0f);
      }
    }
  }
}

// max / first arg-max (row-major) / sum of the LDS plane over the block
__device__ __forceinline__ void plane_reduce(const float* plane, int n, float* red, float& vmax, int& imax, float& sum) {
  float best = -1.0f;
  int bi = 0x7fffffff;
  float s = 0.0f;
  for (int o = threadIdx.x; o < n; o += blockDim.x) {
    const float v = plane[o];
    s += v;
    if (v > best) { best = v; bi = o; }
  }
  wave_argmax(best, bi);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) { red[8 + (threadIdx.x >> 6)] = best; reinterpret_cast<int*>(red)[16 + (threadIdx.x >> 6)] = bi; }
  __syncthreads();
  vmax = red[8];
  imax = reinterpret_cast<int*>(red)[16];
  for (int k = 1; k < (int)(blockDim.x >> 6); ++k) argmax_merge(vmax, imax, red[8 + k], reinterpret_cast<int*>(red)[16 + k]);
  sum = block_sum(s, red);
}

// sub-pixel peak of a plane addressed through `ld` (LDS or global), flat argmax index imax
template <typename F>
__device__ __forceinline__ void subpixel_generic(F ld, int wy, int wx, int imax, int border_mode, float& u, float& v) {
  const int i = imax / wx, j = imax - i * wx;
  if (i <= 0 || i >= wy - 1 || j <= 0 || j >= wx - 1) {
    border_result(border_mode, j - wx / 2, i - wy / 2, u, v);
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

// After plane_reduce / subpixel_generic: does this window go to the float64 rescue pass (common.h, peak_cond)?  One more
// pass over the plane for the runner-up, the fit's own terms again from the five samples; thread 0 appends the record.
// The block-per-window kernels are not the tuned path: they assume twice the plane noise of the fused FFT kernels.
__device__ __forceinline__ void block_rescue_note(const PivParams& p, const float* plane, int n, float* red, float vmax, int imax,
                                                  float u, float v, uint32_t t, bool ok) {
  if (!p.rescue_hdr) return;   // uniform
  // candidates of the arg-max besides imax: samples within tau of the maximum -- how many, and the first of them (row-major).
  // Exactly one other candidate: the rescue pass settles the two by their float64 sums (a "fit" record with pos2), no whole plane.
  const float thr = vmax * (1.0f - p.rescue_tau);
  int cnt = 0, other = 0x7fffffff;
  for (int o = threadIdx.x; o < n; o += blockDim.x) {
    const bool cand = o != imax && plane[o] >= thr;
    cnt += cand ? 1 : 0;
    other = cand ? min(other, o) : other;
  }
  cnt = half_sum_i(cnt);
  cnt += __shfl_xor(cnt, 32, 64);
  other = half_min_i(other);
  other = min(other, __shfl_xor(other, 32, 64));
  int* redi = reinterpret_cast<int*>(red);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) { redi[8 + (threadIdx.x >> 6)] = cnt; redi[16 + (threadIdx.x >> 6)] = other; }
  __syncthreads();
  cnt = redi[8]; other = redi[16];
  for (int k = 1; k < (int)(blockDim.x >> 6); ++k) { cnt += redi[8 + k]; other = min(other, redi[16 + k]); }
  __syncthreads();
  if (threadIdx.x != 0 || !ok) return;
  const int wy = p.wy, wx = p.wx;
  const int i = imax / wx, j = imax - i * wx;
  const bool border = i <= 0 || i >= wy - 1 || j <= 0 || j >= wx - 1;
  float cl = 1.0f, cr = 1.0f, cd = 1.0f, cu = 1.0f, den_v = 1.0f, den_u = 1.0f;
  if (!border) {
    const float l0 = __builtin_amdgcn_logf(plane[imax] + kEpsPeak);
    cl = plane[imax - wx] + kEpsPeak; cr = plane[imax + wx] + kEpsPeak;
    cd = plane[imax - 1] + kEpsPeak; cu = plane[imax + 1] + kEpsPeak;
    gauss_offset_fast(__builtin_amdgcn_logf(cl), l0, __builtin_amdgcn_logf(cr), den_v);
    gauss_offset_fast(__builtin_amdgcn_logf(cd), l0, __builtin_amdgcn_logf(cu), den_u);
  }
  const PeakCond pc = peak_cond(vmax, cnt > 0, border, cl, cr, den_v, v, cd, cu, den_u, u, 2.0f * p.rescue_k);
  const uint32_t pos2 = cnt == 1 ? (((uint32_t)(other / wx) << 16) | (uint32_t)(other - (other / wx) * wx)) : 0xffffffffu;
  if (pc.amb || pc.fit) rescue_note(p.rescue_hdr, p.rescue_fit, p.rescue_cap_fit, p.rescue_amb, p.rescue_cap_amb, t, pc, i, j, pos2);
}

// one window pair -> plane in LDS.  Returns false when the plane is NaN (non-finite input / signal threshold).
template <typename T>
__device__ __forceinline__ bool direct_pair(const PivParams& p, uint32_t pair, uint32_t win, float* a, float* b2,
                                            float* plane, float* red, const DirectGeo& g) {
  const T* frames = static_cast<const T*>(p.frames);
  const uint32_t wrow = win / (uint32_t)p.n_cols, wcol = win - wrow * (uint32_t)p.n_cols;
  const int64_t off = ((int64_t)pair * p.H + (int64_t)wrow * p.sy) * p.W + (int64_t)wcol * p.sx;
  int nza, nzb;
  bool finite = true;
  const float inv_a = stage_window(frames + off, p.W, p.wy, p.wx, a, p.wx, false, p.nz_positive != 0, red, nza, finite, nullptr, p.norm_clip != 0);
  const float inv_b = stage_window(frames + off + p.frame_elems, p.W, p.wy, p.wx, b2, g.bpitch, true, p.nz_positive != 0, red, nzb, finite, nullptr, p.norm_clip != 0);
  __syncthreads();
  bool ok = finite;
  if (p.signal_threshold >= 0.0f) {
    const float fa = (float)nza / (float)g.n, fb = (float)nzb / (float)g.n;
    ok = ok && (fa >= p.signal_threshold) && (fb >= p.signal_threshold);
    if (p.win_keep) ok = ok && p.win_keep[win];   // "stack" mode: one score per window position (A7)
  }
  correlate_direct(a, b2, plane, p.wy, p.wx, g, inv_a * inv_b * p.std_gain2 / (float)g.n);
  __syncthreads();
  return ok;
}

// LDS: a (n) | doubled b (wy * bpitch) | plane (n) | 24 dwords of reduction scratch (8 sums, 8 maxima, 8 indices)
__device__ __forceinline__ void carve(float* smem, const DirectGeo& g, int wy, float*& a, float*& b2, float*& plane, float*& red) {
  a = smem; b2 = a + g.n; plane = b2 + wy * g.bpitch; red = plane + g.n;
}
static size_t direct_lds_bytes(int wy, int wx) {
  const int bpitch = (wx + ((wx + DXB - 1) / DXB) * DXB) | 1;
  return ((size_t)2 * wy * wx + (size_t)wy * bpitch + 24) * sizeof(float);
}

template <typename T>
__global__ __launch_bounds__(DBLOCK_MAX) void piv_direct_kernel(PivParams p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const DirectGeo g(p.wy, p.wx);
  float *a, *b2, *plane, *red;
  carve(smem, g, p.wy, a, b2, plane, red);
  const uint32_t t = blockIdx.x;
  const uint32_t pair = t / p.n_win, win = t - pair * p.n_win;
  const bool ok = direct_pair<T>(p, pair, win, a, b2, plane, red, g);
  float vmax, sum, u, v;
  int imax;
  plane_reduce(plane, g.n, red, vmax, imax, sum);
  subpixel_generic([&](int o) { return plane[o]; }, p.wy, p.wx, imax, p.border_mode, u, v);
  block_rescue_note(p, plane, g.n, red, vmax, imax, u, v, t, ok);
  float cm = vmax, sn = vmax / (sum / (float)g.n);
  if (!ok) u = v = cm = sn = __builtin_nanf("");
  if (threadIdx.x == 0) {
    p.u[t] = u; p.v[t] = v; p.cmax[t] = cm; p.s2n[t] = sn;
  }
  if (p.planes) {
    float* dst = p.planes + (size_t)t * g.n;
    for (int o = threadIdx.x; o < g.n; o += blockDim.x) dst[o] = ok ? plane[o] : __builtin_nanf("");
  }
}

// ensemble: one block owns one window and walks the chunk's pairs in order (see piv_fft_impl.h); the running sum
// of the chunk lives in HBM (corr_sum), one coalesced read-modify-write per kept pair
template <typename T>
__global__ __launch_bounds__(DBLOCK_MAX) void piv_direct_ensemble_kernel(PivParams p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const DirectGeo g(p.wy, p.wx);
  float *a, *b2, *plane, *red;
  carve(smem, g, p.wy, a, b2, plane, red);
  const uint32_t win = blockIdx.x;
  float* dst = p.corr_sum + (size_t)win * g.n;
  float cnt = 0.0f;
  for (uint32_t pair = 0; pair < p.n_pairs; ++pair) {
    const bool ok = direct_pair<T>(p, pair, win, a, b2, plane, red, g);
    float vmax, sum;
    int imax;
    plane_reduce(plane, g.n, red, vmax, imax, sum);
    float cm = vmax, sn = vmax / (sum / (float)g.n);
    const bool keep = ok && (cm >= p.corr_min) && (sn >= p.s2n_min);
    cm = keep ? cm : 0.0f;
    sn = keep ? sn : 0.0f;
    cnt += (cm > 1e-6f) ? 1.0f : 0.0f;
    if (threadIdx.x == 0) {
      p.cmax[(size_t)pair * p.n_win + win] = cm;
      p.s2n[(size_t)pair * p.n_win + win] = sn;
    }
    if (keep)
      for (int o = threadIdx.x; o < g.n; o += blockDim.x) dst[o] += plane[o];
    __syncthreads();
  }
  if (threadIdx.x == 0) p.corr_count[win] += cnt;
}

template <typename T>
static hipError_t launch_t(const PivParams& p, bool ensemble, hipStream_t s) {
  const size_t lds = direct_lds_bytes(p.wy, p.wx);
  const int strips = p.wy * ((p.wx + DXB - 1) / DXB);
  const int threads = std::min(DBLOCK_MAX, std::max(128, ((strips + 63) / 64) * 64));
  if (ensemble)
    hipLaunchKernelGGL(piv_direct_ensemble_kernel<T>, dim3(p.n_win), dim3(threads), lds, s, p);
  else
    hipLaunchKernelGGL(piv_direct_kernel<T>, dim3(p.n_tiles), dim3(threads), lds, s, p);
  return hipGetLastError();
}

// ---- windows above 64 px per side (and any shape that fits LDS): packed 2-D DFT of the window pair, in LDS -----------
// ffpiv.cross_corr has no upper bound on the window (pyorc/api/frames.py:159-168 takes it free-form from the camera
// configuration; 4K footage is routinely processed with 96 / 128 px windows).  Such a window no longer fits the
// lane-per-row register layout of the fused FFT kernels (a 128-sample complex row is 256 VGPRs), so it lives in LDS:
//     z = a'' + i b''  (both windows normalised to unit variance, scaled 1 / n)         n = wy wx, 2 n floats of LDS
//     Z = DFT2(z), in place;   R[k] = conj(A[k]) B[k]  from  Z[k], Z[-k]  (in place, a thread owns the pair k, -k)
//     c = Re IDFT2(R)  ->  clip, fft-shift, max / mean / first arg-max / 3-point fit  (the direct kernel's epilogue)
// The 1-D transforms are plain DFT sums, out[k] = sum_n x[n] w^{n k}, register-blocked 4 lines x 4 frequencies per thread
// (16 LDS reads feed 64 FMAs), twiddles from a per-block table (computed in double, rounded once): O(n^1.5) instead of
// O(n log n), but any length works -- even, odd, prime, non-square -- with one kernel and no size-specific code, and
// 128 x 128 complex samples (128 KB) fit the 160 KB of a CU.  Each window pair is computed on its own: results do not
// depend on the time chunking.  A block is 512 threads; every pass computes all outputs into registers before anything
// is written back, so the transforms run in place.
constexpr int FBLOCK = 512;
constexpr int FTILES = 2;   // 4 x 4 output tiles per thread and pass: 512 x 2 x 16 = 16 384 outputs = 128 x 128

__host__ __device__ inline int fourstep_m(int wy, int wx);
struct DftGeo {
  int n, pitch;   // samples per window; LDS row pitch (odd: the four rows of a tile fall on different banks; four-step
                  // passes read 16 bytes at a time: wx + 4)
  __host__ __device__ DftGeo(int wy, int wx) : n(wy * wx), pitch(fourstep_m(wy, wx) ? wx + 4 : (wx | 1)) {}
  // re plane | im plane | (cos, sin) tables for x and y | reduction scratch
  __host__ __device__ size_t lds_floats(int wy, int wx) const { return (size_t)2 * wy * pitch + 2 * (size_t)(wx + wy) + 24; }
};

// One register-blocked DFT pass over a set of strided lines.  A line is addressed in two levels -- line (o, i), o < n_outer,
// i < n_inner, starts at o * lstride + i * in_istride and its element e sits e * in_estride further -- and frequency k of
// its transform is WRITTEN to o * lstride + i * out_istride + k * out_estride (anywhere: every output of the pass is in
// registers before the first is stored).  The twiddles come from a table of `tw_len` entries, tw[2 m] = cos(2 pi m / tw_len),
// tw[2 m + 1] = sin(...), read with stride `tmul` (a length-len transform uses w_len = w_tw_len^tmul).  `post_tw`: output k
// of line (o, i) is multiplied by w_tw_len^(k i) on its way out.  forward: exp(-i), INV: exp(+i); unnormalised.
struct PassGeo {
  int len, n_outer, n_inner;
  int in_istride, in_estride, out_istride, out_estride, lstride;
  int tmul, tw_len;
  bool post_tw;
};

template <bool INV>
__device__ __forceinline__ void dft_pass_g(float* re, float* im, const PassGeo& g, const float* tw) {
  const int len = g.len, n_lines = g.n_outer * g.n_inner;
  const int ntk = (len + 3) >> 2, ntl_all = (n_lines + 3) >> 2;
  // The block holds FBLOCK FTILES output tiles at a time.  Padding a short length to a multiple of 4 can push a pass over
  // that (126 = 9 x 14: 3 x 441 tiles): it then runs in rounds of whole groups of 4 outer lines -- the inner lines of an
  // outer line read what the others write, so they stay in one round; different outer lines never touch each other.
  int ntl_round = ntl_all;
  if (ntk * ntl_all > FBLOCK * FTILES) ntl_round = max(g.n_inner, (FBLOCK * FTILES / ntk) / g.n_inner * g.n_inner);
  for (int tl_base = 0; tl_base < ntl_all; tl_base += ntl_round) {
  const int ntl = min(ntl_round, ntl_all - tl_base), ntiles = ntk * ntl;
  float ar[FTILES][4][4], ai[FTILES][4][4];
  int l0[FTILES], k0[FTILES];
#pragma unroll
  for (int t = 0; t < FTILES; ++t) {
    const int tile = (int)threadIdx.x + t * FBLOCK;
    const bool on = tile < ntiles;
    const int tl = on ? tile / ntk : 0, tk = on ? tile - tl * ntk : 0;
    l0[t] = 4 * (tl_base + tl); k0[t] = 4 * tk;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) ar[t][j][q] = ai[t][j][q] = 0.0f;
    if (!on) continue;
    int lo[4], step[4], idx[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {   // edge tiles recompute the last line / frequency
      const int l = min(l0[t] + j, n_lines - 1), o = l / g.n_inner, i = l - o * g.n_inner;
      lo[j] = o * g.lstride + i * g.in_istride;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) { step[q] = (min(k0[t] + q, len - 1) * g.tmul) % g.tw_len; idx[q] = 0; }
    for (int n = 0; n < len; ++n) {
      float zr[4], zi[4], c[4], sn[4];
      const int eo = n * g.in_estride;
#pragma unroll
      for (int j = 0; j < 4; ++j) { zr[j] = re[lo[j] + eo]; zi[j] = im[lo[j] + eo]; }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        c[q] = tw[2 * idx[q]];
        sn[q] = INV ? -tw[2 * idx[q] + 1] : tw[2 * idx[q] + 1];
        idx[q] += step[q];
        idx[q] = idx[q] >= g.tw_len ? idx[q] - g.tw_len : idx[q];   // (n k tmul) mod tw_len, kept incrementally
      }
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {   // (zr + i zi)(c - i s)
          ar[t][j][q] = fmaf(zr[j], c[q], fmaf(zi[j], sn[q], ar[t][j][q]));
          ai[t][j][q] = fmaf(zi[j], c[q], fmaf(-zr[j], sn[q], ai[t][j][q]));
        }
    }
  }
  __syncthreads();   // every output is in registers: the lines may be overwritten
#pragma unroll
  for (int t = 0; t < FTILES; ++t) {
    const int tile = (int)threadIdx.x + t * FBLOCK;
    if (tile >= ntiles) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (l0[t] + j >= n_lines) continue;
      const int l = l0[t] + j, o = l / g.n_inner, i = l - o * g.n_inner;
      const int ob = o * g.lstride + i * g.out_istride;
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (k0[t] + q < len) {
          const int k = k0[t] + q;
          float vr = ar[t][j][q], vi = ai[t][j][q];
          if (g.post_tw) {                                   // k i < tw_len: no wrap
            const float c = tw[2 * k * i], sn = INV ? -tw[2 * k * i + 1] : tw[2 * k * i + 1];
            const float wr = fmaf(vr, c, vi * sn), wi = fmaf(vi, c, -vr * sn);
            vr = wr; vi = wi;
          }
          re[ob + k * g.out_estride] = vr;
          im[ob + k * g.out_estride] = vi;
        }
    }
  }
  __syncthreads();
  }
}

// in-place 1-D DFTs of `n_lines` lines of `len` elements: element e of line l at [l * lstride + e * estride]; tw: the table
// of length len.  A composite length N1 N2 takes two passes of lengths N1 and N2 (decimation in time: N2 transforms of
// length N1 over the samples n2 + N2 n1, a twiddle w_len^(k1 n2) on the way out, N1 transforms of length N2 whose output
// k2 lands at k1 + N1 k2): (N1 + N2) instead of N1 N2 complex multiply-adds per output -- 98 = 7 x 14: 21 instead of 98.
__host__ __device__ inline int dft_split(int len) {   // N1: the divisor of len closest to sqrt(len) from below; 1 = prime
  int best = 1;
  for (int d = 2; d * d <= len; ++d)
    if (len % d == 0) best = d;
  return best;
}
template <bool INV>
__device__ __forceinline__ void dft_pass(float* re, float* im, int len, int n_lines, int estride, int lstride, const float* tw) {
  const int n1 = dft_split(len), n2 = len / n1;
  if (n1 < 2 || len < 12) {   // prime (or tiny): one pass
    const PassGeo g{len, n_lines, 1, 0, estride, 0, estride, lstride, 1, len, false};
    dft_pass_g<INV>(re, im, g, tw);
    return;
  }
  const PassGeo a{n1, n_lines, n2, estride, n2 * estride, estride, n2 * estride, lstride, n2, len, true};
  dft_pass_g<INV>(re, im, a, tw);
  const PassGeo b{n2, n_lines, n1, n2 * estride, estride, estride, n1 * estride, lstride, n1, len, false};
  dft_pass_g<INV>(re, im, b, tw);
}

// ---- four-step transforms for the common large sizes: N = R x M with a register FFT of length M ------------------------
// X[k1 + R k2] = sum_{n2 < M} w_M^{n2 k2} [ sum_{n1 < R} x[M n1 + n2] w_N^{k1 (M n1 + n2)} ]: a thread owns one line and one k1;
// it walks the line once (N complex multiply-adds, the R-point DFT and the twiddle in one table factor), runs the
// length-M register transform of fft_regs.h on its M partial sums and has M outputs -- N R tasks of ~1 200 instructions
// per pass instead of N^2 / 32 tasks of ~12 000 in dft_pass.  Lanes of a wave own consecutive lines of the same k1, so
// the twiddle reads are broadcasts and the transposed stores (output (line, k) goes to [k][line]: the next pass reads
// rows again) are conflict-free; the lines are read 16 bytes at a time.  Square windows only (the transposes alternate).
template <bool INV> __device__ __forceinline__ void fs_fft(float (&r)[32], float (&i)[32]) { fft32<INV>(r, i); }
template <bool INV> __device__ __forceinline__ void fs_fft(float (&r)[24], float (&i)[24]) { fft_pfa<INV, 3, 8>(r, i); }
template <bool INV> __device__ __forceinline__ void fs_fft(float (&r)[20], float (&i)[20]) { fft_pfa<INV, 5, 4>(r, i); }
template <bool INV> __device__ __forceinline__ void fs_fft(float (&r)[28], float (&i)[28]) { fft_pfa<INV, 7, 4>(r, i); }
template <bool INV> __device__ __forceinline__ void fs_fft(float (&r)[30], float (&i)[30]) { fft_pfa<INV, 15, 2>(r, i); }

typedef float fs_f32x4 __attribute__((ext_vector_type(4)));
typedef float fs_f32x2 __attribute__((ext_vector_type(2)));

// RR: the compile-time R of the table-free first stage (3 | 4), 0 = any R through the twiddle table
template <bool INV, int M, int RR>
__device__ __forceinline__ void fs_pass_r(float* re, float* im, int N, int R, int pitch, const float* tw) {
  constexpr int VEC = M % 4 == 0 ? 4 : 2;
  float ar[M], ai[M];                                       // left undefined for idle threads (they store nothing)
  const int t = (int)threadIdx.x;
  const bool on = t < N * R;
  const int k1 = on ? t / N : 0, line = on ? t - k1 * N : 0;
  if constexpr (RR == 3 || RR == 4) {
   if (on) {
    // R = 3 | 4: the R-point stage needs no table -- w_4 = -+i costs nothing and w_3 two constants -- so a sample group
    // x[n2], x[M + n2], ... is combined with 8 multiply-adds whose factors are per-thread constants (branch-free: the
    // threads of a wave may differ in k1), and only the combined value meets a twiddle w_N^(k1 n2); k1 n2 < N, so the
    // table index needs no wrap.  15 instead of 9.5 R instructions per n2.
    const float* lr = re + line * pitch;
    const float* li = im + line * pitch;
    const float dir = INV ? -1.0f : 1.0f;
    //   R = 4:  A = x0 + sg x2,  B = x1 + sg x3,  y = A + (c1 + i c2) B
    //   R = 3:  S = x1 + x2,     D = x1 - x2,     y = x0 + c1 S + i c2 D
    const float sg = (k1 & 1) ? -1.0f : 1.0f;
    const float c1 = RR == 4 ? (k1 == 0 ? 1.0f : k1 == 2 ? -1.0f : 0.0f) : (k1 == 0 ? 1.0f : -0.5f);
    const float c2 = RR == 4 ? dir * (k1 == 1 ? -1.0f : k1 == 3 ? 1.0f : 0.0f)
                            : dir * (k1 == 0 ? 0.0f : k1 == 1 ? -0.8660254037844386f : 0.8660254037844386f);
    const float* twk = tw;
    const int tstep = 2 * k1;
#pragma unroll
    for (int n2 = 0; n2 < M; n2 += VEC) {
      float xr[RR][VEC], xi[RR][VEC];
#pragma unroll
      for (int n1 = 0; n1 < RR; ++n1) {
        if constexpr (VEC == 4) {
          const fs_f32x4 a = *reinterpret_cast<const fs_f32x4*>(lr + n1 * M + n2), b = *reinterpret_cast<const fs_f32x4*>(li + n1 * M + n2);
#pragma unroll
          for (int e = 0; e < 4; ++e) { xr[n1][e] = a[e]; xi[n1][e] = b[e]; }
        } else {
          const fs_f32x2 a = *reinterpret_cast<const fs_f32x2*>(lr + n1 * M + n2), b = *reinterpret_cast<const fs_f32x2*>(li + n1 * M + n2);
          xr[n1][0] = a[0]; xr[n1][1] = a[1]; xi[n1][0] = b[0]; xi[n1][1] = b[1];
        }
      }
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        float yr, yi;
        if constexpr (RR == 4) {
          const float Ar = fmaf(sg, xr[2][e], xr[0][e]), Ai = fmaf(sg, xi[2][e], xi[0][e]);
          const float Br = fmaf(sg, xr[3][e], xr[1][e]), Bi = fmaf(sg, xi[3][e], xi[1][e]);
          yr = fmaf(-c2, Bi, fmaf(c1, Br, Ar));
          yi = fmaf(c2, Br, fmaf(c1, Bi, Ai));
        } else {
          const float Sr = xr[1][e] + xr[2][e], Si = xi[1][e] + xi[2][e];
          const float Dr = xr[1][e] - xr[2][e], Di = xi[1][e] - xi[2][e];
          yr = fmaf(-c2, Di, fmaf(c1, Sr, xr[0][e]));
          yi = fmaf(c2, Dr, fmaf(c1, Si, xi[0][e]));
        }
        const fs_f32x2 w = *reinterpret_cast<const fs_f32x2*>(twk);
        twk += tstep;
        const float c = w[0], sn = INV ? -w[1] : w[1];
        ar[n2 + e] = fmaf(yr, c, yi * sn);                  // (yr + i yi)(c - i s)
        ai[n2 + e] = fmaf(yi, c, -yr * sn);
      }
    }
    fs_fft<INV>(ar, ai);
   }
  } else if (on) {
#pragma unroll
    for (int q = 0; q < M; ++q) ar[q] = ai[q] = 0.0f;
    const float* lr = re + line * pitch;
    const float* li = im + line * pitch;
    int idx = 0;                                            // (k1 n) mod N
    for (int n1 = 0; n1 < R; ++n1, lr += M, li += M) {
#pragma unroll
      for (int n2 = 0; n2 < M; n2 += VEC) {
        float xr[VEC], xi[VEC];
        if constexpr (VEC == 4) {
          const fs_f32x4 a = *reinterpret_cast<const fs_f32x4*>(lr + n2), b = *reinterpret_cast<const fs_f32x4*>(li + n2);
          xr[0] = a[0]; xr[1] = a[1]; xr[2] = a[2]; xr[3] = a[3]; xi[0] = b[0]; xi[1] = b[1]; xi[2] = b[2]; xi[3] = b[3];
        } else {
          const fs_f32x2 a = *reinterpret_cast<const fs_f32x2*>(lr + n2), b = *reinterpret_cast<const fs_f32x2*>(li + n2);
          xr[0] = a[0]; xr[1] = a[1]; xi[0] = b[0]; xi[1] = b[1];
        }
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          const fs_f32x2 w = *reinterpret_cast<const fs_f32x2*>(tw + 2 * idx);
          const float c = w[0], sn = INV ? -w[1] : w[1];
          idx += k1;
          idx = idx >= N ? idx - N : idx;
          ar[n2 + e] = fmaf(xr[e], c, fmaf(xi[e], sn, ar[n2 + e]));     // (xr + i xi)(c - i s)
          ai[n2 + e] = fmaf(xi[e], c, fmaf(-xr[e], sn, ai[n2 + e]));
        }
      }
    }
    fs_fft<INV>(ar, ai);
  }
  __syncthreads();   // every output is in registers: the buffer may be overwritten
  if (on) {
#pragma unroll
    for (int k2 = 0; k2 < M; ++k2) {
      const int o = (k1 + R * k2) * pitch + line;           // transposed: the next pass reads rows again
      re[o] = ar[k2];
      im[o] = ai[k2];
    }
  }
  __syncthreads();
}

template <bool INV, int M>
__device__ __forceinline__ void fs_pass(float* re, float* im, int N, int R, int pitch, const float* tw) {
  if (R == 4) fs_pass_r<INV, M, 4>(re, im, N, R, pitch, tw);        // block-uniform
  else if (R == 3) fs_pass_r<INV, M, 3>(re, im, N, R, pitch, tw);
  else fs_pass_r<INV, M, 0>(re, im, N, R, pitch, tw);
}

// the register-FFT length the four-step passes use for a square window of side n (0: none, take the DFT passes)
__host__ __device__ inline int fourstep_m(int wy, int wx) {
  if (wy != wx) return 0;
  const int n = wy;
  const int cand[5] = {32, 24, 30, 20, 28};
  for (int i = 0; i < 5; ++i) {
    const int m = cand[i];
    if (n % m == 0 && n / m >= 2 && n * (n / m) <= FBLOCK) return m;
  }
  return 0;
}

// one window pair -> clipped, fft-shifted plane in `plane` (n floats, aliases the imaginary plane).  false: NaN plane.
template <typename T, int M>
// `data`: the two planes (2 wy pitch floats; LDS, or a slot of HBM scratch for windows that outgrow a CU's LDS), `small`:
// the twiddle tables and the reduction scratch (LDS)
__device__ __forceinline__ bool dft_pair(const PivParams& p, uint32_t pair, uint32_t win, float* data, float* small, const DftGeo& g,
                                         float*& plane, float*& red) {
  const int wy = p.wy, wx = p.wx, P = g.pitch;
  float* re = data;
  float* im = re + wy * P;
  float* twx = small;
  float* twy = twx + 2 * wx;
  red = twy + 2 * wy;
  plane = im;
  const T* frames = static_cast<const T*>(p.frames);
  const uint32_t wrow = win / (uint32_t)p.n_cols, wcol = win - wrow * (uint32_t)p.n_cols;
  const int64_t off = ((int64_t)pair * p.H + (int64_t)wrow * p.sy) * p.W + (int64_t)wcol * p.sx;
  for (int m = threadIdx.x; m < wx + wy; m += blockDim.x) {   // twiddles: double precision, rounded once
    const bool isx = m < wx;
    const int k = isx ? m : m - wx, L = isx ? wx : wy;
    double sv, cv;
    sincospi(2.0 * (double)k / (double)L, &sv, &cv);
    float* tw = isx ? twx : twy;
    tw[2 * k] = (float)cv;
    tw[2 * k + 1] = (float)sv;
  }
  int nza, nzb;
  bool finite = true;
  float sa, sb;                                             // sums of the clipped windows
  const float inv_a = p.std_gain * stage_window(frames + off, p.W, wy, wx, re, P, false, p.nz_positive != 0, red, nza, finite, &sa, p.norm_clip != 0);
  const float inv_b = p.std_gain * stage_window(frames + off + p.frame_elems, p.W, wy, wx, im, P, false, p.nz_positive != 0, red, nzb, finite, &sb, p.norm_clip != 0);
  __syncthreads();
  bool ok = finite;
  if (p.signal_threshold >= 0.0f) {
    const float fa = (float)nza / (float)g.n, fb = (float)nzb / (float)g.n;
    ok = ok && (fa >= p.signal_threshold) && (fb >= p.signal_threshold);
    if (p.win_keep) ok = ok && p.win_keep[win];
  }
  // unit variance and 1 / n on each window: both spectra are O(1) (balanced packing) and the product of two of them
  // carries the 1 / n^2 the plane needs; a zero-variance window gives an exactly-zero plane
  const bool dead = inv_a == 0.0f || inv_b == 0.0f;
  const float ga = dead ? 0.0f : inv_a / (float)g.n, gb = dead ? 0.0f : inv_b / (float)g.n;
  // The clipped windows are non-negative, so their mean (the DC bin) is ~20x a typical AC bin, and every term of the
  // transform sums carries rounding noise relative to IT.  The mean only shifts the plane by a constant,
  //     sum_x (a~ + ma)(x) (b~ + mb)(x + d) = sum_x a~(x) b~(x + d) + n ma mb,
  // so the transforms run on the de-meaned windows and the constant n^2 ma mb is added back before the clip.
  const float ma = sa * ga / (float)g.n, mb = sb * gb / (float)g.n;   // means of the SCALED windows
  const RowCol rc0(wx);
  RowCol rc = rc0;
  for (int o = threadIdx.x; o < g.n; o += blockDim.x, rc.next()) {
    const int a0 = rc.y * P + rc.x;
    re[a0] = re[a0] * ga - ma;
    im[a0] = im[a0] * gb - mb;
  }
  const float plane_dc = (float)g.n * (float)g.n * ma * mb;
  __syncthreads();
  if constexpr (M > 0) {
    fs_pass<false, M>(re, im, wx, wx / M, P, twx);   // along x, written transposed
    fs_pass<false, M>(re, im, wx, wx / M, P, twx);   // along y, transposed back -> Z[ky][kx]
  } else {
    dft_pass<false>(re, im, wx, wy, 1, P, twx);   // along x, one line per row
    dft_pass<false>(re, im, wy, wx, P, 1, twy);   // along y, one line per column -> Z[ky][kx]
  }
  // cross spectrum in place: the thread that owns k also owns -k (k <= -k in row-major order)
  rc = rc0;
  for (int o = threadIdx.x; o < g.n; o += blockDim.x, rc.next()) {
    const int ky = rc.y, kx = rc.x;
    const int my = ky == 0 ? 0 : wy - ky, mx = kx == 0 ? 0 : wx - kx;
    const int om = my * wx + mx;
    if (o > om) continue;
    const int a0 = ky * P + kx, a1 = my * P + mx;
    const float zr = re[a0], zi = im[a0], mr = re[a1], mi = im[a1];
    const float Ar = 0.5f * (zr + mr), Ai = 0.5f * (zi - mi);      // A = (Z[k] + conj Z[-k]) / 2
    const float Br = 0.5f * (zi + mi), Bi = -0.5f * (zr - mr);     // B = (Z[k] - conj Z[-k]) / 2i
    const float Rr = Ar * Br + Ai * Bi, Ri = Ar * Bi - Ai * Br;    // conj(A) B
    re[a0] = Rr; im[a0] = (o == om) ? 0.0f : Ri;                   // self-conjugate bins are real
    if (o != om) { re[a1] = Rr; im[a1] = -Ri; }
  }
  __syncthreads();
  if constexpr (M > 0) {
    fs_pass<true, M>(re, im, wx, wx / M, P, twx);
    fs_pass<true, M>(re, im, wx, wx / M, P, twx);
  } else {
    dft_pass<true>(re, im, wy, wx, P, 1, twy);    // along ky
    dft_pass<true>(re, im, wx, wy, 1, P, twx);    // along kx -> correlation at lag (dy, dx) in re[dy][dx]
  }
  const int cy = wy / 2, cx = wx / 2;
  const float hi = dead ? 0.0f : 1.0f;
  rc = rc0;
  for (int o = threadIdx.x; o < g.n; o += blockDim.x, rc.next()) {   // clip, fft-shift into the (now free) imaginary plane
    const int ip = rc.y, jp = rc.x;
    const int dy = ip - cy < 0 ? ip - cy + wy : ip - cy, dx = jp - cx < 0 ? jp - cx + wx : jp - cx;
    im[o] = __builtin_amdgcn_fmed3f(re[dy * P + dx] + plane_dc, 0.0f, hi);
  }
  __syncthreads();
  return ok;
}

template <typename T, int M>
__global__ __launch_bounds__(FBLOCK, 4) void piv_dft_kernel(PivParams p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const DftGeo g(p.wy, p.wx);
  const uint32_t t = blockIdx.x;
  const uint32_t pair = t / p.n_win, win = t - pair * p.n_win;
  float *plane, *red;
  const bool ok = dft_pair<T, M>(p, pair, win, smem, smem + 2 * p.wy * g.pitch, g, plane, red);
  float vmax, sum, u, v;
  int imax;
  plane_reduce(plane, g.n, red, vmax, imax, sum);
  subpixel_generic([&](int o) { return plane[o]; }, p.wy, p.wx, imax, p.border_mode, u, v);
  block_rescue_note(p, plane, g.n, red, vmax, imax, u, v, t, ok);
  float cm = vmax, sn = vmax / (sum / (float)g.n);
  if (!ok) u = v = cm = sn = __builtin_nanf("");
  if (threadIdx.x == 0) {
    p.u[t] = u; p.v[t] = v; p.cmax[t] = cm; p.s2n[t] = sn;
  }
  if (p.planes) {
    float* dst = p.planes + (size_t)t * g.n;
    for (int o = threadIdx.x; o < g.n; o += blockDim.x) dst[o] = ok ? plane[o] : __builtin_nanf("");
  }
}

// ensemble: one block owns one window and walks the chunk's pairs in order (as piv_direct_ensemble_kernel)
template <typename T, int M>
__global__ __launch_bounds__(FBLOCK) void piv_dft_ensemble_kernel(PivParams p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const DftGeo g(p.wy, p.wx);
  const uint32_t win = blockIdx.x;
  float* dst = p.corr_sum + (size_t)win * g.n;
  float cnt = 0.0f;
  for (uint32_t pair = 0; pair < p.n_pairs; ++pair) {
    float *plane, *red;
    const bool ok = dft_pair<T, M>(p, pair, win, smem, smem + 2 * p.wy * g.pitch, g, plane, red);
    float vmax, sum;
    int imax;
    plane_reduce(plane, g.n, red, vmax, imax, sum);
    float cm = vmax, sn = vmax / (sum / (float)g.n);
    const bool keep = ok && (cm >= p.corr_min) && (sn >= p.s2n_min);
    cm = keep ? cm : 0.0f;
    sn = keep ? sn : 0.0f;
    cnt += (cm > 1e-6f) ? 1.0f : 0.0f;
    if (threadIdx.x == 0) {
      p.cmax[(size_t)pair * p.n_win + win] = cm;
      p.s2n[(size_t)pair * p.n_win + win] = sn;
    }
    if (keep)
      for (int o = threadIdx.x; o < g.n; o += blockDim.x) dst[o] += plane[o];
    __syncthreads();
  }
  if (threadIdx.x == 0) p.corr_count[win] += cnt;
}

// ---- windows above 128 px: the same transforms on a slot of HBM scratch ---------------------------------------------------------
// Two planes of 129 x 129 complex samples no longer fit the 160 KB of a CU.  ffpiv.cross_corr has no upper bound on the
// window, so these sizes run the SAME code -- dft_pair on the DFT passes (a composite length as two shorter passes, oversized
// tile sets in rounds) -- with the planes in a per-block slot of HBM scratch instead of LDS; only the twiddle tables and the
// reduction scratch stay in LDS.  A persistent grid (one slot per block) walks the (pair, window) tiles.  This is a
// functional path, not a fast one: every pass goes through L2 / MALL.
template <typename T>
__global__ __launch_bounds__(FBLOCK) void piv_dft_global_kernel(PivParams p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const DftGeo g(p.wy, p.wx);
  float* data = p.dft_scratch + (size_t)blockIdx.x * p.dft_slot;
  for (uint32_t t = blockIdx.x; t < p.n_tiles; t += gridDim.x) {
    const uint32_t pair = t / p.n_win, win = t - pair * p.n_win;
    float *plane, *red;
    const bool ok = dft_pair<T, 0>(p, pair, win, data, smem, g, plane, red);
    float vmax, sum, u, v;
    int imax;
    plane_reduce(plane, g.n, red, vmax, imax, sum);
    subpixel_generic([&](int o) { return plane[o]; }, p.wy, p.wx, imax, p.border_mode, u, v);
    block_rescue_note(p, plane, g.n, red, vmax, imax, u, v, t, ok);
    float cm = vmax, sn = vmax / (sum / (float)g.n);
    if (!ok) u = v = cm = sn = __builtin_nanf("");
    if (threadIdx.x == 0) {
      p.u[t] = u; p.v[t] = v; p.cmax[t] = cm; p.s2n[t] = sn;
    }
    if (p.planes) {
      float* dst = p.planes + (size_t)t * g.n;
      for (int o = threadIdx.x; o < g.n; o += blockDim.x) dst[o] = ok ? plane[o] : __builtin_nanf("");
    }
    __syncthreads();   // the next tile overwrites the slot
  }
}

template <typename T>
__global__ __launch_bounds__(FBLOCK) void piv_dft_global_ensemble_kernel(PivParams p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const DftGeo g(p.wy, p.wx);
  float* data = p.dft_scratch + (size_t)blockIdx.x * p.dft_slot;
  for (uint32_t win = blockIdx.x; win < p.n_win; win += gridDim.x) {   // a block owns a window and walks the chunk's pairs in order
    float* dst = p.corr_sum + (size_t)win * g.n;
    float cnt = 0.0f;
    for (uint32_t pair = 0; pair < p.n_pairs; ++pair) {
      float *plane, *red;
      const bool ok = dft_pair<T, 0>(p, pair, win, data, smem, g, plane, red);
      float vmax, sum;
      int imax;
      plane_reduce(plane, g.n, red, vmax, imax, sum);
      float cm = vmax, sn = vmax / (sum / (float)g.n);
      const bool keep = ok && (cm >= p.corr_min) && (sn >= p.s2n_min);
      cm = keep ? cm : 0.0f;
      sn = keep ? sn : 0.0f;
      cnt += (cm > 1e-6f) ? 1.0f : 0.0f;
      if (threadIdx.x == 0) {
        p.cmax[(size_t)pair * p.n_win + win] = cm;
        p.s2n[(size_t)pair * p.n_win + win] = sn;
      }
      if (keep)
        for (int o = threadIdx.x; o < g.n; o += blockDim.x) dst[o] += plane[o];
      __syncthreads();
    }
    if (threadIdx.x == 0) p.corr_count[win] += cnt;
  }
}

size_t piv_dft_global_slot_floats(int wy, int wx) { return (((size_t)2 * wy * (wx | 1)) + 63) & ~(size_t)63; }
int piv_dft_global_blocks(uint32_t n_work) { return (int)std::min<uint32_t>(n_work, 512u); }   // two slots per CU

hipError_t launch_piv_dft_global(const PivParams& p, int dtype, bool ensemble, hipStream_t s) {
  if (!p.dft_scratch) return hipErrorInvalidValue;
  const size_t lds = ((size_t)2 * (p.wx + p.wy) + 24) * sizeof(float);
  const dim3 grid((unsigned)piv_dft_global_blocks(ensemble ? p.n_win : p.n_tiles));
#define LSPIV_DFT_G(T)                                                                                       \
  do {                                                                                                       \
    if (ensemble) hipLaunchKernelGGL((piv_dft_global_ensemble_kernel<T>), grid, dim3(FBLOCK), lds, s, p);    \
    else hipLaunchKernelGGL((piv_dft_global_kernel<T>), grid, dim3(FBLOCK), lds, s, p);                      \
  } while (0)
  switch (dtype) {
    case 0: LSPIV_DFT_G(uint8_t); break;
    case 1: LSPIV_DFT_G(float); break;
    case 2: LSPIV_DFT_G(double); break;
    default: return hipErrorInvalidValue;
  }
#undef LSPIV_DFT_G
  return hipGetLastError();
}

size_t piv_dft_lds_bytes(int wy, int wx) { return DftGeo(wy, wx).lds_floats(wy, wx) * sizeof(float); }
bool piv_dft_fits(int wy, int wx) {
  const int tiles = ((wy + 3) / 4) * ((wx + 3) / 4);
  return tiles <= FBLOCK * FTILES && piv_dft_lds_bytes(wy, wx) <= (size_t)160 * 1024 - 512;
}

template <typename T, int M>
static hipError_t launch_dft_tm(const PivParams& p, bool ensemble, hipStream_t s) {
  const size_t lds = piv_dft_lds_bytes(p.wy, p.wx);
  hipError_t e;
  if (ensemble) {
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&piv_dft_ensemble_kernel<T, M>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((piv_dft_ensemble_kernel<T, M>), dim3(p.n_win), dim3(FBLOCK), lds, s, p);
  } else {
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&piv_dft_kernel<T, M>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((piv_dft_kernel<T, M>), dim3(p.n_tiles), dim3(FBLOCK), lds, s, p);
  }
  return hipGetLastError();
}

template <typename T>
static hipError_t launch_dft_t(const PivParams& p, bool ensemble, hipStream_t s) {
  static const bool no_fourstep = getenv("LSPIV_NO_FOURSTEP") != nullptr;   // A/B and cross-check: DFT passes for every size
  switch (no_fourstep ? 0 : fourstep_m(p.wy, p.wx)) {
    case 32: return launch_dft_tm<T, 32>(p, ensemble, s);
    case 24: return launch_dft_tm<T, 24>(p, ensemble, s);
    case 30: return launch_dft_tm<T, 30>(p, ensemble, s);
    case 20: return launch_dft_tm<T, 20>(p, ensemble, s);
    case 28: return launch_dft_tm<T, 28>(p, ensemble, s);
    default: return launch_dft_tm<T, 0>(p, ensemble, s);
  }
}

hipError_t launch_piv_dft(const PivParams& p, int dtype, bool ensemble, hipStream_t s) {
  if (!piv_dft_fits(p.wy, p.wx)) return hipErrorInvalidValue;
  switch (dtype) {
    case 0: return launch_dft_t<uint8_t>(p, ensemble, s);
    case 1: return launch_dft_t<float>(p, ensemble, s);
    case 2: return launch_dft_t<double>(p, ensemble, s);
    default: return hipErrorInvalidValue;
  }
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
__global__ __launch_bounds__(64) void peaks_kernel(const float* planes, uint32_t n_planes, int wy, int wx, int border_mode,
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
  // a NaN plane (a window skipped by the signal threshold) stays NaN whatever the border mode
  subpixel_generic([&](int o) { return pl[o]; }, wy, wx, bi, pl[bi] != pl[bi] ? 0 : border_mode, uu, vv);
  if (lane == 0) { u[g] = uu; v[g] = vv; }
}

hipError_t launch_peaks_from_planes(const float* planes, uint32_t n_planes, int wy, int wx, int border_mode, float* u, float* v,
                                    hipStream_t s) {
  if (n_planes == 0) return hipSuccess;
  hipLaunchKernelGGL(peaks_kernel, dim3(n_planes), dim3(64), 0, s, planes, n_planes, wy, wx, border_mode, u, v);
  return hipGetLastError();
}

// ---- "stack" signal mode (second reading of pyorc/velocimetry/ffpiv.py:93-97) ----------
// One score per window POSITION: the fraction of non-zero (or positive) samples of that window over all T frames of the
// chunk; positions below the threshold are dropped for every pair.  One block per window position, exact integer count.
template <typename T>
__global__ __launch_bounds__(256) void window_signal_kernel(const T* frames, int64_t n_frames, PivParams p, float thr,
                                                            uint8_t* keep) {
  __shared__ int red[8];
  const uint32_t win = blockIdx.x;
  const uint32_t wrow = win / (uint32_t)p.n_cols, wcol = win - wrow * (uint32_t)p.n_cols;
  const T* base = frames + (int64_t)wrow * p.sy * p.W + (int64_t)wcol * p.sx;
  const int n = p.wy * p.wx;
  unsigned long long cnt = 0;
  for (int64_t f = 0; f < n_frames; ++f) {
    const T* src = base + f * p.frame_elems;
    for (int o = threadIdx.x; o < n; o += blockDim.x) {
      const int y = o / p.wx, x = o - y * p.wx;
      const float v = to_f32(src[(int64_t)y * p.W + x]);
      cnt += (p.nz_positive ? v > 0.0f : v != 0.0f) ? 1 : 0;
    }
  }
  // counts fit 32 bits per lane for any realistic chunk; reduce in two steps to stay exact
  int lo = (int)(cnt & 0xffffu), hi = (int)(cnt >> 16);
  lo = block_sum_i(lo, red);
  __syncthreads();
  hi = block_sum_i(hi, red);
  if (threadIdx.x == 0) {
    const double total = (double)hi * 65536.0 + (double)lo;
    const float score = (float)(total / ((double)n_frames * (double)n));
    keep[win] = score >= thr ? 1 : 0;
  }
}

hipError_t launch_window_signal(const void* frames, int dtype, int64_t T, const PivParams& p, float thr, uint8_t* keep, hipStream_t s) {
  switch (dtype) {
    case 0: hipLaunchKernelGGL(window_signal_kernel<uint8_t>, dim3(p.n_win), dim3(256), 0, s, (const uint8_t*)frames, T, p, thr, keep); break;
    case 1: hipLaunchKernelGGL(window_signal_kernel<float>, dim3(p.n_win), dim3(256), 0, s, (const float*)frames, T, p, thr, keep); break;
    case 2: hipLaunchKernelGGL(window_signal_kernel<double>, dim3(p.n_win), dim3(256), 0, s, (const double*)frames, T, p, thr, keep); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

}  // namespace lspiv
