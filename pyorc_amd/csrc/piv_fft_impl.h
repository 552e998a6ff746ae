// Fused N x N interrogation-window kernels for gfx950 (CDNA4, wave64): N = 32 and 64 (the sizes the design is tuned for),
// 8 and 16, and every other even N = P * 2^m up to 62 through prime-factor transforms (fft_regs.h); one instantiation
// file per size (piv_fftNN.hip).
//
// Replaces, in ONE launch per frame chunk, what the reference does in three passes over a
// (T-1, n_win, N, N) float volume (pyorc/velocimetry/ffpiv.py:446-474):
//   ffpiv.cross_corr       window gather, per-window normalise, rfft2 . conj-mul . irfft2,
//                          fftshift, /N^2, clip[0,1]                     
//   numpy reductions       corr_max = nanmax(plane), s2n = corr_max / nanmean(plane)     (K6)
//   ffpiv.u_v_displacement argmax + 3-point log-Gaussian sub-pixel fit                   (K7)
// The correlation planes never leave the CU unless the caller asks for them.
//
// Mapping (there is no reference kernel; this is an MI355X design):
//   * a "job" is TWO windows of one frame pair processed by a group of N lanes (a half-wave for
//     N = 32, a whole wave for N = 64), lane = tile row (tile column after a transpose), the N
//     complex values of that row in VGPRs;
//   * window pair (a from frame t, b from frame t+1) is packed z = a + i b, so the two real
//     forward FFTs cost one complex 2-D FFT;  R = conj(A) B follows from Z[k], Z[-k]:
//         4 R[k] = 2 Im(Z[k] Z[-k]) - i (|Z[k]|^2 - |Z[-k]|^2)
//     R is Hermitian (the correlation is real), bit-exactly so in this formula, therefore a lane
//     computes and keeps only R[ky][kx] for ky = 0..N/2; the rest is the conjugate of what the
//     mirrored lane (-kx) holds;
//   * the two windows of a job share ONE inverse transform: IFFT(R1 + i R2) = c1 + i c2;
//   * length-N transforms are straight-line register code (fft_regs.h); the 2-D transposes go
//     through a padded per-group LDS tile, real and imaginary plane one after the other (N = 32:
//     four 4-wave workgroups fit the 160 KB of a CU); Z[-k] comes from the mirrored lane with
//     ds_bpermute; per-window mean / variance / max / argmax / sum are DPP reductions;
//   * a wave can issue one VALU instruction every 4 cycles while a SIMD retires ~1.6 from >= 3
//     waves (tools/ubench/valu_rate.hip), so N = 32 is built for FOUR waves per SIMD (<= 128 VGPRs,
//     9 KB of LDS per wave); N = 64 needs 254 VGPRs and runs two.
// MFMA is deliberately unused: this is FFT + pointwise work (BASELINE.json north_star).
#pragma once
#include <algorithm>
#include <cstdlib>

#include "common.h"
#include "fft_regs.h"

namespace lspiv {

constexpr int WAVES_PER_BLOCK = 4;
constexpr int BLOCK = 64 * WAVES_PER_BLOCK;

template <int N>
struct Geo {
  static constexpr int NN = N * N;                  // samples per window
  static constexpr int HALF = N / 2;
  static constexpr bool POW2 = (N & (N - 1)) == 0;
  static constexpr int LG = N <= 16 ? 16 : N <= 32 ? 32 : 64;   // lanes per job; lanes >= N idle along (N = 12, 20, 24, 28, 36 ...)
  static constexpr bool FULL = N == LG;             // every lane of the group owns a tile row (16, 32, 64)
  static constexpr int GROUPS = 64 / LG;            // jobs per wave
  static constexpr int LDS_ROW = N % 4 == 0 ? N + 4 : N + 2;   // dwords per padded row: 16-byte aligned, an odd number of 4-bank slots
  static constexpr int LDS_JOB = N * LDS_ROW;       // dwords per group buffer
  static constexpr int LDS_BYTES = WAVES_PER_BLOCK * GROUPS * LDS_JOB * 4;
};

// index helpers valid for the non-power-of-two sizes too (3 * 2^m): i in [0, 2N)
template <int N> __device__ __forceinline__ int wrap_n(int i) {
  if constexpr (Geo<N>::POW2) return i & (N - 1);
  else return i >= N ? i - N : i;
}
// lane -> the tile row it works on: lanes >= N of a group clone row N - 1 (they load, transform and reduce the same
// values as lane N - 1 but never write LDS or HBM, and are masked out of sums / arg-min reductions)
template <int N> __device__ __forceinline__ int row_of(int lg) {
  if constexpr (Geo<N>::FULL) return lg;
  else return lg < N ? lg : N - 1;
}
template <int N> __device__ __forceinline__ bool lane_active(int lg) {
  if constexpr (Geo<N>::FULL) return true;
  else return lg < N;
}
template <int N> __device__ __forceinline__ int partner_byte_of(int lane, int lg) {   // byte address of lane -kx
  constexpr int LG = Geo<N>::LG;
  if constexpr (Geo<N>::FULL) return ((lane & ~(LG - 1)) | ((N - lg) & (N - 1))) << 2;
  else { const int r = row_of<N>(lg); return ((lane & ~(LG - 1)) | (r == 0 ? 0 : N - r)) << 2; }
}

template <int N> __device__ __forceinline__ int group_lane() { return (int)(threadIdx.x & (unsigned)(Geo<N>::LG - 1)); }
template <int N> __device__ __forceinline__ int lane0_byte_of() {   // byte address of the group's lane 0 (ds_bpermute)
  return (int)((threadIdx.x & 63u) & ~(unsigned)(Geo<N>::LG - 1)) << 2;
}

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef u32x2 u32x2_u __attribute__((aligned(1)));
typedef u32x4 u32x4_u __attribute__((aligned(1)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef f32x4 f32x4_u __attribute__((aligned(4)));
typedef double f64x2 __attribute__((ext_vector_type(2)));
typedef f64x2 f64x2_u __attribute__((aligned(8)));

__device__ __forceinline__ float bperm_f(int addr, float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(addr, __builtin_bit_cast(int, v)));
}

// ---- reductions over the N lanes of a group ----------------------------------------------------
template <int N> __device__ __forceinline__ float group_sum(float x) { return Geo<N>::LG == 16 ? row_sum(x) : Geo<N>::LG == 32 ? half_sum(x) : wave_sum(x); }
template <int N> __device__ __forceinline__ int group_sum_i(int x) {
  if (Geo<N>::LG == 16) return row_sum_i(x);
  x = half_sum_i(x);
  if (Geo<N>::LG == 64) x = cross_half_sum_i(x);
  return x;
}
template <int N> __device__ __forceinline__ float group_max(float x) {
  if (Geo<N>::LG == 16) return row_max(x);
  x = half_max(x);
  if (Geo<N>::LG == 64) x = cross_half_max(x);
  return x;
}
// maximum of NON-NEGATIVE floats (clipped planes) over the group, on the bit patterns (common.h, half_max_i)
template <int N> __device__ __forceinline__ float group_max_nonneg(float x) {
  int b = __builtin_bit_cast(int, x);
  if (Geo<N>::LG == 16) return __builtin_bit_cast(float, row_max_i(b));
  b = half_max_i(b);
  if (Geo<N>::LG == 64) b = cross_half_max_i(b);
  return __builtin_bit_cast(float, b);
}
// "some lane of my group has `cond`": one ballot and scalar / per-lane mask arithmetic instead of a DPP reduction
template <int N> __device__ __forceinline__ bool group_any(bool cond) {
  const uint64_t m = __builtin_amdgcn_ballot_w64(cond);
  if constexpr (Geo<N>::LG == 64) return m != 0;
  else if constexpr (Geo<N>::LG == 32) return ((threadIdx.x & 32u) ? (uint32_t)(m >> 32) : (uint32_t)m) != 0u;
  else return ((m >> (threadIdx.x & 48u)) & 0xffffull) != 0ull;
}
template <int N> __device__ __forceinline__ int group_min_i(int x) {
  if (Geo<N>::LG == 16) return row_min_i(x);
  x = half_min_i(x);
  if (Geo<N>::LG == 64) x = cross_half_min_i(x);
  return x;
}

// pairwise (tree) sum of a register row: exact for constant rows, short dependency chains
template <int N>
__device__ __forceinline__ float tree_sum(const float (&x)[N]) {
  float s[N / 2];
#pragma unroll
  for (int k = 0; k < N / 2; ++k) s[k] = x[2 * k] + x[2 * k + 1];
  if constexpr (Geo<N>::POW2) {
#pragma unroll
    for (int w = N / 4; w >= 1; w >>= 1) {
#pragma unroll
      for (int k = 0; k < w; ++k) s[k] = s[2 * k] + s[2 * k + 1];
    }
  } else {
#pragma unroll
    for (int cnt = N / 2; cnt > 1;) {
      const int h = cnt / 2;
#pragma unroll
      for (int k = 0; k < h; ++k) s[k] = s[2 * k] + s[2 * k + 1];
      if (cnt & 1) s[h] = s[cnt - 1];
      cnt = h + (cnt & 1);
    }
  }
  return s[0];
}

#ifndef LSPIV_ENS64_NT
#define LSPIV_ENS64_NT 0   // 1: non-temporal frame loads in the 64 x 64 ensemble kernel (RowRaw::fetch<NT>).  Measured and left OFF (round 6):
                           // written bytes 26.9 -> 21.4 GB only, fetched 3.3 -> 15.4 GB (the 75 % overlap of the windows lives on L2 hits
                           // of the frame lines, which non-temporal loads give up), 28.7 -> 31.5 ms.  What evicts the slots is their own
                           // address pattern, not the frames: see kEnsSplitHalves below
#endif
// ---- one tile row as fetched from HBM -----------------------------------------------------------
// uint8 rows are N/4 dwords, cheap enough to prefetch for BOTH windows of a job before any arithmetic
// starts; float rows are N..2N dwords, so only their address is kept and the load is issued where the
// samples are consumed (the other waves of the SIMD cover that latency).
template <typename T, int N>
struct RowRaw {
  const T* p;
  __device__ __forceinline__ void fetch(const T* q) { p = q; }
  __device__ __forceinline__ void mask(bool) {}
};
template <int N>
struct RowRaw<uint8_t, N> {
  static constexpr int W = (N + 3) / 4;   // N % 4 == 2: the last word holds two samples and two zero bytes
  uint32_t w[W];
  // NT: non-temporal loads -- the frame bytes stream through the L2 without displacing what lives there (an A/B switch of the 64 x 64
  // ensemble kernel, LSPIV_ENS64_NT; off: it costs the frame lines' own reuse more than it saves)
  template <bool NT = false>
  __device__ __forceinline__ void fetch(const uint8_t* q) {
#pragma unroll
    for (int k = 0; k < N / 16; ++k) {
      const u32x4 v = NT ? __builtin_nontemporal_load(reinterpret_cast<const u32x4_u*>(q + 16 * k)) : *reinterpret_cast<const u32x4_u*>(q + 16 * k);
      w[4 * k] = v[0]; w[4 * k + 1] = v[1]; w[4 * k + 2] = v[2]; w[4 * k + 3] = v[3];
    }
    constexpr int done = N / 16 * 16;
    if constexpr (N - done >= 8) {
      const u32x2 v = *reinterpret_cast<const u32x2_u*>(q + done);
      w[done / 4] = v[0]; w[done / 4 + 1] = v[1];
    }
    constexpr int done8 = N / 8 * 8;
    if constexpr (N - done8 >= 4) {
      uint32_t v;
      __builtin_memcpy(&v, q + done8, 4);
      w[done8 / 4] = v;
    }
    if constexpr (N % 4 == 2) {
      uint16_t v;
      __builtin_memcpy(&v, q + (N - 2), 2);
      w[W - 1] = v;
    }
    if constexpr (!Geo<N>::FULL) mask(lane_active<N>(group_lane<N>()));
  }
  // idle lanes of a group (row_of): zero bytes count for nothing in the window sums
  __device__ __forceinline__ void mask(bool active) {
#pragma unroll
    for (int k = 0; k < W; ++k) w[k] = active ? w[k] : 0u;
  }
};

struct RowStats {
  float mean;     // window mean
  float inv_std;  // 1 / population std, 0 for a zero-variance window
};

// ffpiv normalize_intensity (A3): (a - mean)/std (0 if std == 0), clipped to >= 0.
// uint8: sum and sum of squares are exact integers (v_dot4_u32_u8 on the packed bytes), variance from
// n sum(x^2) - (sum x)^2 in 64 bits -- no per-sample arithmetic besides convert / fma / clip.
template <int N>
__device__ __forceinline__ RowStats stats_u8(const RowRaw<uint8_t, N>& raw, bool want_nz, int& nonzero) {
  constexpr int NN = Geo<N>::NN;
  uint32_t s = 0, q = 0;
#pragma unroll
  for (int k = 0; k < (N + 3) / 4; ++k) {
    s = __builtin_amdgcn_udot4(raw.w[k], 0x01010101u, s, false);
    q = __builtin_amdgcn_udot4(raw.w[k], raw.w[k], q, false);
  }
  if (want_nz) {
    int nz = 0;
#pragma unroll
    for (int k = 0; k < (N + 3) / 4; ++k) {  // 0x80 in every zero byte (the padding bytes of a last half word are zero)
      const uint32_t t = ~(((raw.w[k] & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | raw.w[k] | 0x7F7F7F7Fu);
      nz += 4 - __builtin_popcount(t);
    }
    nonzero = group_sum_i<N>(nz);
  }
  const uint32_t S = (uint32_t)group_sum_i<N>((int)s);  // <= 255 N^2
  const uint32_t Q = (uint32_t)group_sum_i<N>((int)q);  // <= 255^2 N^2 < 2^31 for N <= 64
  RowStats st;
  st.mean = (float)S * (1.0f / NN);                      // exact
  const uint64_t n2var = (uint64_t)Q * NN - (uint64_t)S * (uint64_t)S;
  const float var = (float)n2var * (1.0f / ((float)NN * (float)NN));
  st.inv_std = n2var != 0 ? __builtin_amdgcn_rsqf(var) : 0.0f;
  return st;
}

// x = max((byte - mean) * g, 0), g >= 0: convert + fma + max per sample.  BELOW1: the caller guarantees every result
// is < 1 (the walking kernels fold 1 / (2 N^2) into g, and |x - mean| / std <= sqrt(N^2 - 1) for any window), so the
// lower clip can be written as med3(., 0, 1), which folds into the clamp modifier of the FMA: convert + fma per sample.
template <bool BELOW1> __device__ __forceinline__ float clip0(float v) { return BELOW1 ? __builtin_amdgcn_fmed3f(v, 0.0f, 1.0f) : fmaxf(v, 0.0f); }
template <int N, bool BELOW1 = false>
__device__ __forceinline__ void center_u8(const RowRaw<uint8_t, N>& raw, float mean, float g, float (&x)[N]) {
  const float off = -mean * g;
#pragma unroll
  for (int k = 0; k < N / 4; ++k) {
    const uint32_t w = raw.w[k];
    x[4 * k + 0] = clip0<BELOW1>(fmaf((float)(w & 0xffu), g, off));
    x[4 * k + 1] = clip0<BELOW1>(fmaf((float)((w >> 8) & 0xffu), g, off));
    x[4 * k + 2] = clip0<BELOW1>(fmaf((float)((w >> 16) & 0xffu), g, off));
    x[4 * k + 3] = clip0<BELOW1>(fmaf((float)(w >> 24), g, off));
  }
  if constexpr (N % 4 == 2) {
    const uint32_t w = raw.w[N / 4];
    x[N - 2] = clip0<BELOW1>(fmaf((float)(w & 0xffu), g, off));
    x[N - 1] = clip0<BELOW1>(fmaf((float)((w >> 8) & 0xffu), g, off));
  }
}

// float rows: pairwise sum (a constant window gives its value, hence zero variance, exactly)
template <int N, bool DEFER_CLIP = false>
__device__ __forceinline__ float center_clip_f(float (&x)[N], bool want_nz, bool nz_pos, int& nonzero, bool& finite) {
  constexpr float inv_nn = 1.0f / Geo<N>::NN;
  const bool active = lane_active<N>(group_lane<N>());   // constant true for the power-of-two sizes
  if (want_nz) {
    int c = 0;
#pragma unroll
    for (int k = 0; k < N; ++k) c += (nz_pos ? x[k] > 0.0f : x[k] != 0.0f) ? 1 : 0;   // "non-zero" | "above zero" (A7)
    nonzero = group_sum_i<N>(active ? c : 0);
  }
  if constexpr (!Geo<N>::POW2) {   // N^2 is no power of two: shifted mean, so a constant window has exactly zero variance
    const float x0 = bperm_f(lane0_byte_of<N>(), x[0]);
#pragma unroll
    for (int k = 0; k < N; ++k) x[k] -= x0;
  }
  const float s = group_sum<N>(active ? tree_sum<N>(x) : 0.0f);
  const float mean = s * inv_nn;
  float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
  for (int k = 0; k < N; ++k) {
    const float d = x[k] - mean;
    acc[k & 3] = fmaf(d, d, acc[k & 3]);
    x[k] = DEFER_CLIP ? d : fmaxf(d, 0.0f);   // DEFER_CLIP: the caller clips while it scales (prepare_one)
  }
  const float ssq = group_sum<N>(active ? (acc[0] + acc[1]) + (acc[2] + acc[3]) : 0.0f);
  finite = finite && (fabsf(s) <= 3.0e38f) && (ssq <= 3.0e38f);
  const float var = ssq * inv_nn;
  return var > 0.0f ? __builtin_amdgcn_rsqf(var) : 0.0f;
}
template <int N>
__device__ __forceinline__ void load_row_f32(const float* p, float (&x)[N]) {
#pragma unroll
  for (int k = 0; k < N / 4; ++k) {
    const f32x4 v = *reinterpret_cast<const f32x4_u*>(p + 4 * k);
    x[4 * k + 0] = v[0]; x[4 * k + 1] = v[1]; x[4 * k + 2] = v[2]; x[4 * k + 3] = v[3];
  }
  if constexpr (N % 4 == 2) { x[N - 2] = p[N - 2]; x[N - 1] = p[N - 1]; }
}
template <int N, bool DEFER_CLIP = false>
__device__ __forceinline__ float load_center(const RowRaw<float, N>& raw, float (&x)[N], bool want_nz, bool nz_pos,
                                             int& nonzero, bool& finite) {
  load_row_f32<N>(raw.p, x);
  return center_clip_f<N, DEFER_CLIP>(x, want_nz, nz_pos, nonzero, finite);
}
template <int N, bool DEFER_CLIP = false>
__device__ __forceinline__ float load_center(const RowRaw<double, N>& raw, float (&x)[N], bool want_nz, bool nz_pos,
                                             int& nonzero, bool& finite) {
#pragma unroll
  for (int k = 0; k < N / 2; ++k) {
    const f64x2 v = *reinterpret_cast<const f64x2_u*>(raw.p + 2 * k);
    x[2 * k + 0] = (float)v[0]; x[2 * k + 1] = (float)v[1];
  }
  return center_clip_f<N, DEFER_CLIP>(x, want_nz, nz_pos, nonzero, finite);
}

// Both windows of a pair -> xr = a'' (mean-offset, zero-clipped), xi = rho b''.
// Balance: b is rescaled to a's variance (rho = inv_b / inv_a) so |A| ~ |B| and the
// |Z[k]|^2 - |Z[-k]|^2 difference of the cross spectrum does not cancel catastrophically when one
// window is much fainter than the other; corr = inv_a inv_b corr(a'', b'') = inv_a^2 corr(a'', rho b'').
// `scale` (= inv_a^2 / (4 N^4)) goes onto R BEFORE the two windows of a job are packed into one inverse
// transform: both planes then peak at <= 1, so float32 rounding of the shared inverse is relative to
// O(1) for each of them (scaling after the inverse lets a bright window's rounding noise swamp a faint
// neighbour packed with it).  A zero-variance window gives an exactly-zero plane (scale 0, clip ceiling
// hi = 0), the reference's zeros-if-std-is-0 rule (A3).
template <int N>
__device__ __forceinline__ void finish_pair(float inv_a, float inv_b, float& rho, float& scale, float& hi) {
  constexpr float k = 1.0f / (4.0f * (float)Geo<N>::NN * (float)Geo<N>::NN);
  const bool dead = (inv_a == 0.0f) || (inv_b == 0.0f);
  rho = dead ? 0.0f : inv_b * __builtin_amdgcn_rcpf(inv_a);
  scale = dead ? 0.0f : inv_a * inv_a * k;
  hi = dead ? 0.0f : 1.0f;
}
template <int N>
__device__ __forceinline__ bool below_threshold(int nza, int nzb, float thr) {
  constexpr float inv_nn = 1.0f / Geo<N>::NN;
  const float fa = (float)nza * inv_nn, fb = (float)nzb * inv_nn;
  return !(fa >= thr && fb >= thr);
}
template <int N>
__device__ __forceinline__ void prepare_pair(const RowRaw<uint8_t, N>& ra, const RowRaw<uint8_t, N>& rb,
                                             float (&xr)[N], float (&xi)[N], bool want_nz, float thr, bool /*nz_pos*/,
                                             float& scale, float& hi, bool& skip) {
  int nza = Geo<N>::NN, nzb = Geo<N>::NN;
  const RowStats sa = stats_u8<N>(ra, want_nz, nza);
  const RowStats sb = stats_u8<N>(rb, want_nz, nzb);
  float rho;
  finish_pair<N>(sa.inv_std, sb.inv_std, rho, scale, hi);
  center_u8<N>(ra, sa.mean, 1.0f, xr);
  center_u8<N>(rb, sb.mean, rho, xi);  // the balance factor rides on the conversion
  skip = want_nz && below_threshold<N>(nza, nzb, thr);
}
template <typename T, int N>
__device__ __forceinline__ void prepare_pair(const RowRaw<T, N>& ra, const RowRaw<T, N>& rb, float (&xr)[N],
                                             float (&xi)[N], bool want_nz, float thr, bool nz_pos, float& scale,
                                             float& hi, bool& skip) {
  bool finite = true;
  int nza = Geo<N>::NN, nzb = Geo<N>::NN;
  const float inv_a = load_center<N>(ra, xr, want_nz, nz_pos, nza, finite);
  const float inv_b = load_center<N>(rb, xi, want_nz, nz_pos, nzb, finite);
  float rho;
  finish_pair<N>(inv_a, inv_b, rho, scale, hi);
#pragma unroll
  for (int j = 0; j < N; ++j) xi[j] *= rho;
  skip = !finite || (want_nz && below_threshold<N>(nza, nzb, thr));
}

struct TileRef {
  uint32_t pair;   // frame pair index inside the chunk
  uint32_t win;    // window index k * n_cols + m
  bool valid;
};


// ---- embedded mode: a square n x n window, 4 <= n <= N/2, through the N-point transforms ------------------------
// Serves the ODD square windows (which only the C ABI can ask for: pyorc rounds window sizes to even, and every even size
// has FFT kernels of its own) and the LSPIV_NO_PFA=1 cross-check path.  Such a size has no FFT of its own here, but its
// CIRCULAR correlation is exact inside a larger transform: a' zero-padded to
// N x N, b' extended periodically (b'[y mod n][x mod n]); then for lags 0 <= k < n
//     sum_{m < n} a'[m] b'_per[m + k] = sum_m a'[m] b'[(m + k) mod n]        (m + k <= 2n - 2 < N: no wrap of the big FFT)
// in both axes, i.e. the top-left n x n corner of the N x N plane IS the n-point circular correlation; the other lags
// are never looked at.  Statistics (mean, std, non-zero fraction) are those of the n x n windows; the mean is taken as
// x0 + mean(x - x0), x0 the first sample, so a constant window has exactly zero variance for any n (the power-of-two
// kernels get that from pairwise sums).  All sample types take the float path here.
//
// One window of the pair: load (zero-padded rows for a, periodic rows for b), statistics over the n x n samples,
// x <- max(x - mean, 0); returns 1/std (0 for a zero-variance window).  The n x n samples are columns j < n (a uniform
// test: scalar branches, no per-element lane masks) of rows < n (one lane mask, applied to the row totals).
// Branch-free: "column j belongs to the window" is the uniform float m_j = (j < n) (a scalar select) and "this lane's
// row belongs to it" the lane float r; masking is multiplication, so load + statistics are one basic block (the
// per-column scalar branches of the other version cost more than the ~200 extra multiplies).  Used by the 64-point
// variant; in the 32-point variant it needs 40 more VGPRs (2 waves instead of 3) and loses.
template <typename T, int N, bool WANT_NZ, bool PERIODIC>
__device__ __forceinline__ float load_center_embed_bf(const T* row, int n, bool row_in, int lane0_byte, bool nz_pos,
                                                      float (&x)[N], int& nonzero, bool& finite) {
  const float inv_nn = 1.0f / (float)(n * n);
  const float r = row_in ? 1.0f : 0.0f;
  int jm = 0;
#pragma unroll
  for (int j = 0; j < N; ++j) {
    if (PERIODIC) {
      x[j] = to_f32(row[jm]);
      jm = (jm + 1 == n) ? 0 : jm + 1;
    } else {
      x[j] = to_f32(row[j < n ? j : n - 1]);   // always in bounds; columns >= n are masked below
    }
    if (sizeof(T) == 8 && (j & 15) == 15) __builtin_amdgcn_sched_barrier(0);   // 8-byte samples: consume in chunks
  }
  const float x0 = bperm_f(lane0_byte, x[0]);
  float s = 0.0f, nz = 0.0f;
#pragma unroll
  for (int j = 0; j < N; ++j) {
    const float m = j < n ? 1.0f : 0.0f;
    s = fmaf(x[j] - x0, m, s);
    if (WANT_NZ) nz += (nz_pos ? x[j] > 0.0f : x[j] != 0.0f) ? m : 0.0f;
  }
  if (WANT_NZ) nonzero = group_sum_i<N>((int)(nz * r));
  const float mean = x0 + group_sum<N>(s * r) * inv_nn;
  float q = 0.0f;
#pragma unroll
  for (int j = 0; j < N; ++j) {
    const float m = j < n ? 1.0f : 0.0f;
    const float d = x[j] - mean;
    q = fmaf(d * m, d, q);
    x[j] = PERIODIC ? fmaxf(d, 0.0f) : fmaxf(d, 0.0f) * (m * r);
  }
  q = group_sum<N>(q * r);
  finite = finite && (fabsf(mean) <= 3.0e38f) && (q <= 3.0e38f);
  const float var = q * inv_nn;
  return var > 0.0f ? __builtin_amdgcn_rsqf(var) : 0.0f;
}
// uniform scalar branches on the column test (32-point variant)
template <typename T, int N, bool WANT_NZ, bool PERIODIC>
__device__ __forceinline__ float load_center_embed_br(const T* row, int n, bool row_in, int lane0_byte, bool nz_pos,
                                                      float (&x)[N], int& nonzero, bool& finite) {
  const float inv_nn = 1.0f / (float)(n * n);
  int jm = 0;
#pragma unroll
  for (int j = 0; j < N; ++j) {
    if (PERIODIC) {
      x[j] = to_f32(row[jm]);
      jm = (jm + 1 == n) ? 0 : jm + 1;
    } else {
      x[j] = (j < n && row_in) ? to_f32(row[j]) : 0.0f;   // j < n is uniform: the padding costs no load
    }
    if (sizeof(T) == 8 && (j & 15) == 15) __builtin_amdgcn_sched_barrier(0);   // 8-byte samples: consume in chunks
  }
  const float x0 = bperm_f(lane0_byte, x[0]);
  float s = 0.0f;
  int nz = 0;
#pragma unroll
  for (int j = 0; j < N; ++j) {
    if (j < n) {
      s += x[j] - x0;
      if (WANT_NZ) nz += (nz_pos ? x[j] > 0.0f : x[j] != 0.0f) ? 1 : 0;
    }
  }
  if (WANT_NZ) nonzero = group_sum_i<N>(row_in ? nz : 0);
  const float mean = x0 + group_sum<N>(row_in ? s : 0.0f) * inv_nn;
  float q = 0.0f;
#pragma unroll
  for (int j = 0; j < N; ++j) {
    if (PERIODIC || j < n) {
      const float d = x[j] - mean;
      if (j < n) q += d * d;
      x[j] = fmaxf(d, 0.0f);
    }
  }
  if (!PERIODIC) {
#pragma unroll
    for (int j = 0; j < N; ++j) x[j] = row_in ? x[j] : 0.0f;
  }
  q = group_sum<N>(row_in ? q : 0.0f);
  finite = finite && (fabsf(mean) <= 3.0e38f) && (q <= 3.0e38f);
  const float var = q * inv_nn;
  return var > 0.0f ? __builtin_amdgcn_rsqf(var) : 0.0f;
}

template <typename T, int N, bool WANT_NZ, bool PERIODIC>
__device__ __forceinline__ float load_center_embed(const T* row, int n, bool row_in, int lane0_byte, bool nz_pos,
                                                   float (&x)[N], int& nonzero, bool& finite) {
  if constexpr (N == 64) return load_center_embed_bf<T, N, WANT_NZ, PERIODIC>(row, n, row_in, lane0_byte, nz_pos, x, nonzero, finite);
  else return load_center_embed_br<T, N, WANT_NZ, PERIODIC>(row, n, row_in, lane0_byte, nz_pos, x, nonzero, finite);
}

template <typename T, int N, bool WANT_NZ>
__device__ __forceinline__ void prepare_pair_embed(const PivParams& p, const TileRef& t, int lg, float (&xr)[N],
                                                   float (&xi)[N], float& scale, float& hi, bool& skip) {
  const int n = p.wy;
  const float inv_nn = 1.0f / (float)(n * n);
  const T* frames = static_cast<const T*>(p.frames);
  const uint32_t wrow = p.div_ncols.div(t.win);
  const uint32_t wcol = t.win - wrow * (uint32_t)p.n_cols;
  const int64_t base = ((int64_t)t.pair * p.H + (int64_t)wrow * p.sy) * p.W + (int64_t)wcol * p.sx;
  const bool row_in = lg < n;
  const int lane0_byte = lane0_byte_of<N>();
  bool finite = true;
  int nza = 0, nzb = 0;
  const bool nz_pos = p.nz_positive != 0;
  const float inv_a = load_center_embed<T, N, WANT_NZ, false>(frames + base + (int64_t)(row_in ? lg : 0) * p.W, n, row_in,
                                                               lane0_byte, nz_pos, xr, nza, finite);
  __builtin_amdgcn_sched_barrier(0);   // one window after the other: only one raw row in flight next to the finished one
  const float inv_b = load_center_embed<T, N, WANT_NZ, true>(frames + base + p.frame_elems + (int64_t)(lg % n) * p.W, n,
                                                              row_in, lane0_byte, nz_pos, xi, nzb, finite);
  const bool dead = (inv_a == 0.0f) || (inv_b == 0.0f);
  const float rho = dead ? 0.0f : inv_b * __builtin_amdgcn_rcpf(inv_a);
  // plane = (unnormalised inverse N x N transform) / N^2 / n^2, and the cross-spectrum formula carries a factor 4
  scale = dead ? 0.0f : inv_a * inv_a * inv_nn * (1.0f / (4.0f * (float)Geo<N>::NN)) * p.std_gain2;
  hi = dead ? 0.0f : 1.0f;
#pragma unroll
  for (int j = 0; j < N; ++j) xi[j] *= rho;
  skip = !finite;
  if (WANT_NZ) {
    const float fa = (float)nza * inv_nn, fb = (float)nzb * inv_nn;
    skip = skip || !(fa >= p.signal_threshold && fb >= p.signal_threshold);
    if (p.win_keep) skip = skip || !p.win_keep[t.win];   // "stack" mode: one score per window position (A7)
  }
}

// one row of a parked plane <-> registers: ds_read/write_b128, plus one b64 for the two last samples when N % 4 == 2
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int N>
__device__ __forceinline__ void lds_row_read(const float* row, float (&x)[N]) {
  const f32x4* r4 = reinterpret_cast<const f32x4*>(row);
#pragma unroll
  for (int q = 0; q < N / 4; ++q) {
    const f32x4 v = r4[q];
    x[4 * q] = v[0]; x[4 * q + 1] = v[1]; x[4 * q + 2] = v[2]; x[4 * q + 3] = v[3];
  }
  if constexpr (N % 4 == 2) {
    const f32x2 v = *reinterpret_cast<const f32x2*>(row + (N - 2));
    x[N - 2] = v[0]; x[N - 1] = v[1];
  }
}
template <int N>
__device__ __forceinline__ void lds_row_write(float* row, const float (&c)[N]) {
  f32x4* w4 = reinterpret_cast<f32x4*>(row);
#pragma unroll
  for (int q = 0; q < N / 4; ++q) {
    const f32x4 w = {c[4 * q], c[4 * q + 1], c[4 * q + 2], c[4 * q + 3]};
    w4[q] = w;
  }
  if constexpr (N % 4 == 2) {
    const f32x2 w = {c[N - 2], c[N - 1]};
    *reinterpret_cast<f32x2*>(row + (N - 2)) = w;
  }
}

// LDS transpose of one real N x N plane held as lane = row: lane r scatters its row down column r of
// the buffer (ds_write_b32, the lanes of a group hit consecutive banks), then reads buffer row r =
// tile column r with ds_read_b128 (row stride N+4 dwords: 16-byte aligned, and the 16 lanes of a b128
// group land on 16 distinct 4-bank slots since (N/4+1) l mod 16 is a bijection).
// (The compiler pairs the 32 column stores into ds_write2_b32, whose 8-bit offsets reach only 1 KB, and keeps six extra base
// addresses in VGPRs for it; an inline-asm scatter with immediate offsets off one base register was measured for the
// 4-waves-per-SIMD attempts of the walking kernel and dropped with them -- DESIGN.md section 3.1b, git history.)
template <int N>
__device__ __forceinline__ void transpose_plane(float* buf, int lg, float (&x)[N]) {
  constexpr int LR = Geo<N>::LDS_ROW;
  float* wcol = buf + lg;
  if (lane_active<N>(lg)) {
#pragma unroll
    for (int j = 0; j < N; ++j) wcol[j * LR] = x[j];
  }
  __builtin_amdgcn_wave_barrier();  // same wave: LDS ops execute in order, this only pins the compiler
  lds_row_read<N>(buf + row_of<N>(lg) * LR, x);
  __builtin_amdgcn_wave_barrier();
}
// 64 x 64 transpose through HALF the tile: [[A B] [C D]]^T = [[A^T C^T] [B^T D^T]].  The two lane halves first exchange B and C (32
// v_permlane32_swap), then each half transposes its two 32 x 32 blocks in place through its own 32 x 36-float region (the upper
// half's region shifted by 32 dwords: other banks): kHalfTileDwords = 2 x 1152 + 32 dwords = 9.3 KB instead of 17.4, no transient
// registers.  + 2.4 % in the per-timestep kernel (128 swaps per iteration), + 0.4 % in the ensemble kernel, which uses the 8 KB it frees.
constexpr int kHalfTileDwords = 2 * 32 * 36 + 32;
__device__ __forceinline__ void transpose_plane_half64(float* buf, int lg, float (&x)[64]) {
  constexpr int P = 36;
  const int l = lg & 31;
  float* reg = buf + (lg >> 5) * (32 * P + 32);
#pragma unroll
  for (int j = 0; j < 32; ++j) {
    unsigned a = __builtin_bit_cast(unsigned, x[j]), b = __builtin_bit_cast(unsigned, x[32 + j]);
    permlane32_swap(a, b);                       // x[j] of the upper lanes <-> x[32 + j] of the lower lanes
    x[j] = __builtin_bit_cast(float, a); x[32 + j] = __builtin_bit_cast(float, b);
  }
#pragma unroll
  for (int h = 0; h < 2; ++h) {
#pragma unroll
    for (int j = 0; j < 32; ++j) reg[j * P + l] = x[32 * h + j];
    __builtin_amdgcn_wave_barrier();
    const f32x4* r4 = reinterpret_cast<const f32x4*>(reg + l * P);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const f32x4 v = r4[q];
      x[32 * h + 4 * q] = v[0]; x[32 * h + 4 * q + 1] = v[1]; x[32 * h + 4 * q + 2] = v[2]; x[32 * h + 4 * q + 3] = v[3];
    }
    __builtin_amdgcn_wave_barrier();
  }
}
template <int N, bool HALF = false>
__device__ __forceinline__ void transpose2(float* buf, int lg, float (&xr)[N], float (&xi)[N]) {
  if constexpr (HALF && N == 64) {
    transpose_plane_half64(buf, lg, xr);
    transpose_plane_half64(buf, lg, xi);
  } else {
    transpose_plane<N>(buf, lg, xr);
    transpose_plane<N>(buf, lg, xi);
  }
}

// inverse transform whose outputs are clipped to [0, 1] for free (clamp modifier of the last butterfly stage) -- the power-of-two
// sizes; kClampInFft<N> tells the callers whether they still have to clip
template <int N> constexpr bool kClampInFft = N == 8 || N == 16 || N == 32 || N == 64;
__device__ __forceinline__ void ifft_clamped(float (&xr)[8], float (&xi)[8]) { bfly8<true, true>(xr, xi); }
__device__ __forceinline__ void ifft_clamped(float (&xr)[16], float (&xi)[16]) { fft16<true, true>(xr, xi); }
__device__ __forceinline__ void ifft_clamped(float (&xr)[32], float (&xi)[32]) { fft32<true, true>(xr, xi); }
__device__ __forceinline__ void ifft_clamped(float (&xr)[64], float (&xi)[64]) { fft64<true, true>(xr, xi); }
template <bool INV> __device__ __forceinline__ void fft_n(float (&xr)[8], float (&xi)[8]) { bfly8<INV>(xr, xi); }
template <bool INV> __device__ __forceinline__ void fft_n(float (&xr)[16], float (&xi)[16]) { fft16<INV>(xr, xi); }
template <bool INV> __device__ __forceinline__ void fft_n(float (&xr)[32], float (&xi)[32]) { fft32<INV>(xr, xi); }
template <bool INV> __device__ __forceinline__ void fft_n(float (&xr)[64], float (&xi)[64]) { fft64<INV>(xr, xi); }
template <bool INV> __device__ __forceinline__ void fft_n(float (&xr)[12], float (&xi)[12]) { fft_pfa<INV, 3, 4>(xr, xi); }
template <bool INV> __device__ __forceinline__ void fft_n(float (&xr)[20], float (&xi)[20]) { fft_pfa<INV, 5, 4>(xr, xi); }
template <bool INV> __device__ __forceinline__ void fft_n(float (&xr)[24], float (&xi)[24]) { fft_pfa<INV, 3, 8>(xr, xi); }
template <bool INV> __device__ __forceinline__ void fft_n(float (&xr)[40], float (&xi)[40]) { fft_pfa<INV, 5, 8>(xr, xi); }
template <bool INV> __device__ __forceinline__ void fft_n(float (&xr)[48], float (&xi)[48]) { fft_pfa<INV, 3, 16>(xr, xi); }
template <bool INV> __device__ __forceinline__ void fft_n(float (&xr)[6], float (&xi)[6]) { fft_pfa<INV, 3, 2>(xr, xi); }
template <bool INV> __device__ __forceinline__ void fft_n(float (&xr)[10], float (&xi)[10]) { fft_pfa<INV, 5, 2>(xr, xi); }
template <bool INV> __device__ __forceinline__ void fft_n(float (&xr)[14], float (&xi)[14]) { fft_pfa<INV, 7, 2>(xr, xi); }
template <bool INV> __device__ __forceinline__ void fft_n(float (&xr)[18], float (&xi)[18]) { fft_pfa<INV, 9, 2>(xr, xi); }
template <bool INV> __device__ __forceinline__ void fft_n(float (&xr)[22], float (&xi)[22]) { fft_pfa<INV, 11, 2>(xr, xi); }
template <bool INV> __device__ __forceinline__ void fft_n(float (&xr)[26], float (&xi)[26]) { fft_pfa<INV, 13, 2>(xr, xi); }
template <bool INV> __device__ __forceinline__ void fft_n(float (&xr)[30], float (&xi)[30]) { fft_pfa<INV, 15, 2>(xr, xi); }
template <bool INV> __device__ __forceinline__ void fft_n(float (&xr)[34], float (&xi)[34]) { fft_pfa<INV, 17, 2>(xr, xi); }
template <bool INV> __device__ __forceinline__ void fft_n(float (&xr)[38], float (&xi)[38]) { fft_pfa<INV, 19, 2>(xr, xi); }
template <bool INV> __device__ __forceinline__ void fft_n(float (&xr)[42], float (&xi)[42]) { fft_pfa<INV, 21, 2>(xr, xi); }
template <bool INV> __device__ __forceinline__ void fft_n(float (&xr)[46], float (&xi)[46]) { fft_pfa<INV, 23, 2>(xr, xi); }
template <bool INV> __device__ __forceinline__ void fft_n(float (&xr)[50], float (&xi)[50]) { fft_pfa<INV, 25, 2>(xr, xi); }
template <bool INV> __device__ __forceinline__ void fft_n(float (&xr)[54], float (&xi)[54]) { fft_pfa<INV, 27, 2>(xr, xi); }
template <bool INV> __device__ __forceinline__ void fft_n(float (&xr)[58], float (&xi)[58]) { fft_pfa<INV, 29, 2>(xr, xi); }
template <bool INV> __device__ __forceinline__ void fft_n(float (&xr)[62], float (&xi)[62]) { fft_pfa<INV, 31, 2>(xr, xi); }
template <bool INV> __device__ __forceinline__ void fft_n(float (&xr)[28], float (&xi)[28]) { fft_pfa<INV, 7, 4>(xr, xi); }
template <bool INV> __device__ __forceinline__ void fft_n(float (&xr)[36], float (&xi)[36]) { fft_pfa<INV, 9, 4>(xr, xi); }
template <bool INV> __device__ __forceinline__ void fft_n(float (&xr)[44], float (&xi)[44]) { fft_pfa<INV, 11, 4>(xr, xi); }
template <bool INV> __device__ __forceinline__ void fft_n(float (&xr)[52], float (&xi)[52]) { fft_pfa<INV, 13, 4>(xr, xi); }
template <bool INV> __device__ __forceinline__ void fft_n(float (&xr)[56], float (&xi)[56]) { fft_pfa<INV, 7, 8>(xr, xi); }
template <bool INV> __device__ __forceinline__ void fft_n(float (&xr)[60], float (&xi)[60]) { fft_pfa<INV, 15, 4>(xr, xi); }


// lane = kx, registers = ky hold Z = FFT2(a + i b).  Writes s * 4 conj(A) B for ky = 0..N/2 into
// (rr, ri); Z[-k] = mirrored lane's Z[N - ky].
template <int N>
__device__ __forceinline__ void cross_spectrum_half(int partner_byte, const float (&zr)[N], const float (&zi)[N],
                                                    float s, float (&rr)[N / 2 + 1], float (&ri)[N / 2 + 1]) {
#pragma unroll
  for (int ky = 0; ky <= N / 2; ++ky) {
    const int kn = (N - ky) % N;
    const float wr = bperm_f(partner_byte, zr[kn]);
    const float wi = bperm_f(partner_byte, zi[kn]);
    const float ar = zr[ky], ai = zi[ky];
    rr[ky] = (2.0f * s) * (ar * wi + ai * wr);
    ri[ky] = s * ((wr * wr + wi * wi) - (ar * ar + ai * ai));
  }
}

// Everything between "two window pairs" and "two clipped correlation planes in registers".
// On return xr = plane of tile 0, xi = plane of tile 1, natural (un-shifted) order: lane = row y,
// register = column x;  skip[k] = plane k is NaN (signal pre-mask / non-finite input).
template <typename T, int N, bool WANT_NZ, bool EMBED = false>
__device__ __forceinline__ void correlate_job(const PivParams& p, const TileRef (&t)[2], float* buf, int lg,
                                              int partner_byte, float (&xr)[N], float (&xi)[N], bool (&skip)[2],
                                              float (&mean)[2]) {
  constexpr int H = N / 2;
  float R1r[H + 1], R1i[H + 1];  // s1 * 4 conj(A1) B1, ky = 0..N/2 (Hermitian half)
  float hi[2];                   // clip ceiling: 1, or 0 for a zero-variance window (plane exactly 0)
  const T* frames = static_cast<const T*>(p.frames);
  constexpr bool want_nz = WANT_NZ;  // compile-time: a run-time branch here splits the pipeline into basic
                                     // blocks and the register allocator spills across them
  RowRaw<T, N> raw[2][2];
  auto fetch_rows = [&](int k) {
    const uint32_t wrow = p.div_ncols.div(t[k].win);
    const uint32_t wcol = t[k].win - wrow * (uint32_t)p.n_cols;
    const int64_t off = ((int64_t)t[k].pair * p.H + (int64_t)(wrow * p.sy + row_of<N>(lg))) * p.W + (int64_t)wcol * p.sx;
    raw[k][0].fetch(frames + off);
    raw[k][1].fetch(frames + off + p.frame_elems);
  };
  // uint8 rows (8 VGPRs each) are all fetched up front; wider samples are addressed only when their window's turn
  // comes -- four live 64-bit row pointers were exactly the 8 VGPRs that kept the float kernel above 128
  if constexpr (sizeof(T) == 1 && !EMBED) { fetch_rows(0); fetch_rows(1); }
  // 64-point embedding: ONE window per job.  Holding window 0's half spectrum (66 VGPRs) through window 1's scalar
  // loads and masked statistics does not fit 256 VGPRs (200-900 B/lane of scratch, and slower than doing without the
  // shared inverse), so the second slot of the inverse transform stays empty there: 2 instead of 1.5 transforms per
  // window, no spills.
  constexpr bool SINGLE = EMBED && N == 64;
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    // keep the two windows' register-hungry phases apart: the scheduler otherwise interleaves window 1's
    // conversion with window 0's column FFT and spills
    __builtin_amdgcn_sched_barrier(0);
    float scale;
    if constexpr (SINGLE) {
      if (k == 1) {   // Q = R1 for ky <= N/2, conj of the mirrored lane above: the imaginary plane comes out ~0
        hi[1] = 0.0f;
        skip[1] = true;
#pragma unroll
        for (int ky = 0; ky <= H; ++ky) { xr[ky] = R1r[ky]; xi[ky] = R1i[ky]; }
#pragma unroll
        for (int ky = 1; ky < H; ++ky) {
          xr[N - ky] = bperm_f(partner_byte, R1r[ky]);
          xi[N - ky] = -bperm_f(partner_byte, R1i[ky]);
        }
        break;
      }
    }
    if constexpr (EMBED) {
      prepare_pair_embed<T, N, WANT_NZ>(p, t[k], lg, xr, xi, scale, hi[k], skip[k]);
    } else {
      if constexpr (sizeof(T) != 1) fetch_rows(k);
      prepare_pair(raw[k][0], raw[k][1], xr, xi, want_nz, p.signal_threshold, p.nz_positive != 0, scale, hi[k], skip[k]);
      scale *= p.std_gain2;   // 1, or (n - 1) / n under the "std_ddof" option
      if (WANT_NZ && p.win_keep) skip[k] = skip[k] || !p.win_keep[t[k].win];   // "stack" mode (A7)
    }
    fft_n<false>(xr, xi);              // along x
    transpose2<N>(buf, lg, xr, xi);    // lane = kx, regs = y
    fft_n<false>(xr, xi);              // along y -> Z[ky][kx]
    if (k == 0) {
      cross_spectrum_half<N>(partner_byte, xr, xi, scale, R1r, R1i);
    } else {
      float R2r[H + 1], R2i[H + 1];
      cross_spectrum_half<N>(partner_byte, xr, xi, scale, R2r, R2i);
      // Q = R1 + i R2 for ky = 0..N/2 directly; for ky > N/2 use R[ky][kx] = conj(R[N-ky][-kx]):
      // Q[ky][kx] = conj( (R1 - i R2)[N-ky][-kx] ), fetched from the mirrored lane.
#pragma unroll
      for (int ky = 0; ky <= H; ++ky) {
        xr[ky] = R1r[ky] - R2i[ky];
        xi[ky] = R1i[ky] + R2r[ky];
      }
#pragma unroll
      for (int ky = 1; ky < H; ++ky) {
        const float mr = R1r[ky] + R2i[ky];   // (R1 - i R2).re
        const float mi = R1i[ky] - R2r[ky];   // (R1 - i R2).im
        xr[N - ky] = bperm_f(partner_byte, mr);
        xi[N - ky] = -bperm_f(partner_byte, mi);
      }
    }
  }
  // mean of a plane = its DC bin: sum_x IFFT(Q)[x] = N^2 Q[0][0], and Q[0][0] = s1 R1[0][0] + i s2 R2[0][0] sits
  // in lane kx = 0 of the group, register ky = 0.  (The reference averages the CLIPPED plane; the clip only
  // removes float rounding below zero -- a ~1e-8 relative difference, measured in the parity tests.)
  const int lane0_byte = lane0_byte_of<N>();
  mean[0] = bperm_f(lane0_byte, xr[0]);
  mean[1] = bperm_f(lane0_byte, xi[0]);
  __builtin_amdgcn_sched_barrier(0);
  fft_n<true>(xr, xi);                 // along ky
  transpose2<N>(buf, lg, xr, xi);      // lane = y, regs = kx
  fft_n<true>(xr, xi);                 // along kx -> c1 + i c2
#pragma unroll
  for (int j = 0; j < N; ++j) {
    xr[j] = __builtin_amdgcn_fmed3f(xr[j], 0.0f, hi[0]);
    xi[j] = __builtin_amdgcn_fmed3f(xi[j], 0.0f, hi[1]);
  }
}

// Row maximum of a plane held as lane = y, reg = x (v_max3 tree) reduced over the group -> plane maximum.
template <int N>
__device__ __forceinline__ float plane_max(const float (&c)[N], float& row_max) {
  // v_max3 tree from the first level on: N values -> ceil(N / 3) -> ... (32 values: 11 + 4 + 2 instructions)
  float m[N];
#pragma unroll
  for (int k = 0; k < N; ++k) m[k] = c[k];
#pragma unroll
  for (int w = N; w > 1;) {
    const int t = w / 3, rem = w - 3 * t;
#pragma unroll
    for (int k = 0; k < t; ++k) m[k] = fmaxf(fmaxf(m[3 * k], m[3 * k + 1]), m[3 * k + 2]);
    if (rem >= 1) m[t] = m[3 * t];
    if (rem == 2) m[t] = fmaxf(m[t], m[3 * t + 1]);
    w = t + (rem ? 1 : 0);
  }
  row_max = m[0];
  return group_max_nonneg<N>(row_max);   // the planes are clipped to [0, 1]
}

// Peak of one plane: parks the plane in LDS (row y at buf[y * LDS_ROW + x]), finds np.argmax of the
// fft-shifted plane (first maximum in row-major order: smallest shifted row holding the maximum, then the
// smallest shifted column of that row -- one lane per column compares its LDS sample, DPP min-reductions),
// and fits the 3-point log-Gaussian.  u, v in pixels; NaN when the peak sits on the plane border.
// note / g: append the window (result index g) to the lists of the float64 rescue pass if its float32 fit cannot be trusted
// (common.h, peak_cond); decided and written here, so that nothing of it stays live in the callers.
// first shifted row holding the plane maximum (register-only: DPP reductions, no LDS), so that a caller with two planes can
// start the second plane's search while the first plane's LDS round trips are in flight
template <int N>
__device__ __forceinline__ int peak_row(int lg, float vmax, float row_max) {
  constexpr int C = N / 2, NONE = 1 << 12;
  const int sh = wrap_n<N>(row_of<N>(lg) + C);
  return group_min_i<N>((lane_active<N>(lg) && row_max == vmax) ? sh : NONE);
}
template <int N>
__device__ __forceinline__ void find_peak(float* buf, int lg, const float (&c)[N], float vmax, float row_max, const PivParams& p,
                                          float& u, float& v, bool note, uint32_t g, int ip_known = -1) {
  const int border_mode = p.border_mode;
  int ip, jp;
  constexpr int LR = Geo<N>::LDS_ROW;
  constexpr int M = N - 1, C = N / 2, NONE = 1 << 12;
  const bool active = lane_active<N>(lg);
  const int lr = row_of<N>(lg);
  if (active) lds_row_write<N>(buf + lg * LR, c);
  __builtin_amdgcn_wave_barrier();
  const int sh = wrap_n<N>(lr + C);                                                  // this lane's shifted row AND column
  ip = ip_known >= 0 ? ip_known : peak_row<N>(lg, vmax, row_max);                    // first shifted row with the maximum
  const int y = wrap_n<N>(ip + C);
  const float rowv = buf[y * LR + lr];                                               // the peak row, one sample per lane
  jp = group_min_i<N>((active && rowv == vmax) ? sh : NONE);                         // first shifted column in that row
  // is any sample other than (ip, jp) within tau of the maximum?  The other rows through their maxima, the rest of the peak
  // row through this lane's sample of it
  const float thr = vmax * (1.0f - p.rescue_tau);
  const bool near_tie = group_any<N>(active && ((row_max >= thr && sh != ip) || (rowv >= thr && sh != jp)));
  const bool border = (ip == 0 || ip == M || jp == 0 || jp == M);
  const int x = wrap_n<N>(jp + C);
  const int ym = wrap_n<N>(ip + C - 1), yp = wrap_n<N>(ip + C + 1);
  const int xm = wrap_n<N>(jp + C - 1), xp = wrap_n<N>(jp + C + 1);
  const float c0 = vmax + kEpsPeak;
  const float cl = buf[ym * LR + x] + kEpsPeak;
  const float cr = buf[yp * LR + x] + kEpsPeak;
  const float cd = buf[y * LR + xm] + kEpsPeak;
  const float cu = buf[y * LR + xp] + kEpsPeak;
  __builtin_amdgcn_wave_barrier();
  // the fit is a ratio of log differences, so any base works: v_log_f32 (log2, 1 ulp) on inputs >= 1e-7
  const float l0 = __builtin_amdgcn_logf(c0);
  float den_v, den_u;
  v = (float)ip + gauss_offset_fast(__builtin_amdgcn_logf(cl), l0, __builtin_amdgcn_logf(cr), den_v) - (float)C;
  u = (float)jp + gauss_offset_fast(__builtin_amdgcn_logf(cd), l0, __builtin_amdgcn_logf(cu), den_u) - (float)C;
  const PeakCond pc = peak_cond(vmax, near_tie, border, cl, cr, den_v, v, cd, cu, den_u, u, p.rescue_k);
  if (note) {
    uint32_t pos2 = 0xffffffffu;
    if (__builtin_amdgcn_ballot_w64(pc.amb) != 0) {
      // cold path (a few windows in 100 000): how many samples are within tau of the maximum, and where is the other one?
      // With exactly two candidates the rescue pass compares their two float64 sums instead of rebuilding the whole plane.
      const int pos1 = (ip << 16) | jp;
      int cnt = 0, other = 0x7fffffff;
#pragma unroll 1
      for (int yy = 0; yy < N; ++yy) {
        const bool cand = active && buf[yy * LR + lr] >= thr;   // column lr of un-shifted row yy
        const int pos = (wrap_n<N>(yy + C) << 16) | sh;
        cnt += cand ? 1 : 0;
        other = (cand && pos != pos1) ? min(other, pos) : other;
      }
      cnt = group_sum_i<N>(cnt);
      other = group_min_i<N>(other);
      if (cnt == 2) pos2 = (uint32_t)other;
    }
    if (lg == 0 && (pc.amb || pc.fit))
      rescue_note(p.rescue_hdr, p.rescue_fit, p.rescue_cap_fit, p.rescue_amb, p.rescue_cap_amb, g, pc, ip, jp, pos2);
  }
  if (border) border_result(border_mode, jp - C, ip - C, u, v);
}

// ---- the two planes of a walking iteration side by side (64 x 64) ----------------------------------------------------
// find_peak does one plane at a time through ONE parked copy in LDS: plane b cannot start before plane a has read its samples,
// and each plane's chain (park -> row search -> LDS read -> column search -> LDS reads -> logs) is latency, not work.  Here
// plane a parks whole in the job's tile as before, plane b parks only the THREE rows its fit reads (the peak's row and its two
// neighbours, written by the lanes that own them once the peak row is known -- 3 * LDS_ROW floats per job of extra LDS), and
// both fits are straight-line code in one block, so the scheduler overlaps the two chains.
// Measured on one box (1000 pairs, interleaved): 64 x 64 -- 2 waves per SIMD, nobody else to fill a wave's LDS waits -- gains
// 1.8 % (30.8 -> 30.2 ms); 32 x 32 at 3 waves loses 1.3 % to the extra parked rows (6.92 -> 7.03 ms) and keeps the
// one-plane-at-a-time fit with plane b's maximum and peak row hoisted.
template <int N> constexpr bool kTwoPlaneEpilogue = N == 64;
template <int N>
struct PeakFit {
  float u, v;
  PeakCond pc;
  int ip, jp;
  float thr;
};
template <int N>
__device__ __forceinline__ void park_three_rows(float* mini, int lg, const float (&c)[N], int ip) {
  constexpr int LR = Geo<N>::LDS_ROW, C = N / 2;
  const int lr = row_of<N>(lg);
  const int y = wrap_n<N>(ip + C), ym = wrap_n<N>(ip + C - 1), yp = wrap_n<N>(ip + C + 1);
  const int slot = lr == ym ? 0 : lr == y ? 1 : lr == yp ? 2 : -1;   // (N >= 3: the three rows are distinct)
  if (lane_active<N>(lg) && slot >= 0) lds_row_write<N>(mini + slot * LR, c);
}
// rows: base + r0 / r1 / r2 = the rows above / of / below the peak (whole plane: un-shifted row indices; three-row park: 0, 1, 2)
template <int N, bool MINI>
__device__ __forceinline__ PeakFit<N> peak_fit(const float* base, int lg, float vmax, float row_max, int ip, const PivParams& p) {
  constexpr int LR = Geo<N>::LDS_ROW;
  constexpr int M = N - 1, C = N / 2, NONE = 1 << 12;
  const bool active = lane_active<N>(lg);
  const int lr = row_of<N>(lg);
  const int sh = wrap_n<N>(lr + C);
  const int r1 = MINI ? 1 : wrap_n<N>(ip + C), r0 = MINI ? 0 : wrap_n<N>(ip + C - 1), r2 = MINI ? 2 : wrap_n<N>(ip + C + 1);
  PeakFit<N> o;
  o.ip = ip;
  const float rowv = base[r1 * LR + lr];                                             // the peak row, one sample per lane
  o.jp = group_min_i<N>((active && rowv == vmax) ? sh : NONE);                       // first shifted column in that row
  o.thr = vmax * (1.0f - p.rescue_tau);
  const bool near_tie = group_any<N>(active && ((row_max >= o.thr && sh != ip) || (rowv >= o.thr && sh != o.jp)));
  const bool border = (ip == 0 || ip == M || o.jp == 0 || o.jp == M);
  const int x = wrap_n<N>(o.jp + C), xm = wrap_n<N>(o.jp + C - 1), xp = wrap_n<N>(o.jp + C + 1);
  const float c0 = vmax + kEpsPeak;
  const float cl = base[r0 * LR + x] + kEpsPeak;
  const float cr = base[r2 * LR + x] + kEpsPeak;
  const float cd = base[r1 * LR + xm] + kEpsPeak;
  const float cu = base[r1 * LR + xp] + kEpsPeak;
  const float l0 = __builtin_amdgcn_logf(c0);
  float den_v, den_u;
  o.v = (float)ip + gauss_offset_fast(__builtin_amdgcn_logf(cl), l0, __builtin_amdgcn_logf(cr), den_v) - (float)C;
  o.u = (float)o.jp + gauss_offset_fast(__builtin_amdgcn_logf(cd), l0, __builtin_amdgcn_logf(cu), den_u) - (float)C;
  o.pc = peak_cond(vmax, near_tie, border, cl, cr, den_v, o.v, cd, cu, den_u, o.u, p.rescue_k);
  if (border) border_result(p.border_mode, o.jp - C, ip - C, o.u, o.v);
  return o;
}
// the rescue record of one plane; `buf` holds the WHOLE plane (cold path: count the candidates within tau of the maximum)
template <int N>
__device__ __forceinline__ void peak_note(const float* buf, int lg, const PeakFit<N>& o, const PivParams& p, uint32_t g) {
  constexpr int LR = Geo<N>::LDS_ROW, C = N / 2;
  const bool active = lane_active<N>(lg);
  const int lr = row_of<N>(lg);
  const int sh = wrap_n<N>(lr + C);
  uint32_t pos2 = 0xffffffffu;
  if (__builtin_amdgcn_ballot_w64(o.pc.amb) != 0) {
    const int pos1 = (o.ip << 16) | o.jp;
    int cnt = 0, other = 0x7fffffff;
#pragma unroll 1
    for (int yy = 0; yy < N; ++yy) {
      const bool cand = active && buf[yy * LR + lr] >= o.thr;   // column lr of un-shifted row yy
      const int pos = (wrap_n<N>(yy + C) << 16) | sh;
      cnt += cand ? 1 : 0;
      other = (cand && pos != pos1) ? min(other, pos) : other;
    }
    cnt = group_sum_i<N>(cnt);
    other = group_min_i<N>(other);
    if (cnt == 2) pos2 = (uint32_t)other;
  }
  if (lg == 0 && (o.pc.amb || o.pc.fit))
    rescue_note(p.rescue_hdr, p.rescue_fit, p.rescue_cap_fit, p.rescue_amb, p.rescue_cap_amb, g, o.pc, o.ip, o.jp, pos2);
}

template <int N>
__device__ __forceinline__ void store_plane_rows(float* dst, int lg, const float (&c)[N], bool nan_plane, bool zero_plane = false) {
  // shifted row i' = (y + N/2) % N receives columns x = N/2..N-1, 0..N/2-1
  if (!lane_active<N>(lg)) return;
  float* row = dst + wrap_n<N>(lg + N / 2) * N;
  const float nanv = __builtin_nanf("");
  if constexpr (N % 4 == 0) {
#pragma unroll
    for (int q = 0; q < N / 4; ++q) {
      f32x4 v;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = nan_plane ? nanv : zero_plane ? 0.0f : c[(4 * q + e + N / 2) % N];
      *reinterpret_cast<f32x4*>(row + 4 * q) = v;
    }
  } else {   // rows of N = 4 m + 2 floats are 8-byte aligned only
#pragma unroll
    for (int q = 0; q < N / 2; ++q) {
      f32x2 v;
#pragma unroll
      for (int e = 0; e < 2; ++e) v[e] = nan_plane ? nanv : zero_plane ? 0.0f : c[(2 * q + e + N / 2) % N];
      *reinterpret_cast<f32x2*>(row + 2 * q) = v;
    }
  }
}

// registers decide the occupancy: 32x32 uint8 fits 127 VGPRs, no scratch -> FOUR waves per SIMD (interleaved A/B on
// one box: 111.0 k pairs/s at 4 waves vs 103-104 k at the 129 VGPRs = 3 waves of an otherwise identical build);
// 32x32 float rows are loaded where they are consumed and need a few more; 64x64 holds 128 + 66 + temporaries
// (two waves, 254 VGPRs)
#ifndef LSPIV_WALK_WAVES
#define LSPIV_WALK_WAVES 3
#endif
#ifndef LSPIV_WAVES_MID
#define LSPIV_WAVES_MID 3
#endif
#ifndef LSPIV_WAVES_32F
#define LSPIV_WAVES_32F 4
#endif
#ifndef LSPIV_WAVES_32U8
#define LSPIV_WAVES_32U8 4
#endif
#ifndef LSPIV_WALK_WAVES_MID
#define LSPIV_WALK_WAVES_MID 3
#endif
// walking kernels: the carried spectrum costs 32 x 32 one wave per SIMD
#ifndef LSPIV_WALK_WAVES_64
#define LSPIV_WALK_WAVES_64 2
#endif
template <typename T, int N>
constexpr int kWalkWaves = N <= 16 ? 4 : (N < 32 && sizeof(T) < 8) ? LSPIV_WALK_WAVES_MID : (N == 32 && sizeof(T) < 8) ? LSPIV_WALK_WAVES : N == 64 ? LSPIV_WALK_WAVES_64 : 2;
template <typename T, int N>
constexpr int kWavesPerSimd = N <= 16 ? 4 : N <= 24 ? LSPIV_WAVES_MID : (N == 32 && sizeof(T) == 1) ? LSPIV_WAVES_32U8 : (N == 32 && sizeof(T) == 4) ? LSPIV_WAVES_32F : 2;

// ---- per-timestep kernel: one job (two neighbouring windows of one pair) per lane group ---------
template <typename T, int N, bool PLANES, bool WANT_NZ>
__global__ __launch_bounds__(BLOCK, (kWavesPerSimd<T, N>)) void piv_fft_kernel(PivParams p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  using G = Geo<N>;
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int grp = lane / G::LG;
  const int lg = lane & (G::LG - 1);
  float* buf = smem + (wave * G::GROUPS + grp) * G::LDS_JOB;
  const int partner_byte = partner_byte_of<N>(lane, lg);

  // XCD-aware block order: block b runs on XCD b % 8; give every XCD one contiguous range of
  // jobs (= contiguous frame pairs) so a frame is pulled into one L2, not eight.
  const uint32_t nb = gridDim.x;
  const uint32_t q = nb >> 3, r = nb & 7u;
  const uint32_t xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3;
  const uint32_t blk = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;

  // A job is windows (2j, 2j+1) of ONE frame pair: a window's partner in the shared inverse
  // transform is then fixed by the window grid alone, so results do not depend on how the time
  // axis was chunked (bit-identical chunk / halo equivalence).  An odd last window pairs with
  // itself.  Jobs past the end recompute the last job and store nothing.
  const uint32_t jobs_per_pair = (p.n_win + 1) >> 1;
  uint32_t job = (blk * WAVES_PER_BLOCK + wave) * G::GROUPS + grp;
  const bool job_valid = job < p.n_pairs * jobs_per_pair;
  job = job_valid ? job : p.n_pairs * jobs_per_pair - 1;
  const uint32_t pair = p.div_jobs.div(job);
  const uint32_t w0 = (job - pair * jobs_per_pair) * 2;
  TileRef t[2];
  t[0].pair = t[1].pair = pair;
  t[0].win = w0;
  t[0].valid = job_valid;
  t[1].valid = job_valid && (w0 + 1 < p.n_win);
  t[1].win = (w0 + 1 < p.n_win) ? w0 + 1 : w0;

  float xr[N], xi[N], mean[2];
  bool skip[2];
  correlate_job<T, N, WANT_NZ>(p, t, buf, lg, partner_byte, xr, xi, skip, mean);

  const float nanv = __builtin_nanf("");
  {
    float row_max, u, v;
    const uint32_t g = t[0].pair * p.n_win + t[0].win;
    const float vmax = plane_max<N>(xr, row_max);
    find_peak<N>(buf, lg, xr, vmax, row_max, p, u, v, p.rescue_hdr && t[0].valid && !skip[0], g);
    float cm = vmax, sn = vmax * __builtin_amdgcn_rcpf(mean[0]);
    if (skip[0]) u = v = cm = sn = nanv;
    if (t[0].valid && lg == 0) {
      p.u[g] = u; p.v[g] = v; p.cmax[g] = cm; p.s2n[g] = sn;
    }
  }
  {
    float row_max, u, v;
    const uint32_t g = t[1].pair * p.n_win + t[1].win;
    const float vmax = plane_max<N>(xi, row_max);
    find_peak<N>(buf, lg, xi, vmax, row_max, p, u, v, p.rescue_hdr && t[1].valid && !skip[1], g);
    float cm = vmax, sn = vmax * __builtin_amdgcn_rcpf(mean[1]);
    if (skip[1]) u = v = cm = sn = nanv;
    if (t[1].valid && lg == 0) {
      p.u[g] = u; p.v[g] = v; p.cmax[g] = cm; p.s2n[g] = sn;
    }
  }
  if constexpr (PLANES) {
    if (t[0].valid) store_plane_rows<N>(p.planes + ((size_t)t[0].pair * p.n_win + t[0].win) * G::NN, lg, xr, skip[0]);
    if (t[1].valid) store_plane_rows<N>(p.planes + ((size_t)t[1].pair * p.n_win + t[1].win) * G::NN, lg, xi, skip[1]);
  }
}

// ---- time-walking kernel: a job owns ONE window and walks a run of consecutive frame pairs ---------------------------
// The normalised window of frame t is the "b" of pair (t-1, t) and the "a" of pair (t, t+1), and its normalisation
// (own mean / std / clip) is the same in both roles, so its spectrum F_t can be computed once.  An iteration takes TWO
// new frames of the window, packs them z = x_f + i x_{f+1} (both already unit-variance, so they are balanced), and
//   one forward 2-D FFT        -> Z;  P = Z[k] + conj Z[-k] = 2 F_f,  Q = (Z[k] - conj Z[-k]) / i = 2 F_{f+1}
//   R_a = conj(F_prev) P  (pair f-1),  R_b = conj(P) Q  (pair f),  both scaled by 1 / (4 N^4)
//   one inverse 2-D FFT of R_a + i R_b -> the two correlation planes;  F_prev <- Q
// i.e. 2 complex transforms and 2 window conversions per two pairs, where the per-pair kernel above needs 3 and 4.
// The only state carried between iterations is the Hermitian half of F_prev (N/2 + 1 complex values per lane), the
// same 34 registers (N = 32) the per-pair kernel spends on R1, so the budget of 128 VGPRs / 4 waves per SIMD holds.
// The time axis of a chunk is cut into segments of `seg_len` pairs (odd: seg_len + 1 frames = whole iterations) to
// have enough jobs; a segment's first iteration has no F_prev and yields one plane.  Which frame a window shares its
// transforms with depends on where the segment starts, so results are bit-reproducible for a given chunk but differ
// in the last float32 bit between different chunkings (the per-pair kernel does not; LSPIV_WALK=0 selects it).
template <typename T, int N, bool WANT_NZ>
__device__ __forceinline__ void prepare_one(const RowRaw<T, N>& raw, float (&x)[N], bool nz_pos, float std_gain, int& nonzero,
                                            bool& finite, bool& dead) {
  // every frame carries 1 / (2 N^2) on top of 1 / std, so each cross spectrum (a product of two frames' spectra)
  // comes out scaled by the 1 / (4 N^4) the planes need -- no multiply in the un-packing loop; a power of two for
  // N = 8 ... 64, i.e. the same bits as scaling the product
  const float kHalf = std_gain * (1.0f / (2.0f * (float)Geo<N>::NN));   // std_gain: 1, or sqrt((n - 1) / n) under the "std_ddof" option
  if constexpr (sizeof(T) == 1) {
    const RowStats st = stats_u8<N>(raw, WANT_NZ, nonzero);
    center_u8<N, true>(raw, st.mean, st.inv_std * kHalf, x);   // max((byte - mean) / std, 0) / (2 N^2) < 1; all zero for a constant window
    dead = st.inv_std == 0.0f;
  } else {
    // max(d, 0) * g == max(d * g, 0) exactly (g >= 0), and d * g < 1 (see center_u8): the clip rides on the multiply
    const float inv = load_center<N, true>(raw, x, WANT_NZ, nz_pos, nonzero, finite);
    const float g = inv * kHalf;
#pragma unroll
    for (int j = 0; j < N; ++j) x[j] = __builtin_amdgcn_fmed3f(x[j] * g, 0.0f, 1.0f);
    dead = inv == 0.0f;
  }
}

// Scheduling barriers between the phases of a walking iteration (conversion | FFT | transpose | FFT | un-pack | FFT |
// transpose | FFT | epilogue): they stop the scheduler from stretching live ranges across phases.  32 x 32: removed the
// last spills at 3 waves/SIMD (+7 %); 64 x 64 (256 VGPRs, 2 waves/SIMD, the carried spectrum is spilled across the
// inverse transform either way): 50 -> 44 spilled registers and +2.8 %, +3.7 % together with the barriers inside fft64
// (LSPIV_FFT64_SB; interleaved A/B on one box, 1080p 64 x 64 @ 75 %: 28.7 k -> 29.8 k pairs/s).  The other sizes were
// measured without and are left alone.
// VALU arbitration priority per phase of a walking iteration (s_setprio; the arbiter serves priority first, then age).  The
// register FFTs are long runs of independent VALU work; the epilogue (reductions, LDS round trips, logarithms), the loads +
// row conversion at the top of the next iteration, the transposes and the un-packing are short dependent chains that wait for
// LDS or memory between a few instructions.  With everything at priority 0 a wave in such a chain queues behind its
// neighbours' FFT streams for every one of those instructions; at priority 1 it gets them issued at once, is back in its
// next FFT sooner, and the SIMD has two FFT streams to pair more often.  Measured (interleaved builds, two boxes, 1000 pairs,
// rescue off): 32 x 32 6.34 -> 5.95-6.06 ms (-4.5 ... -6 %), 64 x 64 18.0 -> 16.9-17.1 ms (-5.9 %).  Which phases: epilogue +
// conversion alone bring the same on one box and 1.3 % less on the other; transposes alone or un-packing alone are SLOWER
// than no priorities (6.52 / 6.44 ms); conversion left at 0 costs 1.5 %; levels 2 / 3 change nothing (docs/history.md).
// T / U / E / P = priority of transposes / un-packing / epilogue / loads + conversion; the FFTs run at 0.
#ifndef LSPIV_PRIO_T
#define LSPIV_PRIO_T 1
#endif
#ifndef LSPIV_PRIO_U
#define LSPIV_PRIO_U 1
#endif
#ifndef LSPIV_PRIO_E
#define LSPIV_PRIO_E 1
#endif
#ifndef LSPIV_PRIO_P
#define LSPIV_PRIO_P LSPIV_PRIO_E   // loads + row conversion at the top of an iteration
#endif
#define LSPIV_PRIO_ANY (LSPIV_PRIO_T || LSPIV_PRIO_U || LSPIV_PRIO_E || LSPIV_PRIO_P)
#define LSPIV_SETPRIO(x) do { if constexpr (LSPIV_PRIO_ANY) __builtin_amdgcn_s_setprio(x); } while (0)
#ifndef LSPIV_WALK_RELAX
#define LSPIV_WALK_RELAX 0   // experiment: the per-timestep walking kernels without the barriers (bit mask: 1 = 32 x 32, 2 = 64 x 64)
#endif
template <typename T, int N> constexpr bool kWalkRelax = ((LSPIV_WALK_RELAX & 1) && N == 32) || ((LSPIV_WALK_RELAX & 2) && N == 64);
#ifndef LSPIV_WALK_SB
#define LSPIV_WALK_SB do { if constexpr ((N == 32 || N == 64) && !RELAX) __builtin_amdgcn_sched_barrier(0); } while (0)
#endif

// float32 rows: every lane reads its own 4 N-byte row 16 bytes at a time.  (Fetching the rows coalesced -- N / 4 consecutive
// lanes per row -- and staging them through the job's transpose tile was measured and dropped: 104.9 k against 107.5 k pairs/s
// at 1080p 32 x 32, the 16 extra LDS instructions per frame and 6 spilled registers cost more than the strided loads, which
// hit L2; DESIGN.md section 4, git history.)
#ifndef LSPIV_F32_EARLY
#define LSPIV_F32_EARLY 1
#endif
template <int N> constexpr bool kF32Early = LSPIV_F32_EARLY && Geo<N>::FULL;

// ---- 64 x 64 walking ENSEMBLE kernel: the job's partial sum through the idle transpose tile ----------------------------------
// One job per wave, 256 VGPRs (no room for an accumulator), no LDS to spare next to the tile: the partial sum lives in the job's
// 16 KB HBM slot.  Round 3 read-modified-wrote it in the plane's row-major layout -- every lane its own 256-byte row, 16 bytes at a
// time: 64 different lines per load and per store instruction, waves waiting for the loads 56 % of the time, 51.5 ms per 1000
// pairs against the per-timestep kernel's 28 (profiles/r04a_ens64).  Now:
//   * the slot is stored in the order the lanes hold it: element (row y = lane, column x = register j) at
//     slot[(j / 4) * 256 + lane * 4 + j % 4] -- a wave instruction moves 1 KB of consecutive bytes;
//   * after the last transpose of an iteration the tile is idle until the next iteration's first one: the slot is fetched
//     into it by 16 global_load_lds_dwordx4 (asynchronous, no registers) while the last FFT stage and the plane maxima run;
//   * then: wait for it, 16 ds_read_b128, acc = (acc + plane a) + plane b -- the very additions of the round-3 kernel, so the
//     sums keep their bits --, 16 global_store_dwordx4.  The first iteration of a job stores without reading (slots are
//     never zeroed).  ensemble_merge_kernel un-permutes when it adds the segments' slots to corr_sum.
// (No-return float atomics into lane-major slots were measured first: 44.6 ms -- the L2 atomic units take ~3.5 TB/s, 61 GB of
// adds per 1000 pairs do not hide behind the arithmetic; at 32 x 32 the same scheme LOSES to the register accumulator, 11.6
// against 7.0 ms.  docs/history.md, round 4.)
#ifndef LSPIV_XCD_ORDER_DEFAULT
#define LSPIV_XCD_ORDER_DEFAULT 1
#endif
#ifndef LSPIV_ENS_LDS_RMW
#define LSPIV_ENS_LDS_RMW 1
#endif
template <int N> constexpr bool kEnsLdsRmw = LSPIV_ENS_LDS_RMW && N == 64;
// ... and HALF of the partial sum stays in LDS for the whole segment: with the half-tile transposes above a wave's LDS is a 9.3 KB tile
// + 8 KB = the 17.4 KB it had, so columns 0 .. 31 of every lane's row (the first 8 KB of the lane-ordered slot) are accumulated in
// LDS and written to the slot ONCE per segment; columns 32 .. 63 keep the round trip through the tile (now 8 KB per iteration each way).
// Measured (round 5, sessions a / b, 1080p 64 x 64 @ 75 %, 1000 pairs, interleaved with the round-4 kernel on one box): 34.6 -> 31.6 ms
// (28.9 k -> 31.6 k pairs/s), kernel 30.75 ms in the trace; counters: fetched 6.2 GB (was 60: the 8 KB halves of the 256 live jobs of
// an XCD are 2 MB and now STAY in its 4 MB L2 -- TCC hit rate 94 %), written 27.8 GB (was 64; every store of the upper half still goes
// out), 67.6 M cycles per launch -- fewer than the per-timestep kernel's 68.3 M -- at 2 198 MHz (was 2 138).  Against the round-4
// kernel the results move by rounding only (tools/ens_hash.py + ens_diff.py, 120 pairs: 17 % of the samples of corr_sum differ, by at
// most 7e-7 of the plane maximum; masked corr_max / s2n by at most 2.4e-7 relative; u, v of 89 of 7 488 windows by at most 3.8e-6 px;
// the counts are identical): the additions are the same in the same order, but the compiler contracts the symmetric a b + c d
// products of the un-packing step the other way round in the re-shaped iteration, so the planes themselves move by an ulp.  Across
// chunkings and job orders the kernel agrees with itself bit for bit (tests/test_gpu_strip_order.py, the ensemble chunking tests).
#ifndef LSPIV_ENS_HALF_ACC
#define LSPIV_ENS_HALF_ACC 1
#endif
template <int N> constexpr bool kEnsHalfAcc = LSPIV_ENS_HALF_ACC && kEnsLdsRmw<N>;
// Round 6: why do ~27 GB of stores per 1000 pairs reach HBM when the hot halves of an XCD's ~256 live slots are
// 2 MB in a 4 MB L2?  What was established (tools/ubench/l2_rmw.hip + rocprofv3 WRITE_SIZE; A/B builds of this kernel):
//   * the L2 IS write-back: 2 MB per XCD re-written 500 times reach HBM once (0.017 GB for 8.4 GB stored), also with 55 us between
//     two passes (no ageing of dirty lines) and also when the read is a global_load_lds_dwordx4 like here;
//   * a stream of plain loads next to the slots evicts them (7.0 GB written back at 2 stream bytes per slot byte, 0.018 GB at 1/8),
//     non-temporal loads do not (0.35 GB) -- but in THIS kernel non-temporal frame loads (LSPIV_ENS64_NT) save 20 % of the written
//     bytes and quintuple the fetched ones (26.9 -> 21.4 GB, 3.3 -> 15.4 GB, 28.7 -> 31.5 ms): the frames are not what evicts;
//   * hot bytes that are the upper 8 KB of every 16 KB (address bit 13 set: half the sets) are written back 78 x as often in the
//     micro-benchmark as contiguous ones -- but the split layout below (all cold halves, then all hot halves) moves nothing here:
//     27.26 vs 27.48 GB written, 30.18 vs 30.19 ms, same bits (tools/sessions_ens64_ab.sh).
// So: neither a write-through L2, nor the frame stream, nor the slots' address pattern, nor the LDS-direct loads.  Every store of the
// hot half still reaches the fabric although its line is re-read from the L2 an iteration later (hit rate 95 %); the kernel does not
// wait for those bytes (VALU-bound).  Dead end recorded; both switches stay for whoever looks next.
#ifndef LSPIV_ENS_SPLIT_HALVES
#define LSPIV_ENS_SPLIT_HALVES 0
#endif
template <int N> constexpr bool kEnsSplitHalves = LSPIV_ENS_SPLIT_HALVES && kEnsHalfAcc<N>;
constexpr int kEnsHalfAccWaveDwords = kHalfTileDwords + 2048;   // tile | half accumulator (lane-ordered: (j / 4) * 256 + lane * 4 + j % 4, j < 32)
typedef float __attribute__((address_space(1))) * GlobalF32;
typedef float __attribute__((address_space(3))) * LdsF32;
__device__ __forceinline__ GlobalF32 uniform_global_ptr(const float* p) {   // visibly wave-uniform: scalar base + lane offset addressing
  const uint64_t a = reinterpret_cast<uint64_t>(p);
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)a), hi = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
  return (GlobalF32)(((uint64_t)hi << 32) | lo);
}
// Addressing of the slot: SGPR base + the lane's 32-bit byte offset + an immediate (written out: left to the compiler the 16
// store addresses and 4 prefetch bases become loop-invariant VGPR pairs that are spilled and re-loaded every iteration -- 30 of
// the kernel's 43 scratch loads per iteration, ~40 GB of the 109 GB it fetched per 1000 pairs, profiles/r04_ens64).
template <int N>
__device__ __forceinline__ void slot_prefetch_lds(GlobalF32 slot, float* buf) {   // slot: wave-uniform (uniform_global_ptr)
  static_assert(N == 64 && Geo<N>::GROUPS == 1 && Geo<N>::LDS_JOB >= N * N, "one job per wave, the tile holds a plane");
  const uint32_t voff = (threadIdx.x & 63u) * 16u;
  const uint32_t lds = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uintptr_t)(LdsF32)buf);
  __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0): the transpose's own reads of the tile have returned
  // M0 = LDS base of the piece (the hardware adds lane * 16), four immediates move the global AND the LDS address by 1 KB each.
  // (The compiler reserves M0 and rejects it in a clobber list: the block saves and restores it.)
#pragma unroll
  for (int qq = 0; qq < N / 16; ++qq) {
    const uint64_t base = reinterpret_cast<uint64_t>(slot) + (uint64_t)qq * 4096u;
    const uint32_t l = lds + qq * 4096u;
    uint32_t m0_saved;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, %2\n\tglobal_load_lds_dwordx4 %1, %2 offset:1024\n\t"
                 "global_load_lds_dwordx4 %1, %2 offset:2048\n\tglobal_load_lds_dwordx4 %1, %2 offset:3072\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(m0_saved) : "v"(voff), "s"(base), "s"(l) : "memory");
  }
}
// the half-accumulator variant: only the slot's second 8 KB (columns 32 .. 63) come in, to the start of the (half) tile
// (`slot`: the address of the slot's UPPER half -- slot + 8 KB in the interleaved layout, the slot's entry of the hot array in the split one)
__device__ __forceinline__ void slot_prefetch_lds_upper(GlobalF32 slot, float* buf) {
  const uint32_t voff = (threadIdx.x & 63u) * 16u;
  const uint32_t lds = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uintptr_t)(LdsF32)buf);
  __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0): the transpose's own reads of the tile have returned
#pragma unroll
  for (int qq = 0; qq < 2; ++qq) {
    const uint64_t base = reinterpret_cast<uint64_t>(slot) + (uint64_t)qq * 4096u;
    const uint32_t l = lds + qq * 4096u;
    uint32_t m0_saved;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, %2\n\tglobal_load_lds_dwordx4 %1, %2 offset:1024\n\t"
                 "global_load_lds_dwordx4 %1, %2 offset:2048\n\tglobal_load_lds_dwordx4 %1, %2 offset:3072\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(m0_saved) : "v"(voff), "s"(base), "s"(l) : "memory");
  }
}
__device__ __forceinline__ void slot_store4(uint64_t base, uint32_t voff, const f32x4 (&v)[4]) {   // 4 x 1 KB of the wave, 1 KB apart
  asm volatile("global_store_dwordx4 %0, %2, %1\n\tglobal_store_dwordx4 %0, %3, %1 offset:1024\n\t"
               "global_store_dwordx4 %0, %4, %1 offset:2048\n\tglobal_store_dwordx4 %0, %5, %1 offset:3072"
               :: "v"(voff), "s"(base), "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]) : "memory");
}
// acc <- (acc + m0 c0) + m1 c1, m in {0, 1} (exact products: the additions of the masked planes in pair order); `init`: the
// slot holds nothing yet (no prefetch was issued)
template <int N>
__device__ __forceinline__ void slot_accumulate(GlobalF32 slot, const float* buf, const float (&c0)[N], bool keep0, const float (&c1)[N],
                                                bool keep1, bool init_) {
  const int lane = threadIdx.x & 63;
  const bool init = __builtin_amdgcn_readfirstlane((int)init_) != 0;   // one job per wave: uniform, and the compiler may know it
  const bool any = __builtin_amdgcn_readfirstlane((int)(keep0 || keep1)) != 0;
  if (!init) __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): the slot has landed in the tile -- and nothing of it arrives later
  if (!init && !any) return;
  const float m0 = keep0 ? 1.0f : 0.0f, m1 = keep1 ? 1.0f : 0.0f;
  const float* lsrc = buf + lane * 4;
  const uint32_t voff = (uint32_t)lane * 16u;
#pragma unroll
  for (int qq = 0; qq < N / 16; ++qq) {
    f32x4 a[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int q = 4 * qq + k;
      if (init) a[k] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
      else a[k] = *reinterpret_cast<const f32x4*>(lsrc + q * 256);
#pragma unroll
      for (int e = 0; e < 4; ++e) a[k][e] = fmaf(c1[4 * q + e], m1, fmaf(c0[4 * q + e], m0, a[k][e]));
    }
    slot_store4(reinterpret_cast<uint64_t>(slot) + (uint64_t)qq * 4096u, voff, a);
    __builtin_amdgcn_sched_barrier(0);   // sixteen registers of sums at a time
  }
}

// the half-accumulator variant of slot_accumulate: columns 0 .. 31 read-modify-write `hacc` (LDS, stays), columns 32 .. 63 take the
// prefetched upper half of the slot from the tile and go back to HBM
// (`slot`: the slot's UPPER half, as for slot_prefetch_lds_upper)
__device__ __forceinline__ void slot_accumulate_half(GlobalF32 slot, const float* buf, float* hacc, const float (&c0)[64], bool keep0,
                                                     const float (&c1)[64], bool keep1, bool init_) {
  const int lane = threadIdx.x & 63;
  const bool init = __builtin_amdgcn_readfirstlane((int)init_) != 0;
  const bool any = __builtin_amdgcn_readfirstlane((int)(keep0 || keep1)) != 0;
  if (!init && !any) { __builtin_amdgcn_s_waitcnt(0x0F70); return; }   // nothing to add; the prefetch still has to land before the tile is reused
  const float m0 = keep0 ? 1.0f : 0.0f, m1 = keep1 ? 1.0f : 0.0f;
  float* lacc = hacc + lane * 4;
#pragma unroll
  for (int q = 0; q < 8; ++q) {           // columns 4 q .. 4 q + 3 < 32: LDS only
    f32x4 a = init ? f32x4{0.0f, 0.0f, 0.0f, 0.0f} : *reinterpret_cast<const f32x4*>(lacc + q * 256);
#pragma unroll
    for (int e = 0; e < 4; ++e) a[e] = fmaf(c1[4 * q + e], m1, fmaf(c0[4 * q + e], m0, a[e]));
    *reinterpret_cast<f32x4*>(lacc + q * 256) = a;
  }
  if (!init) __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): the upper half of the slot has landed in the tile
  const float* lsrc = buf + lane * 4;
  const uint32_t voff = (uint32_t)lane * 16u;
#pragma unroll
  for (int qq = 2; qq < 4; ++qq) {
    f32x4 a[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int q = 4 * qq + k;
      if (init) a[k] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
      else a[k] = *reinterpret_cast<const f32x4*>(lsrc + (q - 8) * 256);
#pragma unroll
      for (int e = 0; e < 4; ++e) a[k][e] = fmaf(c1[4 * q + e], m1, fmaf(c0[4 * q + e], m0, a[k][e]));
    }
    slot_store4(reinterpret_cast<uint64_t>(slot) + (uint64_t)(qq - 2) * 4096u, voff, a);
    __builtin_amdgcn_sched_barrier(0);
  }
}
// end of the segment: the LDS half goes to the first 8 KB of the slot (ensemble_merge_kernel reads the slot as before)
__device__ __forceinline__ void slot_flush_half(GlobalF32 slot, const float* hacc) {
  const int lane = threadIdx.x & 63;
  const float* lacc = hacc + lane * 4;
  const uint32_t voff = (uint32_t)lane * 16u;
#pragma unroll
  for (int qq = 0; qq < 2; ++qq) {
    f32x4 a[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) a[k] = *reinterpret_cast<const f32x4*>(lacc + (4 * qq + k) * 256);
    slot_store4(reinterpret_cast<uint64_t>(slot) + (uint64_t)qq * 4096u, voff, a);
  }
}

// Order of a segment's jobs over the window grid.  Row-major (strip_w = 0), or column strips of strip_w windows, each strip top
// to bottom: the jobs that are resident together then span MORE window rows and fewer columns.  64 x 64 @ 75 % overlap: a
// window row shares 48 of its 64 image rows with the next one, and 256 resident jobs per XCD are 2.2 rows of a 1080p grid -- the
// other three rows that need a band come rounds later, when the band has left the 4 MB L2 (5.7 GB fetched for a 2.1 GB stack,
// profiles/r03_c3); with strips of 32 windows a round is 8 rows deep.
__device__ __forceinline__ uint32_t strip_order(uint32_t idx, uint32_t strip_w, uint32_t n_rows, uint32_t n_cols) {
  if (strip_w == 0 || strip_w >= n_cols) return idx;
  const uint32_t n_strips = (n_cols + strip_w - 1) / strip_w, per = strip_w * n_rows;
  const uint32_t sidx = min(idx / per, n_strips - 1);
  const uint32_t rem = idx - sidx * per;
  const uint32_t w = sidx == n_strips - 1 ? n_cols - sidx * strip_w : strip_w;
  const uint32_t row = rem / w;
  return row * n_cols + sidx * strip_w + (rem - row * w);
}

// Which (segment, index in the segment's job order) a lane group of a walking kernel works on.  The hardware hands block b to XCD
// b % 8, and an XCD works through its blocks in order.
//   by_windows = 0 (rounds 2 - 4): every XCD gets ONE CONTIGUOUS RANGE of the jobs segment * n_win + window, i.e. whole segments, so a
//     frame is pulled into one L2 only -- but the partition is static: where segments differ in length (the shorter last one; anchors
//     that do not divide the chunk) the XCD that holds the short jobs runs dry while the others work (1080p 64 x 64, 300 pairs at an
//     anchor of 125: 26.2 k against 32.8 k pairs/s at 25).
//   by_windows = 1 (round 5): every XCD gets an eighth of the WINDOWS -- a contiguous range of the job order, i.e. whole rows / strips
//     -- of EVERY segment, segment after segment: the same work per XCD whatever the segments' lengths, a frame's band still goes to
//     one L2 (neighbouring bands share the windows' overlap).
// `local`: the lane group's index within its block (wave * GROUPS + group); JPB: jobs per block.  Blocks per launch: walk_blocks.
struct WalkJob { uint32_t seg, widx; bool valid; };
template <uint32_t JPB>
__device__ __forceinline__ WalkJob walk_job(uint32_t local, uint32_t n_seg, uint32_t n_win, const FastDiv& div_nwin, uint32_t by_windows) {
  WalkJob j;
  const uint32_t xcd = blockIdx.x & 7u, lb = blockIdx.x >> 3;
  if (by_windows) {
    const uint32_t wlo = (uint32_t)(((uint64_t)xcd * n_win) >> 3), nw = (uint32_t)(((uint64_t)(xcd + 1) * n_win) >> 3) - wlo;
    const uint32_t i = lb * JPB + local;                  // index among this XCD's jobs
    j.valid = i < n_seg * nw;                             // (an XCD without windows, n_win < 8: never)
    const uint32_t ii = j.valid ? i : 0u;
    j.seg = nw ? ii / nw : 0u;
    j.widx = wlo + (ii - j.seg * nw);
    if (!j.valid) { j.seg = n_seg - 1; j.widx = n_win - 1; }   // a job past the end recomputes the last one and stores nothing
  } else {
    const uint32_t nb = gridDim.x;
    const uint32_t q = nb >> 3, r = nb & 7u;
    const uint32_t blk = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + lb;
    uint32_t job = blk * JPB + local;
    j.valid = job < n_seg * n_win;
    job = j.valid ? job : n_seg * n_win - 1;
    j.seg = div_nwin.div(job);
    j.widx = job - j.seg * n_win;
  }
  return j;
}
static inline uint32_t walk_blocks(uint32_t n_seg, uint32_t n_win, uint32_t jobs_per_block, uint32_t by_windows) {
  if (by_windows) {   // every XCD: the blocks of its largest possible share, n_seg * ceil(n_win / 8) jobs
    const uint64_t per_xcd = (uint64_t)n_seg * ((n_win + 7) / 8);
    return (uint32_t)(8 * ((per_xcd + jobs_per_block - 1) / jobs_per_block));
  }
  return (uint32_t)(((uint64_t)n_seg * n_win + jobs_per_block - 1) / jobs_per_block);
}
// LSPIV_XCD_ORDER: 0 / 1 = by_windows above (A/B; read once per process)
static inline uint32_t walk_xcd_by_windows() {
  static const int env = getenv("LSPIV_XCD_ORDER") ? atoi(getenv("LSPIV_XCD_ORDER")) : LSPIV_XCD_ORDER_DEFAULT;
  return env != 0 ? 1u : 0u;
}

// what a walking job carries from one iteration to the next
template <int N>
struct WalkCarry {
  float fpr[N / 2 + 1], fpi[N / 2 + 1];   // 2 F_prev, ky = 0 .. N/2
  bool prev_dead, prev_finite;
  int prev_nz;
  __device__ __forceinline__ void reset() {
#pragma unroll
    for (int ky = 0; ky <= N / 2; ++ky) fpr[ky] = fpi[ky] = 0.0f;
    prev_dead = true; prev_finite = true; prev_nz = N * N;
  }
};

// One iteration: rows of frames f (and f + 1 when `has2`) of the job's window -> clipped planes xr (pair f-1) and xi
// (pair f), their means (DC bins) and NaN flags; the carry moves on to frame f + 1.
// RELAX: without the scheduling barriers between the phases (a caller with registers to spare: the 32 x 32 ensemble kernel runs two
// waves per SIMD for its register accumulator and may use 256 VGPRs)
template <typename T, int N, bool WANT_NZ, bool RELAX = false, bool HALF_TILE = false>
__device__ __forceinline__ void walk_iteration(const PivParams& p, const T* row, bool has2, float* buf, int lg,
                                               int partner_byte, int lane0_byte, WalkCarry<N>& c, float (&xr)[N],
                                               float (&xi)[N], float& mean_a, float& mean_b, bool& skip_a, bool& skip_b,
                                               bool& dead_a, bool& dead_b, GlobalF32 acc_slot = nullptr, bool one_job_per_wave_ens = false) {
  using G = Geo<N>;
  constexpr int H = N / 2;
  bool dead0, dead1, fin0 = true, fin1 = true;
  int nz0 = G::NN, nz1 = G::NN;
  if constexpr (LSPIV_PRIO_P != LSPIV_PRIO_E) LSPIV_SETPRIO(LSPIV_PRIO_P);
  if constexpr (sizeof(T) == 4 && kF32Early<N>) {
    // float32 rows: the loads of BOTH frames are in flight before the first is consumed (they land in xr / xi, the registers
    // they are converted in), so an iteration waits for memory once instead of twice
    const float kHalf = p.std_gain * (1.0f / (2.0f * (float)G::NN));
    const float* r0 = reinterpret_cast<const float*>(row);
    const float* r1 = has2 ? r0 + p.frame_elems : r0;
    load_row_f32<N>(r0, xr);
    load_row_f32<N>(r1, xi);
    const float inv0 = center_clip_f<N, true>(xr, WANT_NZ, p.nz_positive != 0, nz0, fin0);
    const float g0 = inv0 * kHalf;
#pragma unroll
    for (int j = 0; j < N; ++j) xr[j] = __builtin_amdgcn_fmed3f(xr[j] * g0, 0.0f, 1.0f);
    dead0 = inv0 == 0.0f;
    LSPIV_WALK_SB;
    const float inv1 = center_clip_f<N, true>(xi, WANT_NZ, p.nz_positive != 0, nz1, fin1);
    const float g1 = inv1 * kHalf;
#pragma unroll
    for (int j = 0; j < N; ++j) xi[j] = __builtin_amdgcn_fmed3f(xi[j] * g1, 0.0f, 1.0f);
    dead1 = inv1 == 0.0f;
  } else {
    RowRaw<T, N> raw0, raw1;
    if constexpr (sizeof(T) == 1 && HALF_TILE && LSPIV_ENS64_NT) {   // the 64 x 64 ensemble kernel: frames past the L2-resident slots
      raw0.template fetch<true>(row);
      raw1.template fetch<true>(has2 ? row + p.frame_elems : row);
    } else {
      raw0.fetch(row);
      raw1.fetch(has2 ? row + p.frame_elems : row);
    }
    prepare_one<T, N, WANT_NZ>(raw0, xr, p.nz_positive != 0, p.std_gain, nz0, fin0, dead0);
    LSPIV_WALK_SB;
    prepare_one<T, N, WANT_NZ>(raw1, xi, p.nz_positive != 0, p.std_gain, nz1, fin1, dead1);
  }
  if constexpr (kEnsLdsRmw<N>) {
    if (one_job_per_wave_ens) {   // (compile-time false in the per-timestep kernel)
      // 64 x 64 ensemble kernel: the two "zero-variance window" flags as scalars now -- left as they are, the compiler keeps the
      // two 1 / std in VGPRs across all four transforms (spilled and re-loaded) to compare them with zero at the very end
      dead0 = __builtin_amdgcn_readfirstlane((int)dead0) != 0;
      dead1 = __builtin_amdgcn_readfirstlane((int)dead1) != 0;
    }
  }
  LSPIV_WALK_SB;
  LSPIV_SETPRIO(0);
  fft_n<false>(xr, xi);              // along x
  LSPIV_WALK_SB;
  LSPIV_SETPRIO(LSPIV_PRIO_T);
  transpose2<N, HALF_TILE>(buf, lg, xr, xi);    // lane = kx, regs = y
  LSPIV_WALK_SB;
  LSPIV_SETPRIO(0);
  fft_n<false>(xr, xi);              // along y -> Z[ky][kx]
  LSPIV_WALK_SB;
  LSPIV_SETPRIO(LSPIV_PRIO_U);
  // un-pack the two spectra, form both cross spectra and pack them for the shared inverse, one ky at a time (a
  // step only touches registers ky and N - ky of this lane and of the mirrored lane, so it can run in place)
#pragma unroll
  for (int ky = 0; ky <= H; ++ky) {
    const int kn = (N - ky) % N;
    const float mr = bperm_f(partner_byte, xr[kn]);
    const float mi = bperm_f(partner_byte, xi[kn]);
    const float pr = xr[ky] + mr, pi = xi[ky] - mi;     // 2 F_f      (each with its frame's 1 / (2 N^2))
    const float ar = c.fpr[ky] * pr + c.fpi[ky] * pi, ai = c.fpr[ky] * pi - c.fpi[ky] * pr;   // conj(F_prev) P
    // the new carry is formed AFTER the last use of the old one, straight into its place: no copies on the loop back-edge
    c.fpr[ky] = xi[ky] + mi;                            // 2 F_{f+1}
    c.fpi[ky] = mr - xr[ky];
    const float qr = c.fpr[ky], qi = c.fpi[ky];
    const float br = pr * qr + pi * qi, bi = pr * qi - pi * qr;                               // conj(P) Q
    xr[ky] = ar - bi;                // (R_a + i R_b)[ky][kx]
    xi[ky] = ai + br;
    if (ky >= 1 && ky < H) {         // rows above N/2: conj of (R_a - i R_b) at the mirrored lane
      xr[kn] = bperm_f(partner_byte, ar + bi);
      xi[kn] = -bperm_f(partner_byte, ai - br);
    }
  }
  mean_a = bperm_f(lane0_byte, xr[0]);   // plane means = DC bins
  mean_b = bperm_f(lane0_byte, xi[0]);
  __builtin_amdgcn_sched_barrier(0);
  LSPIV_SETPRIO(0);
  fft_n<true>(xr, xi);                 // along ky
  LSPIV_WALK_SB;
  LSPIV_SETPRIO(LSPIV_PRIO_T);
  transpose2<N, HALF_TILE>(buf, lg, xr, xi);      // lane = y, regs = kx
  if constexpr (kEnsLdsRmw<N>) {
    // 64 x 64 ensemble kernel: the job's tile is free from here to the next iteration's first transpose -- the running partial
    // sum of the job starts its way from HBM into it now (asynchronously, no registers) and is there when the planes are final
    if (acc_slot) { if constexpr (HALF_TILE) slot_prefetch_lds_upper(acc_slot, buf); else slot_prefetch_lds<N>(acc_slot, buf); }
  }
  LSPIV_WALK_SB;
  LSPIV_SETPRIO(0);
  dead_a = c.prev_dead || dead0;
  dead_b = dead0 || dead1;
  if constexpr (kClampInFft<N>) {
    // clip [0, 1] rides on the last butterfly stage (clamp modifier).  A pair with a zero-variance window has an
    // exactly-zero cross spectrum, but its plane shares the inverse transform with the other pair's and would come out
    // as rounding noise of that one: the callers override such a pair's results (corr 0, s2n / u / v NaN, zero plane)
    ifft_clamped(xr, xi);              // along kx -> c_a + i c_b
    LSPIV_WALK_SB;
  } else {
    fft_n<true>(xr, xi);
    LSPIV_WALK_SB;
    const float hi_a = dead_a ? 0.0f : 1.0f, hi_b = dead_b ? 0.0f : 1.0f;
#pragma unroll
    for (int j = 0; j < N; ++j) {
      xr[j] = __builtin_amdgcn_fmed3f(xr[j], 0.0f, hi_a);
      xi[j] = __builtin_amdgcn_fmed3f(xi[j], 0.0f, hi_b);
    }
  }
  LSPIV_SETPRIO(LSPIV_PRIO_E);         // epilogue, and the next iteration's loads + row conversion
  skip_a = !(c.prev_finite && fin0);
  skip_b = !(fin0 && fin1);
  if (WANT_NZ) {
    skip_a = skip_a || below_threshold<N>(c.prev_nz, nz0, p.signal_threshold);
    skip_b = skip_b || below_threshold<N>(nz0, nz1, p.signal_threshold);
  }
  c.prev_dead = dead1; c.prev_finite = fin1; c.prev_nz = nz1;
}

template <typename T, int N, bool PLANES, bool WANT_NZ>
__global__ __launch_bounds__(BLOCK, (kWalkWaves<T, N>)) void piv_fft_walk_kernel(PivParams p) {
  const uint32_t seg_len = p.seg_len, n_seg = p.n_seg;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  using G = Geo<N>;
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int grp = lane / G::LG;
  const int lg = lane & (G::LG - 1);
  float* buf = smem + (wave * G::GROUPS + grp) * G::LDS_JOB;
  float* mini = smem + WAVES_PER_BLOCK * G::GROUPS * G::LDS_JOB + (wave * G::GROUPS + grp) * 3 * G::LDS_ROW;   // plane b's three rows (64 x 64)
  const int partner_byte = partner_byte_of<N>(lane, lg);
  const int lane0_byte = lane0_byte_of<N>();

  // XCD-aware job order (walk_job): which segment, which window
  const WalkJob wj = walk_job<WAVES_PER_BLOCK * G::GROUPS>((uint32_t)(wave * G::GROUPS + grp), n_seg, p.n_win, p.div_nwin, p.xcd_by_windows);
  const bool job_valid = wj.valid;
  const uint32_t seg = wj.seg;
  const uint32_t win = strip_order(wj.widx, p.strip_w, (uint32_t)p.n_rows, (uint32_t)p.n_cols);
  // segments are anchored at absolute pair indices (common.h, kWalkAnchor): segment 0 ends at the first anchor
  const uint32_t p0 = seg == 0 ? 0u : p.seg_first + (seg - 1) * seg_len;
  const uint32_t p1 = min(seg == 0 ? p.seg_first : p0 + seg_len, p.n_pairs);   // pairs [p0, p1) = frames p0 .. p1
  const uint32_t wrow = p.div_ncols.div(win);
  const uint32_t wcol = win - wrow * (uint32_t)p.n_cols;
  const T* row = static_cast<const T*>(p.frames) + ((int64_t)p0 * p.H + (int64_t)(wrow * p.sy + row_of<N>(lg))) * p.W +
                 (int64_t)wcol * p.sx;
  const float nanv = __builtin_nanf("");

  const bool win_dropped = WANT_NZ && p.win_keep && !p.win_keep[win];   // "stack" mode: one score per window position (A7)
  WalkCarry<N> carry;
  carry.reset();
  for (uint32_t f = p0; f <= p1; f += 2, row += 2 * p.frame_elems) {
    const bool has2 = f + 1 <= p1;
    float xr[N], xi[N], mean_a, mean_b;
    bool skip_a, skip_b, dead_a, dead_b;
    walk_iteration<T, N, WANT_NZ, kWalkRelax<T, N>>(p, row, has2, buf, lg, partner_byte, lane0_byte, carry, xr, xi, mean_a, mean_b, skip_a, skip_b,
                                                    dead_a, dead_b);
    if (WANT_NZ && win_dropped) skip_a = skip_b = true;
    const bool valid_a = job_valid && f > p0, valid_b = job_valid && has2;
    if constexpr (kTwoPlaneEpilogue<N>) {
      // both planes side by side (PeakFit above): maxima and peak rows from registers, plane a parked whole in the tile, plane b's
      // three rows in the job's slice of the extra LDS, then two independent straight-line fits
      float row_max_a, row_max_b;
      const float vmax_a = plane_max<N>(xr, row_max_a);
      const float vmax_b = plane_max<N>(xi, row_max_b);
      const int ip_a = peak_row<N>(lg, vmax_a, row_max_a);
      const int ip_b = peak_row<N>(lg, vmax_b, row_max_b);
      if (lane_active<N>(lg)) lds_row_write<N>(buf + lg * G::LDS_ROW, xr);
      park_three_rows<N>(mini, lg, xi, ip_b);
      __builtin_amdgcn_wave_barrier();
      const PeakFit<N> fa = peak_fit<N, false>(buf, lg, vmax_a, row_max_a, ip_a, p);
      const PeakFit<N> fb = peak_fit<N, true>(mini, lg, vmax_b, row_max_b, ip_b, p);
      __builtin_amdgcn_wave_barrier();
      const uint32_t g_a = (f - 1) * p.n_win + win, g_b = f * p.n_win + win;
      if (p.rescue_hdr) {   // uniform
        const bool note_a = valid_a && !dead_a && !skip_a, note_b = valid_b && !dead_b && !skip_b;
        if (note_a) peak_note<N>(buf, lg, fa, p, g_a);
        if (__builtin_amdgcn_ballot_w64(note_b && (fb.pc.amb || fb.pc.fit)) != 0) {   // rare: plane b's cold path wants the whole plane in LDS
          __builtin_amdgcn_wave_barrier();
          if (lane_active<N>(lg)) lds_row_write<N>(buf + lg * G::LDS_ROW, xi);
          __builtin_amdgcn_wave_barrier();
          if (note_b) peak_note<N>(buf, lg, fb, p, g_b);
        }
      }
      {
        float u = fa.u, v = fa.v, cm = vmax_a, sn = vmax_a * __builtin_amdgcn_rcpf(mean_a);
        if (dead_a) { u = v = sn = nanv; cm = 0.0f; }   // zero-variance window: an exactly-zero plane (corr 0, s2n 0/0, peak on the border)
        if (skip_a) u = v = cm = sn = nanv;
        if (valid_a && lg == 0) { p.u[g_a] = u; p.v[g_a] = v; p.cmax[g_a] = cm; p.s2n[g_a] = sn; }
      }
      {
        float u = fb.u, v = fb.v, cm = vmax_b, sn = vmax_b * __builtin_amdgcn_rcpf(mean_b);
        if (dead_b) { u = v = sn = nanv; cm = 0.0f; }
        if (skip_b) u = v = cm = sn = nanv;
        if (valid_b && lg == 0) { p.u[g_b] = u; p.v[g_b] = v; p.cmax[g_b] = cm; p.s2n[g_b] = sn; }
      }
      __builtin_amdgcn_wave_barrier();   // the parked samples are read before the next iteration's transposes reuse the tile
    } else {
      // plane b's maximum and peak row first: register-only work the scheduler can slot into the LDS waits of plane a's fit
      float row_max_b;
      const float vmax_b = plane_max<N>(xi, row_max_b);
      const int ip_b = peak_row<N>(lg, vmax_b, row_max_b);
      {
        float row_max, u, v;
        const uint32_t g = (f - 1) * p.n_win + win;
        const float vmax = plane_max<N>(xr, row_max);
        find_peak<N>(buf, lg, xr, vmax, row_max, p, u, v, p.rescue_hdr && valid_a && !dead_a && !skip_a, g);
        float cm = vmax, sn = vmax * __builtin_amdgcn_rcpf(mean_a);
        if (dead_a) { u = v = sn = nanv; cm = 0.0f; }   // zero-variance window: an exactly-zero plane (corr 0, s2n 0/0, peak on the border)
        if (skip_a) u = v = cm = sn = nanv;
        if (valid_a && lg == 0) {
          p.u[g] = u; p.v[g] = v; p.cmax[g] = cm; p.s2n[g] = sn;
        }
      }
      {
        float u, v;
        const uint32_t g = f * p.n_win + win;
        const float vmax = vmax_b, row_max = row_max_b;
        find_peak<N>(buf, lg, xi, vmax, row_max, p, u, v, p.rescue_hdr && valid_b && !dead_b && !skip_b, g, ip_b);
        float cm = vmax, sn = vmax * __builtin_amdgcn_rcpf(mean_b);
        if (dead_b) { u = v = sn = nanv; cm = 0.0f; }
        if (skip_b) u = v = cm = sn = nanv;
        if (valid_b && lg == 0) {
          p.u[g] = u; p.v[g] = v; p.cmax[g] = cm; p.s2n[g] = sn;
        }
      }
    }
    if constexpr (PLANES) {
      if (valid_a) store_plane_rows<N>(p.planes + ((size_t)(f - 1) * p.n_win + win) * G::NN, lg, xr, skip_a, dead_a);
      if (valid_b) store_plane_rows<N>(p.planes + ((size_t)f * p.n_win + win) * G::NN, lg, xi, skip_b, dead_b);
    }
  }
}

// ---- embedded mode epilogue: corr_max, mean, np.argmax and the sub-pixel fit over the n x n corner ----------------
// The plane is parked in LDS un-shifted (row ky at buf[ky * LDS_ROW + kx]); the reference's plane is its fftshift,
// shifted index = (k + n/2) mod n.  Same first-maximum rule and arithmetic as find_peak, with run-time n.
template <int N>
__device__ __forceinline__ void find_peak_embed(float* buf, int lg, const float (&c)[N], int n, const PivParams& p, float& vmax,
                                                float& mean, float& u, float& v, PeakCond& pc, int& ip, int& jp) {
  const int border_mode = p.border_mode;
  constexpr int LR = Geo<N>::LDS_ROW;
  constexpr int NONE = 1 << 12;
  f32x4* wrow = reinterpret_cast<f32x4*>(buf + lg * LR);
#pragma unroll
  for (int q = 0; q < N / 4; ++q) {
    const f32x4 w = {c[4 * q], c[4 * q + 1], c[4 * q + 2], c[4 * q + 3]};
    wrow[q] = w;
  }
  __builtin_amdgcn_wave_barrier();
  const bool row_in = lg < n;
  float rmax = -1.0f, rsum = 0.0f;
#pragma unroll
  for (int j = 0; j < N; ++j)
    if (j < n) { rmax = fmaxf(rmax, c[j]); rsum += c[j]; }
  rmax = row_in ? rmax : -1.0f;
  rsum = row_in ? rsum : 0.0f;
  vmax = group_max<N>(rmax);
  mean = group_sum<N>(rsum) / (float)(n * n);
  const int C = n / 2, M = n - 1;
  const int sh = lg + C >= n ? lg + C - n : lg + C;                       // this lane's shifted row AND column
  ip = group_min_i<N>((row_in && rmax == vmax) ? sh : NONE);              // first shifted row with the maximum
  const int y = ip - C < 0 ? ip - C + n : ip - C;
  const float rowv = buf[y * LR + lg];                                    // the peak row, one sample per lane
  jp = group_min_i<N>((row_in && rowv == vmax) ? sh : NONE);              // first shifted column in that row
  const float thr = vmax * (1.0f - p.rescue_tau);
  const bool near_tie = group_any<N>(row_in && ((rmax >= thr && sh != ip) || (rowv >= thr && sh != jp)));   // another sample within tau of the maximum
  const bool border = (ip == 0 || ip == M || jp == 0 || jp == M);
  const int x = jp - C < 0 ? jp - C + n : jp - C;
  const int ym = y == 0 ? M : y - 1, yp = y == M ? 0 : y + 1;
  const int xm = x == 0 ? M : x - 1, xp = x == M ? 0 : x + 1;
  const float c0 = vmax + kEpsPeak;
  const float cl = buf[ym * LR + x] + kEpsPeak;
  const float cr = buf[yp * LR + x] + kEpsPeak;
  const float cd = buf[y * LR + xm] + kEpsPeak;
  const float cu = buf[y * LR + xp] + kEpsPeak;
  const float l0 = __builtin_amdgcn_logf(c0);
  float den_v, den_u;
  v = (float)ip + gauss_offset_fast(__builtin_amdgcn_logf(cl), l0, __builtin_amdgcn_logf(cr), den_v) - (float)C;
  u = (float)jp + gauss_offset_fast(__builtin_amdgcn_logf(cd), l0, __builtin_amdgcn_logf(cu), den_u) - (float)C;
  pc = peak_cond(vmax, near_tie, border, cl, cr, den_v, v, cd, cu, den_u, u, p.rescue_k);
  if (border) border_result(border_mode, jp - C, ip - C, u, v);
}

// fft-shifted n x n plane out of the parked LDS copy (cross_corr's volume); call before the buffer is reused
template <int N>
__device__ __forceinline__ void store_plane_embed(float* dst, const float* buf, int lg, int n, bool nan_plane) {
  constexpr int LR = Geo<N>::LDS_ROW;
  if (lg < n) {
    const int C = n / 2;
    const int sh = lg + C >= n ? lg + C - n : lg + C;
    for (int jp = 0; jp < n; ++jp) {
      const int x = jp - C < 0 ? jp - C + n : jp - C;
      dst[sh * n + jp] = nan_plane ? __builtin_nanf("") : buf[lg * LR + x];
    }
  }
}

template <typename T, int N, bool PLANES, bool WANT_NZ>
__global__ __launch_bounds__(BLOCK, 2) void piv_fft_embed_kernel(PivParams p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  using G = Geo<N>;
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int grp = lane / G::LG;
  const int lg = lane & (G::LG - 1);
  float* buf = smem + (wave * G::GROUPS + grp) * G::LDS_JOB;
  const int partner_byte = partner_byte_of<N>(lane, lg);
  const uint32_t nb = gridDim.x;                                   // XCD-aware block order, as piv_fft_kernel
  const uint32_t q = nb >> 3, r = nb & 7u;
  const uint32_t xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3;
  const uint32_t blk = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
  constexpr bool SINGLE = N == 64;                                 // one window per job (see correlate_job)
  const uint32_t jobs_per_pair = SINGLE ? p.n_win : (p.n_win + 1) >> 1;
  uint32_t job = (blk * WAVES_PER_BLOCK + wave) * G::GROUPS + grp;
  const bool job_valid = job < p.n_pairs * jobs_per_pair;
  job = job_valid ? job : p.n_pairs * jobs_per_pair - 1;
  const uint32_t pair = SINGLE ? p.div_nwin.div(job) : p.div_jobs.div(job);
  const uint32_t w0 = (job - pair * jobs_per_pair) * (SINGLE ? 1 : 2);
  TileRef t[2];
  t[0].pair = t[1].pair = pair;
  t[0].win = w0;
  t[0].valid = job_valid;
  t[1].valid = !SINGLE && job_valid && (w0 + 1 < p.n_win);
  t[1].win = (!SINGLE && w0 + 1 < p.n_win) ? w0 + 1 : w0;

  float xr[N], xi[N], dc[2];
  bool skip[2];
  correlate_job<T, N, WANT_NZ, true>(p, t, buf, lg, partner_byte, xr, xi, skip, dc);
  const int n = p.wy;
  const float nanv = __builtin_nanf("");
#pragma unroll
  for (int k = 0; k < (SINGLE ? 1 : 2); ++k) {
    float vmax, mean, u, v;
    PeakCond pc;
    int ip, jp;
    find_peak_embed<N>(buf, lg, k == 0 ? xr : xi, n, p, vmax, mean, u, v, pc, ip, jp);
    float cm = vmax, sn = vmax * __builtin_amdgcn_rcpf(mean);
    if (skip[k]) u = v = cm = sn = nanv;
    if (t[k].valid && lg == 0) {
      const uint32_t g = t[k].pair * p.n_win + t[k].win;
      p.u[g] = u; p.v[g] = v; p.cmax[g] = cm; p.s2n[g] = sn;
      if (p.rescue_hdr && !skip[k] && (pc.amb || pc.fit))
        rescue_note(p.rescue_hdr, p.rescue_fit, p.rescue_cap_fit, p.rescue_amb, p.rescue_cap_amb, g, pc, ip, jp);
    }
    if constexpr (PLANES) {
      if (t[k].valid) store_plane_embed<N>(p.planes + ((size_t)t[k].pair * p.n_win + t[k].win) * n * n, buf, lg, n, skip[k]);
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// Ensemble correlation for the embedded sizes: a job owns one window and walks the chunk's pairs in order (two per
// iteration in the 32-point variant, sharing the inverse transform like piv_fft_ensemble_kernel; one in the 64-point
// variant), adding every kept plane to its n x n slice of corr_sum (fft-shifted layout) through the parked LDS copy:
// row by row, lane = column, so the read-modify-write is coalesced.
template <typename T, int N, bool WANT_NZ>
__global__ __launch_bounds__(BLOCK, 2) void piv_fft_embed_ensemble_kernel(PivParams p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  using G = Geo<N>;
  constexpr int LR = G::LDS_ROW;
  constexpr bool SINGLE = N == 64;
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int grp = lane / G::LG;
  const int lg = lane & (G::LG - 1);
  float* buf = smem + (wave * G::GROUPS + grp) * G::LDS_JOB;
  const int partner_byte = partner_byte_of<N>(lane, lg);
  const uint32_t job = (blockIdx.x * WAVES_PER_BLOCK + wave) * G::GROUPS + grp;
  const bool valid = job < p.n_win;
  const uint32_t w = valid ? job : p.n_win - 1;
  const int n = p.wy, C = n / 2;
  const bool row_in = lg < n;
  const int xs = lg - C < 0 ? lg - C + n : lg - C;       // un-shifted column of this lane's shifted column
  float* sum = p.corr_sum + (size_t)w * n * n;
  float cnt = 0.0f;
  for (uint32_t pair = 0; pair < p.n_pairs; pair += SINGLE ? 1 : 2) {
    const bool two = !SINGLE && pair + 1 < p.n_pairs;
    TileRef t[2] = {{pair, w, valid}, {two ? pair + 1 : pair, w, valid && two}};
    float xr[N], xi[N], dc[2];
    bool skip[2];
    correlate_job<T, N, WANT_NZ, true>(p, t, buf, lg, partner_byte, xr, xi, skip, dc);
#pragma unroll
    for (int k = 0; k < (SINGLE ? 1 : 2); ++k) {
      const float (&c)[N] = k == 0 ? xr : xi;
      float rmax = -1.0f, rsum = 0.0f;
#pragma unroll
      for (int j = 0; j < N; ++j)
        if (j < n) { rmax = fmaxf(rmax, c[j]); rsum += c[j]; }
      const float vmax = group_max<N>(row_in ? rmax : -1.0f);
      const float mean = group_sum<N>(row_in ? rsum : 0.0f) / (float)(n * n);
      float cm = vmax, sn = vmax * __builtin_amdgcn_rcpf(mean);
      const bool keep = t[k].valid && !skip[k] && (cm >= p.corr_min) && (sn >= p.s2n_min);  // NaN s2n compares false
      cm = keep ? cm : 0.0f;
      sn = keep ? sn : 0.0f;
      cnt += (cm > 1e-6f) ? 1.0f : 0.0f;
      if (t[k].valid && lg == 0) {
        p.cmax[(size_t)t[k].pair * p.n_win + w] = cm;
        p.s2n[(size_t)t[k].pair * p.n_win + w] = sn;
      }
      if (keep) {
        f32x4* wrow = reinterpret_cast<f32x4*>(buf + lg * LR);
#pragma unroll
        for (int q = 0; q < N / 4; ++q) {
          const f32x4 v = {c[4 * q], c[4 * q + 1], c[4 * q + 2], c[4 * q + 3]};
          wrow[q] = v;
        }
        __builtin_amdgcn_wave_barrier();
        if (row_in) {
          for (int ip = 0; ip < n; ++ip) {
            const int y = ip - C < 0 ? ip - C + n : ip - C;
            sum[ip * n + lg] += buf[y * LR + xs];
          }
        }
        __builtin_amdgcn_wave_barrier();
      }
    }
  }
  if (valid && lg == 0) p.corr_count[w] += cnt;
}

template <typename T, int N, bool WANT_NZ>
static hipError_t launch_embed_t(const PivParams& p, bool ensemble, hipStream_t s) {
  using G = Geo<N>;
  constexpr uint32_t jobs_per_block = WAVES_PER_BLOCK * G::GROUPS;
  if (ensemble) {
    hipLaunchKernelGGL((piv_fft_embed_ensemble_kernel<T, N, WANT_NZ>), dim3((p.n_win + jobs_per_block - 1) / jobs_per_block),
                       dim3(BLOCK), G::LDS_BYTES, s, p);
    return hipGetLastError();
  }
  const uint32_t jobs = p.n_pairs * (N == 64 ? p.n_win : (p.n_win + 1) / 2);
  const uint32_t blocks = (jobs + jobs_per_block - 1) / jobs_per_block;
  if (p.planes)
    hipLaunchKernelGGL((piv_fft_embed_kernel<T, N, true, WANT_NZ>), dim3(blocks), dim3(BLOCK), G::LDS_BYTES, s, p);
  else
    hipLaunchKernelGGL((piv_fft_embed_kernel<T, N, false, WANT_NZ>), dim3(blocks), dim3(BLOCK), G::LDS_BYTES, s, p);
  return hipGetLastError();
}

template <int N>
static hipError_t launch_embed(const PivParams& p, int dtype, bool ensemble, hipStream_t s) {
  const bool nz = p.signal_threshold >= 0.0f;
  switch (dtype) {
    case 0: return nz ? launch_embed_t<uint8_t, N, true>(p, ensemble, s) : launch_embed_t<uint8_t, N, false>(p, ensemble, s);
    case 1: return nz ? launch_embed_t<float, N, true>(p, ensemble, s) : launch_embed_t<float, N, false>(p, ensemble, s);
    case 2: return nz ? launch_embed_t<double, N, true>(p, ensemble, s) : launch_embed_t<double, N, false>(p, ensemble, s);
    default: return hipErrorInvalidValue;
  }
}

// ---- ensemble kernel: a job owns ONE window and walks the pairs of the chunk in order, two at a time ---------
// (pyorc/velocimetry/ffpiv.py:222-241,361-363): planes failing corr_min / s2n_min / finite are
// zeroed, corr_sum += plane, corr_count += (corr_max > 1e-6).  The two planes that share an inverse FFT are the
// SAME window of two consecutive frame pairs (2k, 2k+1), not two windows of one pair: that gives n_win jobs instead of
// n_win / 2 (a 1080p grid then fills all 8192 half-wave slots of the chip instead of 48 % of them) and one
// read-modify-write of the running sum per two pairs.  The accumulation order is still the pair order, one owner per
// window => bit-reproducible, no atomics.  The running sums live in HBM (MALL-resident read-modify-write by their
// single owner) so the kernel keeps the register budget of the per-timestep kernel.
template <int N>
__device__ __forceinline__ void accumulate_planes(float* dst, int lg, const float (&c0)[N], bool keep0,
                                                  const float (&c1)[N], bool keep1, bool init = false) {
  // corr_sum is kept in fft-shifted layout (what u_v_displacement expects); `init`: the slot holds nothing yet, start from 0
  if (!lane_active<N>(lg)) return;
  float* row = dst + wrap_n<N>(lg + N / 2) * N;
  if constexpr (N % 4 == 0) {
#pragma unroll
    for (int qd = 0; qd < N / 4; ++qd) {
      f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
      if (!init) acc = *reinterpret_cast<f32x4*>(row + 4 * qd);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int j = (4 * qd + e + N / 2) % N;
        acc[e] += keep0 ? c0[j] : 0.0f;   // pair 2k first, then 2k+1: the reference's summation order
        acc[e] += keep1 ? c1[j] : 0.0f;
      }
      *reinterpret_cast<f32x4*>(row + 4 * qd) = acc;
    }
  } else {
#pragma unroll
    for (int qd = 0; qd < N / 2; ++qd) {
      f32x2 acc = {0.0f, 0.0f};
      if (!init) acc = *reinterpret_cast<f32x2*>(row + 2 * qd);
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int j = (2 * qd + e + N / 2) % N;
        acc[e] += keep0 ? c0[j] : 0.0f;
        acc[e] += keep1 ? c1[j] : 0.0f;
      }
      *reinterpret_cast<f32x2*>(row + 2 * qd) = acc;
    }
  }
}

template <typename T, int N, bool WANT_NZ>
__global__ __launch_bounds__(BLOCK, (kWavesPerSimd<T, N>)) void piv_fft_ensemble_kernel(PivParams p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  using G = Geo<N>;
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int grp = lane / G::LG;
  const int lg = lane & (G::LG - 1);
  float* buf = smem + (wave * G::GROUPS + grp) * G::LDS_JOB;
  const int partner_byte = partner_byte_of<N>(lane, lg);
  const uint32_t job = (blockIdx.x * WAVES_PER_BLOCK + wave) * G::GROUPS + grp;
  const bool valid = job < p.n_win;
  const uint32_t w = valid ? job : p.n_win - 1;
  float cnt = 0.0f;
  for (uint32_t pair = 0; pair < p.n_pairs; pair += 2) {
    const bool two = pair + 1 < p.n_pairs;   // an odd chunk ends with a lone pair (its partner recomputes it, unused)
    TileRef t[2] = {{pair, w, valid}, {two ? pair + 1 : pair, w, valid && two}};
    float xr[N], xi[N], mean[2];
    bool skip[2], keep[2];
    correlate_job<T, N, WANT_NZ>(p, t, buf, lg, partner_byte, xr, xi, skip, mean);
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      float row_max;
      const float vmax = plane_max<N>(k == 0 ? xr : xi, row_max);
      float cm = vmax, sn = vmax * __builtin_amdgcn_rcpf(mean[k]);
      keep[k] = t[k].valid && !skip[k] && (cm >= p.corr_min) && (sn >= p.s2n_min);  // NaN s2n compares false
      cm = keep[k] ? cm : 0.0f;
      sn = keep[k] ? sn : 0.0f;
      cnt += (cm > 1e-6f) ? 1.0f : 0.0f;
      if (t[k].valid && lg == 0) {
        p.cmax[(size_t)t[k].pair * p.n_win + w] = cm;
        p.s2n[(size_t)t[k].pair * p.n_win + w] = sn;
      }
    }
    if (keep[0] || keep[1]) accumulate_planes<N>(p.corr_sum + (size_t)w * G::NN, lg, xr, keep[0], xi, keep[1]);
  }
  if (valid && lg == 0) p.corr_count[w] += cnt;
}

// Walking ENSEMBLE kernel: the same iteration, but the planes are masked (corr_min, s2n_min, finite) and added to the
// job's partial sum instead of being searched for a peak.  A job = (time segment, window); each segment has its own
// partial sum in HBM (zeroed by the caller), merged afterwards in segment order (ensemble_merge_kernel): fixed
// summation order, no atomics, and ~3 rounds of jobs on the chip where one job per window would leave it 2/3 idle.
// LSPIV_ENS_REGACC: the job's partial sum lives in N VGPRs per lane for the whole segment (lane = row y, register =
// column x, the layout the planes come out of the inverse transform in) and is stored once at the end, instead of a
// read-modify-write of its HBM slot every iteration.  Same sequence of float additions per element => same bits.
#ifndef LSPIV_ENS_REGACC
#define LSPIV_ENS_REGACC 1
#endif
template <int N> constexpr bool kEnsRegAcc = LSPIV_ENS_REGACC && N <= 32;
#ifndef LSPIV_WALK_ENS_WAVES_32
#define LSPIV_WALK_ENS_WAVES_32 2
#endif
#ifndef LSPIV_ENS32_RELAX
#define LSPIV_ENS32_RELAX 1   // measured: 6.78 -> 6.02 ms per 1000 pairs (147 k -> 166 k pairs/s): without the barriers the allocator needs 168 VGPRs instead of 194 and a third wave fits
#endif
template <typename T, int N> constexpr bool kEnsRelax = LSPIV_ENS32_RELAX && N == 32 && kEnsRegAcc<N> && sizeof(T) < 8;
template <typename T, int N>
constexpr int kWalkEnsWaves = (kEnsRegAcc<N> && N == 32 && sizeof(T) < 8) ? LSPIV_WALK_ENS_WAVES_32 : kWalkWaves<T, N>;
// Relaxed (no barriers), the uint8 kernel without a signal threshold allocates 168 VGPRs by itself: a third wave fits although the
// bound says two (166 k pairs/s; bounded at three it spills 28 bytes: 163 k).  The variants that count non-zero samples and the
// float32 ones come out at 173 - 174 and would run two waves: they are bounded at three (168 VGPRs + 28 ... 44 bytes of scratch).
template <typename T, int N, bool WANT_NZ>
constexpr int kWalkEnsBound = (kEnsRelax<T, N> && (WANT_NZ || sizeof(T) == 4)) ? 3 : kWalkEnsWaves<T, N>;

template <typename T, int N, bool WANT_NZ>
__global__ __launch_bounds__(BLOCK, (kWalkEnsBound<T, N, WANT_NZ>)) void piv_fft_walk_ensemble_kernel(PivParams p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  using G = Geo<N>;
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int grp = lane / G::LG;
  const int lg = lane & (G::LG - 1);
  float* buf = smem + (wave * G::GROUPS + grp) * (kEnsHalfAcc<N> ? kEnsHalfAccWaveDwords : G::LDS_JOB);
  float* hacc = buf + kHalfTileDwords;                             // (kEnsHalfAcc: the job's columns 0 .. 31 of the partial sum)
  const int partner_byte = partner_byte_of<N>(lane, lg);
  const int lane0_byte = lane0_byte_of<N>();
  // XCD-aware job order, as piv_fft_walk_kernel (walk_job).
  // one job per wave (N > 32): say so -- segment, window, pair range, loop counter, result and slot addresses then live in SGPRs
  // instead of one VGPR (pair) each (in-loop scratch 15 + 3 -> 11 + 3; the per-timestep kernel got WORSE with the same line,
  // 8 + 0 -> 12 + 2, and does without)
  uint32_t local = (uint32_t)(wave * G::GROUPS + grp);
  if constexpr (G::GROUPS == 1) local = (uint32_t)__builtin_amdgcn_readfirstlane((int)local);
  const WalkJob wj = walk_job<WAVES_PER_BLOCK * G::GROUPS>(local, p.n_seg, p.n_win, p.div_nwin, p.xcd_by_windows);
  const bool job_valid = wj.valid;
  const uint32_t seg = wj.seg;
  const uint32_t win = strip_order(wj.widx, p.strip_w, (uint32_t)p.n_rows, (uint32_t)p.n_cols);
  const uint32_t p0 = seg == 0 ? 0u : p.seg_first + (seg - 1) * p.seg_len;
  const uint32_t p1 = min(seg == 0 ? p.seg_first : p0 + p.seg_len, p.n_pairs);
  const uint32_t wrow = p.div_ncols.div(win);
  const uint32_t wcol = win - wrow * (uint32_t)p.n_cols;
  const T* row = static_cast<const T*>(p.frames) + ((int64_t)p0 * p.H + (int64_t)(wrow * p.sy + row_of<N>(lg))) * p.W +
                 (int64_t)wcol * p.sx;
  // the job's partial-sum slot is that of its (segment, WINDOW): ensemble_merge_kernel adds the segments' slots of a window in
  // segment order, whatever order the jobs ran in (strip_order permutes the windows of a segment)
  const uint32_t pslot = seg * p.n_win + win;
  float* part = p.part_sum + (size_t)pslot * (kEnsSplitHalves<N> ? G::NN / 2 : G::NN);      // split: the slot's entry of the cold array
  const GlobalF32 part_u = kEnsLdsRmw<N> ? uniform_global_ptr(part) : nullptr;   // one job per wave: the slot pointer lives in SGPRs
  // the half that is re-written every iteration: the slot's entry of the hot array (split layout), or its second 8 KB
  const GlobalF32 part_hi = !kEnsHalfAcc<N> ? part_u
                            : uniform_global_ptr(kEnsSplitHalves<N> ? p.part_sum + ((size_t)p.n_seg * p.n_win + pslot) * (G::NN / 2) : part + G::NN / 2);
  const bool win_dropped = WANT_NZ && p.win_keep && !p.win_keep[win];
  float cnt = 0.0f;
  float acc[kEnsRegAcc<N> ? N : 1];
  if constexpr (kEnsRegAcc<N>) {
#pragma unroll
    for (int j = 0; j < N; ++j) acc[j] = 0.0f;
  }
  WalkCarry<N> carry;
  carry.reset();
  bool first = job_valid;   // the slot is not zeroed beforehand: the job's first iteration stores, the later ones add
  for (uint32_t f = p0; f <= p1; f += 2, row += 2 * p.frame_elems) {
    const bool has2 = f + 1 <= p1;
    float xr[N], xi[N], mean[2];
    bool skip[2], keep[2], dead[2];
    walk_iteration<T, N, WANT_NZ, kEnsRelax<T, N>, kEnsHalfAcc<N>>(p, row, has2, buf, lg, partner_byte, lane0_byte, carry, xr, xi, mean[0], mean[1], skip[0], skip[1],
                                                   dead[0], dead[1], (kEnsLdsRmw<N> && !first) ? part_hi : nullptr, kEnsLdsRmw<N>);
    if (WANT_NZ && win_dropped) skip[0] = skip[1] = true;
    const bool valid[2] = {job_valid && f > p0, job_valid && has2};
    float vmaxs[2];
    if constexpr (kEnsLdsRmw<N>) {
      // 256 VGPRs, both planes and the carry live: four running maxima per plane instead of the 22-wide first level of plane_max's tree
      float ma[4] = {xr[0], xr[1], xr[2], xr[3]}, mb[4] = {xi[0], xi[1], xi[2], xi[3]};
#pragma unroll
      for (int j = 4; j < N; j += 4) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { ma[e] = fmaxf(ma[e], xr[j + e]); mb[e] = fmaxf(mb[e], xi[j + e]); }
      }
      vmaxs[0] = group_max_nonneg<N>(fmaxf(fmaxf(ma[0], ma[1]), fmaxf(ma[2], ma[3])));
      vmaxs[1] = group_max_nonneg<N>(fmaxf(fmaxf(mb[0], mb[1]), fmaxf(mb[2], mb[3])));
    } else {
      float row_max;
      vmaxs[0] = plane_max<N>(xr, row_max);
      vmaxs[1] = plane_max<N>(xi, row_max);
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const float vmax = vmaxs[k];
      float cm = dead[k] ? 0.0f : vmax, sn = dead[k] ? __builtin_nanf("") : vmax * __builtin_amdgcn_rcpf(mean[k]);   // dead: zero plane
      keep[k] = valid[k] && !skip[k] && (cm >= p.corr_min) && (sn >= p.s2n_min);  // NaN s2n compares false
      cm = keep[k] ? cm : 0.0f;
      sn = keep[k] ? sn : 0.0f;
      cnt += (cm > 1e-6f) ? 1.0f : 0.0f;
      if (valid[k] && lg == 0) {
        const size_t g = (size_t)(f - 1 + k) * p.n_win + win;
        p.cmax[g] = cm;
        p.s2n[g] = sn;
      }
    }
    if constexpr (kEnsRegAcc<N>) {
#pragma unroll
      for (int j = 0; j < N; ++j) {
        acc[j] += keep[0] ? xr[j] : 0.0f;   // pair f-1 first, then f: the reference's summation order
        acc[j] += keep[1] ? xi[j] : 0.0f;
      }
    } else if constexpr (kEnsLdsRmw<N>) {
      if constexpr (kEnsHalfAcc<N>) {
        if (job_valid) slot_accumulate_half(part_hi, buf, hacc, xr, keep[0], xi, keep[1], first);
        else __builtin_amdgcn_s_waitcnt(0x0F70);
      } else if (job_valid) slot_accumulate<N>(part_u, buf, xr, keep[0], xi, keep[1], first);
      else __builtin_amdgcn_s_waitcnt(0x0F70);   // (a job past the end stores nothing, but its prefetch still has to land before the tile is reused)
      first = false;
    } else {
      if (first || keep[0] || keep[1]) accumulate_planes<N>(part, lg, xr, keep[0], xi, keep[1], first);
      first = false;
    }
  }
  if constexpr (kEnsRegAcc<N>) {
    if (job_valid) store_plane_rows<N>(part, lg, acc, false);   // fft-shifted layout, like accumulate_planes
  }
  if constexpr (kEnsHalfAcc<N>) {
    if (job_valid) slot_flush_half(part_u, hacc);
  }
  if (job_valid && lg == 0) p.part_cnt[pslot] = cnt;
}

// strip width of the walking kernels' job order (strip_order): 32 windows for 64 x 64 (HBM fetch of C3 5.2 -> 2.6 GB), 24 for
// 32 x 32 windows of float32 / float64 frames (13.3 -> 9.5 GB = 1.15 x the stack; uint8 frames fetch MORE in strips, 2.2 -> 3.1 GB,
// and stay row-major like every other size); times do not move either way.  LSPIV_STRIP_W overrides it for every kernel
// (0 = row-major; measurements)
template <typename T, int N>
static uint32_t walk_strip_width() {
  static const int env = getenv("LSPIV_STRIP_W") ? atoi(getenv("LSPIV_STRIP_W")) : -1;
  return env >= 0 ? (uint32_t)env : (N == 64 ? 32u : (N == 32 && sizeof(T) >= 4) ? 24u : 0u);
}

template <typename T, int N, bool WANT_NZ>
static hipError_t launch_t(const PivParams& p, bool ensemble, hipStream_t s) {
  using G = Geo<N>;
  constexpr uint32_t jobs_per_block = WAVES_PER_BLOCK * G::GROUPS;
  if (ensemble && p.part_sum) {   // walking ensemble kernel + ordered merge of the per-segment partial sums
    PivParams q = p;
    q.strip_w = walk_strip_width<T, N>();
    q.xcd_by_windows = walk_xcd_by_windows();
    constexpr size_t ens_lds = kEnsHalfAcc<N> ? (size_t)WAVES_PER_BLOCK * kEnsHalfAccWaveDwords * 4 : (size_t)G::LDS_BYTES;
    hipLaunchKernelGGL((piv_fft_walk_ensemble_kernel<T, N, WANT_NZ>), dim3(walk_blocks(p.n_seg, p.n_win, jobs_per_block, q.xcd_by_windows)),
                       dim3(BLOCK), ens_lds, s, q);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    return launch_ensemble_merge(p.part_sum, p.part_cnt, p.n_seg, p.n_win, G::NN, p.corr_sum, p.corr_count, s, kEnsLdsRmw<N> ? N : 0, kEnsSplitHalves<N>);
  }
  if (ensemble) {
    const uint32_t jobs = p.n_win;
    const uint32_t blocks = (jobs + jobs_per_block - 1) / jobs_per_block;
    hipLaunchKernelGGL((piv_fft_ensemble_kernel<T, N, WANT_NZ>), dim3(blocks), dim3(BLOCK), G::LDS_BYTES, s, p);
    return hipGetLastError();
  }
  // LSPIV_WALK: 0 = per-pair kernel; unset / 1 = time-walking kernel, segments anchored every walk_anchor(N, n_win) = 25 / 75 pairs of the
  // absolute pair index (results independent of the chunking); n > 1 = anchor length n (odd values waste no half iteration)
  const int walk = walk_setting();   // option or environment, read per launch
  if (walk != 0) {
    PivParams q = p;
    const WalkSegments w = walk_segments(p.n_pairs, p.pair_offset, walk > 1 ? (uint32_t)walk : walk_anchor(N, p.n_win));
    q.seg_len = w.seg_len; q.seg_first = w.seg_first; q.n_seg = w.n_seg;
    q.strip_w = walk_strip_width<T, N>();
    q.xcd_by_windows = walk_xcd_by_windows();
    const uint32_t wblocks = walk_blocks(w.n_seg, p.n_win, jobs_per_block, q.xcd_by_windows);
    constexpr size_t walk_lds = G::LDS_BYTES + (kTwoPlaneEpilogue<N> ? (size_t)WAVES_PER_BLOCK * G::GROUPS * 3 * G::LDS_ROW * 4 : 0);   // + plane b's three rows
    if (p.planes)
      hipLaunchKernelGGL((piv_fft_walk_kernel<T, N, true, WANT_NZ>), dim3(wblocks), dim3(BLOCK), walk_lds, s, q);
    else
      hipLaunchKernelGGL((piv_fft_walk_kernel<T, N, false, WANT_NZ>), dim3(wblocks), dim3(BLOCK), walk_lds, s, q);
    return hipGetLastError();
  }
  const uint32_t jobs = p.n_pairs * ((p.n_win + 1) / 2);
  const uint32_t blocks = (jobs + jobs_per_block - 1) / jobs_per_block;
  // LSPIV_DEBUG_EXTRA_LDS: occupancy experiments only (pads the LDS request so fewer blocks fit a CU)
  static const int extra_lds = getenv("LSPIV_DEBUG_EXTRA_LDS") ? atoi(getenv("LSPIV_DEBUG_EXTRA_LDS")) : 0;
  if (p.planes)
    hipLaunchKernelGGL((piv_fft_kernel<T, N, true, WANT_NZ>), dim3(blocks), dim3(BLOCK), G::LDS_BYTES + extra_lds, s, p);
  else
    hipLaunchKernelGGL((piv_fft_kernel<T, N, false, WANT_NZ>), dim3(blocks), dim3(BLOCK), G::LDS_BYTES + extra_lds, s, p);
  return hipGetLastError();
}

template <int N>
static hipError_t launch_fft(const PivParams& p, int dtype, bool ensemble, hipStream_t s) {
  const bool nz = p.signal_threshold >= 0.0f;
  switch (dtype) {
    case 0: return nz ? launch_t<uint8_t, N, true>(p, ensemble, s) : launch_t<uint8_t, N, false>(p, ensemble, s);
    case 1: return nz ? launch_t<float, N, true>(p, ensemble, s) : launch_t<float, N, false>(p, ensemble, s);
    case 2: return nz ? launch_t<double, N, true>(p, ensemble, s) : launch_t<double, N, false>(p, ensemble, s);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace lspiv
