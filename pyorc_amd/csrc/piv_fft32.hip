// Fused 32x32 interrogation-window kernel for gfx950 (CDNA4, wave64).
//
// Replaces, in ONE launch per frame chunk, what the reference does in three passes over a
// (T-1, n_win, 32, 32) float volume (pyorc/velocimetry/ffpiv.py:446-474):
//   ffpiv.cross_corr       window gather, per-window normalise, rfft2 . conj-mul . irfft2,
//                          fftshift, /N, clip[0,1]                       
//   numpy reductions       corr_max = nanmax(plane), s2n = corr_max / nanmean(plane)     (K6)
//   ffpiv.u_v_displacement argmax + 3-point log-Gaussian sub-pixel fit                   (K7)
// The correlation planes never leave the CU unless the caller asks for them.
//
// Mapping (there is no reference kernel; this is an MI355X design):
//   * a "job" is TWO windows processed by the 32 lanes of a half-wave, lane = tile row (then
//     tile column after a transpose), all 32 complex values of that row live in VGPRs;
//   * window pair (a from frame t, b from frame t+1) is packed z = a + i b, so the two real
//     forward FFTs cost one complex 2-D FFT;  R = conj(A) B follows from Z[k], Z[-k]:
//         4 R[k] = 2 Im(Z[k] Z[-k]) - i (|Z[k]|^2 - |Z[-k]|^2)
//   * the two windows of a job share ONE inverse transform: IFFT(R1 + i R2) = c1 + i c2;
//   * length-32 transforms are straight-line register code (fft_regs.h); the 2-D transposes go
//     through a padded per-half LDS tile (ds_write_b128 rows, ds_read_b64 columns, conflict
//     free); Z[-k] is fetched from the mirrored lane with ds_bpermute;
//   * per-window mean / variance / max / argmax / sum use DPP row reductions inside the half.
// MFMA is deliberately unused: this is FFT + pointwise work (BASELINE.json north_star).
#include "common.h"
#include "fft_regs.h"

namespace lspiv {

constexpr int TILE = 32;
constexpr int LDS_ROW = 68;                   // dwords per padded row: 32 complex + 4 pad
constexpr int LDS_JOB = TILE * LDS_ROW;       // dwords per half-wave buffer (8704 B)
constexpr int WAVES_PER_BLOCK = 4;
constexpr int BLOCK = 64 * WAVES_PER_BLOCK;
constexpr int LDS_BYTES = WAVES_PER_BLOCK * 2 * LDS_JOB * 4;  // 69632 B -> 2 blocks per CU

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef u32x4 u32x4_u __attribute__((aligned(1)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef f32x4 f32x4_u __attribute__((aligned(4)));
typedef double f64x2 __attribute__((ext_vector_type(2)));
typedef f64x2 f64x2_u __attribute__((aligned(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// ---- one tile row (32 samples) from global memory into registers ------------------------------
__device__ __forceinline__ void load_row(const uint8_t* p, float (&x)[32]) {
  u32x4 lo = *reinterpret_cast<const u32x4_u*>(p);
  u32x4 hi = *reinterpret_cast<const u32x4_u*>(p + 16);
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    uint32_t w = lo[k];
    x[4 * k + 0] = (float)(w & 0xffu);
    x[4 * k + 1] = (float)((w >> 8) & 0xffu);
    x[4 * k + 2] = (float)((w >> 16) & 0xffu);
    x[4 * k + 3] = (float)(w >> 24);
    w = hi[k];
    x[16 + 4 * k + 0] = (float)(w & 0xffu);
    x[16 + 4 * k + 1] = (float)((w >> 8) & 0xffu);
    x[16 + 4 * k + 2] = (float)((w >> 16) & 0xffu);
    x[16 + 4 * k + 3] = (float)(w >> 24);
  }
}
__device__ __forceinline__ void load_row(const float* p, float (&x)[32]) {
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    f32x4 v = *reinterpret_cast<const f32x4_u*>(p + 4 * k);
    x[4 * k + 0] = v[0]; x[4 * k + 1] = v[1]; x[4 * k + 2] = v[2]; x[4 * k + 3] = v[3];
  }
}
__device__ __forceinline__ void load_row(const double* p, float (&x)[32]) {
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    f64x2 v = *reinterpret_cast<const f64x2_u*>(p + 2 * k);
    x[2 * k + 0] = (float)v[0]; x[2 * k + 1] = (float)v[1];
  }
}

// pairwise (tree) sum of a register row: exact for constant rows, short dependency chains
__device__ __forceinline__ float tree_sum32(const float (&x)[32]) {
  float s[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) s[k] = x[2 * k] + x[2 * k + 1];
#pragma unroll
  for (int k = 0; k < 8; ++k) s[k] = s[2 * k] + s[2 * k + 1];
#pragma unroll
  for (int k = 0; k < 4; ++k) s[k] = s[2 * k] + s[2 * k + 1];
  return (s[0] + s[1]) + (s[2] + s[3]);
}

// mean-offset, variance, clip at zero (ffpiv normalize_intensity, A3) -- the 1/std factor is
// bilinear in the correlation and is applied once at the end.  Returns 1/std (0 if std == 0).
__device__ __forceinline__ float center_clip(float (&x)[32], bool& finite) {
  float s = half_sum(tree_sum32(x));
  float mean = s * (1.0f / 1024.0f);
  float q[32];
#pragma unroll
  for (int k = 0; k < 32; ++k) {
    float d = x[k] - mean;
    q[k] = d * d;
    x[k] = fmaxf(d, 0.0f);
  }
  float ssq = half_sum(tree_sum32(q));
  finite = finite && (fabsf(s) <= 3.0e38f) && (ssq <= 3.0e38f);
  float var = ssq * (1.0f / 1024.0f);
  return var > 0.0f ? 1.0f / sqrtf(var) : 0.0f;
}

__device__ __forceinline__ int count_nonzero32(const float (&x)[32]) {
  int c = 0;
#pragma unroll
  for (int k = 0; k < 32; ++k) c += (x[k] != 0.0f) ? 1 : 0;
  return half_sum_i(c);
}

// LDS transpose of the half-wave's 32x32 complex tile: lane r scatters its row r down column r
// of the buffer (ds_write_b64, lanes contiguous -> conflict free), then reads buffer row r, which
// is column r of the tile, with ds_read_b128 (row stride 68 dwords -> the 16 lanes of a b128
// group hit 16 distinct 4-bank slots).  Wide reads move 2x the bytes per LDS cycle of b64 pairs
// that the compiler would otherwise fuse into half-rate ds_read2_b64.
__device__ __forceinline__ void transpose32(float* buf, int l32, float (&xr)[32], float (&xi)[32]) {
  f32x2* wcol = reinterpret_cast<f32x2*>(buf + 2 * l32);
#pragma unroll
  for (int j = 0; j < 32; ++j) {
    f32x2 v = {xr[j], xi[j]};
    wcol[j * (LDS_ROW / 2)] = v;
  }
  __builtin_amdgcn_wave_barrier();  // same wave: LDS ops execute in order, this only pins the compiler
  const f32x4* rrow = reinterpret_cast<const f32x4*>(buf + l32 * LDS_ROW);
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    f32x4 v = rrow[q];
    xr[2 * q] = v[0]; xi[2 * q] = v[1];
    xr[2 * q + 1] = v[2]; xi[2 * q + 1] = v[3];
  }
  __builtin_amdgcn_wave_barrier();
}

// lane = kx, registers = ky hold Z = FFT2(a + i b).  In place: 4 conj(A) B.
__device__ __forceinline__ void cross_spectrum(int partner_byte, float (&zr)[32], float (&zi)[32]) {
  // self-paired rows ky = 0 and 16; pairs (ky, 32 - ky) for ky = 1..15
#pragma unroll
  for (int ky = 0; ky <= 16; ++ky) {
    const int kn = (32 - ky) & 31;
    // partner lane's Z[kn] is Z[-k] for this lane's Z[ky]; partner's Z[ky] serves Z[kn]
    float wr_n = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(partner_byte, __builtin_bit_cast(int, zr[kn])));
    float wi_n = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(partner_byte, __builtin_bit_cast(int, zi[kn])));
    float ar = zr[ky], ai = zi[ky];
    if (kn != ky) {
      float wr_k = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(partner_byte, __builtin_bit_cast(int, zr[ky])));
      float wi_k = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(partner_byte, __builtin_bit_cast(int, zi[ky])));
      float br = zr[kn], bi = zi[kn];
      zr[kn] = 2.0f * (br * wi_k + bi * wr_k);
      zi[kn] = (wr_k * wr_k + wi_k * wi_k) - (br * br + bi * bi);
    }
    zr[ky] = 2.0f * (ar * wi_n + ai * wr_n);
    zi[ky] = (wr_n * wr_n + wi_n * wi_n) - (ar * ar + ai * ai);
  }
}

struct TileRef {
  uint32_t pair;   // frame pair index inside the chunk
  uint32_t win;    // window index k * n_cols + m
  bool valid;
};

// Everything between "two window pairs" and "two clipped correlation planes in registers".
// On return xr = plane of tile 0, xi = plane of tile 1, natural (un-shifted) order: lane = row y,
// register = column x;  skip[k] = plane k is NaN (signal pre-mask / non-finite input).
template <typename T>
__device__ __forceinline__ void correlate_job(const PivParams& p, const TileRef (&t)[2], float* buf, int l32,
                                              int partner_byte, float (&xr)[32], float (&xi)[32], bool (&skip)[2]) {
  float Rr[32], Ri[32];
  float scale[2], hi[2];  // hi = clip ceiling: 1, or 0 for a zero-variance window (plane exactly 0)
  const T* frames = static_cast<const T*>(p.frames);
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const uint32_t wrow = t[k].win / (uint32_t)p.n_cols;
    const uint32_t wcol = t[k].win - wrow * (uint32_t)p.n_cols;
    const int64_t off = ((int64_t)t[k].pair * p.H + (int64_t)(wrow * p.sy + l32)) * p.W + (int64_t)wcol * p.sx;
    load_row(frames + off, xr);
    load_row(frames + off + p.frame_elems, xi);
    bool finite = true;
    skip[k] = false;
    if (p.signal_threshold >= 0.0f) {
      const float fa = (float)count_nonzero32(xr) * (1.0f / 1024.0f);
      const float fb = (float)count_nonzero32(xi) * (1.0f / 1024.0f);
      skip[k] = !(fa >= p.signal_threshold && fb >= p.signal_threshold);
    }
    const float inv_a = center_clip(xr, finite);
    const float inv_b = center_clip(xi, finite);
    skip[k] = skip[k] || !finite;
    // Balance the packed pair: b is rescaled to a's variance so |A| ~ |B| and the
    // |Z[k]|^2 - |Z[-k]|^2 difference in cross_spectrum() does not cancel catastrophically when
    // one window is much fainter than the other.  corr = inv_a inv_b corr(a'', b'') =
    // inv_a^2 corr(a'', rho b'') with rho = inv_b / inv_a.  A zero-variance window gives an
    // exactly-zero plane (scale 0), like the reference's zeros-if-std-is-0 rule (A3).
    const bool dead = (inv_a == 0.0f) || (inv_b == 0.0f);
    const float rho = dead ? 0.0f : inv_b / inv_a;
#pragma unroll
    for (int j = 0; j < 32; ++j) xi[j] *= rho;
    scale[k] = dead ? 0.0f : inv_a * inv_a * (1.0f / (4.0f * 1024.0f * 1024.0f));
    hi[k] = dead ? 0.0f : 1.0f;
    __builtin_amdgcn_sched_barrier(0);
    fft32<false>(xr, xi);            // along x
    __builtin_amdgcn_sched_barrier(0);
    transpose32(buf, l32, xr, xi);   // lane = kx, regs = y
    __builtin_amdgcn_sched_barrier(0);
    fft32<false>(xr, xi);            // along y -> Z[ky][kx]
    __builtin_amdgcn_sched_barrier(0);
    cross_spectrum(partner_byte, xr, xi);
    __builtin_amdgcn_sched_barrier(0);
    // The per-window scale goes onto R BEFORE the two windows are packed into one inverse
    // transform: both planes then peak at <= 1, so float32 rounding of the shared inverse is
    // relative to O(1) for each of them.  (Scaling after the inverse lets a bright window's
    // rounding noise, ~1e-7 of ITS magnitude, swamp a faint neighbour packed with it.)
    if (k == 0) {
#pragma unroll
      for (int j = 0; j < 32; ++j) { Rr[j] = xr[j] * scale[0]; Ri[j] = xi[j] * scale[0]; }
    } else {
#pragma unroll
      for (int j = 0; j < 32; ++j) {  // Q = s1 R1 + i s2 R2
        const float r2r = xr[j], r2i = xi[j];
        xr[j] = fmaf(-r2i, scale[1], Rr[j]);
        xi[j] = fmaf(r2r, scale[1], Ri[j]);
      }
    }
  }
  __builtin_amdgcn_sched_barrier(0);
  fft32<true>(xr, xi);               // along ky
  __builtin_amdgcn_sched_barrier(0);
  transpose32(buf, l32, xr, xi);     // lane = y, regs = kx
  __builtin_amdgcn_sched_barrier(0);
  fft32<true>(xr, xi);               // along kx -> c1 + i c2
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int j = 0; j < 32; ++j) {
    xr[j] = fminf(fmaxf(xr[j], 0.0f), hi[0]);
    xi[j] = fminf(fmaxf(xi[j], 0.0f), hi[1]);
  }
}

// max / first-argmax (in fft-shifted row-major order) / sum of one plane held as lane = y, reg = x
__device__ __forceinline__ void plane_stats(const float (&c)[32], int l32, float& vmax, int& imax, float& sum) {
  float best = c[16];
  int bj = 0;
#pragma unroll
  for (int jj = 1; jj < 32; ++jj) {  // shifted column jj <-> x = (jj + 16) & 31
    const float v = c[(jj + 16) & 31];
    const bool g = v > best;
    best = g ? v : best;
    bj = g ? jj : bj;
  }
  vmax = best;
  imax = (((l32 + 16) & 31) << 5) | bj;
  half_argmax(vmax, imax);
  sum = half_sum(tree_sum32(c));
}

// planes of both tiles are in LDS as float2 (plane0, plane1) at [y][x]; returns u, v in pixels
__device__ __forceinline__ void subpixel(const float* buf, int which, int imax, float& u, float& v) {
  const int ip = imax >> 5, jp = imax & 31;  // shifted coordinates
  if (ip == 0 || ip == 31 || jp == 0 || jp == 31) {
    u = v = __builtin_nanf("");
    return;
  }
  const int y = (ip + 16) & 31, x = (jp + 16) & 31;
  const int ym = (ip - 1 + 16) & 31, yp = (ip + 1 + 16) & 31;
  const int xm = (jp - 1 + 16) & 31, xp = (jp + 1 + 16) & 31;
  const float c = buf[y * LDS_ROW + 2 * x + which] + kEpsPeak;
  const float cl = buf[ym * LDS_ROW + 2 * x + which] + kEpsPeak;
  const float cr = buf[yp * LDS_ROW + 2 * x + which] + kEpsPeak;
  const float cd = buf[y * LDS_ROW + 2 * xm + which] + kEpsPeak;
  const float cu = buf[y * LDS_ROW + 2 * xp + which] + kEpsPeak;
  const float l0 = logf(c);
  v = (float)ip + gauss_offset(logf(cl), l0, logf(cr)) - 16.0f;
  u = (float)jp + gauss_offset(logf(cd), l0, logf(cu)) - 16.0f;
}

__device__ __forceinline__ void store_plane_rows(float* dst, int l32, const float (&c)[32], bool nan_plane) {
  // shifted row i' = (y + 16) & 31 receives columns x = 16..31, 0..15
  float* row = dst + ((l32 + 16) & 31) * 32;
  const float nanv = __builtin_nanf("");
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    f32x4 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = nan_plane ? nanv : c[(4 * q + e + 16) & 31];
    *reinterpret_cast<f32x4*>(row + 4 * q) = v;
  }
}

// ---- per-timestep kernel: one job (two consecutive windows) per half-wave ---------------------
template <typename T, bool PLANES>
__global__ __launch_bounds__(BLOCK, 2) void piv_fft32_kernel(PivParams p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int half = lane >> 5;
  const int l32 = lane & 31;
  float* buf = smem + (wave * 2 + half) * LDS_JOB;
  const int partner_byte = ((lane & 32) | ((32 - l32) & 31)) << 2;

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
  uint32_t job = (blk * WAVES_PER_BLOCK + wave) * 2 + half;
  const bool job_valid = job < p.n_pairs * jobs_per_pair;
  job = job_valid ? job : p.n_pairs * jobs_per_pair - 1;
  const uint32_t pair = job / jobs_per_pair;
  const uint32_t w0 = (job - pair * jobs_per_pair) * 2;
  TileRef t[2];
  t[0].pair = t[1].pair = pair;
  t[0].win = w0;
  t[0].valid = job_valid;
  t[1].valid = job_valid && (w0 + 1 < p.n_win);
  t[1].win = (w0 + 1 < p.n_win) ? w0 + 1 : w0;
  float xr[32], xi[32];
  bool skip[2];
  correlate_job<T>(p, t, buf, l32, partner_byte, xr, xi, skip);

  float vmax[2], sum[2];
  int imax[2];
  plane_stats(xr, l32, vmax[0], imax[0], sum[0]);
  plane_stats(xi, l32, vmax[1], imax[1], sum[1]);
  // park both planes in LDS for the 5-point neighbourhood reads
  {
    f32x4* wrow = reinterpret_cast<f32x4*>(buf + l32 * LDS_ROW);
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      f32x4 v = {xr[2 * j], xi[2 * j], xr[2 * j + 1], xi[2 * j + 1]};
      wrow[j] = v;
    }
    __builtin_amdgcn_wave_barrier();
  }
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    float u, v;
    subpixel(buf, k, imax[k], u, v);
    float cm = vmax[k];
    float sn = vmax[k] / (sum[k] * (1.0f / 1024.0f));
    if (skip[k]) u = v = cm = sn = __builtin_nanf("");
    if (t[k].valid && l32 == 0) {
      const uint32_t g = t[k].pair * p.n_win + t[k].win;
      p.u[g] = u;
      p.v[g] = v;
      p.cmax[g] = cm;
      p.s2n[g] = sn;
    }
  }
  if constexpr (PLANES) {
    if (t[0].valid) store_plane_rows(p.planes + ((size_t)t[0].pair * p.n_win + t[0].win) * 1024, l32, xr, skip[0]);
    if (t[1].valid) store_plane_rows(p.planes + ((size_t)t[1].pair * p.n_win + t[1].win) * 1024, l32, xi, skip[1]);
  }
  __builtin_amdgcn_wave_barrier();
}

// ---- ensemble kernel: a job owns two windows and walks all pairs of the chunk in order ---------
// (pyorc/velocimetry/ffpiv.py:222-241,361-363): planes failing corr_min / s2n_min / finite are
// zeroed, corr_sum += plane, corr_count += (corr_max > 1e-6).  The accumulation order is the
// pair order, one owner per window => bit-reproducible, no atomics.
template <typename T>
__global__ __launch_bounds__(BLOCK, 2) void piv_fft32_ensemble_kernel(PivParams p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int half = lane >> 5;
  const int l32 = lane & 31;
  float* buf = smem + (wave * 2 + half) * LDS_JOB;
  const int partner_byte = ((lane & 32) | ((32 - l32) & 31)) << 2;
  const uint32_t job = (blockIdx.x * WAVES_PER_BLOCK + wave) * 2 + half;
  uint32_t w[2];
  bool valid[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    w[k] = job * 2 + k;
    valid[k] = w[k] < p.n_win;
    w[k] = valid[k] ? w[k] : p.n_win - 1;
  }
  float acc0[32], acc1[32];
#pragma unroll
  for (int j = 0; j < 32; ++j) acc0[j] = acc1[j] = 0.0f;
  float cnt[2] = {0.0f, 0.0f};
  for (uint32_t pair = 0; pair < p.n_pairs; ++pair) {
    TileRef t[2] = {{pair, w[0], valid[0]}, {pair, w[1], valid[1]}};
    float xr[32], xi[32];
    bool skip[2];
    correlate_job<T>(p, t, buf, l32, partner_byte, xr, xi, skip);
    float vmax[2], sum[2];
    int imax[2];
    plane_stats(xr, l32, vmax[0], imax[0], sum[0]);
    plane_stats(xi, l32, vmax[1], imax[1], sum[1]);
    bool keep[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      float cm = vmax[k];
      float sn = vmax[k] / (sum[k] * (1.0f / 1024.0f));
      keep[k] = !skip[k] && (cm >= p.corr_min) && (sn >= p.s2n_min);  // NaN s2n compares false
      cm = keep[k] ? cm : 0.0f;
      sn = keep[k] ? sn : 0.0f;
      cnt[k] += (cm > 1e-6f) ? 1.0f : 0.0f;
      if (valid[k] && l32 == 0) {
        p.cmax[(size_t)pair * p.n_win + w[k]] = cm;
        p.s2n[(size_t)pair * p.n_win + w[k]] = sn;
      }
    }
#pragma unroll
    for (int j = 0; j < 32; ++j) {
      acc0[j] += keep[0] ? xr[j] : 0.0f;
      acc1[j] += keep[1] ? xi[j] : 0.0f;
    }
  }
  // corr_sum is kept in fft-shifted layout (what u_v_displacement expects)
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    if (!valid[k]) continue;
    float* row = p.corr_sum + (size_t)w[k] * 1024 + ((l32 + 16) & 31) * 32;
#pragma unroll
    for (int qd = 0; qd < 8; ++qd) {
      f32x4 old = *reinterpret_cast<f32x4*>(row + 4 * qd);
#pragma unroll
      for (int e = 0; e < 4; ++e) old[e] += (k == 0 ? acc0 : acc1)[(4 * qd + e + 16) & 31];
      *reinterpret_cast<f32x4*>(row + 4 * qd) = old;
    }
    if (l32 == 0) p.corr_count[w[k]] += cnt[k];
  }
}

template <typename T>
static hipError_t launch_t(const PivParams& p, bool ensemble, hipStream_t s) {
  if (ensemble) {
    const uint32_t jobs = (p.n_win + 1) / 2;
    const uint32_t blocks = (jobs + 2 * WAVES_PER_BLOCK - 1) / (2 * WAVES_PER_BLOCK);
    hipLaunchKernelGGL(piv_fft32_ensemble_kernel<T>, dim3(blocks), dim3(BLOCK), LDS_BYTES, s, p);
    return hipGetLastError();
  }
  const uint32_t jobs = p.n_pairs * ((p.n_win + 1) / 2);
  const uint32_t blocks = (jobs + 2 * WAVES_PER_BLOCK - 1) / (2 * WAVES_PER_BLOCK);
  if (p.planes)
    hipLaunchKernelGGL((piv_fft32_kernel<T, true>), dim3(blocks), dim3(BLOCK), LDS_BYTES, s, p);
  else
    hipLaunchKernelGGL((piv_fft32_kernel<T, false>), dim3(blocks), dim3(BLOCK), LDS_BYTES, s, p);
  return hipGetLastError();
}

hipError_t launch_piv_fft32(const PivParams& p, int dtype, bool ensemble, hipStream_t s) {
  switch (dtype) {
    case 0: return launch_t<uint8_t>(p, ensemble, s);
    case 1: return launch_t<float>(p, ensemble, s);
    case 2: return launch_t<double>(p, ensemble, s);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace lspiv
