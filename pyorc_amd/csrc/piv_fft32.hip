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
//   * a "job" is TWO windows of one frame pair processed by the 32 lanes of a half-wave,
//     lane = tile row (tile column after a transpose), the 32 complex values of that row in VGPRs;
//   * window pair (a from frame t, b from frame t+1) is packed z = a + i b, so the two real
//     forward FFTs cost one complex 2-D FFT;  R = conj(A) B follows from Z[k], Z[-k]:
//         4 R[k] = 2 Im(Z[k] Z[-k]) - i (|Z[k]|^2 - |Z[-k]|^2)
//     R is Hermitian (the correlation is real), bit-exactly so in this formula, therefore a lane
//     computes and keeps only R[ky][kx] for ky = 0..16; the rest is the conjugate of what the
//     mirrored lane (-kx) holds;
//   * the two windows of a job share ONE inverse transform: IFFT(R1 + i R2) = c1 + i c2;
//   * length-32 transforms are straight-line register code (fft_regs.h); the 2-D transposes go
//     through a padded per-half LDS tile, real and imaginary plane one after the other so that
//     three 4-wave workgroups fit the 160 KB of a CU; Z[-k] comes from the mirrored lane with
//     ds_bpermute; per-window mean / variance / max / argmax / sum are DPP reductions.
//   * a wave can issue one VALU instruction every 4 cycles while a SIMD retires two, so the
//     kernel is built for THREE waves per SIMD (<= 168 VGPRs, 9 KB of LDS per wave).
// MFMA is deliberately unused: this is FFT + pointwise work (BASELINE.json north_star).
#include <cstdlib>

#include "common.h"
#include "fft_regs.h"

namespace lspiv {

constexpr int TILE = 32;
constexpr int LDS_ROW = 36;                   // dwords per padded row (16-byte aligned rows, 9 l mod 16 slots)
constexpr int LDS_JOB = TILE * LDS_ROW;       // dwords per half-wave buffer (4608 B)
constexpr int WAVES_PER_BLOCK = 4;
constexpr int BLOCK = 64 * WAVES_PER_BLOCK;
constexpr int LDS_BYTES = WAVES_PER_BLOCK * 2 * LDS_JOB * 4;  // 36864 B -> 3 (4) blocks per CU

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef u32x4 u32x4_u __attribute__((aligned(1)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef f32x4 f32x4_u __attribute__((aligned(4)));
typedef double f64x2 __attribute__((ext_vector_type(2)));
typedef f64x2 f64x2_u __attribute__((aligned(8)));

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

// ---- one tile row from global memory: mean-offset, zero-clipped samples + window statistics ----
// ffpiv normalize_intensity (A3): (a - mean)/std (0 if std == 0), clipped to >= 0.  The 1/std factor
// is bilinear in the correlation and is applied to the spectrum later; this returns it.
//   nonzero : number of non-zero samples of the window (signal pre-mask, A7); only if want_nz
//   finite  : cleared when the window holds NaN / Inf
// uint8: sum and sum of squares are exact integers (v_dot4_u32_u8), variance from
// n sum(x^2) - (sum x)^2 in 64 bits -- no per-sample arithmetic besides convert / subtract / clip.
// A row as fetched from HBM.  uint8 rows are 8 dwords, cheap enough to prefetch for BOTH windows of a
// job before any arithmetic starts; float rows are 32-64 dwords, so only their address is kept and the
// load is issued where the samples are consumed (the other waves of the SIMD cover that latency).
template <typename T>
struct RowRaw {
  const T* p;
  __device__ __forceinline__ void fetch(const T* q) { p = q; }
};
template <>
struct RowRaw<uint8_t> {
  uint32_t w[8];
  __device__ __forceinline__ void fetch(const uint8_t* q) {
    const u32x4 lo = *reinterpret_cast<const u32x4_u*>(q);
    const u32x4 hi = *reinterpret_cast<const u32x4_u*>(q + 16);
    w[0] = lo[0]; w[1] = lo[1]; w[2] = lo[2]; w[3] = lo[3];
    w[4] = hi[0]; w[5] = hi[1]; w[6] = hi[2]; w[7] = hi[3];
  }
};

struct RowStats {
  float mean;     // window mean
  float inv_std;  // 1 / population std, 0 for a zero-variance window
};

// window statistics of a uint8 row set, straight from the packed bytes
__device__ __forceinline__ RowStats stats_u8(const RowRaw<uint8_t>& raw, bool want_nz, int& nonzero) {
  const uint32_t (&w)[8] = raw.w;
  uint32_t s = 0, q = 0;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    s = __builtin_amdgcn_udot4(w[k], 0x01010101u, s, false);
    q = __builtin_amdgcn_udot4(w[k], w[k], q, false);
  }
  if (want_nz) {
    int nz = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {  // 0x80 in every zero byte
      const uint32_t t = ~(((w[k] & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | w[k] | 0x7F7F7F7Fu);
      nz += 4 - __builtin_popcount(t);
    }
    nonzero = half_sum_i(nz);
  }
  const uint32_t S = (uint32_t)half_sum_i((int)s);  // <= 255 * 1024
  const uint32_t Q = (uint32_t)half_sum_i((int)q);  // <= 255^2 * 1024 < 2^31
  RowStats st;
  st.mean = (float)S * (1.0f / 1024.0f);            // exact
  const uint64_t n2var = ((uint64_t)Q << 10) - (uint64_t)S * (uint64_t)S;
  const float var = (float)n2var * (1.0f / (1024.0f * 1024.0f));
  st.inv_std = n2var != 0 ? __builtin_amdgcn_rsqf(var) : 0.0f;
  return st;
}

// x = max((byte - mean) * g, 0), g >= 0: convert + fma + max per sample
__device__ __forceinline__ void center_u8(const RowRaw<uint8_t>& raw, float mean, float g, float (&x)[32]) {
  const uint32_t (&w)[8] = raw.w;
  const float off = -mean * g;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    x[4 * k + 0] = fmaxf(fmaf((float)(w[k] & 0xffu), g, off), 0.0f);
    x[4 * k + 1] = fmaxf(fmaf((float)((w[k] >> 8) & 0xffu), g, off), 0.0f);
    x[4 * k + 2] = fmaxf(fmaf((float)((w[k] >> 16) & 0xffu), g, off), 0.0f);
    x[4 * k + 3] = fmaxf(fmaf((float)(w[k] >> 24), g, off), 0.0f);
  }
}

__device__ __forceinline__ float center_clip_f(float (&x)[32], bool want_nz, int& nonzero, bool& finite) {
  if (want_nz) {
    int c = 0;
#pragma unroll
    for (int k = 0; k < 32; ++k) c += (x[k] != 0.0f) ? 1 : 0;
    nonzero = half_sum_i(c);
  }
  const float s = half_sum(tree_sum32(x));  // pairwise: a constant window gives its value exactly
  const float mean = s * (1.0f / 1024.0f);
  float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
  for (int k = 0; k < 32; ++k) {
    const float d = x[k] - mean;
    acc[k & 3] = fmaf(d, d, acc[k & 3]);
    x[k] = fmaxf(d, 0.0f);
  }
  const float ssq = half_sum((acc[0] + acc[1]) + (acc[2] + acc[3]));
  finite = finite && (fabsf(s) <= 3.0e38f) && (ssq <= 3.0e38f);
  const float var = ssq * (1.0f / 1024.0f);
  return var > 0.0f ? __builtin_amdgcn_rsqf(var) : 0.0f;
}
__device__ __forceinline__ float load_center(const RowRaw<float>& raw, float (&x)[32], bool want_nz, int& nonzero,
                                             bool& finite) {
  const float* p = raw.p;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const f32x4 v = *reinterpret_cast<const f32x4_u*>(p + 4 * k);
    x[4 * k + 0] = v[0]; x[4 * k + 1] = v[1]; x[4 * k + 2] = v[2]; x[4 * k + 3] = v[3];
  }
  return center_clip_f(x, want_nz, nonzero, finite);
}
__device__ __forceinline__ float load_center(const RowRaw<double>& raw, float (&x)[32], bool want_nz, int& nonzero,
                                             bool& finite) {
  const double* p = raw.p;
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const f64x2 v = *reinterpret_cast<const f64x2_u*>(p + 2 * k);
    x[2 * k + 0] = (float)v[0]; x[2 * k + 1] = (float)v[1];
  }
  return center_clip_f(x, want_nz, nonzero, finite);
}

// Both windows of a pair -> xr = a'' (mean-offset, zero-clipped), xi = rho b''.
// Balance: b is rescaled to a's variance (rho = inv_b / inv_a) so |A| ~ |B| and the
// |Z[k]|^2 - |Z[-k]|^2 difference of the cross spectrum does not cancel catastrophically when one
// window is much fainter than the other; corr = inv_a inv_b corr(a'', b'') = inv_a^2 corr(a'', rho b'').
// `scale` (= inv_a^2 / (4 N^2)) goes onto R BEFORE the two windows of a job are packed into one inverse
// transform: both planes then peak at <= 1, so float32 rounding of the shared inverse is relative to
// O(1) for each of them (scaling after the inverse lets a bright window's rounding noise swamp a faint
// neighbour packed with it).  A zero-variance window gives an exactly-zero plane (scale 0, clip ceiling
// hi = 0), the reference's zeros-if-std-is-0 rule (A3).
__device__ __forceinline__ void finish_pair(float inv_a, float inv_b, float& rho, float& scale, float& hi) {
  const bool dead = (inv_a == 0.0f) || (inv_b == 0.0f);
  rho = dead ? 0.0f : inv_b * __builtin_amdgcn_rcpf(inv_a);
  scale = dead ? 0.0f : inv_a * inv_a * (1.0f / (4.0f * 1024.0f * 1024.0f));
  hi = dead ? 0.0f : 1.0f;
}
__device__ __forceinline__ bool below_threshold(int nza, int nzb, float thr) {
  const float fa = (float)nza * (1.0f / 1024.0f), fb = (float)nzb * (1.0f / 1024.0f);
  return !(fa >= thr && fb >= thr);
}
__device__ __forceinline__ void prepare_pair(const RowRaw<uint8_t>& ra, const RowRaw<uint8_t>& rb, float (&xr)[32],
                                             float (&xi)[32], bool want_nz, float thr, float& scale, float& hi,
                                             bool& skip) {
  int nza = 1024, nzb = 1024;
  const RowStats sa = stats_u8(ra, want_nz, nza);
  const RowStats sb = stats_u8(rb, want_nz, nzb);
  float rho;
  finish_pair(sa.inv_std, sb.inv_std, rho, scale, hi);
  center_u8(ra, sa.mean, 1.0f, xr);
  center_u8(rb, sb.mean, rho, xi);  // the balance factor rides on the conversion
  skip = want_nz && below_threshold(nza, nzb, thr);
}
template <typename T>
__device__ __forceinline__ void prepare_pair(const RowRaw<T>& ra, const RowRaw<T>& rb, float (&xr)[32],
                                             float (&xi)[32], bool want_nz, float thr, float& scale, float& hi,
                                             bool& skip) {
  bool finite = true;
  int nza = 1024, nzb = 1024;
  const float inv_a = load_center(ra, xr, want_nz, nza, finite);
  const float inv_b = load_center(rb, xi, want_nz, nzb, finite);
  float rho;
  finish_pair(inv_a, inv_b, rho, scale, hi);
#pragma unroll
  for (int j = 0; j < 32; ++j) xi[j] *= rho;
  skip = !finite || (want_nz && below_threshold(nza, nzb, thr));
}

// LDS transpose of one real 32x32 plane held as lane = row: lane r scatters its row down column r of
// the buffer (ds_write_b32, the 32 lanes of a half hit 32 consecutive banks), then reads buffer row r
// = tile column r with ds_read_b128 (row stride 36 dwords: 16-byte aligned, and the 16 lanes of a b128
// group land on 16 distinct 4-bank slots since 9 l mod 16 is a bijection).
__device__ __forceinline__ void transpose_plane(float* buf, int l32, float (&x)[32]) {
  float* wcol = buf + l32;
#pragma unroll
  for (int j = 0; j < 32; ++j) wcol[j * LDS_ROW] = x[j];
  __builtin_amdgcn_wave_barrier();  // same wave: LDS ops execute in order, this only pins the compiler
  const f32x4* rrow = reinterpret_cast<const f32x4*>(buf + l32 * LDS_ROW);
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const f32x4 v = rrow[q];
    x[4 * q] = v[0]; x[4 * q + 1] = v[1]; x[4 * q + 2] = v[2]; x[4 * q + 3] = v[3];
  }
  __builtin_amdgcn_wave_barrier();
}
__device__ __forceinline__ void transpose32(float* buf, int l32, float (&xr)[32], float (&xi)[32]) {
  transpose_plane(buf, l32, xr);
  transpose_plane(buf, l32, xi);
}

__device__ __forceinline__ float bperm_f(int addr, float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(addr, __builtin_bit_cast(int, v)));
}

// lane = kx, registers = ky hold Z = FFT2(a + i b).  Writes s * 4 conj(A) B for ky = 0..16 into
// (rr, ri); Z[-k] = mirrored lane's Z[32 - ky].
__device__ __forceinline__ void cross_spectrum_half(int partner_byte, const float (&zr)[32], const float (&zi)[32],
                                                    float s, float (&rr)[17], float (&ri)[17]) {
#pragma unroll
  for (int ky = 0; ky <= 16; ++ky) {
    const int kn = (32 - ky) & 31;
    const float wr = bperm_f(partner_byte, zr[kn]);
    const float wi = bperm_f(partner_byte, zi[kn]);
    const float ar = zr[ky], ai = zi[ky];
    rr[ky] = (2.0f * s) * (ar * wi + ai * wr);
    ri[ky] = s * ((wr * wr + wi * wi) - (ar * ar + ai * ai));
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
template <typename T, bool WANT_NZ>
__device__ __forceinline__ void correlate_job(const PivParams& p, const TileRef (&t)[2], float* buf, int l32,
                                              int partner_byte, float (&xr)[32], float (&xi)[32], bool (&skip)[2]) {
  float R1r[17], R1i[17];  // s1 * 4 conj(A1) B1, ky = 0..16 (Hermitian half)
  float hi[2];             // clip ceiling: 1, or 0 for a zero-variance window (plane exactly 0)
  const T* frames = static_cast<const T*>(p.frames);
  constexpr bool want_nz = WANT_NZ;  // compile-time: a run-time branch here splits the pipeline into basic
                                     // blocks and the register allocator spills across them
  RowRaw<T> raw[2][2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const uint32_t wrow = p.div_ncols.div(t[k].win);
    const uint32_t wcol = t[k].win - wrow * (uint32_t)p.n_cols;
    const int64_t off = ((int64_t)t[k].pair * p.H + (int64_t)(wrow * p.sy + l32)) * p.W + (int64_t)wcol * p.sx;
    raw[k][0].fetch(frames + off);
    raw[k][1].fetch(frames + off + p.frame_elems);
  }
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    // keep the two windows' register-hungry phases apart: the scheduler otherwise interleaves window 1's
    // conversion with window 0's column FFT and spills (168-VGPR budget for three waves per SIMD)
    __builtin_amdgcn_sched_barrier(0);
    float scale;
    prepare_pair(raw[k][0], raw[k][1], xr, xi, want_nz, p.signal_threshold, scale, hi[k], skip[k]);
    fft32<false>(xr, xi);            // along x
    transpose32(buf, l32, xr, xi);   // lane = kx, regs = y
    fft32<false>(xr, xi);            // along y -> Z[ky][kx]
    if (k == 0) {
      cross_spectrum_half(partner_byte, xr, xi, scale, R1r, R1i);
    } else {
      float R2r[17], R2i[17];
      cross_spectrum_half(partner_byte, xr, xi, scale, R2r, R2i);
      // Q = R1 + i R2 for ky = 0..16 directly; for ky = 17..31 use R[ky][kx] = conj(R[32-ky][-kx]):
      // Q[ky][kx] = conj( (R1 - i R2)[32-ky][-kx] ), fetched from the mirrored lane.
#pragma unroll
      for (int ky = 0; ky <= 16; ++ky) {
        xr[ky] = R1r[ky] - R2i[ky];
        xi[ky] = R1i[ky] + R2r[ky];
      }
#pragma unroll
      for (int ky = 1; ky <= 15; ++ky) {
        const float mr = R1r[ky] + R2i[ky];   // (R1 - i R2).re
        const float mi = R1i[ky] - R2r[ky];   // (R1 - i R2).im
        xr[32 - ky] = bperm_f(partner_byte, mr);
        xi[32 - ky] = -bperm_f(partner_byte, mi);
      }
    }
  }
  __builtin_amdgcn_sched_barrier(0);
  fft32<true>(xr, xi);               // along ky
  transpose32(buf, l32, xr, xi);     // lane = y, regs = kx
  fft32<true>(xr, xi);               // along kx -> c1 + i c2
#pragma unroll
  for (int j = 0; j < 32; ++j) {
    xr[j] = __builtin_amdgcn_fmed3f(xr[j], 0.0f, hi[0]);
    xi[j] = __builtin_amdgcn_fmed3f(xi[j], 0.0f, hi[1]);
  }
}

// max / first-argmax (in fft-shifted row-major order) / sum of one plane held as lane = y, reg = x.
// The maximum is a v_max3 tree + DPP reduction; the arg-max is the smallest shifted flat index whose
// value equals it (np.argmax: first occurrence), found by an equality scan and a DPP min-reduction.
__device__ __forceinline__ void plane_stats(const float (&c)[32], int l32, float& vmax, int& imax, float& sum) {
  float m[11];
#pragma unroll
  for (int k = 0; k < 10; ++k) m[k] = fmaxf(fmaxf(c[3 * k], c[3 * k + 1]), c[3 * k + 2]);  // v_max3_f32
  m[10] = fmaxf(c[30], c[31]);
  float r = fmaxf(fmaxf(m[0], m[1]), m[2]);
  r = fmaxf(fmaxf(r, m[3]), m[4]);
  r = fmaxf(fmaxf(r, m[5]), m[6]);
  r = fmaxf(fmaxf(r, m[7]), m[8]);
  r = fmaxf(fmaxf(r, m[9]), m[10]);
  vmax = half_max(r);
  int bj = 1 << 10;  // "not in this row"
#pragma unroll
  for (int jj = 31; jj >= 0; --jj) bj = (c[(jj + 16) & 31] == vmax) ? jj : bj;  // ends on the smallest jj
  imax = half_min_i(((((l32 + 16) & 31) << 5) + bj));
  sum = half_sum(tree_sum32(c));
}

// park one plane in LDS (row y at buf[y * LDS_ROW + x]) and fit the peak: u, v in pixels
__device__ __forceinline__ void subpixel(float* buf, int l32, const float (&c)[32], int imax, float& u, float& v) {
  f32x4* wrow = reinterpret_cast<f32x4*>(buf + l32 * LDS_ROW);
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const f32x4 w = {c[4 * q], c[4 * q + 1], c[4 * q + 2], c[4 * q + 3]};
    wrow[q] = w;
  }
  __builtin_amdgcn_wave_barrier();
  const int ip = imax >> 5, jp = imax & 31;  // shifted coordinates
  const bool border = (ip == 0 || ip == 31 || jp == 0 || jp == 31);
  const int y = (ip + 16) & 31, x = (jp + 16) & 31;
  const int ym = (ip + 15) & 31, yp = (ip + 17) & 31;
  const int xm = (jp + 15) & 31, xp = (jp + 17) & 31;
  const float c0 = buf[y * LDS_ROW + x] + kEpsPeak;
  const float cl = buf[ym * LDS_ROW + x] + kEpsPeak;
  const float cr = buf[yp * LDS_ROW + x] + kEpsPeak;
  const float cd = buf[y * LDS_ROW + xm] + kEpsPeak;
  const float cu = buf[y * LDS_ROW + xp] + kEpsPeak;
  __builtin_amdgcn_wave_barrier();
  // the fit is a ratio of log differences, so any base works: v_log_f32 (log2, 1 ulp) on inputs >= 1e-7
  const float l0 = __builtin_amdgcn_logf(c0);
  v = (float)ip + gauss_offset_fast(__builtin_amdgcn_logf(cl), l0, __builtin_amdgcn_logf(cr)) - 16.0f;
  u = (float)jp + gauss_offset_fast(__builtin_amdgcn_logf(cd), l0, __builtin_amdgcn_logf(cu)) - 16.0f;
  if (border) u = v = __builtin_nanf("");
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

// ---- per-timestep kernel: one job (two neighbouring windows of one pair) per half-wave ---------
// uint8 frames fit the 168-VGPR budget of three waves per SIMD; float / double rows are loaded where they are
// consumed (no cheap prefetch) and need a few more registers: two waves per SIMD for those.
template <typename T>
constexpr int kWavesPerSimd = sizeof(T) == 1 ? 3 : 2;

template <typename T, bool PLANES, bool WANT_NZ>
__global__ __launch_bounds__(BLOCK, kWavesPerSimd<T>) void piv_fft32_kernel(PivParams p) {
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
  const uint32_t pair = p.div_jobs.div(job);
  const uint32_t w0 = (job - pair * jobs_per_pair) * 2;
  TileRef t[2];
  t[0].pair = t[1].pair = pair;
  t[0].win = w0;
  t[0].valid = job_valid;
  t[1].valid = job_valid && (w0 + 1 < p.n_win);
  t[1].win = (w0 + 1 < p.n_win) ? w0 + 1 : w0;

  float xr[32], xi[32];
  bool skip[2];
  correlate_job<T, WANT_NZ>(p, t, buf, l32, partner_byte, xr, xi, skip);

  const float nanv = __builtin_nanf("");
  {
    float vmax, sum, u, v;
    int imax;
    plane_stats(xr, l32, vmax, imax, sum);
    subpixel(buf, l32, xr, imax, u, v);
    float cm = vmax, sn = vmax * __builtin_amdgcn_rcpf(sum * (1.0f / 1024.0f));
    if (skip[0]) u = v = cm = sn = nanv;
    if (t[0].valid && l32 == 0) {
      const uint32_t g = t[0].pair * p.n_win + t[0].win;
      p.u[g] = u; p.v[g] = v; p.cmax[g] = cm; p.s2n[g] = sn;
    }
  }
  {
    float vmax, sum, u, v;
    int imax;
    plane_stats(xi, l32, vmax, imax, sum);
    subpixel(buf, l32, xi, imax, u, v);
    float cm = vmax, sn = vmax * __builtin_amdgcn_rcpf(sum * (1.0f / 1024.0f));
    if (skip[1]) u = v = cm = sn = nanv;
    if (t[1].valid && l32 == 0) {
      const uint32_t g = t[1].pair * p.n_win + t[1].win;
      p.u[g] = u; p.v[g] = v; p.cmax[g] = cm; p.s2n[g] = sn;
    }
  }
  if constexpr (PLANES) {
    if (t[0].valid) store_plane_rows(p.planes + ((size_t)t[0].pair * p.n_win + t[0].win) * 1024, l32, xr, skip[0]);
    if (t[1].valid) store_plane_rows(p.planes + ((size_t)t[1].pair * p.n_win + t[1].win) * 1024, l32, xi, skip[1]);
  }
}

// ---- ensemble kernel: a job owns two windows and walks all pairs of the chunk in order ---------
// (pyorc/velocimetry/ffpiv.py:222-241,361-363): planes failing corr_min / s2n_min / finite are
// zeroed, corr_sum += plane, corr_count += (corr_max > 1e-6).  The accumulation order is the
// pair order, one owner per window => bit-reproducible, no atomics.
template <typename T, bool WANT_NZ>
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
  float cnt0 = 0.0f, cnt1 = 0.0f;
  for (uint32_t pair = 0; pair < p.n_pairs; ++pair) {
    TileRef t[2] = {{pair, w[0], valid[0]}, {pair, w[1], valid[1]}};
    float xr[32], xi[32];
    bool skip[2];
    correlate_job<T, WANT_NZ>(p, t, buf, l32, partner_byte, xr, xi, skip);
    float vmax, sum;
    int imax;
    plane_stats(xr, l32, vmax, imax, sum);
    {
      float cm = vmax, sn = vmax * __builtin_amdgcn_rcpf(sum * (1.0f / 1024.0f));
      const bool keep = !skip[0] && (cm >= p.corr_min) && (sn >= p.s2n_min);  // NaN s2n compares false
      cm = keep ? cm : 0.0f;
      sn = keep ? sn : 0.0f;
      cnt0 += (cm > 1e-6f) ? 1.0f : 0.0f;
      if (valid[0] && l32 == 0) {
        p.cmax[(size_t)pair * p.n_win + w[0]] = cm;
        p.s2n[(size_t)pair * p.n_win + w[0]] = sn;
      }
#pragma unroll
      for (int j = 0; j < 32; ++j) acc0[j] += keep ? xr[j] : 0.0f;
    }
    plane_stats(xi, l32, vmax, imax, sum);
    {
      float cm = vmax, sn = vmax * __builtin_amdgcn_rcpf(sum * (1.0f / 1024.0f));
      const bool keep = !skip[1] && (cm >= p.corr_min) && (sn >= p.s2n_min);
      cm = keep ? cm : 0.0f;
      sn = keep ? sn : 0.0f;
      cnt1 += (cm > 1e-6f) ? 1.0f : 0.0f;
      if (valid[1] && l32 == 0) {
        p.cmax[(size_t)pair * p.n_win + w[1]] = cm;
        p.s2n[(size_t)pair * p.n_win + w[1]] = sn;
      }
#pragma unroll
      for (int j = 0; j < 32; ++j) acc1[j] += keep ? xi[j] : 0.0f;
    }
  }
  // corr_sum is kept in fft-shifted layout (what u_v_displacement expects)
  if (valid[0]) {
    float* row = p.corr_sum + (size_t)w[0] * 1024 + ((l32 + 16) & 31) * 32;
#pragma unroll
    for (int qd = 0; qd < 8; ++qd) {
      f32x4 old = *reinterpret_cast<f32x4*>(row + 4 * qd);
#pragma unroll
      for (int e = 0; e < 4; ++e) old[e] += acc0[(4 * qd + e + 16) & 31];
      *reinterpret_cast<f32x4*>(row + 4 * qd) = old;
    }
    if (l32 == 0) p.corr_count[w[0]] += cnt0;
  }
  if (valid[1]) {
    float* row = p.corr_sum + (size_t)w[1] * 1024 + ((l32 + 16) & 31) * 32;
#pragma unroll
    for (int qd = 0; qd < 8; ++qd) {
      f32x4 old = *reinterpret_cast<f32x4*>(row + 4 * qd);
#pragma unroll
      for (int e = 0; e < 4; ++e) old[e] += acc1[(4 * qd + e + 16) & 31];
      *reinterpret_cast<f32x4*>(row + 4 * qd) = old;
    }
    if (l32 == 0) p.corr_count[w[1]] += cnt1;
  }
}

template <typename T, bool WANT_NZ>
static hipError_t launch_t(const PivParams& p, bool ensemble, hipStream_t s) {
  if (ensemble) {
    const uint32_t jobs = (p.n_win + 1) / 2;
    const uint32_t blocks = (jobs + 2 * WAVES_PER_BLOCK - 1) / (2 * WAVES_PER_BLOCK);
    hipLaunchKernelGGL((piv_fft32_ensemble_kernel<T, WANT_NZ>), dim3(blocks), dim3(BLOCK), LDS_BYTES, s, p);
    return hipGetLastError();
  }
  const uint32_t jobs = p.n_pairs * ((p.n_win + 1) / 2);
  const uint32_t blocks = (jobs + 2 * WAVES_PER_BLOCK - 1) / (2 * WAVES_PER_BLOCK);
  // LSPIV_DEBUG_EXTRA_LDS: occupancy experiments only (pads the LDS request so fewer blocks fit a CU)
  static const int extra_lds = getenv("LSPIV_DEBUG_EXTRA_LDS") ? atoi(getenv("LSPIV_DEBUG_EXTRA_LDS")) : 0;
  if (p.planes)
    hipLaunchKernelGGL((piv_fft32_kernel<T, true, WANT_NZ>), dim3(blocks), dim3(BLOCK), LDS_BYTES + extra_lds, s, p);
  else
    hipLaunchKernelGGL((piv_fft32_kernel<T, false, WANT_NZ>), dim3(blocks), dim3(BLOCK), LDS_BYTES + extra_lds, s, p);
  return hipGetLastError();
}

template <typename T>
static hipError_t launch_nz(const PivParams& p, bool ensemble, hipStream_t s) {
  return p.signal_threshold >= 0.0f ? launch_t<T, true>(p, ensemble, s) : launch_t<T, false>(p, ensemble, s);
}

hipError_t launch_piv_fft32(const PivParams& p, int dtype, bool ensemble, hipStream_t s) {
  switch (dtype) {
    case 0: return launch_nz<uint8_t>(p, ensemble, s);
    case 1: return launch_nz<float>(p, ensemble, s);
    case 2: return launch_nz<double>(p, ensemble, s);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace lspiv
