// Shared declarations of the LSPIV HIP kernels (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>

namespace lspiv {

// Division by a launch-invariant 32-bit divisor without the ~25-instruction expansion of `/`:
// q = (umulhi(magic, n) + ((n - umulhi(magic, n)) >> 1)) >> shift   (round-up method, branch free).
struct FastDiv {
  uint32_t magic, shift, d;
  static __host__ FastDiv make(uint32_t d) {
    FastDiv f;
    f.d = d;
    uint32_t l = 31;
    while (l > 0 && !((d >> l) & 1u)) --l;  // floor(log2 d), d >= 1
    if ((d & (d - 1)) == 0) {
      f.magic = 0;
      f.shift = l == 0 ? 0 : l - 1;
      return f;
    }
    const uint64_t num = (uint64_t)1 << (32 + l);
    uint64_t m = num / d;
    const uint64_t rem = num - m * d;
    m += m;
    const uint64_t twice_rem = rem + rem;
    if (twice_rem >= d) m += 1;
    f.magic = (uint32_t)(m + 1);
    f.shift = l;
    return f;
  }
  __host__ __device__ __forceinline__ uint32_t div(uint32_t n) const {
    if (d == 1) return n;  // uniform branch, never taken for real grids
#ifdef __HIP_DEVICE_COMPILE__
    const uint32_t q = __umulhi(magic, n);
#else
    const uint32_t q = (uint32_t)(((uint64_t)magic * n) >> 32);
#endif
    return (((n - q) >> 1) + q) >> shift;
  }
};

// header of the rescue lists in device memory (zeroed once; the rescue kernel's last block resets it after every pass)
struct RescueHdr {
  uint32_t n_fit, n_amb;         // records appended by the PIV kernel of this pass (may exceed the capacities: the excess is dropped)
  uint32_t spare0, spare1;
  uint32_t done_blocks, pad;
  uint32_t last_fit, last_amb;   // counts of the last completed pass
  unsigned long long total_fit, total_amb, total_windows;
};

// One launch = all interrogation-window pairs of a frame chunk.
// Replaces, fused: ffpiv.cross_corr + corr_max/s2n reductions + ffpiv.u_v_displacement
// (pyorc/velocimetry/ffpiv.py:446-474).
struct PivParams {
  const void* frames;      // device (T, H, W), dtype by template
  int64_t frame_elems;     // H * W
  int H, W;
  int wy, wx;              // window (== search area, pyorc/api/frames.py:168)
  int sy, sx;              // window stride = window - overlap
  int n_rows, n_cols;
  uint32_t n_win;          // n_rows * n_cols
  uint32_t n_tiles;        // (T-1) * n_win
  float signal_threshold;  // < 0: off
  // the three choices of the engine the oracle could not pin on a real ffpiv run, as
  // run-time options so that the default can follow whatever ffpiv turns out to do (lspiv_set_option)
  int border_mode;         // arg-max on the plane border: 0 NaN (default), 1 the plane centre (zero displacement), 2 the integer peak
  int nz_positive;         // signal score counts samples != 0 (0, default) or > 0 (1)
  const uint8_t* win_keep; // nullptr, or n_win flags of the "stack" signal mode: 0 = this window position is dropped
  // two more unpinned readings (A3), run-time options like the three above: the standard deviation of the window
  // normalisation (std_gain = 1 for the population value, sqrt((n - 1) / n) for the sample value -- a scale on every
  // normalised window, std_gain2 = its square on every plane), and whether negative lobes are clipped after it (norm_clip = 0
  // is served by the block-per-window kernels only: lspiv_kernel_kind)
  float std_gain, std_gain2;
  int norm_clip;
  float* u;                // each n_tiles float32
  float* v;
  float* cmax;
  float* s2n;
  float* planes;           // nullptr or n_tiles * wy * wx
  // ensemble mode (lspiv_ensemble_accumulate): per-pair masks, running plane sum
  float corr_min, s2n_min;
  float* corr_sum;         // n_win * wy * wx
  float* corr_count;       // n_win
  // walking ensemble kernels: per-segment partial sums / counts (zeroed by the caller), merged in segment order
  float* part_sum;         // n_seg * n_win * wy * wx, or nullptr: the single-owner kernels
  float* part_cnt;         // n_seg * n_win
  float* dft_scratch;      // windows above 128 px: per-block slots of HBM for the two planes (piv_dft_global_kernel), else nullptr
  size_t dft_slot;         // floats per slot
  uint32_t seg_len, n_seg; // pairs per segment (odd), number of segments
  uint32_t seg_first;      // pairs in segment 0: seg_len, or what is left up to the next anchor when the chunk starts off-anchor
  uint32_t n_pairs;        // T-1
  uint32_t xcd_by_windows; // walking kernels: 0 = every XCD gets a contiguous range of (segment, window) jobs; 1 = an eighth of the windows of every segment (piv_fft_impl.h, walk_job)
  uint32_t strip_w;        // walking kernels: 0 = a segment's jobs run over the window grid row by row; w > 0 = in column strips of w
                           // windows, every strip top to bottom (the vertically overlapping windows of a band share a round of jobs)
  int64_t pair_offset;     // absolute index of the chunk's first pair in the caller's stack: the walking kernels cut
                           // segments at multiples of the anchor length of THAT index, so results do not depend on the chunking
  // float64 rescue pass (piv_rescue.hip): the kernels' epilogues append the windows whose float32
  // sub-pixel result cannot be trusted to 1e-4 -- the arg-max is not unique under float32 plane noise ("amb"), or the
  // log-Gaussian fit amplifies that noise beyond the tolerance ("fit") -- to two device lists; nullptr: off
  RescueHdr* rescue_hdr;
  uint4* rescue_fit;       // {result index, (ip << 16) | jp, second candidate or ~0, -}: the five samples of the fit in float64
  uint32_t* rescue_amb;    // result index: re-evaluate the whole plane in float64
  uint32_t rescue_cap_fit, rescue_cap_amb;
  float rescue_k;          // 2 kappa / (ln 2 * 1e-4), kappa = absolute plane noise / plane maximum the flag assumes
  float rescue_tau;        // relative arg-max gap below which the arg-max counts as not unique
  FastDiv div_ncols;       // window index -> (row, col)
  FastDiv div_jobs;        // fft kernels: job index -> (pair, job in pair), divisor (n_win + 1) / 2
  FastDiv div_nwin;        // one-window-per-job kernels: job index -> (pair, window), divisor n_win
};

// ---- float64 rescue of the ENSEMBLE's final fit (piv_rescue.hip) ---------------------------------
// lspiv_ensemble_finish fits the MEAN plane; where that float32 fit is ill-conditioned (same flag model as above) the five
// samples it reads are re-evaluated in float64 from the retained frames: c[k] = (1 / count) sum over the KEPT pairs of
// clip01((1 / n) sum_m a'[m] b'[m + k]).  A record lists the window and up to four candidates of the arg-max (the samples of
// the float32 mean plane within tau of its maximum, ascending row-major index): each gets its five float64 sums, the largest
// centre wins.  Partial sums per (record, block of pairs) are merged in block order: deterministic, no atomics on results.
constexpr int kEnsMaxCand = 4;
constexpr int kEnsPairBlock = 16;       // pairs per partial sum
struct EnsRescueRec {
  uint32_t w;                 // window
  uint32_t ncand;             // 1 .. kEnsMaxCand; 0: more candidates than that -- the float32 result stays
  uint32_t pos[kEnsMaxCand];  // (ip << 16) | jp in the fft-shifted plane
  uint32_t pad[2];
};
struct EnsRescueHdr { uint32_t n_rec, n_skipped, pad[2]; };
struct EnsRescueArgs {
  const EnsRescueRec* recs;
  uint32_t n_rec;
  const float* cmax;          // this chunk's masked per-pair corr_max, (n_pairs, n_win): > 0 <=> the pair was added to the sum
  uint32_t n_pairs;           // pairs of this chunk
  uint32_t blk0, n_blk;       // this chunk's first pair-block in `partial`, pair-blocks over all retained chunks
  double* partial;            // (n_rec, n_blk, kEnsMaxCand, 5)
  const float* count;         // (n_win) pairs in the sum
};

// ---- wave64 cross-lane helpers -----------------------------------------------------------------
// DPP row operations act inside rows of 16 lanes; ds_swizzle(SWAP,16) joins the two rows of a
// 32-lane half, ds_swizzle cannot cross the 32-lane boundary (which is what we want: the two
// halves of a wave work on different tiles).
template <int CTRL>
__device__ __forceinline__ float dpp_f(float x) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xF, 0xF, true));
}
template <int CTRL>
__device__ __forceinline__ int dpp_i(int x) {
  return __builtin_amdgcn_update_dpp(0, x, CTRL, 0xF, 0xF, true);
}
constexpr int DPP_XOR1 = 0xB1;         // quad_perm [1,0,3,2]
constexpr int DPP_XOR2 = 0x4E;         // quad_perm [2,3,0,1]
constexpr int DPP_HALF_MIRROR = 0x141; // lane ^ 7 inside 8
constexpr int DPP_MIRROR = 0x140;      // lane ^ 15 inside 16
constexpr int SWZ_XOR16 = 0x401F;      // ds_swizzle bit mode: and 0x1f, or 0, xor 0x10
constexpr int SWZ_XOR8 = 0x201F;

__device__ __forceinline__ float swz16_f(float x) {
  return __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, x), SWZ_XOR16));
}
__device__ __forceinline__ int swz16_i(int x) { return __builtin_amdgcn_ds_swizzle(x, SWZ_XOR16); }

// sum over the 32 lanes of a half-wave; every lane gets the total (fixed order => deterministic)
__device__ __forceinline__ float half_sum(float x) {
  x += dpp_f<DPP_XOR1>(x);
  x += dpp_f<DPP_XOR2>(x);
  x += dpp_f<DPP_HALF_MIRROR>(x);
  x += dpp_f<DPP_MIRROR>(x);
  x += swz16_f(x);
  return x;
}
__device__ __forceinline__ int half_sum_i(int x) {
  x += dpp_i<DPP_XOR1>(x);
  x += dpp_i<DPP_XOR2>(x);
  x += dpp_i<DPP_HALF_MIRROR>(x);
  x += dpp_i<DPP_MIRROR>(x);
  x += swz16_i(x);
  return x;
}
// the same over one DPP row of 16 lanes (16 x 16 windows: four jobs per wave)
__device__ __forceinline__ float row_sum(float x) {
  x += dpp_f<DPP_XOR1>(x);
  x += dpp_f<DPP_XOR2>(x);
  x += dpp_f<DPP_HALF_MIRROR>(x);
  x += dpp_f<DPP_MIRROR>(x);
  return x;
}
__device__ __forceinline__ int row_sum_i(int x) {
  x += dpp_i<DPP_XOR1>(x);
  x += dpp_i<DPP_XOR2>(x);
  x += dpp_i<DPP_HALF_MIRROR>(x);
  x += dpp_i<DPP_MIRROR>(x);
  return x;
}
// The two 32-lane halves of a wave joined with gfx950's v_permlane32_swap: ONE VALU instruction hands every lane l the pair
// (x[l mod 32], x[32 + l mod 32]) -- where __shfl_xor(x, 32) is a ds_bpermute through the LDS pipe plus an s_waitcnt.  Both
// halves combine the pair in the same operand order (lower, upper); for +, max and min that gives the bits of the exchange.
// Inline assembly: ROCm 7.2's __builtin_amdgcn_permlane32_swap maps BOTH of its results to the first register (a one-line
// test kernel stores v1 twice), so the instruction is written out, with the two wait states the compiler itself puts between
// a VALU write and this read.
__device__ __forceinline__ void permlane32_swap(unsigned& a, unsigned& b) {   // a[32..63] <-> b[0..31]
  asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}
struct HalfPair { float lo, hi; };
__device__ __forceinline__ HalfPair half_pair_f(float x) {
  unsigned a = __builtin_bit_cast(unsigned, x), b = a;
  permlane32_swap(a, b);   // a = lower half's values in both halves, b = upper half's
  return {__builtin_bit_cast(float, a), __builtin_bit_cast(float, b)};
}
__device__ __forceinline__ float cross_half_sum(float x) { const HalfPair h = half_pair_f(x); return h.lo + h.hi; }
__device__ __forceinline__ float cross_half_max(float x) { const HalfPair h = half_pair_f(x); return fmaxf(h.lo, h.hi); }
__device__ __forceinline__ int cross_half_sum_i(int x) {
  unsigned a = (unsigned)x, b = a;
  permlane32_swap(a, b);
  return (int)(a + b);
}
__device__ __forceinline__ int cross_half_max_i(int x) {
  unsigned a = (unsigned)x, b = a;
  permlane32_swap(a, b);
  return max((int)a, (int)b);
}
__device__ __forceinline__ int cross_half_min_i(int x) {
  unsigned a = (unsigned)x, b = a;
  permlane32_swap(a, b);
  return min((int)a, (int)b);
}
// sum over all 64 lanes (two halves joined by the cross-half swap)
__device__ __forceinline__ float wave_sum(float x) {
  x = half_sum(x);
  return cross_half_sum(x);
}

// lexicographic arg-max step: keep (v, idx) unless partner is larger, or equal with smaller idx
__device__ __forceinline__ void argmax_merge(float& v, int& idx, float pv, int pidx) {
  bool take = (pv > v) || (pv == v && pidx < idx);
  v = take ? pv : v;
  idx = take ? pidx : idx;
}
__device__ __forceinline__ void half_argmax(float& v, int& idx) {
  argmax_merge(v, idx, dpp_f<DPP_XOR1>(v), dpp_i<DPP_XOR1>(idx));
  argmax_merge(v, idx, dpp_f<DPP_XOR2>(v), dpp_i<DPP_XOR2>(idx));
  argmax_merge(v, idx, dpp_f<DPP_HALF_MIRROR>(v), dpp_i<DPP_HALF_MIRROR>(idx));
  argmax_merge(v, idx, dpp_f<DPP_MIRROR>(v), dpp_i<DPP_MIRROR>(idx));
  argmax_merge(v, idx, swz16_f(v), swz16_i(idx));
}

// 3-point log-Gaussian sub-pixel offset; zero denominator -> 0 (ffpiv peak_position, A5).
// The ratio is independent of the logarithm base.
__device__ __forceinline__ float gauss_offset(float lm, float l0, float lp) {
  float nom = lm - lp;
  float den = 2.0f * lm - 4.0f * l0 + 2.0f * lp;
  return den != 0.0f ? nom / den : 0.0f;
}
// same with a 1-ulp reciprocal instead of the ~10-instruction IEEE division (fused FFT kernels)
__device__ __forceinline__ float gauss_offset_fast(float lm, float l0, float lp) {
  float nom = lm - lp;
  float den = 2.0f * lm - 4.0f * l0 + 2.0f * lp;
  return den != 0.0f ? nom * __builtin_amdgcn_rcpf(den) : 0.0f;
}
__device__ __forceinline__ float gauss_offset_fast(float lm, float l0, float lp, float& den) {
  float nom = lm - lp;
  den = 2.0f * lm - 4.0f * l0 + 2.0f * lp;
  return den != 0.0f ? nom * __builtin_amdgcn_rcpf(den) : 0.0f;
}
// min over the 32 lanes of a half-wave
__device__ __forceinline__ int half_min_i(int x) {
  x = min(x, dpp_i<DPP_XOR1>(x));
  x = min(x, dpp_i<DPP_XOR2>(x));
  x = min(x, dpp_i<DPP_HALF_MIRROR>(x));
  x = min(x, dpp_i<DPP_MIRROR>(x));
  x = min(x, swz16_i(x));
  return x;
}
__device__ __forceinline__ float half_max(float x) {
  x = fmaxf(x, dpp_f<DPP_XOR1>(x));
  x = fmaxf(x, dpp_f<DPP_XOR2>(x));
  x = fmaxf(x, dpp_f<DPP_HALF_MIRROR>(x));
  x = fmaxf(x, dpp_f<DPP_MIRROR>(x));
  x = fmaxf(x, swz16_f(x));
  return x;
}

// max over a half-wave / a DPP row as signed integers: one v_max_i32 with a DPP operand per step.  The float reductions above
// cost three instructions per step (fmaxf has to quiet signalling NaNs: v_mov_dpp + v_max(x, x) + v_max), so the maxima of
// NON-NEGATIVE floats -- clipped correlation planes -- are taken on their bit patterns, which order like the values (a NaN
// ranks above every finite value and survives, as with fmaxf of a quiet NaN in every lane)
__device__ __forceinline__ int half_max_i(int x) {
  x = max(x, dpp_i<DPP_XOR1>(x));
  x = max(x, dpp_i<DPP_XOR2>(x));
  x = max(x, dpp_i<DPP_HALF_MIRROR>(x));
  x = max(x, dpp_i<DPP_MIRROR>(x));
  x = max(x, swz16_i(x));
  return x;
}
__device__ __forceinline__ int row_max_i(int x) {
  x = max(x, dpp_i<DPP_XOR1>(x));
  x = max(x, dpp_i<DPP_XOR2>(x));
  x = max(x, dpp_i<DPP_HALF_MIRROR>(x));
  x = max(x, dpp_i<DPP_MIRROR>(x));
  return x;
}
__device__ __forceinline__ int row_min_i(int x) {
  x = min(x, dpp_i<DPP_XOR1>(x));
  x = min(x, dpp_i<DPP_XOR2>(x));
  x = min(x, dpp_i<DPP_HALF_MIRROR>(x));
  x = min(x, dpp_i<DPP_MIRROR>(x));
  return x;
}
__device__ __forceinline__ float row_max(float x) {
  x = fmaxf(x, dpp_f<DPP_XOR1>(x));
  x = fmaxf(x, dpp_f<DPP_XOR2>(x));
  x = fmaxf(x, dpp_f<DPP_HALF_MIRROR>(x));
  x = fmaxf(x, dpp_f<DPP_MIRROR>(x));
  return x;
}

constexpr float kEpsPeak = 1e-7f;

// what u, v become when the arg-max sits on the plane border (no 3-point fit possible there): (dx, dy) = the integer
// peak minus the plane centre
__device__ __forceinline__ void border_result(int mode, int dx, int dy, float& u, float& v) {
  const float nanv = __builtin_nanf("");
  u = mode == 0 ? nanv : mode == 1 ? 0.0f : (float)dx;
  v = mode == 0 ? nanv : mode == 1 ? 0.0f : (float)dy;
}

// ---- which float32 peaks go to the float64 rescue pass ------------------------------------------------------------------
// A float32 plane carries absolute noise of <= ~3e-7 of its maximum (measured against the float64 oracle on 330 k windows of
// the benchmark stacks: 4.2e-7 worst).  (i) "amb": the runner-up is within rescue_tau of the maximum -- which sample is
// the arg-max (and whether it sits on the border => NaN) is then a matter of rounding.  (ii) "fit": the 3-point log-Gaussian
// offset nom / den moves by about 2 (d_m + d_0 + d_p) / |den| when the logs move by d_x = noise / c_x; with the smaller
// neighbour c_min that is <= 2 kappa (2 vmax / c_min + 1) / (ln 2 |den|) for logs to base 2, and the window is flagged when
// this exceeds 1e-4 max(|result|, 0.05 px) -- the parity gate of SURVEY.md section 8d.  Typical cause: a neighbour of the
// peak that is exactly 0 in exact arithmetic (clipped), where log(c + 1e-7) turns 1e-8 of rounding noise into 1e-3 px.
// cm_* = min of the two neighbours + eps, den_* in log2 units, res_* the float32 result of that axis.
struct PeakCond { bool amb, fit; };
// min / max of POSITIVE floats on their bit patterns (fminf / fmaxf cost an extra v_max(x, x) per operand to quiet signalling NaNs)
__device__ __forceinline__ float min_pos(float a, float b) { return __builtin_bit_cast(float, min(__builtin_bit_cast(int, a), __builtin_bit_cast(int, b))); }
__device__ __forceinline__ float max_abs(float a, float floor_) {   // max(|a|, floor_), floor_ > 0; a NaN stays a NaN
  return __builtin_bit_cast(float, max(__builtin_bit_cast(int, a) & 0x7fffffff, __builtin_bit_cast(int, floor_)));
}
// cl_, cr_ (cd_, cu_): the two neighbours of the peak along v (u), eps already added (> 0)
// near_tie: some sample other than the arg-max is >= vmax (1 - tau)
__device__ __forceinline__ PeakCond peak_cond(float vmax, bool near_tie, bool border, float cl_, float cr_, float den_v, float res_v,
                                              float cd_, float cu_, float den_u, float res_u, float k) {
  PeakCond c;
  const bool live = vmax > 0.0f;   // an all-zero plane is NaN by construction (first arg-max on the border)
  c.amb = live && near_tie;
  const float cm_v = min_pos(cl_, cr_), cm_u = min_pos(cd_, cu_);
  const float a_v = k * fmaf(2.0f, vmax, cm_v), a_u = k * fmaf(2.0f, vmax, cm_u);
  const bool bad_v = !(a_v <= max_abs(res_v, 0.05f) * fabsf(den_v) * cm_v);   // NaN compares false => flagged
  const bool bad_u = !(a_u <= max_abs(res_u, 0.05f) * fabsf(den_u) * cm_u);
  c.fit = live && !border && !c.amb && (bad_v || bad_u);
  return c;
}
// one lane appends the record (g = result index inside this launch); pos2: the only other arg-max candidate of an "amb"
// window ((ip << 16) | jp), or ~0 -- with exactly two candidates their two float64 sums settle it, no whole plane needed
__device__ __forceinline__ void rescue_note(RescueHdr* hdr, uint4* fit, uint32_t cap_fit, uint32_t* amb, uint32_t cap_amb,
                                            uint32_t g, PeakCond c, int ip, int jp, uint32_t pos2 = 0xffffffffu) {
  if (c.amb && pos2 == 0xffffffffu) {
    const uint32_t i = atomicAdd(&hdr->n_amb, 1u);
    if (i < cap_amb) amb[i] = g;
  } else if (c.amb || c.fit) {
    const uint32_t i = atomicAdd(&hdr->n_fit, 1u);
    if (i < cap_fit) fit[i] = make_uint4(g, ((uint32_t)ip << 16) | (uint32_t)jp, pos2, 0u);
  }
}

// element -> float conversion of the three frame dtypes
__device__ __forceinline__ float to_f32(uint8_t x) { return (float)x; }
__device__ __forceinline__ float to_f32(float x) { return x; }
__device__ __forceinline__ float to_f32(double x) { return (float)x; }

// launch entry points implemented by the kernel translation units
hipError_t launch_piv_fft8(const PivParams& p, int dtype, bool ensemble, hipStream_t s);
hipError_t launch_piv_fft16(const PivParams& p, int dtype, bool ensemble, hipStream_t s);
// square even windows that are no power of two (P * 2^m, P odd) with prime-factor FFT kernels of their own, one translation unit each (piv_fftNN.hip;
// keep in step with PFA_SIZES in the Makefile)
#define LSPIV_PFA_SIZES(X) X(6) X(10) X(12) X(14) X(18) X(20) X(22) X(24) X(26) X(28) X(30) X(34) X(36) X(38) X(40) X(42) X(44) X(46) \
  X(48) X(50) X(52) X(54) X(56) X(58) X(60) X(62)
#define LSPIV_PFA_DECL(n) hipError_t launch_piv_fft##n(const PivParams& p, int dtype, bool ensemble, hipStream_t s);
LSPIV_PFA_SIZES(LSPIV_PFA_DECL)
#undef LSPIV_PFA_DECL
hipError_t launch_piv_fft32(const PivParams& p, int dtype, bool ensemble, hipStream_t s);
hipError_t launch_piv_fft64(const PivParams& p, int dtype, bool ensemble, hipStream_t s);
hipError_t launch_piv_direct(const PivParams& p, int dtype, bool ensemble, hipStream_t s);
// windows above 64 px per side (any shape that fits LDS): packed 2-D DFT in LDS (piv_direct.hip)
hipError_t launch_piv_dft(const PivParams& p, int dtype, bool ensemble, hipStream_t s);
bool piv_dft_fits(int wy, int wx);
// windows above 128 px: the DFT passes on per-block slots of HBM scratch (piv_direct.hip); PivParams::dft_scratch / dft_slot
size_t piv_dft_global_slot_floats(int wy, int wx);
int piv_dft_global_blocks(uint32_t n_work);
hipError_t launch_piv_dft_global(const PivParams& p, int dtype, bool ensemble, hipStream_t s);
// square windows 4..16 / 17..31 through the 32- / 64-point transforms
hipError_t launch_piv_embed16(const PivParams& p, int dtype, bool ensemble, hipStream_t s);
hipError_t launch_piv_embed32(const PivParams& p, int dtype, bool ensemble, hipStream_t s);
hipError_t launch_piv_embed64(const PivParams& p, int dtype, bool ensemble, hipStream_t s);
// float64 re-evaluation of the windows the PIV kernel of this pass appended to p.rescue_* (piv_rescue.hip)
hipError_t launch_piv_rescue(const PivParams& p, int dtype, hipStream_t s);
// ensemble mode: flag the windows of the float32 mean planes (u, v = their float32 fits) whose fit cannot be trusted; partial
// float64 sums of one retained chunk (p: geometry + options + the chunk's frames); merge + fit, overwriting u, v
hipError_t launch_ens_flag(const float* mean, uint32_t n_win, int wy, int wx, const float* u, const float* v, float k, float tau,
                           EnsRescueHdr* hdr, EnsRescueRec* recs, uint32_t cap, hipStream_t s);
hipError_t launch_ens_partial(const PivParams& p, int dtype, const EnsRescueArgs& a, hipStream_t s);
// totals (n_rec, kEnsMaxCand * 5) = the partial sums of one handle merged in pair-block order
hipError_t launch_ens_merge(const EnsRescueArgs& a, double* totals, hipStream_t s);
// fit of the flagged windows from totals over ALL pairs of the sum (p: wy, wx, border_mode; a: recs, n_rec, count)
hipError_t launch_ens_final(const PivParams& p, const EnsRescueArgs& a, const double* totals, float* u, float* v, hipStream_t s);
hipError_t launch_peaks_from_planes(const float* planes, uint32_t n_planes, int wy, int wx, int border_mode,
                                    float* u, float* v, hipStream_t s);
// "stack" signal mode: keep[w] = fraction of non-zero (or positive) samples of window position w over ALL frames >= thr
hipError_t launch_window_signal(const void* frames, int dtype, int64_t T, const PivParams& p, float thr, uint8_t* keep, hipStream_t s);
// mean[w][o] = count[w] < min_count ? NaN : sum[w][o] / count[w]   (pyorc/velocimetry/ffpiv.py:280-282)
// LSPIV_WALK as an integer (0 per-pair kernels, 1 default walking kernels, n > 1 forced segment length): the value set
// through lspiv_set_option("walk", v) if any, else the environment variable read at every launch, else 1
int walk_setting();
// Segments of the time-walking kernels.  A job walks one window through a run of consecutive frame pairs and shares
// every frame's spectrum between the two pairs it belongs to; which frames share transforms depends on where a run
// starts, so the runs ("segments") are ANCHORED: they start at the absolute pair indices k * kWalkAnchor of the caller's
// stack (PivParams::pair_offset + local index), whatever the chunk.  Two chunkings whose boundaries are multiples of the
// anchor length (lspiv_chunk_alignment; the Python planner only makes such chunks) then run the very same jobs and give
// the same bits, like the reference, which computes every window independently (pyorc/velocimetry/ffpiv.py:140,399-442).
// A chunk that starts off-anchor gets a shorter first segment (correct, but its first pairs differ in the last bit from
// an aligned run).  25: a job of L pairs lasts L / 2 + 1 iterations (its first iteration yields one plane only) and the
// n_win * n_seg jobs run in rounds of as many lane groups as the chip holds; over chunks of 20 ... 4000 pairs and grids
// of 2.5 k ... 32 k windows, rounds x iterations of L = 25 stays within 2.5 % of the best per-chunk choice at 1000 pairs
// and within 6 % at 200 (59 ... 63, the best length for 1000-pair chunks, loses 15 - 40 % at 200).
constexpr uint32_t kWalkAnchor = 25;
// Round 5: LONG anchors for grids with enough windows.  What a longer segment saves is per-segment overhead -- the first iteration of a
// segment yields one plane instead of two (13 iterations per 25 pairs, 38 per 75: - 2.6 %), and in ensemble mode every segment flushes
// and later merges a partial sum per window.  What it costs: (1) parallelism per segment -- a segment is n_win jobs, and with fewer of
// them than the chip has lane groups the tail of every round of jobs idles for a longer job --, so the anchor is a function of the
// window grid alone (NOT of the chunk: results must not depend on the chunking): long when the grid has at least as many windows as
// the chip has lane groups of that window family (nominal MI355X: 256 CUs x 4 SIMDs x waves x groups per wave), 25 otherwise;
// (2) bytes -- the jobs of a round drift apart over a long segment, overlapping windows are no longer at the same frame and share
// fewer rows in L2 (HBM fetch per 1000 pairs at 1080p, anchors 25 / 51 / 75 / 125: 32 x 32 2.31 / 2.96 / 3.12 / 3.13 GB, 64 x 64 2.81 /
// 3.46 / 4.06 / 4.44 GB); the kernels are VALU-bound and do not wait for them.  1080p, 1000 pairs, one box (tools/gpu_round5_k.sh),
// anchors 25 / 51 / 75 / 125: per-timestep 32 x 32 166.5 / 168.7 / 167.9 / 163.7 k pairs/s, 64 x 64 35.3 / 36.3 / 36.7 / 36.3 k,
// ensemble 64 x 64 33.2 / 34.3 / 34.7 / 34.4 k, 32 x 32 165.7 / 174.1 / 174.9 / 175.3 k: 75 takes what there is at three quarters of
// 125's extra bytes.  It needs the XCD partition BY WINDOWS (piv_fft_impl.h, walk_job): with whole segments per XCD unequal segments
// run XCDs dry (300 pairs at 125: 26 k instead of 33 k).  Chunks must be cut on multiples of lspiv_chunk_alignment_grid() -- 75 for
// such grids -- to reproduce one call bit for bit.
constexpr uint32_t kWalkAnchorLong = 75;
inline uint32_t walk_long_min_windows(int n) {          // lane groups on the chip, per window family
  return n <= 16 ? 256u * 4u * 3u * 4u : n <= 32 ? 256u * 4u * 3u * 2u : 256u * 4u * 2u;   // 12 288 / 6 144 / 2 048
}
inline uint32_t walk_anchor(int n, uint32_t n_win) { return n_win >= walk_long_min_windows(n) ? kWalkAnchorLong : kWalkAnchor; }
struct WalkSegments { uint32_t seg_len, seg_first, n_seg; };
inline WalkSegments walk_segments(uint32_t n_pairs, int64_t pair_offset, uint32_t seg_len) {
  WalkSegments w;
  w.seg_len = seg_len < 1 ? 1 : seg_len;
  const uint32_t head = (uint32_t)(pair_offset % (int64_t)w.seg_len);
  w.seg_first = std::min<uint32_t>(w.seg_len - head, n_pairs);
  w.n_seg = 1 + (n_pairs - w.seg_first + w.seg_len - 1) / w.seg_len;
  return w;
}
// concurrent lane groups of a kernel that runs `waves_per_simd` waves with `groups` jobs per wave (CU count queried once)
uint32_t job_slots(int waves_per_simd, int groups);
// lane_major_n: 0 = the partial sums are fft-shifted row-major planes like corr_sum; N = the N x N slots of the 64 x 64 walking
// ensemble kernel, element (row y, column x) at slot[(x / 4) * 4 N + y * 4 + x % 4], un-shifted (piv_fft_impl.h, slot_accumulate);
// split_halves: the first and the second half of every slot live in two arrays, all first halves, then all second halves (kEnsSplitHalves)
hipError_t launch_ensemble_merge(const float* part_sum, const float* part_cnt, uint32_t n_seg, uint32_t n_win, int plane_elems,
                                 float* corr_sum, float* corr_count, hipStream_t s, int lane_major_n = 0, bool split_halves = false);
hipError_t launch_ensemble_mean(const float* sum, const float* count, float min_count, uint32_t n_win,
                                int plane_elems, float* mean, hipStream_t s);
// orthoprojection gather (project.hip) and int16 result packing
hipError_t launch_project(const void* frames, int dtype, int64_t src_elems, int n_frames, const int* nn_src,
                          const int* grp_of, const int* grp_off, const int* grp_src, float* out, int n_out,
                          hipStream_t s);
// uint8 frames through the quad-window plan (two 8-byte source windows per four output cells, project.hip)
hipError_t launch_division_check(int* d_mismatches, hipStream_t s);   // test hook of project_mix_kernel's quotient
// uint8 frames through a plan WITH group means: per quad NW (2 or 4) 8-byte windows, per cell one byte mask per window + the count
hipError_t launch_project_mix(const uint8_t* frames, int64_t src_elems, int n_frames, int nw, const int* qwin, const uint32_t* qcell,
                              const int* slow_q, int n_slow, const int* nn_src, const int* grp_of, const int* grp_off,
                              const int* grp_src, float* out, int n_out, hipStream_t s);
hipError_t launch_project_win(const uint8_t* frames, int64_t src_elems, int n_frames, const int* qlo1, const int* qlo2,
                              const uint32_t* qdesc, const int* slow_q, int n_slow, const int* nn_src, const int* grp_of,
                              const int* grp_off, const int* grp_src, float* out, int n_out, hipStream_t s);
// project_cv: one fixed-point bilinear remap (cv2.remap INTER_LINEAR, BORDER_CONSTANT 0); dtype 0 (uint8) or 1 (float32), output of the same type
hipError_t launch_remap(const void* frames, int dtype, int64_t src_elems, int Hs, int Ws, int n_frames, const int* mx, const int* my,
                        const uint16_t* mf, void* out, int n_out, hipStream_t s);
// uint8 frames through the quad plan (two 8-byte source windows per four destination pixels; project.hip)
hipError_t launch_remap_win(const uint8_t* frames, int64_t src_elems, int Hs, int Ws, int n_frames, const int* qbase, const uint64_t* qdesc,
                            const int* slow_q, int n_slow, const int* mx, const int* my, const uint16_t* mf, uint8_t* out, int n_out,
                            hipStream_t s);
// x[i] = -x[i]: the "v_sign" option (a reading of ffpiv nothing in the reference decides), applied after the kernels
hipError_t launch_negate(float* x, int64_t n, hipStream_t s);
hipError_t launch_pack_int16(const float* in, int64_t n, float scale, int fill, int16_t* out, hipStream_t s);
// element-wise pre-processing filters (filters.hip)
hipError_t launch_time_diff(const void* frames, int dtype, int64_t frame_elems, int64_t n_frames, float thres, int use_abs,
                            float* out, hipStream_t s);
hipError_t launch_minmax(const float* in, int64_t n, float lo, float hi, float* out, hipStream_t s);
// Frames.reduce_rolling on uint8 frames: trailing rolling mean removed, >= 0, per-frame
// stretch to uint8; scratch of reduce_rolling_scratch_bytes()
size_t reduce_rolling_scratch_bytes(int64_t frame_elems, int n_frames);
hipError_t launch_reduce_rolling(const uint8_t* frames, int64_t frame_elems, int n_frames, int samples, double* scratch,
                                 uint8_t* out, hipStream_t s);
// Frames.range: (max - min over time) in the frames' own dtype, out (frame_elems) of that dtype
hipError_t launch_time_range(const void* frames, int dtype, int64_t frame_elems, int64_t n_frames, void* out, hipStream_t s);
// d_part: normalize_part_bytes() of scratch for the per-wave (min, max) pairs of the space-major passes (nullptr: frame-major kernels)
size_t normalize_part_bytes(int64_t frame_elems, int n_frames);
hipError_t launch_normalize(const uint8_t* frames, int64_t frame_elems, int n_frames, int interval, float* d_mean,
                            int* d_mn, int* d_mx, float* d_part, uint8_t* out, hipStream_t s);
// the two halves of launch_normalize: the float32 mean of frames [::interval], and the per-frame stretch against a given mean
hipError_t launch_sample_mean(const uint8_t* frames, int64_t frame_elems, int n_frames, int interval, float* d_mean, hipStream_t s);
hipError_t launch_normalize_apply(const uint8_t* frames, int64_t frame_elems, int n_frames, const float* d_mean, int* d_mn,
                                  int* d_mx, float* d_part, uint8_t* out, hipStream_t s);
// Gaussian blur (ksize_b == 0) or band filter blur(ksize_b) - blur(ksize_a); odd sizes 1..31
hipError_t launch_blur(const void* frames, int dtype, int n_frames, int H, int W, int ksize_a, int ksize_b, float* out,
                       hipStream_t s);
// post-PIV masks on the [v_x | v_y | corr | s2n] block (masks.hip); kinds / params as in include/lspiv.h
hipError_t launch_mask(const float* f, int64_t T, int R, int C, int kind, const double* p, uint8_t* mask, hipStream_t s);
hipError_t launch_mask_apply(float* f, int64_t T, int64_t n, const uint8_t* mask, int mask_has_time, hipStream_t s);
hipError_t launch_time_mean(const float* f, int64_t T, int64_t n, float* out, hipStream_t s);
hipError_t launch_window_replace(const float* in, int64_t planes, int R, int C, int x_min, int x_max, int y_min, int y_max,
                                 float* out, hipStream_t s);
hipError_t launch_scale_velocity(float* f, int64_t T, int64_t n, float res_x, float res_y, const double* d_dt, hipStream_t s);
// test hook: `count` independent length-n register FFTs (fft_debug.hip); interleaved re/im, n = 8, 16, 32, 64 or a prime-factor length
hipError_t launch_fft_debug(int n, bool inverse, const float* in, float* out, int count, hipStream_t s);
// synthetic particle-image stack (bench / test utility)
hipError_t launch_synth_particles(uint8_t* d_frames, int64_t T, int H, int W, uint64_t seed, float density,
                                  hipStream_t s);

}  // namespace lspiv
