// Float64 rescue pass of the PIV kernels (gfx950).
//
// The fused kernels compute the correlation planes in float32: every sample carries absolute noise of ~1e-7 .. 4e-7 of the
// plane maximum.  The reference (ffpiv behind pyorc/velocimetry/ffpiv.py:450-471) fits EVERY plane, also those on which the
// 3-point log-Gaussian fit amplifies such noise beyond the 1e-4 parity gate -- a neighbour of the peak that is exactly zero
// in exact arithmetic (clipped), a flat ridge, two samples that tie for the maximum.  The kernels' epilogues recognise those
// windows (common.h, peak_cond) and append them to two device lists; this pass re-evaluates them from the FRAMES in float64
// and overwrites u and v (and with them the NaN decision of a border peak):
//   "fit" records  (g, pos, [pos2]): the arg-max is trusted -- or is one of exactly two candidates, settled first by their two
//                   float64 sums --; the five samples the fit reads are five circular cross-correlation
//                   sums  c[k] = (1 / n) sum_m a'[m] b'[m + k],  a' = max((a - mean a) / std a, 0)  -- 5 n multiply-adds, one wave;
//   "amb" records  (g):          the whole plane, n^2 multiply-adds by one block (normalised windows in LDS up to 64 x 64,
//                   zero samples of a' skipped -- a wave-uniform branch --, from L2 above), then first arg-max in fft-shifted
//                   row-major order and the same five-sample fit.
// Same semantics as the float64 CPU restatement the tests check against (mean as x0 + mean(x - x0), population std,
// clip [0, 1], eps 1e-7, zero denominator -> 0, border -> NaN / option); the summation order differs (direct sums here, FFT
// there), i.e. agreement to ~1e-13, not bit identity.  Exact float64 ties (e.g. two lags of a sparse integer-valued window
// with the very same products) stay a matter of rounding in ANY implementation.
//
// A fixed grid strides over both lists; its last block moves the counts to the statistics fields and zeroes the counters, so
// a pass costs one extra launch and no memset.
#include "common.h"

namespace lspiv {

namespace {

constexpr int RBLOCK = 256;
constexpr int RESCUE_LDS_SAMPLES = 4096;   // both normalised windows as doubles: 64 KB
constexpr int RESCUE_GENERIC_MAX = 16384;  // whole-plane re-evaluation of windows that do not fit LDS: up to 128 x 128 samples

// float64 sum over the wave, every lane gets the total: the DPP / swizzle / permlane32 steps of the float32 reductions
// (common.h) on the two halves of the double -- __shfl_xor on a double is two ds_bpermute plus a wait per step
template <int CTRL>
__device__ __forceinline__ double dpp_d(double x) {
  const unsigned long long b = __builtin_bit_cast(unsigned long long, x);
  const unsigned lo = (unsigned)dpp_i<CTRL>((int)(unsigned)b), hi = (unsigned)dpp_i<CTRL>((int)(unsigned)(b >> 32));
  return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ double wave_sum_d(double x) {
  x += dpp_d<DPP_XOR1>(x);
  x += dpp_d<DPP_XOR2>(x);
  x += dpp_d<DPP_HALF_MIRROR>(x);
  x += dpp_d<DPP_MIRROR>(x);
  {
    const unsigned long long b = __builtin_bit_cast(unsigned long long, x);
    const unsigned lo = (unsigned)swz16_i((int)(unsigned)b), hi = (unsigned)swz16_i((int)(unsigned)(b >> 32));
    x += __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
  }
  {
    const unsigned long long b = __builtin_bit_cast(unsigned long long, x);
    unsigned lo_a = (unsigned)b, lo_b = lo_a, hi_a = (unsigned)(b >> 32), hi_b = hi_a;
    permlane32_swap(lo_a, lo_b);
    permlane32_swap(hi_a, hi_b);
    x = __builtin_bit_cast(double, ((unsigned long long)hi_a << 32) | lo_a) + __builtin_bit_cast(double, ((unsigned long long)hi_b << 32) | lo_b);
  }
  return x;
}

// mean / population std of both windows of a pair in float64, by one wave (every lane gets the totals); the two windows go
// through the same loops and the loops are unrolled so that a batch of independent loads is in flight per wait
// lane -> its samples e = lane, lane + 64, ...: (y, x) advanced without a division per sample
struct LaneWalk {
  int qy, qx, wx;   // 64 = qy * wx + qx
  __device__ LaneWalk(int wx_) : qy(64 / wx_), qx(64 - (64 / wx_) * wx_), wx(wx_) {}
  __device__ __forceinline__ void start(int lane, int& y, int& x) const { y = lane / wx; x = lane - y * wx; }
  __device__ __forceinline__ void next(int& y, int& x) const {
    y += qy; x += qx;
    if (x >= wx) { x -= wx; ++y; }
  }
};

template <typename T, bool STAGE = false>
__device__ __forceinline__ void window_stats_wave2(const T* A, const T* B, int W, int wy, int wx, int lane, double& mean_a,
                                                   double& sd_a, double& mean_b, double& sd_b, T* la = nullptr, T* lb = nullptr) {
  const int n = wy * wx;
  const LaneWalk lw(wx);
  // one pass over shifted samples d = x - x[0]: mean = x0 + S1 / n, variance = (S2 - S1^2 / n) / n.  With the shift the
  // cancellation costs ~1e-16 (mean - x0)^2 / variance relative -- 1e-13 at worst for 8-bit imagery -- and a constant window
  // still has exactly zero variance (all d are 0).
  const double x0a = (double)A[0], x0b = (double)B[0];
  double sa = 0.0, sb = 0.0, qa = 0.0, qb = 0.0;
  int y, x;
  lw.start(lane, y, x);
#pragma unroll 8
  for (int e = lane; e < n; e += 64, lw.next(y, x)) {
    const int64_t off = (int64_t)y * W + x;
    const T ra = A[off], rb = B[off];
    if constexpr (STAGE) { la[e] = ra; lb[e] = rb; }   // dense copy of the window (pitch wx) in the wave's LDS slice
    const double da = (double)ra - x0a, db = (double)rb - x0b;
    sa += da; qa = fma(da, da, qa);
    sb += db; qb = fma(db, db, qb);
  }
  sa = wave_sum_d(sa); sb = wave_sum_d(sb); qa = wave_sum_d(qa); qb = wave_sum_d(qb);
  const double inv_n = 1.0 / (double)n;
  mean_a = x0a + sa * inv_n;
  mean_b = x0b + sb * inv_n;
  const double va = (qa - sa * sa * inv_n) * inv_n, vb = (qb - sb * sb * inv_n) * inv_n;
  sd_a = va > 0.0 ? sqrt(va) : 0.0;
  sd_b = vb > 0.0 ? sqrt(vb) : 0.0;
}

// max((x - mean) / std, 0) with the reciprocal of std formed once per window (1 ulp of float64 from the division)
// (inv_sd < 0 encodes the "norm_clip" = 0 option: |inv_sd| is the factor and the negative lobes stay)
__device__ __forceinline__ double norm_clip(double x, double mean, double inv_sd) {
  const double d = (x - mean) * fabs(inv_sd);
  return (d > 0.0 || inv_sd < 0.0) ? d : 0.0;
}

// un-shifted lag of a shifted plane coordinate
__device__ __forceinline__ int unshift(int ip, int c, int w) { return ip - c < 0 ? ip - c + w : ip - c; }

// which of two shifted plane positions holds the larger float64 correlation (ties: the smaller row-major index, np.argmax)
template <typename PA>
__device__ __forceinline__ uint32_t choose_wave(const PivParams& p, PA A, PA B, int pitch, double mean_a, double inv_a, double mean_b,
                                                double inv_b, uint32_t pos1, uint32_t pos2, int lane) {
  const int wy = p.wy, wx = p.wx, n = wy * wx, cy = wy / 2, cx = wx / 2;
  const int ky1 = unshift((int)(pos1 >> 16), cy, wy), kx1 = unshift((int)(pos1 & 0xffffu), cx, wx);
  const int ky2 = unshift((int)(pos2 >> 16), cy, wy), kx2 = unshift((int)(pos2 & 0xffffu), cx, wx);
  const LaneWalk lw(wx);
  double acc1 = 0.0, acc2 = 0.0;
  int y, x;
  lw.start(lane, y, x);
#pragma unroll 4
  for (int e = lane; e < n; e += 64, lw.next(y, x)) {
    const double av = norm_clip((double)A[y * pitch + x], mean_a, inv_a);
    int y1 = y + ky1; y1 = y1 >= wy ? y1 - wy : y1;
    int x1 = x + kx1; x1 = x1 >= wx ? x1 - wx : x1;
    int y2 = y + ky2; y2 = y2 >= wy ? y2 - wy : y2;
    int x2 = x + kx2; x2 = x2 >= wx ? x2 - wx : x2;
    acc1 += av * norm_clip((double)B[y1 * pitch + x1], mean_b, inv_b);
    acc2 += av * norm_clip((double)B[y2 * pitch + x2], mean_b, inv_b);
  }
  const double c1 = wave_sum_d(acc1), c2 = wave_sum_d(acc2);   // same scale and clip for both: compare the sums
  const uint32_t o1 = (pos1 >> 16) * (uint32_t)wx + (pos1 & 0xffffu), o2 = (pos2 >> 16) * (uint32_t)wx + (pos2 & 0xffffu);
  return (c2 > c1 || (c2 == c1 && o2 < o1)) ? pos2 : pos1;
}

// the five circular cross-correlation sums around shifted position (ip, jp), (1 / n) sum_m a'[m] b'[m + k] clipped to [0, 1],
// k = the peak and its four neighbours, all in float64; every lane gets the five values
// (A, B: the two windows, in global memory (pitch = frame width) or staged in LDS (pitch = window width))
struct Lag5 { double c0, cu, cd, cl, cr; };   // centre, row above, row below, column left, column right
template <typename PA>
__device__ __forceinline__ Lag5 lag5_wave(const PivParams& p, PA A, PA B, int pitch, double mean_a, double inv_a, double mean_b,
                                          double inv_b, int ip, int jp, int lane) {
  const int wy = p.wy, wx = p.wx, n = wy * wx, cy = wy / 2, cx = wx / 2;
  // un-shifted lags of the peak and its four neighbours
  const int ky0 = unshift(ip, cy, wy), kx0 = unshift(jp, cx, wx);
  const int kym = ky0 == 0 ? wy - 1 : ky0 - 1, kyp = ky0 == wy - 1 ? 0 : ky0 + 1;
  const int kxm = kx0 == 0 ? wx - 1 : kx0 - 1, kxp = kx0 == wx - 1 ? 0 : kx0 + 1;
  const LaneWalk lw(wx);
  double acc0 = 0.0, accu = 0.0, accd = 0.0, accl = 0.0, accr = 0.0;
  int y, x;
  lw.start(lane, y, x);
#pragma unroll 4
  for (int e = lane; e < n; e += 64, lw.next(y, x)) {
    const double av = norm_clip((double)A[y * pitch + x], mean_a, inv_a);
    int y0 = y + ky0; y0 = y0 >= wy ? y0 - wy : y0;
    int ym = y + kym; ym = ym >= wy ? ym - wy : ym;
    int yp = y + kyp; yp = yp >= wy ? yp - wy : yp;
    int x0 = x + kx0; x0 = x0 >= wx ? x0 - wx : x0;
    int xm = x + kxm; xm = xm >= wx ? xm - wx : xm;
    int xp = x + kxp; xp = xp >= wx ? xp - wx : xp;
    acc0 += av * norm_clip((double)B[y0 * pitch + x0], mean_b, inv_b);
    accu += av * norm_clip((double)B[ym * pitch + x0], mean_b, inv_b);
    accd += av * norm_clip((double)B[yp * pitch + x0], mean_b, inv_b);
    accl += av * norm_clip((double)B[y0 * pitch + xm], mean_b, inv_b);
    accr += av * norm_clip((double)B[y0 * pitch + xp], mean_b, inv_b);
  }
  const double inv_n = 1.0 / (double)n;
  auto clip01 = [](double c) { return c < 0.0 ? 0.0 : (c > 1.0 ? 1.0 : c); };
  Lag5 o;
  o.c0 = clip01(wave_sum_d(acc0) * inv_n); o.cu = clip01(wave_sum_d(accu) * inv_n); o.cd = clip01(wave_sum_d(accd) * inv_n);
  o.cl = clip01(wave_sum_d(accl) * inv_n); o.cr = clip01(wave_sum_d(accr) * inv_n);
  return o;
}
// 3-point log-Gaussian fit of five float64 samples around shifted position (ip, jp) (eps 1e-7, zero denominator -> 0)
__device__ __forceinline__ void fit5_d(const Lag5& c, int ip, int jp, int cy, int cx, float& u, float& v) {
  const double eps = 1e-7;
  const double l0 = log(c.c0 + eps), lu = log(c.cu + eps), ld = log(c.cd + eps), ll = log(c.cl + eps), lr = log(c.cr + eps);
  const double den1 = 2 * lu - 4 * l0 + 2 * ld, den2 = 2 * ll - 4 * l0 + 2 * lr;
  const double di = den1 != 0.0 ? (lu - ld) / den1 : 0.0;
  const double dj = den2 != 0.0 ? (ll - lr) / den2 : 0.0;
  v = (float)((double)ip + di - (double)cy);
  u = (float)((double)jp + dj - (double)cx);
}

// the five-sample fit at shifted position (ip, jp), all in float64; one wave, lane 0 stores
template <typename PA>
__device__ __forceinline__ void fit_wave(const PivParams& p, PA A, PA B, int pitch, double mean_a, double inv_a, double mean_b,
                                         double inv_b, uint32_t g, int ip, int jp, int lane) {
  const int wy = p.wy, wx = p.wx, cy = wy / 2, cx = wx / 2;
  if (ip <= 0 || ip >= wy - 1 || jp <= 0 || jp >= wx - 1) {   // border peak: no fit (A5)
    if (lane == 0) {
      float u, v;
      border_result(p.border_mode, jp - cx, ip - cy, u, v);
      p.u[g] = u; p.v[g] = v;
    }
    return;
  }
  const Lag5 c = lag5_wave(p, A, B, pitch, mean_a, inv_a, mean_b, inv_b, ip, jp, lane);
  // the five logarithms side by side on five lanes instead of one after the other on lane 0
  const double eps = 1e-7;
  const double mine = log((lane == 0 ? c.c0 : lane == 1 ? c.cu : lane == 2 ? c.cd : lane == 3 ? c.cl : c.cr) + eps);
  const double l0 = __shfl(mine, 0, 64), lu = __shfl(mine, 1, 64), ld = __shfl(mine, 2, 64), ll = __shfl(mine, 3, 64), lr = __shfl(mine, 4, 64);
  if (lane == 0) {
    const double den1 = 2 * lu - 4 * l0 + 2 * ld, den2 = 2 * ll - 4 * l0 + 2 * lr;
    const double di = den1 != 0.0 ? (lu - ld) / den1 : 0.0;
    const double dj = den2 != 0.0 ? (ll - lr) / den2 : 0.0;
    p.v[g] = (float)((double)ip + di - (double)cy);
    p.u[g] = (float)((double)jp + dj - (double)cx);
  }
}

template <typename T>
__device__ __forceinline__ const T* window_base(const PivParams& p, uint32_t g) {
  const uint32_t pair = g / p.n_win, win = g - pair * p.n_win;
  const uint32_t wrow = win / (uint32_t)p.n_cols, wcol = win - wrow * (uint32_t)p.n_cols;
  return static_cast<const T*>(p.frames) + ((int64_t)pair * p.H + (int64_t)wrow * p.sy) * p.W + (int64_t)wcol * p.sx;
}

// first arg-max merge in shifted row-major order: larger value wins, equal values -> smaller index
__device__ __forceinline__ void amax_merge_d(double& v, int& idx, double pv, int pidx) {
  const bool take = (pv > v) || (pv == v && pidx < idx);
  v = take ? pv : v;
  idx = take ? pidx : idx;
}

// ---- "fit" records: one wave each, statically strided over the grid (every lane of a wave holds the same index: no
// cross-lane hand-over of a work counter, nothing the compiler has to prove uniform -- a dynamic counter handed round with
// readfirstlane was structurised into a loop that never left its first record) -------------------------------------------
// Staging: a record's two windows are read ONCE from L2 into the wave's slice of LDS (in the frames' own sample type), the
// statistics ride on that pass, and the six samples per element of the fit come out of LDS -- read straight from the frames the
// fit issues 6 n single-sample loads through the texture path, which is what bounded this kernel (140 -> ~50 us per 32 k records).
// Windows whose two copies exceed the slice (float32 64 x 64 and up) take the direct path.
constexpr int FIT_LDS_PER_WAVE = 8192;

template <typename T>
__global__ __launch_bounds__(RBLOCK) void piv_rescue_fit_kernel(PivParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char fsm[];
  typedef const T __attribute__((address_space(3))) * LdsPtr;
  const RescueHdr* hdr = p.rescue_hdr;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wy = p.wy, wx = p.wx, n = wy * wx;
  const bool staged = (size_t)2 * n * sizeof(T) <= (size_t)FIT_LDS_PER_WAVE;
  T* la = reinterpret_cast<T*>(fsm + (size_t)wave * FIT_LDS_PER_WAVE);
  T* lb = la + n;
  const uint32_t n_fit = min(hdr->n_fit, p.rescue_cap_fit);
  const uint32_t n_waves = gridDim.x * (RBLOCK / 64);
  for (uint32_t i = blockIdx.x * (RBLOCK / 64) + (uint32_t)wave; i < n_fit; i += n_waves) {
    const uint4 rec = p.rescue_fit[i];
    const uint32_t g = rec.x;
    if (g >= p.n_tiles) continue;   // a record of another launch (two host threads interleaving on one stream): never write out of bounds
    const T* A = window_base<T>(p, g);
    const T* B = A + p.frame_elems;
    double mean_a, sd_a, mean_b, sd_b;
    if (staged) window_stats_wave2<T, true>(A, B, p.W, wy, wx, lane, mean_a, sd_a, mean_b, sd_b, la, lb);
    else window_stats_wave2<T, false>(A, B, p.W, wy, wx, lane, mean_a, sd_a, mean_b, sd_b, nullptr, nullptr);
    // (a zero-variance window is NaN already and is never listed)
    if (sd_a != 0.0 && sd_b != 0.0) {
      const double sg = p.norm_clip ? (double)p.std_gain : -(double)p.std_gain;   // options "std_ddof" / "norm_clip"
      const double inv_a = sg / sd_a, inv_b = sg / sd_b;
      uint32_t pos = rec.y;
      if (staged) {
        __builtin_amdgcn_wave_barrier();   // the wave's own LDS writes above are in order with the reads below; this pins the compiler
        const LdsPtr SA = (LdsPtr)la, SB = (LdsPtr)lb;
        if (rec.z != 0xffffffffu) pos = choose_wave(p, SA, SB, wx, mean_a, inv_a, mean_b, inv_b, rec.y, rec.z, lane);   // two candidates
        fit_wave(p, SA, SB, wx, mean_a, inv_a, mean_b, inv_b, g, (int)(pos >> 16), (int)(pos & 0xffffu), lane);
        __builtin_amdgcn_wave_barrier();
      } else {
        if (rec.z != 0xffffffffu) pos = choose_wave(p, A, B, p.W, mean_a, inv_a, mean_b, inv_b, rec.y, rec.z, lane);
        fit_wave(p, A, B, p.W, mean_a, inv_a, mean_b, inv_b, g, (int)(pos >> 16), (int)(pos & 0xffffu), lane);
      }
    }
  }
}

// ---- "amb" records: one block each.  Fast path (wx a multiple of 4, both windows fit the LDS budget): the normalised
// windows sit in LDS as doubles, b with every row doubled (b2[y][x] = b'[y][x mod wx], 2 wx entries) so that a lag never
// wraps inside a row; a thread owns strips of FOUR consecutive lags kx .. kx + 3 of one ky; the rows of a' are compacted to
// their non-zero samples, so a step is two broadcast reads (sample, its x) and four reads of b2 for four float64 FMAs. ------
constexpr int AMB_R = 4;

template <typename T>
__global__ __launch_bounds__(RBLOCK) void piv_rescue_amb_kernel(PivParams p) {
  extern __shared__ __attribute__((aligned(16))) double dsm[];
  __shared__ double red_v[RBLOCK / 64];
  __shared__ int red_i[RBLOCK / 64];
  RescueHdr* hdr = p.rescue_hdr;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wy = p.wy, wx = p.wx, n = wy * wx, cy = wy / 2, cx = wx / 2;
  const uint32_t n_amb = min(hdr->n_amb, p.rescue_cap_amb);
  const bool fast = n <= RESCUE_LDS_SAMPLES && (wx % AMB_R) == 0;
  const int pitch = 2 * wx;
  double* la = dsm;
  double* lb2 = dsm + n;
  int* lx = reinterpret_cast<int*>(dsm + 3 * n);   // x of the compacted samples, row by row
  int* lcnt = lx + n;                              // non-zero samples per row
  for (uint32_t i = blockIdx.x; i < n_amb; i += gridDim.x) {
    __syncthreads();   // the previous record's LDS windows and reduction slots are free
    const uint32_t g = min(p.rescue_amb[i], p.n_tiles - 1);   // (clamped, not skipped: the barriers below stay uniform)
    const bool g_ok = p.rescue_amb[i] < p.n_tiles;
    const T* A = window_base<T>(p, g);
    const T* B = A + p.frame_elems;
    double mean_a, sd_a, mean_b, sd_b;
    window_stats_wave2<T>(A, B, p.W, wy, wx, lane, mean_a, sd_a, mean_b, sd_b);   // every wave computes the same totals
    const bool dead = sd_a == 0.0 || sd_b == 0.0;   // never listed; kept out of the control flow around the barriers below
    const double sg = p.norm_clip ? (double)p.std_gain : -(double)p.std_gain;   // options "std_ddof" / "norm_clip"
    const double inv_a = dead ? 1.0 : sg / sd_a, inv_b = dead ? 1.0 : sg / sd_b;
    double best = -1.0;
    int bi = 0x7fffffff;
    if (fast) {
      for (int e = threadIdx.x; e < n; e += RBLOCK) {
        const int y = e / wx, x = e - y * wx;
        la[e] = norm_clip((double)A[(int64_t)y * p.W + x], mean_a, inv_a);
        const double bv = norm_clip((double)B[(int64_t)y * p.W + x], mean_b, inv_b);
        lb2[y * pitch + x] = bv;
        lb2[y * pitch + wx + x] = bv;
      }
      __syncthreads();
      // The clip at zero leaves most samples of a' exactly zero on particle imagery (and nearly all of them in the sparse
      // windows that end up here): compact each row to its non-zero samples, in place and in x order (one thread per row:
      // a fixed order, so the sums are reproducible), and let the lag loops walk those only.
      for (int r = threadIdx.x; r < wy; r += RBLOCK) {   // (tall narrow windows have more rows than the block has threads: 512 x 8)
        double* ar = la + r * wx;
        int* xr = lx + r * wx;
        int k = 0;
        for (int x = 0; x < wx; ++x) {
          const double av = ar[x];
          if (av != 0.0) { ar[k] = av; xr[k] = x; ++k; }
        }
        lcnt[r] = k;
      }
      __syncthreads();
      const int strips_per_row = wx / AMB_R, strips = wy * strips_per_row;
      for (int sidx = threadIdx.x; sidx < strips; sidx += RBLOCK) {
        const int ky = sidx / strips_per_row, kx0 = (sidx - ky * strips_per_row) * AMB_R;
        double acc[AMB_R];
#pragma unroll
        for (int r = 0; r < AMB_R; ++r) acc[r] = 0.0;
        int yb = ky;
        for (int y = 0; y < wy; ++y) {
          const double* ar = la + y * wx;     // the same words for every thread: broadcast reads, uniform trip count
          const int* xr = lx + y * wx;
          const double* br = lb2 + yb * pitch + kx0;
          const int cnt = lcnt[y];
#pragma unroll 2
          for (int k = 0; k < cnt; ++k) {
            const double av = ar[k];
            const double* bq = br + xr[k];
#pragma unroll
            for (int r = 0; r < AMB_R; ++r) acc[r] = fma(av, bq[r], acc[r]);
          }
          yb = yb + 1 == wy ? 0 : yb + 1;
        }
        // shifted plane index o = ip * wx + jp of lag (ky, kx): ascending kx inside a strip may wrap jp once, so merge by index
        const int ipo = ky + cy >= wy ? ky + cy - wy : ky + cy;
#pragma unroll
        for (int r = 0; r < AMB_R; ++r) {
          const int kx = kx0 + r;
          const int jpo = kx + cx >= wx ? kx + cx - wx : kx + cx;
          double c = acc[r] / (double)n;
          c = c < 0.0 ? 0.0 : (c > 1.0 ? 1.0 : c);
          amax_merge_d(best, bi, c, ipo * wx + jpo);
        }
      }
    } else if (n <= RESCUE_GENERIC_MAX) {
      // any other shape: one lag at a time, samples normalised on the fly from L2 (functional path: windows above 64 px,
      // widths that are no multiple of 4).  n^2 multiply-adds by ONE block: bounded at 128 x 128 samples (2.7e8, milliseconds);
      // a three-way near-tie of the maximum in a larger window keeps its float32 result (include/lspiv.h, "rescue")
      for (int o = threadIdx.x; o < n; o += RBLOCK) {
        const int ipo = o / wx, jpo = o - ipo * wx;
        const int ky = ipo - cy < 0 ? ipo - cy + wy : ipo - cy, kx = jpo - cx < 0 ? jpo - cx + wx : jpo - cx;
        double acc = 0.0;
        int yb = ky;
        for (int y = 0; y < wy; ++y) {
          int xb = kx;
          for (int x = 0; x < wx; ++x) {
            const double av = norm_clip((double)A[(int64_t)y * p.W + x], mean_a, inv_a);
            if (av != 0.0) acc += av * norm_clip((double)B[(int64_t)yb * p.W + xb], mean_b, inv_b);
            xb = xb + 1 == wx ? 0 : xb + 1;
          }
          yb = yb + 1 == wy ? 0 : yb + 1;
        }
        double c = acc / (double)n;
        c = c < 0.0 ? 0.0 : (c > 1.0 ? 1.0 : c);
        amax_merge_d(best, bi, c, o);
      }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) amax_merge_d(best, bi, __shfl_xor(best, o, 64), __shfl_xor(bi, o, 64));
    if (lane == 0) { red_v[wave] = best; red_i[wave] = bi; }
    __syncthreads();
    best = red_v[0]; bi = red_i[0];
#pragma unroll
    for (int k = 1; k < RBLOCK / 64; ++k) amax_merge_d(best, bi, red_v[k], red_i[k]);
    if (wave == 0 && !dead && g_ok && (fast || n <= RESCUE_GENERIC_MAX)) {
      const int ip = bi / wx, jp = bi - ip * wx;
      fit_wave(p, A, B, p.W, mean_a, inv_a, mean_b, inv_b, g, ip, jp, lane);
    }
  }

  // ---- the last block to finish publishes the counts and clears the counters for the next pass (the fit kernel of this
  // pass ran before this one on the same stream) -----------------------------------------------------------------------------
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence();
    const uint32_t d = atomicAdd(&hdr->done_blocks, 1u);
    if (d == gridDim.x - 1) {
      hdr->last_fit = hdr->n_fit;
      hdr->last_amb = hdr->n_amb;
      hdr->total_fit += hdr->n_fit;
      hdr->total_amb += hdr->n_amb;
      hdr->total_windows += p.n_tiles;
      hdr->n_fit = 0; hdr->n_amb = 0; hdr->done_blocks = 0;
      __threadfence();
    }
  }
}

// ======== ensemble mode: the final fit of the MEAN plane (lspiv_ensemble_finish) =========================================
// (i) flag: one wave per window on the float32 mean plane.  Candidates of the arg-max = samples within tau of the maximum; the
// fit's conditioning by the same model and noise allowance as the per-pair epilogues (peak_cond: the mean of float32 planes
// summed in float32 carries no more noise relative to its maximum than one plane, 1.5e-7 median / 3.8e-7 worst over ensembles
// of 2 ... 400 pairs, tools/ens_diag.py).  A NaN plane (count filter) or an all-zero one is NaN by construction, never listed.
__device__ __forceinline__ void wave_argmax_first(float& v, int& idx) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) {
    const float pv = __shfl_xor(v, o, 64);
    const int pi = __shfl_xor(idx, o, 64);
    const bool take = (pv > v) || (pv == v && pi < idx);
    v = take ? pv : v;
    idx = take ? pi : idx;
  }
}
__device__ __forceinline__ int wave_min_i(int x) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) x = min(x, __shfl_xor(x, o, 64));
  return x;
}
__device__ __forceinline__ int wave_sum_i(int x) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) x += __shfl_xor(x, o, 64);
  return x;
}

__global__ __launch_bounds__(RBLOCK) void ens_flag_kernel(const float* mean, uint32_t n_win, int wy, int wx, const float* u, const float* v,
                                                          float k, float tau, EnsRescueHdr* hdr, EnsRescueRec* recs, uint32_t cap) {
  const int lane = threadIdx.x & 63;
  const uint32_t w = blockIdx.x * (RBLOCK / 64) + (threadIdx.x >> 6);
  if (w >= n_win) return;   // whole waves
  const int n = wy * wx;
  const float* pl = mean + (size_t)w * n;
  float best = -1.0f;
  int bi = 0x7fffffff;
  bool bad = false;
  for (int o = lane; o < n; o += 64) {
    const float x = pl[o];
    bad = bad || !(x == x);
    if (x > best) { best = x; bi = o; }
  }
  if (__builtin_amdgcn_ballot_w64(bad) != 0) return;
  wave_argmax_first(best, bi);
  if (!(best > 0.0f)) return;
  const float thr = best * (1.0f - tau);
  int cnt = 0;
  for (int o = lane; o < n; o += 64) cnt += pl[o] >= thr ? 1 : 0;
  cnt = wave_sum_i(cnt);
  const int i = bi / wx, j = bi - i * wx;
  const bool border = i <= 0 || i >= wy - 1 || j <= 0 || j >= wx - 1;
  bool fit = false;
  if (cnt == 1 && !border) {
    const float cl = pl[bi - wx] + kEpsPeak, cr = pl[bi + wx] + kEpsPeak, cd = pl[bi - 1] + kEpsPeak, cu = pl[bi + 1] + kEpsPeak;
    const float l0 = __builtin_amdgcn_logf(best + kEpsPeak);
    float den_v, den_u;
    gauss_offset_fast(__builtin_amdgcn_logf(cl), l0, __builtin_amdgcn_logf(cr), den_v);
    gauss_offset_fast(__builtin_amdgcn_logf(cd), l0, __builtin_amdgcn_logf(cu), den_u);
    // the results the fit was flagged for are relative to the plane centre, like the per-pair epilogues'
    fit = peak_cond(best, false, false, cl, cr, den_v, v[w], cd, cu, den_u, u[w], k).fit;
  }
  if (cnt < 2 && !fit) return;
  EnsRescueRec r;
  r.w = w; r.ncand = cnt <= kEnsMaxCand ? (uint32_t)cnt : 0u;
  r.pad[0] = r.pad[1] = 0;
  int prev = -1;
#pragma unroll
  for (int c = 0; c < kEnsMaxCand; ++c) {
    int m = 0x7fffffff;
    if (c < cnt && cnt <= kEnsMaxCand) {
      for (int o = lane; o < n; o += 64) m = (pl[o] >= thr && o > prev) ? min(m, o) : m;
      m = wave_min_i(m);
      prev = m;
    }
    r.pos[c] = m == 0x7fffffff ? 0xffffffffu : (((uint32_t)(m / wx) << 16) | (uint32_t)(m - (m / wx) * wx));
  }
  if (lane == 0) {
    if (r.ncand == 0) atomicAdd(&hdr->n_skipped, 1u);
    const uint32_t slot = atomicAdd(&hdr->n_rec, 1u);
    if (slot < cap) recs[slot] = r;
  }
}

// (ii) partial sums: one wave per (record, block of kEnsPairBlock pairs of this chunk).  For every pair of the block that was
// ADDED to the sum (its masked corr_max is > 0: the decision the float32 kernel took and returned), the two windows are staged
// once, their float64 statistics taken, and for every candidate the five clipped lag sums are added up in pair order.
template <typename T>
__global__ __launch_bounds__(RBLOCK) void ens_partial_kernel(PivParams p, EnsRescueArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char fsm[];
  typedef const T __attribute__((address_space(3))) * LdsPtr;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wy = p.wy, wx = p.wx, n = wy * wx;
  const bool staged = (size_t)2 * n * sizeof(T) <= (size_t)FIT_LDS_PER_WAVE;
  T* la = reinterpret_cast<T*>(fsm + (size_t)wave * FIT_LDS_PER_WAVE);
  T* lb = la + n;
  const uint32_t chunk_blks = (a.n_pairs + kEnsPairBlock - 1) / kEnsPairBlock;
  const uint32_t items = a.n_rec * chunk_blks, n_waves = gridDim.x * (RBLOCK / 64);
  for (uint32_t it = blockIdx.x * (RBLOCK / 64) + (uint32_t)wave; it < items; it += n_waves) {
    const uint32_t ri = it / chunk_blks, blk = it - ri * chunk_blks;
    const EnsRescueRec rec = a.recs[ri];
    double acc[kEnsMaxCand][5];
#pragma unroll
    for (int c = 0; c < kEnsMaxCand; ++c)
#pragma unroll
      for (int q = 0; q < 5; ++q) acc[c][q] = 0.0;
    const uint32_t pa = blk * kEnsPairBlock, pb = min(pa + (uint32_t)kEnsPairBlock, a.n_pairs);
    for (uint32_t pair = pa; pair < pb && rec.ncand != 0; ++pair) {
      if (!(a.cmax[(size_t)pair * p.n_win + rec.w] > 0.0f)) continue;   // not in the sum (uniform over the wave)
      const T* A = window_base<T>(p, pair * p.n_win + rec.w);
      const T* B = A + p.frame_elems;
      double mean_a, sd_a, mean_b, sd_b;
      if (staged) window_stats_wave2<T, true>(A, B, p.W, wy, wx, lane, mean_a, sd_a, mean_b, sd_b, la, lb);
      else window_stats_wave2<T, false>(A, B, p.W, wy, wx, lane, mean_a, sd_a, mean_b, sd_b, nullptr, nullptr);
      if (sd_a == 0.0 || sd_b == 0.0) continue;   // (a zero-variance window has corr_max 0 and is never kept)
      const double sg = p.norm_clip ? (double)p.std_gain : -(double)p.std_gain;
      const double inv_a = sg / sd_a, inv_b = sg / sd_b;
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int c = 0; c < kEnsMaxCand; ++c) {
        if (c >= (int)rec.ncand) break;
        const int ip = (int)(rec.pos[c] >> 16), jp = (int)(rec.pos[c] & 0xffffu);
        const Lag5 l = staged ? lag5_wave(p, (LdsPtr)la, (LdsPtr)lb, wx, mean_a, inv_a, mean_b, inv_b, ip, jp, lane)
                              : lag5_wave(p, A, B, p.W, mean_a, inv_a, mean_b, inv_b, ip, jp, lane);
        acc[c][0] += l.c0; acc[c][1] += l.cu; acc[c][2] += l.cd; acc[c][3] += l.cl; acc[c][4] += l.cr;
      }
      __builtin_amdgcn_wave_barrier();
    }
    if (lane == 0) {
      double* dst = a.partial + ((size_t)ri * a.n_blk + a.blk0 + blk) * (kEnsMaxCand * 5);
#pragma unroll
      for (int c = 0; c < kEnsMaxCand; ++c)
#pragma unroll
        for (int q = 0; q < 5; ++q) dst[c * 5 + q] = acc[c][q];
    }
  }
}

// (A variant for power-of-two widths -- both windows normalised once per pair into the slice as float32, a lane owning one column so
// that the column wraps of a lag are lane constants -- was built and measured against this kernel with hundreds of windows flagged:
// 6.6 vs 4.9 ms (32 x 32, 395 windows x 2000 pairs), 24.2 vs 16.5 ms (64 x 64, 385 windows), 21.9 vs 16.2 and 78.4 vs 54.3 ms at 1 450
// windows; the per-pair fit kernel through the same machinery: 215 vs 132 us per 32 k records.  The second pass over the samples
// costs more than the repeated normalisation: a record is a chain of short dependent steps either way.  Removed.)

// (iii) merge a handle's partial sums in pair-block order (one thread per (record, candidate, sample): a fixed order)
__global__ __launch_bounds__(RBLOCK) void ens_merge_kernel(EnsRescueArgs a, double* totals) {
  constexpr int ROW = kEnsMaxCand * 5;
  const uint32_t i = blockIdx.x * RBLOCK + threadIdx.x;
  if (i >= a.n_rec * ROW) return;
  const uint32_t ri = i / ROW, q = i - ri * ROW;
  const double* src = a.partial + (size_t)ri * a.n_blk * ROW + q;
  double sum = 0.0;
  for (uint32_t b = 0; b < a.n_blk; ++b) sum += src[(size_t)b * ROW];
  totals[i] = sum;
}

// (iv) totals over ALL pairs of the sum (one handle's, or the all-reduced ones of several): divide by the count, pick the
// candidate with the largest centre (ties: the smaller row-major index -- candidates are listed in that order), fit in float64
__global__ __launch_bounds__(RBLOCK) void ens_final_kernel(PivParams p, EnsRescueArgs a, const double* totals, float* u, float* v) {
  constexpr int ROW = kEnsMaxCand * 5;
  const uint32_t ri = blockIdx.x * RBLOCK + threadIdx.x;
  if (ri >= a.n_rec) return;
  const EnsRescueRec rec = a.recs[ri];
  if (rec.ncand == 0) return;
  const double cnt = (double)a.count[rec.w];
  if (!(cnt > 0.0)) return;
  const int cy = p.wy / 2, cx = p.wx / 2;
  int best = -1;
  Lag5 bl = {0, 0, 0, 0, 0};
  for (int c = 0; c < (int)rec.ncand; ++c) {
    const double* t = totals + (size_t)ri * ROW + c * 5;
    const Lag5 l = {t[0] / cnt, t[1] / cnt, t[2] / cnt, t[3] / cnt, t[4] / cnt};
    if (best < 0 || l.c0 > bl.c0) { best = c; bl = l; }
  }
  const int ip = (int)(rec.pos[best] >> 16), jp = (int)(rec.pos[best] & 0xffffu);
  float uu, vv;
  if (ip <= 0 || ip >= p.wy - 1 || jp <= 0 || jp >= p.wx - 1) border_result(p.border_mode, jp - cx, ip - cy, uu, vv);
  else fit5_d(bl, ip, jp, cy, cx, uu, vv);
  u[rec.w] = uu; v[rec.w] = vv;
}

template <typename T>
hipError_t launch_rescue_t(const PivParams& p, hipStream_t s) {
  const int n = p.wy * p.wx;
  const bool fast = n <= RESCUE_LDS_SAMPLES && (p.wx % AMB_R) == 0;
  const size_t lds = fast ? (size_t)3 * n * sizeof(double) + (size_t)(n + p.wy) * sizeof(int) : 0;
  // fixed grids (the record counts live on the device): empty blocks leave within microseconds
  const uint32_t fit_blocks = std::min<uint32_t>(4096u, std::max<uint32_t>(64u, p.n_tiles / 512u + 1u));
  const uint32_t amb_blocks = std::min<uint32_t>(1024u, std::max<uint32_t>(64u, p.n_tiles / 2048u + 1u));
  const size_t fit_lds = (size_t)2 * n * sizeof(T) <= (size_t)FIT_LDS_PER_WAVE ? (size_t)(RBLOCK / 64) * FIT_LDS_PER_WAVE : 0;
  hipLaunchKernelGGL(piv_rescue_fit_kernel<T>, dim3(fit_blocks), dim3(RBLOCK), fit_lds, s, p);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  // (per launch: the attribute belongs to the current device, and a process may drive several)
  e = hipFuncSetAttribute(reinterpret_cast<const void*>(&piv_rescue_amb_kernel<T>), hipFuncAttributeMaxDynamicSharedMemorySize,
                          (int)(3 * RESCUE_LDS_SAMPLES * sizeof(double) + (RESCUE_LDS_SAMPLES + 512) * sizeof(int)));
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(piv_rescue_amb_kernel<T>, dim3(amb_blocks), dim3(RBLOCK), lds, s, p);
  return hipGetLastError();
}

}  // namespace

hipError_t launch_ens_flag(const float* mean, uint32_t n_win, int wy, int wx, const float* u, const float* v, float k, float tau,
                           EnsRescueHdr* hdr, EnsRescueRec* recs, uint32_t cap, hipStream_t s) {
  if (n_win == 0) return hipSuccess;
  hipLaunchKernelGGL(ens_flag_kernel, dim3((n_win + RBLOCK / 64 - 1) / (RBLOCK / 64)), dim3(RBLOCK), 0, s, mean, n_win, wy, wx, u, v, k, tau,
                     hdr, recs, cap);
  return hipGetLastError();
}

template <typename T>
static void launch_ens_partial_t(const PivParams& p, const EnsRescueArgs& a, uint32_t blocks, hipStream_t s) {
  const int n = p.wy * p.wx;
  const size_t lds = (size_t)2 * n * sizeof(T) <= (size_t)FIT_LDS_PER_WAVE ? (size_t)(RBLOCK / 64) * FIT_LDS_PER_WAVE : 0;
  hipLaunchKernelGGL(ens_partial_kernel<T>, dim3(blocks), dim3(RBLOCK), lds, s, p, a);
}

hipError_t launch_ens_partial(const PivParams& p, int dtype, const EnsRescueArgs& a, hipStream_t s) {
  const uint32_t chunk_blks = (a.n_pairs + kEnsPairBlock - 1) / kEnsPairBlock;
  const uint64_t items = (uint64_t)a.n_rec * chunk_blks;
  if (items == 0) return hipSuccess;
  const uint32_t blocks = (uint32_t)std::min<uint64_t>((items + RBLOCK / 64 - 1) / (RBLOCK / 64), 65536);
  switch (dtype) {
    case 0: launch_ens_partial_t<uint8_t>(p, a, blocks, s); break;
    case 1: launch_ens_partial_t<float>(p, a, blocks, s); break;
    case 2: launch_ens_partial_t<double>(p, a, blocks, s); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

hipError_t launch_ens_merge(const EnsRescueArgs& a, double* totals, hipStream_t s) {
  if (a.n_rec == 0) return hipSuccess;
  const uint32_t n = a.n_rec * kEnsMaxCand * 5;
  hipLaunchKernelGGL(ens_merge_kernel, dim3((n + RBLOCK - 1) / RBLOCK), dim3(RBLOCK), 0, s, a, totals);
  return hipGetLastError();
}

hipError_t launch_ens_final(const PivParams& p, const EnsRescueArgs& a, const double* totals, float* u, float* v, hipStream_t s) {
  if (a.n_rec == 0) return hipSuccess;
  hipLaunchKernelGGL(ens_final_kernel, dim3((a.n_rec + RBLOCK - 1) / RBLOCK), dim3(RBLOCK), 0, s, p, a, totals, u, v);
  return hipGetLastError();
}

hipError_t launch_piv_rescue(const PivParams& p, int dtype, hipStream_t s) {
  if (!p.rescue_hdr) return hipSuccess;
  switch (dtype) {
    case 0: return launch_rescue_t<uint8_t>(p, s);
    case 1: return launch_rescue_t<float>(p, s);
    case 2: return launch_rescue_t<double>(p, s);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace lspiv
