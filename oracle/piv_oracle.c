/*
 * piv_oracle.c -- plain-C (float64) restatement of the LSPIV hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * ** PARITY UNPINNED ** (same caveat as oracle/piv_oracle.py, which is the primary checker and the
 * place where every semantic choice A1..A8 is documented).  This file exists so that the CPU
 * baseline of bench.py can use all host cores (OpenMP over interrogation windows), which is how
 * the reference's numba engine parallelises (prange inside ffpiv), instead of single-threaded numpy.
 * It is validated against oracle/piv_oracle.py in tests/test_oracle.py.
 *
 * Restates (the arithmetic lives in the third-party ffpiv >= 0.2.1, absent from /root/reference):
 *   ffpiv.window.get_axis_shape / sliding windows      call sites pyorc/api/frames.py:85-90
 *   ffpiv.pivnp.normalize_intensity / ncc               call sites pyorc/velocimetry/ffpiv.py:222,450
 *   corr_max = nanmax, s2n = corr_max / nanmean         pyorc/velocimetry/ffpiv.py:465-466
 *   ffpiv.pivnp.peak_position / u_v_displacement        call sites pyorc/velocimetry/ffpiv.py:324,471
 *
 * Algorithm per window pair (a from frame t, b from frame t+1):
 *   a' = max((a - mean a)/std a, 0), b' likewise (zeros when std == 0)
 *   plane = clip(fftshift(IFFT2(conj(FFT2 a') FFT2 b')) / (wy wx)^... , 0, 1)    [numpy irfft2 scaling]
 *   both forward transforms are obtained from ONE complex transform of a' + i b'.
 * Power-of-two sides use an iterative radix-2 FFT, any other side a direct DFT (test sizes only).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define MAXW 512
#define EPS_PEAK 1e-7

typedef struct { double re, im; } cpx;

typedef struct {
  int n, pow2, log2n;
  cpx tw[MAXW];          /* exp(-2 pi i k / n) */
  int rev[MAXW];
} plan1d;

static void plan_init(plan1d* p, int n) {
  p->n = n;
  p->pow2 = (n & (n - 1)) == 0;
  p->log2n = 0;
  while ((1 << p->log2n) < n) p->log2n++;
  for (int k = 0; k < n; ++k) {
    double ang = -2.0 * M_PI * (double)k / (double)n;
    p->tw[k].re = cos(ang);
    p->tw[k].im = sin(ang);
  }
  if (p->pow2) {
    for (int i = 0; i < n; ++i) {
      int r = 0;
      for (int b = 0; b < p->log2n; ++b) if (i & (1 << b)) r |= 1 << (p->log2n - 1 - b);
      p->rev[i] = r;
    }
  }
}

/* in-place transform of x[0], x[stride], ...; inverse = conjugate twiddles, unnormalised */
static void fft1d(const plan1d* p, cpx* x, int stride, int inverse) {
  const int n = p->n;
  cpx t[MAXW];
  if (p->pow2) {
    for (int i = 0; i < n; ++i) t[p->rev[i]] = x[i * stride];
    for (int len = 2; len <= n; len <<= 1) {
      const int half = len >> 1, step = n / len;
      for (int s = 0; s < n; s += len) {
        for (int k = 0; k < half; ++k) {
          const cpx w = p->tw[k * step];
          const double wi = inverse ? -w.im : w.im;
          cpx* a = &t[s + k];
          cpx* b = &t[s + k + half];
          const double br = b->re * w.re - b->im * wi;
          const double bi = b->re * wi + b->im * w.re;
          b->re = a->re - br; b->im = a->im - bi;
          a->re += br;        a->im += bi;
        }
      }
    }
    for (int i = 0; i < n; ++i) x[i * stride] = t[i];
  } else {
    for (int k = 0; k < n; ++k) {
      double sr = 0.0, si = 0.0;
      for (int j = 0; j < n; ++j) {
        const cpx w = p->tw[(int)(((long)j * k) % n)];
        const double wi = inverse ? -w.im : w.im;
        const cpx v = x[j * stride];
        sr += v.re * w.re - v.im * wi;
        si += v.re * wi + v.im * w.re;
      }
      t[k].re = sr; t[k].im = si;
    }
    for (int i = 0; i < n; ++i) x[i * stride] = t[i];
  }
}

static void fft2d(const plan1d* py, const plan1d* px, cpx* z, int wy, int wx, int inverse) {
  for (int y = 0; y < wy; ++y) fft1d(px, z + (size_t)y * wx, 1, inverse);
  for (int x = 0; x < wx; ++x) fft1d(py, z + x, wx, inverse);
}

static double load_px(const void* frames, int dtype, size_t idx) {
  switch (dtype) {
    case 0: return (double)((const uint8_t*)frames)[idx];
    case 1: return (double)((const float*)frames)[idx];
    default: return ((const double*)frames)[idx];
  }
}

/* normalise one window in place (re or im part of z); returns count of non-zero raw samples */
static int normalize_part(cpx* z, int n, int part, int* dead) {
  /* mean as x0 + mean(x - x0): algebraically the mean, and exactly x0 for a constant window, so that "zeros if
     std == 0" (A3) does not depend on whether n * c happens to be representable */
  const double x0 = part ? z[0].im : z[0].re;
  double s = 0.0;
  int nz = 0;
  for (int i = 0; i < n; ++i) {
    const double v = part ? z[i].im : z[i].re;
    s += v - x0;
    nz += (v != 0.0);
  }
  const double mean = x0 + s / n;
  double ss = 0.0;
  for (int i = 0; i < n; ++i) {
    const double d = (part ? z[i].im : z[i].re) - mean;
    ss += d * d;
  }
  const double sd = sqrt(ss / n);
  if (sd == 0.0) *dead = 1;
  for (int i = 0; i < n; ++i) {
    double d = (part ? z[i].im : z[i].re) - mean;
    d = (sd != 0.0) ? d / sd : 0.0;
    d = d > 0.0 ? d : 0.0;
    if (part) z[i].im = d; else z[i].re = d;
  }
  return nz;
}

/*
 * frames (T,H,W); outputs (T-1)*n_rows*n_cols float32 each; planes NULL or (T-1)*n_win*wy*wx float64.
 * signal_threshold < 0: off.  cond: NULL or 3 floats per window that grade how well-posed the window is for a
 * float32 implementation: [0] (max - runner-up)/max over the whole plane (argmax stability), [1] smallest of the
 * five peak-neighbourhood values / max (log-Gaussian sensitivity; 1 for a border peak, whose result is NaN
 * anyway), [2] min(|den_row|, |den_col|) of the log-Gaussian fit (a flat ridge divides rounding noise by a tiny
 * curvature; 1 for a border peak); all 0 for NaN / all-zero planes.
 * Returns 0, or -1 on bad arguments.
 */
int piv_oracle_pairs(const void* frames, int dtype, long T, long H, long W, int wy, int wx, int oy, int ox,
                     double signal_threshold, float* u, float* v, float* cmax, float* s2n, double* planes,
                     float* cond, int nthreads) {
  if (!frames || T < 2 || wy < 2 || wx < 2 || wy > MAXW || wx > MAXW || oy >= wy || ox >= wx || oy < 0 || ox < 0)
    return -1;
  if (H < wy || W < wx) return -1;
  const long n_rows = (H - wy) / (wy - oy) + 1, n_cols = (W - wx) / (wx - ox) + 1;
  const long n_win = n_rows * n_cols, n_tiles = (T - 1) * n_win;
  const int sy = wy - oy, sx = wx - ox, n = wy * wx;
  const int cy = wy / 2, cx = wx / 2;
  plan1d py, px;
  plan_init(&py, wy);
  plan_init(&px, wx);
#ifdef _OPENMP
  if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
#pragma omp parallel
  {
    cpx* z = (cpx*)malloc(sizeof(cpx) * n);
    cpx* r = (cpx*)malloc(sizeof(cpx) * n);
    double* pl = (double*)malloc(sizeof(double) * n);
#pragma omp for schedule(static)
    for (long g = 0; g < n_tiles; ++g) {
      const long pair = g / n_win, win = g % n_win;
      const long wr = win / n_cols, wc = win % n_cols;
      const size_t base = ((size_t)pair * H + (size_t)wr * sy) * W + (size_t)wc * sx;
      for (int y = 0; y < wy; ++y)
        for (int x = 0; x < wx; ++x) {
          const size_t idx = base + (size_t)y * W + x;
          z[y * wx + x].re = load_px(frames, dtype, idx);
          z[y * wx + x].im = load_px(frames, dtype, idx + (size_t)H * W);
        }
      int dead = 0; /* a zero-variance window normalises to zeros: numpy's transform of it is exactly 0 */
      const int nza = normalize_part(z, n, 0, &dead);
      const int nzb = normalize_part(z, n, 1, &dead);
      int skip = 0;
      if (signal_threshold >= 0.0)
        skip = !((double)nza / n >= signal_threshold && (double)nzb / n >= signal_threshold);
      float uo = NAN, vo = NAN, cm = NAN, sn = NAN;
      if (!skip) {
        fft2d(&py, &px, z, wy, wx, 0);
        /* R = conj(A) B with A = (Z[k] + conj Z[-k])/2, B = (Z[k] - conj Z[-k])/(2i) */
        for (int ky = 0; ky < wy; ++ky)
          for (int kx = 0; kx < wx; ++kx) {
            const cpx zk = z[ky * wx + kx];
            const cpx zn = z[((wy - ky) % wy) * wx + (wx - kx) % wx];
            const double ar = 0.5 * (zk.re + zn.re), ai = 0.5 * (zk.im - zn.im);
            const double br = 0.5 * (zk.im + zn.im), bi = -0.5 * (zk.re - zn.re);
            r[ky * wx + kx].re = ar * br + ai * bi;   /* conj(A) * B */
            r[ky * wx + kx].im = ar * bi - ai * br;
          }
        fft2d(&py, &px, r, wy, wx, 1);
        if (dead) memset(r, 0, sizeof(cpx) * n);
        /* irfft2 scaling 1/n, then the /n of ncc; fftshift; clip */
        double mx = -1.0, sum = 0.0;
        int imax = 0;
        const double sc = 1.0 / ((double)n * (double)n);
        for (int ip = 0; ip < wy; ++ip)
          for (int jp = 0; jp < wx; ++jp) {
            const int ysrc = (ip - cy + wy) % wy, xsrc = (jp - cx + wx) % wx;
            double c = r[ysrc * wx + xsrc].re * sc;
            c = c < 0.0 ? 0.0 : (c > 1.0 ? 1.0 : c);
            pl[ip * wx + jp] = c;
            sum += c;
            if (c > mx) { mx = c; imax = ip * wx + jp; }
          }
        cm = (float)mx;
        sn = (float)(mx / (sum / n));
        const int i = imax / wx, j = imax % wx;
        if (i > 0 && i < wy - 1 && j > 0 && j < wx - 1) {
          const double l0 = log(pl[i * wx + j] + EPS_PEAK);
          const double lu = log(pl[(i - 1) * wx + j] + EPS_PEAK), ld = log(pl[(i + 1) * wx + j] + EPS_PEAK);
          const double ll = log(pl[i * wx + j - 1] + EPS_PEAK), lr = log(pl[i * wx + j + 1] + EPS_PEAK);
          const double den1 = 2 * lu - 4 * l0 + 2 * ld, den2 = 2 * ll - 4 * l0 + 2 * lr;
          const double di = den1 != 0.0 ? (lu - ld) / den1 : 0.0;
          const double dj = den2 != 0.0 ? (ll - lr) / den2 : 0.0;
          vo = (float)(i + di - cy);
          uo = (float)(j + dj - cx);
        }
      }
      u[g] = uo; v[g] = vo; cmax[g] = cm; s2n[g] = sn;
      if (cond) {
        float gap = 0.0f, mnb = 0.0f, curv = 0.0f;
        if (!skip && cm > 0.0f) {
          const double mx = cm;
          int imax = 0;
          double best = -1.0, second = -1.0;
          for (int i = 0; i < n; ++i) if (pl[i] > best) { best = pl[i]; imax = i; }
          for (int i = 0; i < n; ++i) if (i != imax && pl[i] > second) second = pl[i];
          gap = (float)((best - second) / best);
          const int i = imax / wx, j = imax % wx;
          if (i > 0 && i < wy - 1 && j > 0 && j < wx - 1) {
            double m = pl[imax];
            if (pl[(i - 1) * wx + j] < m) m = pl[(i - 1) * wx + j];
            if (pl[(i + 1) * wx + j] < m) m = pl[(i + 1) * wx + j];
            if (pl[i * wx + j - 1] < m) m = pl[i * wx + j - 1];
            if (pl[i * wx + j + 1] < m) m = pl[i * wx + j + 1];
            mnb = (float)(m / mx);
            const double l0 = log(pl[imax] + EPS_PEAK);
            const double d1 = fabs(2 * log(pl[(i - 1) * wx + j] + EPS_PEAK) - 4 * l0 + 2 * log(pl[(i + 1) * wx + j] + EPS_PEAK));
            const double d2 = fabs(2 * log(pl[i * wx + j - 1] + EPS_PEAK) - 4 * l0 + 2 * log(pl[i * wx + j + 1] + EPS_PEAK));
            curv = (float)(d1 < d2 ? d1 : d2);
          } else {
            mnb = 1.0f; /* border peak: the result is NaN whatever the neighbours are */
            curv = 1.0f;
          }
        }
        cond[3 * g] = gap; cond[3 * g + 1] = mnb; cond[3 * g + 2] = curv;
      }
      if (planes) {
        double* dst = planes + (size_t)g * n;
        for (int i = 0; i < n; ++i) dst[i] = skip ? NAN : pl[i];
      }
    }
    free(z); free(r); free(pl);
  }
  return 0;
}

int piv_oracle_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
