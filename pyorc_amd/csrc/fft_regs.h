// In-register complex FFTs for gfx950 wavefronts (one transform per lane, all indices static).
//
// A lane owns a whole row (or column) of an interrogation tile in VGPRs, so a length-N
// transform is straight-line VALU code: no LDS, no cross-lane traffic, twiddles are
// compile-time literals.  N = 16 is 4 radix-4 -> 9 twiddles -> 4 radix-4; N = 32 is 8 radix-4 butterflies ->
// 21 twiddle multiplies -> 4 radix-8 butterflies; N = 64 is 8 radix-8 -> 49 twiddles -> 8 radix-8.
//
// There is no reference kernel to mirror: pyorc delegates the FFT to rocket_fft/pocketfft on
// the CPU.  Sign convention matches numpy: forward = exp(-2 pi i nk/N),
// inverse = exp(+2 pi i nk/N), both unnormalised (the 1/N^2 is folded into the final scale).
#pragma once
#include <hip/hip_runtime.h>

namespace lspiv {

// cos(2 pi m / 64), m = 0..63, rounded from double
static constexpr float kCos64[64] = {
    1.0f, 0.9951847195625305f, 0.9807852506637573f, 0.9569403529167175f,
    0.9238795042037964f, 0.8819212913513184f, 0.8314695954322815f, 0.7730104327201843f,
    0.7071067690849304f, 0.6343932747840881f, 0.5555702447891235f, 0.4713967442512512f,
    0.3826834261417389f, 0.290284663438797f, 0.19509032368659973f, 0.0980171412229538f,
    0.0f, -0.0980171412229538f, -0.19509032368659973f, -0.290284663438797f,
    -0.3826834261417389f, -0.4713967442512512f, -0.5555702447891235f, -0.6343932747840881f,
    -0.7071067690849304f, -0.7730104327201843f, -0.8314695954322815f, -0.8819212913513184f,
    -0.9238795042037964f, -0.9569403529167175f, -0.9807852506637573f, -0.9951847195625305f,
    -1.0f, -0.9951847195625305f, -0.9807852506637573f, -0.9569403529167175f,
    -0.9238795042037964f, -0.8819212913513184f, -0.8314695954322815f, -0.7730104327201843f,
    -0.7071067690849304f, -0.6343932747840881f, -0.5555702447891235f, -0.4713967442512512f,
    -0.3826834261417389f, -0.290284663438797f, -0.19509032368659973f, -0.0980171412229538f,
    0.0f, 0.0980171412229538f, 0.19509032368659973f, 0.290284663438797f,
    0.3826834261417389f, 0.4713967442512512f, 0.5555702447891235f, 0.6343932747840881f,
    0.7071067690849304f, 0.7730104327201843f, 0.8314695954322815f, 0.8819212913513184f,
    0.9238795042037964f, 0.9569403529167175f, 0.9807852506637573f, 0.9951847195625305f};

__host__ __device__ constexpr float cos64(int m) { return kCos64[m & 63]; }
__host__ __device__ constexpr float sin64(int m) { return kCos64[(m - 16) & 63]; }

constexpr float kSqrtHalf = 0.70710678118654757f;

// x *= exp(-/+ 2 pi i m / 64)   (forward: minus; INV: plus)
template <bool INV, int M>
__device__ __forceinline__ void twiddle64(float& xr, float& xi) {
  constexpr int m = M & 63;
  if constexpr (m == 0) {
    return;
  } else if constexpr (m == 16) {  // -i (fwd) / +i (inv)
    float t = xr;
    if constexpr (!INV) { xr = xi; xi = -t; } else { xr = -xi; xi = t; }
  } else if constexpr (m == 32) {
    xr = -xr; xi = -xi;
  } else if constexpr (m == 48) {
    float t = xr;
    if constexpr (!INV) { xr = -xi; xi = t; } else { xr = xi; xi = -t; }
  } else {
    constexpr float c = cos64(m);
    constexpr float s = INV ? sin64(m) : -sin64(m);
    float tr = xr * c - xi * s;
    float ti = xr * s + xi * c;
    xr = tr; xi = ti;
  }
}

// CLAMP: the outputs of the LAST butterfly stage of an inverse transform are correlation-plane samples, which the
// caller clips to [0, 1]; written as med3(x, 0, 1) of the final add / sub, the clip folds into the VALU clamp modifier of
// that instruction (v_add_f32 ... clamp) and costs nothing.
template <bool CLAMP>
__device__ __forceinline__ float out01(float x) { return CLAMP ? __builtin_amdgcn_fmed3f(x, 0.0f, 1.0f) : x; }

// radix-4 butterfly, natural-order outputs
template <bool INV, bool CLAMP = false>
__device__ __forceinline__ void bfly4(float& r0, float& i0, float& r1, float& i1,
                                      float& r2, float& i2, float& r3, float& i3) {
  float ar = r0 + r2, ai = i0 + i2;
  float br = r0 - r2, bi = i0 - i2;
  float cr = r1 + r3, ci = i1 + i3;
  float dr = r1 - r3, di = i1 - i3;
  r0 = out01<CLAMP>(ar + cr); i0 = out01<CLAMP>(ai + ci);
  r2 = out01<CLAMP>(ar - cr); i2 = out01<CLAMP>(ai - ci);
  if constexpr (!INV) {  // X1 = b - i d, X3 = b + i d
    r1 = out01<CLAMP>(br + di); i1 = out01<CLAMP>(bi - dr);
    r3 = out01<CLAMP>(br - di); i3 = out01<CLAMP>(bi + dr);
  } else {               // X1 = b + i d, X3 = b - i d
    r1 = out01<CLAMP>(br - di); i1 = out01<CLAMP>(bi + dr);
    r3 = out01<CLAMP>(br + di); i3 = out01<CLAMP>(bi - dr);
  }
}

// radix-8 butterfly on 8 values (natural order in, natural order out)
template <bool INV, bool CLAMP = false>
__device__ __forceinline__ void bfly8(float (&r)[8], float (&i)[8]) {
  // even / odd radix-4
  bfly4<INV>(r[0], i[0], r[2], i[2], r[4], i[4], r[6], i[6]);  // E0..E3 in slots 0,2,4,6
  bfly4<INV>(r[1], i[1], r[3], i[3], r[5], i[5], r[7], i[7]);  // O0..O3 in slots 1,3,5,7
  // O1 *= W8^1, O2 *= W8^2, O3 *= W8^3
  {
    float tr = r[3], ti = i[3];
    if constexpr (!INV) { r[3] = (tr + ti) * kSqrtHalf; i[3] = (ti - tr) * kSqrtHalf; }
    else                { r[3] = (tr - ti) * kSqrtHalf; i[3] = (ti + tr) * kSqrtHalf; }
    tr = r[5]; ti = i[5];
    if constexpr (!INV) { r[5] = ti; i[5] = -tr; } else { r[5] = -ti; i[5] = tr; }
    tr = r[7]; ti = i[7];
    if constexpr (!INV) { r[7] = (ti - tr) * kSqrtHalf; i[7] = -(tr + ti) * kSqrtHalf; }
    else                { r[7] = -(tr + ti) * kSqrtHalf; i[7] = (tr - ti) * kSqrtHalf; }
  }
  float er[4] = {r[0], r[2], r[4], r[6]}, ei[4] = {i[0], i[2], i[4], i[6]};
  float orr[4] = {r[1], r[3], r[5], r[7]}, oi[4] = {i[1], i[3], i[5], i[7]};
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    r[k] = out01<CLAMP>(er[k] + orr[k]);     i[k] = out01<CLAMP>(ei[k] + oi[k]);
    r[k + 4] = out01<CLAMP>(er[k] - orr[k]); i[k + 4] = out01<CLAMP>(ei[k] - oi[k]);
  }
}

template <bool INV, int N2, int K1, int STEP>
struct TwiddleRow {
  // x[n2] *= W^(STEP * n2 * K1) for n2 = 1..7 (W = 64th root); unrolled at compile time
  template <int n2>
  static __device__ __forceinline__ void apply(float (&r)[8], float (&i)[8]) {
    if constexpr (n2 < N2) {
      twiddle64<INV, STEP * n2 * K1>(r[n2], i[n2]);
      apply<n2 + 1>(r, i);
    }
  }
};

// ---- length-16 transform: n = 4 n1 + n2, k = k1 + 4 k2 ------------------------------------------------
//   X[k1 + 4 k2] = DFT4_{n2}( W16^{n2 k1} * DFT4_{n1} x[4 n1 + n2] ),  W16 = W64^4
template <bool INV, bool CLAMP = false>
__device__ __forceinline__ void fft16(float (&xr)[16], float (&xi)[16]) {
#pragma unroll
  for (int n2 = 0; n2 < 4; ++n2)
    bfly4<INV>(xr[n2], xi[n2], xr[4 + n2], xi[4 + n2], xr[8 + n2], xi[8 + n2], xr[12 + n2], xi[12 + n2]);   // slot (k1, n2) = 4 k1 + n2
  twiddle64<INV, 4 * 1 * 1>(xr[4 + 1], xi[4 + 1]);  twiddle64<INV, 4 * 2 * 1>(xr[4 + 2], xi[4 + 2]);  twiddle64<INV, 4 * 3 * 1>(xr[4 + 3], xi[4 + 3]);
  twiddle64<INV, 4 * 1 * 2>(xr[8 + 1], xi[8 + 1]);  twiddle64<INV, 4 * 2 * 2>(xr[8 + 2], xi[8 + 2]);  twiddle64<INV, 4 * 3 * 2>(xr[8 + 3], xi[8 + 3]);
  twiddle64<INV, 4 * 1 * 3>(xr[12 + 1], xi[12 + 1]); twiddle64<INV, 4 * 2 * 3>(xr[12 + 2], xi[12 + 2]); twiddle64<INV, 4 * 3 * 3>(xr[12 + 3], xi[12 + 3]);
  float yr[16], yi[16];
#pragma unroll
  for (int k1 = 0; k1 < 4; ++k1) {
    float r0 = xr[4 * k1], i0 = xi[4 * k1], r1 = xr[4 * k1 + 1], i1 = xi[4 * k1 + 1];
    float r2 = xr[4 * k1 + 2], i2 = xi[4 * k1 + 2], r3 = xr[4 * k1 + 3], i3 = xi[4 * k1 + 3];
    bfly4<INV, CLAMP>(r0, i0, r1, i1, r2, i2, r3, i3);
    yr[k1] = r0; yi[k1] = i0; yr[k1 + 4] = r1; yi[k1 + 4] = i1; yr[k1 + 8] = r2; yi[k1 + 8] = i2; yr[k1 + 12] = r3; yi[k1 + 12] = i3;
  }
#pragma unroll
  for (int k = 0; k < 16; ++k) { xr[k] = yr[k]; xi[k] = yi[k]; }
}

// ---- length-32 transform, in place, natural order in and out --------------------------------
// n = 8 n1 + n2, k = k1 + 4 k2:  X[k1 + 4 k2] = DFT8_{n2}( W32^{n2 k1} * DFT4_{n1} x[8 n1 + n2] )
template <bool INV, bool CLAMP = false>
__device__ __forceinline__ void fft32(float (&xr)[32], float (&xi)[32]) {
#pragma unroll
  for (int n2 = 0; n2 < 8; ++n2)
    bfly4<INV>(xr[n2], xi[n2], xr[8 + n2], xi[8 + n2], xr[16 + n2], xi[16 + n2], xr[24 + n2], xi[24 + n2]);
  float yr[32], yi[32];
  // slot (k1, n2) lives at 8*k1 + n2
  {
    float r[8], i[8];
#pragma unroll
    for (int n2 = 0; n2 < 8; ++n2) { r[n2] = xr[n2]; i[n2] = xi[n2]; }
    bfly8<INV, CLAMP>(r, i);
#pragma unroll
    for (int k2 = 0; k2 < 8; ++k2) { yr[4 * k2] = r[k2]; yi[4 * k2] = i[k2]; }
  }
  {
    float r[8], i[8];
#pragma unroll
    for (int n2 = 0; n2 < 8; ++n2) { r[n2] = xr[8 + n2]; i[n2] = xi[8 + n2]; }
    TwiddleRow<INV, 8, 1, 2>::template apply<1>(r, i);
    bfly8<INV, CLAMP>(r, i);
#pragma unroll
    for (int k2 = 0; k2 < 8; ++k2) { yr[1 + 4 * k2] = r[k2]; yi[1 + 4 * k2] = i[k2]; }
  }
  {
    float r[8], i[8];
#pragma unroll
    for (int n2 = 0; n2 < 8; ++n2) { r[n2] = xr[16 + n2]; i[n2] = xi[16 + n2]; }
    TwiddleRow<INV, 8, 2, 2>::template apply<1>(r, i);
    bfly8<INV, CLAMP>(r, i);
#pragma unroll
    for (int k2 = 0; k2 < 8; ++k2) { yr[2 + 4 * k2] = r[k2]; yi[2 + 4 * k2] = i[k2]; }
  }
  {
    float r[8], i[8];
#pragma unroll
    for (int n2 = 0; n2 < 8; ++n2) { r[n2] = xr[24 + n2]; i[n2] = xi[24 + n2]; }
    TwiddleRow<INV, 8, 3, 2>::template apply<1>(r, i);
    bfly8<INV, CLAMP>(r, i);
#pragma unroll
    for (int k2 = 0; k2 < 8; ++k2) { yr[3 + 4 * k2] = r[k2]; yi[3 + 4 * k2] = i[k2]; }
  }
#pragma unroll
  for (int k = 0; k < 32; ++k) { xr[k] = yr[k]; xi[k] = yi[k]; }
}

// ---- length-64 transform: n = 8 n1 + n2, k = k1 + 8 k2 --------------------------------------
template <bool INV, int K1, bool CLAMP = false>
__device__ __forceinline__ void fft64_col(const float (&xr)[64], const float (&xi)[64],
                                          float (&yr)[64], float (&yi)[64]) {
  float r[8], i[8];
#pragma unroll
  for (int n2 = 0; n2 < 8; ++n2) { r[n2] = xr[8 * K1 + n2]; i[n2] = xi[8 * K1 + n2]; }
  TwiddleRow<INV, 8, K1, 1>::template apply<1>(r, i);
  bfly8<INV, CLAMP>(r, i);
#pragma unroll
  for (int k2 = 0; k2 < 8; ++k2) { yr[K1 + 8 * k2] = r[k2]; yi[K1 + 8 * k2] = i[k2]; }
}

// LSPIV_FFT64_SB: scheduling barriers between the butterflies of the two stages -- left alone the scheduler interleaves
// several column transforms for ILP, which stretches live ranges in the register-bound 64 x 64 kernels
#ifndef LSPIV_FFT64_SB
#define LSPIV_FFT64_SB 1
#endif
#ifndef LSPIV_FFT64_PRIO
#define LSPIV_FFT64_PRIO 0
#endif
#define LSPIV_FFT64_BAR do { if (LSPIV_FFT64_SB) __builtin_amdgcn_sched_barrier(0); } while (0)
template <bool INV, bool CLAMP = false>
__device__ __forceinline__ void fft64(float (&xr)[64], float (&xi)[64]) {
#pragma unroll
  for (int n2 = 0; n2 < 8; ++n2) {
    float r[8], i[8];
#pragma unroll
    for (int n1 = 0; n1 < 8; ++n1) { r[n1] = xr[8 * n1 + n2]; i[n1] = xi[8 * n1 + n2]; }
    bfly8<INV>(r, i);
#pragma unroll
    for (int k1 = 0; k1 < 8; ++k1) { xr[8 * k1 + n2] = r[k1]; xi[8 * k1 + n2] = i[k1]; }
    if (LSPIV_FFT64_SB) __builtin_amdgcn_sched_barrier(0);
  }
  float yr[64], yi[64];
#if LSPIV_FFT64_PRIO
  __builtin_amdgcn_s_setprio(1);   // A/B variant: the second radix stage ahead of a neighbour's first
#endif
  fft64_col<INV, 0, CLAMP>(xr, xi, yr, yi); LSPIV_FFT64_BAR;
  fft64_col<INV, 1, CLAMP>(xr, xi, yr, yi); LSPIV_FFT64_BAR;
  fft64_col<INV, 2, CLAMP>(xr, xi, yr, yi); LSPIV_FFT64_BAR;
  fft64_col<INV, 3, CLAMP>(xr, xi, yr, yi); LSPIV_FFT64_BAR;
  fft64_col<INV, 4, CLAMP>(xr, xi, yr, yi); LSPIV_FFT64_BAR;
  fft64_col<INV, 5, CLAMP>(xr, xi, yr, yi); LSPIV_FFT64_BAR;
  fft64_col<INV, 6, CLAMP>(xr, xi, yr, yi); LSPIV_FFT64_BAR;
  fft64_col<INV, 7, CLAMP>(xr, xi, yr, yi); LSPIV_FFT64_BAR;
#if LSPIV_FFT64_PRIO
  __builtin_amdgcn_s_setprio(0);
#endif
#pragma unroll
  for (int k = 0; k < 64; ++k) { xr[k] = yr[k]; xi[k] = yi[k]; }
}

// ---- lengths P * 2^m, P odd (3, 5: hand-written butterflies; 7..31: dft_odd): Good-Thomas prime-factor split, NO twiddles --
// P and 2^m are coprime, so with the index maps  n = (N2 n1 + P n2) mod N  (input) and  k = (e1 k1 + e2 k2) mod N
// (output; e1 = 1 mod P, 0 mod N2;  e2 = 0 mod P, 1 mod N2)  the length-N DFT is exactly a P x N2 two-dimensional
// DFT:  W_N^{nk} = W_P^{n1 k1} W_N2^{n2 k2}.  All indices are compile-time, so the permutations cost nothing.
template <bool INV>
__device__ __forceinline__ void dft_small(float (&r)[3], float (&i)[3]) {
  constexpr float s3 = 0.86602540378443865f;   // sin(2 pi / 3)
  const float tr = r[1] + r[2], ti = i[1] + i[2];
  const float ur = (r[1] - r[2]) * s3, ui = (i[1] - i[2]) * s3;
  const float mr = fmaf(tr, -0.5f, r[0]), mi = fmaf(ti, -0.5f, i[0]);
  r[0] += tr; i[0] += ti;
  if constexpr (!INV) {   // X1 = m - i s3 (b - c),  X2 = m + i s3 (b - c)
    r[1] = mr + ui; i[1] = mi - ur;
    r[2] = mr - ui; i[2] = mi + ur;
  } else {
    r[1] = mr - ui; i[1] = mi + ur;
    r[2] = mr + ui; i[2] = mi - ur;
  }
}
template <bool INV>
__device__ __forceinline__ void dft_small(float (&r)[5], float (&i)[5]) {
  constexpr float c1 = 0.30901699437494742f, c2 = -0.80901699437494742f;   // cos(2 pi / 5), cos(4 pi / 5)
  constexpr float s1 = 0.95105651629515357f, s2 = 0.58778525229247313f;    // sin(2 pi / 5), sin(4 pi / 5)
  const float t1r = r[1] + r[4], t1i = i[1] + i[4], t2r = r[2] + r[3], t2i = i[2] + i[3];
  const float t3r = r[1] - r[4], t3i = i[1] - i[4], t4r = r[2] - r[3], t4i = i[2] - i[3];
  const float m1r = fmaf(c2, t2r, fmaf(c1, t1r, r[0])), m1i = fmaf(c2, t2i, fmaf(c1, t1i, i[0]));
  const float m2r = fmaf(c1, t2r, fmaf(c2, t1r, r[0])), m2i = fmaf(c1, t2i, fmaf(c2, t1i, i[0]));
  const float u1r = fmaf(s2, t4r, s1 * t3r), u1i = fmaf(s2, t4i, s1 * t3i);
  const float u2r = fmaf(-s1, t4r, s2 * t3r), u2i = fmaf(-s1, t4i, s2 * t3i);
  r[0] += t1r + t2r; i[0] += t1i + t2i;
  if constexpr (!INV) {   // X1 = m1 - i u1, X4 = m1 + i u1, X2 = m2 - i u2, X3 = m2 + i u2
    r[1] = m1r + u1i; i[1] = m1i - u1r;
    r[4] = m1r - u1i; i[4] = m1i + u1r;
    r[2] = m2r + u2i; i[2] = m2i - u2r;
    r[3] = m2r - u2i; i[3] = m2i + u2r;
  } else {
    r[1] = m1r - u1i; i[1] = m1i + u1r;
    r[4] = m1r + u1i; i[4] = m1i - u1r;
    r[2] = m2r - u2i; i[2] = m2i + u2r;
    r[3] = m2r + u2i; i[3] = m2i - u2r;
  }
}
// cos / sin (2 pi j / P), rounded from double, for the odd factors without a hand-written butterfly
template <int P> struct RootTab;
template <> struct RootTab<7> {
  static constexpr float c[7] = {1.0f, 0.6234897971153259f, -0.22252093255519867f, -0.9009688496589661f, -0.9009688496589661f, -0.22252093255519867f, 0.6234897971153259f};
  static constexpr float s[7] = {0.0f, 0.7818315029144287f, 0.9749279022216797f, 0.4338837265968323f, -0.4338837265968323f, -0.9749279022216797f, -0.7818315029144287f};
};
template <> struct RootTab<9> {
  static constexpr float c[9] = {1.0f, 0.7660444378852844f, 0.1736481785774231f, -0.5f, -0.9396926164627075f, -0.9396926164627075f, -0.5f, 0.1736481785774231f, 0.7660444378852844f};
  static constexpr float s[9] = {0.0f, 0.6427876353263855f, 0.9848077297210693f, 0.8660253882408142f, 0.3420201539993286f, -0.3420201539993286f, -0.8660253882408142f, -0.9848077297210693f, -0.6427876353263855f};
};
template <> struct RootTab<11> {
  static constexpr float c[11] = {1.0f, 0.8412535190582275f, 0.4154150187969208f, -0.1423148363828659f, -0.6548607349395752f, -0.9594929814338684f, -0.9594929814338684f, -0.6548607349395752f, -0.1423148363828659f, 0.4154150187969208f, 0.8412535190582275f};
  static constexpr float s[11] = {0.0f, 0.5406408309936523f, 0.9096319675445557f, 0.9898214340209961f, 0.7557495832443237f, 0.28173255920410156f, -0.28173255920410156f, -0.7557495832443237f, -0.9898214340209961f, -0.9096319675445557f, -0.5406408309936523f};
};
template <> struct RootTab<13> {
  static constexpr float c[13] = {1.0f, 0.8854560256004333f, 0.5680647492408752f, 0.1205366775393486f, -0.35460489988327026f, -0.7485107779502869f, -0.9709418416023254f, -0.9709418416023254f, -0.7485107779502869f, -0.35460489988327026f, 0.1205366775393486f, 0.5680647492408752f, 0.8854560256004333f};
  static constexpr float s[13] = {0.0f, 0.4647231698036194f, 0.8229838609695435f, 0.9927088618278503f, 0.9350162148475647f, 0.6631226539611816f, 0.23931565880775452f, -0.23931565880775452f, -0.6631226539611816f, -0.9350162148475647f, -0.9927088618278503f, -0.8229838609695435f, -0.4647231698036194f};
};
template <> struct RootTab<15> {
  static constexpr float c[15] = {1.0f, 0.9135454297065735f, 0.6691306233406067f, 0.30901700258255005f, -0.10452846437692642f, -0.5f, -0.80901700258255f, -0.9781476259231567f, -0.9781476259231567f, -0.80901700258255f, -0.5f, -0.10452846437692642f, 0.30901700258255005f, 0.6691306233406067f, 0.9135454297065735f};
  static constexpr float s[15] = {0.0f, 0.4067366421222687f, 0.7431448101997375f, 0.9510565400123596f, 0.9945219159126282f, 0.8660253882408142f, 0.5877852439880371f, 0.2079116851091385f, -0.2079116851091385f, -0.5877852439880371f, -0.8660253882408142f, -0.9945219159126282f, -0.9510565400123596f, -0.7431448101997375f, -0.4067366421222687f};
};
template <> struct RootTab<17> {
  static constexpr float c[17] = {1.0f, 0.9324722290039062f, 0.739008903503418f, 0.4457383453845978f, 0.09226836264133453f, -0.2736629843711853f, -0.602634608745575f, -0.8502171635627747f, -0.9829730987548828f, -0.9829730987548828f, -0.8502171635627747f, -0.602634608745575f, -0.2736629843711853f, 0.09226836264133453f, 0.4457383453845978f, 0.739008903503418f, 0.9324722290039062f};
  static constexpr float s[17] = {0.0f, 0.3612416684627533f, 0.6736956238746643f, 0.8951632976531982f, 0.9957341551780701f, 0.9618256688117981f, 0.7980172038078308f, 0.5264321565628052f, 0.1837495118379593f, -0.1837495118379593f, -0.5264321565628052f, -0.7980172038078308f, -0.9618256688117981f, -0.9957341551780701f, -0.8951632976531982f, -0.6736956238746643f, -0.3612416684627533f};
};
template <> struct RootTab<19> {
  static constexpr float c[19] = {1.0f, 0.945817232131958f, 0.789140522480011f, 0.5469481348991394f, 0.24548548460006714f, -0.0825793445110321f, -0.4016954302787781f, -0.6772815585136414f, -0.8794737458229065f, -0.9863613247871399f, -0.9863613247871399f, -0.8794737458229065f, -0.6772815585136414f, -0.4016954302787781f, -0.0825793445110321f, 0.24548548460006714f, 0.5469481348991394f, 0.789140522480011f, 0.945817232131958f};
  static constexpr float s[19] = {0.0f, 0.3246994614601135f, 0.614212691783905f, 0.8371664881706238f, 0.9694002866744995f, 0.9965844750404358f, 0.915773332118988f, 0.7357239127159119f, 0.47594738006591797f, 0.1645945906639099f, -0.1645945906639099f, -0.47594738006591797f, -0.7357239127159119f, -0.915773332118988f, -0.9965844750404358f, -0.9694002866744995f, -0.8371664881706238f, -0.614212691783905f, -0.3246994614601135f};
};
template <> struct RootTab<21> {
  static constexpr float c[21] = {1.0f, 0.955572783946991f, 0.826238751411438f, 0.6234897971153259f, 0.36534103751182556f, 0.07473009079694748f, -0.22252093255519867f, -0.5f, -0.7330518960952759f, -0.9009688496589661f, -0.9888308048248291f, -0.9888308048248291f, -0.9009688496589661f, -0.7330518960952759f, -0.5f, -0.22252093255519867f, 0.07473009079694748f, 0.36534103751182556f, 0.6234897971153259f, 0.826238751411438f, 0.955572783946991f};
  static constexpr float s[21] = {0.0f, 0.29475516080856323f, 0.5633200407028198f, 0.7818315029144287f, 0.9308737516403198f, 0.9972038269042969f, 0.9749279022216797f, 0.8660253882408142f, 0.6801727414131165f, 0.4338837265968323f, 0.1490422636270523f, -0.1490422636270523f, -0.4338837265968323f, -0.6801727414131165f, -0.8660253882408142f, -0.9749279022216797f, -0.9972038269042969f, -0.9308737516403198f, -0.7818315029144287f, -0.5633200407028198f, -0.29475516080856323f};
};
template <> struct RootTab<23> {
  static constexpr float c[23] = {1.0f, 0.9629172682762146f, 0.8544194102287292f, 0.6825531721115112f, 0.4600650370121002f, 0.20345601439476013f, -0.06824241578578949f, -0.334879606962204f, -0.5766803026199341f, -0.7757112979888916f, -0.9172112941741943f, -0.9906859397888184f, -0.9906859397888184f, -0.9172112941741943f, -0.7757112979888916f, -0.5766803026199341f, -0.334879606962204f, -0.06824241578578949f, 0.20345601439476013f, 0.4600650370121002f, 0.6825531721115112f, 0.8544194102287292f, 0.9629172682762146f};
  static constexpr float s[23] = {0.0f, 0.269796758890152f, 0.5195839405059814f, 0.7308359742164612f, 0.8878852128982544f, 0.9790840744972229f, 0.9976687431335449f, 0.9422609210014343f, 0.8169698715209961f, 0.6310879588127136f, 0.39840108156204224f, 0.13616664707660675f, -0.13616664707660675f, -0.39840108156204224f, -0.6310879588127136f, -0.8169698715209961f, -0.9422609210014343f, -0.9976687431335449f, -0.9790840744972229f, -0.8878852128982544f, -0.7308359742164612f, -0.5195839405059814f, -0.269796758890152f};
};
template <> struct RootTab<25> {
  static constexpr float c[25] = {1.0f, 0.9685831665992737f, 0.8763066530227661f, 0.728968620300293f, 0.5358268022537231f, 0.30901700258255005f, 0.06279052048921585f, -0.187381312251091f, -0.4257792830467224f, -0.6374239921569824f, -0.80901700258255f, -0.9297764897346497f, -0.9921147227287292f, -0.9921147227287292f, -0.9297764897346497f, -0.80901700258255f, -0.6374239921569824f, -0.4257792830467224f, -0.187381312251091f, 0.06279052048921585f, 0.30901700258255005f, 0.5358268022537231f, 0.728968620300293f, 0.8763066530227661f, 0.9685831665992737f};
  static constexpr float s[25] = {0.0f, 0.24868988990783691f, 0.4817536771297455f, 0.6845471262931824f, 0.8443279266357422f, 0.9510565400123596f, 0.9980267286300659f, 0.9822872281074524f, 0.9048270583152771f, 0.7705132365226746f, 0.5877852439880371f, 0.3681245446205139f, 0.12533323466777802f, -0.12533323466777802f, -0.3681245446205139f, -0.5877852439880371f, -0.7705132365226746f, -0.9048270583152771f, -0.9822872281074524f, -0.9980267286300659f, -0.9510565400123596f, -0.8443279266357422f, -0.6845471262931824f, -0.4817536771297455f, -0.24868988990783691f};
};
template <> struct RootTab<27> {
  static constexpr float c[27] = {1.0f, 0.9730448722839355f, 0.8936326503753662f, 0.7660444378852844f, 0.5971586108207703f, 0.39607977867126465f, 0.1736481785774231f, -0.05814483016729355f, -0.2868032455444336f, -0.5f, -0.686241626739502f, -0.8354877829551697f, -0.9396926164627075f, -0.9932383298873901f, -0.9932383298873901f, -0.9396926164627075f, -0.8354877829551697f, -0.686241626739502f, -0.5f, -0.2868032455444336f, -0.05814483016729355f, 0.1736481785774231f, 0.39607977867126465f, 0.5971586108207703f, 0.7660444378852844f, 0.8936326503753662f, 0.9730448722839355f};
  static constexpr float s[27] = {0.0f, 0.23061586916446686f, 0.448799192905426f, 0.6427876353263855f, 0.8021231889724731f, 0.9182161092758179f, 0.9848077297210693f, 0.9983081817626953f, 0.957989513874054f, 0.8660253882408142f, 0.7273736596107483f, 0.5495089888572693f, 0.3420201539993286f, 0.11609291285276413f, -0.11609291285276413f, -0.3420201539993286f, -0.5495089888572693f, -0.7273736596107483f, -0.8660253882408142f, -0.957989513874054f, -0.9983081817626953f, -0.9848077297210693f, -0.9182161092758179f, -0.8021231889724731f, -0.6427876353263855f, -0.448799192905426f, -0.23061586916446686f};
};
template <> struct RootTab<29> {
  static constexpr float c[29] = {1.0f, 0.9766205549240112f, 0.9075754284858704f, 0.7960930466651917f, 0.6473863124847412f, 0.4684084355831146f, 0.26752832531929016f, 0.0541389100253582f, -0.16178199648857117f, -0.37013816833496094f, -0.5611870884895325f, -0.7259954810142517f, -0.856857180595398f, -0.9476531744003296f, -0.9941379427909851f, -0.9941379427909851f, -0.9476531744003296f, -0.856857180595398f, -0.7259954810142517f, -0.5611870884895325f, -0.37013816833496094f, -0.16178199648857117f, 0.0541389100253582f, 0.26752832531929016f, 0.4684084355831146f, 0.6473863124847412f, 0.7960930466651917f, 0.9075754284858704f, 0.9766205549240112f};
  static constexpr float s[29] = {0.0f, 0.2149704396724701f, 0.41988909244537354f, 0.6051742434501648f, 0.7621620297431946f, 0.883512020111084f, 0.9635499715805054f, 0.9985334277153015f, 0.9868265390396118f, 0.9289767146110535f, 0.827688992023468f, 0.6876994371414185f, 0.5155538320541382f, 0.3193015158176422f, 0.10811901837587357f, -0.10811901837587357f, -0.3193015158176422f, -0.5155538320541382f, -0.6876994371414185f, -0.827688992023468f, -0.9289767146110535f, -0.9868265390396118f, -0.9985334277153015f, -0.9635499715805054f, -0.883512020111084f, -0.7621620297431946f, -0.6051742434501648f, -0.41988909244537354f, -0.2149704396724701f};
};
template <> struct RootTab<31> {
  static constexpr float c[31] = {1.0f, 0.9795299172401428f, 0.9189578294754028f, 0.8207634687423706f, 0.6889669299125671f, 0.5289639830589294f, 0.3473052382469177f, 0.15142777562141418f, -0.05064916983246803f, -0.2506525218486786f, -0.44039416313171387f, -0.6121059656143188f, -0.7587581276893616f, -0.8743466138839722f, -0.954139232635498f, -0.9948693513870239f, -0.9948693513870239f, -0.954139232635498f, -0.8743466138839722f, -0.7587581276893616f, -0.6121059656143188f, -0.44039416313171387f, -0.2506525218486786f, -0.05064916983246803f, 0.15142777562141418f, 0.3473052382469177f, 0.5289639830589294f, 0.6889669299125671f, 0.8207634687423706f, 0.9189578294754028f, 0.9795299172401428f};
  static constexpr float s[31] = {0.0f, 0.2012985199689865f, 0.3943558633327484f, 0.5712682008743286f, 0.7247927784919739f, 0.8486442565917969f, 0.9377521276473999f, 0.98846834897995f, 0.9987165331840515f, 0.9680771231651306f, 0.8978045582771301f, 0.790775716304779f, 0.651372492313385f, 0.4853019714355469f, 0.2993631362915039f, 0.10116831958293915f, -0.10116831958293915f, -0.2993631362915039f, -0.4853019714355469f, -0.651372492313385f, -0.790775716304779f, -0.8978045582771301f, -0.9680771231651306f, -0.9987165331840515f, -0.98846834897995f, -0.9377521276473999f, -0.8486442565917969f, -0.7247927784919739f, -0.5712682008743286f, -0.3943558633327484f, -0.2012985199689865f};
};
// odd P: X_k = x0 + sum_j cos(2 pi jk/P) (x_j + x_{P-j}) -/+ i sum_j sin(2 pi jk/P) (x_j - x_{P-j}),  j = 1..(P-1)/2 --
// (P-1)^2 real multiply-adds; k and P - k share both sums
template <bool INV, int P>
__device__ __forceinline__ void dft_odd(float (&r)[P], float (&i)[P]) {
  constexpr int H = (P - 1) / 2;
  float tr[H + 1], ti[H + 1], ur[H + 1], ui[H + 1];
#pragma unroll
  for (int j = 1; j <= H; ++j) {
    tr[j] = r[j] + r[P - j]; ti[j] = i[j] + i[P - j];
    ur[j] = r[j] - r[P - j]; ui[j] = i[j] - i[P - j];
  }
  const float x0r = r[0], x0i = i[0];
  float sr = x0r, si = x0i;
#pragma unroll
  for (int j = 1; j <= H; ++j) { sr += tr[j]; si += ti[j]; }
  r[0] = sr; i[0] = si;
#pragma unroll
  for (int k = 1; k <= H; ++k) {
    float mr = x0r, mi = x0i, vr = 0.0f, vi = 0.0f;
#pragma unroll
    for (int j = 1; j <= H; ++j) {
      const float c = RootTab<P>::c[(j * k) % P], sn = RootTab<P>::s[(j * k) % P];
      mr = fmaf(c, tr[j], mr); mi = fmaf(c, ti[j], mi);
      vr = fmaf(sn, ur[j], vr); vi = fmaf(sn, ui[j], vi);
    }
    if constexpr (!INV) {   // X_k = m - i v,  X_{P-k} = m + i v
      r[k] = mr + vi; i[k] = mi - vr;
      r[P - k] = mr - vi; i[P - k] = mi + vr;
    } else {
      r[k] = mr - vi; i[k] = mi + vr;
      r[P - k] = mr + vi; i[P - k] = mi - vr;
    }
  }
}
template <bool INV> __device__ __forceinline__ void dft_small(float (&r)[7], float (&i)[7]) { dft_odd<INV, 7>(r, i); }
template <bool INV> __device__ __forceinline__ void dft_small(float (&r)[9], float (&i)[9]) { dft_odd<INV, 9>(r, i); }
template <bool INV> __device__ __forceinline__ void dft_small(float (&r)[11], float (&i)[11]) { dft_odd<INV, 11>(r, i); }
template <bool INV> __device__ __forceinline__ void dft_small(float (&r)[13], float (&i)[13]) { dft_odd<INV, 13>(r, i); }
template <bool INV> __device__ __forceinline__ void dft_small(float (&r)[17], float (&i)[17]) { dft_odd<INV, 17>(r, i); }
template <bool INV> __device__ __forceinline__ void dft_small(float (&r)[19], float (&i)[19]) { dft_odd<INV, 19>(r, i); }
template <bool INV> __device__ __forceinline__ void dft_small(float (&r)[23], float (&i)[23]) { dft_odd<INV, 23>(r, i); }
template <bool INV> __device__ __forceinline__ void dft_small(float (&r)[25], float (&i)[25]) { dft_odd<INV, 25>(r, i); }
template <bool INV> __device__ __forceinline__ void dft_small(float (&r)[27], float (&i)[27]) { dft_odd<INV, 27>(r, i); }
template <bool INV> __device__ __forceinline__ void dft_small(float (&r)[29], float (&i)[29]) { dft_odd<INV, 29>(r, i); }
template <bool INV> __device__ __forceinline__ void dft_small(float (&r)[31], float (&i)[31]) { dft_odd<INV, 31>(r, i); }
template <bool INV> __device__ __forceinline__ void fft_pow2(float (&r)[2], float (&i)[2]) {
  const float ar = r[0] + r[1], ai = i[0] + i[1];
  r[1] = r[0] - r[1]; i[1] = i[0] - i[1];
  r[0] = ar; i[0] = ai;
}
template <bool INV> __device__ __forceinline__ void fft_pow2(float (&r)[4], float (&i)[4]) { bfly4<INV>(r[0], i[0], r[1], i[1], r[2], i[2], r[3], i[3]); }
template <bool INV> __device__ __forceinline__ void fft_pow2(float (&r)[8], float (&i)[8]) { bfly8<INV>(r, i); }
template <bool INV> __device__ __forceinline__ void fft_pow2(float (&r)[16], float (&i)[16]) { fft16<INV>(r, i); }

template <bool INV> __device__ __forceinline__ void fft_pow2(float (&r)[5], float (&i)[5]) { dft_small<INV>(r, i); }   // second factor of 15
template <bool INV> __device__ __forceinline__ void fft_pow2(float (&r)[7], float (&i)[7]) { dft_small<INV>(r, i); }   // ... of 21

template <bool INV> __device__ __forceinline__ void dft_small(float (&r)[15], float (&i)[15]);   // nested splits, defined below
template <bool INV> __device__ __forceinline__ void dft_small(float (&r)[21], float (&i)[21]);
constexpr int pfa_unit(int n_self, int n_other) {   // e = 1 mod n_self, 0 mod n_other
  for (int e = n_other; e < n_self * n_other; e += n_other)
    if (e % n_self == 1) return e;
  return 0;
}

template <bool INV, int P, int N2>
__device__ __forceinline__ void fft_pfa(float (&xr)[P * N2], float (&xi)[P * N2]) {
  constexpr int N = P * N2;
  constexpr int e1 = pfa_unit(P, N2), e2 = pfa_unit(N2, P);
  float yr[N], yi[N];   // slot (k1, n2) at N2 k1 + n2
#pragma unroll
  for (int n2 = 0; n2 < N2; ++n2) {
    float r[P], i[P];
#pragma unroll
    for (int n1 = 0; n1 < P; ++n1) { r[n1] = xr[(N2 * n1 + P * n2) % N]; i[n1] = xi[(N2 * n1 + P * n2) % N]; }
    dft_small<INV>(r, i);
#pragma unroll
    for (int k1 = 0; k1 < P; ++k1) { yr[N2 * k1 + n2] = r[k1]; yi[N2 * k1 + n2] = i[k1]; }
  }
#pragma unroll
  for (int k1 = 0; k1 < P; ++k1) {
    float r[N2], i[N2];
#pragma unroll
    for (int n2 = 0; n2 < N2; ++n2) { r[n2] = yr[N2 * k1 + n2]; i[n2] = yi[N2 * k1 + n2]; }
    fft_pow2<INV>(r, i);
#pragma unroll
    for (int k2 = 0; k2 < N2; ++k2) { xr[(e1 * k1 + e2 * k2) % N] = r[k2]; xi[(e1 * k1 + e2 * k2) % N] = i[k2]; }
  }
}

// 15 = 3 x 5 and 21 = 3 x 7 are coprime products themselves: the same split once more instead of the (P-1)^2 form
template <bool INV> __device__ __forceinline__ void dft_small(float (&r)[15], float (&i)[15]) { fft_pfa<INV, 3, 5>(r, i); }
template <bool INV> __device__ __forceinline__ void dft_small(float (&r)[21], float (&i)[21]) { fft_pfa<INV, 3, 7>(r, i); }

}  // namespace lspiv
