// Per-edge geometry helpers shared by the edge prologue / reverse kernels (aa_edge.hip) and the fused per-atom-tile
// kernels (aa_fused.hip): real spherical harmonics and their gradient, polynomial cutoff, spline basis.
#pragma once
#include "aa_common.h"

namespace aa {

// --------------------------------------------------------------------------------------------
// real spherical harmonics, component normalised, y polar, m = -l..l (same polynomials as
// oracle/restatement.py; l <= 3)
// --------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ void sh_eval(int l_max, T x, T y, T z, T* Y) {
  Y[0] = T(1);
  if (l_max >= 1) {
    const T s3 = T(1.7320508075688772);
    Y[1] = s3 * x;
    Y[2] = s3 * y;
    Y[3] = s3 * z;
  }
  if (l_max >= 2) {
    const T s15 = T(3.872983346207417), s5 = T(2.23606797749979);
    Y[4] = s15 * x * z;
    Y[5] = s15 * x * y;
    Y[6] = s5 * (y * y - T(0.5) * (x * x + z * z));
    Y[7] = s15 * y * z;
    Y[8] = T(0.5) * s15 * (z * z - x * x);
  }
  if (l_max >= 3) {
    const T c70 = T(2.091650066335189), c105 = T(10.246950765959598), c42 = T(1.620185174601965),
            c7 = T(1.3228756555322954);
    Y[9] = c70 * x * (T(3) * z * z - x * x);
    Y[10] = c105 * x * y * z;
    Y[11] = c42 * x * (T(5) * y * y - T(1));
    Y[12] = c7 * y * (T(5) * y * y - T(3));
    Y[13] = c42 * z * (T(5) * y * y - T(1));
    Y[14] = T(0.5) * c105 * y * (z * z - x * x);
    Y[15] = c70 * z * (z * z - T(3) * x * x);
  }
}

// g = sum_i gY[i] * dY_i/d(x,y,z)  (polynomials as written; the caller projects onto the tangent plane)
template <typename T>
__device__ __forceinline__ void sh_grad(int l_max, T x, T y, T z, const T* gY, T& gx, T& gy, T& gz) {
  gx = gy = gz = T(0);
  if (l_max >= 1) {
    const T s3 = T(1.7320508075688772);
    gx += s3 * gY[1];
    gy += s3 * gY[2];
    gz += s3 * gY[3];
  }
  if (l_max >= 2) {
    const T s15 = T(3.872983346207417), s5 = T(2.23606797749979);
    gx += s15 * z * gY[4] + s15 * y * gY[5] - s5 * x * gY[6] - s15 * x * gY[8];
    gy += s15 * x * gY[5] + T(2) * s5 * y * gY[6] + s15 * z * gY[7];
    gz += s15 * x * gY[4] - s5 * z * gY[6] + s15 * y * gY[7] + s15 * z * gY[8];
  }
  if (l_max >= 3) {
    const T c70 = T(2.091650066335189), c105 = T(10.246950765959598), c42 = T(1.620185174601965),
            c7 = T(1.3228756555322954);
    T a = T(5) * y * y - T(1);
    gx += c70 * T(3) * (z * z - x * x) * gY[9] + c105 * y * z * gY[10] + c42 * a * gY[11] - c105 * x * y * gY[14] -
          c70 * T(6) * x * z * gY[15];
    gy += c105 * x * z * gY[10] + c42 * T(10) * x * y * gY[11] + c7 * (T(15) * y * y - T(3)) * gY[12] +
          c42 * T(10) * y * z * gY[13] + T(0.5) * c105 * (z * z - x * x) * gY[14];
    gz += c70 * T(6) * x * z * gY[9] + c105 * x * y * gY[10] + c42 * a * gY[13] + c105 * y * z * gY[14] +
          c70 * T(3) * (z * z - x * x) * gY[15];
  }
}

template <typename T>
__device__ __forceinline__ void cutoff_and_grad(T x, T p, T& f, T& df) {
  if (x < T(1)) {
    T a = (p + T(1)) * (p + T(2)) * T(0.5), b = p * (p + T(2)), c = p * (p + T(1)) * T(0.5);
    T xp1 = aa_pow(x, p - T(1));  // x^(p-1)
    T xp = xp1 * x, xq = xp * x, xr = xq * x;
    f = T(1) - a * xp + b * xq - c * xr;
    df = -a * p * xp1 + b * (p + T(1)) * xp - c * (p + T(2)) * xq;
  } else {
    f = T(0);
    df = T(0);
  }
}

constexpr int kMaxBessel = 16;

// spline basis of PerClassSpline._get_basis (allegro/nn/spline.py:81-89):
//   b_s(x) = 0.25 (1 - cos(k (clamp(x, lo_s, lo_s + diff) - lo_s)))^2,  lo_s = (s - span)/n, diff = (span+1)/n, k = 2 pi/diff
// value and d/dx (zero where the clamp is active)
template <typename T>
__device__ __forceinline__ void spline_basis_and_grad(T x, int s, int n, int span, T& bv, T& dbv) {
  const T lo = T(s - span) / T(n), diff = T(span + 1) / T(n);
  const T k = T(6.283185307179586476925286766559) / diff;
  const T xc = x < lo ? lo : (x > lo + diff ? lo + diff : x);
  const T th = k * (xc - lo);
  const T omc = T(1) - aa_cos(th);
  bv = T(0.25) * omc * omc;
  dbv = (x > lo && x < lo + diff) ? T(0.5) * omc * aa_sin(th) * k : T(0);
}


}  // namespace aa
