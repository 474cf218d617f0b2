// Wave-level helpers shared by the tensor-product kernels (gfx950, wave64).
#pragma once
#include "aa_common.h"

namespace aa {

// index of the irrep (l) a component of a spherical-harmonics-ordered vector belongs to: 0 | 1..3 | 4..8 | 9..15
template <int LMAX>
__device__ __forceinline__ constexpr int r_of(int i) {
  return i < 1 ? 0 : (i < 4 ? 1 : (i < 9 ? 2 : 3));
}

// 2-vector of T: the packed two-edge evaluation instantiates the straight-line CG code on these
template <typename T>
struct Pk;
template <>
struct Pk<float> {
  typedef float type __attribute__((ext_vector_type(2)));
};
template <>
struct Pk<double> {
  typedef double type __attribute__((ext_vector_type(2)));
};

// Sum D (<=16) per-lane values over the 64 lanes of a wave with a reduce-scatter butterfly
// (8+4+2+1+1+1 = 17 exchange steps instead of 6*D) and store total k to dst[k].
// fp32: the exchanges are v_permlane32/16_swap (no selects needed: the swap leaves each half holding exactly the
// two addends it keeps) and DPP-fused adds -- ~35 VALU instructions, no LDS.  fp64: shuffles.
template <typename T, int D>
__device__ __forceinline__ void wave_sum_store(const T* v, T* dst, bool act, bool atomic) {
  static_assert(D <= 16, "at most 16 values");
  const int lane = threadIdx.x & 63;
  T x[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) x[k] = k < D ? v[k] : T(0);
  T r;
  if constexpr (sizeof(T) == 4) {
    float y[8], z[4], q[2];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      float a = x[k], b = x[k + 8];
      permlane32_swap(a, b);  // lower half: {x_lo[k], x_hi[k]}, upper half: {x_lo[k+8], x_hi[k+8]}
      y[k] = a + b;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float a = y[k], b = y[k + 4];
      permlane16_swap(a, b);
      z[k] = a + b;
    }
    {
      const bool hi = lane & 8;
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const float send = hi ? z[k] : z[k + 2], keep = hi ? z[k + 2] : z[k];
        q[k] = keep + dpp_move<kDppRowRor8>(send);
      }
    }
    {
      const bool hi = lane & 4;
      const float send = hi ? q[0] : q[1], keep = hi ? q[1] : q[0];
      r = keep + dpp_move<kDppHalfMirror>(send);
    }
    r += dpp_move<kDppQuad1032>(r);
    r += dpp_move<kDppQuad2301>(r);
  } else {
    T y[8], z[4], q[2];
    {
      const bool hi = lane & 32;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        T recv = __shfl_xor(hi ? x[k] : x[k + 8], 32);
        y[k] = (hi ? x[k + 8] : x[k]) + recv;
      }
    }
    {
      const bool hi = lane & 16;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        T recv = __shfl_xor(hi ? y[k] : y[k + 4], 16);
        z[k] = (hi ? y[k + 4] : y[k]) + recv;
      }
    }
    {
      const bool hi = lane & 8;
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        T recv = __shfl_xor(hi ? z[k] : z[k + 2], 8);
        q[k] = (hi ? z[k + 2] : z[k]) + recv;
      }
    }
    {
      const bool hi = lane & 4;
      T recv = __shfl_xor(hi ? q[0] : q[1], 4);
      r = (hi ? q[1] : q[0]) + recv;
    }
    r += __shfl_xor(r, 2);
    r += __shfl_xor(r, 1);
  }
  const int idx = ((lane >> 5) & 1) * 8 + ((lane >> 4) & 1) * 4 + ((lane >> 3) & 1) * 2 + ((lane >> 2) & 1);
  if ((lane & 3) == 0 && idx < D && act) {
    if (atomic)
      atomicAdd(&dst[idx], r);
    else
      dst[idx] = r;
  }
}

// Two independent D-value sums (the two edges of a pair) in one pass: the butterflies are interleaved so that each
// one's dependent exchange -> add chain hides the other's latency, and the permlane swaps are issued four to a block.
template <typename T, int D>
__device__ __forceinline__ void wave_sum_store2(const T* va, const T* vb, T* dsta, T* dstb, bool acta, bool actb) {
  static_assert(D <= 16, "at most 16 values");
  if constexpr (sizeof(T) != 4) {
    wave_sum_store<T, D>(va, dsta, acta, false);
    wave_sum_store<T, D>(vb, dstb, actb, false);
  } else {
    const int lane = threadIdx.x & 63;
    float xa[16], xb[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      xa[k] = k < D ? va[k] : 0.f;
      xb[k] = k < D ? vb[k] : 0.f;
    }
    float ya[8], yb[8], za[4], zb[4], qa[2], qb[2];
    // bit 5: {x[k], x[k+8]} -> y[k]
    permlane32_swap4(xa, xa + 8);
    permlane32_swap4(xb, xb + 8);
    permlane32_swap4(xa + 4, xa + 12);
    permlane32_swap4(xb + 4, xb + 12);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      ya[k] = xa[k] + xa[k + 8];
      yb[k] = xb[k] + xb[k + 8];
    }
    // bit 4
    permlane16_swap4(ya, ya + 4);
    permlane16_swap4(yb, yb + 4);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      za[k] = ya[k] + ya[k + 4];
      zb[k] = yb[k] + yb[k + 4];
    }
    {
      const bool hi = lane & 8;
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const float sa = hi ? za[k] : za[k + 2], ka_ = hi ? za[k + 2] : za[k];
        const float sb = hi ? zb[k] : zb[k + 2], kb_ = hi ? zb[k + 2] : zb[k];
        qa[k] = ka_ + dpp_move<kDppRowRor8>(sa);
        qb[k] = kb_ + dpp_move<kDppRowRor8>(sb);
      }
    }
    float ra, rb;
    {
      const bool hi = lane & 4;
      const float sa = hi ? qa[0] : qa[1], ka_ = hi ? qa[1] : qa[0];
      const float sb = hi ? qb[0] : qb[1], kb_ = hi ? qb[1] : qb[0];
      ra = ka_ + dpp_move<kDppHalfMirror>(sa);
      rb = kb_ + dpp_move<kDppHalfMirror>(sb);
    }
    ra += dpp_move<kDppQuad1032>(ra);
    rb += dpp_move<kDppQuad1032>(rb);
    ra += dpp_move<kDppQuad2301>(ra);
    rb += dpp_move<kDppQuad2301>(rb);
    const int idx = ((lane >> 5) & 1) * 8 + ((lane >> 4) & 1) * 4 + ((lane >> 3) & 1) * 2 + ((lane >> 2) & 1);
    if ((lane & 3) == 0 && idx < D) {
      if (acta) dsta[idx] = ra;
      if (actb) dstb[idx] = rb;
    }
  }
}


}  // namespace aa
