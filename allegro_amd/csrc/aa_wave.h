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
    {
      // lower half: {x_lo[k], x_hi[k]}, upper half: {x_lo[k+8], x_hi[k+8]}; swaps in blocks of four (one pair of hazard nops per block)
      float a[8], b[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        a[k] = x[k];
        b[k] = x[k + 8];
      }
      permlane32_swap4(a, b);
      permlane32_swap4(a + 4, b + 4);
#pragma unroll
      for (int k = 0; k < 8; ++k) y[k] = a[k] + b[k];
    }
    {
      float a[4], b[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        a[k] = y[k];
        b[k] = y[k + 4];
      }
      permlane16_swap4(a, b);
#pragma unroll
      for (int k = 0; k < 4; ++k) z[k] = a[k] + b[k];
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

// N independent v_permlane32/16_swap of (a[i], b[i]) in blocks of four (permlane*_swap4: one pair of hazard nops per block)
template <int W, int N>
__device__ __forceinline__ void permlane_swap_n(float* a, float* b) {
  constexpr int N4 = N & ~3;
#pragma unroll
  for (int i = 0; i < N4; i += 4) {
    if constexpr (W == 32)
      permlane32_swap4(a + i, b + i);
    else
      permlane16_swap4(a + i, b + i);
  }
#pragma unroll
  for (int i = N4; i < N; ++i) {
    if constexpr (W == 32)
      permlane32_swap(a[i], b[i]);
    else
      permlane16_swap(a[i], b[i]);
  }
}

// One level of a reduce-scatter over the lane bit BIT: of the N values a lane holds it keeps ceil(N / 2) (the lower ones if its
// bit is clear, the upper ones -- padded with a zero when N is odd -- if it is set) and adds its partner's partial sums of the
// same values.  N == 1: a plain sum over the bit.  Bits 5 / 4: v_permlane32/16_swap (the swap leaves each half holding exactly
// the two addends it keeps); bits 3..0: DPP-fused adds of the value selected for sending.
template <int BIT, int N>
__device__ __forceinline__ void scatter_level(const float* in, float* out, int lane) {
  constexpr int H = (N + 1) / 2;
  if constexpr (N == 1) {
    if constexpr (BIT >= 4) {
      out[0] = in[0] + __shfl_xor(in[0], 1 << BIT);
    } else if constexpr (BIT == 3) {
      out[0] = in[0] + dpp_move<kDppRowRor8>(in[0]);
    } else if constexpr (BIT == 2) {
      out[0] = in[0] + dpp_move<kDppHalfMirror>(in[0]);
    } else if constexpr (BIT == 1) {
      out[0] = in[0] + dpp_move<kDppQuad2301>(in[0]);
    } else {
      out[0] = in[0] + dpp_move<kDppQuad1032>(in[0]);
    }
  } else if constexpr (BIT >= 4) {
    float sa[H], sb[H];
#pragma unroll
    for (int i = 0; i < H; ++i) {
      sa[i] = in[i];
      sb[i] = i + H < N ? in[i + H] : 0.f;
    }
    permlane_swap_n<(BIT == 5 ? 32 : 16), H>(sa, sb);  // (blocks of four swaps: see wave_sum_store2)
#pragma unroll
    for (int i = 0; i < H; ++i) out[i] = sa[i] + sb[i];
  } else {
    const bool hi = lane & (1 << BIT);
#pragma unroll
    for (int i = 0; i < H; ++i) {
      const float lo_v = in[i], hi_v = i + H < N ? in[i + H] : 0.f;
      const float send = hi ? lo_v : hi_v, keep = hi ? hi_v : lo_v;
      if constexpr (BIT == 3)
        out[i] = keep + dpp_move<kDppRowRor8>(send);
      else if constexpr (BIT == 2)
        out[i] = keep + dpp_move<kDppHalfMirror>(send);
      else if constexpr (BIT == 1)
        out[i] = keep + dpp_move<kDppQuad2301>(send);
      else
        out[i] = keep + dpp_move<kDppQuad1032>(send);
    }
  }
}

// Two independent D-value sums (the two edges of a pair) in ONE reduce-scatter of 2 D values: bit 5 separates the edges (the
// swap pairs value i of edge a with value i of edge b), every further bit halves what a lane holds -- 9 + 5 + 3 + 2 + 1 + 1
// exchanges for D = 9 instead of 2 x (8 + 4 + 2 + 1 + 1 + 1) with each edge padded to 16 values.  Total j of its edge ends
// up in the lanes whose bits 4..0 spell j in the mixed radix of the level sizes; one lane per total stores it.
template <typename T, int D>
__device__ __forceinline__ void wave_sum_store2(const T* va, const T* vb, T* dsta, T* dstb, bool acta, bool actb) {
  static_assert(D <= 16, "at most 16 values");
  if constexpr (sizeof(T) != 4) {
    wave_sum_store<T, D>(va, dsta, acta, false);
    wave_sum_store<T, D>(vb, dstb, actb, false);
  } else {
    const int lane = threadIdx.x & 63;
    constexpr int N4 = D, N3 = (N4 + 1) / 2, N2 = (N3 + 1) / 2, N1 = (N2 + 1) / 2, N0 = (N1 + 1) / 2;
    float x4[N4], x3[N3], x2[N2], x1[N1], x0[N0], r[1];
    {
      // lower half: both halves' partial sums of edge a, upper half: of edge b.  The swaps go in blocks of four (one pair of hazard
      // nops per block instead of per swap: 6 instead of 18 s_nop issue slots per edge pair at D = 9)
      float sa[D], sb[D];
#pragma unroll
      for (int i = 0; i < D; ++i) {
        sa[i] = va[i];
        sb[i] = vb[i];
      }
      permlane_swap_n<32, D>(sa, sb);
#pragma unroll
      for (int i = 0; i < D; ++i) x4[i] = sa[i] + sb[i];
    }
    scatter_level<4, N4>(x4, x3, lane);
    scatter_level<3, N3>(x3, x2, lane);
    scatter_level<2, N2>(x2, x1, lane);
    scatter_level<1, N1>(x1, x0, lane);
    scatter_level<0, N0>(x0, r, lane);
    // which total the lane holds: at a level that still had N > 1 values, a set bit selected the upper part (index offset =
    // size of the lower part H, and N - H values, one fewer than H when N is odd: the pad); at a level with one value the
    // lanes of both bit values hold copies and the clear one owns the store
    int j = 0, cnt = D;
    bool owner = true;
    auto level = [&](bool bit, int n, int h) {
      if (n > 1) {
        if (bit) {
          j += h;
          cnt -= h;
        } else {
          cnt = cnt < h ? cnt : h;
        }
      } else {
        owner = owner && !bit;
      }
    };
    level(lane & 16, N4, N3);
    level(lane & 8, N3, N2);
    level(lane & 4, N2, N1);
    level(lane & 2, N1, N0);
    level(lane & 1, N0, (N0 + 1) / 2);
    if (owner && cnt >= 1) {
      if (lane & 32) {
        if (actb) dstb[j] = r[0];
      } else {
        if (acta) dsta[j] = r[0];
      }
    }
  }
}


// streamed-once operands of the per-atom edge loops (moments / operator kernels) carry the non-temporal hint: each row is read by exactly one wave and
// written rows are not read again before they have left the caches (C4: 2-4 % per kernel, profiles/archive/r02_v12_nt_stages_*;
// tools/ubench/hbm_stream.hip: +5-10 % for this one-row-per-instruction pattern; -DAA_NO_NT builds without)
template <typename T>
__device__ __forceinline__ T ld_stream(const T* p) {
#ifndef AA_NO_NT
  return __builtin_nontemporal_load(p);
#else
  return *p;
#endif
}
template <typename T>
__device__ __forceinline__ void st_stream(T* p, T v) {
#ifndef AA_NO_NT
  __builtin_nontemporal_store(v, p);
#else
  *p = v;
#endif
}

}  // namespace aa
