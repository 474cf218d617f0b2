// bf16x3 split-precision MFMA building blocks shared by the linear-layer kernels (aa_gemm.hip) and the fused
// per-atom-tile kernels (aa_fused.hip).  See aa_gemm.hip for the arithmetic ("fp32 GEMM on the bf16 matrix cores by
// exact 3-way splitting") and the fragment layouts.
#pragma once
#include "aa_common.h"

namespace aa {

typedef float v16f __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned f2u(float x) { return __builtin_bit_cast(unsigned, x); }
__device__ __forceinline__ float u2f(unsigned x) { return __builtin_bit_cast(float, x); }

// 16 floats -> three levels of 8 packed bf16 pairs (element 2q in the low half, 2q+1 in the high half)
__device__ __forceinline__ void split3_pack(const v4f* a, u32x4* lv1, u32x4* lv2, u32x4* lv3) {
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    u32x4 o1, o2, o3;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      unsigned h1[2], h2[2], h3[2];
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int idx = half * 8 + q * 2 + e;
        const float x = a[idx >> 2][idx & 3];
        h1[e] = f2u(x) & 0xFFFF0000u;
        const float r = x - u2f(h1[e]);
        h2[e] = f2u(r) & 0xFFFF0000u;
        const float r2 = r - u2f(h2[e]);
        h3[e] = f2u(r2) & 0xFFFF0000u;
      }
      o1[q] = (h1[0] >> 16) | h1[1];
      o2[q] = (h2[0] >> 16) | h2[1];
      o3[q] = (h3[0] >> 16) | h3[1];
    }
    lv1[half] = o1;
    lv2[half] = o2;
    lv3[half] = o3;
  }
}

__device__ __forceinline__ v16f mma_bf16(const u32x4& w, const u32x4& x, v16f acc) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, w), __builtin_bit_cast(bf16x8, x), acc, 0, 0, 0);
}

// a 32-feature activation tile in accumulator layout, split into its three bf16 levels (= the MFMA B operand of one
// 32-deep k chunk of the next layer: the weights' k order is the accumulator order)
struct XSplit {
  u32x4 l1[2], l2[2], l3[2];
};
__device__ __forceinline__ void xsplit_from_acc(const v16f& acc, XSplit& x) {
  v4f a[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) a[q] = v4f{acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]};
  split3_pack(a, x.l1, x.l2, x.l3);
}

constexpr int kWStep = 2 * 6 * 64;  // u32x4 per staged weight step (tile pair x 32-deep chunk x 3 levels)

}  // namespace aa
