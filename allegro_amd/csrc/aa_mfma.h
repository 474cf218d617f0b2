// bf16x3 split-precision MFMA building blocks shared by the linear-layer kernels (aa_gemm.hip) and the fused
// per-atom-tile kernels (aa_fused.hip).  See aa_gemm.hip for the arithmetic ("fp32 GEMM on the bf16 matrix cores by
// exact 3-way splitting") and the fragment layouts.
#pragma once
#include <type_traits>

#include "aa_common.h"

namespace aa {

typedef float v16f __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned f2u(float x) { return __builtin_bit_cast(unsigned, x); }
__device__ __forceinline__ float u2f(unsigned x) { return __builtin_bit_cast(float, x); }

// the high halves of two words as one word: {hi[31:16], lo[31:16]} (one v_perm_b32)
__device__ __forceinline__ unsigned pack_high_halves(unsigned lo, unsigned hi) { return __builtin_amdgcn_perm(hi, lo, 0x07060302u); }

// 16 floats -> three levels of 8 packed bf16 pairs (element 2q in the low half, 2q+1 in the high half).  Level l holds the
// top 16 bits of what levels < l left over; the pairs are packed straight from the unmasked words (the mask is only needed
// for the value that is subtracted), 5.5 instead of 8 VALU operations per float -- the splits are a third of the fused
// kernels' vector instructions.
__device__ __forceinline__ void split3_pack(const v4f* a, u32x4* lv1, u32x4* lv2, u32x4* lv3) {
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    u32x4 o1, o2, o3;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int i0 = half * 8 + q * 2, i1 = i0 + 1;
      const float x0 = a[i0 >> 2][i0 & 3], x1 = a[i1 >> 2][i1 & 3];
      o1[q] = pack_high_halves(f2u(x0), f2u(x1));
      const float r0 = x0 - u2f(f2u(x0) & 0xFFFF0000u), r1 = x1 - u2f(f2u(x1) & 0xFFFF0000u);
      o2[q] = pack_high_halves(f2u(r0), f2u(r1));
      const float s0 = r0 - u2f(f2u(r0) & 0xFFFF0000u), s1 = r1 - u2f(f2u(r1) & 0xFFFF0000u);
      o3[q] = pack_high_halves(f2u(s0), f2u(s1));
    }
    lv1[half] = o1;
    lv2[half] = o2;
    lv3[half] = o3;
  }
}

// the same split with the pairs packed from the MASKED words (shift + or): 8 operations per float, but lower register
// pressure in the scheduler's hands -- the chain kernel's PRE variants (255 of 256 VGPRs at two waves per SIMD) spill 35-37
// registers with the form above and lose 60 % (gc_64x64_64x128 at C4: 1.01 -> 1.62 ms), so they keep this one
__device__ __forceinline__ void split3_pack_masked(const v4f* a, u32x4* lv1, u32x4* lv2, u32x4* lv3) {
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

// workgroup barrier that orders LDS traffic only: unlike __syncthreads() it carries no vmcnt(0), so global loads that
// were requested ahead of their use (operand rows, weight blocks) stay in flight across it
__device__ __forceinline__ void lds_barrier() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

constexpr int kWStep = 2 * 6 * 64;  // u32x4 per staged weight step (tile pair x 32-deep chunk x 3 levels)

// ---- helpers shared by the kernels that work on 32-edge tiles in the accumulator layout (aa_fused.hip, aa_tp_mfma.hip,
//      the tensor-track epilogue of gemm_chain_bf16x3_kernel)
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

constexpr int kTileLdT = 36;  // row stride (floats) of the wave-private [32][32] store-transpose patch: 32 + 4 keeps b128 accesses conflict-light

// store one 32-feature tile (accumulator layout) to rows [row0, row0 + cnt) of a row-major [E, ld] array through the
// wave-private transpose patch sT [32][kTileLdT], so that every store instruction writes whole 128-B lines
__device__ __forceinline__ void tile_store_rows(float* sT, const v16f& acc, float* dst, int64_t row0, int cnt, int ld, int lane) {
  const int el = lane & 31, hh = lane >> 5;
  float* st = sT + el * kTileLdT + 4 * hh;
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int q = 0; q < 4; ++q) *reinterpret_cast<v4f*>(st + 8 * q) = v4f{acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]};
  __builtin_amdgcn_wave_barrier();
  const int pr = lane >> 3, pc = 4 * (lane & 7);
  v4f v[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) v[q] = *reinterpret_cast<const v4f*>(sT + (8 * q + pr) * kTileLdT + pc);
  __builtin_amdgcn_wave_barrier();
  float* p = dst + (row0 + pr) * ld + pc;
#pragma unroll
  for (int q = 0; q < 4; ++q)
    if (pr + 8 * q < cnt) *reinterpret_cast<v4f*>(p + int64_t(8 * q) * ld) = v[q];
}

// Tensor-track scalars of a tile:  s[e][ch] += w[e][r][ch] * sum_{a in irrep RR} Y[e][a] * B[a][ch]  for the tile pair
// (w0a: channels 0..31, w0b: 32..63) of irrep RR.  bb: the lane's view of the per-atom vector block B [D][64], already
// offset by 4 * (lane >> 5) -- in LDS (broadcast reads) or in global memory (gathered by center atom: the lanes of one
// atom read the same 16-B cells).  Groups of four channels are evaluated one after the other with the next group's
// cells in flight; the anchors pin that order (unconstrained, the optimizer gathers all cells at the front and sinks
// the arithmetic below the following MFMA phases -- see aa::anchor).
template <int RR>
__device__ __forceinline__ void tile_scal_accumulate(const float* bb, const float* Y, const v16f& w0a, const v16f& w0b, v16f& s0, v16f& s1) {
  constexpr int a0 = RR * RR, na = 2 * RR + 1;
  v4f b[2][na];
  auto request = [&](int g, v4f* d) {
#pragma unroll
    for (int a = 0; a < na; ++a) d[a] = *reinterpret_cast<const v4f*>(bb + (a0 + a) * 64 + 32 * (g >> 2) + 8 * (g & 3));
  };
  request(0, b[0]);
#pragma unroll
  for (int g = 0; g < 8; ++g) {
    if (g + 1 < 8) request(g + 1, b[(g + 1) & 1]);
    v4f T4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int a = 0; a < na; ++a) {
#pragma unroll
      for (int i = 0; i < 4; ++i) T4[i] += Y[a0 + a] * b[g & 1][a][i];
    }
    const int q = g & 3;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (g < 4)
        s0[4 * q + i] += w0a[4 * q + i] * T4[i];
      else
        s1[4 * q + i] += w0b[4 * q + i] * T4[i];
    }
    if (g < 4)
      anchor(s0);
    else
      anchor(s1);
    __builtin_amdgcn_sched_barrier(0);
  }
}

}  // namespace aa
