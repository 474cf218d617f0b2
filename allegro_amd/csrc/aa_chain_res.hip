// One-layer reverse chains with their weights RESIDENT in LDS (round 6).
//
// gemm_chain_bf16x3_kernel (aa_gemm.hip) gives every 128-row workgroup its own copy of the layer: the 12-KB weight steps are
// fetched from L2 and staged through LDS per workgroup, one barrier per step -- for a ONE-layer chain that is 48-96 KB of weight
// traffic, 12-24 staging instructions per thread and 4-8 barriers per 128 rows of 0.2 MB.  The latent-0 reverse of the 2-layer
// stack (Runner::backward "B2": K = 64 -> N = 128 behind the operand transform A = (a + add) silu'(z)) is 48 KB of bf16x3
// fragments: here ONE persistent workgroup per CU loads them once, and its eight waves then stream 32-row tiles independently --
// no staging, no barrier inside the loop, the next tile's operand rows in flight while the current one is transformed, split and
// multiplied.  Same arithmetic in the same order as the chain kernel (bit-equal results).
// Semantics: the reverse of `ScalarMLPFunction` of latent 0, allegro/nn/_allegro.py:192-213, 272-283.
#include "aa_fused_tile.h"

namespace aa {
namespace {

struct RawRows {
  v4f a[8], add[8], z[8];  // two 32-feature chunks x four 16-byte pieces of the lane's row, accumulator-order k
};

}  // namespace

__global__ __launch_bounds__(512, 2) void chain_b2_resident_kernel(ChainB2Args g) {
  u32x4* wres = reinterpret_cast<u32x4*>(aa_smem);  // [4 steps][kWStep]: step 2 p + kc = tile pair p, k chunk kc
  const int tid = threadIdx.x, lane = tid & 63, hh = lane >> 5, el = lane & 31;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  float* sT = reinterpret_cast<float*>(wres + 4 * kWStep) + wv * 32 * kTileLdT;
  {
    const u32x4* Wq = static_cast<const u32x4*>(g.Wq);
    for (int i = tid; i < 4 * kWStep; i += 512) {
      const int s = i / kWStep, j = i % kWStep, p = s >> 1, kc = s & 1, t = 2 * p + (j >= 384 ? 1 : 0);
      wres[i] = Wq[size_t(t * 2 + kc) * 384 + (j % 384)];
    }
  }
  __syncthreads();
  const int64_t ntiles = (g.M + 31) / 32, stride = int64_t(gridDim.x) * 8;
  int64_t tile = int64_t(blockIdx.x) * 8 + wv;
  auto load = [&](int64_t t, RawRows& r) {
    int64_t row = t * 32 + el;
    row = row < g.M ? row : g.M - 1;
    const float* pa = g.a + row * g.lda + 4 * hh;
    const float* pd = g.add + row * g.ldadd + 4 * hh;
    const float* pz = g.z + row * g.ldz + 4 * hh;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      r.a[q] = *reinterpret_cast<const v4f*>(pa + 8 * q);
      r.add[q] = *reinterpret_cast<const v4f*>(pd + 8 * q);
      r.z[q] = *reinterpret_cast<const v4f*>(pz + 8 * q);
    }
  };
  RawRows nxt;
  if (tile < ntiles) load(tile, nxt);
  for (; tile < ntiles; tile += stride) {
    const int64_t row0 = tile * 32;
    const int cnt = int(g.M - row0 < 32 ? g.M - row0 : 32);
    // operand transform of this tile
    XSplit xs[2];
    {
      v16f A0, A1;
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          A0[4 * q + e] = (nxt.a[q][e] + nxt.add[q][e]) * dsilu(nxt.z[q][e]);
          A1[4 * q + e] = (nxt.a[4 + q][e] + nxt.add[4 + q][e]) * dsilu(nxt.z[4 + q][e]);
        }
      xsplit_from_acc(A0, xs[0]);
      xsplit_from_acc(A1, xs[1]);
    }
    // the destination rows that are accumulated into: requested now, needed behind the 48 MFMAs of the first tile pair
    v4f old[8];
    {
      int64_t row = row0 + el;
      row = row < g.M ? row : g.M - 1;
      const float* pc = g.c0 + row * g.ldc0 + 4 * hh;
#pragma unroll
      for (int q = 0; q < 8; ++q) old[q] = *reinterpret_cast<const v4f*>(pc + 8 * q);
    }
    v16f acc0, acc1;
    // tile pair 0: accumulated into c0
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      acc0[r] = 0.f;
      acc1[r] = 0.f;
    }
    fused_mma_step(wres + 0 * kWStep, lane, xs[0], acc0, acc1);
    fused_mma_step(wres + 1 * kWStep, lane, xs[1], acc0, acc1);
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        acc0[4 * q + e] += old[q][e];
        acc1[4 * q + e] += old[4 + q][e];
      }
    tile_store_rows(sT, acc0, g.c0, row0, cnt, g.ldc0, lane);
    tile_store_rows(sT, acc1, g.c0 + 32, row0, cnt, g.ldc0, lane);
    // the next tile's operand rows travel behind the second tile pair (and the other wave of the SIMD)
    if (tile + stride < ntiles) load(tile + stride, nxt);
    // tile pair 1: stored to c1
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      acc0[r] = 0.f;
      acc1[r] = 0.f;
    }
    fused_mma_step(wres + 2 * kWStep, lane, xs[0], acc0, acc1);
    fused_mma_step(wres + 3 * kWStep, lane, xs[1], acc0, acc1);
    tile_store_rows(sT, acc0, g.c1, row0, cnt, g.ldc1, lane);
    tile_store_rows(sT, acc1, g.c1 + 32, row0, cnt, g.ldc1, lane);
  }
}

int launch_chain_b2_resident(const ChainB2Args& g, hipStream_t stream) {
  if (g.M <= 0) return AA_OK;
  if (!(g.a && g.add && g.z && g.Wq && g.c0 && g.c1)) return fail(AA_ERR_INVALID, "resident chain: null argument");
  if ((g.lda | g.ldadd | g.ldz | g.ldc0 | g.ldc1) & 3) return fail(AA_ERR_INVALID, "resident chain: row strides must be multiples of 4 floats");
  const size_t smem = sizeof(u32x4) * 4 * kWStep + sizeof(float) * 8 * 32 * kTileLdT;
  const int64_t groups = ((g.M + 31) / 32 + 7) / 8;
  dim3 grid((unsigned)std::min<int64_t>(groups, fused_num_cus()));
  AA_CHECK_HIP(hipFuncSetAttribute((const void*)chain_b2_resident_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, int(smem)));
  hipLaunchKernelGGL(chain_b2_resident_kernel, grid, dim3(512), smem, stream, g);
  AA_CHECK_HIP(hipGetLastError());
  return AA_OK;
}

}  // namespace aa
