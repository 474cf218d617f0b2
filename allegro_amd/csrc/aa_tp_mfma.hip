// Tensor-product track of the standard 2-layer, 64-wide fp32 stack with the first-layer x1 weights RECOMPUTED on the
// matrix cores (gfx950).
//
// The x1 operand of every tensor product of the stack is w0[e] (x) Y[e] with w0[e] = EDGE_EMBEDDING[e] @ Wg
// (allegro/nn/_allegro.py:251-258: env_embed_linear | first_layer_env_embed_projection), a [E, R*64] array that the
// staged pipeline writes once and every tensor-product kernel reads again (768 B/edge each at l_max = 2 -- the largest
// single operand of the step).  Here a wave owns one center atom as before, but walks its edge segment in 32-edge MFMA
// tiles: the 256-B embedding row of an edge is the B operand of a 64 x (R*64) bf16x3 GEMM against Wg, whose fragments
// (R * 24 KB) stay resident in LDS for the whole persistent kernel, so w0 exists only as accumulator tiles and there is
// no barrier after the prologue.
//
//   phase 1  moments of the layer's env input over the segment   M[j][k] = sum_e Y[e][j] a[e][k]   (_contract.py:195-205
//            by linearity, see aa_tp_spec.hip): tile -> wave-private LDS patch -> lane = k walks the rows
//   phase 2  x2s = f * M @ Wenv (lane = channel), per-atom Clebsch-Gordan vector B (Sig^T_x1), B -> LDS
//   phase 3  per tile and irrep r: w0[.][r][.] = emb @ Wg[:, r] (48 MFMAs), scal[e][ch] += w0[e][r][ch] * sum_{a in r}
//            Y[e][a] B[a][ch] in the accumulator layout (lane = edge), full-line stores through a transpose patch
//
// Layouts as in aa_gemm.hip / aa_fused.hip (accumulator layout = next B operand; k order = accumulator order).
#include <type_traits>

#include "aa_cg_gen.h"
#include "aa_wave.h"
#include "aa_common.h"
#include "aa_mfma.h"

namespace aa {

namespace {

constexpr int kLdA = 68;   // row stride (floats) of the [32 edges][64 k] patch
constexpr int kLdT = kTileLdT;  // row stride of the [32][32] store-transpose patch
constexpr int kLdY = 16;   // row stride of sY [32 edges][<=16] and sM [64 k][<=16]
constexpr int kOffB = 32 * kLdT;                     // sB [D][64] sits behind the store patch inside the wave region
constexpr int kWaveFloats = 32 * kLdA + 32 * kLdY;   // wave region: max(sA, sM | sT + sB) then sY
constexpr int kWBlock = 6 * 64;                      // u32x4 per (tile, chunk) block of bf16x3 fragments (6 KB)
constexpr int kWaves = 8;                            // waves per workgroup (two per SIMD)
static_assert(kOffB + 16 * 64 <= 32 * kLdA, "per-atom vectors must fit behind the store patch");

// pointer with a wave-uniform value -> SGPR pair (global loads then use the scalar-base + lane-offset form)
template <class P>
__device__ __forceinline__ const P* uniform_ptr(const P* p) {
  const unsigned long long v = reinterpret_cast<unsigned long long>(p);
  const unsigned lo = __builtin_amdgcn_readfirstlane(unsigned(v)), hi = __builtin_amdgcn_readfirstlane(unsigned(v >> 32));
  return reinterpret_cast<const P*>((static_cast<unsigned long long>(hi) << 32) | lo);
}

// 24 MFMAs of one 32-deep chunk for a tile pair (6 cross products x 2 k halves x 2 tiles); wa / wb: the chunk's
// fragment blocks of the two tiles in LDS ([level x half][lane] 16-B cells)
__device__ __forceinline__ void mma_pair(const u32x4* wa, const u32x4* wb, const XSplit& x, v16f& acc0, v16f& acc1) {
  {
    const u32x4 a0 = wa[4 * 64], b0 = wb[4 * 64], a1 = wa[5 * 64], b1 = wb[5 * 64];  // level 3
    acc0 = mma_bf16(a0, x.l1[0], acc0);
    acc1 = mma_bf16(b0, x.l1[0], acc1);
    acc0 = mma_bf16(a1, x.l1[1], acc0);
    acc1 = mma_bf16(b1, x.l1[1], acc1);
  }
  {
    const u32x4 a0 = wa[2 * 64], b0 = wb[2 * 64], a1 = wa[3 * 64], b1 = wb[3 * 64];  // level 2
    acc0 = mma_bf16(a0, x.l2[0], acc0);
    acc1 = mma_bf16(b0, x.l2[0], acc1);
    acc0 = mma_bf16(a1, x.l2[1], acc0);
    acc1 = mma_bf16(b1, x.l2[1], acc1);
    acc0 = mma_bf16(a0, x.l1[0], acc0);
    acc1 = mma_bf16(b0, x.l1[0], acc1);
    acc0 = mma_bf16(a1, x.l1[1], acc0);
    acc1 = mma_bf16(b1, x.l1[1], acc1);
  }
  {
    const u32x4 a0 = wa[0], b0 = wb[0], a1 = wa[64], b1 = wb[64];  // level 1
    acc0 = mma_bf16(a0, x.l3[0], acc0);
    acc1 = mma_bf16(b0, x.l3[0], acc1);
    acc0 = mma_bf16(a1, x.l3[1], acc0);
    acc1 = mma_bf16(b1, x.l3[1], acc1);
    acc0 = mma_bf16(a0, x.l2[0], acc0);
    acc1 = mma_bf16(b0, x.l2[0], acc1);
    acc0 = mma_bf16(a1, x.l2[1], acc0);
    acc1 = mma_bf16(b1, x.l2[1], acc1);
    acc0 = mma_bf16(a0, x.l1[0], acc0);
    acc1 = mma_bf16(b0, x.l1[0], acc1);
    acc0 = mma_bf16(a1, x.l1[1], acc0);
    acc1 = mma_bf16(b1, x.l1[1], acc1);
  }
}

// rows [row0, row0 + cnt) of a row-major [E, 64] array as two 32-feature tiles in accumulator layout (lane = (edge
// el, half hh); register s holds feature 8 (s >> 2) + 4 hh + (s & 3)); rows beyond cnt repeat the last row
template <bool ACT>
__device__ __forceinline__ void load_tile_pair(const float* src, int64_t row0, int cnt, int lane, v16f& t0, v16f& t1) {
  const int el = lane & 31, hh = lane >> 5;
  const float* p = src + (row0 + (el < cnt ? el : cnt - 1)) * 64 + 4 * hh;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const v4f a = *reinterpret_cast<const v4f*>(p + 8 * q);
    const v4f b = *reinterpret_cast<const v4f*>(p + 32 + 8 * q);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      t0[4 * q + i] = ACT ? silu(a[i]) : a[i];
      t1[4 * q + i] = ACT ? silu(b[i]) : b[i];
    }
  }
}

// M[j] (lane = k) += sum over the tile's rows of Y[e][j] * a[e][k].  Rows beyond the segment carry Y = 0 in sY.
template <int D>
__device__ __forceinline__ void tile_moments_acc(float* sA, const float* sY, const v16f& t0, const v16f& t1, int cnt, int lane, float* M) {
  const int el = lane & 31, hh = lane >> 5;
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    *reinterpret_cast<v4f*>(sA + el * kLdA + 8 * q + 4 * hh) = v4f{t0[4 * q], t0[4 * q + 1], t0[4 * q + 2], t0[4 * q + 3]};
    *reinterpret_cast<v4f*>(sA + el * kLdA + 32 + 8 * q + 4 * hh) = v4f{t1[4 * q], t1[4 * q + 1], t1[4 * q + 2], t1[4 * q + 3]};
  }
  __builtin_amdgcn_wave_barrier();
  const int c4 = (cnt + 3) & ~3;
  for (int e0 = 0; e0 < c4; e0 += 4) {
#pragma unroll
    for (int ee = 0; ee < 4; ++ee) {
      const int e = e0 + ee;
      const float a = sA[e * kLdA + lane];
      float y[16];
#pragma unroll
      for (int q = 0; q < (D + 3) / 4; ++q) {
        const v4f yy = *reinterpret_cast<const v4f*>(sY + e * kLdY + 4 * q);
#pragma unroll
        for (int i = 0; i < 4; ++i) y[4 * q + i] = yy[i];
      }
#pragma unroll
      for (int j = 0; j < D; ++j) M[j] += y[j] * a;
    }
  }
  __builtin_amdgcn_wave_barrier();
}

// x2s[j] (lane = channel) = f * sum_k M[j][k] * Wk[k][r(j)][ch]; M handed over through sM [k][kLdY]; weight rows come
// from L2, 8 k ahead
template <int D, int R>
__device__ __forceinline__ void project_moments(float* sM, const float* M, const float* Wk, float sf, int lane, float* x2s) {
  constexpr int KB = 8;
#pragma unroll
  for (int q = 0; q < (D + 3) / 4; ++q) {
    v4f mm;
#pragma unroll
    for (int i = 0; i < 4; ++i) mm[i] = 4 * q + i < D ? M[4 * q + i] : 0.f;
    *reinterpret_cast<v4f*>(sM + lane * kLdY + 4 * q) = mm;
  }
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int j = 0; j < D; ++j) x2s[j] = 0.f;
  float wc[KB][R], wn[KB][R];
  // (wave-uniform row base + one lane offset: with per-lane 64-bit addresses the compiler hoists one address pair per
  //  (row, irrep) out of the persistent loop and spills them)
  auto loadw = [&](int k0, float(*w)[R]) {
    const float* base = uniform_ptr(Wk + (k0 < 64 ? k0 : 64 - KB) * R * 64);
#pragma unroll
    for (int i = 0; i < KB; ++i) {
#pragma unroll
      for (int r = 0; r < R; ++r) w[i][r] = base[(i * R + r) * 64 + lane];
    }
  };
  loadw(0, wc);
  for (int k0 = 0; k0 < 64; k0 += KB) {
    loadw(k0 + KB, wn);
#pragma unroll
    for (int i = 0; i < KB; ++i) {
      float m[16];
#pragma unroll
      for (int q = 0; q < (D + 3) / 4; ++q) {
        const v4f mm = *reinterpret_cast<const v4f*>(sM + (k0 + i) * kLdY + 4 * q);
#pragma unroll
        for (int t = 0; t < 4; ++t) m[4 * q + t] = mm[t];
      }
#pragma unroll
      for (int j = 0; j < D; ++j) x2s[j] += m[j] * wc[i][r_of<0>(j)];
    }
#pragma unroll
    for (int i = 0; i < KB; ++i)
#pragma unroll
      for (int r = 0; r < R; ++r) wc[i][r] = wn[i][r];
  }
#pragma unroll
  for (int j = 0; j < D; ++j) x2s[j] *= sf;
  __builtin_amdgcn_wave_barrier();
}

// the lane's harmonics (lane = edge el of the tile; zero beyond the segment), also left in sY [32][kLdY] for phase 1
template <int D>
__device__ __forceinline__ void load_harmonics(const float* sh, int ld_sh, int64_t row0, int cnt, int lane, float* Y, float* sY) {
  const int el = lane & 31, hh = lane >> 5;
  const float* p = sh + (row0 + (el < cnt ? el : cnt - 1)) * ld_sh;
#pragma unroll
  for (int j = 0; j < D; ++j) Y[j] = el < cnt ? p[j] : 0.f;
  if (sY && hh == 0) {
#pragma unroll
    for (int q = 0; q < (D + 3) / 4; ++q) {
      v4f yy;
#pragma unroll
      for (int i = 0; i < 4; ++i) yy[i] = 4 * q + i < D ? Y[4 * q + i] : 0.f;
      *reinterpret_cast<v4f*>(sY + el * kLdY + 4 * q) = yy;
    }
  }
}

}  // namespace

template <class Sig0, class Sig1, bool LAST>
__global__ __launch_bounds__(64 * kWaves, 2) void tp_mfma_fwd_kernel(TpMfmaArgs A) {
  constexpr int D = Sig0::D2, R = Sig0::LMAX + 1;
  static_assert(Sig0::D1 == D && Sig0::DOUT == D && Sig1::D1 == D && Sig1::DOUT == 1, "standard 2-layer stack");
  static_assert(D <= 16, "l_max <= 3");
  u32x4* wbuf = reinterpret_cast<u32x4*>(aa_smem);  // [2R tiles][2 chunks][kWBlock]
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, hh = lane >> 5;
  float* sW = reinterpret_cast<float*>(wbuf + 4 * R * kWBlock) + wv * kWaveFloats;
  float* sY = sW + 32 * kLdA;
  float* sBv = sW + kOffB;
  {
    const u32x4* src = static_cast<const u32x4*>(A.wq);
    for (int i = tid; i < 4 * R * kWBlock; i += 64 * kWaves) wbuf[i] = src[i];
  }
  float wp0[Sig0::P], wp1[LAST ? Sig1::P : 1];
#pragma unroll
  for (int q = 0; q < Sig0::P; ++q) wp0[q] = A.coupling ? A.tpw0[lane * Sig0::P + q] : A.tpw0[q];
  if constexpr (LAST) {
#pragma unroll
    for (int q = 0; q < Sig1::P; ++q) wp1[q] = A.coupling ? A.tpw1[lane * Sig1::P + q] : A.tpw1[q];
  }
  __syncthreads();  // weights staged; from here on the waves are independent
  for (int64_t atom = A.atom0 + int64_t(blockIdx.x) * kWaves + wv; atom < A.N; atom += int64_t(gridDim.x) * kWaves) {
    const int beg = __builtin_amdgcn_readfirstlane(A.rowptr[atom]), end = __builtin_amdgcn_readfirstlane(A.rowptr[atom + 1]);
    if (beg >= end) continue;  // (no edges: nothing to contribute or receive, as in the staged kernels)
    const bool single = end - beg <= 32;
    // ---- phase 1: moments of the env input
    float M[D], Y[D];
#pragma unroll
    for (int j = 0; j < D; ++j) M[j] = 0.f;
    for (int t0 = beg; t0 < end; t0 += 32) {
      const int cnt = end - t0 < 32 ? end - t0 : 32;
      load_harmonics<D>(A.sh, A.ld_sh, t0, cnt, lane, Y, sY);
      v16f a0, a1;
      load_tile_pair<LAST>(A.a, t0, cnt, lane, a0, a1);
      tile_moments_acc<D>(sW, sY, a0, a1, cnt, lane, M);
    }
    // ---- phase 2: x2s, per-atom Clebsch-Gordan vector
    {
      float x2s[D], B[D];
      project_moments<D, R>(sW, M, A.wk, A.sf, lane, x2s);
      float* xo = (LAST ? A.x2s1 : A.x2s0) + atom * D * 64 + lane;
#pragma unroll
      for (int j = 0; j < D; ++j) xo[j * 64] = x2s[j];
      if constexpr (LAST) {
        float x2s0[D], one[1] = {1.f}, v[D];
        const float* xi = A.x2s0 + atom * D * 64 + lane;
#pragma unroll
        for (int j = 0; j < D; ++j) x2s0[j] = xi[j * 64];
        Sig1::template bx1<float>(one, x2s, wp1, v);
        Sig0::template bx1<float>(v, x2s0, wp0, B);
      } else {
        float e0[D];
#pragma unroll
        for (int k = 0; k < D; ++k) e0[k] = k == 0 ? 1.f : 0.f;
        Sig0::template bx1<float>(e0, x2s, wp0, B);
      }
#pragma unroll
      for (int a = 0; a < D; ++a) sBv[a * 64 + lane] = B[a];
      __builtin_amdgcn_wave_barrier();
    }
    // ---- phase 3: w0 tiles on the matrix cores, scalars of the layer
    for (int t0 = beg; t0 < end; t0 += 32) {
      const int cnt = end - t0 < 32 ? end - t0 : 32;
      // (the embedding rows come from L2 -- this wave read them microseconds ago; holding the tile across phase 2
      //  instead costs 32 registers and spills)
      v16f e0t, e1t;
      load_tile_pair<false>(A.emb, t0, cnt, lane, e0t, e1t);
      if (!single) load_harmonics<D>(A.sh, A.ld_sh, t0, cnt, lane, Y, nullptr);
      XSplit xs[2];
      xsplit_from_acc(e0t, xs[0]);
      xsplit_from_acc(e1t, xs[1]);
      v16f sc0, sc1;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        sc0[r] = 0.f;
        sc1[r] = 0.f;
      }
      static_for<0, R>([&](auto rr) {
        constexpr int r = decltype(rr)::value;
        v16f acc0, acc1;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          acc0[i] = 0.f;
          acc1[i] = 0.f;
        }
        const u32x4* wa = wbuf + (2 * r) * 2 * kWBlock + lane;      // tile 2r:     chunks 0, 1
        const u32x4* wb = wbuf + (2 * r + 1) * 2 * kWBlock + lane;  // tile 2r + 1
        // (one scheduling region per chunk: otherwise the LDS fragment reads of all irreps are hoisted to the front)
        __builtin_amdgcn_sched_barrier(0);
        mma_pair(wa, wb, xs[0], acc0, acc1);
        __builtin_amdgcn_sched_barrier(0);
        mma_pair(wa + kWBlock, wb + kWBlock, xs[1], acc0, acc1);
        __builtin_amdgcn_sched_barrier(0);
        tile_scal_accumulate<r>(sBv + 4 * hh, Y, acc0, acc1, sc0, sc1);
      });
      __builtin_amdgcn_sched_barrier(0);
      tile_store_rows(sW, sc0, A.scal, t0, cnt, 64, lane);
      tile_store_rows(sW, sc1, A.scal + 32, t0, cnt, 64, lane);
    }
    __builtin_amdgcn_wave_barrier();  // (the next atom's patches alias this one's)
  }
}

size_t tp_mfma_lds_bytes(int R) { return sizeof(u32x4) * 4 * R * kWBlock + sizeof(float) * kWaves * kWaveFloats; }

int launch_tp_mfma_fwd(int pair, bool last, const TpMfmaArgs& a, hipStream_t stream) {
  if (a.N <= a.atom0) return AA_OK;
  const int R = pair == 0 ? 2 : 3;
  const size_t smem = tp_mfma_lds_bytes(R);
  if (pair < 0 || pair > 1 || smem > 160 * 1024) return fail(AA_ERR_INVALID, "tp_mfma_fwd: l_max <= 2 only");
  static int num_cu = 0;
  if (num_cu == 0) {
    int dev = 0, n = 0;
    AA_CHECK_HIP(hipGetDevice(&dev));
    AA_CHECK_HIP(hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev));
    num_cu = n > 0 ? n : 256;
  }
  const int64_t ngroups = (a.N - a.atom0 + kWaves - 1) / kWaves;
  dim3 grid((unsigned)std::min<int64_t>(ngroups, num_cu));
#define AA_TPM_LAUNCH(S0_, S1_, L_)                                                                   \
  {                                                                                                   \
    const void* fn = (const void*)tp_mfma_fwd_kernel<cg::S0_, cg::S1_, L_>;                           \
    AA_CHECK_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, int(smem)));     \
    hipLaunchKernelGGL((tp_mfma_fwd_kernel<cg::S0_, cg::S1_, L_>), grid, dim3(64 * kWaves), smem, stream, a); \
  }
  if (pair == 0) {
    if (last) AA_TPM_LAUNCH(Sig1, Sig0, true) else AA_TPM_LAUNCH(Sig1, Sig0, false)
  } else {
    if (last) AA_TPM_LAUNCH(Sig5, Sig4, true) else AA_TPM_LAUNCH(Sig5, Sig4, false)
  }
#undef AA_TPM_LAUNCH
  AA_CHECK_HIP(hipGetLastError());
  return AA_OK;
}

}  // namespace aa
