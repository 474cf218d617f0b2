// Fused per-atom-tile kernels of the standard 2-layer Allegro stack (gfx950, fp32 via bf16x3 MFMA).
//
// Strict locality (the reference's own model test, tests/model/test_allegro.py:68-70) makes every dependency of the
// forward pass local to one center atom: its edge segment and per-atom sums over that segment.  One WAVE therefore
// owns one center atom's edge tile (<= 32 edges = one 32-column MFMA tile) and runs the WHOLE module chain of
// allegro/model/allegro_models.py:222-297 on it with every activation in registers / LDS:
//
//   geometry, spherical harmonics, Bessel x cutoff x type-pair embedding   (tensorembed.py:86-92, scalarembed.py:60-81)
//   scalar_embed_mlp  ->  env_embed_linear | first_layer projection         (allegro_models.py:173-183, _allegro.py:251)
//   layer 0: moments -> x2s0 -> per-atom vector B0 -> scal0 -> latent 0      (_channels.py:44-57, _contract.py:185-251)
//   layer 1: moments -> x2s1 -> per-atom vector B1 -> scal1 -> latent 1      (_allegro.py:263-294)
//   edge readout + EdgewiseReduce + PerTypeScaleShift                        (allegro_models.py:231-260, edgewise.py:40-60)
//
// What goes to HBM is only what the reverse pass reads again (pre-activations of the four hidden layers, the edge
// embedding, x2s per atom) -- the 9 stage boundaries of the staged pipeline (aa_model.hip) disappear.
//
// Layouts.  Per-edge activations live in the MFMA accumulator layout of the swapped-operand GEMM (aa_gemm.hip):
// lane = (edge el = lane & 31, half hh = lane >> 5); a 32-feature tile is 16 registers, register s holds feature
// 8 (s >> 2) + 4 hh + (s & 3).  That IS the B operand of the next layer (weights pre-permuted on the host), so linear
// layers chain with no data movement.  Per-atom quantities (moments M[j][k], x2s[j][ch], the Clebsch-Gordan vectors
// B[a][ch]) live in the transposed view lane = k / channel; the two views exchange through a wave-private LDS patch:
//   moments   M[j][k] = sum_e Y[e][j] a[e][k]      tile -> LDS [e][k] -> lane k walks the 32 rows
//   scalars   scal[e][ch] = sum_r w0[e][r][ch] * (sum_{a in r} Y[e][a] B[a][ch])      B from LDS (broadcast reads),
//             evaluated in the epilogue of the GEMM tile pair that produces w0[.][r][.] -- w0 is never stored for
//             the forward's own use and is recomputed (6 MFMA steps) for the second layer instead of being held in
//             96 registers or re-read from HBM.
// Weights stream L2 -> LDS once per workgroup (4 waves = 4 atoms share every 12-KB step, double buffered), exactly
// the staging of gemm_chain_bf16x3_kernel.
//
// Limits of this first version: every center atom of the block has <= 32 edges (aa_graph.max_degree; larger
// segments run the staged pipeline), u = S = all MLP widths = 64, embedding table path (<= 2 species, 8 basis
// functions), fp32.
#include <type_traits>

#include "aa_cg_gen.h"
#include "aa_wave.h"
#include "aa_common.h"
#include "aa_geom.h"
#include "aa_mfma.h"

namespace aa {

namespace {

constexpr int kLdA = 68;   // row stride (floats) of the [32 edges][64 features] patch: 16-B aligned rows, conflict-light
constexpr int kLdT = 36;   // row stride of the [32][32] store-transpose patch (as in the chain kernel)
constexpr int kLdY = 16;   // row stride of sY [32 edges][D <= 16] and sM [64 k][D]
constexpr int kOffB = 32 * kLdT;          // sB [D][64] sits behind the store patch inside the wave region
constexpr int kWaveRegion = 32 * kLdA;    // floats: max(sA, sT + sB, sM + sB)
static_assert(kOffB + 16 * 64 <= kWaveRegion, "per-atom vectors must fit behind the store patch");

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

struct FusedCtx {
  int tid, lane, hh, el;
  u32x4* wbuf;  // [2][kWStep]
  int step;
};

// block-cooperative staging of one weight step (tile pair nt, nt+1; chunk kc) -- see gemm_chain_bf16x3_kernel
__device__ __forceinline__ void fused_stage_load(const FusedLayerDev& L, int nt, int kc, int tid, u32x4* r) {
  const size_t chunk_stride = 64 * 6, tile_stride = size_t(L.KC) * chunk_stride;
  const u32x4* W = static_cast<const u32x4*>(L.Wq);
  const u32x4* s0 = W + size_t(nt) * tile_stride + size_t(kc) * chunk_stride;
  const u32x4* s1 = W + size_t(nt + 1) * tile_stride + size_t(kc) * chunk_stride;
  r[0] = s0[tid];
  r[1] = tid < 128 ? s0[256 + tid] : s1[tid - 128];
  r[2] = s1[128 + tid];
}
__device__ __forceinline__ void fused_stage_write(u32x4* wbuf, int b, int tid, const u32x4* r) {
  u32x4* d = wbuf + b * kWStep;
  d[tid] = r[0];
  d[256 + tid] = r[1];
  d[512 + tid] = r[2];
}

// 24 MFMAs of one step (6 cross products x 2 k halves x 2 tiles); weight levels read from LDS just in time
__device__ __forceinline__ void fused_mma_step(const u32x4* wb, int lane, const XSplit& x, v16f& acc0, v16f& acc1) {
  const u32x4* w = wb + lane;
#define AA_W(T_, Q_) w[((T_)*6 + (Q_)) * 64]
  {
    const u32x4 a0 = AA_W(0, 4), b0 = AA_W(1, 4), a1 = AA_W(0, 5), b1 = AA_W(1, 5);  // level 3
    acc0 = mma_bf16(a0, x.l1[0], acc0);
    acc1 = mma_bf16(b0, x.l1[0], acc1);
    acc0 = mma_bf16(a1, x.l1[1], acc0);
    acc1 = mma_bf16(b1, x.l1[1], acc1);
  }
  {
    const u32x4 a0 = AA_W(0, 2), b0 = AA_W(1, 2), a1 = AA_W(0, 3), b1 = AA_W(1, 3);  // level 2
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
    const u32x4 a0 = AA_W(0, 0), b0 = AA_W(1, 0), a1 = AA_W(0, 1), b1 = AA_W(1, 1);  // level 1
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
#undef AA_W
}

// One linear layer on the wave's tile: KC 32-deep operand chunks (op(kc) -> the v16f tile that is chunk kc), NT output
// tiles in pairs (epi(pair, acc0, acc1) after each pair).  Every step stages the NEXT step's weights (next chunk /
// next pair / first step of layer Ln) while its own MFMAs issue; one block barrier per step.
template <int KC, int NT, class OpF, class EpiF>
__device__ __forceinline__ void fused_layer(FusedCtx& c, const FusedLayerDev& L, const FusedLayerDev& Ln, bool kernel_last,
                                            OpF&& op, EpiF&& epi) {
  static_assert(NT % 2 == 0, "output tiles come in pairs");
  constexpr bool PRE = KC <= 2 && NT > 2;  // few chunks, many pairs: split the operand once
  XSplit ps[PRE ? KC : 1];
  if constexpr (PRE) {
    static_for<0, KC>([&](auto kc) { const v16f t = op(kc); xsplit_from_acc(t, ps[kc]); });
  }
  static_for<0, NT / 2>([&](auto ntp) {
    constexpr int nt = 2 * decltype(ntp)::value;
    constexpr bool last_pair = nt + 2 >= NT;
    v16f acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      acc0[r] = 0.f;
      acc1[r] = 0.f;
    }
    static_for<0, KC>([&](auto kcc) {
      constexpr int kc = decltype(kcc)::value;
      constexpr bool lastc = kc + 1 >= KC;
      u32x4 r[3];
      if constexpr (!lastc) {
        fused_stage_load(L, nt, kc + 1, c.tid, r);
      } else if constexpr (!last_pair) {
        fused_stage_load(L, nt + 2, 0, c.tid, r);
      } else {
        if (kernel_last)
          fused_stage_load(L, nt, kc, c.tid, r);  // (nothing follows: re-stage own step, uniform instruction stream)
        else
          fused_stage_load(Ln, 0, 0, c.tid, r);
      }
      if constexpr (PRE) {
        fused_mma_step(c.wbuf + (c.step & 1) * kWStep, c.lane, ps[kc], acc0, acc1);
      } else {
        XSplit x;
        {
          const v16f t = op(kcc);
          xsplit_from_acc(t, x);
        }
        fused_mma_step(c.wbuf + (c.step & 1) * kWStep, c.lane, x, acc0, acc1);
      }
      fused_stage_write(c.wbuf, (c.step + 1) & 1, c.tid, r);
      __syncthreads();
      __builtin_amdgcn_sched_barrier(0);  // keep later steps' operand splits / loads from being hoisted over this one
      ++c.step;
    });
    epi(ntp, acc0, acc1);
  });
}

// store one 32-feature tile (accumulator layout) to rows [row0, row0 + cnt) of a row-major [E, ld] array through the
// wave-private transpose patch, so that every store instruction writes whole 128-B lines
__device__ __forceinline__ void fused_store_tile(float* sT, const v16f& acc, float* dst, int64_t row0, int cnt, int ld, int lane) {
  const int el = lane & 31, hh = lane >> 5;
  float* st = sT + el * kLdT + 4 * hh;
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int q = 0; q < 4; ++q) *reinterpret_cast<v4f*>(st + 8 * q) = v4f{acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]};
  __builtin_amdgcn_wave_barrier();
  const int pr = lane >> 3, pc = 4 * (lane & 7);
  v4f v[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) v[q] = *reinterpret_cast<const v4f*>(sT + (8 * q + pr) * kLdT + pc);
  __builtin_amdgcn_wave_barrier();
  float* p = dst + (row0 + pr) * ld + pc;
#pragma unroll
  for (int q = 0; q < 4; ++q)
    if (pr + 8 * q < cnt) *reinterpret_cast<v4f*>(p + int64_t(8 * q) * ld) = v[q];
}

// A tile pair parked in LDS in accumulator layout ([q][lane] 16-B cells: conflict-free b128 accesses).  The two-body
// scalars and lat0 are operands of three / two later layers; parking them frees 64 registers per lane for the whole
// second half of the kernel (the kernel runs one wave per SIMD, LDS is plentiful).
__device__ __forceinline__ void park_tile(float* slot, const v16f& t, int lane) {
#pragma unroll
  for (int q = 0; q < 4; ++q) *reinterpret_cast<v4f*>(slot + (q * 64 + lane) * 4) = v4f{t[4 * q], t[4 * q + 1], t[4 * q + 2], t[4 * q + 3]};
}
__device__ __forceinline__ v16f fetch_tile(const float* slot, int lane) {
  v16f t;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const v4f v = *reinterpret_cast<const v4f*>(slot + (q * 64 + lane) * 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) t[4 * q + i] = v[i];
  }
  return t;
}
constexpr int kTileFloats = 64 * 16;  // one parked 32-feature tile

template <bool ACT>
__device__ __forceinline__ void keep_tile(const v16f& acc, v16f& k) {
#pragma unroll
  for (int r = 0; r < 16; ++r) k[r] = ACT ? silu(acc[r]) : acc[r];
}

// M[j] (lane = k) = sum over the tile's rows of Y[e][j] * a[e][k]: the two tiles go to the LDS patch in [e][k] order,
// every lane then walks its column.  Rows beyond the segment carry Y = 0 (sY), so they drop out.
template <int D>
__device__ __forceinline__ void tile_moments(float* sA, const float* sY, const v16f& t0, const v16f& t1, int lane, float* M) {
  const int el = lane & 31, hh = lane >> 5;
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    *reinterpret_cast<v4f*>(sA + el * kLdA + 8 * q + 4 * hh) = v4f{t0[4 * q], t0[4 * q + 1], t0[4 * q + 2], t0[4 * q + 3]};
    *reinterpret_cast<v4f*>(sA + el * kLdA + 32 + 8 * q + 4 * hh) = v4f{t1[4 * q], t1[4 * q + 1], t1[4 * q + 2], t1[4 * q + 3]};
  }
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int j = 0; j < D; ++j) M[j] = 0.f;
#pragma unroll 4
  for (int e = 0; e < 32; ++e) {
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
  __builtin_amdgcn_wave_barrier();
}

// x2s[j] (lane = channel) = f * sum_k M[j][k] * Wk[k][r(j)][ch]  with M handed over through sM [k][D]
template <int D, int R>
__device__ __forceinline__ void project_moments(float* sM, const float* M, const float* __restrict__ Wk, float sf, int lane, float* x2s) {
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
  auto loadw = [&](int k0, float(*w)[R]) {
#pragma unroll
    for (int i = 0; i < KB; ++i) {
      const int k = k0 + i < 64 ? k0 + i : 63;
#pragma unroll
      for (int r = 0; r < R; ++r) w[i][r] = Wk[(int64_t(k) * R + r) * 64 + lane];
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

// scal[e][ch] += w[e][r][ch] * sum_{a in irrep r} Y[e][a] * B[a][ch]  for the tile pair (w0a: channels 0..31,
// w0b: 32..63) of irrep r; B[a][ch] from LDS (the lanes of a half read the same address: broadcast)
template <int RR>
__device__ __forceinline__ void scal_accumulate(const float* sB, const float* Y, const v16f& w0a, const v16f& w0b, int hh, v16f& s0, v16f& s1) {
  constexpr int a0 = RR * RR, na = 2 * RR + 1;
#pragma unroll
  for (int t = 0; t < 2; ++t) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      v4f T4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int a = 0; a < na; ++a) {
        const v4f b4 = *reinterpret_cast<const v4f*>(sB + (a0 + a) * 64 + 32 * t + 8 * q + 4 * hh);
#pragma unroll
        for (int i = 0; i < 4; ++i) T4[i] += Y[a0 + a] * b4[i];
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (t == 0)
          s0[4 * q + i] += w0a[4 * q + i] * T4[i];
        else
          s1[4 * q + i] += w0b[4 * q + i] * T4[i];
      }
    }
  }
}

}  // namespace

// ------------------------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------------------------
template <class Sig0, class Sig1>
__global__ __launch_bounds__(256, 1) void fused_fwd_kernel(FusedFwdArgs A) {
  constexpr int D = Sig0::D2, R = Sig0::LMAX + 1;
  static_assert(Sig0::D1 == D && Sig0::DOUT == D && Sig1::D1 == D && Sig1::DOUT == 1, "standard 2-layer stack");
  static_assert(D <= 16, "l_max <= 3");
  u32x4* wbuf = reinterpret_cast<u32x4*>(aa_smem);
  float* sRo = reinterpret_cast<float*>(wbuf + 2 * kWStep);            // [64] last readout weights
  float* sTab = sRo + 64;                                              // [T*T][8][64] two-body table
  const int ntab = A.num_types * A.num_types * 512;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, hh = lane >> 5, el = lane & 31;
  float* sW = sTab + ntab + wv * (kWaveRegion + 32 * kLdY);            // wave region: patch / per-atom vectors ...
  float* sY = sW + kWaveRegion;                                        // ... and the harmonics of the tile [32][kLdY]
  float* sBv = sW + kOffB;
  // parked tiles of this wave: two-body scalars (2 tiles) and lat0 (2 tiles)
  float* sPark = sTab + ntab + 4 * (kWaveRegion + 32 * kLdY) + wv * 4 * kTileFloats;
  for (int i = tid; i < 64; i += 256) sRo[i] = A.ro_w[i];
  for (int i = tid; i < ntab; i += 256) sTab[i] = A.emb_tab[i];
  FusedCtx c{tid, lane, hh, el, wbuf, 0};
  {
    u32x4 r[3];
    fused_stage_load(A.L[0], 0, 0, tid, r);
    fused_stage_write(wbuf, 0, tid, r);
  }
  // ---- the wave's atom and edge tile
  const int64_t atom = A.atom0 + int64_t(blockIdx.x) * 4 + wv;
  const bool atom_ok = atom < A.atom_end;
  int beg = 0, cnt = 0;
  if (atom_ok) {
    beg = A.rowptr[atom];
    cnt = A.rowptr[atom + 1] - beg;
  }
  beg = __builtin_amdgcn_readfirstlane(beg);
  cnt = __builtin_amdgcn_readfirstlane(cnt);
  const bool row_ok = el < cnt;
  // ---- geometry of the lane's edge (rows beyond the segment: a harmless dummy that is masked everywhere)
  float Y[D], basis[8];
  int pair = 0;
  {
    float vx = 1.f, vy = 0.f, vz = 0.f;
    float x = 0.5f;
    if (row_ok) {
      const int64_t e = int64_t(beg) + el;
      const int j = A.nbr[e];
      const float* pi = A.pos + 3 * atom;
      const float* pj = A.pos + 3 * int64_t(j);
      vx = pj[0] - pi[0];
      vy = pj[1] - pi[1];
      vz = pj[2] - pi[2];
      if (A.shift_vec) {
        const float* sv = A.shift_vec + 3 * e;
        vx += sv[0];
        vy += sv[1];
        vz += sv[2];
      }
      pair = A.types[atom] * A.num_types + A.types[j];
    }
    const float rr = aa_sqrt(vx * vx + vy * vy + vz * vz);
    const float inv = 1.f / rr;
    const float nx = vx * inv, ny = vy * inv, nz = vz * inv;
    if (row_ok) x = rr * A.rmax_recip[pair];
    float Yf[16];
    sh_eval<float>(Sig0::LMAX, nx, ny, nz, Yf);
#pragma unroll
    for (int m = 0; m < D; ++m) Y[m] = row_ok ? Yf[m] : 0.f;
    if (row_ok && hh == 0) {
      const int64_t e = int64_t(beg) + el;
      *reinterpret_cast<v4f*>(A.vec + 4 * e) = v4f{nx, ny, nz, rr};
      if (A.sh) {
#pragma unroll
        for (int m = 0; m < D; ++m) A.sh[e * D + m] = Yf[m];
      }
    }
    if (hh == 0) {
#pragma unroll
      for (int m = 0; m < kLdY; ++m) sY[el * kLdY + m] = m < D ? Y[m] : 0.f;
    }
    if (A.embed_kind == 1) {
#pragma unroll
      for (int n = 0; n < 8; ++n) {
        float dbv;
        spline_basis_and_grad<float>(x, n, 8, A.spline_span, basis[n], dbv);
      }
    } else {
      float f, df;
      cutoff_and_grad<float>(x, A.poly_p, f, df);
      const float fx = f / x;
#pragma unroll
      for (int n = 0; n < 8; ++n) basis[n] = aa_sin(A.bessel_w[n] * x) * fx;
    }
  }
  __syncthreads();  // tables + first weight step staged
  // ---- two-body embedding of the lane's 32 features: emb0[c] = sum_n basis[n] * tab[pair][n][c]
  v16f em0, em1;
  {
    const float* tb = sTab + pair * 512 + 4 * hh;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      em0[r] = 0.f;
      em1[r] = 0.f;
    }
#pragma unroll
    for (int n = 0; n < 8; ++n) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const v4f t0 = *reinterpret_cast<const v4f*>(tb + n * 64 + 8 * q);
        const v4f t1 = *reinterpret_cast<const v4f*>(tb + n * 64 + 32 + 8 * q);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          em0[4 * q + i] += basis[n] * t0[i];
          em1[4 * q + i] += basis[n] * t1[i];
        }
      }
    }
  }
  const int64_t row0 = beg;
  v16f k0, k1, sc0, sc1;
  // ---- L0: scalar_embed_mlp layer 0 (pre-activation kept for the reverse pass)
  fused_layer<2, 2>(c, A.L[0], A.L[1], false,
                    [&](auto kc) -> const v16f& { if constexpr (decltype(kc)::value == 0) return em0; else return em1; },
                    [&](auto, const v16f& a0, const v16f& a1) {
                      fused_store_tile(sW, a0, A.se_h, row0, cnt, 64, lane);
                      fused_store_tile(sW, a1, A.se_h + 32, row0, cnt, 64, lane);
                      keep_tile<true>(a0, k0);
                      keep_tile<true>(a1, k1);
                    });
  // ---- L1: scalar_embed_mlp layer 1 -> EDGE_EMBEDDING
  fused_layer<2, 2>(c, A.L[1], A.L[2], false,
                    [&](auto kc) -> const v16f& { if constexpr (decltype(kc)::value == 0) return k0; else return k1; },
                    [&](auto, const v16f& a0, const v16f& a1) {
                      fused_store_tile(sW, a0, A.emb, row0, cnt, 64, lane);
                      fused_store_tile(sW, a1, A.emb + 32, row0, cnt, 64, lane);
                      em0 = a0;
                      em1 = a1;
                    });
  // ---- per-atom part of layer 0: moments of the embedding -> x2s0 -> B0 = Sig0^T_x1(e_0, x2s0)
  float wp0[Sig0::P], wp1[Sig1::P];
#pragma unroll
  for (int p = 0; p < Sig0::P; ++p) wp0[p] = A.coupling ? A.tpw0[lane * Sig0::P + p] : A.tpw0[p];
#pragma unroll
  for (int p = 0; p < Sig1::P; ++p) wp1[p] = A.coupling ? A.tpw1[lane * Sig1::P + p] : A.tpw1[p];
  float x2s0[D];
  {
    float M[D];
    tile_moments<D>(sW, sY, em0, em1, lane, M);
    project_moments<D, R>(sW, M, A.wk0, A.sf, lane, x2s0);
    if (atom_ok) {
#pragma unroll
      for (int j = 0; j < D; ++j) A.x2s0[(atom * D + j) * 64 + lane] = x2s0[j];
    }
    float e0[D], B0[D];
#pragma unroll
    for (int k = 0; k < D; ++k) e0[k] = k == 0 ? 1.f : 0.f;
    Sig0::template bx1<float>(e0, x2s0, wp0, B0);
#pragma unroll
    for (int a = 0; a < D; ++a) sBv[a * 64 + lane] = B0[a];
    __builtin_amdgcn_wave_barrier();
  }
  // ---- L2: [two-body scalars | w0 irrep 0 | irrep 1 | ...] = emb @ [first_proj[:, :S] | env_embed_linear]; the
  //          layer-0 tensor-track scalars are accumulated as each irrep's tile pair comes out
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    sc0[r] = 0.f;
    sc1[r] = 0.f;
  }
  fused_layer<2, 2 + 2 * R>(c, A.L[2], A.L[3], false,
                            [&](auto kc) -> const v16f& { if constexpr (decltype(kc)::value == 0) return em0; else return em1; },
                            [&](auto ntp, const v16f& a0, const v16f& a1) {
                              constexpr int p = decltype(ntp)::value;
                              if constexpr (p == 0) {
                                park_tile(sPark, a0, lane);
                                park_tile(sPark + kTileFloats, a1, lane);
                                if (A.fcat) {
                                  fused_store_tile(sW, a0, A.fcat, row0, cnt, 192, lane);
                                  fused_store_tile(sW, a1, A.fcat + 32, row0, cnt, 192, lane);
                                }
                              } else {
                                if (A.w0) {
                                  fused_store_tile(sW, a0, A.w0 + (p - 1) * 64, row0, cnt, 64 * R, lane);
                                  fused_store_tile(sW, a1, A.w0 + (p - 1) * 64 + 32, row0, cnt, 64 * R, lane);
                                }
                                scal_accumulate<p - 1>(sBv, Y, a0, a1, hh, sc0, sc1);
                              }
                            });
  // ---- L3: latent 0, hidden layer: [two-body | scal0] -> h (pre-activation stored), a1 = silu(h)
  fused_layer<4, 2>(c, A.L[3], A.L[4], false,
                    [&](auto kc) -> v16f {
                      constexpr int k = decltype(kc)::value;
                      if constexpr (k < 2) return fetch_tile(sPark + k * kTileFloats, lane); else if constexpr (k == 2) return sc0; else return sc1;
                    },
                    [&](auto, const v16f& a0, const v16f& a1) {
                      fused_store_tile(sW, a0, A.lat_h0, row0, cnt, 64, lane);
                      fused_store_tile(sW, a1, A.lat_h0 + 32, row0, cnt, 64, lane);
                      keep_tile<true>(a0, k0);
                      keep_tile<true>(a1, k1);
                    });
  // ---- per-atom part of layer 1: moments of a1 -> x2s1 -> v = dSig1/dtf1 (x2s1) -> B1 = Sig0^T_x1(v, x2s0)
  {
    float M[D], x2s1[D];
    tile_moments<D>(sW, sY, k0, k1, lane, M);
    project_moments<D, R>(sW, M, A.wk1, A.sf, lane, x2s1);
    if (atom_ok) {
#pragma unroll
      for (int j = 0; j < D; ++j) A.x2s1[(atom * D + j) * 64 + lane] = x2s1[j];
    }
    float one[1] = {1.f}, v[D], B1[D];
    Sig1::template bx1<float>(one, x2s1, wp1, v);
    Sig0::template bx1<float>(v, x2s0, wp0, B1);
#pragma unroll
    for (int a = 0; a < D; ++a) sBv[a * 64 + lane] = B1[a];
    __builtin_amdgcn_wave_barrier();
  }
  // ---- L4: latent 0, output layer -> lat0
  fused_layer<2, 2>(c, A.L[4], A.L[5], false,
                    [&](auto kc) -> const v16f& { if constexpr (decltype(kc)::value == 0) return k0; else return k1; },
                    [&](auto, const v16f& a0, const v16f& a1) {
                      park_tile(sPark + 2 * kTileFloats, a0, lane);
                      park_tile(sPark + 3 * kTileFloats, a1, lane);
                      if (A.fcat) {
                        fused_store_tile(sW, a0, A.fcat + 64, row0, cnt, 192, lane);
                        fused_store_tile(sW, a1, A.fcat + 96, row0, cnt, 192, lane);
                      }
                    });
  // ---- L5: w0 again (recomputed from the embedding still held in registers) -> layer-1 scalars with B1
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    sc0[r] = 0.f;
    sc1[r] = 0.f;
  }
  fused_layer<2, 2 * R>(c, A.L[5], A.L[6], false,
                        [&](auto kc) -> const v16f& { if constexpr (decltype(kc)::value == 0) return em0; else return em1; },
                        [&](auto ntp, const v16f& a0, const v16f& a1) {
                          scal_accumulate<decltype(ntp)::value>(sBv, Y, a0, a1, hh, sc0, sc1);
                        });
  // ---- L6: latent 1, hidden layer: [two-body | lat0 | scal1]
  fused_layer<6, 2>(c, A.L[6], A.L[7], false,
                    [&](auto kc) -> v16f {
                      constexpr int k = decltype(kc)::value;
                      if constexpr (k < 4) return fetch_tile(sPark + k * kTileFloats, lane); else if constexpr (k == 4) return sc0; else return sc1;
                    },
                    [&](auto, const v16f& a0, const v16f& a1) {
                      fused_store_tile(sW, a0, A.lat_h1, row0, cnt, 64, lane);
                      fused_store_tile(sW, a1, A.lat_h1 + 32, row0, cnt, 64, lane);
                      keep_tile<true>(a0, k0);
                      keep_tile<true>(a1, k1);
                    });
  // ---- L7: latent 1, output layer -> lat1
  fused_layer<2, 2>(c, A.L[7], A.L[8], false,
                    [&](auto kc) -> const v16f& { if constexpr (decltype(kc)::value == 0) return k0; else return k1; },
                    [&](auto, const v16f& a0, const v16f& a1) {
                      k0 = a0;
                      k1 = a1;
                      if (A.fcat) {
                        fused_store_tile(sW, a0, A.fcat + 128, row0, cnt, 192, lane);
                        fused_store_tile(sW, a1, A.fcat + 160, row0, cnt, 192, lane);
                      }
                    });
  // ---- L8: edge readout hidden layer on [two-body | lat0 | lat1]; last linear layer + edge sum in the epilogue
  fused_layer<6, 2>(c, A.L[8], A.L[8], true,
                    [&](auto kc) -> v16f {
                      constexpr int k = decltype(kc)::value;
                      if constexpr (k < 4) return fetch_tile(sPark + k * kTileFloats, lane); else if constexpr (k == 4) return k0; else return k1;
                    },
                    [&](auto, const v16f& a0, const v16f& a1) {
                      fused_store_tile(sW, a0, A.ro_h, row0, cnt, 64, lane);
                      fused_store_tile(sW, a1, A.ro_h + 32, row0, cnt, 64, lane);
                      float part = 0.f;
#pragma unroll
                      for (int q = 0; q < 4; ++q) {
                        const v4f w0v = *reinterpret_cast<const v4f*>(sRo + 8 * q + 4 * hh);
                        const v4f w1v = *reinterpret_cast<const v4f*>(sRo + 32 + 8 * q + 4 * hh);
#pragma unroll
                        for (int i = 0; i < 4; ++i) part += silu(a0[4 * q + i]) * w0v[i] + silu(a1[4 * q + i]) * w1v[i];
                      }
                      // E_i = scale_t * factor * sum_{rows of the segment} (both lane halves of a row hold half of it) + shift_t
                      float tot = row_ok ? part : 0.f;
#pragma unroll
                      for (int m = 32; m >= 1; m >>= 1) tot += __shfl_xor(tot, m);
                      if (atom_ok && lane == 0) {
                        float en = tot * A.ro_factor;
                        const int t = A.types[atom];
                        if (A.scales) en *= A.scales[t];
                        if (A.shifts) en += A.shifts[t];
                        A.atom_energy[atom] = en;
                      }
                    });
}

// atoms outside the block the fused kernel covers (other ranks' blocks of an atom partition): E_i = shift_t
__global__ __launch_bounds__(256) void fused_fill_energy_kernel(int64_t N, int64_t a0, int64_t a1, const int32_t* types,
                                                                const float* shifts, float* atom_energy) {
  const int64_t n = int64_t(blockIdx.x) * 256 + threadIdx.x;
  if (n < N && (n < a0 || n >= a1)) atom_energy[n] = shifts ? shifts[types[n]] : 0.f;
}

size_t fused_fwd_lds_bytes(int num_types) {
  return sizeof(u32x4) * 2 * kWStep +
         sizeof(float) * (64 + size_t(num_types) * num_types * 512 + 4 * (kWaveRegion + 32 * kLdY) + 4 * 4 * kTileFloats);
}

int launch_fused_fwd(int pair, const FusedFwdArgs& a, hipStream_t stream) {
  if (a.atom_end <= a.atom0) return AA_OK;
  const size_t smem = fused_fwd_lds_bytes(a.num_types);
  if (smem > 160 * 1024) return fail(AA_ERR_INVALID, "fused forward: LDS budget exceeded");
  if (a.N > 0 && (a.atom0 > 0 || a.atom_end < a.N)) {
    hipLaunchKernelGGL(fused_fill_energy_kernel, dim3((unsigned)((a.N + 255) / 256)), dim3(256), 0, stream, a.N, a.atom0, a.atom_end,
                       a.types, a.shifts, a.atom_energy);
  }
  dim3 grid((unsigned)((a.atom_end - a.atom0 + 3) / 4));
  switch (pair) {
    case 0: {
      const void* fn = (const void*)fused_fwd_kernel<cg::Sig1, cg::Sig0>;
      AA_CHECK_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, int(smem)));
      hipLaunchKernelGGL((fused_fwd_kernel<cg::Sig1, cg::Sig0>), grid, dim3(256), smem, stream, a);
      break;
    }
    case 1: {
      const void* fn = (const void*)fused_fwd_kernel<cg::Sig5, cg::Sig4>;
      AA_CHECK_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, int(smem)));
      hipLaunchKernelGGL((fused_fwd_kernel<cg::Sig5, cg::Sig4>), grid, dim3(256), smem, stream, a);
      break;
    }
    default:
      return fail(AA_ERR_INVALID, "fused forward: unsupported signature pair");
  }
  AA_CHECK_HIP(hipGetLastError());
  return AA_OK;
}

}  // namespace aa
