// Fused forward of the standard 2-layer, 64-wide fp32 Allegro stack -- 16-edge-tile form (gfx950).
//
// Same program as aa_fused.hip (see there for the math and the reference citations): the whole module chain of
// allegro/model/allegro_models.py:222-297 for one center atom with every activation on chip.  The 32-edge-tile form
// needs ~500 registers and ~35 KB of LDS per wave, i.e. ONE wave per SIMD, and a single wave cannot hide its own
// LDS / MFMA / L2 latencies (measured: 6.3 ms at C4 against 5.5 ms for the staged forward).
// Here a wave owns 16 edges and the matrix work runs on v_mfma_f32_16x16x32_bf16:
//
//   * an activation (64 features of the lane's edge) is 16 registers instead of 32: lane = (edge el = lane & 15,
//     group g = lane >> 4), tile t (16 features) holds features 16 t + 4 g + r in register r -- the D layout of the
//     instruction (probed on hardware, tools/ubench/mfma16_probe.hip); the weights' k order is permuted on the host so
//     that two tiles ARE the B operand of a 32-deep chunk of the next layer (no data movement between layers);
//   * an atom's <= 32 edges are two tiles = two waves of the same workgroup; they meet through LDS only where the
//     model sums over the atom's edges (moments, per-atom vectors, the energy);
//   * 8 waves (4 atoms) per workgroup, TWO waves per SIMD: one wave's VALU / LDS work fills the other's MFMA gaps.
//
// Weights stream L2 -> LDS in 12-KB steps shared by the 8 waves, requested two steps ahead (see FusedPipe16).
#include <type_traits>

#include "aa_cg_gen.h"
#include "aa_wave.h"
#include "aa_common.h"
#include "aa_geom.h"
#include "aa_mfma.h"

namespace aa {

namespace {

constexpr int kLdP = 68;   // row stride (floats) of the [rows][64 features] patches
constexpr int kLdJ = 12;   // row stride of sY [32 edges][D] / moment exchange [64 k][D], D <= 9 (l_max <= 2)
constexpr int kSlotFloats = 32 * kLdP + 32 * kLdJ + 2 * 64 * kLdJ + 2 * 9 * 64 + 9 * 64 + 4;  // per atom slot (see carve)

template <int I, int N, class F>
__device__ __forceinline__ void sfor(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    sfor<I + 1, N>(f);
  }
}

struct Act {  // 64 features of the lane's edge in MFMA-D layout: t[tile][r] = feature 16 tile + 4 g + r
  v4f t[4];
};
struct X16 {  // one 32-deep chunk (tiles 2c, 2c+1) split into three bf16 levels: the B operand of 16x16x32
  u32x4 l1, l2, l3;
};

__device__ __forceinline__ void split_chunk(const v4f& lo, const v4f& hi, X16& x) {
  unsigned h1[8], h2[8], h3[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float v = e < 4 ? lo[e] : hi[e - 4];
    h1[e] = f2u(v) & 0xFFFF0000u;
    const float r = v - u2f(h1[e]);
    h2[e] = f2u(r) & 0xFFFF0000u;
    const float r2 = r - u2f(h2[e]);
    h3[e] = f2u(r2) & 0xFFFF0000u;
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    x.l1[q] = (h1[2 * q] >> 16) | h1[2 * q + 1];
    x.l2[q] = (h2[2 * q] >> 16) | h2[2 * q + 1];
    x.l3[q] = (h3[2 * q] >> 16) | h3[2 * q + 1];
  }
}

__device__ __forceinline__ v4f mma16(const u32x4& w, const u32x4& x, v4f acc) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w), __builtin_bit_cast(bf16x8, x), acc, 0, 0, 0);
}

// ---- weight pipeline (as in aa_fused.hip: step S reads LDS buffer S & 1; block S + 2 is requested at the start of
// step S and lands at the end of step S + 1; raw s_barrier with LDS-only fences) -- 512 threads move 768 x 16 B
struct FusedPipe16 {
  u32x4 ra[2], rb[2];
  u32x4* wbuf;
  int tid, lane;
};
__device__ __forceinline__ void lds_barrier16() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}
__device__ __forceinline__ void pipe16_load(const FusedFwdArgs& A, int t, int tid, u32x4* r) {
  const u32x4* s0 = static_cast<const u32x4*>(A.wstep[t][0]);  // 768 contiguous 16-B elements
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  r[0] = s0[tid];
  if (wv < 4) r[1] = (s0 + 512)[tid];  // (wave-uniform: waves 0..3 move the last 256 elements)
}
__device__ __forceinline__ void pipe16_store(u32x4* wbuf, int b, int tid, const u32x4* r) {
  u32x4* d = wbuf + b * kWStep;
  d[tid] = r[0];
  if (tid < 256) d[512 + tid] = r[1];
}
template <int S, int NS>
__device__ __forceinline__ void pipe16_issue(const FusedFwdArgs& A, FusedPipe16& p) {
  if constexpr ((S & 1) == 0)
    pipe16_load(A, (S + 2) % NS, p.tid, p.ra);
  else
    pipe16_load(A, (S + 2) % NS, p.tid, p.rb);
}
template <int S>
__device__ __forceinline__ void pipe16_commit(FusedPipe16& p) {
  if constexpr (((S + 1) & 1) == 0)
    pipe16_store(p.wbuf, 0, p.tid, p.ra);
  else
    pipe16_store(p.wbuf, 1, p.tid, p.rb);
  lds_barrier16();
  __builtin_amdgcn_sched_barrier(0);
}

// 24 MFMAs of one step: 4 output tiles x 6 cross products; fragment f = tile * 3 + level at wb[f * 64 + lane]
__device__ __forceinline__ void mma16_step(const u32x4* wb, int lane, const X16& x, Act& acc) {
  const u32x4* w = wb + lane;
#define AA_W16(T_, L_) w[((T_)*3 + (L_)) * 64]
  {
    const u32x4 a0 = AA_W16(0, 2), a1 = AA_W16(1, 2), a2 = AA_W16(2, 2), a3 = AA_W16(3, 2);  // level 3
    acc.t[0] = mma16(a0, x.l1, acc.t[0]);
    acc.t[1] = mma16(a1, x.l1, acc.t[1]);
    acc.t[2] = mma16(a2, x.l1, acc.t[2]);
    acc.t[3] = mma16(a3, x.l1, acc.t[3]);
  }
  {
    const u32x4 a0 = AA_W16(0, 1), a1 = AA_W16(1, 1), a2 = AA_W16(2, 1), a3 = AA_W16(3, 1);  // level 2
    acc.t[0] = mma16(a0, x.l2, acc.t[0]);
    acc.t[1] = mma16(a1, x.l2, acc.t[1]);
    acc.t[2] = mma16(a2, x.l2, acc.t[2]);
    acc.t[3] = mma16(a3, x.l2, acc.t[3]);
    acc.t[0] = mma16(a0, x.l1, acc.t[0]);
    acc.t[1] = mma16(a1, x.l1, acc.t[1]);
    acc.t[2] = mma16(a2, x.l1, acc.t[2]);
    acc.t[3] = mma16(a3, x.l1, acc.t[3]);
  }
  {
    const u32x4 a0 = AA_W16(0, 0), a1 = AA_W16(1, 0), a2 = AA_W16(2, 0), a3 = AA_W16(3, 0);  // level 1
    acc.t[0] = mma16(a0, x.l3, acc.t[0]);
    acc.t[1] = mma16(a1, x.l3, acc.t[1]);
    acc.t[2] = mma16(a2, x.l3, acc.t[2]);
    acc.t[3] = mma16(a3, x.l3, acc.t[3]);
    acc.t[0] = mma16(a0, x.l2, acc.t[0]);
    acc.t[1] = mma16(a1, x.l2, acc.t[1]);
    acc.t[2] = mma16(a2, x.l2, acc.t[2]);
    acc.t[3] = mma16(a3, x.l2, acc.t[3]);
    acc.t[0] = mma16(a0, x.l1, acc.t[0]);
    acc.t[1] = mma16(a1, x.l1, acc.t[1]);
    acc.t[2] = mma16(a2, x.l1, acc.t[2]);
    acc.t[3] = mma16(a3, x.l1, acc.t[3]);
  }
#undef AA_W16
}

// One linear layer: KC 32-deep operand chunks (op(kc, x) fills the split chunk), NQ groups of 64 output features
// (epi(q, acc) after each).  Steps S0 .. S0 + KC * NQ - 1.  Chunk kc + 1 is split while chunk kc's MFMAs execute.
template <int S0, int NS, int KC, int NQ, class OpF, class EpiF>
__device__ __forceinline__ void layer16(const FusedFwdArgs& A, FusedPipe16& p, OpF&& op, EpiF&& epi) {
  constexpr bool PRE = NQ > 1;
  X16 xs[PRE ? KC : 2];
  if constexpr (PRE) {
    sfor<0, KC>([&](auto kc) { op(kc, xs[decltype(kc)::value]); });
  } else {
    op(std::integral_constant<int, 0>{}, xs[0]);
  }
  sfor<0, NQ>([&](auto qq) {
    Act acc;
#pragma unroll
    for (int t = 0; t < 4; ++t) acc.t[t] = v4f{0.f, 0.f, 0.f, 0.f};
    sfor<0, KC>([&](auto kcc) {
      constexpr int kc = decltype(kcc)::value;
      constexpr int S = S0 + decltype(qq)::value * KC + kc;
      pipe16_issue<S, NS>(A, p);
      mma16_step(p.wbuf + (S & 1) * kWStep, p.lane, xs[PRE ? kc : (kc & 1)], acc);
      if constexpr (!PRE && kc + 1 < KC) op(std::integral_constant<int, kc + 1>{}, xs[(kc + 1) & 1]);
      pipe16_commit<S>(p);
    });
    epi(qq, acc);
  });
}

// chunk c (0 / 1) of an activation as MFMA operand
__device__ __forceinline__ void chunk_of(const Act& a, int c, X16& x) { split_chunk(a.t[2 * c], a.t[2 * c + 1], x); }

// store the wave's 16 rows x 64 features through its LDS patch (rows of the atom's patch it owns) as whole 256-B rows
__device__ __forceinline__ void store_act(float* sP, const Act& a, float* dst, int64_t row0, int nrows, int ld, int lane) {
  const int el = lane & 15, g = lane >> 4;
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int t = 0; t < 4; ++t) *reinterpret_cast<v4f*>(sP + el * kLdP + 16 * t + 4 * g) = a.t[t];
  __builtin_amdgcn_wave_barrier();
  const int pr = lane >> 2, pc = 16 * (lane & 3);
  v4f v[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) v[q] = *reinterpret_cast<const v4f*>(sP + pr * kLdP + pc + 4 * q);
  __builtin_amdgcn_wave_barrier();
  if (pr < nrows) {
    float* o = dst + (row0 + pr) * ld + pc;
#pragma unroll
    for (int q = 0; q < 4; ++q) *reinterpret_cast<v4f*>(o + 4 * q) = v[q];
  }
}

template <bool ACT>
__device__ __forceinline__ void keep_act(const Act& a, Act& k) {
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) k.t[t][r] = ACT ? silu(a.t[t][r]) : a.t[t][r];
}

// scal[e][ch] += w0[e][r][ch] * sum_{a in irrep r} Y[e][a] * B[a][ch]  (w = the 64 channels of irrep RR just computed)
template <int RR>
__device__ __forceinline__ void scal_acc16(const float* sB, const float* Y, const Act& w, int g, Act& sc) {
  constexpr int a0 = RR * RR, na = 2 * RR + 1;
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    v4f T4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int a = 0; a < na; ++a) {
      const v4f b4 = *reinterpret_cast<const v4f*>(sB + (a0 + a) * 64 + 16 * t + 4 * g);
#pragma unroll
      for (int i = 0; i < 4; ++i) T4[i] += Y[a0 + a] * b4[i];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) sc.t[t][i] += w.t[t][i] * T4[i];
    anchor(sc.t[t]);  // (otherwise the arithmetic is sunk below the following MFMA steps and every w tile stays live)
  }
}

}  // namespace

#ifdef AA_FUSED_TIMING
__device__ unsigned long long g_fused16_ticks[32];
#define AA_TICK16(i)                                                                 \
  if (blockIdx.x == gridDim.x / 2 && threadIdx.x == 0) g_fused16_ticks[i] = __builtin_readcyclecounter();
#else
#define AA_TICK16(i)
#endif

template <class Sig0, class Sig1, bool HOLD>
__global__ __launch_bounds__(512, 2) void fused16_fwd_kernel(FusedFwdArgs A) {
  constexpr int D = Sig0::D2, R = Sig0::LMAX + 1;
  static_assert(Sig0::D1 == D && Sig0::DOUT == D && Sig1::D1 == D && Sig1::DOUT == 1 && D <= 9, "standard 2-layer stack, l_max <= 2");
  constexpr int S_L0 = 0, S_L1 = 2, S_P0 = 4, S_L2 = 8, S_L3 = S_L2 + 2 + 2 * R, S_P1 = S_L3 + 4, S_L4 = S_P1 + 4,
                S_L5 = S_L4 + 2, S_L6 = S_L5 + (HOLD ? 0 : 2 * R), S_L7 = S_L6 + 6, S_L8 = S_L7 + 2, NS = S_L8 + 6;
  static_assert(NS % 2 == 0 && NS <= kFusedMaxSteps, "program length");
  // ---- LDS carve
  u32x4* wbuf = reinterpret_cast<u32x4*>(aa_smem);
  float* sRo = reinterpret_cast<float*>(wbuf + 2 * kWStep);  // [64]
  float* sRm = sRo + 64;                                      // [16: T*T <= 9 used] 1/r_max per type pair, then [8] Bessel roots at 16
  float* sTab = sRm + 32;                                     // [T*T][8][64]
  const int ntab = A.num_types * A.num_types * 512;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, slot = wv >> 1, h = wv & 1, el = lane & 15, g = lane >> 4;
  float* sSlot = sTab + ntab + slot * kSlotFloats;
  float* sA = sSlot;                      // [32 rows][kLdP]: moments patch; rows 16 h .. are also the wave's store patch
  float* sY = sA + 32 * kLdP;             // [32 rows][kLdJ] harmonics of the atom's edges
  float* sMp = sY + 32 * kLdJ;            // [2 halves][64 k][kLdJ] partial moments
  float* sX = sMp + 2 * 64 * kLdJ;        // [2 halves][D][64] partial x2s
  float* sB = sX + 2 * 9 * 64;            // [D][64] per-atom Clebsch-Gordan vectors
  float* sE = sB + 9 * 64;                // [4] energy partial of the second tile
  float* sP = sA + 16 * h * kLdP;         // the wave's store patch
  float* sX0 = sTab + ntab + 4 * kSlotFloats + wv * 9 * 64;  // [D][64] the wave's copy of x2s0
  for (int i = tid; i < 64; i += 512) sRo[i] = A.ro_w[i];
  if (tid < A.num_types * A.num_types) sRm[tid] = A.rmax_recip[tid];
  if (tid >= 16 && tid < 24) sRm[tid] = A.embed_kind == 0 ? A.bessel_w[tid - 16] : 0.f;
  for (int i = tid; i < ntab; i += 512) sTab[i] = A.emb_tab[i];
  FusedPipe16 p;
  p.wbuf = wbuf;
  p.tid = tid;
  p.lane = lane;
  {
    u32x4 r[2];
    pipe16_load(A, 0, tid, r);
    pipe16_store(wbuf, 0, tid, r);
    pipe16_load(A, 1, tid, p.rb);
  }
  // (path weights of the lane's channel are re-read from L2 where the per-atom vectors are formed: 14 registers that
  //  would otherwise be live through the whole kernel)
  auto load_wp = [&](float* wp0, float* wp1) {
#pragma unroll
    for (int q = 0; q < Sig0::P; ++q) wp0[q] = A.coupling ? A.tpw0[lane * Sig0::P + q] : A.tpw0[q];
#pragma unroll
    for (int q = 0; q < Sig1::P; ++q) wp1[q] = A.coupling ? A.tpw1[lane * Sig1::P + q] : A.tpw1[q];
  };
  // ---- the wave's atom and 16-edge tile
  const int64_t atom = A.atom0 + int64_t(blockIdx.x) * 4 + slot;
  const bool atom_ok = atom < A.atom_end;
  int beg = 0, cnt = 0;
  if (atom_ok) {
    beg = A.rowptr[atom];
    cnt = A.rowptr[atom + 1] - beg;
  }
  beg = __builtin_amdgcn_readfirstlane(beg);
  cnt = __builtin_amdgcn_readfirstlane(cnt);
  const int nrows = cnt - 16 * h < 0 ? 0 : (cnt - 16 * h > 16 ? 16 : cnt - 16 * h);
  const bool row_ok = el < nrows;
  const int64_t row0 = int64_t(beg) + 16 * h;
  AA_TICK16(0)
  float Y[D], basis[8];
  int pair = 0;
  {
    float vx = 1.f, vy = 0.f, vz = 0.f, x = 0.5f;
    if (row_ok) {
      const int64_t e = row0 + el;
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
    float Yf[16];
    sh_eval<float>(Sig0::LMAX, nx, ny, nz, Yf);
#pragma unroll
    for (int m = 0; m < D; ++m) Y[m] = row_ok ? Yf[m] : 0.f;
    if (row_ok && g == 0) {
      const int64_t e = row0 + el;
      *reinterpret_cast<v4f*>(A.vec + 4 * e) = v4f{nx, ny, nz, rr};
      if (A.sh) {
#pragma unroll
        for (int m = 0; m < D; ++m) A.sh[e * D + m] = Yf[m];
      }
    }
    if (g == 0) {
#pragma unroll
      for (int m = 0; m < kLdJ; ++m) sY[(16 * h + el) * kLdJ + m] = m < D ? Y[m] : 0.f;
    }
    lds_barrier16();  // tables, first weight step, harmonics
    if (row_ok) x = rr * sRm[pair];
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
      for (int n = 0; n < 8; ++n) basis[n] = aa_sin(sRm[16 + n] * x) * fx;
    }
  }
  AA_TICK16(1)
  // ---- two-body embedding of the lane's 16 features: emb0[c] = sum_n basis[n] * tab[pair][n][c]
  Act em;
  {
    const float* tb = sTab + pair * 512 + 4 * g;
#pragma unroll
    for (int t = 0; t < 4; ++t) em.t[t] = v4f{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int n = 0; n < 8; ++n) {
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const v4f t4 = *reinterpret_cast<const v4f*>(tb + n * 64 + 16 * t);
#pragma unroll
        for (int i = 0; i < 4; ++i) em.t[t][i] += basis[n] * t4[i];
      }
    }
  }
  Act k, tb, la, sc;
  Act w0t[HOLD ? R : 1];
  AA_TICK16(2)
  // ---- L0 / L1: scalar_embed_mlp
  layer16<S_L0, NS, 2, 1>(A, p, [&](auto kc, X16& x) { chunk_of(em, decltype(kc)::value, x); },
                          [&](auto, const Act& a) {
                            store_act(sP, a, A.se_h, row0, nrows, 64, lane);
                            keep_act<true>(a, k);
                          });
  AA_TICK16(3)
  layer16<S_L1, NS, 2, 1>(A, p, [&](auto kc, X16& x) { chunk_of(k, decltype(kc)::value, x); },
                          [&](auto, const Act& a) {
                            store_act(sP, a, A.emb, row0, nrows, 64, lane);
                            em = a;
                          });
  AA_TICK16(4)
  // ---- per-atom part of a layer: partial moments of this tile -> (exchange) -> x2s -> per-atom vector B -> sB
  auto per_atom = [&](auto s0c, const Act& a, float* x2s) {
    constexpr int SP = decltype(s0c)::value;
    // the wave's 16 rows to its half of the atom patch, then lane = k walks them
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int t = 0; t < 4; ++t) *reinterpret_cast<v4f*>(sP + el * kLdP + 16 * t + 4 * g) = a.t[t];
    __builtin_amdgcn_wave_barrier();
    float M[D];
#pragma unroll
    for (int j = 0; j < D; ++j) M[j] = 0.f;
#pragma unroll 4
    for (int e = 0; e < 16; ++e) {
      const float av = sP[e * kLdP + lane];
      float y[12];
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        const v4f yy = *reinterpret_cast<const v4f*>(sY + (16 * h + e) * kLdJ + 4 * q);
#pragma unroll
        for (int i = 0; i < 4; ++i) y[4 * q + i] = yy[i];
      }
#pragma unroll
      for (int j = 0; j < D; ++j) M[j] += y[j] * av;
    }
    {
      float* mp = sMp + (h * 64 + lane) * kLdJ;
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        v4f mm;
#pragma unroll
        for (int i = 0; i < 4; ++i) mm[i] = 4 * q + i < D ? M[4 * q + i] : 0.f;
        *reinterpret_cast<v4f*>(mp + 4 * q) = mm;
      }
    }
    lds_barrier16();  // both tiles' partial moments are in LDS
    // projection: this wave covers k rows 32 h .. 32 h + 31 (blocks 2 h, 2 h + 1 of the env-weight matrix)
    float xp[D];
#pragma unroll
    for (int j = 0; j < D; ++j) xp[j] = 0.f;
    sfor<0, 4>([&](auto cc) {
      constexpr int c = decltype(cc)::value;
      constexpr int S = SP + c;
      pipe16_issue<S, NS>(A, p);
      if ((c >> 1) == h) {
        const float* wf = reinterpret_cast<const float*>(p.wbuf + (S & 1) * kWStep) + lane;
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
          float w[R], m[12];
#pragma unroll
          for (int r = 0; r < R; ++r) w[r] = wf[(kk * R + r) * 64];
#pragma unroll
          for (int q = 0; q < 3; ++q) {
            const v4f m0 = *reinterpret_cast<const v4f*>(sMp + (16 * c + kk) * kLdJ + 4 * q);
            const v4f m1 = *reinterpret_cast<const v4f*>(sMp + (64 + 16 * c + kk) * kLdJ + 4 * q);
#pragma unroll
            for (int i = 0; i < 4; ++i) m[4 * q + i] = m0[i] + m1[i];
          }
#pragma unroll
          for (int j = 0; j < D; ++j) xp[j] += m[j] * w[r_of<0>(j)];
          if ((kk & 3) == 3) __builtin_amdgcn_sched_barrier(0);
        }
      }
      if constexpr (c == 3) {
#pragma unroll
        for (int j = 0; j < D; ++j) sX[(h * 9 + j) * 64 + lane] = xp[j];
      }
      pipe16_commit<S>(p);
    });
#pragma unroll
    for (int j = 0; j < D; ++j) x2s[j] = A.sf * (sX[j * 64 + lane] + sX[(9 + j) * 64 + lane]);
  };
  {
    float x2s0[D];
    per_atom(std::integral_constant<int, S_P0>{}, em, x2s0);
    if (atom_ok && h == 0) {
#pragma unroll
      for (int j = 0; j < D; ++j) A.x2s0[(atom * D + j) * 64 + lane] = x2s0[j];
    }
    float e0[D], B0[D], wp0[Sig0::P], wp1[Sig1::P];
    load_wp(wp0, wp1);
#pragma unroll
    for (int q = 0; q < D; ++q) e0[q] = q == 0 ? 1.f : 0.f;
    Sig0::template bx1<float>(e0, x2s0, wp0, B0);
    // x2s0 is needed again for the second layer's per-atom vector: parked in the wave's LDS slot until then
#pragma unroll
    for (int j = 0; j < D; ++j) sX0[j * 64 + lane] = x2s0[j];
    if (h == 0) {
#pragma unroll
      for (int a = 0; a < D; ++a) sB[a * 64 + lane] = B0[a];
    }
  }
  AA_TICK16(5)
  // ---- L2: [two-body | w0 irrep 0 | 1 | ..]; scal0 accumulated per irrep group (sB is visible after the first barrier)
#pragma unroll
  for (int t = 0; t < 4; ++t) sc.t[t] = v4f{0.f, 0.f, 0.f, 0.f};
  layer16<S_L2, NS, 2, 1 + R>(A, p, [&](auto kc, X16& x) { chunk_of(em, decltype(kc)::value, x); },
                              [&](auto qq, const Act& a) {
                                constexpr int q = decltype(qq)::value;
                                if constexpr (q == 0) {
                                  tb = a;
                                } else {
                                  if (A.w0) store_act(sP, a, A.w0 + (q - 1) * 64, row0, nrows, 64 * R, lane);
                                  scal_acc16<q - 1>(sB, Y, a, g, sc);
                                  if constexpr (HOLD) w0t[q - 1] = a;
                                }
                              });
  AA_TICK16(6)
  // ---- L3: latent 0 hidden layer on [two-body | scal0]
  layer16<S_L3, NS, 4, 1>(A, p,
                          [&](auto kc, X16& x) {
                            constexpr int c = decltype(kc)::value;
                            if constexpr (c < 2) chunk_of(tb, c, x); else chunk_of(sc, c - 2, x);
                          },
                          [&](auto, const Act& a) {
                            store_act(sP, a, A.lat_h0, row0, nrows, 64, lane);
                            keep_act<true>(a, k);
                          });
  AA_TICK16(7)
  {
    float x2s1[D];
    per_atom(std::integral_constant<int, S_P1>{}, k, x2s1);
    if (atom_ok && h == 0) {
#pragma unroll
      for (int j = 0; j < D; ++j) A.x2s1[(atom * D + j) * 64 + lane] = x2s1[j];
    }
    float one[1] = {1.f}, v[D], B1[D], x2s0[D], wp0[Sig0::P], wp1[Sig1::P];
    load_wp(wp0, wp1);
#pragma unroll
    for (int j = 0; j < D; ++j) x2s0[j] = sX0[j * 64 + lane];
    Sig1::template bx1<float>(one, x2s1, wp1, v);
    Sig0::template bx1<float>(v, x2s0, wp0, B1);
    if (h == 0) {
#pragma unroll
      for (int a = 0; a < D; ++a) sB[a * 64 + lane] = B1[a];
    }
  }
  AA_TICK16(8)
  // ---- L4: latent 0 output layer -> lat0
  layer16<S_L4, NS, 2, 1>(A, p, [&](auto kc, X16& x) { chunk_of(k, decltype(kc)::value, x); },
                          [&](auto, const Act& a) { la = a; });
  AA_TICK16(9)
  // ---- L5: layer-1 scalars with B1 (w0 held, or recomputed from the embedding)
#pragma unroll
  for (int t = 0; t < 4; ++t) sc.t[t] = v4f{0.f, 0.f, 0.f, 0.f};
  if constexpr (HOLD) {
    sfor<0, R>([&](auto rr) { scal_acc16<decltype(rr)::value>(sB, Y, w0t[decltype(rr)::value], g, sc); });
  } else {
    // (the w0 columns of the first-stage matrix again: groups 1 .. R of the same packed matrix)
    layer16<S_L5, NS, 2, R>(A, p, [&](auto kc, X16& x) { chunk_of(em, decltype(kc)::value, x); },
                            [&](auto qq, const Act& a) { scal_acc16<decltype(qq)::value>(sB, Y, a, g, sc); });
  }
  AA_TICK16(10)
  // ---- L6: latent 1 hidden layer on [two-body | lat0 | scal1]
  layer16<S_L6, NS, 6, 1>(A, p,
                          [&](auto kc, X16& x) {
                            constexpr int c = decltype(kc)::value;
                            if constexpr (c < 2) chunk_of(tb, c, x); else if constexpr (c < 4) chunk_of(la, c - 2, x); else chunk_of(sc, c - 4, x);
                          },
                          [&](auto, const Act& a) {
                            store_act(sP, a, A.lat_h1, row0, nrows, 64, lane);
                            keep_act<true>(a, k);
                          });
  AA_TICK16(11)
  // ---- L7: latent 1 output layer -> lat1
  layer16<S_L7, NS, 2, 1>(A, p, [&](auto kc, X16& x) { chunk_of(k, decltype(kc)::value, x); },
                          [&](auto, const Act& a) { k = a; });
  AA_TICK16(12)
  // ---- L8: edge readout hidden layer on [two-body | lat0 | lat1]; last linear layer + edge sum in the epilogue
  float tot = 0.f;
  layer16<S_L8, NS, 6, 1>(A, p,
                          [&](auto kc, X16& x) {
                            constexpr int c = decltype(kc)::value;
                            if constexpr (c < 2) chunk_of(tb, c, x); else if constexpr (c < 4) chunk_of(la, c - 2, x); else chunk_of(k, c - 4, x);
                          },
                          [&](auto, const Act& a) {
                            store_act(sP, a, A.ro_h, row0, nrows, 64, lane);
                            float part = 0.f;
#pragma unroll
                            for (int t = 0; t < 4; ++t) {
                              const v4f wv4 = *reinterpret_cast<const v4f*>(sRo + 16 * t + 4 * g);
#pragma unroll
                              for (int i = 0; i < 4; ++i) part += silu(a.t[t][i]) * wv4[i];
                            }
                            tot = row_ok ? part : 0.f;
#pragma unroll
                            for (int m = 32; m >= 1; m >>= 1) tot += __shfl_xor(tot, m);
                          });
  // E_i = scale_t * factor * (sum over both tiles) + shift_t
  if (h == 1 && lane == 0) sE[0] = tot;
  lds_barrier16();
  if (atom_ok && h == 0 && lane == 0) {
    float en = (tot + sE[0]) * A.ro_factor;
    const int t = A.types[atom];
    if (A.scales) en *= A.scales[t];
    if (A.shifts) en += A.shifts[t];
    A.atom_energy[atom] = en;
  }
  AA_TICK16(13)
}

__global__ __launch_bounds__(256) void fused16_fill_energy_kernel(int64_t N, int64_t a0, int64_t a1, const int32_t* types,
                                                                  const float* shifts, float* atom_energy) {
  const int64_t n = int64_t(blockIdx.x) * 256 + threadIdx.x;
  if (n < N && (n < a0 || n >= a1)) atom_energy[n] = shifts ? shifts[types[n]] : 0.f;
}

size_t fused16_lds_bytes(int num_types) {
  return sizeof(u32x4) * 2 * kWStep + sizeof(float) * (64 + 32 + size_t(num_types) * num_types * 512 + 4 * kSlotFloats + 8 * 9 * 64);
}

int launch_fused16_fwd(int pair, bool hold_w0, const FusedFwdArgs& a, hipStream_t stream) {
  if (a.atom_end <= a.atom0) return AA_OK;
  const size_t smem = fused16_lds_bytes(a.num_types);
  if (smem > 160 * 1024) return fail(AA_ERR_INVALID, "fused forward (16-edge tiles): LDS budget exceeded");
  if (a.N > 0 && (a.atom0 > 0 || a.atom_end < a.N)) {
    hipLaunchKernelGGL(fused16_fill_energy_kernel, dim3((unsigned)((a.N + 255) / 256)), dim3(256), 0, stream, a.N, a.atom0,
                       a.atom_end, a.types, a.shifts, a.atom_energy);
  }
  dim3 grid((unsigned)((a.atom_end - a.atom0 + 3) / 4));
#define AA_F16_LAUNCH(S0_, S1_, H_)                                                                            \
  {                                                                                                            \
    const void* fn = (const void*)fused16_fwd_kernel<cg::S0_, cg::S1_, H_>;                                    \
    AA_CHECK_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, int(smem)));              \
    hipLaunchKernelGGL((fused16_fwd_kernel<cg::S0_, cg::S1_, H_>), grid, dim3(512), smem, stream, a);          \
  }
  if (pair == 0) {
    if (hold_w0) AA_F16_LAUNCH(Sig1, Sig0, true) else AA_F16_LAUNCH(Sig1, Sig0, false)
  } else if (pair == 1) {
    if (hold_w0) AA_F16_LAUNCH(Sig5, Sig4, true) else AA_F16_LAUNCH(Sig5, Sig4, false)
  } else {
    return fail(AA_ERR_INVALID, "fused forward: unsupported signature pair");
  }
#undef AA_F16_LAUNCH
  AA_CHECK_HIP(hipGetLastError());
#ifdef AA_FUSED_TIMING
  {
    static int calls = 0;
    if (++calls == 8) {
      unsigned long long t[32];
      AA_CHECK_HIP(hipStreamSynchronize(stream));
      AA_CHECK_HIP(hipMemcpyFromSymbol(t, HIP_SYMBOL(g_fused16_ticks), sizeof(t)));
      static const char* nm[13] = {"geometry", "emb0", "L0", "L1", "TPA0", "L2", "L3", "TPA1", "L4", "L5", "L6", "L7", "L8"};
      for (int i = 0; i < 13; ++i) fprintf(stderr, "[fused16 timing] %-16s %8llu cycles\n", nm[i], t[i + 1] - t[i]);
      fprintf(stderr, "[fused16 timing] %-16s %8llu cycles\n", "total", t[13] - t[0]);
    }
  }
#endif
  return AA_OK;
}

}  // namespace aa
