// Tensor-product track of an L-layer Allegro stack in "per-atom operator" form (gfx950).
//
// For a fixed center atom n and channel ch every layer of the track (allegro/nn/_allegro.py:263-278 +
// allegro/nn/_strided/_contract.py:195-251) is LINEAR in the incoming edge features once the per-atom sums
// x2s_i[n] (the scattered, env-weighted harmonics, _contract.py:195-205) are fixed:
//     tf_{i+1}[e] = Sig_i(tf_i[e], x2s_i[n]) = M_i(n) tf_i[e],      tf_0[e] = x1[e] = Y[e] (.) w0[e]
// and the track is only ever read through its scalar channel  scal_l[e] = tf_{l+1}[e][0]  (_allegro.py:272-278).
// Therefore
//     scal_l[e] = < x1[e], B_l(n) >,     B_l = M_0^T M_1^T ... M_l^T e_0                      (forward)
//     d x1[e]   = sum_m G_m[e] B_m(n)                                                         (reverse, edges)
//     d x2s_l   = sum_{m>=l} Sig_l^T_x2( M_{l+1}^T..M_m^T e_0 ,  M_{l-1}..M_0 Q_m ),   Q_m = sum_e G_m[e] x1[e]
// with G_m = dE/d scal_m.  The Clebsch-Gordan contractions (up to 611 non-zeros each at l_max = 3) are evaluated
// a handful of times per ATOM; per EDGE only O(D) dot products remain, for any number of layers, and no
// [E, u, d] tensor-feature array exists at all.  The env weights enter through the moments
//     x2s_l[n,j,ch] = f * sum_k (sum_e Y[e,j] act(a_l[e,k])) Wenv_l[k, r(j), ch]
// exactly as in the 2-layer kernels of aa_tp_spec.hip (which remain the tuned path for L = 2, u = 64).
//
// Launch: one workgroup per atom, one wave per 64-channel slice (u = 64, 128, ...); lane = channel.
#include <type_traits>

#include "aa_cg_gen.h"
#include "aa_common.h"
#include "aa_wave.h"

namespace aa {

namespace {

constexpr int kOpMaxD = 31;   // widest irreps vector on the track (l_max 3, L 3: 0e+1e+1o+2e+2o+3e+3o)
constexpr int kOpMaxKa = 128; // widest env input

struct NoSig {};
template <class A, class B, class C, int N>
struct SigChain {
  typedef A S0;
  typedef B S1;
  typedef C S2;
  static constexpr int L = N;
};
template <class Ch, int I>
struct SigAt;
template <class Ch>
struct SigAt<Ch, 0> {
  typedef typename Ch::S0 type;
};
template <class Ch>
struct SigAt<Ch, 1> {
  typedef typename Ch::S1 type;
};
template <class Ch>
struct SigAt<Ch, 2> {
  typedef typename Ch::S2 type;
};

// per-atom operands of layer I for this lane's channel: x2s_I (from HBM) and the path weights
template <class Sig, typename T>
__device__ __forceinline__ void load_layer(const TpOpArgs& a, int i, int64_t atom, int ch, T* x2s, T* wp) {
  const T* xp = static_cast<const T*>(a.x2s[i]) + atom * Sig::D2 * int64_t(a.u) + ch;
#pragma unroll
  for (int j = 0; j < Sig::D2; ++j) x2s[j] = xp[int64_t(j) * a.u];
  const T* W = static_cast<const T*>(a.tpw[i]);
#pragma unroll
  for (int p = 0; p < Sig::P; ++p) wp[p] = a.coupling ? W[ch * Sig::P + p] : W[p];
}

// g <- M_LO^T ... M_HI^T g   (g enters with DOUT_HI entries, leaves with D1_LO entries)
template <class Ch, int HI, int LO, typename T>
__device__ __forceinline__ void chain_bx1(const TpOpArgs& a, int64_t atom, int ch, T* g) {
  if constexpr (HI >= LO) {
    typedef typename SigAt<Ch, HI>::type S;
    T x2s[S::D2], wp[S::P], t[S::D1];
    load_layer<S, T>(a, HI, atom, ch, x2s, wp);
    S::template bx1<T>(g, x2s, wp, t);
#pragma unroll
    for (int i = 0; i < S::D1; ++i) g[i] = t[i];
    chain_bx1<Ch, HI - 1, LO, T>(a, atom, ch, g);
  }
}

// t <- M_HI ... M_LO t   (t enters with D1_LO entries, leaves with DOUT_HI entries)
template <class Ch, int LO, int HI, typename T>
__device__ __forceinline__ void chain_fwd(const TpOpArgs& a, int64_t atom, int ch, T* t) {
  if constexpr (LO <= HI) {
    typedef typename SigAt<Ch, LO>::type S;
    T x2s[S::D2], wp[S::P], o[S::DOUT];
    load_layer<S, T>(a, LO, atom, ch, x2s, wp);
    S::template fwd<T>(t, x2s, wp, o);
#pragma unroll
    for (int k = 0; k < S::DOUT; ++k) t[k] = o[k];
    chain_fwd<Ch, LO + 1, HI, T>(a, atom, ch, t);
  }
}

// B_M = M_0^T .. M_M^T e_0
template <class Ch, int M, typename T>
__device__ __forceinline__ void atom_vector(const TpOpArgs& a, int64_t atom, int ch, T* g) {
#pragma unroll
  for (int k = 0; k < kOpMaxD; ++k) g[k] = k == 0 ? T(1) : T(0);
  chain_bx1<Ch, M, 0, T>(a, atom, ch, g);
}

// term m of d x2s_L:  Sig_L^T_x2( M_{L+1}^T..M_m^T e_0 , M_{L-1}..M_0 q )  accumulated into g2
template <class Ch, int LI, int M, typename T>
__device__ __forceinline__ void x2s_grad_term(const TpOpArgs& a, int64_t atom, int ch, const T* q, T* g2) {
  typedef typename SigAt<Ch, LI>::type S;
  T t[kOpMaxD], lv[kOpMaxD];
#pragma unroll
  for (int k = 0; k < kOpMaxD; ++k) {
    t[k] = k < SigAt<Ch, 0>::type::D1 ? q[k] : T(0);
    lv[k] = k == 0 ? T(1) : T(0);
  }
  chain_fwd<Ch, 0, LI - 1, T>(a, atom, ch, t);
  chain_bx1<Ch, M, LI + 1, T>(a, atom, ch, lv);
  T x2s[S::D2], wp[S::P], o[S::D2];
  load_layer<S, T>(a, LI, atom, ch, x2s, wp);
  S::template bx2<T>(lv, t, wp, o);
#pragma unroll
  for (int j = 0; j < S::D2; ++j) g2[j] += o[j];
}

template <class Ch, int LI, int M, typename T>
__device__ __forceinline__ void x2s_grad_terms(const TpOpArgs& a, int64_t atom, int ch, const T* q_own, T* g2) {
  if constexpr (M < Ch::L) {
    constexpr int D = SigAt<Ch, 0>::type::D1;
    T q[D];
    if (M == LI) {
#pragma unroll
      for (int i = 0; i < D; ++i) q[i] = q_own[i];
    } else {
      const T* qp = static_cast<const T*>(a.q) + ((atom * Ch::L + M) * D) * int64_t(a.u) + ch;
#pragma unroll
      for (int i = 0; i < D; ++i) q[i] = qp[int64_t(i) * a.u];
    }
    x2s_grad_term<Ch, LI, M, T>(a, atom, ch, q, g2);
    x2s_grad_terms<Ch, LI, M + 1, T>(a, atom, ch, q_own, g2);
  }
}

// B_m for m = 0..L-1 into b[m][.]
template <class Ch, int M, typename T, int D>
__device__ __forceinline__ void all_atom_vectors(const TpOpArgs& a, int64_t atom, int ch, T (*b)[D]) {
  if constexpr (M < Ch::L) {
    T g[kOpMaxD];
    atom_vector<Ch, M, T>(a, atom, ch, g);
#pragma unroll
    for (int i = 0; i < D; ++i) b[M][i] = g[i];
    all_atom_vectors<Ch, M + 1, T, D>(a, atom, ch, b);
  }
}

}  // namespace

// ---- forward of layer LI: x2s_LI from the moments of its env input, B_LI, then scal_LI[e] = <x1[e], B_LI>
template <class Ch, int LI, typename T>
__global__ __launch_bounds__(256) void tp_op_fwd_kernel(TpOpArgs a) {
  typedef typename SigAt<Ch, LI>::type S;
  constexpr int D = S::D2, R = S::LMAX + 1;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int64_t atom = a.atom0 + blockIdx.x;
  const int u = a.u, ch = wv * 64 + lane;
  const int beg = __builtin_amdgcn_readfirstlane(a.rowptr[atom]), end = __builtin_amdgcn_readfirstlane(a.rowptr[atom + 1]);
  if (beg >= end) return;  // (its x2s rows are only ever read by its own, absent, edges)
  T* sM = reinterpret_cast<T*>(aa_smem) + size_t(wv) * D * a.ka_lds;  // wave-private [D][ka]
  const T* sh = static_cast<const T*>(a.sh);
  const T* av = static_cast<const T*>(a.a);
  const int ka = a.ka;
  if (!a.proj_gemm) {  // (proj_gemm: x2s_LI is already in HBM, written by the batched projection launch)
  // moments M[j][k] = sum_e Y[e,j] act(a[e,k])   (every slice recomputes them: they do not depend on the channel)
  if (a.mbuf) {  // split form: tp_op_moments_kernel streamed the edges; fetch this atom's D x ka block
    const T* mp = static_cast<const T*>(a.mbuf) + (atom * D) * int64_t(ka);
    for (int kb = 0; kb < ka; kb += 64)
#pragma unroll
      for (int j = 0; j < D; ++j) sM[j * a.ka_lds + kb + lane] = mp[int64_t(j) * ka + kb + lane];
  } else
  for (int kb = 0; kb < ka; kb += 64) {
    T m[D];
#pragma unroll
    for (int j = 0; j < D; ++j) m[j] = T(0);
#pragma unroll 4
    for (int s = beg; s < end; ++s) {
      T x = av[int64_t(s) * a.ld_a + kb + lane];
      if (a.act) x = silu(x);
      const T* y = sh + int64_t(s) * a.ld_sh;
#pragma unroll
      for (int j = 0; j < D; ++j) m[j] += y[j] * x;
    }
#pragma unroll
    for (int j = 0; j < D; ++j) sM[j * a.ka_lds + kb + lane] = m[j];
  }
  __builtin_amdgcn_wave_barrier();
  T x2s[D];
#pragma unroll
  for (int j = 0; j < D; ++j) x2s[j] = T(0);
  {
    const T* Wk = static_cast<const T*>(a.wk) + ch;  // [ka][R][u]
#pragma unroll 4
    for (int k = 0; k < ka; ++k) {
      T wv3[R];
#pragma unroll
      for (int r = 0; r < R; ++r) wv3[r] = Wk[(int64_t(k) * R + r) * u];
#pragma unroll
      for (int j = 0; j < D; ++j) x2s[j] += sM[j * a.ka_lds + k] * wv3[r_of<0>(j)];
    }
  }
  {
    const T sf = T(a.sf);
    T* xo = static_cast<T*>(const_cast<void*>(a.x2s[LI])) + atom * D * int64_t(u) + ch;
#pragma unroll
    for (int j = 0; j < D; ++j) {
      x2s[j] *= sf;
      xo[int64_t(j) * u] = x2s[j];
    }
  }
  // the chain reads x2s_LI back through the same lane's own store (same thread, same address: program order)
  __threadfence_block();
  }
  T b[kOpMaxD];
  atom_vector<Ch, LI, T>(a, atom, ch, b);
  constexpr int D1 = SigAt<Ch, 0>::type::D1;
  if (a.bvec) {  // split form: tp_op_edge_fwd_kernel streams the edges
    T* bp = static_cast<T*>(a.bvec) + ((atom * a.num_layers + LI) * D1) * int64_t(u) + ch;
#pragma unroll
    for (int i = 0; i < D1; ++i) bp[int64_t(i) * u] = b[i];
    return;
  }
  const T* w0 = static_cast<const T*>(a.w0) + ch;
  T* sc = static_cast<T*>(a.scal) + ch;
#pragma unroll 2
  for (int s = beg; s < end; ++s) {
    const T* y = sh + int64_t(s) * a.ld_sh;
    T wr[R];
#pragma unroll
    for (int r = 0; r < R; ++r) wr[r] = w0[int64_t(s) * a.ld_w0 + r * u];
    T acc = T(0);
#pragma unroll
    for (int i = 0; i < D1; ++i) acc += (y[i] * wr[r_of<0>(i)]) * b[i];
    sc[int64_t(s) * a.ld_scal] = acc;
  }
}

// ---- split form, lean edge-streaming kernels (few registers => many waves per SIMD; the per-atom vectors come
// ---- from HBM).  One workgroup per atom, one wave per 64-channel slice (edge_env: per 64-wide env-input block).
template <typename T, int D1, int R>
__global__ __launch_bounds__(256) void tp_op_edge_fwd_kernel(TpOpArgs a, int layer) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int64_t atom = a.atom0 + blockIdx.x;
  const int u = a.u, ch = wv * 64 + lane;
  const int beg = __builtin_amdgcn_readfirstlane(a.rowptr[atom]), end = __builtin_amdgcn_readfirstlane(a.rowptr[atom + 1]);
  if (beg >= end) return;
  T b[D1];
  {
    const T* bp = static_cast<const T*>(a.bvec) + ((atom * a.num_layers + layer) * D1) * int64_t(u) + ch;
#pragma unroll
    for (int i = 0; i < D1; ++i) b[i] = bp[int64_t(i) * u];
  }
  const T* sh = static_cast<const T*>(a.sh);
  const T* w0 = static_cast<const T*>(a.w0) + ch;
  T* sc = static_cast<T*>(a.scal) + ch;
#pragma unroll 4
  for (int s = beg; s < end; ++s) {
    const T* y = sh + int64_t(s) * a.ld_sh;
    T wr[R];
#pragma unroll
    for (int r = 0; r < R; ++r) wr[r] = ld_stream(w0 + int64_t(s) * a.ld_w0 + r * u);
    T acc = T(0);
#pragma unroll
    for (int i = 0; i < D1; ++i) acc += (y[i] * wr[r_of<0>(i)]) * b[i];
    st_stream(sc + int64_t(s) * a.ld_scal, acc);
  }
}

// moments M[n][j][k] = sum_e Y[e,j] act(a[e,k]); one wave per 64-wide block of the env input
template <typename T, int D>
__global__ __launch_bounds__(256) void tp_op_moments_kernel(TpOpArgs a) {
  const int lane = threadIdx.x & 63, blk = threadIdx.x >> 6;
  const int64_t atom = a.atom0 + blockIdx.x;
  const int beg = __builtin_amdgcn_readfirstlane(a.rowptr[atom]), end = __builtin_amdgcn_readfirstlane(a.rowptr[atom + 1]);
  if (beg >= end) return;
  const int ka = a.ka, k = blk * 64 + lane;
  const T* sh = static_cast<const T*>(a.sh);
  const T* av = static_cast<const T*>(a.a) + k;
  T m[D];
#pragma unroll
  for (int j = 0; j < D; ++j) m[j] = T(0);
#pragma unroll 4
  for (int s = beg; s < end; ++s) {
    T x = ld_stream(av + int64_t(s) * a.ld_a);
    if (a.act) x = silu(x);
    const T* y = sh + int64_t(s) * a.ld_sh;
#pragma unroll
    for (int j = 0; j < D; ++j) m[j] += y[j] * x;
  }
  T* mp = static_cast<T*>(a.mbuf) + (atom * D) * int64_t(ka) + k;
#pragma unroll
  for (int j = 0; j < D; ++j) mp[int64_t(j) * ka] = m[j];
}

// Q_layer = sum_e G_layer[e] x1[e]; FIRST (layer 0): also d w0[e] and d Y[e] through x1 with all L gradient streams
template <typename T, int D1, int R, int L, bool FIRST>
__global__ __launch_bounds__(256) void tp_op_edge_bwd_kernel(TpOpArgs a, int layer) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int64_t atom = a.atom0 + blockIdx.x;
  const int u = a.u, ch = wv * 64 + lane;
  const int beg = __builtin_amdgcn_readfirstlane(a.rowptr[atom]), end = __builtin_amdgcn_readfirstlane(a.rowptr[atom + 1]);
  if (beg >= end) return;
  const T* sh = static_cast<const T*>(a.sh);
  const T* w0 = static_cast<const T*>(a.w0) + ch;
  const int64_t ED = int64_t(a.E) * a.ld_gsh;
  T bm[FIRST ? L : 1][D1];
  if constexpr (FIRST) {
    const T* bp = static_cast<const T*>(a.bvec) + (atom * L * D1) * int64_t(u) + ch;
#pragma unroll
    for (int m = 0; m < L; ++m)
#pragma unroll
      for (int i = 0; i < D1; ++i) bm[m][i] = bp[int64_t(m * D1 + i) * u];
  }
  T q[D1];
#pragma unroll
  for (int i = 0; i < D1; ++i) q[i] = T(0);
  const T* gl = static_cast<const T*>(a.gscal[FIRST ? 0 : 1]) + ch;  // (host passes this layer's stream in slot 1)
  T* gw0 = static_cast<T*>(a.g_w0) + ch;
  T* gsx = static_cast<T*>(a.gsh_x1) + int64_t(wv) * ED;
#pragma unroll 2
  for (int s = beg; s < end; ++s) {
    const T* y = sh + int64_t(s) * a.ld_sh;
    T wr[R];
#pragma unroll
    for (int r = 0; r < R; ++r) wr[r] = ld_stream(w0 + int64_t(s) * a.ld_w0 + r * u);
    const T g = ld_stream(gl + int64_t(s) * a.ld_gscal);
#pragma unroll
    for (int i = 0; i < D1; ++i) q[i] += g * (y[i] * wr[r_of<0>(i)]);
    if constexpr (FIRST) {
      T gm_[L];
      gm_[0] = g;
#pragma unroll
      for (int m = 1; m < L; ++m) gm_[m] = static_cast<const T*>(a.gscal[m])[int64_t(s) * a.ld_gscal + ch];
      T gw[R], gy[D1];
#pragma unroll
      for (int r = 0; r < R; ++r) gw[r] = T(0);
#pragma unroll
      for (int i = 0; i < D1; ++i) {
        T gx = T(0);
#pragma unroll
        for (int m = 0; m < L; ++m) gx += gm_[m] * bm[m][i];
        gw[r_of<0>(i)] += gx * y[i];
        gy[i] = gx * wr[r_of<0>(i)];
      }
#pragma unroll
      for (int r = 0; r < R; ++r) st_stream(gw0 + int64_t(s) * a.ld_gw0 + r * u, gw[r]);
      wave_sum_store<T, D1>(gy, gsx + int64_t(s) * a.ld_gsh, true, false);
    }
  }
  T* qp = static_cast<T*>(a.q) + ((atom * a.num_layers + layer) * D1) * int64_t(u) + ch;
#pragma unroll
  for (int i = 0; i < D1; ++i) qp[int64_t(i) * u] = q[i];
}

// adjoint of the moments on the edges: d a[e,k] = sum_j Y[e,j] GM[j,k], d Y[e,j] += sum_k act(a[e,k]) GM[j,k]
template <typename T, int D>
__global__ __launch_bounds__(256) void tp_op_edge_env_kernel(TpOpArgs a) {
  const int lane = threadIdx.x & 63, blk = threadIdx.x >> 6;
  const int64_t atom = a.atom0 + blockIdx.x;
  const int beg = __builtin_amdgcn_readfirstlane(a.rowptr[atom]), end = __builtin_amdgcn_readfirstlane(a.rowptr[atom + 1]);
  if (beg >= end) return;
  const int ka = a.ka, k = blk * 64 + lane;
  const int64_t ED = int64_t(a.E) * a.ld_gsh;
  T gm[D];
  {
    const T* gp = static_cast<const T*>(a.gmbuf) + (atom * D) * int64_t(ka) + k;
#pragma unroll
    for (int j = 0; j < D; ++j) gm[j] = gp[int64_t(j) * ka];
  }
  const T* sh = static_cast<const T*>(a.sh);
  const T* av = static_cast<const T*>(a.a);
  T* ga = static_cast<T*>(a.g_a) + k;
  T* gse = static_cast<T*>(a.gsh_env) + int64_t(blk) * ED;
#pragma unroll 2
  for (int s = beg; s < end; ++s) {
    const T* y = sh + int64_t(s) * a.ld_sh;
    T x = av[int64_t(s) * a.ld_a + k];
    if (a.act) x = silu(x);
    T d = T(0), gy[D];
#pragma unroll
    for (int j = 0; j < D; ++j) {
      d += y[j] * gm[j];
      gy[j] = x * gm[j];
    }
    ga[int64_t(s) * a.ld_ga] = d;
    wave_sum_store<T, D>(gy, gse + int64_t(s) * a.ld_gsh, true, false);
  }
}

// The same adjoint on the fp64 matrix cores (v_mfma_f64_16x16x4): per center atom both sums are small matrix products over its
// edge segment, 16 edges per tile --
//     d Y[e, j] = sum_k act(a[e, k]) GM[j, k]      [16 x ka] @ [ka x D]   (D <= 16 columns, one accumulator per 64-wide k block:
//                                                                           edge_backward sums those slots, as with the vector form)
//     d a[e, k] = sum_j Y[e, j] GM[j, k]           [16 x D] @ [D x ka]    (ka / 16 column tiles)
// -- instead of D products + D 64-lane reductions per edge and wave (143 vector instructions per edge and 64-wide block, 68 % of
// the wave cycles waiting: 2.05 ms per layer at C5 for 4.1 GB, 1.8 TB/s).  One wave per atom; GM[n] (D x ka doubles, <= 16 KB)
// sits in LDS with padded rows, the operand rows come straight from HBM: with the k index permuted inside every 16-deep chunk
// (lane group g supplies k = 16 c + 4 g + s at MFMA step s) lane (i, g) reads 32 contiguous bytes of its edge's row per chunk.
// Operand layout of the instruction: A[i = lane & 15][k = lane >> 4], B[k = lane >> 4][j = lane & 15], C[i = 4 r + (lane >> 4)][j].
typedef double v4d_op __attribute__((ext_vector_type(4)));
typedef double v2d_op __attribute__((ext_vector_type(2)));
template <int D>
__global__ __launch_bounds__(256) void tp_op_edge_env_mfma_kernel(TpOpArgs a) {
  constexpr int KS2 = (D + 3) / 4;  // MFMA steps of the second product (K = D, zero-padded to a multiple of 4)
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int64_t atom = a.atom0 + int64_t(blockIdx.x) * 4 + wv;
  if (atom >= a.N) return;
  const int beg = __builtin_amdgcn_readfirstlane(a.rowptr[atom]), end = __builtin_amdgcn_readfirstlane(a.rowptr[atom + 1]);
  if (beg >= end) return;
  const int ka = a.ka, LD = ka + 2;  // (padded row: the 16 rows a B-operand read touches start 16 B apart)
  const int li = lane & 15, lg = lane >> 4;
  double* sGM = reinterpret_cast<double*>(aa_smem) + size_t(wv) * D * LD;  // wave-private [D][LD]
  {
    const double* gp = static_cast<const double*>(a.gmbuf) + (atom * D) * int64_t(ka);
    for (int idx = lane; idx < D * ka; idx += 64) sGM[(idx / ka) * LD + idx % ka] = gp[idx];
  }
  __builtin_amdgcn_wave_barrier();  // (wave-private region: LDS operations of one wave execute in order)
  const double* sh = static_cast<const double*>(a.sh);
  const double* av = static_cast<const double*>(a.a);
  double* ga = static_cast<double*>(a.g_a);
  const int64_t ED = int64_t(a.E) * a.ld_gsh;
  double* gse = static_cast<double*>(a.gsh_env);
  const int KC = ka / 16;  // 16-deep chunks of the first product (4 per 64-wide block)
  for (int s0 = beg; s0 < end; s0 += 16) {
    const int row = s0 + li < end ? s0 + li : end - 1;  // this lane's operand row (clamped; stores are masked)
    // ---- d Y = act(a) @ GM^T
    const double* ar = av + int64_t(row) * a.ld_a + 4 * lg;
    for (int blk = 0; blk * 4 < KC; ++blk) {
      v2d_op x[4][2];
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int h = 0; h < 2; ++h) x[c][h] = *reinterpret_cast<const v2d_op*>(ar + 16 * (4 * blk + c) + 2 * h);
      if (a.act) {
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
          for (int h = 0; h < 2; ++h) x[c][h] = v2d_op{silu(x[c][h][0]), silu(x[c][h][1])};
      }
      v4d_op acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const double* bp = sGM + li * LD + 16 * (4 * blk + c) + 4 * lg;  // GM[j = li][16 c' + 4 g + s]
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const double b = li < D ? bp[s] : 0.0;
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64(x[c][s >> 1][s & 1], b, acc, 0, 0, 0);
        }
      }
      if (li < D) {
        double* dst = gse + int64_t(blk) * ED + li;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int e = s0 + 4 * r + lg;
          if (e < end) dst[int64_t(e) * a.ld_gsh] = acc[r];
        }
      }
    }
    // ---- d a = Y @ GM
    double y[KS2];
#pragma unroll
    for (int t = 0; t < KS2; ++t) y[t] = 4 * t + lg < D ? sh[int64_t(row) * a.ld_sh + 4 * t + lg] : 0.0;
    for (int tile = 0; tile < KC; ++tile) {
      v4d_op acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int t = 0; t < KS2; ++t) {
        const double b = 4 * t + lg < D ? sGM[(4 * t + lg) * LD + 16 * tile + li] : 0.0;  // GM[j = 4 t + g][k = 16 tile + li]
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(y[t], b, acc, 0, 0, 0);
      }
      double* dst = ga + 16 * tile + li;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int e = s0 + 4 * r + lg;
        if (e < end) dst[int64_t(e) * a.ld_ga] = acc[r];
      }
    }
  }
}

// ---- split form, register-heavy per-atom kernels of the reverse pass
template <class Ch, typename T>
__global__ __launch_bounds__(256) void tp_op_bvecs_kernel(TpOpArgs a) {
  constexpr int L = Ch::L, D1 = SigAt<Ch, 0>::type::D1;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int64_t atom = a.atom0 + blockIdx.x;
  const int u = a.u, ch = wv * 64 + lane;
  const int beg = __builtin_amdgcn_readfirstlane(a.rowptr[atom]), end = __builtin_amdgcn_readfirstlane(a.rowptr[atom + 1]);
  if (beg >= end) return;
  T bm[L][D1];
  all_atom_vectors<Ch, 0, T, D1>(a, atom, ch, bm);
  T* bp = static_cast<T*>(a.bvec) + (atom * L * D1) * int64_t(u) + ch;
#pragma unroll
  for (int m = 0; m < L; ++m)
#pragma unroll
    for (int i = 0; i < D1; ++i) bp[int64_t(m * D1 + i) * u] = bm[m][i];
}

// d x2s_LI from Q_m (m >= LI, all in HBM) and GM = f * d x2s_LI . Wenv^T  ->  gmbuf
template <class Ch, int LI, typename T>
__global__ __launch_bounds__(256) void tp_op_bwd_mid_kernel(TpOpArgs a) {
  typedef typename SigAt<Ch, LI>::type S;
  constexpr int D = S::D2, R = S::LMAX + 1, L = Ch::L;
  constexpr int D1 = SigAt<Ch, 0>::type::D1;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nsl = blockDim.x >> 6;
  const int64_t atom = a.atom0 + blockIdx.x;
  const int u = a.u, ch = wv * 64 + lane;
  const int beg = __builtin_amdgcn_readfirstlane(a.rowptr[atom]), end = __builtin_amdgcn_readfirstlane(a.rowptr[atom + 1]);
  if (beg >= end) return;
  T* sG = reinterpret_cast<T*>(aa_smem);  // [nsl][D][64]
  T q[D1];
  {
    const T* qp = static_cast<const T*>(a.q) + ((atom * L + LI) * D1) * int64_t(u) + ch;
#pragma unroll
    for (int i = 0; i < D1; ++i) q[i] = qp[int64_t(i) * u];
  }
  T g2[D];
#pragma unroll
  for (int j = 0; j < D; ++j) g2[j] = T(0);
  x2s_grad_terms<Ch, LI, LI, T>(a, atom, ch, q, g2);
  if (a.proj_gemm) {  // (the batched projection launch turns d x2s into GM; f rides in its weights)
    T* dp = static_cast<T*>(a.dx2s) + atom * D * int64_t(u) + ch;
#pragma unroll
    for (int j = 0; j < D; ++j) dp[int64_t(j) * u] = g2[j];
    return;
  }
  {
    const T sf = T(a.sf);
#pragma unroll
    for (int j = 0; j < D; ++j) sG[(wv * D + j) * 64 + lane] = g2[j] * sf;
  }
  __syncthreads();
  const T* Wt = static_cast<const T*>(a.wt);  // [R][u][ka]
  const int ka = a.ka;
  for (int blk = wv; blk * 64 < ka; blk += nsl) {
    const int k = blk * 64 + lane;
    T gm[D];
#pragma unroll
    for (int j = 0; j < D; ++j) gm[j] = T(0);
#pragma unroll 4
    for (int c = 0; c < u; ++c) {
      T w3[R];
#pragma unroll
      for (int r = 0; r < R; ++r) w3[r] = Wt[(int64_t(r) * u + c) * ka + k];
      const T* sg = sG + ((c >> 6) * D) * 64 + (c & 63);
#pragma unroll
      for (int j = 0; j < D; ++j) gm[j] += sg[j * 64] * w3[r_of<0>(j)];
    }
    T* gp = static_cast<T*>(a.gmbuf) + (atom * D) * int64_t(ka) + k;
#pragma unroll
    for (int j = 0; j < D; ++j) gp[int64_t(j) * ka] = gm[j];
  }
}

// ---- reverse of layer LI
template <class Ch, int LI, typename T>
__global__ __launch_bounds__(256) void tp_op_bwd_kernel(TpOpArgs a) {
  typedef typename SigAt<Ch, LI>::type S;
  constexpr int D = S::D2, R = S::LMAX + 1, L = Ch::L;
  constexpr int D1 = SigAt<Ch, 0>::type::D1;
  static_assert(D1 == D && D <= 16, "x1 and the harmonics share their irreps");
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nsl = blockDim.x >> 6;
  const int64_t atom = a.atom0 + blockIdx.x;
  const int u = a.u, ch = wv * 64 + lane;
  const int beg = __builtin_amdgcn_readfirstlane(a.rowptr[atom]), end = __builtin_amdgcn_readfirstlane(a.rowptr[atom + 1]);
  if (beg >= end) return;  // (uniform for the whole workgroup: no barrier is skipped by a subset)
  T* sG = reinterpret_cast<T*>(aa_smem);  // [nsl][D][64]  d x2s_LI of every slice
  const T* sh = static_cast<const T*>(a.sh);
  const T* w0 = static_cast<const T*>(a.w0) + ch;
  const int64_t ED = int64_t(a.E) * a.ld_gsh;

  // per-atom vectors B_m (all layers) for the edge gradients of the first layer's reverse
  T bm[LI == 0 ? L : 1][D1];
  if constexpr (LI == 0) all_atom_vectors<Ch, 0, T, D1>(a, atom, ch, bm);

  // edge loop A: Q_LI = sum_e G_LI[e] x1[e]; for LI == 0 also d w0[e] and d Y[e] through x1
  T q[D1];
#pragma unroll
  for (int i = 0; i < D1; ++i) q[i] = T(0);
  {
    const T* gl = static_cast<const T*>(a.gscal[LI]) + ch;
    T* gw0 = static_cast<T*>(a.g_w0) + ch;
    T* gsx = static_cast<T*>(a.gsh_x1) + int64_t(wv) * ED;  // one slot per slice
    for (int s = beg; s < end; ++s) {
      const T* y = sh + int64_t(s) * a.ld_sh;
      T wr[R];
#pragma unroll
      for (int r = 0; r < R; ++r) wr[r] = w0[int64_t(s) * a.ld_w0 + r * u];
      const T g = gl[int64_t(s) * a.ld_gscal];
#pragma unroll
      for (int i = 0; i < D1; ++i) q[i] += g * (y[i] * wr[r_of<0>(i)]);
      if constexpr (LI == 0) {
        T gm_[L];
        gm_[0] = g;
#pragma unroll
        for (int m = 1; m < L; ++m) gm_[m] = static_cast<const T*>(a.gscal[m])[int64_t(s) * a.ld_gscal + ch];
        T gw[R], gy[D1];
#pragma unroll
        for (int r = 0; r < R; ++r) gw[r] = T(0);
#pragma unroll
        for (int i = 0; i < D1; ++i) {
          T gx = T(0);
#pragma unroll
          for (int m = 0; m < L; ++m) gx += gm_[m] * bm[m][i];
          gw[r_of<0>(i)] += gx * y[i];
          gy[i] = gx * wr[r_of<0>(i)];
        }
#pragma unroll
        for (int r = 0; r < R; ++r) gw0[int64_t(s) * a.ld_gw0 + r * u] = gw[r];
        wave_sum_store<T, D1>(gy, gsx + int64_t(s) * a.ld_gsh, true, false);
      }
    }
  }
  {
    T* qp = static_cast<T*>(a.q) + ((atom * L + LI) * D1) * int64_t(u) + ch;
#pragma unroll
    for (int i = 0; i < D1; ++i) qp[int64_t(i) * u] = q[i];
  }
  // d x2s_LI from all layers m >= LI
  T g2[D];
#pragma unroll
  for (int j = 0; j < D; ++j) g2[j] = T(0);
  x2s_grad_terms<Ch, LI, LI, T>(a, atom, ch, q, g2);
  {
    const T sf = T(a.sf);
#pragma unroll
    for (int j = 0; j < D; ++j) sG[(wv * D + j) * 64 + lane] = g2[j] * sf;
  }
  __syncthreads();
  // adjoint of the moments: k-block b (64 env inputs) belongs to wave b % nsl
  const T* Wt = static_cast<const T*>(a.wt);  // [R][u][ka]
  const T* av = static_cast<const T*>(a.a);
  const int ka = a.ka;
  for (int blk = wv; blk * 64 < ka; blk += nsl) {
    const int k = blk * 64 + lane;
    T gm[D];
#pragma unroll
    for (int j = 0; j < D; ++j) gm[j] = T(0);
    for (int c = 0; c < u; ++c) {
      T w3[R];
#pragma unroll
      for (int r = 0; r < R; ++r) w3[r] = Wt[(int64_t(r) * u + c) * ka + k];
      const T* sg = sG + ((c >> 6) * D) * 64 + (c & 63);
#pragma unroll
      for (int j = 0; j < D; ++j) gm[j] += sg[j * 64] * w3[r_of<0>(j)];
    }
    T* ga = static_cast<T*>(a.g_a) + k;
    T* gse = static_cast<T*>(a.gsh_env) + int64_t(blk) * ED;  // one slot per k-block
    for (int s = beg; s < end; ++s) {
      const T* y = sh + int64_t(s) * a.ld_sh;
      T x = av[int64_t(s) * a.ld_a + k];
      if (a.act) x = silu(x);
      T d = T(0), gy[D];
#pragma unroll
      for (int j = 0; j < D; ++j) {
        d += y[j] * gm[j];
        gy[j] = x * gm[j];
      }
      ga[int64_t(s) * a.ld_ga] = d;
      wave_sum_store<T, D>(gy, gse + int64_t(s) * a.ld_gsh, true, false);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// (first-layer, ..., last-layer) signatures of the standard stacks, by (l_max, L)
typedef SigChain<cg::Sig1, cg::Sig0, NoSig, 2> Chain12;
typedef SigChain<cg::Sig5, cg::Sig4, NoSig, 2> Chain22;
typedef SigChain<cg::Sig9, cg::Sig8, NoSig, 2> Chain32;
typedef SigChain<cg::Sig2, cg::Sig3, cg::Sig0, 3> Chain13;
typedef SigChain<cg::Sig6, cg::Sig7, cg::Sig4, 3> Chain23;
typedef SigChain<cg::Sig10, cg::Sig11, cg::Sig8, 3> Chain33;

int find_op_chain(const int* sigs, int L) {
  static const int table[6][4] = {{2, 1, 0, -1}, {2, 5, 4, -1}, {2, 9, 8, -1}, {3, 2, 3, 0}, {3, 6, 7, 4}, {3, 10, 11, 8}};
  for (int c = 0; c < 6; ++c) {
    if (table[c][0] != L) continue;
    bool ok = true;
    for (int l = 0; l < L; ++l) ok = ok && sigs[l] == table[c][1 + l];
    if (ok) return c;
  }
  return -1;
}

template <typename T>
int launch_tp_op(int chain, int layer, bool reverse, const TpOpArgs& a, hipStream_t stream, int phase) {
  if (a.N <= a.atom0) return AA_OK;
  if ((a.proj_gemm != 0) != (phase != 0) || (a.proj_gemm && !(a.bvec && a.mbuf && (!reverse || (a.gmbuf && a.dx2s)))))
    return fail(AA_ERR_INVALID, "tp_op: the projection-by-GEMM form runs in two phases of the split form");
  if ((a.u & 63) || a.u > 256 || (a.ka & 63) || a.ka > kOpMaxKa || a.ka_lds < a.ka)
    return fail(AA_ERR_INVALID, "tp_op: needs u = 64..256 in steps of 64 and env widths of 64 or 128");
  const int nsl = a.u / 64;
  const int Dsh = chain % 3 == 0 ? 4 : (chain % 3 == 1 ? 9 : 16);
  const size_t smem = reverse ? sizeof(T) * size_t(nsl) * Dsh * 64 : sizeof(T) * size_t(nsl) * Dsh * a.ka_lds;
  dim3 grid((unsigned)(a.N - a.atom0)), block(64 * nsl);
  const bool split = a.bvec != nullptr && (!reverse || a.gmbuf != nullptr);
  // split form: [per-atom heavy] -> HBM -> [lean edge streams]; see the kernels above
#define AA_OP_EDGE(DD, RR, LL)                                                                                  \
  {                                                                                                             \
    if (!reverse) {                                                                                             \
      hipLaunchKernelGGL((tp_op_edge_fwd_kernel<T, DD, RR>), grid, block, 0, stream, a, layer);                 \
    } else {                                                                                                    \
      TpOpArgs e = a;                                                                                           \
      if (layer == 0) {                                                                                         \
        hipLaunchKernelGGL((tp_op_edge_bwd_kernel<T, DD, RR, LL, true>), grid, block, 0, stream, e, layer);     \
      } else {                                                                                                  \
        e.gscal[1] = a.gscal[layer];                                                                            \
        hipLaunchKernelGGL((tp_op_edge_bwd_kernel<T, DD, RR, LL, false>), grid, block, 0, stream, e, layer);    \
      }                                                                                                         \
    }                                                                                                           \
  }
#define AA_OP_ENV(DD)                                                                                                           \
  if (sizeof(T) == 8 && a.env_mfma) {                                                                                           \
    const size_t smem_env = sizeof(double) * 4 * DD * (a.ka + 2);                                                               \
    if (smem_env > 48 * 1024)                                                                                                   \
      AA_CHECK_HIP(hipFuncSetAttribute((const void*)tp_op_edge_env_mfma_kernel<DD>, hipFuncAttributeMaxDynamicSharedMemorySize, int(smem_env))); \
    hipLaunchKernelGGL((tp_op_edge_env_mfma_kernel<DD>), dim3((unsigned)((a.N - a.atom0 + 3) / 4)), dim3(256), smem_env, stream, a); \
  } else {                                                                                                                      \
    hipLaunchKernelGGL((tp_op_edge_env_kernel<T, DD>), grid, dim3(a.ka), 0, stream, a);                                         \
  }
#define AA_OP_LAUNCH(CH, LI)                                                                          \
  {                                                                                                   \
    constexpr int DD_ = SigAt<CH, 0>::type::D1, RR_ = SigAt<CH, 0>::type::LMAX + 1, LL_ = CH::L;      \
    if (!split) {                                                                                     \
      if (reverse)                                                                                    \
        hipLaunchKernelGGL((tp_op_bwd_kernel<CH, LI, T>), grid, block, smem, stream, a);              \
      else                                                                                            \
        hipLaunchKernelGGL((tp_op_fwd_kernel<CH, LI, T>), grid, block, smem, stream, a);              \
    } else if (!reverse) {                                                                            \
      if (a.mbuf && phase != 2) hipLaunchKernelGGL((tp_op_moments_kernel<T, DD_>), grid, dim3(a.ka), 0, stream, a); \
      if (phase != 1) {                                                                               \
        hipLaunchKernelGGL((tp_op_fwd_kernel<CH, LI, T>), grid, block, smem, stream, a);              \
        AA_OP_EDGE(DD_, RR_, LL_)                                                                     \
      }                                                                                               \
    } else {                                                                                          \
      if (phase != 2) {                                                                               \
        if (LI == 0 && !a.bvec_ready) hipLaunchKernelGGL((tp_op_bvecs_kernel<CH, T>), grid, block, 0, stream, a); \
        AA_OP_EDGE(DD_, RR_, LL_)                                                                     \
        hipLaunchKernelGGL((tp_op_bwd_mid_kernel<CH, LI, T>), grid, block, smem, stream, a);          \
      }                                                                                               \
      if (phase != 1) { AA_OP_ENV(DD_) }                                                              \
    }                                                                                                 \
  }
#define AA_OP_CHAIN2(CH)                            \
  if (layer == 0) AA_OP_LAUNCH(CH, 0) else if (layer == 1) AA_OP_LAUNCH(CH, 1) else return fail(AA_ERR_INVALID, "tp_op: bad layer");
#define AA_OP_CHAIN3(CH)                            \
  if (layer == 0) AA_OP_LAUNCH(CH, 0) else if (layer == 1) AA_OP_LAUNCH(CH, 1) else if (layer == 2) AA_OP_LAUNCH(CH, 2) else return fail(AA_ERR_INVALID, "tp_op: bad layer");
  switch (chain) {
    case 0: AA_OP_CHAIN2(Chain12) break;
    case 1: AA_OP_CHAIN2(Chain22) break;
    case 2: AA_OP_CHAIN2(Chain32) break;
    case 3: AA_OP_CHAIN3(Chain13) break;
    case 4: AA_OP_CHAIN3(Chain23) break;
    case 5: AA_OP_CHAIN3(Chain33) break;
    default: return fail(AA_ERR_INVALID, "tp_op: unknown signature chain");
  }
#undef AA_OP_LAUNCH
#undef AA_OP_EDGE
#undef AA_OP_ENV
#undef AA_OP_CHAIN2
#undef AA_OP_CHAIN3
  AA_CHECK_HIP(hipGetLastError());
  return AA_OK;
}

template int launch_tp_op<float>(int, int, bool, const TpOpArgs&, hipStream_t, int);
template int launch_tp_op<double>(int, int, bool, const TpOpArgs&, hipStream_t, int);

}  // namespace aa
