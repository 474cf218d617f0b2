// Specialised tensor-product layer kernels (gfx950): compile-time Clebsch-Gordan tables
// (aa_cg_gen.h), operands in registers, lane = channel, one wave per center atom (x u/64).
//
// Same math as aa_tp.hip (reference: allegro/nn/_strided/_contract.py:185-251 fused with the env
// weighting of allegro/nn/_strided/_channels.py:44-57), selected by the model pipeline when the
// layer's w3j buffer equals a generated signature and u is 16/32/64/128/256.  Internal layouts are
// channel-minor so every global access of a wave is one contiguous 256-B row:
//   env / x1 weights  [E][R][u]     tensor features [E][d][u]     x2s [N][D][u]
// A wave walks its atom's edge segment twice (segment sum, then contraction) with everything in
// VGPRs: no LDS, no barriers, no atomics in the forward; deterministic summation order.
#include "aa_cg_gen.h"
#include "aa_wave.h"
#include "aa_common.h"

namespace aa {

namespace {

struct LaneMap {
  int64_t atom;
  int ch;
  int width;    // lanes cooperating on one atom-slice (min(u,64))
  bool valid;
  bool leader;  // lane 0 of its group
};

__device__ __forceinline__ LaneMap lane_map(int u, int64_t N) {
  LaneMap m;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  if (u >= 64) {
    const int wpa = u >> 6;
    m.atom = int64_t(blockIdx.x) * (4 / wpa) + wave / wpa;
    m.ch = (wave % wpa) * 64 + lane;
    m.width = 64;
    m.leader = lane == 0;
  } else {
    const int apw = 64 / u;
    m.atom = (int64_t(blockIdx.x) * 4 + wave) * apw + lane / u;
    m.ch = lane % u;
    m.width = u;
    m.leader = (lane % u) == 0;
  }
  m.valid = m.atom < N;
  return m;
}

template <typename T>
__device__ __forceinline__ T group_sum(T v, int width) {
  for (int m = width >> 1; m >= 1; m >>= 1) v += __shfl_xor(v, m);
  return v;
}

__device__ __forceinline__ int wave_max(int v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) {
    int o = __shfl_xor(v, m);
    v = o > v ? o : v;
  }
  return v;
}

// wave-uniform value -> SGPR (so that addresses derived from it use the scalar memory path)
__device__ __forceinline__ int uniform_if(bool uni, int v) { return uni ? __builtin_amdgcn_readfirstlane(v) : v; }

}  // namespace

template <class Sig, typename T>
__global__ __launch_bounds__(256) void tp_spec_fwd_kernel(TpSpecFwdArgs a) {
  constexpr int D1 = Sig::D1, D2 = Sig::D2, DOUT = Sig::DOUT, P = Sig::P, R = Sig::LMAX + 1;
  const int u = a.u;
  const LaneMap m = lane_map(u, a.N);
  int beg = 0, end = 0;
  if (m.valid) {
    beg = a.rowptr[m.atom];
    end = a.rowptr[m.atom + 1];
  }
  const int ch = m.ch;
  const T* sh = static_cast<const T*>(a.sh);
  const T* wenv = static_cast<const T*>(a.w_env);
  // ---- phase 1: x2s[j] = f * sum_e sh[e,j] * w_env[e, r(j), ch]        (_contract.py:195-204)
  T x2s[D2];
#pragma unroll
  for (int j = 0; j < D2; ++j) x2s[j] = T(0);
  for (int s = beg; s < end; ++s) {
    const T* y = sh + int64_t(s) * a.ld_sh;
    const T* we = wenv + int64_t(s) * a.ld_we + ch;
    T wr[R];
#pragma unroll
    for (int r = 0; r < R; ++r) wr[r] = we[r * u];
#pragma unroll
    for (int j = 0; j < D2; ++j) x2s[j] += y[j] * wr[r_of<0>(j)];
  }
  const T sf = T(a.sf);
#pragma unroll
  for (int j = 0; j < D2; ++j) x2s[j] *= sf;
  if (m.valid) {
    T* xo = static_cast<T*>(a.x2s) + m.atom * D2 * int64_t(u) + ch;
#pragma unroll
    for (int j = 0; j < D2; ++j) xo[int64_t(j) * u] = x2s[j];
  }
  // ---- path weights of this channel
  T w[P];
  {
    const T* W = static_cast<const T*>(a.weights);
#pragma unroll
    for (int p = 0; p < P; ++p) w[p] = a.coupling ? W[ch * P + p] : W[p];
  }
  // ---- phase 2: contraction per edge                                     (_contract.py:213-251)
  for (int s = beg; s < end; ++s) {
    T x1[D1];
    if (a.x1_dense) {
      const T* xp = static_cast<const T*>(a.x1_dense) + int64_t(s) * D1 * u + ch;
#pragma unroll
      for (int i = 0; i < D1; ++i) x1[i] = xp[int64_t(i) * u];
    } else {
      const T* y = sh + int64_t(s) * a.ld_sh;
      const T* w1 = static_cast<const T*>(a.w_x1) + int64_t(s) * a.ld_w1 + ch;
      T wr[R];
#pragma unroll
      for (int r = 0; r < R; ++r) wr[r] = w1[r * u];
#pragma unroll
      for (int i = 0; i < D1; ++i) x1[i] = y[i] * wr[r_of<0>(i)];
    }
    T out[DOUT];
    Sig::template fwd<T>(x1, x2s, w, out);
    if (a.out) {
      T* op = static_cast<T*>(a.out) + int64_t(s) * DOUT * u + ch;
#pragma unroll
      for (int k = 0; k < DOUT; ++k) op[int64_t(k) * u] = out[k];
    }
    if (a.scal) static_cast<T*>(a.scal)[int64_t(s) * a.ld_scal + ch] = out[0];
  }
}

template <class Sig, typename T>
__global__ __launch_bounds__(256) void tp_spec_bwd_kernel(TpSpecBwdArgs a) {
  constexpr int D1 = Sig::D1, D2 = Sig::D2, DOUT = Sig::DOUT, P = Sig::P, R = Sig::LMAX + 1;
  const int u = a.u;
  const LaneMap m = lane_map(u, a.N);
  int beg = 0, end = 0;
  if (m.valid) {
    beg = a.rowptr[m.atom];
    end = a.rowptr[m.atom + 1];
  }
  const int deg = end - beg;
  const int maxdeg = u >= 64 ? deg : wave_max(deg);  // all lanes of a wave iterate together (shuffles below)
  const int ch = m.ch;
  const bool multi_wave = u > 64;
  const T* sh = static_cast<const T*>(a.sh);
  const T* wenv = static_cast<const T*>(a.w_env);
  T* gsh1 = static_cast<T*>(a.gsh_x1);
  T* gsh2 = static_cast<T*>(a.gsh_env);
  T x2s[D2], g2acc[D2];
  {
    const T* xi = static_cast<const T*>(a.x2s) + (m.valid ? m.atom : 0) * D2 * int64_t(u) + ch;
#pragma unroll
    for (int j = 0; j < D2; ++j) {
      x2s[j] = m.valid ? xi[int64_t(j) * u] : T(0);
      g2acc[j] = T(0);
    }
  }
  T w[P];
  {
    const T* W = static_cast<const T*>(a.weights);
#pragma unroll
    for (int p = 0; p < P; ++p) w[p] = a.coupling ? W[ch * P + p] : W[p];
  }
  // ---- pass 1: per edge, grad wrt x1 and accumulation of grad wrt x2s
  for (int it = 0; it < maxdeg; ++it) {
    const bool act = it < deg;
    const int64_t s = act ? beg + it : 0;
    const T* y = sh + s * a.ld_sh;
    T x1[D1], wr1[R];
    if (a.x1_dense) {
      const T* xp = static_cast<const T*>(a.x1_dense) + s * D1 * u + ch;
#pragma unroll
      for (int i = 0; i < D1; ++i) x1[i] = act ? xp[int64_t(i) * u] : T(0);
    } else {
      const T* w1 = static_cast<const T*>(a.w_x1) + s * a.ld_w1 + ch;
#pragma unroll
      for (int r = 0; r < R; ++r) wr1[r] = act ? w1[r * u] : T(0);
#pragma unroll
      for (int i = 0; i < D1; ++i) x1[i] = y[i] * wr1[r_of<0>(i)];
    }
    T go[DOUT];
#pragma unroll
    for (int k = 0; k < DOUT; ++k) go[k] = T(0);
    if (act) {
      if (a.gout) {
        const T* gp = static_cast<const T*>(a.gout) + s * DOUT * u + ch;
#pragma unroll
        for (int k = 0; k < DOUT; ++k) go[k] = gp[int64_t(k) * u];
      }
      if (a.gscal) go[0] += static_cast<const T*>(a.gscal)[s * a.ld_gscal + ch];
    }
    T g1[D1];
    Sig::template bx1<T>(go, x2s, w, g1);
    if (a.g_x1_dense) {
      if (act) {
        T* gp = static_cast<T*>(a.g_x1_dense) + s * D1 * u + ch;
#pragma unroll
        for (int i = 0; i < D1; ++i) gp[int64_t(i) * u] = g1[i];
      }
    } else {
      // adjoint of x1 = sh[e,i] * w1[e, r(i), ch]
      T gw[R];
#pragma unroll
      for (int r = 0; r < R; ++r) gw[r] = T(0);
#pragma unroll
      for (int i = 0; i < D1; ++i) gw[r_of<0>(i)] += g1[i] * y[i];
      if (act) {
        T* gwp = static_cast<T*>(a.g_w1) + s * a.ld_gw1 + ch;
#pragma unroll
        for (int r = 0; r < R; ++r) gwp[r * u] = gw[r];
      }
      if (m.width == 64 && D1 <= 16) {
        T gy[D1 <= 16 ? D1 : 1];
#pragma unroll
        for (int i = 0; i < (D1 <= 16 ? D1 : 1); ++i) gy[i] = act ? g1[i] * wr1[r_of<0>(i)] : T(0);
        wave_sum_store<T, (D1 <= 16 ? D1 : 1)>(gy, gsh1 + s * a.ld_gsh, act, multi_wave);
      } else {
#pragma unroll
        for (int i = 0; i < D1; ++i) {
          T v = group_sum<T>(act ? g1[i] * wr1[r_of<0>(i)] : T(0), m.width);
          if (m.leader && act) {
            if (multi_wave)
              atomicAdd(&gsh1[s * a.ld_gsh + i], v);
            else
              gsh1[s * a.ld_gsh + i] = v;
          }
        }
      }
    }
    T g2[D2];
    Sig::template bx2<T>(go, x1, w, g2);
#pragma unroll
    for (int j = 0; j < D2; ++j) g2acc[j] += g2[j];
  }
  // ---- pass 2: adjoint of (scale, segment-sum, gather) and of the env weighting
  const T sf = T(a.sf);
#pragma unroll
  for (int j = 0; j < D2; ++j) g2acc[j] *= sf;
  for (int it = 0; it < maxdeg; ++it) {
    const bool act = it < deg;
    const int64_t s = act ? beg + it : 0;
    const T* y = sh + s * a.ld_sh;
    const T* we = wenv + s * a.ld_we + ch;
    T wr[R], gw[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      wr[r] = act ? we[r * u] : T(0);
      gw[r] = T(0);
    }
#pragma unroll
    for (int j = 0; j < D2; ++j) gw[r_of<0>(j)] += y[j] * g2acc[j];
    if (act) {
      T* gwp = static_cast<T*>(a.g_wenv) + s * a.ld_gwe + ch;
#pragma unroll
      for (int r = 0; r < R; ++r) gwp[r * u] = gw[r];
    }
    if (m.width == 64) {
      T gy[D2];
#pragma unroll
      for (int j = 0; j < D2; ++j) gy[j] = act ? wr[r_of<0>(j)] * g2acc[j] : T(0);
      wave_sum_store<T, D2>(gy, gsh2 + s * a.ld_gsh, act, multi_wave);
    } else {
#pragma unroll
      for (int j = 0; j < D2; ++j) {
        T v = group_sum<T>(act ? wr[r_of<0>(j)] * g2acc[j] : T(0), m.width);
        if (m.leader && act) {
          if (multi_wave)
            atomicAdd(&gsh2[s * a.ld_gsh + j], v);
          else
            gsh2[s * a.ld_gsh + j] = v;
        }
      }
    }
  }
}

static unsigned spec_grid(int u, int64_t N) {
  int64_t atoms_per_block = u >= 64 ? 4 / (u / 64) : 4 * (64 / u);
  return (unsigned)((N + atoms_per_block - 1) / atoms_per_block);
}

// =============================================================================================
// chain kernels for 2-layer stacks (Sig0: layer 0 with implicit x1, Sig1: last layer, DOUT == 1)
// =============================================================================================
namespace {
template <class Sig0, typename T>
__device__ __forceinline__ void chain_tf1(const T* y, const T* w0p, int u, bool act, const T* x2s0, const T* wp0, T* x1,
                                          T* wr0, T* tf1) {
  constexpr int R = Sig0::LMAX + 1;
#pragma unroll
  for (int r = 0; r < R; ++r) wr0[r] = act ? w0p[r * u] : T(0);
#pragma unroll
  for (int i = 0; i < Sig0::D1; ++i) x1[i] = y[i] * wr0[r_of<0>(i)];
  Sig0::template fwd<T>(x1, x2s0, wp0, tf1);
}

template <typename T, int D>
__device__ __forceinline__ void reduce_store(const T* gy, T* dst, const LaneMap& m, bool act, bool multi_wave) {
  if (m.width == 64) {
    wave_sum_store<T, D>(gy, dst, act, multi_wave);
  } else {
#pragma unroll
    for (int j = 0; j < D; ++j) {
      T v = group_sum<T>(gy[j], m.width);
      if (m.leader && act) {
        if (multi_wave)
          atomicAdd(&dst[j], v);
        else
          dst[j] = v;
      }
    }
  }
}
}  // namespace

// per-edge operands fetched one edge ahead (software pipelining: the loads of edge s+1 are in flight
// while edge s is being contracted)
template <typename T, int D, int R>
struct EdgeIn {
  T y[D];
  T wa[R];
  T wb[R];
  T g0, g1;
};

template <class Sig0, class Sig1, typename T>
__global__ __launch_bounds__(256) void tp_chain_fwd_last_kernel(TpChainArgs a) {
  static_assert(Sig1::DOUT == 1 && Sig0::DOUT == Sig1::D1 && Sig0::D2 == Sig1::D2, "chain signature mismatch");
  constexpr int D = Sig0::D2, R = Sig0::LMAX + 1;
  const int u = a.u;
  const LaneMap m = lane_map(u, a.N);
  int beg = 0, end = 0;
  if (m.valid) {
    beg = a.rowptr[m.atom];
    end = a.rowptr[m.atom + 1];
  }
  const bool uni = u >= 64;
  beg = uniform_if(uni, beg);
  end = uniform_if(uni, end);
  const int ch = m.ch;
  const T* sh = static_cast<const T*>(a.sh);
  const T* w0g = static_cast<const T*>(a.w0) + ch;
  const T* we1 = static_cast<const T*>(a.wenv1) + ch;
  T x2s1[D], x2s0[D];
#pragma unroll
  for (int j = 0; j < D; ++j) x2s1[j] = T(0);
#pragma unroll 4
  for (int s = beg; s < end; ++s) {
    const T* y = sh + int64_t(s) * a.ld_sh;
    const T* we = we1 + int64_t(s) * a.ld_we1;
    T wr[R];
#pragma unroll
    for (int r = 0; r < R; ++r) wr[r] = we[r * u];
#pragma unroll
    for (int j = 0; j < D; ++j) x2s1[j] += y[j] * wr[r_of<0>(j)];
  }
  const T sf = T(a.sf);
  {
    const int64_t base = (m.valid ? m.atom : 0) * D * int64_t(u) + ch;
    const T* xi = static_cast<const T*>(a.x2s0) + base;
    T* xo = static_cast<T*>(a.x2s1) + base;
#pragma unroll
    for (int j = 0; j < D; ++j) {
      x2s1[j] *= sf;
      x2s0[j] = m.valid ? xi[int64_t(j) * u] : T(0);
      if (m.valid) xo[int64_t(j) * u] = x2s1[j];
    }
  }
  T wp0[Sig0::P], wp1[Sig1::P];
  {
    const T* W0 = static_cast<const T*>(a.weights0);
    const T* W1 = static_cast<const T*>(a.weights1);
#pragma unroll
    for (int p = 0; p < Sig0::P; ++p) wp0[p] = a.coupling ? W0[ch * Sig0::P + p] : W0[p];
#pragma unroll
    for (int p = 0; p < Sig1::P; ++p) wp1[p] = a.coupling ? W1[ch * Sig1::P + p] : W1[p];
  }
  auto fetch = [&](int s, EdgeIn<T, D, R>& in) {
    const T* y = sh + int64_t(s) * a.ld_sh;
    const T* w0p = w0g + int64_t(s) * a.ld_w0;
#pragma unroll
    for (int j = 0; j < D; ++j) in.y[j] = y[j];
#pragma unroll
    for (int r = 0; r < R; ++r) in.wa[r] = w0p[r * u];
  };
  if (beg < end) {
    EdgeIn<T, D, R> cur, nxt;
    fetch(beg, cur);
    for (int s = beg; s < end; ++s) {
      fetch(s + 1 < end ? s + 1 : s, nxt);
      T x1[Sig0::D1], tf1[Sig0::DOUT], out[1];
#pragma unroll
      for (int i = 0; i < Sig0::D1; ++i) x1[i] = cur.y[i] * cur.wa[r_of<0>(i)];
      Sig0::template fwd<T>(x1, x2s0, wp0, tf1);
      Sig1::template fwd<T>(tf1, x2s1, wp1, out);
      static_cast<T*>(a.scal1)[int64_t(s) * a.ld_scal + ch] = out[0];
      cur = nxt;
    }
  }
}

template <class Sig0, class Sig1, typename T>
__global__ __launch_bounds__(256) void tp_chain_bwd_last_kernel(TpChainArgs a) {
  constexpr int D = Sig0::D2, R = Sig0::LMAX + 1;
  const int u = a.u;
  const LaneMap m = lane_map(u, a.N);
  int beg = 0, end = 0;
  if (m.valid) {
    beg = a.rowptr[m.atom];
    end = a.rowptr[m.atom + 1];
  }
  const bool uni = u >= 64;
  beg = uniform_if(uni, beg);
  end = uniform_if(uni, end);
  const int deg = end - beg;
  const int maxdeg = uni ? deg : wave_max(deg);
  const int ch = m.ch;
  const bool multi_wave = u > 64;
  const T* sh = static_cast<const T*>(a.sh);
  const T* w0g = static_cast<const T*>(a.w0) + ch;
  const T* we1 = static_cast<const T*>(a.wenv1) + ch;
  const T* gs1 = static_cast<const T*>(a.gscal1) + ch;
  T x2s0[D], g2acc[D];
  {
    const T* xi = static_cast<const T*>(a.x2s0) + (m.valid ? m.atom : 0) * D * int64_t(u) + ch;
#pragma unroll
    for (int j = 0; j < D; ++j) {
      x2s0[j] = m.valid ? xi[int64_t(j) * u] : T(0);
      g2acc[j] = T(0);
    }
  }
  T wp0[Sig0::P], wp1[Sig1::P];
  {
    const T* W0 = static_cast<const T*>(a.weights0);
    const T* W1 = static_cast<const T*>(a.weights1);
#pragma unroll
    for (int p = 0; p < Sig0::P; ++p) wp0[p] = a.coupling ? W0[ch * Sig0::P + p] : W0[p];
#pragma unroll
    for (int p = 0; p < Sig1::P; ++p) wp1[p] = a.coupling ? W1[ch * Sig1::P + p] : W1[p];
  }
  // pass 1: g2acc[j] = sum_e bx2_1(d_scal1[e], tf1[e])
  {
    auto fetch = [&](int s, EdgeIn<T, D, R>& in) {
      const T* y = sh + int64_t(s) * a.ld_sh;
      const T* w0p = w0g + int64_t(s) * a.ld_w0;
#pragma unroll
      for (int j = 0; j < D; ++j) in.y[j] = y[j];
#pragma unroll
      for (int r = 0; r < R; ++r) in.wa[r] = w0p[r * u];
      in.g1 = gs1[int64_t(s) * a.ld_gscal];
    };
    if (beg < end) {
      EdgeIn<T, D, R> cur, nxt;
      fetch(beg, cur);
      for (int s = beg; s < end; ++s) {
        fetch(s + 1 < end ? s + 1 : s, nxt);
        T x1[Sig0::D1], tf1[Sig0::DOUT], go[1], g2[D];
#pragma unroll
        for (int i = 0; i < Sig0::D1; ++i) x1[i] = cur.y[i] * cur.wa[r_of<0>(i)];
        Sig0::template fwd<T>(x1, x2s0, wp0, tf1);
        go[0] = cur.g1;
        Sig1::template bx2<T>(go, tf1, wp1, g2);
#pragma unroll
        for (int j = 0; j < D; ++j) g2acc[j] += g2[j];
        cur = nxt;
      }
    }
  }
  const T sf = T(a.sf);
#pragma unroll
  for (int j = 0; j < D; ++j) g2acc[j] *= sf;
  // pass 2: adjoint of (scale, segment-sum, gather) and of the env weighting
  for (int it = 0; it < maxdeg; ++it) {
    const bool act = it < deg;
    const int64_t s = act ? beg + it : 0;
    const T* y = sh + s * a.ld_sh;
    const T* we = we1 + s * a.ld_we1;
    T wr[R], gw[R], gy[D];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      wr[r] = act ? we[r * u] : T(0);
      gw[r] = T(0);
    }
#pragma unroll
    for (int j = 0; j < D; ++j) {
      gw[r_of<0>(j)] += y[j] * g2acc[j];
      gy[j] = act ? wr[r_of<0>(j)] * g2acc[j] : T(0);
    }
    if (act) {
      T* gwp = static_cast<T*>(a.g_wenv) + s * a.ld_gwe + ch;
#pragma unroll
      for (int r = 0; r < R; ++r) gwp[r * u] = gw[r];
    }
    reduce_store<T, D>(gy, static_cast<T*>(a.gsh_env) + s * a.ld_gsh, m, act, multi_wave);
  }
}

template <class Sig0, class Sig1, typename T>
__global__ __launch_bounds__(256) void tp_chain_bwd_first_kernel(TpChainArgs a) {
  constexpr int D = Sig0::D2, D1 = Sig0::D1, DOUT = Sig0::DOUT, R = Sig0::LMAX + 1;
  const int u = a.u;
  const LaneMap m = lane_map(u, a.N);
  int beg = 0, end = 0;
  if (m.valid) {
    beg = a.rowptr[m.atom];
    end = a.rowptr[m.atom + 1];
  }
  const bool uni = u >= 64;
  beg = uniform_if(uni, beg);
  end = uniform_if(uni, end);
  const int deg = end - beg;
  const int maxdeg = uni ? deg : wave_max(deg);
  const int ch = m.ch;
  const bool multi_wave = u > 64;
  const T* sh = static_cast<const T*>(a.sh);
  const T* w0g = static_cast<const T*>(a.w0) + ch;
  const T* gs0 = static_cast<const T*>(a.gscal0) + ch;
  const T* gs1 = static_cast<const T*>(a.gscal1) + ch;
  T x2s0[D], x2s1[D], g2acc[D];
  {
    const int64_t base = (m.valid ? m.atom : 0) * D * int64_t(u) + ch;
    const T* x0 = static_cast<const T*>(a.x2s0) + base;
    const T* x1p = static_cast<const T*>(a.x2s1) + base;
#pragma unroll
    for (int j = 0; j < D; ++j) {
      x2s0[j] = m.valid ? x0[int64_t(j) * u] : T(0);
      x2s1[j] = m.valid ? x1p[int64_t(j) * u] : T(0);
      g2acc[j] = T(0);
    }
  }
  T wp0[Sig0::P], wp1[Sig1::P];
  {
    const T* W0 = static_cast<const T*>(a.weights0);
    const T* W1 = static_cast<const T*>(a.weights1);
#pragma unroll
    for (int p = 0; p < Sig0::P; ++p) wp0[p] = a.coupling ? W0[ch * Sig0::P + p] : W0[p];
#pragma unroll
    for (int p = 0; p < Sig1::P; ++p) wp1[p] = a.coupling ? W1[ch * Sig1::P + p] : W1[p];
  }
  {
    auto fetch = [&](int64_t s, bool act, EdgeIn<T, D, R>& in) {
      const T* y = sh + s * a.ld_sh;
      const T* w0p = w0g + s * a.ld_w0;
#pragma unroll
      for (int j = 0; j < D; ++j) in.y[j] = y[j];
#pragma unroll
      for (int r = 0; r < R; ++r) in.wa[r] = act ? w0p[r * u] : T(0);
      in.g0 = act ? gs0[s * a.ld_gscal] : T(0);
      in.g1 = act ? gs1[s * a.ld_gscal] : T(0);
    };
    EdgeIn<T, D, R> cur, nxt;
    fetch(0 < deg ? beg : 0, 0 < deg, cur);
    for (int it = 0; it < maxdeg; ++it) {
      const bool act = it < deg;
      const int64_t s = act ? beg + it : 0;
      const bool actn = it + 1 < deg;
      fetch(actn ? beg + it + 1 : 0, actn, nxt);
      T x1[D1];
#pragma unroll
      for (int i = 0; i < D1; ++i) x1[i] = cur.y[i] * cur.wa[r_of<0>(i)];
      // d_tf1 = bx1 of the last layer, recomputed from d_scal1 and x2s1 (never stored)
      T gn[1], go[DOUT];
      gn[0] = cur.g1;
      Sig1::template bx1<T>(gn, x2s1, wp1, go);
      go[0] += cur.g0;
      T g1[D1], g2[D];
      Sig0::template bx1<T>(go, x2s0, wp0, g1);
      Sig0::template bx2<T>(go, x1, wp0, g2);
#pragma unroll
      for (int j = 0; j < D; ++j) g2acc[j] += g2[j];
      T gw[R], gy[D1];
#pragma unroll
      for (int r = 0; r < R; ++r) gw[r] = T(0);
#pragma unroll
      for (int i = 0; i < D1; ++i) {
        gw[r_of<0>(i)] += g1[i] * cur.y[i];
        gy[i] = act ? g1[i] * cur.wa[r_of<0>(i)] : T(0);
      }
      if (act) {
        T* gwp = static_cast<T*>(a.g_w0) + s * a.ld_gw0 + ch;
#pragma unroll
        for (int r = 0; r < R; ++r) gwp[r * u] = gw[r];
      }
      reduce_store<T, D1>(gy, static_cast<T*>(a.gsh_x1) + s * a.ld_gsh, m, act, multi_wave);
      cur = nxt;
    }
  }
  const T sf = T(a.sf);
#pragma unroll
  for (int j = 0; j < D; ++j) g2acc[j] *= sf;
  const T* we0 = static_cast<const T*>(a.wenv0) + ch;
  for (int it = 0; it < maxdeg; ++it) {
    const bool act = it < deg;
    const int64_t s = act ? beg + it : 0;
    const T* y = sh + s * a.ld_sh;
    const T* we = we0 + s * a.ld_we0;
    T wr[R], gw[R], gy[D];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      wr[r] = act ? we[r * u] : T(0);
      gw[r] = T(0);
    }
#pragma unroll
    for (int j = 0; j < D; ++j) {
      gw[r_of<0>(j)] += y[j] * g2acc[j];
      gy[j] = act ? wr[r_of<0>(j)] * g2acc[j] : T(0);
    }
    if (act) {
      T* gwp = static_cast<T*>(a.g_wenv) + s * a.ld_gwe + ch;
#pragma unroll
      for (int r = 0; r < R; ++r) gwp[r * u] = gw[r];
    }
    reduce_store<T, D>(gy, static_cast<T*>(a.gsh_env) + s * a.ld_gsh, m, act, multi_wave);
  }
}

// =============================================================================================
// moments kernels (u == 64, one wave per atom): see TpMomArgs in aa_common.h
//
// Every loop over the atom's edge segment is software-pipelined by hand: the rows of the NEXT batch / edge
// pair are requested before the current one is consumed, so a wave always has several KB in flight instead of
// paying one memory latency per edge (these kernels are latency-, not bandwidth- or VALU-bound).  The spherical
// harmonics of the segment are staged once in wave-private LDS, pair-interleaved, so that the packed two-edge
// math reads them as 2-vectors.
// =============================================================================================
namespace {
constexpr int kMaxKa = 128;
constexpr int kSegCap = 64;  // edges of a segment staged per pass (longer segments are walked in chunks of this size)
constexpr int kPB = 8;       // edge pairs per load batch in the moment loops

// stage sh[cb..ce) as sY[(pair*D + j)*2 + half]; an odd tail repeats the last edge, and pairs up to the next
// multiple of kPB are zero-filled (so batch loops need no bounds checks on the harmonics)
template <typename T, int D>
__device__ __forceinline__ void stage_sh(const T* sh, int ld_sh, int cb, int ce, int lane, T* sY) {
  const int n = ce - cb;
  const int npad = (n + 1) & ~1;
  const int nfill = ((((n + 1) >> 1) + kPB - 1) / kPB) * kPB * 2;
  __builtin_amdgcn_wave_barrier();
  for (int i = lane; i < nfill * D; i += 64) {
    const int e = i / D, j = i - e * D;
    const int es = e < n ? e : n - 1;
    const T v = sh[int64_t(cb + es) * ld_sh + j];
    sY[((e >> 1) * D + j) * 2 + (e & 1)] = e < npad ? v : T(0);
  }
  __builtin_amdgcn_wave_barrier();
}

// per-pair env-input values for channel block kb: {a[s][kb+lane], a[s+1][kb+lane]}, zero beyond the segment
template <typename T>
__device__ __forceinline__ void load_a_batch(const T* a, int ld_a, int kb, bool act, int s0, int ce, int lane,
                                             typename Pk<T>::type* v) {
  typedef typename Pk<T>::type T2;
#pragma unroll
  for (int i = 0; i < kPB; ++i) {
    const int e0 = s0 + 2 * i, e1 = e0 + 1;
    const int c0 = e0 < ce ? e0 : ce - 1, c1 = e1 < ce ? e1 : ce - 1;
    T x0 = ld_stream(a + int64_t(c0) * ld_a + kb + lane), x1 = ld_stream(a + int64_t(c1) * ld_a + kb + lane);
    if (act) {
      x0 = silu(x0);
      x1 = silu(x1);
    }
    v[i] = T2{e0 < ce ? x0 : T(0), e1 < ce ? x1 : T(0)};
  }
}

// sM[j][kb + lane] (+)= sum over the staged chunk of sh[e,j] * act(a[e, kb + lane])
template <typename T, int D>
__device__ __forceinline__ void mom_accumulate(const T* a, int ld_a, int ka, int ka_lds, bool act, int cb, int ce, bool first,
                                               int lane, const T* sY, T* sM) {
  typedef typename Pk<T>::type T2;
  const T2* sY2 = reinterpret_cast<const T2*>(sY);
  const int np = (ce - cb + 1) >> 1;
  for (int kb = 0; kb < ka; kb += 64) {
    T2 m2[D];
#pragma unroll
    for (int j = 0; j < D; ++j) m2[j] = T2{T(0), T(0)};
    T2 cur[kPB], nxt[kPB];
    load_a_batch<T>(a, ld_a, kb, act, cb, ce, lane, cur);
    for (int q0 = 0; q0 < np; q0 += kPB) {
      load_a_batch<T>(a, ld_a, kb, act, cb + 2 * (q0 + kPB), ce, lane, nxt);
#pragma unroll
      for (int i = 0; i < kPB; ++i) {
#pragma unroll
        for (int j = 0; j < D; ++j) m2[j] += sY2[(q0 + i) * D + j] * cur[i];
      }
#pragma unroll
      for (int i = 0; i < kPB; ++i) cur[i] = nxt[i];
    }
#pragma unroll
    for (int j = 0; j < D; ++j) {
      const T v = m2[j][0] + m2[j][1];
      T* d = sM + j * ka_lds + kb + lane;
      *d = first ? v : *d + v;
    }
  }
}

// x2s[j] (this lane's channel) = f * sum_k M[j][k] * Wk[k][r(j)][ch]; weight rows are fetched 8 k ahead
template <typename T, int D, int R>
__device__ __forceinline__ void mom_project(const T* sM, int ka, int ka_lds, const T* Wk, T sf, int lane, T* x2s) {
  constexpr int KB = 8;
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int j = 0; j < D; ++j) x2s[j] = T(0);
  T wc[KB][R], wn[KB][R];
  auto loadw = [&](int k0, T(*w)[R]) {
#pragma unroll
    for (int i = 0; i < KB; ++i) {
      const int k = k0 + i < ka ? k0 + i : ka - 1;
#pragma unroll
      for (int r = 0; r < R; ++r) w[i][r] = Wk[(int64_t(k) * R + r) * 64 + lane];
    }
  };
  loadw(0, wc);
  for (int k0 = 0; k0 < ka; k0 += KB) {
    loadw(k0 + KB, wn);
#pragma unroll
    for (int i = 0; i < KB; ++i) {
#pragma unroll
      for (int j = 0; j < D; ++j) x2s[j] += sM[j * ka_lds + k0 + i] * wc[i][r_of<0>(j)];
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

constexpr int kGmChannels = 64;  // channels the GM projection contracts (all of them; the round-4 pricing experiment that
                                 // contracted 8 -- wrong results by design -- is described in HISTORY.md, not kept in this file)
// plain form of mom_gm below (sG as [j][ch], weight rows copied between two register sets): what tp_mom_bwd_first keeps --
// same-box A/B on MI355X (profiles/r04_v6_ab_c4.txt): the packed form is 6.5 % faster in tp_mom_bwd_last (1.019 -> 0.952 ms)
// and 3 % slower in tp_mom_bwd_first (220 instead of 204 registers at two waves per SIMD)
template <typename T, int D, int R>
__device__ __forceinline__ void mom_gm_plain(const T* sG, const T* Wt, int ka, int kb, int lane, T* gm) {
  constexpr int CB = 8;
#pragma unroll
  for (int j = 0; j < D; ++j) gm[j] = T(0);
  T wc[CB][R], wn[CB][R];
  auto loadw = [&](int c0, T(*w)[R]) {
#pragma unroll
    for (int i = 0; i < CB; ++i) {
      const int ch = c0 + i < 64 ? c0 + i : 63;
#pragma unroll
      for (int r = 0; r < R; ++r) w[i][r] = Wt[(int64_t(r) * 64 + ch) * ka + kb + lane];
    }
  };
  loadw(0, wc);
  for (int c0 = 0; c0 < kGmChannels; c0 += CB) {
    loadw(c0 + CB, wn);
#pragma unroll
    for (int i = 0; i < CB; ++i) {
#pragma unroll
      for (int j = 0; j < D; ++j) gm[j] += sG[j * 64 + c0 + i] * wc[i][r_of<0>(j)];
    }
#pragma unroll
    for (int i = 0; i < CB; ++i)
#pragma unroll
      for (int r = 0; r < R; ++r) wc[i][r] = wn[i][r];
  }
}

// GM[j] (k = kb + lane) = sum_ch g[j][ch] * Wt[r(j)][ch][k]   (g already carries the scatter factor)
// Instruction-issue-bound (a fifth of the reverse kernels' vector instructions): the components of one irrep share their weight,
// so they are accumulated two per instruction (v_pk_fma_f32: the g pair comes from LDS as an aligned 8-B cell, the weight is
// broadcast), and the weight rows are double-buffered by unrolling instead of being copied.  sG holds the per-atom gradient
// as [ch][kGmLd] with the PAIRS of every irrep first (components base+1+2i, base+2+2i of irrep r, i < r) and the leftover
// component of every irrep (base = r^2) behind them: gm_slot() below.
template <int R>
__device__ __forceinline__ constexpr int gm_num_pairs() {
  return R * (R - 1) / 2;
}
template <int R>
__device__ __forceinline__ constexpr int gm_ld() {
  return (R * R + 3) / 4 * 4;  // row stride (floats): 16-B cells
}
// position of component j inside a row of sG
template <int R>
__device__ __forceinline__ constexpr int gm_slot(int j) {
  const int r = j < 1 ? 0 : (j < 4 ? 1 : (j < 9 ? 2 : 3));
  const int base = r * r, off = j - base;
  if (off == 0) return 2 * gm_num_pairs<R>() + r;  // the leftover of irrep r
  return 2 * (r * (r - 1) / 2) + (off - 1);        // pairs of the irreps before + (off - 1)
}
template <typename T, int D, int R>
__device__ __forceinline__ void mom_gm_store(T* sG, const T* g2acc, int lane) {
  static_assert(64 * gm_ld<R>() <= D * (64 + 64), "the staging area is the wave's moments + gradient region: D x (ka_lds + 64), ka_lds >= 64");
#pragma unroll
  for (int j = 0; j < D; ++j) sG[lane * gm_ld<R>() + gm_slot<R>(j)] = g2acc[j];
}
template <typename T, int D, int R>
__device__ __forceinline__ void mom_gm(const T* sG, const T* Wt, int ka, int kb, int lane, T* gm) {
  typedef typename Pk<T>::type T2;
  constexpr int CB = 4, NP = gm_num_pairs<R>(), LD = gm_ld<R>();
  static_assert(64 % (2 * CB) == 0 && D == R * R, "spherical-harmonics-ordered components");
  T2 gp[NP > 0 ? NP : 1];
  T gs[R];
#pragma unroll
  for (int i = 0; i < NP; ++i) gp[i] = T2{T(0), T(0)};
#pragma unroll
  for (int r = 0; r < R; ++r) gs[r] = T(0);
  T wa[CB][R], wb[CB][R];
  auto loadw = [&](int c0, T(*w)[R]) {
#pragma unroll
    for (int i = 0; i < CB; ++i) {
#pragma unroll
      for (int r = 0; r < R; ++r) w[i][r] = Wt[(int64_t(r) * 64 + c0 + i) * ka + kb + lane];
    }
  };
  auto consume = [&](int c0, const T(*w)[R]) {
#pragma unroll
    for (int i = 0; i < CB; ++i) {
      const T* row = sG + (c0 + i) * LD;  // (wave-uniform address: broadcast reads)
      T2 gpair[NP > 0 ? NP : 1];
#pragma unroll
      for (int q = 0; q < NP; ++q) gpair[q] = *reinterpret_cast<const T2*>(row + 2 * q);
      T gsing[R];
#pragma unroll
      for (int r = 0; r < R; ++r) gsing[r] = row[2 * NP + r];
#pragma unroll
      for (int r = 0; r < R; ++r) {
        gs[r] += gsing[r] * w[i][r];
#pragma unroll
        for (int q = 0; q < r; ++q) gp[r * (r - 1) / 2 + q] += gpair[r * (r - 1) / 2 + q] * T2{w[i][r], w[i][r]};
      }
    }
  };
  loadw(0, wa);
#pragma unroll 1
  for (int c0 = 0; c0 < kGmChannels; c0 += 2 * CB) {  // (not unrolled further: the reverse kernels run at up to 4 waves per SIMD, 128 registers)
    loadw(c0 + CB, wb);
    consume(c0, wa);
    if (c0 + 2 * CB < kGmChannels) loadw(c0 + 2 * CB, wa);
    consume(c0 + CB, wb);
  }
#pragma unroll
  for (int r = 0; r < R; ++r) {
    gm[r * r] = gs[r];
#pragma unroll
    for (int q = 0; q < r; ++q) {
      gm[r * r + 1 + 2 * q] = gp[r * (r - 1) / 2 + q][0];
      gm[r * r + 2 + 2 * q] = gp[r * (r - 1) / 2 + q][1];
    }
  }
}

// adjoint of the moments for every edge of the segment:
//   d_a[e,k] = sum_j sh[e,j] * GM[j][k]        d_sh[e,j] = sum_k act(a[e,k]) * GM[j][k]
template <typename T, int D, int R, bool KA2, bool PACKED_GM>
__device__ __forceinline__ void mom_backward_edges(const T* sh, int ld_sh, const T* a, int ld_a, int ka, bool act,
                                                   const T* g2acc, const T* Wt, int beg, int end, int lane, T* sG, T* sY,
                                                   int& staged_cb, T* g_a, int ld_ga, T* gsh, int ld_gsh) {
  typedef typename Pk<T>::type T2;
  constexpr int B = 4;  // pairs per batch here (two channel blocks may be live)
  __builtin_amdgcn_wave_barrier();
  if constexpr (PACKED_GM) {
    mom_gm_store<T, D, R>(sG, g2acc, lane);
  } else {
#pragma unroll
    for (int j = 0; j < D; ++j) sG[j * 64 + lane] = g2acc[j];
  }
  __builtin_amdgcn_wave_barrier();
  T gm0[D], gm1[D];
  const bool two = KA2 && ka > 64;
  if constexpr (PACKED_GM) {
    mom_gm<T, D, R>(sG, Wt, ka, 0, lane, gm0);
    if (two) mom_gm<T, D, R>(sG, Wt, ka, 64, lane, gm1);
  } else {
    mom_gm_plain<T, D, R>(sG, Wt, ka, 0, lane, gm0);
    if (two) mom_gm_plain<T, D, R>(sG, Wt, ka, 64, lane, gm1);
  }
  const T2* sY2 = reinterpret_cast<const T2*>(sY);
  auto loadb = [&](int s0, int ce, T2* v0, T2* v1) {
#pragma unroll
    for (int i = 0; i < B; ++i) {
      const int e0 = s0 + 2 * i, e1 = e0 + 1;
      const int c0 = e0 < ce ? e0 : ce - 1, c1 = e1 < ce ? e1 : ce - 1;
      T x0 = ld_stream(a + int64_t(c0) * ld_a + lane), x1 = ld_stream(a + int64_t(c1) * ld_a + lane);
      if (act) {
        x0 = silu(x0);
        x1 = silu(x1);
      }
      v0[i] = T2{x0, x1};
      if (two) {
        T z0 = a[int64_t(c0) * ld_a + 64 + lane], z1 = a[int64_t(c1) * ld_a + 64 + lane];
        if (act) {
          z0 = silu(z0);
          z1 = silu(z1);
        }
        v1[i] = T2{z0, z1};
      }
    }
  };
  for (int cb = beg; cb < end; cb += kSegCap) {
    const int ce = cb + kSegCap < end ? cb + kSegCap : end;
    if (staged_cb != cb) {
      stage_sh<T, D>(sh, ld_sh, cb, ce, lane, sY);
      staged_cb = cb;
    }
    T2 c0v[B], c1v[B], n0v[B], n1v[B];
    loadb(cb, ce, c0v, c1v);
    for (int s0 = cb; s0 < ce; s0 += 2 * B) {
      loadb(s0 + 2 * B, ce, n0v, n1v);
#pragma unroll
      for (int i = 0; i < B; ++i) {
        const int s = s0 + 2 * i;
        if (s < ce) {
          const bool vb = s + 1 < ce;
          const T2* y = sY2 + ((s - cb) >> 1) * D;
          T2 d0 = T2{T(0), T(0)}, d1 = T2{T(0), T(0)}, gy[D];
#pragma unroll
          for (int j = 0; j < D; ++j) {
            const T2 yj = y[j];
            d0 += yj * gm0[j];
            gy[j] = c0v[i] * gm0[j];
            if (two) {
              d1 += yj * gm1[j];
              gy[j] += c1v[i] * gm1[j];
            }
          }
          T ga[D], gb[D];
#pragma unroll
          for (int j = 0; j < D; ++j) {
            ga[j] = gy[j][0];
            gb[j] = gy[j][1];
          }
          st_stream(g_a + int64_t(s) * ld_ga + lane, d0[0]);
          if (two) g_a[int64_t(s) * ld_ga + 64 + lane] = d1[0];
          if (vb) {
            st_stream(g_a + int64_t(s + 1) * ld_ga + lane, d0[1]);
            if (two) g_a[int64_t(s + 1) * ld_ga + 64 + lane] = d1[1];
          }
          wave_sum_store2<T, D>(ga, gb, gsh + int64_t(s) * ld_gsh, gsh + int64_t(s + (vb ? 1 : 0)) * ld_gsh, true, vb);
        }
      }
#pragma unroll
      for (int i = 0; i < B; ++i) {
        c0v[i] = n0v[i];
        c1v[i] = n1v[i];
      }
    }
  }
}

// ---- forward kernels: harmonics through the scalar path (SGPR operands), one pair of edges ahead.  They run
// at 4-5 waves/SIMD and within ~20% of the achievable HBM rate; the LDS-staged, deeper-pipelined form used by the
// reverse kernels costs them occupancy (measured slower), so they keep this leaner shape.
// M[j][k] = sum_e sh[e,j] * act(a[e,k]) into wave-private LDS sM[D][ka], then
// x2s[j] (this lane's channel) = f * sum_k M[j][k] * Wk[k][r(j)][ch]
template <typename T, int D, int R>
__device__ __forceinline__ void mom_x2s(const T* sh, int ld_sh, const T* a, int ld_a, int ka, bool act, const T* Wk, int beg,
                                        int end, T sf, int lane, T* sM, T* x2s) {
  for (int kb = 0; kb < ka; kb += 64) {
    T m[D];
#pragma unroll
    for (int j = 0; j < D; ++j) m[j] = T(0);
#pragma unroll 4
    for (int s = beg; s < end; ++s) {
      T av = ld_stream(a + int64_t(s) * ld_a + kb + lane);
      if (act) av = silu(av);
      const T* y = sh + int64_t(s) * ld_sh;
#pragma unroll
      for (int j = 0; j < D; ++j) m[j] += y[j] * av;
    }
#pragma unroll
    for (int j = 0; j < D; ++j) sM[j * ka + kb + lane] = m[j];
  }
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int j = 0; j < D; ++j) x2s[j] = T(0);
#pragma unroll 4
  for (int k = 0; k < ka; ++k) {
    T wv[R];
#pragma unroll
    for (int r = 0; r < R; ++r) wv[r] = Wk[(int64_t(k) * R + r) * 64 + lane];
#pragma unroll
    for (int j = 0; j < D; ++j) x2s[j] += sM[j * ka + k] * wv[r_of<0>(j)];
  }
#pragma unroll
  for (int j = 0; j < D; ++j) x2s[j] *= sf;
  __builtin_amdgcn_wave_barrier();
}

template <typename T, int D, int R>
struct EdgeIn2 {
  typename Pk<T>::type y[D];
  typename Pk<T>::type wa[R];
  typename Pk<T>::type g0, g1;
};


// per-pair operands of the contraction loops that come from HBM (the harmonics come from LDS)
template <typename T, int R>
struct PairIn {
  typename Pk<T>::type wa[R];
  typename Pk<T>::type g0, g1;
};
}  // namespace

// common prologue of the moments kernels
#define AA_MOM_PROLOGUE(DVAL)                                                                                          \
  const TpChainArgs& a = ma.c;                                                                                         \
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;                                                            \
  const int64_t atom = a.atom0 + int64_t(blockIdx.x) * (blockDim.x >> 6) + wv;                                         \
  const int ka_lds = ma.ka_lds;                                                                                        \
  T* sM = reinterpret_cast<T*>(aa_smem) + size_t(wv) * (DVAL) * (ka_lds + 64 + kSegCap);                               \
  T* sG = sM + (DVAL) * ka_lds;                                                                                        \
  T* sY = sG + (DVAL) * 64;                                                                                            \
  const T2* sY2 = reinterpret_cast<const T2*>(sY);                                                                     \
  (void)sG;                                                                                                            \
  (void)sM;                                                                                                            \
  (void)sY2;                                                                                                           \
  if (atom >= a.N) return;                                                                                             \
  const int beg = __builtin_amdgcn_readfirstlane(a.rowptr[atom]), end = __builtin_amdgcn_readfirstlane(a.rowptr[atom + 1]); \
  /* an atom without edges (isolated, or owned by another rank of an atom-block partition) has nothing to        \
     contribute or receive: its x2s rows are only ever read by its own (absent) edges */                           \
  if (beg >= end) return;                                                                                              \
  const T* sh = static_cast<const T*>(a.sh);                                                                           \
  const T* w0g = static_cast<const T*>(a.w0) + lane;                                                                   \
  int staged_cb = -1;                                                                                                 \
  (void)staged_cb;

// moments of the whole segment -> x2s (leaves the LAST chunk staged in sY)
#define AA_MOM_X2S(AIN, LDA, KA, ACT, WK, OUT)                                                                         \
  {                                                                                                                    \
    if (beg >= end) {                                                                                                  \
      for (int kb = 0; kb < (KA); kb += 64)                                                                            \
        for (int j = 0; j < D; ++j) sM[j * ka_lds + kb + lane] = T(0);                                                 \
    }                                                                                                                  \
    for (int cb = beg; cb < end; cb += kSegCap) {                                                                      \
      const int ce = cb + kSegCap < end ? cb + kSegCap : end;                                                          \
      stage_sh<T, D>(sh, a.ld_sh, cb, ce, lane, sY);                                                                   \
      staged_cb = cb;                                                                                                  \
      mom_accumulate<T, D>(static_cast<const T*>(AIN), LDA, KA, ka_lds, ACT, cb, ce, cb == beg, lane, sY, sM);         \
    }                                                                                                                  \
    mom_project<T, D, R>(sM, KA, ka_lds, static_cast<const T*>(WK), T(a.sf), lane, OUT);                               \
  }

// loop over the segment in staged chunks and edge pairs, with the HBM operands of the pair after next in flight:
// fetch(s, ce, PairIn&) requests the rows of pair (s, s+1) (clamped to the chunk), body(s, vb, y, cur) consumes
template <typename T, int D, int R, int AHEAD, class F, class B>
__device__ __forceinline__ void mom_pair_loop(const T* sh, int ld_sh, int beg, int end, int lane, T* sY, int& staged_cb,
                                              F&& fetch, B&& body) {
  typedef typename Pk<T>::type T2;
  const T2* sY2 = reinterpret_cast<const T2*>(sY);
  for (int cb = beg; cb < end; cb += kSegCap) {
    const int ce = cb + kSegCap < end ? cb + kSegCap : end;
    if (staged_cb != cb) {
      stage_sh<T, D>(sh, ld_sh, cb, ce, lane, sY);
      staged_cb = cb;
    }
    PairIn<T, R> cur, nx1, nx2;
    fetch(cb, ce, cur);
    if (AHEAD == 2) fetch(cb + 2, ce, nx1);
    for (int s = cb; s < ce; s += 2) {
      if (AHEAD == 2) {
        fetch(s + 4, ce, nx2);
      } else {
        fetch(s + 2, ce, nx1);
      }
      body(s, s + 1 < ce, sY2 + ((s - cb) >> 1) * D, cur);
      cur = nx1;
      if (AHEAD == 2) nx1 = nx2;
    }
  }
}

template <class Sig0, typename T, bool KA2>
__global__ __launch_bounds__(256) void tp_mom_fwd_first_kernel(TpMomArgs ma) {
  constexpr int D = Sig0::D2, R = Sig0::LMAX + 1;
  typedef typename Pk<T>::type T2;
  AA_MOM_PROLOGUE(D)
  T x2s0[D];
  mom_x2s<T, D, R>(sh, a.ld_sh, static_cast<const T*>(ma.a0), ma.ld_a0, ma.ka0, false, static_cast<const T*>(ma.wk0), beg, end,
                   T(a.sf), lane, sM, x2s0);
  {
    T* xo = static_cast<T*>(const_cast<void*>(a.x2s0)) + atom * D * 64 + lane;
#pragma unroll
    for (int j = 0; j < D; ++j) xo[int64_t(j) * 64] = x2s0[j];
  }
  T wp0[Sig0::P];
  {
    const T* W0 = static_cast<const T*>(a.weights0);
#pragma unroll
    for (int p = 0; p < Sig0::P; ++p) wp0[p] = a.coupling ? W0[lane * Sig0::P + p] : W0[p];
  }
  auto fetch = [&](int s, EdgeIn2<T, D, R>& in) {
    const int sb = s + 1 < end ? s + 1 : s;
    const T* ya = sh + int64_t(s) * a.ld_sh;
    const T* yb = sh + int64_t(sb) * a.ld_sh;
    const T* wa = w0g + int64_t(s) * a.ld_w0;
    const T* wb = w0g + int64_t(sb) * a.ld_w0;
#pragma unroll
    for (int j = 0; j < D; ++j) in.y[j] = T2{ya[j], yb[j]};
#pragma unroll
    for (int r = 0; r < R; ++r) in.wa[r] = T2{ld_stream(wa + r * 64), ld_stream(wb + r * 64)};
  };
  if (beg < end) {
    EdgeIn2<T, D, R> cur, nxt;
    fetch(beg, cur);
    T* sc = static_cast<T*>(a.scal1) + lane;  // (scal1 field carries the layer-0 scalars here)
    for (int s = beg; s < end; s += 2) {
      fetch(s + 2 < end ? s + 2 : s, nxt);
      T2 x1[Sig0::D1], tf1[Sig0::DOUT];
#pragma unroll
      for (int i = 0; i < Sig0::D1; ++i) x1[i] = cur.y[i] * cur.wa[r_of<0>(i)];
      Sig0::template fwd4<T2, T2, T, T>(x1, x2s0, wp0, tf1);
      st_stream(sc + int64_t(s) * a.ld_scal, tf1[0][0]);
      if (s + 1 < end) st_stream(sc + int64_t(s + 1) * a.ld_scal, tf1[0][1]);
      cur = nxt;
    }
  }
}

template <class Sig0, class Sig1, typename T, bool KA2>
__global__ __launch_bounds__(256) void tp_mom_fwd_last_kernel(TpMomArgs ma) {
  constexpr int D = Sig0::D2, R = Sig0::LMAX + 1;
  typedef typename Pk<T>::type T2;
  AA_MOM_PROLOGUE(D)
  T x2s1[D], x2s0[D];
  mom_x2s<T, D, R>(sh, a.ld_sh, static_cast<const T*>(ma.a1), ma.ld_a1, ma.ka1, true, static_cast<const T*>(ma.wk1), beg, end,
                   T(a.sf), lane, sM, x2s1);
  {
    const int64_t base = atom * D * 64 + lane;
    const T* xi = static_cast<const T*>(a.x2s0) + base;
    T* xo = static_cast<T*>(a.x2s1) + base;
#pragma unroll
    for (int j = 0; j < D; ++j) {
      x2s0[j] = xi[int64_t(j) * 64];
      xo[int64_t(j) * 64] = x2s1[j];
    }
  }
  T wp0[Sig0::P], wp1[Sig1::P];
  {
    const T* W0 = static_cast<const T*>(a.weights0);
    const T* W1 = static_cast<const T*>(a.weights1);
#pragma unroll
    for (int p = 0; p < Sig0::P; ++p) wp0[p] = a.coupling ? W0[lane * Sig0::P + p] : W0[p];
#pragma unroll
    for (int p = 0; p < Sig1::P; ++p) wp1[p] = a.coupling ? W1[lane * Sig1::P + p] : W1[p];
  }
  // The last layer contracts to one scalar and layer 0 is linear in x1 for fixed x2s0, so the whole per-edge chain
  //   scal1[e] = Sig1(Sig0(x1[e], x2s0), x2s1) = sum_a x1[e][a] * B1[a],   B1 = Sig0^T_x1(v, x2s0),  v = dSig1/dtf1 (x2s1)
  // needs the Clebsch-Gordan contractions once per ATOM (here) instead of once per edge.
  T B1[Sig0::D1];
  {
    T one[1] = {T(1)}, v[Sig1::D1];
    Sig1::template bx1<T>(one, x2s1, wp1, v);
    Sig0::template bx1<T>(v, x2s0, wp0, B1);
  }
  auto fetch = [&](int s, EdgeIn2<T, D, R>& in) {
    const int sb = s + 1 < end ? s + 1 : s;
    const T* ya = sh + int64_t(s) * a.ld_sh;
    const T* yb = sh + int64_t(sb) * a.ld_sh;
    const T* wa = w0g + int64_t(s) * a.ld_w0;
    const T* wb = w0g + int64_t(sb) * a.ld_w0;
#pragma unroll
    for (int j = 0; j < D; ++j) in.y[j] = T2{ya[j], yb[j]};
#pragma unroll
    for (int r = 0; r < R; ++r) in.wa[r] = T2{ld_stream(wa + r * 64), ld_stream(wb + r * 64)};
  };
  if (beg < end) {
    EdgeIn2<T, D, R> cur, nxt;
    fetch(beg, cur);
    T* sc = static_cast<T*>(a.scal1) + lane;
    for (int s = beg; s < end; s += 2) {
      fetch(s + 2 < end ? s + 2 : s, nxt);
      // scal1[e] = <x1[e], B1>: both layers are linear in x1 once the per-atom vectors are fixed (see B1 above)
      T2 out[1];
      out[0] = T2{T(0), T(0)};
#pragma unroll
      for (int i = 0; i < Sig0::D1; ++i) out[0] += (cur.y[i] * cur.wa[r_of<0>(i)]) * B1[i];
      st_stream(sc + int64_t(s) * a.ld_scal, out[0][0]);
      if (s + 1 < end) st_stream(sc + int64_t(s + 1) * a.ld_scal, out[0][1]);
      cur = nxt;
    }
  }
}

// KA2: env inputs wider than 64 (two k blocks per lane) -- a compile-time switch because the second block costs the
// reverse kernels ~50 VGPRs (4 instead of 3 waves/SIMD without it)
template <class Sig0, class Sig1, typename T, bool KA2>
__global__ __launch_bounds__(256, (sizeof(T) == 4 && Sig0::LMAX <= 2) ? (KA2 ? 3 : 4) : 1) void tp_mom_bwd_last_kernel(TpMomArgs ma) {
  constexpr int D = Sig0::D2, R = Sig0::LMAX + 1;
  typedef typename Pk<T>::type T2;
  AA_MOM_PROLOGUE(D)
  const T* gs1 = static_cast<const T*>(a.gscal1) + lane;
  T x2s0[D], g2acc[D];
  {
    const T* xi = static_cast<const T*>(a.x2s0) + atom * D * 64 + lane;
#pragma unroll
    for (int j = 0; j < D; ++j) {
      x2s0[j] = xi[int64_t(j) * 64];
      g2acc[j] = T(0);
    }
  }
  T wp0[Sig0::P], wp1[Sig1::P];
  {
    const T* W0 = static_cast<const T*>(a.weights0);
    const T* W1 = static_cast<const T*>(a.weights1);
#pragma unroll
    for (int p = 0; p < Sig0::P; ++p) wp0[p] = a.coupling ? W0[lane * Sig0::P + p] : W0[p];
#pragma unroll
    for (int p = 0; p < Sig1::P; ++p) wp1[p] = a.coupling ? W1[lane * Sig1::P + p] : W1[p];
  }
  T2 Q1[Sig0::D1];
#pragma unroll
  for (int i = 0; i < Sig0::D1; ++i) Q1[i] = T2{T(0), T(0)};
  auto fetch = [&](int s, int ce, PairIn<T, R>& in) {
    const int sa = s < ce ? s : ce - 1, sb = s + 1 < ce ? s + 1 : ce - 1;
    const T* wa = w0g + int64_t(sa) * a.ld_w0;
    const T* wb = w0g + int64_t(sb) * a.ld_w0;
#pragma unroll
    for (int r = 0; r < R; ++r) in.wa[r] = T2{ld_stream(wa + r * 64), ld_stream(wb + r * 64)};
    const T ga = ld_stream(gs1 + int64_t(sa) * a.ld_gscal), gb = ld_stream(gs1 + int64_t(sb) * a.ld_gscal);
    in.g1 = T2{ga, s + 1 < ce ? gb : T(0)};  // a padded second edge contributes nothing
  };
  mom_pair_loop<T, D, R, 2>(sh, a.ld_sh, beg, end, lane, sY, staged_cb, fetch, [&](int s, bool vb, const T2* y, const PairIn<T, R>& cur) {
    // d x2s1 = Sig1^T_x2(1, sum_e g[e] * tf1[e]) and sum_e g[e] * tf1[e] = Sig0(sum_e g[e] * x1[e], x2s0): only the
    // g-weighted moment of x1 is accumulated per edge; the contractions run once per atom after the loop
#pragma unroll
    for (int i = 0; i < Sig0::D1; ++i) Q1[i] += cur.g1 * (y[i] * cur.wa[r_of<0>(i)]);
  });
  {
    T q[Sig0::D1], tq[Sig0::DOUT], one[1] = {T(1)};
#pragma unroll
    for (int i = 0; i < Sig0::D1; ++i) q[i] = Q1[i][0] + Q1[i][1];
    Sig0::template fwd<T>(q, x2s0, wp0, tq);
    Sig1::template bx2<T>(one, tq, wp1, g2acc);
  }
  const T sf = T(a.sf);
#pragma unroll
  for (int j = 0; j < D; ++j) g2acc[j] *= sf;
  mom_backward_edges<T, D, R, KA2, true>(sh, a.ld_sh, static_cast<const T*>(ma.a1), ma.ld_a1, ma.ka1, true, g2acc,
                              static_cast<const T*>(ma.wt1), beg, end, lane, sM /* [64][gm_ld]: spans the (here unused) moments area and sG */, sY, staged_cb, static_cast<T*>(ma.g_a),
                              ma.ld_ga, static_cast<T*>(a.gsh_env), a.ld_gsh);
}

template <class Sig0, class Sig1, typename T, bool KA2>
#ifndef AA_MOM_FIRST_WAVES
#define AA_MOM_FIRST_WAVES 2  // waves per SIMD the register budget is sized for (A/B: 3 = 168 VGPRs)
#endif
__global__ __launch_bounds__(256, (sizeof(T) == 4 && Sig0::LMAX <= 2) ? AA_MOM_FIRST_WAVES : 1) void tp_mom_bwd_first_kernel(TpMomArgs ma) {
  constexpr int D = Sig0::D2, D1 = Sig0::D1, DOUT = Sig0::DOUT, R = Sig0::LMAX + 1;
  typedef typename Pk<T>::type T2;
  AA_MOM_PROLOGUE(D)
  // (uniform row pointers + one 32-bit lane offset: every stream access of the pair loop is a saddr + voffset instruction)
  const T* gs0 = static_cast<const T*>(a.gscal0);
  const T* gs1 = static_cast<const T*>(a.gscal1);
  const T* w0u = static_cast<const T*>(a.w0);
  const unsigned lane_b0 = unsigned(lane) * unsigned(sizeof(T));
  T x2s0[D], x2s1[D], g2acc[D];
  {
    const int64_t base = atom * D * 64 + lane;
    const T* x0 = static_cast<const T*>(a.x2s0) + base;
    const T* x1p = static_cast<const T*>(a.x2s1) + base;
#pragma unroll
    for (int j = 0; j < D; ++j) {
      x2s0[j] = x0[int64_t(j) * 64];
      x2s1[j] = x1p[int64_t(j) * 64];
      g2acc[j] = T(0);
    }
  }
  T wp0[Sig0::P], wp1[Sig1::P];
  {
    const T* W0 = static_cast<const T*>(a.weights0);
    const T* W1 = static_cast<const T*>(a.weights1);
#pragma unroll
    for (int p = 0; p < Sig0::P; ++p) wp0[p] = a.coupling ? W0[lane * Sig0::P + p] : W0[p];
#pragma unroll
    for (int p = 0; p < Sig1::P; ++p) wp1[p] = a.coupling ? W1[lane * Sig1::P + p] : W1[p];
  }
  // per-atom vectors (see the loop body)
  T vv[DOUT], B1[D1], B0[D1];
  {
    T one[1] = {T(1)}, e0[DOUT];
#pragma unroll
    for (int k = 0; k < DOUT; ++k) e0[k] = k == 0 ? T(1) : T(0);
    Sig1::template bx1<T>(one, x2s1, wp1, vv);
    Sig0::template bx1<T>(vv, x2s0, wp0, B1);
    Sig0::template bx1<T>(e0, x2s0, wp0, B0);
  }
  T2 Q1[D1], Q0[D1];
#pragma unroll
  for (int i = 0; i < D1; ++i) {
    Q1[i] = T2{T(0), T(0)};
    Q0[i] = T2{T(0), T(0)};
  }
  auto fetch = [&](int s, int ce, PairIn<T, R>& in) {
    const int sa = s < ce ? s : ce - 1, sb = s + 1 < ce ? s + 1 : ce - 1;
    unsigned lane_b = lane_b0;
    opaque_vector(lane_b);  // (keeps the zero-extension of the lane offset next to its loads: hoisted as a 64-bit pair it defeats the saddr pattern)
    const T* wa = w0u + int64_t(sa) * a.ld_w0;
    const T* wb = w0u + int64_t(sb) * a.ld_w0;
#pragma unroll
    for (int r = 0; r < R; ++r) in.wa[r] = T2{ld_stream(lane_at(wa + r * 64, lane_b)), ld_stream(lane_at(wb + r * 64, lane_b))};
    const bool vb2 = s + 1 < ce;
    const T g0a = ld_stream(lane_at(gs0 + int64_t(sa) * a.ld_gscal, lane_b)), g0b = ld_stream(lane_at(gs0 + int64_t(sb) * a.ld_gscal, lane_b));
    const T g1a = ld_stream(lane_at(gs1 + int64_t(sa) * a.ld_gscal, lane_b)), g1b = ld_stream(lane_at(gs1 + int64_t(sb) * a.ld_gscal, lane_b));
    in.g0 = T2{g0a, vb2 ? g0b : T(0)};
    in.g1 = T2{g1a, vb2 ? g1b : T(0)};
  };
  T* gw0 = static_cast<T*>(a.g_w0);
  T* gsx = static_cast<T*>(a.gsh_x1);
#ifndef AA_MOM_FIRST_AHEAD
#define AA_MOM_FIRST_AHEAD 1  // pairs of HBM operands in flight ahead of the contraction (A/B: profiles/r05_v10_ab_*)
#endif
  mom_pair_loop<T, D, R, AA_MOM_FIRST_AHEAD>(sh, a.ld_sh, beg, end, lane, sY, staged_cb, fetch, [&](int s, bool vb, const T2* y, const PairIn<T, R>& cur) {
    // go[e] = g1[e] * v + g0[e] * e_0 (v = dSig1/dtf1), so with the per-atom vectors B1 = Sig0^T_x1(v), B0 = Sig0^T_x1(e_0)
    //   d x1[e] = g1[e] * B1 + g0[e] * B0      and      d x2s0 = Sig0^T_x2(v, sum_e g1 x1) + Sig0^T_x2(e_0, sum_e g0 x1)
    T2 gw[R];
    T gya[D1], gyb[D1];
#pragma unroll
    for (int r = 0; r < R; ++r) gw[r] = T2{T(0), T(0)};
#pragma unroll
    for (int i = 0; i < D1; ++i) {
      const T2 x1 = y[i] * cur.wa[r_of<0>(i)];
      Q1[i] += cur.g1 * x1;
      Q0[i] += cur.g0 * x1;
      const T2 gx = cur.g1 * B1[i] + cur.g0 * B0[i];
      gw[r_of<0>(i)] += gx * y[i];
      const T2 t = gx * cur.wa[r_of<0>(i)];
      gya[i] = t[0];
      gyb[i] = t[1];
    }
    unsigned lane_b = lane_b0;
    opaque_vector(lane_b);
#pragma unroll
    for (int r = 0; r < R; ++r) st_stream(lane_at(gw0 + int64_t(s) * a.ld_gw0 + r * 64, lane_b), gw[r][0]);
    if (vb) {
#pragma unroll
      for (int r = 0; r < R; ++r) st_stream(lane_at(gw0 + int64_t(s + 1) * a.ld_gw0 + r * 64, lane_b), gw[r][1]);
    }
    wave_sum_store2<T, D1>(gya, gyb, gsx + int64_t(s) * a.ld_gsh, gsx + int64_t(s + (vb ? 1 : 0)) * a.ld_gsh, true, vb);
  });
  {
    T q1[D1], q0[D1], ga[D], gb[D], e0[DOUT];
#pragma unroll
    for (int i = 0; i < D1; ++i) {
      q1[i] = Q1[i][0] + Q1[i][1];
      q0[i] = Q0[i][0] + Q0[i][1];
    }
#pragma unroll
    for (int k = 0; k < DOUT; ++k) e0[k] = k == 0 ? T(1) : T(0);
    Sig0::template bx2<T>(vv, q1, wp0, ga);
    Sig0::template bx2<T>(e0, q0, wp0, gb);
#pragma unroll
    for (int j = 0; j < D; ++j) g2acc[j] = ga[j] + gb[j];
  }
  const T sf = T(a.sf);
#pragma unroll
  for (int j = 0; j < D; ++j) g2acc[j] *= sf;
  mom_backward_edges<T, D, R, KA2, false>(sh, a.ld_sh, static_cast<const T*>(ma.a0), ma.ld_a0, ma.ka0, false, g2acc,
                              static_cast<const T*>(ma.wt0), beg, end, lane, sG, sY, staged_cb, static_cast<T*>(ma.g_a),
                              ma.ld_ga, static_cast<T*>(a.gsh_env), a.ld_gsh);
}

// (layer-0 signature, last-layer signature) pairs of 2-layer stacks at l_max = 1, 2, 3
#define AA_FOREACH_CHAIN(X) X(0, Sig1, Sig0) X(1, Sig5, Sig4) X(2, Sig9, Sig8)
int find_chain_pair(int sig0, int sig1) {
  if (sig0 == 1 && sig1 == 0) return 0;
  if (sig0 == 5 && sig1 == 4) return 1;
  if (sig0 == 9 && sig1 == 8) return 2;
  return -1;
}

#define AA_CHAIN_LAUNCHER(NAME)                                                                          \
  template <typename T>                                                                                  \
  int launch_##NAME(int pair, const TpChainArgs& a, hipStream_t stream) {                                \
    if (a.N == 0) return AA_OK;                                                                          \
    dim3 grid(spec_grid(a.u, a.N));                                                                      \
    switch (pair) {                                                                                      \
      case 0: hipLaunchKernelGGL((NAME##_kernel<cg::Sig1, cg::Sig0, T>), grid, dim3(256), 0, stream, a); break; \
      case 1: hipLaunchKernelGGL((NAME##_kernel<cg::Sig5, cg::Sig4, T>), grid, dim3(256), 0, stream, a); break; \
      case 2: hipLaunchKernelGGL((NAME##_kernel<cg::Sig9, cg::Sig8, T>), grid, dim3(256), 0, stream, a); break; \
      default: return fail(AA_ERR_INVALID, #NAME ": unknown chain pair");                                \
    }                                                                                                    \
    AA_CHECK_HIP(hipGetLastError());                                                                     \
    return AA_OK;                                                                                        \
  }                                                                                                      \
  template int launch_##NAME<float>(int, const TpChainArgs&, hipStream_t);                               \
  template int launch_##NAME<double>(int, const TpChainArgs&, hipStream_t);

// ---------------------------------------------------------------------------------------------
int find_spec_sig(const aa_tp_desc& d) {
  if (d.mul < 1 || d.mul > 256 || (d.mul & (d.mul - 1)) != 0) return -1;  // power of two <= 256
  for (int sgi = 0; sgi < cg::kNumSigs; ++sgi) {
    const cg::SigInfo& s = cg::kSigs[sgi];
    if (s.d1 != d.d1 || s.d2 != d.d2 || s.dout != d.dout || s.num_paths != d.num_paths || s.nnz != d.nnz) continue;
    // compare as sets of (i,j,k,p) -> val
    bool ok = true;
    for (int n = 0; n < d.nnz && ok; ++n) {
      bool found = false;
      for (int q = 0; q < s.nnz; ++q) {
        if (s.nz[q][0] == d.nz_i[n] && s.nz[q][1] == d.nz_j[n] && s.nz[q][2] == d.nz_k[n] && s.nz[q][3] == d.nz_path[n]) {
          found = std::fabs(s.val[q] - d.nz_val[n]) <= 1e-6 * std::max(1.0, std::fabs(s.val[q]));
          break;
        }
      }
      ok = found;
    }
    if (ok) return sgi;
  }
  return -1;
}


template <typename T>
int launch_tp_spec_fwd(int sig, const TpSpecFwdArgs& a, hipStream_t stream) {
  if (a.N == 0) return AA_OK;
  dim3 grid(spec_grid(a.u, a.N));
  switch (sig) {
#define AA_CASE(ID, SIG)                                                                     \
  case ID:                                                                                   \
    hipLaunchKernelGGL((tp_spec_fwd_kernel<cg::SIG, T>), grid, dim3(256), 0, stream, a);     \
    break;
    AA_FOREACH_SIG(AA_CASE)
#undef AA_CASE
    default:
      return fail(AA_ERR_INVALID, "tp spec fwd: unknown signature");
  }
  AA_CHECK_HIP(hipGetLastError());
  return AA_OK;
}

template <typename T>
int launch_tp_spec_bwd(int sig, const TpSpecBwdArgs& a, hipStream_t stream) {
  if (a.N == 0) return AA_OK;
  dim3 grid(spec_grid(a.u, a.N));
  switch (sig) {
#define AA_CASE(ID, SIG)                                                                     \
  case ID:                                                                                   \
    hipLaunchKernelGGL((tp_spec_bwd_kernel<cg::SIG, T>), grid, dim3(256), 0, stream, a);     \
    break;
    AA_FOREACH_SIG(AA_CASE)
#undef AA_CASE
    default:
      return fail(AA_ERR_INVALID, "tp spec bwd: unknown signature");
  }
  AA_CHECK_HIP(hipGetLastError());
  return AA_OK;
}

#define AA_MOM_LAUNCHER(NAME, K1, K2, K3)                                                                 \
  template <typename T>                                                                                  \
  int launch_##NAME(int pair, const TpMomArgs& a, hipStream_t stream) {                                  \
    if (a.c.N == 0) return AA_OK;                                                                        \
    if (a.c.u != 64 || a.ka0 > kMaxKa || a.ka1 > kMaxKa || (a.ka0 & 63) || (a.ka1 & 63))                \
      return fail(AA_ERR_INVALID, #NAME ": needs u == 64 and env-input widths of 64 or 128");            \
    /* one wave per atom and no inter-wave cooperation: single-wave workgroups give the dispatcher the finest    \
       granularity (shorter tail on small boxes / per-rank shards) */                                            \
    const int wpb = std::max(1, std::min(4, a.waves_per_block));                                                 \
    if (a.c.N <= a.c.atom0) return AA_OK;                                                                \
    dim3 grid((unsigned)((a.c.N - a.c.atom0 + wpb - 1) / wpb));                                          \
    const int dpair = pair == 0 ? 4 : (pair == 1 ? 9 : 16);                                              \
    TpMomArgs b = a;                                                                                     \
    b.ka_lds = a.ka0 > a.ka1 ? a.ka0 : a.ka1;                                                            \
    size_t smem = sizeof(T) * wpb * dpair * (b.ka_lds + 64 + kSegCap);                                   \
    if (smem > 160 * 1024) return fail(AA_ERR_INVALID, #NAME ": LDS patch too large for this dtype/l_max"); \
    if (smem > 64 * 1024) {                                                                              \
      const void* fn = pair == 0 ? (const void*)NAME##_kernel<K1, T, true>                               \
                                 : (pair == 1 ? (const void*)NAME##_kernel<K2, T, true> : (const void*)NAME##_kernel<K3, T, true>); \
      const void* fn1 = pair == 0 ? (const void*)NAME##_kernel<K1, T, false>                             \
                                  : (pair == 1 ? (const void*)NAME##_kernel<K2, T, false> : (const void*)NAME##_kernel<K3, T, false>); \
      AA_CHECK_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, int(smem)));      \
      AA_CHECK_HIP(hipFuncSetAttribute(fn1, hipFuncAttributeMaxDynamicSharedMemorySize, int(smem)));     \
    }                                                                                                    \
    const bool wide = b.ka_lds > 64;                                                                     \
    switch (pair * 2 + (wide ? 1 : 0)) {                                                                 \
      case 0: hipLaunchKernelGGL((NAME##_kernel<K1, T, false>), grid, dim3(64 * wpb), smem, stream, b); break; \
      case 1: hipLaunchKernelGGL((NAME##_kernel<K1, T, true>), grid, dim3(64 * wpb), smem, stream, b); break;  \
      case 2: hipLaunchKernelGGL((NAME##_kernel<K2, T, false>), grid, dim3(64 * wpb), smem, stream, b); break; \
      case 3: hipLaunchKernelGGL((NAME##_kernel<K2, T, true>), grid, dim3(64 * wpb), smem, stream, b); break;  \
      case 4: hipLaunchKernelGGL((NAME##_kernel<K3, T, false>), grid, dim3(64 * wpb), smem, stream, b); break; \
      case 5: hipLaunchKernelGGL((NAME##_kernel<K3, T, true>), grid, dim3(64 * wpb), smem, stream, b); break;  \
      default: return fail(AA_ERR_INVALID, #NAME ": unknown chain pair");                                \
    }                                                                                                    \
    AA_CHECK_HIP(hipGetLastError());                                                                     \
    return AA_OK;                                                                                        \
  }                                                                                                      \
  template int launch_##NAME<float>(int, const TpMomArgs&, hipStream_t);                                 \
  template int launch_##NAME<double>(int, const TpMomArgs&, hipStream_t);
#define AA_COMMA ,
AA_MOM_LAUNCHER(tp_mom_fwd_first, cg::Sig1, cg::Sig5, cg::Sig9)
AA_MOM_LAUNCHER(tp_mom_fwd_last, cg::Sig1 AA_COMMA cg::Sig0, cg::Sig5 AA_COMMA cg::Sig4, cg::Sig9 AA_COMMA cg::Sig8)
AA_MOM_LAUNCHER(tp_mom_bwd_last, cg::Sig1 AA_COMMA cg::Sig0, cg::Sig5 AA_COMMA cg::Sig4, cg::Sig9 AA_COMMA cg::Sig8)
AA_MOM_LAUNCHER(tp_mom_bwd_first, cg::Sig1 AA_COMMA cg::Sig0, cg::Sig5 AA_COMMA cg::Sig4, cg::Sig9 AA_COMMA cg::Sig8)

AA_CHAIN_LAUNCHER(tp_chain_fwd_last)
AA_CHAIN_LAUNCHER(tp_chain_bwd_last)
AA_CHAIN_LAUNCHER(tp_chain_bwd_first)

template int launch_tp_spec_fwd<float>(int, const TpSpecFwdArgs&, hipStream_t);
template int launch_tp_spec_fwd<double>(int, const TpSpecFwdArgs&, hipStream_t);
template int launch_tp_spec_bwd<float>(int, const TpSpecBwdArgs&, hipStream_t);
template int launch_tp_spec_bwd<double>(int, const TpSpecBwdArgs&, hipStream_t);

}  // namespace aa
