// Edge prologue / epilogue of the Allegro hot path (gfx950):
//   prologue : with_edge_vectors_ (called at allegro/nn/tensorembed.py:86), EdgeLengthNormalizer +
//              Bessel x polynomial cutoff + ProductTypeEmbedding (allegro/nn/scalarembed.py:60-81,
//              allegro/nn/_edgeembed.py:68-84), SphericalHarmonics (tensorembed.py:92)
//   readout  : last linear of edge_readout + EdgewiseReduce + PerTypeScaleShift
//              (allegro_models.py:231-260, allegro/nn/edgewise.py:40-60)
//   backward : hand-written reverse of the prologue -> dE/dr_ij -> forces on center and neighbor
//              (what ForceStressOutput's autograd does in the reference, allegro_models.py:101-103)
#include "aa_common.h"
#include "aa_geom.h"

namespace aa {

__host__ __device__ inline int prologue_ldb(int B) { return B == 8 ? 12 : B + 1; }

// STAGE_SH: the block's harmonics go through LDS and leave as one contiguous run (fp32 stacks: +9-16 KB of LDS);
// without it every lane stores its own D values (large fp64 stacks, where the extra LDS would halve the occupancy)
template <typename T, bool STAGE_SH>
__global__ __launch_bounds__(256) void edge_prologue_kernel(EdgeGeomArgs a) {
  const int B = a.num_bessels, S0 = a.S0, D = (a.l_max + 1) * (a.l_max + 1), Tn = a.num_types;
  const int ldb = prologue_ldb(B);                 // row stride of sB (B = 8: 12, so that a row is two aligned 16-B reads)
  T* sB = reinterpret_cast<T*>(aa_smem);           // [256][ldb]
  T* sWb = sB + 256 * ldb;                         // [B][S0]
  T* sSh = sWb + B * S0;                           // [256][D]: the block's harmonics, written out as one contiguous run
  int* sTy = reinterpret_cast<int*>(sSh + (STAGE_SH ? 256 * D : 0));  // [256][2]
  const int tid = threadIdx.x;
  const int64_t e0 = int64_t(blockIdx.x) * 256;
  const T* pos = static_cast<const T*>(a.pos);
  for (int idx = tid; idx < B * S0; idx += 256) sWb[idx] = static_cast<const T*>(a.basis_w)[idx];
  {
    int64_t e = e0 + tid;
    if (e < a.E) {
      int i = a.center[e], j = a.nbr[e];
      T vx = pos[3 * int64_t(j)] - pos[3 * int64_t(i)];
      T vy = pos[3 * int64_t(j) + 1] - pos[3 * int64_t(i) + 1];
      T vz = pos[3 * int64_t(j) + 2] - pos[3 * int64_t(i) + 2];
      if (a.shift_vec) {
        const T* sv = static_cast<const T*>(a.shift_vec) + 3 * e;
        vx += sv[0];
        vy += sv[1];
        vz += sv[2];
      }
      T r = aa_sqrt(vx * vx + vy * vy + vz * vz);
      T inv = T(1) / r;
      T nx = vx * inv, ny = vy * inv, nz = vz * inv;
      typedef T V4 __attribute__((ext_vector_type(4)));
      *reinterpret_cast<V4*>(static_cast<T*>(a.vec) + 4 * e) = V4{nx, ny, nz, r};
      T Y[16];
      sh_eval<T>(a.l_max, nx, ny, nz, Y);
      if (STAGE_SH) {
        for (int m = 0; m < D; ++m) sSh[tid * D + m] = Y[m];
      } else {
        T* sh = static_cast<T*>(a.sh) + e * D;
        for (int m = 0; m < D; ++m) sh[m] = Y[m];
      }
      int ti = a.types[i], tj = a.types[j];
      sTy[2 * tid] = ti;
      sTy[2 * tid + 1] = tj;
      T x = r * static_cast<const T*>(a.rmax_recip)[ti * Tn + tj];
      if (a.embed_kind == 1) {
        for (int nb = 0; nb < B; ++nb) {
          T bv, dbv;
          spline_basis_and_grad<T>(x, nb, B, a.spline_span, bv, dbv);
          sB[tid * ldb + nb] = bv;
        }
      } else {
        T f, df;
        cutoff_and_grad<T>(x, T(a.poly_p), f, df);
        const T fx = f / x;
        for (int nb = 0; nb < B; ++nb) {
          T w = static_cast<const T*>(a.bessel_w)[nb];
          sB[tid * ldb + nb] = aa_sin(w * x) * fx;
        }
      }
    }
  }
  __syncthreads();
  if (STAGE_SH) {
    // harmonics of the block's edges: rows [e0, e0 + 256) of sh are one contiguous run
    const int64_t n = (a.E - e0 < 256 ? a.E - e0 : 256) * D;
    T* sh = static_cast<T*>(a.sh) + e0 * D;
    for (int idx = tid; idx < n; idx += 256) sh[idx] = sSh[idx];
  }
  if (a.embed_kind == 1) {
    // emb0[e][c] = sum_s W[class(e)][c][s] b_s(x)  (spline.py:69-79), class = t_center * T + t_neighbor
    // (scalarembed.py:170); table is basis-major [class][s][c] so a wave reads contiguous rows
    const T* tab = static_cast<const T*>(a.emb_tab);
    for (int idx = tid; idx < 256 * S0; idx += 256) {
      int le = idx / S0, c = idx % S0;
      int64_t e = e0 + le;
      if (e >= a.E) break;
      const T* tc = tab + size_t(sTy[2 * le] * Tn + sTy[2 * le + 1]) * B * S0 + c;
      T acc = T(0);
      for (int nb = 0; nb < B; ++nb) acc += sB[le * ldb + nb] * tc[size_t(nb) * S0];
      static_cast<T*>(a.emb0)[e * S0 + c] = acc;
    }
    return;
  }
  const int half = S0 / 2;
  const T* cemb = static_cast<const T*>(a.center_embed);
  const T* nemb = static_cast<const T*>(a.neighbor_embed);
  if (S0 == 64 && B <= kMaxBessel) {
    // a wave writes one 256-B row per iteration; the lane's column is fixed, so its basis weights live in registers
    const int c = tid & 63;
    T wreg[kMaxBessel];
#pragma unroll
    for (int nb = 0; nb < kMaxBessel; ++nb) wreg[nb] = nb < B ? sWb[nb * S0 + c] : T(0);
    const T* tab = c < half ? cemb + c : nemb + (c - half);
    const int tsel = c < half ? 0 : 1;
#pragma unroll 4
    for (int le = tid >> 6; le < 256; le += 4) {
      const int64_t e = e0 + le;
      if (e >= a.E) break;
      T basis = T(0);
      if (B == 8) {  // (the usual basis size: the row is two aligned vector reads, broadcast to the wave)
        typedef T V4 __attribute__((ext_vector_type(4)));
        const V4 b0 = *reinterpret_cast<const V4*>(sB + le * ldb), b1 = *reinterpret_cast<const V4*>(sB + le * ldb + 4);
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) basis += b0[nb] * wreg[nb] + b1[nb] * wreg[4 + nb];
      } else {
#pragma unroll
        for (int nb = 0; nb < kMaxBessel; ++nb)
          if (nb < B) basis += sB[le * ldb + nb] * wreg[nb];
      }
      const T te = tab[sTy[2 * le + tsel] * half];
      static_cast<T*>(a.emb0)[e * S0 + c] = te * basis;
    }
    return;
  }
  for (int idx = tid; idx < 256 * S0; idx += 256) {
    int le = idx / S0, c = idx % S0;
    int64_t e = e0 + le;
    if (e >= a.E) break;
    T basis = T(0);
    for (int nb = 0; nb < B; ++nb) basis += sB[le * ldb + nb] * sWb[nb * S0 + c];
    T te = c < half ? cemb[sTy[2 * le] * half + c] : nemb[sTy[2 * le + 1] * half + (c - half)];
    static_cast<T*>(a.emb0)[e * S0 + c] = te * basis;
  }
}

template <typename T, int COLS>
__device__ __forceinline__ void edge_bwd_phase1_b8(const EdgeBwdArgs& b, int64_t e0, int tid, const int* sTy, const T* sEmb, T* sT) {
  const EdgeGeomArgs& a = b.g;
  constexpr int S0 = COLS * 8, half = S0 / 2, B = 8;
  const int sub = tid & 7;
  T w[COLS][B];
  {
    const T* wb = static_cast<const T*>(a.basis_w);  // [B][S0]
#pragma unroll
    for (int cc = 0; cc < COLS; ++cc)
#pragma unroll
      for (int nb = 0; nb < B; ++nb) w[cc][nb] = wb[nb * S0 + sub * COLS + cc];
  }
#pragma unroll
  for (int pb = 0; pb < 8; pb += 4) {
    T g[4][COLS];
#pragma unroll
    for (int ps = 0; ps < 4; ++ps) {
      const int le = (pb + ps) * 32 + (tid >> 3);
      const int64_t e = e0 + le < a.E ? e0 + le : a.E - 1;
      const T* gp = static_cast<const T*>(b.g_emb0) + e * S0 + sub * COLS;
#pragma unroll
      for (int cc = 0; cc < COLS; ++cc) g[ps][cc] = gp[cc];
    }
#pragma unroll
    for (int ps = 0; ps < 4; ++ps) {
      const int le = (pb + ps) * 32 + (tid >> 3);
      const int ty = sTy[le];  // ti | tj << 16, or -1 beyond the edge list
      const int tyc = ty >= 0 ? ty : 0;
      // columns [sub*COLS, +COLS) lie entirely in the center half (sub < 4) or the neighbor half of the embedding
      const T* te = sub < 4 ? sEmb + (tyc & 0xffff) * half + sub * COLS : sEmb + a.num_types * half + (tyc >> 16) * half + (sub - 4) * COLS;
      T part[B];
#pragma unroll
      for (int nb = 0; nb < B; ++nb) part[nb] = T(0);
#pragma unroll
      for (int cc = 0; cc < COLS; ++cc) {
        const T gv = ty >= 0 ? g[ps][cc] * te[cc] : T(0);
#pragma unroll
        for (int nb = 0; nb < B; ++nb) part[nb] += gv * w[cc][nb];
      }
      T r;
      if constexpr (sizeof(T) == 4) {
        float y[4], z[2];
        {
          const bool hi = sub & 4;
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const float send = hi ? part[k] : part[k + 4], keep = hi ? part[k + 4] : part[k];
            y[k] = keep + dpp_move<kDppHalfMirror>(send);
          }
        }
        {
          const bool hi = sub & 2;
#pragma unroll
          for (int k = 0; k < 2; ++k) {
            const float send = hi ? y[k] : y[k + 2], keep = hi ? y[k + 2] : y[k];
            z[k] = keep + dpp_move<kDppQuad2301>(send);
          }
        }
        {
          const bool hi = sub & 1;
          const float send = hi ? z[0] : z[1], keep = hi ? z[1] : z[0];
          r = keep + dpp_move<kDppQuad1032>(send);
        }
      } else {
        T y[4], z[2];
        {
          const bool hi = sub & 4;
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const T recv = __shfl_xor(hi ? part[k] : part[k + 4], 4);
            y[k] = (hi ? part[k + 4] : part[k]) + recv;
          }
        }
        {
          const bool hi = sub & 2;
#pragma unroll
          for (int k = 0; k < 2; ++k) {
            const T recv = __shfl_xor(hi ? y[k] : y[k + 2], 2);
            z[k] = (hi ? y[k + 2] : y[k]) + recv;
          }
        }
        {
          const bool hi = sub & 1;
          const T recv = __shfl_xor(hi ? z[0] : z[1], 1);
          r = (hi ? z[1] : z[0]) + recv;
        }
      }
      sT[le * (B + 1) + sub] = r;  // lane `sub` ends up with Bessel index sub = 4*bit2 + 2*bit1 + bit0
    }
  }
}

// S0 = 128 (C5): 16 lanes per edge -- lanes 0-7 the center half of the embedding, 8-15 the neighbor half, 8 consecutive columns each;
// a wave reads 4 whole 128-column rows per pass.  The two halves are added first, then the 8 partial sums are combined and
// scattered over the first 8 lanes as above.  (the general path below walks every row with 8-byte accesses from 8 lanes at a
// time: 2.46 ms at C5, 1.5 TB/s)
template <typename T>
__device__ __forceinline__ void edge_bwd_phase1_b16(const EdgeBwdArgs& b, int64_t e0, int tid, const int* sTy, const T* sEmb, T* sT) {
  const EdgeGeomArgs& a = b.g;
  constexpr int S0 = 128, half = 64, B = 8, COLS = 8;
  const int sub16 = tid & 15, sub = tid & 7;
  T w[COLS][B];
  {
    const T* wb = static_cast<const T*>(a.basis_w);  // [B][S0]
#pragma unroll
    for (int cc = 0; cc < COLS; ++cc)
#pragma unroll
      for (int nb = 0; nb < B; ++nb) w[cc][nb] = wb[nb * S0 + sub16 * COLS + cc];
  }
  for (int pb = 0; pb < 16; pb += 4) {
    T g[4][COLS];
#pragma unroll
    for (int ps = 0; ps < 4; ++ps) {
      const int le = (pb + ps) * 16 + (tid >> 4);
      const int64_t e = e0 + le < a.E ? e0 + le : a.E - 1;
      const T* gp = static_cast<const T*>(b.g_emb0) + e * S0 + sub16 * COLS;
#pragma unroll
      for (int cc = 0; cc < COLS; ++cc) g[ps][cc] = gp[cc];
    }
#pragma unroll
    for (int ps = 0; ps < 4; ++ps) {
      const int le = (pb + ps) * 16 + (tid >> 4);
      const int ty = sTy[le];  // ti | tj << 16, or -1 beyond the edge list
      const int tyc = ty >= 0 ? ty : 0;
      const T* te = sub16 < 8 ? sEmb + (tyc & 0xffff) * half + sub * COLS : sEmb + a.num_types * half + (tyc >> 16) * half + sub * COLS;
      T part[B];
#pragma unroll
      for (int nb = 0; nb < B; ++nb) part[nb] = T(0);
#pragma unroll
      for (int cc = 0; cc < COLS; ++cc) {
        const T gv = ty >= 0 ? g[ps][cc] * te[cc] : T(0);
#pragma unroll
        for (int nb = 0; nb < B; ++nb) part[nb] += gv * w[cc][nb];
      }
#pragma unroll
      for (int nb = 0; nb < B; ++nb) part[nb] += __shfl_xor(part[nb], 8);  // center half + neighbor half
      T y[4], z[2], r;
      {
        const bool hi = sub & 4;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const T recv = __shfl_xor(hi ? part[k] : part[k + 4], 4);
          y[k] = (hi ? part[k + 4] : part[k]) + recv;
        }
      }
      {
        const bool hi = sub & 2;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const T recv = __shfl_xor(hi ? y[k] : y[k + 2], 2);
          z[k] = (hi ? y[k + 2] : y[k]) + recv;
        }
      }
      {
        const bool hi = sub & 1;
        const T recv = __shfl_xor(hi ? z[0] : z[1], 1);
        r = (hi ? z[1] : z[0]) + recv;
      }
      if (sub16 < 8) sT[le * (B + 1) + sub] = r;
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void edge_backward_kernel(EdgeBwdArgs b) {
  const EdgeGeomArgs& a = b.g;
  const int B = a.num_bessels, S0 = a.S0, D = (a.l_max + 1) * (a.l_max + 1), Tn = a.num_types;
  T* sT = reinterpret_cast<T*>(aa_smem);          // [256][B+1]  dE/d(bessel_n * cutoff)
  T* sWb = sT + 256 * (B + 1);                    // [B][S0] (general path only)
  T* sEmb = sWb + size_t(S0) * kMaxBessel;        // [2][Tn][S0/2] center / neighbor type embeddings
  int* sTy = reinterpret_cast<int*>(sEmb + size_t(Tn) * S0);  // [256] ti | tj << 16
  const int tid = threadIdx.x;
  const int64_t e0 = int64_t(blockIdx.x) * 256;
  const int half = S0 / 2;
  const T* cemb = static_cast<const T*>(a.center_embed);
  const T* nemb = static_cast<const T*>(a.neighbor_embed);
  const bool spline = a.embed_kind == 1;
  const bool fast = !spline && (S0 == 16 || S0 == 32 || S0 == 64 || S0 == 128) && B == 8 && Tn < 32768;
  // this lane's own edge (used again in phase 2)
  const int64_t e = e0 + tid;
  int ci = -1, cj = -1, ti = 0, tj = 0;  // ci: center of this lane's edge (-1: no edge)
  if (e < a.E) {
    ci = a.center[e];
    cj = a.nbr[e];
    ti = a.types[ci];
    tj = a.types[cj];
  }
  if (fast) {
    for (int idx = tid; idx < Tn * half; idx += 256) {
      sEmb[idx] = cemb[idx];
      sEmb[Tn * half + idx] = nemb[idx];
    }
    sTy[tid] = ci >= 0 ? (ti | (tj << 16)) : -1;
  } else if (!spline) {
    for (int idx = tid; idx < B * S0; idx += 256) sWb[idx] = static_cast<const T*>(a.basis_w)[idx];
  }
  __syncthreads();
  // phase 1: t[le][n] = sum_c g_emb0[e][c] * type_embed[c] * Wb[n][c]   (or already done by the producer: t_in)
  if (b.t_in) {
    if (e < a.E)
      for (int nb = 0; nb < B; ++nb) sT[tid * (B + 1) + nb] = static_cast<const T*>(b.t_in)[e * B + nb];
  } else if (fast) {
    // 8 lanes per edge, each owning S0/8 consecutive columns: a wave reads 8 whole rows per pass (coalesced);
    // the B partial sums are then combined across the 8 lanes
    if (S0 == 128)
      edge_bwd_phase1_b16<T>(b, e0, tid, sTy, sEmb, sT);
    else if (S0 == 64)
      edge_bwd_phase1_b8<T, 8>(b, e0, tid, sTy, sEmb, sT);
    else if (S0 == 32)
      edge_bwd_phase1_b8<T, 4>(b, e0, tid, sTy, sEmb, sT);
    else
      edge_bwd_phase1_b8<T, 2>(b, e0, tid, sTy, sEmb, sT);
  } else if (spline) {
    // t[le][s] = sum_c g_emb0[e][c] * W[class(e)][c][s]
    const T* tab = static_cast<const T*>(a.emb_tab);
    for (int idx = tid; idx < 256 * B; idx += 256) {
      int le = idx / B, nb = idx % B;
      int64_t el = e0 + le;
      T acc = T(0);
      if (el < a.E) {
        const int cls = a.types[a.center[el]] * Tn + a.types[a.nbr[el]];
        const T* g = static_cast<const T*>(b.g_emb0) + el * S0;
        const T* tc = tab + (size_t(cls) * B + nb) * S0;
        for (int c = 0; c < S0; ++c) acc += g[c] * tc[c];
      }
      sT[le * (B + 1) + nb] = acc;
    }
  } else {
    for (int idx = tid; idx < 256 * B; idx += 256) {
      int le = idx / B, nb = idx % B;
      int64_t el = e0 + le;
      T acc = T(0);
      if (el < a.E) {
        int t_i = a.types[a.center[el]], t_j = a.types[a.nbr[el]];
        const T* g = static_cast<const T*>(b.g_emb0) + el * S0;
        for (int c = 0; c < S0; ++c) {
          T te = c < half ? cemb[t_i * half + c] : nemb[t_j * half + (c - half)];
          acc += g[c] * te * sWb[nb * S0 + c];
        }
      }
      sT[le * (B + 1) + nb] = acc;
    }
  }
  __syncthreads();
  // phase 2: per edge chain rule to the edge vector, then scatter to both atoms
  T fx = T(0), fy = T(0), fz = T(0);
  if (e < a.E) {
    const int j = cj;
    typedef T V4 __attribute__((ext_vector_type(4)));
    const V4 vec = *reinterpret_cast<const V4*>(static_cast<const T*>(a.vec) + 4 * e);
    T nx = vec[0], ny = vec[1], nz = vec[2], r = vec[3];
    T recip = static_cast<const T*>(a.rmax_recip)[ti * Tn + tj];
    T x = r * recip;
    T dEdx = T(0);
    if (spline) {
      for (int nb = 0; nb < B; ++nb) {
        T bv, dbv;
        spline_basis_and_grad<T>(x, nb, B, a.spline_span, bv, dbv);
        dEdx += sT[tid * (B + 1) + nb] * dbv;
      }
    } else {
      T f, df;
      cutoff_and_grad<T>(x, T(a.poly_p), f, df);
      for (int nb = 0; nb < B; ++nb) {
        T w = static_cast<const T*>(a.bessel_w)[nb];
        T s = aa_sin(w * x), c = aa_cos(w * x);
        T bv = s / x;
        T dbv = (w * c * x - s) / (x * x);
        dEdx += sT[tid * (B + 1) + nb] * (dbv * f + bv * df);
      }
    }
    T dEdr = dEdx * recip;
    // (staging the block's dE/dY rows through LDS for coalesced reads measured slower here: 191 vs 169 us at C4 --
    //  the per-lane strided loads overlap with phase 1, the staged copy does not)
    T gY[16];
    const T* gsh = static_cast<const T*>(b.g_sh) + e * D;
    for (int m = 0; m < D; ++m) gY[m] = gsh[m];
    for (int sl = 1; sl < b.num_gsh; ++sl) {
      const T* g2 = gsh + int64_t(sl) * a.E * D;
      for (int m = 0; m < D; ++m) gY[m] += g2[m];
    }
    T gx, gy, gz;
    sh_grad<T>(a.l_max, nx, ny, nz, gY, gx, gy, gz);
    T dot = gx * nx + gy * ny + gz * nz;
    T inv = T(1) / r;
    T dx = dEdr * nx + (gx - dot * nx) * inv;
    T dy = dEdr * ny + (gy - dot * ny) * inv;
    T dz = dEdr * nz + (gz - dot * nz) * inv;
    // r_ij = pos_j - pos_i : dE/dpos_j = +d, dE/dpos_i = -d ; F = -dE/dpos
    fx = dx;
    fy = dy;
    fz = dz;
    if (b.dvec) {
      *reinterpret_cast<V4*>(static_cast<T*>(b.dvec) + 4 * e) = V4{dx, dy, dz, T(0)};
    }
    if (!b.gather) {
      T* F = static_cast<T*>(b.forces);
      atomicAdd(&F[3 * int64_t(j)], -dx);
      atomicAdd(&F[3 * int64_t(j) + 1], -dy);
      atomicAdd(&F[3 * int64_t(j) + 2], -dz);
    }
  }
  if (b.gather) return;  // force_gather_kernel sums them per atom
  // center-atom contributions: the edges are sorted by center, so the lanes of a wave hold runs of equal centers;
  // a segmented shuffle sum leaves one atomic per run instead of one per edge (same-address atomics serialise)
  {
    const int lane = tid & 63;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int ko = __shfl_down(ci, off);
      const T ox = __shfl_down(fx, off), oy = __shfl_down(fy, off), oz = __shfl_down(fz, off);
      if (lane + off < 64 && ko == ci) {
        fx += ox;
        fy += oy;
        fz += oz;
      }
    }
    const int kp = __shfl_up(ci, 1);
    if (ci >= 0 && (lane == 0 || kp != ci)) {
      T* F = static_cast<T*>(b.forces);
      atomicAdd(&F[3 * int64_t(ci)], fx);
      atomicAdd(&F[3 * int64_t(ci) + 1], fy);
      atomicAdd(&F[3 * int64_t(ci) + 2], fz);
    }
  }
}

// 8 lanes per atom walk its own segment (+d) and its transposed segment (-d) in a fixed order
template <typename T>
__global__ __launch_bounds__(256) void force_gather_kernel(ForceGatherArgs a) {
  const int sub = threadIdx.x & 7;
  const int64_t n = (int64_t(blockIdx.x) * 256 + threadIdx.x) >> 3;
  T fx = T(0), fy = T(0), fz = T(0);
  if (n < a.N) {
    const T* dv = static_cast<const T*>(a.dvec);
    for (int e = a.rowptr[n] + sub; e < a.rowptr[n + 1]; e += 8) {
      fx += dv[4 * int64_t(e)];
      fy += dv[4 * int64_t(e) + 1];
      fz += dv[4 * int64_t(e) + 2];
    }
    for (int k = a.t_rowptr[n] + sub; k < a.t_rowptr[n + 1]; k += 8) {
      const int64_t e = a.t_perm[k];
      fx -= dv[4 * e];
      fy -= dv[4 * e + 1];
      fz -= dv[4 * e + 2];
    }
  }
#pragma unroll
  for (int m = 4; m >= 1; m >>= 1) {
    fx += __shfl_xor(fx, m);
    fy += __shfl_xor(fy, m);
    fz += __shfl_xor(fz, m);
  }
  if (n < a.N && sub == 0) {
    T* F = static_cast<T*>(a.forces) + 3 * n;
    F[0] = fx;
    F[1] = fy;
    F[2] = fz;
  }
}

// one wave per atom: E_i = scale_t * factor * sum_{e in seg(i)} sum_c act(h[e,c]) w[c] + shift_t
template <typename T>
__global__ __launch_bounds__(256) void readout_reduce_kernel(ReadoutArgs a) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int64_t n = int64_t(blockIdx.x) * 4 + wv;
  T acc = T(0);
  if (n < a.N) {
    int beg = a.rowptr[n], end = a.rowptr[n + 1];
    const T* w = static_cast<const T*>(a.w);
    if (a.edge_sum) {
      for (int s = beg + lane; s < end; s += 64) acc += static_cast<const T*>(a.edge_sum)[s];
    } else
    for (int c = lane; c < a.H; c += 64) {
      T wc = w[c];
      if (a.g_h) {
        // forces requested: d E / d h[e, c] = factor * scale(type of the center) * w[c] * act'(h) depends on nothing but what this
        // loop reads -- written here, so that no second kernel streams the hidden layer again (readout_backward_kernel)
        T gf = T(a.factor) * wc;
        if (a.scales) gf *= static_cast<const T*>(a.scales)[a.types[n]];
        T* gh = static_cast<T*>(a.g_h) + c;
        for (int s = beg; s < end; ++s) {
          T h = static_cast<const T*>(a.h)[int64_t(s) * a.ld + c];
          if (a.act && a.act_kind == AA_ACT_SILU) {
            const T sg = sigmoid_(h);  // (one exponential for the value and the slope)
            acc += h * sg * wc;
            gh[int64_t(s) * a.H] = gf * sg * (T(1) + h * (T(1) - sg));
          } else {
            acc += (a.act ? act_apply(a.act_kind, h) : h) * wc;
            gh[int64_t(s) * a.H] = a.act ? gf * act_grad(a.act_kind, h) : gf;
          }
        }
        continue;
      }
      for (int s = beg; s < end; ++s) {
        T h = static_cast<const T*>(a.h)[int64_t(s) * a.ld + c];
        acc += (a.act ? act_apply(a.act_kind, h) : h) * wc;
      }
    }
  }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) acc += __shfl_xor(acc, m);
  if (n < a.N && lane == 0) {
    T en = acc * T(a.factor);
    int t = a.types[n];
    if (a.scales) en *= static_cast<const T*>(a.scales)[t];
    if (a.shifts) en += static_cast<const T*>(a.shifts)[t];
    static_cast<T*>(a.atom_energy)[n] = en;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void readout_backward_kernel(ReadoutArgs a) {
  const int64_t total = a.E * a.H;
  for (int64_t idx = int64_t(blockIdx.x) * 256 + threadIdx.x; idx < total; idx += int64_t(gridDim.x) * 256) {
    int64_t e = idx / a.H;
    int c = int(idx % a.H);
    T g = T(a.factor) * static_cast<const T*>(a.w)[c];
    if (a.scales) g *= static_cast<const T*>(a.scales)[a.types[a.center[e]]];
    if (a.act) g *= act_grad(a.act_kind, static_cast<const T*>(a.h)[e * a.ld + c]);
    static_cast<T*>(a.g_h)[e * a.H + c] = g;
  }
}

template <typename T>
int launch_edge_prologue(const EdgeGeomArgs& a, hipStream_t stream) {
  if (a.E == 0) return AA_OK;
  AA_REQUIRE(a.num_bessels <= kMaxBessel && a.l_max >= 1 && a.l_max <= 3 && a.S0 % 2 == 0, "prologue: unsupported sizes");
  const int D = (a.l_max + 1) * (a.l_max + 1);
  const size_t base = sizeof(T) * (256 * size_t(prologue_ldb(a.num_bessels)) + size_t(a.num_bessels) * a.S0) + sizeof(int) * 512;
  const size_t staged = base + sizeof(T) * 256 * size_t(D);
  const bool stage = staged <= 40 * 1024;  // (four workgroups per CU stay resident)
  const size_t smem = stage ? staged : base;
  if (smem > 64 * 1024) return fail(AA_ERR_INVALID, "prologue: LDS tables too large");
  if (stage)
    hipLaunchKernelGGL((edge_prologue_kernel<T, true>), dim3((unsigned)((a.E + 255) / 256)), dim3(256), smem, stream, a);
  else
    hipLaunchKernelGGL((edge_prologue_kernel<T, false>), dim3((unsigned)((a.E + 255) / 256)), dim3(256), smem, stream, a);
  AA_CHECK_HIP(hipGetLastError());
  return AA_OK;
}

template <typename T>
int launch_edge_backward(const EdgeBwdArgs& b, hipStream_t stream) {
  if (b.g.E == 0) return AA_OK;
  size_t smem = sizeof(T) * (256 * size_t(b.g.num_bessels + 1) + size_t(kMaxBessel) * b.g.S0 + size_t(b.g.num_types) * b.g.S0) + sizeof(int) * 256;
  if (smem > 160 * 1024) return fail(AA_ERR_INVALID, "edge_backward: too many types / embedding columns for the LDS tables");
  if (smem > 64 * 1024)
    AA_CHECK_HIP(hipFuncSetAttribute((const void*)edge_backward_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, int(smem)));
  hipLaunchKernelGGL(edge_backward_kernel<T>, dim3((unsigned)((b.g.E + 255) / 256)), dim3(256), smem, stream, b);
  AA_CHECK_HIP(hipGetLastError());
  return AA_OK;
}

// strain derivative: per-block partial sums in double (fixed order), then one block adds the partials
template <typename T>
__global__ __launch_bounds__(256) void virial_partial_kernel(VirialArgs a) {
  double* sP = reinterpret_cast<double*>(aa_smem);  // [256][9]
  double w[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) w[k] = 0.0;
  const T* dv = static_cast<const T*>(a.dvec);
  const T* vc = static_cast<const T*>(a.vec);
  for (int64_t e = int64_t(blockIdx.x) * 256 + threadIdx.x; e < a.E; e += int64_t(gridDim.x) * 256) {
    const double r = double(vc[4 * e + 3]);
    const double rx = double(vc[4 * e]) * r, ry = double(vc[4 * e + 1]) * r, rz = double(vc[4 * e + 2]) * r;
    const double dx = double(dv[4 * e]), dy = double(dv[4 * e + 1]), dz = double(dv[4 * e + 2]);
    w[0] += dx * rx; w[1] += dx * ry; w[2] += dx * rz;
    w[3] += dy * rx; w[4] += dy * ry; w[5] += dy * rz;
    w[6] += dz * rx; w[7] += dz * ry; w[8] += dz * rz;
  }
#pragma unroll
  for (int k = 0; k < 9; ++k) sP[threadIdx.x * 9 + k] = w[k];
  __syncthreads();
  for (int st = 128; st >= 1; st >>= 1) {
    if (int(threadIdx.x) < st)
      for (int k = 0; k < 9; ++k) sP[threadIdx.x * 9 + k] += sP[(threadIdx.x + st) * 9 + k];
    __syncthreads();
  }
  if (threadIdx.x < 9) a.partial[blockIdx.x * 9 + threadIdx.x] = sP[threadIdx.x];
}
template <typename T>
__global__ __launch_bounds__(64) void virial_final_kernel(VirialArgs a, int nblocks) {
  if (threadIdx.x < 9) {
    double s = 0.0;
    for (int b2 = 0; b2 < nblocks; ++b2) s += a.partial[b2 * 9 + threadIdx.x];
    static_cast<T*>(a.out)[threadIdx.x] = T(s);
  }
}
template <typename T>
int launch_virial(const VirialArgs& a, hipStream_t stream) {
  const int nb = int(std::min<int64_t>(kVirialBlocks, std::max<int64_t>(1, (a.E + 255) / 256)));
  hipLaunchKernelGGL(virial_partial_kernel<T>, dim3(nb), dim3(256), sizeof(double) * 256 * 9, stream, a);
  hipLaunchKernelGGL(virial_final_kernel<T>, dim3(1), dim3(64), 0, stream, a, nb);
  AA_CHECK_HIP(hipGetLastError());
  return AA_OK;
}

template <typename T>
int launch_force_gather(const ForceGatherArgs& a, hipStream_t stream) {
  if (a.N == 0) return AA_OK;
  hipLaunchKernelGGL(force_gather_kernel<T>, dim3((unsigned)((a.N * 8 + 255) / 256)), dim3(256), 0, stream, a);
  AA_CHECK_HIP(hipGetLastError());
  return AA_OK;
}

// Runs LAST in a step that carries an atom-block hint.  Edges whose center lies outside [a0, a1) were skipped by every
// per-atom kernel, so the step's outputs are not the model's: besides raising the status word (reported by the next call /
// aa_model_check) the outputs of THIS step are overwritten with NaN -- a host that integrates the forces before it looks at
// the status sees the failure in the same step.  One small block; the fill is the failure path only.
__global__ void graph_hint_check_kernel(const int32_t* rowptr, int64_t N, int64_t a0, int64_t a1, int32_t* status, void* atom_energy,
                                        void* forces, int esize) {
  if (rowptr[a0] == rowptr[0] && rowptr[a1] == rowptr[N]) return;
  if (threadIdx.x == 0) *reinterpret_cast<volatile int32_t*>(status) = -2;
  for (int64_t i = threadIdx.x; i < 4 * N; i += blockDim.x) {
    void* dst = i < N ? atom_energy : forces;
    const int64_t k = i < N ? i : i - N;
    if (!dst) continue;
    if (esize == 4) static_cast<float*>(dst)[k] = __builtin_nanf("");
    else static_cast<double*>(dst)[k] = __builtin_nan("");
  }
}
int launch_graph_hint_check(const int32_t* rowptr, int64_t N, int64_t a0, int64_t a1, int32_t* status, void* atom_energy, void* forces,
                            int esize, hipStream_t stream) {
  hipLaunchKernelGGL(graph_hint_check_kernel, dim3(1), dim3(256), 0, stream, rowptr, N, a0, a1, status, atom_energy, forces, esize);
  AA_CHECK_HIP(hipGetLastError());
  return AA_OK;
}

template <typename T>
int launch_readout_reduce(const ReadoutArgs& a, hipStream_t stream) {
  if (a.N == 0) return AA_OK;
  hipLaunchKernelGGL(readout_reduce_kernel<T>, dim3((unsigned)((a.N + 3) / 4)), dim3(256), 0, stream, a);
  AA_CHECK_HIP(hipGetLastError());
  return AA_OK;
}

template <typename T>
int launch_readout_backward(const ReadoutArgs& a, hipStream_t stream) {
  if (a.E == 0) return AA_OK;
  int64_t total = a.E * a.H;
  unsigned blocks = (unsigned)std::min<int64_t>((total + 255) / 256, 256 * 8);
  hipLaunchKernelGGL(readout_backward_kernel<T>, dim3(blocks), dim3(256), 0, stream, a);
  AA_CHECK_HIP(hipGetLastError());
  return AA_OK;
}

#define AA_INST(T)                                                              \
  template int launch_edge_prologue<T>(const EdgeGeomArgs&, hipStream_t);      \
  template int launch_edge_backward<T>(const EdgeBwdArgs&, hipStream_t);       \
  template int launch_force_gather<T>(const ForceGatherArgs&, hipStream_t);    \
  template int launch_virial<T>(const VirialArgs&, hipStream_t);               \
  template int launch_readout_reduce<T>(const ReadoutArgs&, hipStream_t);      \
  template int launch_readout_backward<T>(const ReadoutArgs&, hipStream_t);
AA_INST(float)
AA_INST(double)

}  // namespace aa
