// Dense per-edge scalar-MLP GEMMs of the Allegro hot path (gfx950).
//
// Computes, for one linear layer of a nequip ScalarMLPFunction (reference call sites
// allegro/nn/_allegro.py:251,278; tensorembed.py:89; allegro_models.py:173-183,231-241):
//     C (=|+=) ( act(A)[M,K] @ B[K,N] ) (* silu'(Z))
// A and C are column-segmented views so that the reference's torch.cat (dense-net concat,
// _allegro.py:278,300) and torch.narrow (:253-258,284-294) never materialise.  The same kernel serves
// the reverse pass (B = W^T, Z = forward pre-activation, accumulate into shared dense-net columns).
//
// f32: v_mfma_f32_32x32x2_f32 (exact fp32 FMA chain, 157 TF peak) -- the only place MFMA is used,
// because only the scalar MLPs are true GEMMs.  f64 (and the A/B check path): LDS-tiled VALU kernel.
#include "aa_common.h"
#include "aa_mfma.h"

namespace aa {


template <typename T>
__device__ __forceinline__ T seg_load(const SegList& sl, int64_t row, int col) {
  int c = col;
#pragma unroll
  for (int s = 0; s < 3; ++s) {
    if (s < sl.count) {
      if (c < sl.s[s].n) return static_cast<const T*>(sl.s[s].p)[row * sl.s[s].ld + c];
      c -= sl.s[s].n;
    }
  }
  return T(0);
}

// is column k of A activated on load?  (wave-uniform wherever k is a chunk base: the range is 32-granular)
__device__ __forceinline__ bool a_act(const GemmArgs& g, int k) { return g.act_a && (g.act_hi == 0 || (k >= g.act_lo && k < g.act_hi)); }

template <typename T>
__device__ __forceinline__ void seg_store(const GemmArgs& g, int64_t row, int col, T v) {
  int c = col;
#pragma unroll
  for (int s = 0; s < 3; ++s) {
    if (s < g.c.count) {
      if (c < g.c.s[s].n) {
        if (g.has_add) v += static_cast<const T*>(g.add.s[s].p)[row * g.add.s[s].ld + c];
        if (g.has_z) {
          T z = static_cast<const T*>(g.z.s[s].p)[row * g.z.s[s].ld + c];
          v *= act_grad(g.act_kind, z);
        }
        T* p = static_cast<T*>(g.c.s[s].p) + row * g.c.s[s].ld + c;
        if (g.c_accum[s]) v += *p;
        *p = v;
        return;
      }
      c -= g.c.s[s].n;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// fp32 MFMA kernel: block tile 128(M) x 64(N), K step 32; 4 waves, each 32 rows x 64 cols
// (two 32x32 accumulators); A tile stored k-major in LDS (+1 pad) so MFMA operand reads are
// conflict-free ds_read_b32.
// ---------------------------------------------------------------------------------------------
constexpr int GM_BM = 128, GM_BN = 64, GM_BK = 32, GM_LDA = GM_BM + 1;

__global__ __launch_bounds__(256) void gemm_mfma_f32_kernel(GemmArgs g) {
  float* As = reinterpret_cast<float*>(aa_smem);  // [BK][LDA]
  float* Bs = As + GM_BK * GM_LDA;                // [BK][BN]
  const int tid = threadIdx.x;
  const int lane = tid & 63, wv = tid >> 6;
  const int64_t m0 = int64_t(blockIdx.x) * GM_BM;
  const int n0 = blockIdx.y * GM_BN;
  const float* B = static_cast<const float*>(g.B);

  v16f acc0, acc1;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    acc0[r] = 0.f;
    acc1[r] = 0.f;
  }

  for (int k0 = 0; k0 < g.K; k0 += GM_BK) {
    // stage A (with optional activation) and B
    for (int idx = tid; idx < GM_BM * GM_BK; idx += 256) {
      int m = idx / GM_BK, k = idx % GM_BK;
      int64_t gm = m0 + m;
      int gk = k0 + k;
      float v = 0.f;
      if (gm < g.M && gk < g.K) {
        v = seg_load<float>(g.a, gm, gk);
        if (a_act(g, gk)) v = silu(v);
      }
      As[k * GM_LDA + m] = v;
    }
    for (int idx = tid; idx < GM_BK * GM_BN; idx += 256) {
      int k = idx / GM_BN, j = idx % GM_BN;
      int gk = k0 + k, gn = n0 + j;
      Bs[idx] = (gk < g.K && gn < g.N) ? B[int64_t(gk) * g.N + gn] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < GM_BK; kk += 2) {
      int kr = kk + (lane >> 5);
      float a = As[kr * GM_LDA + wv * 32 + (lane & 31)];
      float b0 = Bs[kr * GM_BN + (lane & 31)];
      float b1 = Bs[kr * GM_BN + 32 + (lane & 31)];
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b0, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b1, acc1, 0, 0, 0);
    }
    __syncthreads();
  }
  // epilogue: C/D fragment layout col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    int64_t gm = m0 + wv * 32 + row;
    if (gm < g.M) {
      int gn = n0 + (lane & 31);
      if (gn < g.N) seg_store<float>(g, gm, gn, acc0[r]);
      gn += 32;
      if (gn < g.N) seg_store<float>(g, gm, gn, acc1[r]);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// generic VALU kernel (fp64; fp32 cross-check): 64x64 tile, 4x4 micro-tile per thread
// ---------------------------------------------------------------------------------------------
constexpr int GV_BM = 64, GV_BN = 64, GV_BK = 16, GV_LDA = GV_BM + 1;

template <typename T>
__global__ __launch_bounds__(256) void gemm_valu_kernel(GemmArgs g) {
  T* As = reinterpret_cast<T*>(aa_smem);  // [BK][LDA]
  T* Bs = As + GV_BK * GV_LDA;            // [BK][BN]
  const int tid = threadIdx.x;
  const int tx = tid & 15, ty = tid >> 4;
  const int64_t m0 = int64_t(blockIdx.x) * GV_BM;
  const int n0 = blockIdx.y * GV_BN;
  const T* B = static_cast<const T*>(g.B);
  T acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = T(0);

  for (int k0 = 0; k0 < g.K; k0 += GV_BK) {
    for (int idx = tid; idx < GV_BM * GV_BK; idx += 256) {
      int m = idx / GV_BK, k = idx % GV_BK;
      int64_t gm = m0 + m;
      int gk = k0 + k;
      T v = T(0);
      if (gm < g.M && gk < g.K) {
        v = seg_load<T>(g.a, gm, gk);
        if (a_act(g, gk)) v = act_apply(g.act_kind, v);
      }
      As[k * GV_LDA + m] = v;
    }
    for (int idx = tid; idx < GV_BK * GV_BN; idx += 256) {
      int k = idx / GV_BN, j = idx % GV_BN;
      int gk = k0 + k, gn = n0 + j;
      Bs[idx] = (gk < g.K && gn < g.N) ? B[int64_t(gk) * g.N + gn] : T(0);
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < GV_BK; ++k) {
      T a[4], b[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) a[i] = As[k * GV_LDA + ty * 4 + i];
#pragma unroll
      for (int j = 0; j < 4; ++j) b[j] = Bs[k * GV_BN + tx * 4 + j];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] += a[i] * b[j];
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int64_t gm = m0 + ty * 4 + i;
    if (gm >= g.M) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      int gn = n0 + tx * 4 + j;
      if (gn < g.N) seg_store<T>(g, gm, gn, acc[i][j]);
    }
  }
}

// fp64 GEMM on the f64 matrix cores (v_mfma_f64_16x16x4_f64): same 64x64x16 LDS tiling and generic segmented
// loads / epilogue as gemm_valu_kernel, the inner product on MFMA.  Each of the four waves owns a 32x32 sub-tile
// (2x2 MFMA tiles).  Operand layout: A[i = lane&15][k = lane>>4], B[k = lane>>4][j = lane&15]; result
// C[i = 4*r + (lane>>4)][j = lane&15], r = 0..3 (probed on hardware: unlike the f32 16x16 layout, register r holds
// rows 4r..4r+3 across the four 16-lane groups).
typedef double v4d __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void gemm_mfma_f64_kernel(GemmArgs g) {
  double* As = reinterpret_cast<double*>(aa_smem);  // [BK][LDA]
  double* Bs = As + GV_BK * GV_LDA;                 // [BK][BN]
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int wr = wv >> 1, wc = wv & 1;
  const int64_t m0 = int64_t(blockIdx.x) * GV_BM;
  const int n0 = blockIdx.y * GV_BN;
  const double* B = static_cast<const double*>(g.B);
  v4d acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = v4d{0.0, 0.0, 0.0, 0.0};
  for (int k0 = 0; k0 < g.K; k0 += GV_BK) {
    for (int idx = tid; idx < GV_BM * GV_BK; idx += 256) {
      const int m = idx / GV_BK, k = idx % GV_BK;
      const int64_t gm = m0 + m;
      const int gk = k0 + k;
      double v = 0.0;
      if (gm < g.M && gk < g.K) {
        v = seg_load<double>(g.a, gm, gk);
        if (a_act(g, gk)) v = silu(v);
      }
      As[k * GV_LDA + m] = v;
    }
    for (int idx = tid; idx < GV_BK * GV_BN; idx += 256) {
      const int k = idx / GV_BN, j = idx % GV_BN;
      const int gk = k0 + k, gn = n0 + j;
      Bs[idx] = (gk < g.K && gn < g.N) ? B[int64_t(gk) * g.N + gn] : 0.0;
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < GV_BK; kk += 4) {
      const int kr = kk + (lane >> 4);
      double a[2], b[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) a[i] = As[kr * GV_LDA + wr * 32 + i * 16 + (lane & 15)];
#pragma unroll
      for (int j = 0; j < 2; ++j) b[j] = Bs[kr * GV_BN + wc * 32 + j * 16 + (lane & 15)];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int gn = n0 + wc * 32 + j * 16 + (lane & 15);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int64_t gm = m0 + wr * 32 + i * 16 + 4 * r + (lane >> 4);
        if (gm < g.M && gn < g.N) seg_store<double>(g, gm, gn, acc[i][j][r]);
      }
    }
}

// Pipelined variant for the model's shapes (every A segment a multiple of 16 columns wide, 16-B aligned rows):
// Epilogue of one 16-column MFMA tile pair of a wave (v_mfma_f64_16x16x4 layout: register r of tile i = row 16 i + 4 r +
// lane / 16, column lane % 16): out = (acc [+ add]) [* silu'(z)] [+ C].  The operand loads of the eight elements are issued
// TOGETHER, ahead of the arithmetic -- element-by-element code waits out the full memory latency eight times per tile
// (each load sits behind a wave-uniform branch, which the compiler does not hoist loads across).  Also tried: the loads of
// tile j + 1 issued before the arithmetic and stores of tile j (two operand sets in flight) -- 3 % on the accumulator-
// resident kernel's z layers, but 57-150 spilled registers in the staged and the operand-resident kernel, whose K loops
// then run 25-50 % slower (profiles/archive/r03_n_stages_c5.log); not kept.
// NT: non-temporal result stores -- the operand-resident kernel's wide layers (128 -> 640 at C5: 8.8 GB of results nobody re-reads from
// a cache) gain 4 % (profiles/r06_v43_ab_c5_f64_epilogue_nt_stores.txt); the N <= 128 layers do not (+1 %) and keep plain stores.
template <bool NT = false>
__device__ __forceinline__ void f64_tile_epilogue(const GemmArgs& g, int t0, int64_t m_base, int li, int lg, const v4d& acc0, const v4d& acc1,
                                                  int64_t c_shift = 0) {
  if (t0 >= g.N) return;
  int cc = t0, si = -1;
#pragma unroll
  for (int s2 = 0; s2 < 3; ++s2) {
    if (s2 < g.c.count && si < 0) {
      if (cc < g.c.s[s2].n)
        si = s2;
      else
        cc -= g.c.s[s2].n;
    }
  }
  if (si < 0) return;
  double* cp = static_cast<double*>(g.c.s[si].p);
  if (!cp) return;
  cp += c_shift;
  const int ldc = g.c.s[si].ld, col = cc + li;
  const double* zp = g.has_z ? static_cast<const double*>(g.z.s[si].p) : nullptr;
  const double* ap = g.has_add ? static_cast<const double*>(g.add.s[si].p) : nullptr;
  const int ldz = g.has_z ? g.z.s[si].ld : 0, lda2 = g.has_add ? g.add.s[si].ld : 0;
  const bool accum = g.c_accum[si] != 0;
  int64_t gm[8], gl[8];  // row, row clamped for the loads
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    gm[e] = m_base + 16 * (e >> 2) + 4 * (e & 3) + lg;
    gl[e] = gm[e] < g.M ? gm[e] : g.M - 1;
  }
  double v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = (e >> 2) ? acc1[e & 3] : acc0[e & 3];
  if (ap) {
    double t[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) t[e] = ap[gl[e] * lda2 + col];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] += t[e];
  }
  if (zp) {
    double t[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) t[e] = zp[gl[e] * ldz + col];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] *= dsilu(t[e]);
  }
  if (accum) {
    double t[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) t[e] = cp[gl[e] * ldc + col];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] += t[e];
  }
#pragma unroll
  for (int e = 0; e < 8; ++e)
    if (gm[e] < g.M) {
      if constexpr (NT)
        __builtin_nontemporal_store(v[e], &cp[gm[e] * ldc + col]);
      else
        cp[gm[e] * ldc + col] = v[e];
    }
}

// 128 x 64 block tile, 16-deep steps; the next step's global loads are in flight while this step's 32 MFMAs per
// wave issue, LDS double-buffered (one barrier per step).  Each wave owns 32 rows x 64 columns (2 x 4 MFMA tiles).
constexpr int G6_BM = 128, G6_BN = 64, G6_BK = 16, G6_LDA = G6_BM + 4;
typedef double v2d __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(256, 3) void gemm_mfma_f64_pipe_kernel(GemmArgs g) {
  double* smem = reinterpret_cast<double*>(aa_smem);
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int64_t m0 = int64_t(blockIdx.x) * G6_BM;
  const double* B = static_cast<const double*>(g.B);
  // gridDim.y == 1: this workgroup walks ALL column tiles of its 128 rows, so the A rows come from HBM once and from
  // L2 afterwards (a 2-D grid re-reads A from HBM once per column tile: 8x at N = 512)
  const int n_tiles = gridDim.y == 1 ? (g.N + G6_BN - 1) / G6_BN : 1;
  for (int nt = 0; nt < n_tiles; ++nt) {
  const int n0 = (gridDim.y == 1 ? nt : int(blockIdx.y)) * G6_BN;
  auto As = [&](int b) { return smem + b * (G6_BK * G6_LDA + G6_BK * G6_BN); };
  auto Bs = [&](int b) { return As(b) + G6_BK * G6_LDA; };
  // staging roles: A: thread -> (row = tid >> 1, 8 consecutive k); B: thread -> (k = tid >> 4, 4 consecutive columns)
  const int ar = tid >> 1, ak = (tid & 1) * 8;
  const int64_t arow = m0 + ar < g.M ? m0 + ar : g.M - 1;
  const int bk = tid >> 4, bn = (tid & 15) * 4;
  v2d ra[4], rb[2];
  int ra_k0 = 0;  // first column of the operand tile held in ra (its activation is decided when it is stored)
  auto load_tiles = [&](int k0, int nb0) {
    ra_k0 = k0;
    // the segment holding columns [k0, k0+16): wave-uniform
    int c = k0;
    const double* base = nullptr;
    int ld = 0;
#pragma unroll
    for (int s2 = 0; s2 < 3; ++s2) {
      if (s2 < g.a.count && base == nullptr) {
        if (c < g.a.s[s2].n) {
          base = static_cast<const double*>(g.a.s[s2].p) + c;
          ld = g.a.s[s2].ld;
        } else {
          c -= g.a.s[s2].n;
        }
      }
    }
    const double* ap = base + arow * ld + ak;
#pragma unroll
    for (int q = 0; q < 4; ++q) ra[q] = (base && k0 + ak + 2 * q < g.K) ? *reinterpret_cast<const v2d*>(ap + 2 * q) : v2d{0.0, 0.0};
    const int gk = k0 + bk;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int gn = nb0 + bn + 2 * q;
      rb[q] = (gk < g.K && gn + 1 < g.N) ? *reinterpret_cast<const v2d*>(B + int64_t(gk) * g.N + gn)
                                          : v2d{(gk < g.K && gn < g.N) ? B[int64_t(gk) * g.N + gn] : 0.0, 0.0};
    }
  };
  auto store_tiles = [&](int b) {
    double* a_ = As(b);
    double* b_ = Bs(b);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      double x0 = ra[q][0], x1 = ra[q][1];
      if (a_act(g, ra_k0)) {
        x0 = silu(x0);
        x1 = silu(x1);
      }
      a_[(ak + 2 * q) * G6_LDA + ar] = x0;
      a_[(ak + 2 * q + 1) * G6_LDA + ar] = x1;
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) *reinterpret_cast<v2d*>(b_ + bk * G6_BN + bn + 2 * q) = rb[q];
  };
  v4d acc[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = v4d{0.0, 0.0, 0.0, 0.0};
  if (nt == 0) load_tiles(0, n0);  // (later column tiles: issued before the previous tile's epilogue, see below)
  store_tiles(0);
  __syncthreads();
  int buf = 0;
  for (int k0 = 0; k0 < g.K; k0 += G6_BK) {
    const bool more = k0 + G6_BK < g.K;
    if (more) load_tiles(k0 + G6_BK, n0);
    const double* a_ = As(buf);
    const double* b_ = Bs(buf);
#pragma unroll
    for (int kk = 0; kk < G6_BK; kk += 4) {
      const int kr = kk + (lane >> 4);
      double a[2], b[4];
#pragma unroll
      for (int i = 0; i < 2; ++i) a[i] = a_[kr * G6_LDA + wv * 32 + i * 16 + (lane & 15)];
#pragma unroll
      for (int j = 0; j < 4; ++j) b[j] = b_[kr * G6_BN + j * 16 + (lane & 15)];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    if (more) store_tiles(buf ^ 1);
    __syncthreads();
    buf ^= 1;
  }
  // the next column tile's first operands travel while this tile's epilogue runs
  if (nt + 1 < n_tiles) load_tiles(0, n0 + G6_BN);
  // epilogue: a 16-column tile never straddles a C segment here (widths are multiples of 16, checked by the
  // launcher), so the destination / z / add rows are resolved once per tile, wave-uniformly
#pragma unroll
  for (int j = 0; j < 4; ++j) f64_tile_epilogue(g, n0 + j * 16, m0 + wv * 32, lane & 15, lane >> 4, acc[0][j], acc[1][j]);
  }  // column tiles
}

// ---------------------------------------------------------------------------------------------
// fp64 linear layers, row-resident form: every wave owns 32 rows and the OPERAND ROWS NEVER TOUCH LDS.
// v_mfma_f64_16x16x4 takes A[i = lane & 15][k = lane >> 4]; with the k index permuted inside each 16-deep chunk (lane
// group g supplies k = 4 g + s at MFMA step s, the weights are read from LDS with the same mapping) lane (i, g) needs
// A[row i][16 c + 4 g .. + 3]: 32 contiguous bytes of its own row, fetched straight into VGPRs.  Each A element is read
// from HBM exactly once per layer:
//   CSTAT (N <= 128, any K): the wave holds its whole 32 x N output in accumulators (128 VGPRs) and streams A by chunks;
//   ASTAT (K <= 128, any N): the wave holds its 32 x K operand rows in registers (128 VGPRs) and walks the column
//                            tiles of 64.
// (the staged kernel above re-reads the 128-row A tile once per 64-column tile -- at N = 512 that is 8 reads, which no
// longer come from L2 once ~770 workgroups are resident: rocprofv3 FETCH_SIZE showed 2.5x the algorithmic read bytes)
// Only the weights go through LDS: a [16][NB] block per chunk, staged by the four waves together, double-buffered,
// one barrier per 32-64 MFMAs of 64 cycles each.
template <bool ASTAT>
__global__ __launch_bounds__(256, 2) void gemm_f64_rows_kernel(GemmArgs g) {
  constexpr int NB = ASTAT ? 64 : 128;  // columns per pass
  constexpr int JT = NB / 16;           // MFMA column tiles per pass
  constexpr int LDB = NB + 4;           // (rows 4 g + s of the four lane groups land in different banks)
  constexpr int SPT = NB / 16;          // doubles per thread of a staged [16][NB] block
  constexpr int KCMAX = ASTAT ? 8 : 1;  // resident operand chunks
  double* smem = reinterpret_cast<double*>(aa_smem);  // [2][16][LDB]
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int li = lane & 15, lg = lane >> 4;
  const int64_t m_base = (int64_t(blockIdx.x) * 4 + wv) * 32;
  // batched launch (GemmArgs::batch): this workgroup's problem is blockIdx.z -- its operand / result views are the stated ones
  // shifted by whole elements, its weights one matrix of the set.  (shifts applied where pointers are formed: a modified COPY of
  // the argument struct would live in scratch memory, its segment tables are indexed at run time)
  const int bz = g.batch > 1 ? int(blockIdx.z) : 0;
  const int64_t a_sh = bz * g.a_bs, c_sh = bz * g.c_bs;
  const double* B = static_cast<const double*>(g.B) + int64_t((g.bsel4 >> (4 * bz)) & 15ull) * g.b_bs;
  const int KC = g.K / 16;
  int64_t arow[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int64_t r = m_base + 16 * i + li;
    arow[i] = r < g.M ? r : g.M - 1;
  }
  const int sr = tid >> 4, sc = (tid & 15) * SPT;
  v2d rb[SPT / 2];
  // Branch-free fetches: a weight column beyond N is read from a clamped (valid) address instead of being predicated -- the output
  // columns it feeds are never stored (f64_tile_epilogue resolves only columns < N; N is a multiple of 16 here, so a thread's SPT
  // columns are inside or outside together).  The predicated form, and a per-chunk search for the operand segment (see a_load below),
  // compiled to four exec-masked branches around the staging loads and a dozen scalar branches per chunk.
  auto stage_load = [&](int c, int n0) {
    int col = n0 + sc;
    col = col + SPT <= g.N ? col : g.N - SPT;
    const double* src = B + int64_t(16 * c + sr) * g.N + col;
#pragma unroll
    for (int q = 0; q < SPT / 2; ++q) rb[q] = *reinterpret_cast<const v2d*>(src + 2 * q);
  };
  auto stage_write = [&](int b) {
    double* d = smem + b * 16 * LDB + sr * LDB + sc;
#pragma unroll
    for (int q = 0; q < SPT / 2; ++q) *reinterpret_cast<v2d*>(d + 2 * q) = rb[q];
  };
  // operand chunk c of the lane's two rows: A[row][16 c + 4 lg .. + 3]
  // The chunks are fetched in order, so the operand segment of the NEXT chunk is a running state (pointer, row stride, chunks left
  // in the segment) advanced once per fetch; a segment boundary is one rarely-taken uniform branch.
  const double* nx_p = static_cast<const double*>(g.a.s[0].p) + a_sh;
  int nx_ld = g.a.s[0].ld, nx_left = g.a.s[0].n >> 4, nx_seg = 0;
  auto a_next_segment = [&]() {  // (also steps over empty segments)
    while (nx_left <= 0 && nx_seg + 1 < g.a.count) {
      ++nx_seg;
      if (nx_seg == 1) {  // (compile-time member indices: a run-time index into the argument struct would move it to scratch memory)
        nx_p = static_cast<const double*>(g.a.s[1].p) + a_sh;
        nx_ld = g.a.s[1].ld;
        nx_left = g.a.s[1].n >> 4;
      } else {
        nx_p = static_cast<const double*>(g.a.s[2].p) + a_sh;
        nx_ld = g.a.s[2].ld;
        nx_left = g.a.s[2].n >> 4;
      }
    }
  };
  a_next_segment();
  auto a_load = [&](int c, v2d (*a)[2]) {
    (void)c;
    const double* base = nx_p;
    const int ld = nx_ld;
    nx_p += 16;
    --nx_left;
    a_next_segment();
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const double* ap = base + arow[i] * ld + 4 * lg;
#pragma unroll
      for (int h = 0; h < 2; ++h) a[i][h] = *reinterpret_cast<const v2d*>(ap + 2 * h);
    }
  };
  // the activation of a fetched chunk, applied when the chunk is about to be used -- NOT next to its loads: there it
  // makes every load wait out its own latency before the next one is issued
  auto a_activate = [&](int c, v2d (*a)[2]) {
    if (!a_act(g, 16 * c)) return;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int h = 0; h < 2; ++h) a[i][h] = v2d{silu(a[i][h][0]), silu(a[i][h][1])};
  };
  // 2 x JT x 4 MFMAs of one chunk against the staged weight block
  auto mma_chunk = [&](const double* bs, const v2d (*a)[2], v4d (*acc)[JT]) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      double b[JT];
#pragma unroll
      for (int j = 0; j < JT; ++j) b[j] = bs[(4 * lg + s) * LDB + 16 * j + li];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < JT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i][s >> 1][s & 1], b[j], acc[i][j], 0, 0, 0);
    }
  };
  auto epilogue = [&](int n0, v4d (*acc)[JT]) {
#pragma unroll
    for (int j = 0; j < JT; ++j) f64_tile_epilogue<ASTAT>(g, n0 + j * 16, m_base, li, lg, acc[0][j], acc[1][j], c_sh);
  };
  v4d acc[2][JT];
  auto zero_acc = [&]() {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < JT; ++j) acc[i][j] = v4d{0.0, 0.0, 0.0, 0.0};
  };
  if constexpr (ASTAT) {
    v2d areg[KCMAX][2][2];
#pragma unroll
    for (int c = 0; c < KCMAX; ++c)
      if (c < KC) a_load(c, areg[c]);
    stage_load(0, 0);
    a_activate(0, areg[0]);
    stage_write(0);
    __syncthreads();
    int buf = 0;
    for (int n0 = 0; n0 < g.N; n0 += NB) {
      zero_acc();
#pragma unroll
      for (int c = 0; c < KCMAX; ++c) {
        if (c < KC) {  // (wave-uniform)
          const bool last_c = c + 1 >= KC;
          const bool more = !last_c || n0 + NB < g.N;
          if (more) stage_load(last_c ? 0 : c + 1, last_c ? n0 + NB : n0);
          mma_chunk(smem + buf * 16 * LDB, areg[c], acc);
          // (first column pass: the next resident chunk is activated behind this chunk's MFMAs -- all eight up front cost
          //  0.8 ms on the 128 -> 640 first stage of C5, where nothing overlaps them)
          if (n0 == 0 && c + 1 < KCMAX && c + 1 < KC) a_activate(c + 1, areg[c + 1]);
          if (more) stage_write(buf ^ 1);
          __syncthreads();
          buf ^= 1;
        }
      }
      epilogue(n0, acc);
    }
  } else {
    v2d acur[2][2], anext[2][2];
    zero_acc();
    stage_load(0, 0);
    a_load(0, acur);
    stage_write(0);
    wait_vmem_all();  // (enter the loop with nothing in flight, see wait_vmem_all)
    a_activate(0, acur);
    __syncthreads();
    int buf = 0;
    for (int c = 0; c < KC; ++c) {
      const bool more = c + 1 < KC;
      if (more) {
        stage_load(c + 1, 0);
        a_load(c + 1, anext);
      }
      mma_chunk(smem + buf * 16 * LDB, acur, acc);
      if (more) {
        stage_write(buf ^ 1);
        a_activate(c + 1, anext);
      }
      __syncthreads();
      buf ^= 1;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int h = 0; h < 2; ++h) acur[i][h] = anext[i][h];
    }
    epilogue(0, acc);
  }
}


// destination of one output column, resolved once per 32-column tile (not per element)
struct ColDst {
  float* c;
  const float* z;
  int ld, zld, accum, valid;
};
__device__ __forceinline__ ColDst resolve_col(const GemmArgs& g, int gn) {
  ColDst d;
  d.c = nullptr;
  d.z = nullptr;
  d.ld = d.zld = d.accum = d.valid = 0;
  int c = gn;
#pragma unroll
  for (int s = 0; s < 3; ++s) {
    const bool live = s < g.c.count;
    const bool in = live && !d.valid && c >= 0 && c < g.c.s[s].n && gn < g.N;
    if (in) {
      d.c = static_cast<float*>(g.c.s[s].p) + c;
      d.ld = g.c.s[s].ld;
      d.accum = g.c_accum[s];
      if (g.has_z) {
        d.z = static_cast<const float*>(g.z.s[s].p) + c;
        d.zld = g.z.s[s].ld;
      }
      d.valid = 1;
    }
    if (live) c -= g.c.s[s].n;
  }
  return d;
}
__device__ __forceinline__ void store_tile(const ColDst& d, const v16f& acc, int64_t m0, int64_t M, int lane) {
  if (!d.valid) return;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    const int64_t gm = m0 + row;
    if (gm < M) {
      float v = acc[r];
      if (d.z) v *= dsilu(d.z[gm * d.zld]);
      float* p = d.c + gm * d.ld;
      if (d.accum) v += *p;
      *p = v;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// fp32 MFMA kernel v3 -- no LDS.  Each wave owns 32 rows.  With the k index permuted inside every
// 32-deep chunk (lane half h supplies k = 16h + s at MFMA step s; B is pre-permuted on the host the
// same way), lane (i, h) needs A[row i][chunk*32 + 16h .. +15]: 64 contiguous bytes of its own row,
// fetched straight into VGPRs with 4 x 16-B loads -- every A element is loaded (and activated) by exactly
// one lane, nothing is staged, nothing is synchronised, occupancy is bounded by registers only.
// Column tiles are processed in pairs sharing the A fragment; the next chunk's operands are fetched into
// a second register set before the current chunk's 32 MFMAs issue.
// Requires every A segment width to be a multiple of 16 and 16-B aligned rows (else the v1 kernel runs).
// ---------------------------------------------------------------------------------------------
struct RowSrc {  // per-lane source of one 16-float half-chunk
  const float* p;  // nullptr -> zeros
};

__device__ __forceinline__ bool seglist_frag_ok(const SegList& sl) {
  for (int s = 0; s < sl.count; ++s)
    if ((sl.s[s].n & 15) || (sl.s[s].ld & 3) || (reinterpret_cast<uintptr_t>(sl.s[s].p) & 15)) return false;
  return true;
}

// pointer to A[gm][k .. k+15] (k multiple of 16) or nullptr when out of range
__device__ __forceinline__ const float* a_half_ptr(const GemmArgs& g, int64_t gm, int k, int64_t shift = 0) {
  const float* out = nullptr;
  if (gm < g.M && k < g.K) {
    int c = k;
    bool done = false;
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      const bool live = s < g.a.count;
      if (live && !done && c < g.a.s[s].n) {
        out = static_cast<const float*>(g.a.s[s].p) + gm * g.a.s[s].ld + c + shift;
        done = true;
      }
      if (live) c -= g.a.s[s].n;
    }
  }
  return out;
}

__device__ __forceinline__ void load_a_frag(const GemmArgs& g, int64_t gm, int k, v4f* a) {
  const float* p = a_half_ptr(g, gm, k);
  if (p) {
#pragma unroll
    for (int q = 0; q < 4; ++q) a[q] = reinterpret_cast<const v4f*>(p)[q];
    if (a_act(g, k)) {
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) a[q][e] = silu(a[q][e]);
    }
  } else {
#pragma unroll
    for (int q = 0; q < 4; ++q) a[q] = v4f{0.f, 0.f, 0.f, 0.f};
  }
}

// destination of 4 consecutive output features of one row (a 16-B store), resolved per tile group
struct Dst4 {
  float* c;
  const float* z;
  const float* add;
  int accum, nvalid;  // nvalid: how many of the 4 features exist (0..4)
};
__device__ __forceinline__ Dst4 resolve4(const GemmArgs& g, int64_t gm, int f0, int64_t c_shift = 0) {
  Dst4 d;
  d.c = nullptr;
  d.z = nullptr;
  d.add = nullptr;
  d.accum = 0;
  d.nvalid = 0;
  if (gm >= g.M || f0 >= g.N) return d;
  int c = f0;
  bool done = false;
#pragma unroll
  for (int s = 0; s < 3; ++s) {
    const bool live = s < g.c.count;
    if (live && !done && c < g.c.s[s].n) {
      d.c = g.c.s[s].p ? static_cast<float*>(g.c.s[s].p) + gm * g.c.s[s].ld + c + c_shift : nullptr;
      d.accum = g.c_accum[s];
      if (g.has_z) d.z = static_cast<const float*>(g.z.s[s].p) + gm * g.z.s[s].ld + c;
      if (g.has_add) d.add = static_cast<const float*>(g.add.s[s].p) + gm * g.add.s[s].ld + c;
      d.nvalid = g.c.s[s].n - c < 4 ? g.c.s[s].n - c : 4;
      done = true;
    }
    if (live) c -= g.c.s[s].n;
  }
  return d;
}

// acc holds C^T: column = row index (edge) lane&31, feature = n0 + (r&3) + 8*(r>>2) + 4*(lane>>5)
__device__ __forceinline__ void store_tile_t(const GemmArgs& g, const v16f& acc, int64_t gm, int n0, int lane, bool vec_ok, int64_t c_shift = 0) {
#pragma unroll
  for (int gq = 0; gq < 4; ++gq) {
    const int f0 = n0 + 8 * gq + 4 * (lane >> 5);
    const Dst4 d = resolve4(g, gm, f0, c_shift);
    if (d.nvalid == 0) continue;
    v4f v = {acc[4 * gq], acc[4 * gq + 1], acc[4 * gq + 2], acc[4 * gq + 3]};
    if (vec_ok && d.nvalid == 4) {
      if (d.add) {
        const v4f ad = *reinterpret_cast<const v4f*>(d.add);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += ad[e];
      }
      if (d.z) {
        const v4f z = *reinterpret_cast<const v4f*>(d.z);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] *= dsilu(z[e]);
      }
      if (d.accum) {
        const v4f o = *reinterpret_cast<const v4f*>(d.c);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += o[e];
      }
      *reinterpret_cast<v4f*>(d.c) = v;
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (e < d.nvalid) {
          float x = v[e];
          if (d.add) x += d.add[e];
          if (d.z) x *= dsilu(d.z[e]);
          if (d.accum) x += d.c[e];
          d.c[e] = x;
        }
    }
  }
}

// Epilogue of a tile pair through a wave-private LDS patch [32 rows][64 + 4 features]: the accumulators are
// written in MFMA order (lane = row, 4 consecutive features per register group) and read back so that 16
// consecutive lanes cover one 256-B row segment -> z loads, accumulate loads and stores are full-line
// accesses (the direct epilogue issued 32-B partial-line writes: 64x448 spent 155 of 269 us in its stores).
constexpr int EP_LD = 68;  // floats per patch row
__device__ __forceinline__ void store_pair_lds(const GemmArgs& g, const v16f& acc0, const v16f& acc1, bool two, float* patch,
                                               int64_t m0, int n0, int lane) {
  const int row = lane & 31, h = lane >> 5;
#pragma unroll
  for (int gq = 0; gq < 4; ++gq) {
    *reinterpret_cast<v4f*>(patch + row * EP_LD + 8 * gq + 4 * h) =
        v4f{acc0[4 * gq], acc0[4 * gq + 1], acc0[4 * gq + 2], acc0[4 * gq + 3]};
    if (two)
      *reinterpret_cast<v4f*>(patch + row * EP_LD + 32 + 8 * gq + 4 * h) =
          v4f{acc1[4 * gq], acc1[4 * gq + 1], acc1[4 * gq + 2], acc1[4 * gq + 3]};
  }
  __builtin_amdgcn_wave_barrier();
  const int c4 = (lane & 15) * 4;  // feature offset inside the pair
  const int f0 = n0 + c4;
  const bool col_ok = (c4 < 32 || two) && f0 < g.N;
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    const int r = it * 4 + (lane >> 4);
    const int64_t gm = m0 + r;
    if (!col_ok) continue;
    const Dst4 d = resolve4(g, gm, f0);
    if (d.nvalid != 4) continue;  // (segments are 4-granular on this path)
    v4f v = *reinterpret_cast<const v4f*>(patch + r * EP_LD + c4);
    if (d.add) {
      const v4f ad = *reinterpret_cast<const v4f*>(d.add);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] += ad[e];
    }
    if (d.z) {
      const v4f z = *reinterpret_cast<const v4f*>(d.z);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] *= dsilu(z[e]);
    }
    if (d.accum) {
      const v4f o = *reinterpret_cast<const v4f*>(d.c);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] += o[e];
    }
    *reinterpret_cast<v4f*>(d.c) = v;
  }
  __builtin_amdgcn_wave_barrier();
}

// KCR > 0: K <= 32*KCR and the row's activation fragments stay in registers for all column tiles;
// KCR == 0: fragments are streamed (and double-buffered) per k chunk.
template <int KCR>
__global__ __launch_bounds__(256) void gemm_mfma_f32_v3_kernel(GemmArgs g, int vec_ok) {
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int64_t m0 = (int64_t(blockIdx.x) * 4 + wv) * 32;
  const int64_t gm = m0 + (lane & 31);
  const int kh = (lane >> 5) * 16;
  const int KC = (g.K + 31) >> 5;
  const int NT = (g.N + 31) >> 5;
  const v4f* Wp = static_cast<const v4f*>(g.Bp) + size_t(lane) * 4;
  const size_t tile_stride = size_t(KC) * 64 * 4;  // v4f units between consecutive feature tiles
  v4f xr[KCR > 0 ? KCR : 1][4];
  if (KCR > 0) {
#pragma unroll
    for (int kc = 0; kc < KCR; ++kc) load_a_frag(g, gm, kc * 32 + kh, xr[kc]);
  }
  int nt = 0;
  for (; nt + 1 < NT; nt += 2) {
    v16f acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      acc0[r] = 0.f;
      acc1[r] = 0.f;
    }
    const v4f* wp0 = Wp + size_t(nt) * tile_stride;
    const v4f* wp1 = wp0 + tile_stride;
    if (KCR > 0) {
      // weights of step (pair, kc+1) are fetched while the 32 MFMAs of step (pair, kc) issue; the first
      // chunk of the NEXT pair is fetched during the last chunk of this one
      v4f w0[4], w1[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        w0[q] = wp0[q];
        w1[q] = wp1[q];
      }
#pragma unroll
      for (int kc = 0; kc < KCR; ++kc) {
        if (kc < KC) {
          v4f w0n[4], w1n[4];
          const bool more = kc + 1 < KC;
          const v4f* n0p = more ? wp0 + size_t(kc + 1) * 256 : wp0;
          const v4f* n1p = more ? wp1 + size_t(kc + 1) * 256 : wp1;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            w0n[q] = n0p[q];
            w1n[q] = n1p[q];
          }
#pragma unroll
          for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(w0[q][e], xr[kc][q][e], acc0, 0, 0, 0);
              acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(w1[q][e], xr[kc][q][e], acc1, 0, 0, 0);
            }
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            w0[q] = w0n[q];
            w1[q] = w1n[q];
          }
        }
      }
    } else {
      v4f x[4], w0[4], w1[4];
      load_a_frag(g, gm, kh, x);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        w0[q] = wp0[q];
        w1[q] = wp1[q];
      }
      for (int kc = 0; kc < KC; ++kc) {
        v4f xn[4], w0n[4], w1n[4];
        const int kn = kc + 1 < KC ? kc + 1 : kc;  // the last iteration re-reads its own chunk (harmless)
        load_a_frag(g, gm, kn * 32 + kh, xn);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          w0n[q] = wp0[size_t(kn) * 256 + q];
          w1n[q] = wp1[size_t(kn) * 256 + q];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(w0[q][e], x[q][e], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(w1[q][e], x[q][e], acc1, 0, 0, 0);
          }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          x[q] = xn[q];
          w0[q] = w0n[q];
          w1[q] = w1n[q];
        }
      }
    }
    store_tile_t(g, acc0, gm, nt * 32, lane, vec_ok);
    store_tile_t(g, acc1, gm, nt * 32 + 32, lane, vec_ok);
  }
  if (nt < NT) {  // odd tail tile
    v16f acc0;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc0[r] = 0.f;
    const v4f* wp0 = Wp + size_t(nt) * tile_stride;
    for (int kc = 0; kc < KC; ++kc) {
      v4f x[4], w0[4];
      load_a_frag(g, gm, kc * 32 + kh, x);
#pragma unroll
      for (int q = 0; q < 4; ++q) w0[q] = wp0[size_t(kc) * 256 + q];
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(w0[q][e], x[q][e], acc0, 0, 0, 0);
    }
    store_tile_t(g, acc0, gm, nt * 32, lane, vec_ok);
  }
}

// "accumulator order" k-permutation used by the bf16x3 kernels: element s (0..15) of lane half h in chunk c is
// feature k = 32c + 8*(s>>2) + 4h + (s&3) -- exactly the feature a lane holds in accumulator register s of the
// swapped-operand C layout, so a layer's accumulators can be fed to the next layer's MFMAs without any data
// movement (gemm_chain_bf16x3_kernel).  From global memory it is four 16-B pieces of the lane's own row.
__device__ __forceinline__ void load_a_frag_acc(const GemmArgs& g, int64_t gm, int chunk, int h, v4f* a, int64_t shift = 0) {
  const float* p = a_half_ptr(g, gm, chunk * 32, shift);  // pointer to A[gm][32*chunk] (segments are 32-granular here)
  if (p) {
#pragma unroll
    for (int q = 0; q < 4; ++q) a[q] = *reinterpret_cast<const v4f*>(p + 8 * q + 4 * h);
    if (a_act(g, chunk * 32)) {
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) a[q][e] = silu(a[q][e]);
    }
  } else {
#pragma unroll
    for (int q = 0; q < 4; ++q) a[q] = v4f{0.f, 0.f, 0.f, 0.f};
  }
}

// ---------------------------------------------------------------------------------------------
// fp32 GEMM on the bf16 matrix cores by exact 3-way splitting ("bf16x3").
// Every fp32 value is the exact sum of three bf16 numbers (x = x1 + x2 + x3: 3 x 8 significand bits,
// obtained by truncation and exact residuals).  x*w is reproduced to ~2^-24 by the 6 leading cross
// products x1w1, x1w2, x2w1, x1w3, x2w2, x3w1, each exact in fp32 and accumulated in fp32 by
// v_mfma_f32_32x32x16_bf16.  Six bf16 MFMAs (K=16, 32 cycles/SIMD) replace sixteen fp32 MFMAs (K=2,
// 64 cycles) per 32-deep chunk and tile pair: 2.7x less matrix-pipe time at fp32-class accuracy.
// Same data flow as v3: activation fragments straight from global (64 contiguous bytes per lane and
// chunk), weights pre-split and pre-permuted on the host, swapped operands, 16-B epilogue accesses.
// ---------------------------------------------------------------------------------------------

// one 32-deep chunk of a tile pair: 2 k-halves x 6 cross products per tile; the two tiles' accumulator
// chains are interleaved so consecutive MFMAs are independent
__device__ __forceinline__ void chunk_pair_bf16x3(const u32x4* w0, const u32x4* w1, const u32x4* x1, const u32x4* x2,
                                                  const u32x4* x3, v16f& acc0, v16f& acc1) {
#define AA_STEP(WL, XL)                          \
  acc0 = mma_bf16(w0[2 * WL + 0], XL[0], acc0);  \
  acc1 = mma_bf16(w1[2 * WL + 0], XL[0], acc1);  \
  acc0 = mma_bf16(w0[2 * WL + 1], XL[1], acc0);  \
  acc1 = mma_bf16(w1[2 * WL + 1], XL[1], acc1);
  AA_STEP(2, x1)  // w3 x1
  AA_STEP(1, x2)  // w2 x2
  AA_STEP(0, x3)  // w1 x3
  AA_STEP(1, x1)  // w2 x1
  AA_STEP(0, x2)  // w1 x2
  AA_STEP(0, x1)  // w1 x1
#undef AA_STEP
}

// weights: Wq[tile][chunk][lane][level 0..2][half 0..1] as u32x4 (6 x 16 B per lane and chunk).
// Software pipeline: the weight fragments of step (pair, chunk)+1 are in flight while the 24 MFMAs of the
// current step issue (also across tile-pair boundaries); streamed activations are fetched two chunks ahead.
template <int KCR, bool LDS_EPI>
__global__ __launch_bounds__(256) void gemm_bf16x3_kernel(GemmArgs g, const u32x4* __restrict__ Wq, int vec_ok) {
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int64_t m0 = (int64_t(blockIdx.x) * 4 + wv) * 32;
  const int64_t gm = m0 + (lane & 31);
  const int hh = lane >> 5;
  const int KC = (g.K + 31) >> 5;
  const int NT = (g.N + 31) >> 5;
  // batched launch (GemmArgs::batch): problem blockIdx.z, see gemm_f64_rows_kernel
  const int bz = g.batch > 1 ? int(blockIdx.z) : 0;
  const int64_t a_sh = bz * g.a_bs, c_sh = bz * g.c_bs;
  if (g.batch > 1) Wq = reinterpret_cast<const u32x4*>(static_cast<const float*>(g.Bq) + int64_t((g.bsel4 >> (4 * bz)) & 15ull) * g.bq_bs);
  const u32x4* Wl = Wq + lane;  // fragments are stored [tile][chunk][q][lane]
  const size_t chunk_stride = 64 * 6;                    // u32x4 units between chunks
  const size_t tile_stride = size_t(KC) * chunk_stride;  // between feature tiles
  auto wptr = [&](int tile, int kc) { return Wl + size_t(tile < NT ? tile : NT - 1) * tile_stride + size_t(kc) * chunk_stride; };
  u32x4 xr1[KCR > 0 ? KCR : 1][2], xr2[KCR > 0 ? KCR : 1][2], xr3[KCR > 0 ? KCR : 1][2];
  if (KCR > 0) {
#pragma unroll
    for (int kc = 0; kc < KCR; ++kc) {
      v4f a[4];
      load_a_frag_acc(g, gm, kc, hh, a, a_sh);
      split3_pack(a, xr1[kc], xr2[kc], xr3[kc]);
    }
  }
  u32x4 wc0[6], wc1[6];
#pragma unroll
  for (int q = 0; q < 6; ++q) {
    wc0[q] = wptr(0, 0)[q * 64];
    wc1[q] = wptr(1, 0)[q * 64];
  }
  for (int nt = 0; nt < NT; nt += 2) {
    const bool two = nt + 1 < NT;
    v16f acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      acc0[r] = 0.f;
      acc1[r] = 0.f;
    }
    if (KCR > 0) {
#pragma unroll
      for (int kc = 0; kc < KCR; ++kc) {
        if (kc < KC) {
          const bool last = kc + 1 >= KC;
          const int nnt = last ? nt + 2 : nt, nkc = last ? 0 : kc + 1;
          u32x4 wn0[6], wn1[6];
          const u32x4* p0 = wptr(nnt, nkc);
          const u32x4* p1 = wptr(nnt + 1, nkc);
#pragma unroll
          for (int q = 0; q < 6; ++q) {
            wn0[q] = p0[q * 64];
            wn1[q] = p1[q * 64];
          }
          chunk_pair_bf16x3(wc0, wc1, xr1[kc], xr2[kc], xr3[kc], acc0, acc1);
#pragma unroll
          for (int q = 0; q < 6; ++q) {
            wc0[q] = wn0[q];
            wc1[q] = wn1[q];
          }
        }
      }
    } else {
      v4f a0[4], a1[4];
      load_a_frag_acc(g, gm, 0, hh, a0, a_sh);
      load_a_frag_acc(g, gm, KC > 1 ? 1 : 0, hh, a1, a_sh);
      for (int kc = 0; kc < KC; ++kc) {
        // L2-resident weights are requested BEFORE the HBM activations: loads return in order, so the other
        // way round every step would wait a full HBM latency for its weights
        const bool last = kc + 1 >= KC;
        const int nnt = last ? nt + 2 : nt, nkc = last ? 0 : kc + 1;
        u32x4 wn0[6], wn1[6];
        const u32x4* p0 = wptr(nnt, nkc);
        const u32x4* p1 = wptr(nnt + 1, nkc);
#pragma unroll
        for (int q = 0; q < 6; ++q) {
          wn0[q] = p0[q * 64];
          wn1[q] = p1[q * 64];
        }
        __builtin_amdgcn_sched_barrier(0);
        v4f a2[4];
        const int k2 = kc + 2 < KC ? kc + 2 : KC - 1;
        load_a_frag_acc(g, gm, k2, hh, a2, a_sh);  // two chunks ahead
        __builtin_amdgcn_sched_barrier(0);
        u32x4 x1[2], x2[2], x3[2];
        split3_pack(a0, x1, x2, x3);
        chunk_pair_bf16x3(wc0, wc1, x1, x2, x3, acc0, acc1);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          a0[q] = a1[q];
          a1[q] = a2[q];
        }
#pragma unroll
        for (int q = 0; q < 6; ++q) {
          wc0[q] = wn0[q];
          wc1[q] = wn1[q];
        }
      }
    }
    if (LDS_EPI) {
      store_pair_lds(g, acc0, acc1, two, reinterpret_cast<float*>(aa_smem) + wv * 32 * EP_LD, m0, nt * 32, lane);
    } else {
      store_tile_t(g, acc0, gm, nt * 32, lane, vec_ok, c_sh);
      if (two) store_tile_t(g, acc1, gm, nt * 32 + 32, lane, vec_ok, c_sh);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Fused chain of linear layers (see ChainArgs in aa_common.h).  One wave = 32 edges; a layer's 64 kept output
// features stay in accumulator registers (optionally activated) and are split into bf16 levels as the last
// two k chunks of the next layer -- the hidden activations of the reference's ScalarMLPFunction chains never
// travel through HBM except for the one store of the pre-activation that the reverse pass needs.
// ---------------------------------------------------------------------------------------------

// epilogue of one tile in accumulator layout, in place: acc <- (acc + add) * silu'(z), then stored (=|+=) unless
// the destination segment is a "computed only" (null) segment
__device__ __forceinline__ void tile_epilogue_store(const GemmArgs& g, v16f& acc, int64_t gm, int n0, int lane) {
#pragma unroll
  for (int gq = 0; gq < 4; ++gq) {
    const int f0 = n0 + 8 * gq + 4 * (lane >> 5);
    const Dst4 d = resolve4(g, gm, f0);
    if (d.nvalid == 0) continue;
    v4f v = {acc[4 * gq], acc[4 * gq + 1], acc[4 * gq + 2], acc[4 * gq + 3]};
    if (d.add) {
      const v4f ad = *reinterpret_cast<const v4f*>(d.add);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] += ad[e];
    }
    if (d.z) {
      const v4f z = *reinterpret_cast<const v4f*>(d.z);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] *= dsilu(z[e]);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[4 * gq + e] = v[e];
    if (d.c) {
      if (d.accum) {
        const v4f o = *reinterpret_cast<const v4f*>(d.c);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += o[e];
      }
      *reinterpret_cast<v4f*>(d.c) = v;
    }
  }
}

// Device-side form of a chain: every segment list is resolved on the host into one pointer per 32-column
// chunk (operands) / 32-feature tile (destinations), so the kernel's inner code is branch-free pointer
// arithmetic.  (Chains require 32-column-granular segments; the model's chain path has 64-granular ones.)
constexpr int kChainMaxBlocks = 12;  // 32-wide chunks / tiles per layer
struct ChainTileDev {
  float* c;          // destination of the tile (row 0), nullptr: computed only
  const float* z;    // silu' factor operand or nullptr
  const float* add;  // addend or nullptr
  int ldc, ldz, ldadd, accum;
};
struct ChainLayerDev {
  const u32x4* Wq;
  int KCg, KC, NT;  // k chunks from global memory, total k chunks (+2 chained), output tiles
  int keep_tile, keep_act, a_mode, pad_;
  int use_prev, pad2_;
  float* edge_sum_out;
  float* embrev_out;
  float* kept_out;
  int ld_kept, pad3_;
  const float *a2_add, *a2_z;  // a_mode 2 operand transform
  int ld_a2add, ld_a2z;
  const float* a[kChainMaxBlocks];
  int lda[kChainMaxBlocks];
  ChainTileDev t[kChainMaxBlocks];
};
struct ChainDev {
  int64_t M;
  int nlayers;
  float ro_factor;
  const float* ro_w;
  const float* ro_scales;
  const int32_t* types;
  const int32_t* center;
  int ro_n;  // entries of ro_w used by a_mode-1 layers
  const int32_t* nbr;
  const float* emb_table;  // [T*T][8][64] or nullptr
  int num_types;
  ChainLayerDev L[4];
};

// Weights of one step (tile pair x 32-deep chunk: 2 x 6 x 64 fragments of 16 B = 12 KB) are staged through LDS
// by the whole block (the four waves run the same layer/tile/chunk sequence on different rows): one L2 fetch
// instead of four, the next step's weights -- also across layer boundaries -- are in flight while this step's
// MFMAs issue (double buffer, one barrier per step), and the MFMA operands are read just in time with
// conflict-free ds_read_b128.  Operand rows are loaded one step ahead, also across tile-pair and layer
// boundaries; rows beyond M are clamped for loads and masked for stores.
constexpr int kEpLd = 36;          // row stride (floats) of the store-transpose patch: 32 + 4 keeps b128 accesses conflict-light

// PRE: layers with at most two k chunks and more than two output tiles (a chained 64-wide operand feeding a wide
// layer) split their operand once and reuse it for every tile pair; that costs 48 VGPRs, so the variant without
// such layers runs at 3 waves/SIMD and the one with them at 2.
// EMB: the embrev_out epilogue (reverse of the two-body basis expansion) is compiled in; only the last reverse chain
// of a step uses it, and it costs registers the other chains should not pay for.
template <bool PRE, bool EMB>
__global__ __launch_bounds__(256, PRE ? 2 : 3) void gemm_chain_bf16x3_kernel(ChainDev c) {
  // (which split: see split3_pack_masked)
  auto chain_split = [](const v4f* a, u32x4* l1, u32x4* l2, u32x4* l3) {
    if constexpr (PRE)
      split3_pack_masked(a, l1, l2, l3);
    else
      split3_pack(a, l1, l2, l3);
  };
  u32x4* wbuf = reinterpret_cast<u32x4*>(aa_smem);  // [2][kWStep]
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int64_t m0 = (int64_t(blockIdx.x) * 4 + wv) * 32;
  const int64_t gm = m0 + (lane & 31);
  const int hh = lane >> 5;
  const bool row_ok = gm < c.M;
  const int64_t gmc = row_ok ? gm : c.M - 1;
  v16f kept0, kept1;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    kept0[r] = 0.f;
    kept1[r] = 0.f;
  }
  float rofac = c.ro_factor;  // readout-reverse transform factor of this row
  if (c.ro_scales) rofac *= c.ro_scales[c.types[c.center[gmc]]];
  float* sRo = reinterpret_cast<float*>(wbuf + 2 * kWStep);  // [ro_n] last readout weights (a_mode 1)
  for (int i = tid; i < c.ro_n; i += 256) sRo[i] = c.ro_w[i];
  float* sT = sRo + 32 * kChainMaxBlocks + wv * 32 * kEpLd;  // wave-private [32 rows][kEpLd] store-transpose patch
  // two-body embedding table of this row's type pair (embrev_out)
  const float* sTab = sRo + 32 * kChainMaxBlocks + 4 * 32 * kEpLd;  // [T*T][8][64]
  if (EMB && c.emb_table) {
    float* tab = const_cast<float*>(sTab);
    for (int i = tid; i < c.num_types * c.num_types * 512; i += 256) tab[i] = c.emb_table[i];
    sTab += (c.types[c.center[gmc]] * c.num_types + c.types[c.nbr[gmc]]) * 512;
  }
  // block-cooperative staging: thread t moves elements t, t+256, t+512 of the 768-element step
  auto stage_load = [&](const ChainLayerDev& L, int nt, int kc, u32x4* r) {
    const size_t chunk_stride = 64 * 6, tile_stride = size_t(L.KC) * chunk_stride;
    const int t1 = nt + 1 < L.NT ? nt + 1 : nt;
    const u32x4* s0 = L.Wq + size_t(nt) * tile_stride + size_t(kc) * chunk_stride;
    const u32x4* s1 = L.Wq + size_t(t1) * tile_stride + size_t(kc) * chunk_stride;
    r[0] = s0[tid];
    r[1] = tid < 128 ? s0[256 + tid] : s1[tid - 128];
    r[2] = s1[128 + tid];
  };
  auto stage_write = [&](int b, const u32x4* r) {
    u32x4* d = wbuf + b * kWStep;
    d[tid] = r[0];
    d[256 + tid] = r[1];
    d[512 + tid] = r[2];
  };
  // raw operand rows of global chunk kc (four 16-B pieces of the lane's row, accumulator-order k)
  auto load_a = [&](const ChainLayerDev& L, int kc, v4f* a) {
    const float* p = L.a[kc] + gmc * L.lda[kc] + 4 * hh;
#pragma unroll
    for (int q = 0; q < 4; ++q) a[q] = *reinterpret_cast<const v4f*>(p + 8 * q);
  };
  // raw -> operand (readout-reverse transform of a_mode 1)
  auto finish_a = [&](const ChainLayerDev& L, int kc, v4f* a) {
    if (L.a_mode == 1) {
      const float* rw = sRo + kc * 32 + 4 * hh;  // (LDS copy: no vector-memory access inside a step's control flow)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const v4f wv4 = *reinterpret_cast<const v4f*>(rw + 8 * q);
#pragma unroll
        for (int e = 0; e < 4; ++e) a[q][e] = rofac * wv4[e] * dsilu(a[q][e]);
      }
    }
  };
  // a_mode 2 (only in layers whose operand is split once, OUTSIDE the step loop -- the streaming steps keep their uniform
  // vector-memory sequence; launch_gemm_chain checks the shape): A = (a + add) * silu'(z)
  auto finish_a2 = [&](const ChainLayerDev& L, int kc, v4f* a) {
    const float* pa = L.a2_add + gmc * L.ld_a2add + kc * 32 + 4 * hh;
    const float* pz = L.a2_z + gmc * L.ld_a2z + kc * 32 + 4 * hh;
    v4f ad[4], zz[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      ad[q] = *reinterpret_cast<const v4f*>(pa + 8 * q);
      zz[q] = *reinterpret_cast<const v4f*>(pz + 8 * q);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int e = 0; e < 4; ++e) a[q][e] = (a[q][e] + ad[q][e]) * dsilu(zz[q][e]);
  };
  // chained chunk j of a layer: the kept pair (use_prev)
  auto kept_a = [&](const ChainLayerDev& L, int j, v4f* a) {
    const bool first = (j & 1) == 0;
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int e = 0; e < 4; ++e) a[q][e] = first ? kept0[4 * q + e] : kept1[4 * q + e];
  };
  // 24 MFMAs of one step; weight levels are read from LDS just in time (level l is reused by 3-l products)
  auto mma_step = [&](int b, const u32x4* x1, const u32x4* x2, const u32x4* x3, v16f& acc0, v16f& acc1) {
    const u32x4* w = wbuf + b * kWStep + lane;
#define AA_W(T_, Q_) w[((T_)*6 + (Q_)) * 64]
    {
      const u32x4 a0 = AA_W(0, 4), b0 = AA_W(1, 4), a1 = AA_W(0, 5), b1 = AA_W(1, 5);  // level 3
      acc0 = mma_bf16(a0, x1[0], acc0);
      acc1 = mma_bf16(b0, x1[0], acc1);
      acc0 = mma_bf16(a1, x1[1], acc0);
      acc1 = mma_bf16(b1, x1[1], acc1);
    }
    {
      const u32x4 a0 = AA_W(0, 2), b0 = AA_W(1, 2), a1 = AA_W(0, 3), b1 = AA_W(1, 3);  // level 2
      acc0 = mma_bf16(a0, x2[0], acc0);
      acc1 = mma_bf16(b0, x2[0], acc1);
      acc0 = mma_bf16(a1, x2[1], acc0);
      acc1 = mma_bf16(b1, x2[1], acc1);
      acc0 = mma_bf16(a0, x1[0], acc0);
      acc1 = mma_bf16(b0, x1[0], acc1);
      acc0 = mma_bf16(a1, x1[1], acc0);
      acc1 = mma_bf16(b1, x1[1], acc1);
    }
    {
      const u32x4 a0 = AA_W(0, 0), b0 = AA_W(1, 0), a1 = AA_W(0, 1), b1 = AA_W(1, 1);  // level 1
      acc0 = mma_bf16(a0, x3[0], acc0);
      acc1 = mma_bf16(b0, x3[0], acc1);
      acc0 = mma_bf16(a1, x3[1], acc0);
      acc1 = mma_bf16(b1, x3[1], acc1);
      acc0 = mma_bf16(a0, x2[0], acc0);
      acc1 = mma_bf16(b0, x2[0], acc1);
      acc0 = mma_bf16(a1, x2[1], acc0);
      acc1 = mma_bf16(b1, x2[1], acc1);
      acc0 = mma_bf16(a0, x1[0], acc0);
      acc1 = mma_bf16(b0, x1[0], acc1);
      acc0 = mma_bf16(a1, x1[1], acc0);
      acc1 = mma_bf16(b1, x1[1], acc1);
    }
#undef AA_W
  };
  // epilogue of one tile in accumulator layout, in place: acc <- (acc + add) * silu'(z), then stored (=|+=).
  // All operand loads of the tile are issued together (one memory latency per tile, not one per 16-B group).
  auto epilogue = [&](const ChainTileDev& d, v16f& acc) {
    const bool has_add = d.add != nullptr, has_z = d.z != nullptr, has_old = d.c != nullptr && d.accum;
    {
      v4f ad[4], zz[4];
      if (has_add) {
        const float* p = d.add + gmc * d.ldadd + 4 * hh;
#pragma unroll
        for (int q = 0; q < 4; ++q) ad[q] = *reinterpret_cast<const v4f*>(p + 8 * q);
      }
      if (has_z) {
        const float* p = d.z + gmc * d.ldz + 4 * hh;
#pragma unroll
        for (int q = 0; q < 4; ++q) zz[q] = *reinterpret_cast<const v4f*>(p + 8 * q);
      }
      if (has_add) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] += ad[r >> 2][r & 3];
      }
      if (has_z) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] *= dsilu(zz[r >> 2][r & 3]);
      }
    }
    if (d.c != nullptr) {
      // Stores go through a wave-private LDS patch so that every store instruction writes whole 128-B lines
      // (8 lanes x 16 B per row, 8 rows): in the accumulator layout an instruction would write 32 B of each of 32
      // lines, and partial-line writes stream ~25 % slower on this memory system (tools/ubench/gemm_steps.hip:
      // 4.3 vs 5.3 TB/s for a copy).  Tiles are 32 features = one line wide and rows are 128-B aligned here.
      float* st = sT + (lane & 31) * kEpLd + 4 * hh;
#pragma unroll
      for (int q = 0; q < 4; ++q) *reinterpret_cast<v4f*>(st + 8 * q) = v4f{acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]};
      __builtin_amdgcn_wave_barrier();
      const int pr = lane >> 3, pc = 4 * (lane & 7);
      v4f v[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) v[q] = *reinterpret_cast<const v4f*>(sT + (8 * q + pr) * kEpLd + pc);
      __builtin_amdgcn_wave_barrier();
      float* p = d.c + (m0 + pr) * d.ldc + pc;
      if (has_old) {  // (no tile of the model's chains has both an accumulated destination and add / z operands)
        v4f old[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int64_t rr = m0 + pr + 8 * q < c.M ? m0 + pr + 8 * q : c.M - 1;
          old[q] = *reinterpret_cast<const v4f*>(d.c + rr * d.ldc + pc);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int e = 0; e < 4; ++e) v[q][e] += old[q][e];
      }
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (m0 + pr + 8 * q < c.M) *reinterpret_cast<v4f*>(p + int64_t(8 * q) * d.ldc) = v[q];
    }
  };

  // Every step issues the same vector-memory sequence -- 3 weight-staging loads, then 4 operand-row loads -- on
  // every control path (targets are chosen with scalar selects; a step with nothing to prefetch re-reads layer
  // 0's first chunk, an L2 hit).  With conditional loads the compiler's s_waitcnt placement has to assume the
  // worst path and drains the operand prefetch together with the staging loads, which serialises a full memory
  // latency into every step.
  const float* const dummy_a = c.L[0].a[0];
  const int dummy_lda = c.L[0].lda[0];
  auto load_rows = [&](const float* base, int ld, v4f* a) {
    const float* p = base + gmc * ld + 4 * hh;
#pragma unroll
    for (int q = 0; q < 4; ++q) a[q] = *reinterpret_cast<const v4f*>(p + 8 * q);
  };
  auto streams = [&](const ChainLayerDev& X) { return X.KCg > 0 && !(PRE && X.KC <= 2 && X.NT > 2); };
  int step = 0;
  {
    u32x4 r[3];
    stage_load(c.L[0], 0, 0, r);
    stage_write(0, r);
  }
  // operand rows of the next streamed step, always loaded one step ahead
  v4f a0[4];
  load_rows(dummy_a, dummy_lda, a0);  // (layer 0, chunk 0: the first streamed operand when layer 0 streams)
  __syncthreads();
  for (int li = 0; li < c.nlayers; ++li) {
    const ChainLayerDev& L = c.L[li];
    const int KCg = L.KCg, KC = L.KC, NT = L.NT;
    const bool last_layer = li + 1 >= c.nlayers;
    const ChainLayerDev& Ln = c.L[last_layer ? li : li + 1];
    const bool next_streams = !last_layer && streams(Ln);
    // few k chunks but several tile pairs: split the operands once, not once per pair
    const bool pre = PRE && KC <= 2 && NT > 2;
    u32x4 ps1[PRE ? 2 : 1][2], ps2[PRE ? 2 : 1][2], ps3[PRE ? 2 : 1][2];
    if constexpr (PRE) {
      if (pre) {
#pragma unroll
        for (int kc = 0; kc < 2; ++kc) {
          if (kc < KC) {
            v4f a[4];
            if (kc >= KCg) {
              kept_a(L, kc - KCg, a);
            } else {
              load_a(L, kc, a);
              if (L.a_mode == 2)
                finish_a2(L, kc, a);
              else
                finish_a(L, kc, a);
            }
            chain_split(a, ps1[kc], ps2[kc], ps3[kc]);
          }
        }
      }
    }
    for (int nt = 0; nt < NT; nt += 2) {
      const bool two = nt + 1 < NT;
      const bool last_pair = nt + 2 >= NT;
      v16f acc0, acc1;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        acc0[r] = 0.f;
        acc1[r] = 0.f;
      }
      for (int kc = 0; kc < KC; ++kc) {
        const bool lastc = kc + 1 >= KC;
        const bool layer_end = lastc && last_pair;
        const bool kernel_end = layer_end && last_layer;
        // next step's weights (the last step of the kernel re-stages its own: uniform instruction stream)
        u32x4 r[3];
        stage_load(layer_end && !kernel_end ? Ln : L, (layer_end && !kernel_end) ? 0 : (lastc && !kernel_end ? nt + 2 : nt),
                   (lastc && !kernel_end) ? 0 : (kernel_end ? kc : kc + 1), r);
        // this step's operand, split into its three bf16 levels: the prefetched rows or the chained accumulators
        u32x4 x1[2], x2[2], x3[2];
        if (!pre) {
          if (kc >= KCg) {
            v4f a[4];
            kept_a(L, kc - KCg, a);
            chain_split(a, x1, x2, x3);
          } else {
            finish_a(L, kc, a0);
            chain_split(a0, x1, x2, x3);
          }
        }
        // rows of the next streamed step (next chunk / next pair / next layer), in flight during this step's MFMAs
        {
          const float* tp = dummy_a;
          int tl = dummy_lda;
          if (!pre && !lastc && kc + 1 < KCg) {
            tp = L.a[kc + 1];
            tl = L.lda[kc + 1];
          } else if (lastc && !last_pair && KCg > 0 && !pre) {
            tp = L.a[0];
            tl = L.lda[0];
          } else if (layer_end && next_streams) {
            tp = Ln.a[0];
            tl = Ln.lda[0];
          }
          load_rows(tp, tl, a0);
        }
        if constexpr (PRE) {
          if (pre) {
            if (kc == 0)
              mma_step(step & 1, ps1[0], ps2[0], ps3[0], acc0, acc1);
            else
              mma_step(step & 1, ps1[1], ps2[1], ps3[1], acc0, acc1);
          } else {
            mma_step(step & 1, x1, x2, x3, acc0, acc1);
          }
        } else {
          mma_step(step & 1, x1, x2, x3, acc0, acc1);
        }
        stage_write((step + 1) & 1, r);
#ifdef AA_CHAIN_SYNCTHREADS
        __syncthreads();
#else
        lds_barrier();  // (not __syncthreads(): its vmcnt(0) would wait for the operand rows just requested for the NEXT step)
#endif
        ++step;
      }
      epilogue(L.t[nt], acc0);
      if (two) epilogue(L.t[nt + 1], acc1);
      if (EMB && L.embrev_out) {  // (64-wide layer: this is its only tile pair)
        float part[8];
#pragma unroll
        for (int n = 0; n < 8; ++n) {
          float p = 0.f;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const v4f t0 = *reinterpret_cast<const v4f*>(sTab + n * 64 + 8 * q + 4 * hh);
            const v4f t1 = *reinterpret_cast<const v4f*>(sTab + n * 64 + 32 + 8 * q + 4 * hh);
#pragma unroll
            for (int e = 0; e < 4; ++e) p += acc0[4 * q + e] * t0[e] + acc1[4 * q + e] * t1[e];
          }
          part[n] = p;
        }
        float other[8];
#pragma unroll
        for (int n = 0; n < 8; ++n) other[n] = part[n];
        permlane32_swap4(part, other);
        permlane32_swap4(part + 4, other + 4);
        if (row_ok) {  // each lane half stores four of the eight sums of its row
          const int o4 = 4 * hh;
          *reinterpret_cast<v4f*>(L.embrev_out + gm * 8 + o4) =
              v4f{part[o4] + other[o4], part[o4 + 1] + other[o4 + 1], part[o4 + 2] + other[o4 + 2], part[o4 + 3] + other[o4 + 3]};
        }
      }
      if (L.edge_sum_out) {  // (64-wide layer: this is its only tile pair)
        float part = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const v4f w0v = *reinterpret_cast<const v4f*>(sRo + 8 * q + 4 * hh);
          const v4f w1v = *reinterpret_cast<const v4f*>(sRo + 32 + 8 * q + 4 * hh);
#pragma unroll
          for (int e = 0; e < 4; ++e) part += silu(acc0[4 * q + e]) * w0v[e] + silu(acc1[4 * q + e]) * w1v[e];
        }
        float t = part, o = part;
        permlane32_swap(t, o);  // both lane halves of a row: own + partner
        if (row_ok && hh == 0) L.edge_sum_out[gm] = t + o;
      }
      if (nt == L.keep_tile) {  // (the kept pair is the last pair of its layer: nothing reads the old kept tiles any more)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          kept0[r] = L.keep_act ? silu(acc0[r]) : acc0[r];
          kept1[r] = L.keep_act ? silu(acc1[r]) : acc1[r];
        }
        if (L.kept_out && row_ok) {  // (accumulator layout: register 4 q + e of tile t holds feature 32 t + 8 q + 4 hh + e of this lane's row)
          float* ko = L.kept_out + gm * L.ld_kept + 4 * hh;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            *reinterpret_cast<v4f*>(ko + 8 * q) = v4f{kept0[4 * q], kept0[4 * q + 1], kept0[4 * q + 2], kept0[4 * q + 3]};
            *reinterpret_cast<v4f*>(ko + 32 + 8 * q) = v4f{kept1[4 * q], kept1[4 * q + 1], kept1[4 * q + 2], kept1[4 * q + 3]};
          }
        }
      }
    }
  }
}

int launch_gemm_chain(const ChainArgs& c, hipStream_t stream) {
  if (c.M == 0) return AA_OK;
  ChainDev d{};
  d.M = c.M;
  d.nlayers = c.nlayers;
  d.ro_factor = float(c.ro_factor);
  d.ro_w = static_cast<const float*>(c.ro_w);
  d.ro_scales = static_cast<const float*>(c.ro_scales);
  d.types = c.types;
  d.center = c.center;
  d.nbr = c.nbr;
  d.emb_table = static_cast<const float*>(c.emb_table);
  d.num_types = c.num_types;
  if (c.emb_table && (c.num_types < 1 || c.num_types > 2 || !c.types || !c.center || !c.nbr))
    return fail(AA_ERR_INVALID, "gemm chain: the embedding table needs 1..2 types and the edge / type arrays");
  if (c.nlayers < 1 || c.nlayers > 4) return fail(AA_ERR_INVALID, "gemm chain: 1..4 layers");
  bool have_kept = false;
  static_assert(sizeof(ChainDev) <= 4096, "kernel argument block too large");
  for (int li = 0; li < c.nlayers; ++li) {
    const ChainLayer& L = c.L[li];
    const GemmArgs& g = L.g;
    ChainLayerDev& D = d.L[li];
    int ka = 0, nc = 0, nchunk = 0, ntile = 0;
    if (g.act_a) return fail(AA_ERR_INVALID, "gemm chain: act_a is not supported");
    for (int s = 0; s < g.a.count; ++s) {
      const Seg& sg = g.a.s[s];
      if ((sg.n & 31) || (sg.ld & 3) || (reinterpret_cast<uintptr_t>(sg.p) & 15) || !sg.p)
        return fail(AA_ERR_INVALID, "gemm chain: A segments must be 32-column granular and 16-B aligned");
      for (int k = 0; k < sg.n; k += 32) {
        if (nchunk >= kChainMaxBlocks) return fail(AA_ERR_INVALID, "gemm chain: too many k chunks");
        D.a[nchunk] = static_cast<const float*>(sg.p) + k;
        D.lda[nchunk] = sg.ld;
        ++nchunk;
      }
      ka += sg.n;
    }
    if (g.has_z && g.z.count != g.c.count) return fail(AA_ERR_INVALID, "gemm chain: z must be split like c");
    if (g.has_add && g.add.count != g.c.count) return fail(AA_ERR_INVALID, "gemm chain: add must be split like c");
    for (int s = 0; s < g.c.count; ++s) {
      const Seg& sg = g.c.s[s];
      if ((sg.n & 31) || (sg.ld & 3) || (reinterpret_cast<uintptr_t>(sg.p) & 15))
        return fail(AA_ERR_INVALID, "gemm chain: C segments must be 32-column granular and 16-B aligned");
      if (g.has_z && (g.z.s[s].n != sg.n || (g.z.s[s].ld & 3) || (reinterpret_cast<uintptr_t>(g.z.s[s].p) & 15) || !g.z.s[s].p))
        return fail(AA_ERR_INVALID, "gemm chain: bad z segment");
      if (g.has_add && (g.add.s[s].n != sg.n || (g.add.s[s].ld & 3) || (reinterpret_cast<uintptr_t>(g.add.s[s].p) & 15) || !g.add.s[s].p))
        return fail(AA_ERR_INVALID, "gemm chain: bad add segment");
      for (int k = 0; k < sg.n; k += 32) {
        if (ntile >= kChainMaxBlocks) return fail(AA_ERR_INVALID, "gemm chain: too many output tiles");
        ChainTileDev& t = D.t[ntile];
        t.c = sg.p ? static_cast<float*>(sg.p) + k : nullptr;
        t.ldc = sg.ld;
        t.accum = g.c_accum[s];
        t.z = g.has_z ? static_cast<const float*>(g.z.s[s].p) + k : nullptr;
        t.ldz = g.has_z ? g.z.s[s].ld : 0;
        t.add = g.has_add ? static_cast<const float*>(g.add.s[s].p) + k : nullptr;
        t.ldadd = g.has_add ? g.add.s[s].ld : 0;
        ++ntile;
      }
      nc += sg.n;
    }
    if (ka + (L.use_prev ? 64 : 0) != g.K || nc != g.N || !g.Bq) return fail(AA_ERR_INVALID, "gemm chain: bad layer shape");
    // (a layer that chains from the kept pair may only replace it behind its last tile pair; one that does not may keep any pair)
    if (L.keep_tile >= 0 && ((L.keep_tile & 1) || L.keep_tile * 32 + 64 > g.N || (L.use_prev && L.keep_tile * 32 + 64 != g.N)))
      return fail(AA_ERR_INVALID, "gemm chain: the kept 64 features must be a tile pair (the last one of a layer that chains)");
    if (L.use_prev && !have_kept) return fail(AA_ERR_INVALID, "gemm chain: nothing to chain from");
    have_kept = have_kept || L.keep_tile >= 0;  // (kept features stay available until a later layer replaces them)
    D.Wq = static_cast<const u32x4*>(g.Bq);
    D.KCg = nchunk;
    D.KC = nchunk + (L.use_prev ? 2 : 0);
    D.use_prev = L.use_prev;
    D.NT = ntile;
    D.keep_tile = L.keep_tile;
    D.keep_act = L.keep_act;
    D.a_mode = L.a_mode;
    if (L.a_mode == 2) {
      if (!L.a2_add || !L.a2_z || (L.ld_a2add & 3) || (L.ld_a2z & 3) || L.use_prev || g.a.count != 1 || !(nchunk <= 2 && ntile > 2))
        return fail(AA_ERR_INVALID, "gemm chain: a_mode 2 needs one A segment of <= 2 chunks, > 2 output tiles, and its add / z arrays");
      D.a2_add = static_cast<const float*>(L.a2_add);
      D.a2_z = static_cast<const float*>(L.a2_z);
      D.ld_a2add = L.ld_a2add;
      D.ld_a2z = L.ld_a2z;
    }
    if (L.a_mode == 1) {
      if (!c.ro_w) return fail(AA_ERR_INVALID, "gemm chain: a_mode 1 needs ro_w");
      d.ro_n = std::max(d.ro_n, nchunk * 32);
    }
    D.kept_out = static_cast<float*>(L.kept_out);
    D.ld_kept = L.ld_kept;
    if (L.kept_out && (L.keep_tile < 0 || (L.ld_kept & 3) || (reinterpret_cast<uintptr_t>(L.kept_out) & 15)))
      return fail(AA_ERR_INVALID, "gemm chain: kept_out needs a kept tile pair and 16-B aligned rows");
    D.embrev_out = static_cast<float*>(L.embrev_out);
    if (L.embrev_out && (!c.emb_table || g.N != 64)) return fail(AA_ERR_INVALID, "gemm chain: embrev_out needs the table and a 64-wide layer");
    D.edge_sum_out = static_cast<float*>(L.edge_sum_out);
    if (L.edge_sum_out) {
      if (!c.ro_w || g.N != 64) return fail(AA_ERR_INVALID, "gemm chain: edge_sum_out needs ro_w and a 64-wide layer");
      d.ro_n = std::max(d.ro_n, 64);
    }
  }
  dim3 grid((unsigned)((c.M + 127) / 128));
  bool any_pre = false;
  for (int li = 0; li < d.nlayers; ++li) any_pre = any_pre || (d.L[li].KC <= 2 && d.L[li].NT > 2);
  const size_t smem = sizeof(u32x4) * 2 * kWStep + sizeof(float) * (32 * kChainMaxBlocks + 4 * 32 * kEpLd) +
                      (c.emb_table ? sizeof(float) * 512 * c.num_types * c.num_types : 0);
  bool emb = false;
  for (int li = 0; li < d.nlayers; ++li) emb = emb || d.L[li].embrev_out != nullptr;
#define AA_CHAIN_LAUNCH(P_, E_) hipLaunchKernelGGL((gemm_chain_bf16x3_kernel<P_, E_>), grid, dim3(256), smem, stream, d)
  if (any_pre) {
    if (emb) AA_CHAIN_LAUNCH(true, true); else AA_CHAIN_LAUNCH(true, false);
  } else {
    if (emb) AA_CHAIN_LAUNCH(false, true); else AA_CHAIN_LAUNCH(false, false);
  }
#undef AA_CHAIN_LAUNCH
  AA_CHECK_HIP(hipGetLastError());
  return AA_OK;
}

// element count (32-bit words) of the split weight copy, and the host-side packer (from the fp32 matrix)
size_t gemm_bf16x3_words(int K, int N) { return size_t((N + 31) / 32) * size_t((K + 31) / 32) * 64 * 24; }
void gemm_pack_bf16x3(const float* B, int K, int N, unsigned* out) {
  const int NT = (N + 31) / 32, KC = (K + 31) / 32;
  auto trunc = [](float x) {
    unsigned u;
    memcpy(&u, &x, 4);
    return u & 0xFFFF0000u;
  };
  auto tof = [](unsigned u) {
    float x;
    memcpy(&x, &u, 4);
    return x;
  };
  for (int nt = 0; nt < NT; ++nt)
    for (int kc = 0; kc < KC; ++kc)
      for (int lane = 0; lane < 64; ++lane) {
        unsigned* o = out + (size_t(nt) * KC + kc) * 64 * 24 + size_t(lane) * 4;  // [q = level*2+half][lane][4 words]
        for (int half = 0; half < 2; ++half)
          for (int q = 0; q < 4; ++q) {
            unsigned h[3][2];
            for (int e = 0; e < 2; ++e) {
              int s = half * 8 + q * 2 + e;
              int k = kc * 32 + 8 * (s >> 2) + 4 * (lane >> 5) + (s & 3), n = nt * 32 + (lane & 31);  // accumulator order
              float x = (k < K && n < N) ? B[size_t(k) * N + n] : 0.f;
              h[0][e] = trunc(x);
              float r = x - tof(h[0][e]);
              h[1][e] = trunc(r);
              float r2 = r - tof(h[1][e]);
              h[2][e] = trunc(r2);
            }
            for (int lv = 0; lv < 3; ++lv) o[size_t(lv * 2 + half) * 64 * 4 + q] = (h[lv][0] >> 16) | h[lv][1];
          }
      }
}

size_t gemm_packed_elems(int K, int N) { return size_t((N + 31) / 32) * size_t((K + 31) / 32) * 64 * 16; }

// out[nt][kc][lane][s] = B[kc*32 + (lane>>5)*16 + s][nt*32 + (lane&31)]   (zero padded)
void gemm_pack_b(const double* B, int K, int N, double* out) {
  const int NT = (N + 31) / 32, KC = (K + 31) / 32;
  for (int nt = 0; nt < NT; ++nt)
    for (int kc = 0; kc < KC; ++kc)
      for (int lane = 0; lane < 64; ++lane)
        for (int s = 0; s < 16; ++s) {
          int k = kc * 32 + (lane >> 5) * 16 + s, n = nt * 32 + (lane & 31);
          out[((size_t(nt) * KC + kc) * 64 + lane) * 16 + s] = (k < K && n < N) ? B[size_t(k) * N + n] : 0.0;
        }
}

static bool seglist_frag_ok_host(const SegList& sl) {
  for (int s = 0; s < sl.count; ++s)
    if ((sl.s[s].n & 15) || (sl.s[s].ld & 3) || (reinterpret_cast<uintptr_t>(sl.s[s].p) & 15)) return false;
  return true;
}

static int check_args(const GemmArgs& g) {
  int ka = 0, nc = 0;
  for (int s = 0; s < g.a.count; ++s) ka += g.a.s[s].n;
  for (int s = 0; s < g.c.count; ++s) nc += g.c.s[s].n;
  if (ka != g.K || nc != g.N) return fail(AA_ERR_INVALID, "gemm: segment widths do not sum to K/N");
  if (g.batch > 1 && (g.batch > 16 || g.has_z || g.has_add)) return fail(AA_ERR_INVALID, "gemm: batched launches take up to 16 problems and no z / add operands");
  if (g.act_a && g.act_hi != 0 && ((g.act_lo & 31) || (g.act_hi & 31) || g.act_lo < 0 || g.act_hi <= g.act_lo || g.act_hi > g.K))
    return fail(AA_ERR_INVALID, "gemm: the activated column range of A must be 32-granular and inside [0, K)");
  if (g.has_add) {
    if (g.add.count != g.c.count) return fail(AA_ERR_INVALID, "gemm: add/c segment mismatch");
    for (int s = 0; s < g.c.count; ++s)
      if (g.add.s[s].n != g.c.s[s].n) return fail(AA_ERR_INVALID, "gemm: add/c segment mismatch");
  }
  if (g.has_z) {
    if (g.z.count != g.c.count) return fail(AA_ERR_INVALID, "gemm: z/c segment mismatch");
    for (int s = 0; s < g.c.count; ++s)
      if (g.z.s[s].n != g.c.s[s].n) return fail(AA_ERR_INVALID, "gemm: z/c segment mismatch");
  }
  return AA_OK;
}

// batched problem set on a kernel without the batched form: one launch per problem on shifted views
template <typename T>
static int launch_gemm_each(const GemmArgs& g, hipStream_t stream) {
  for (int b = 0; b < g.batch; ++b) {
    GemmArgs s = g;
    s.batch = 0;
    const int sel = int((g.bsel4 >> (4 * b)) & 15ull);
    for (int q = 0; q < s.a.count; ++q) s.a.s[q].p = static_cast<T*>(s.a.s[q].p) + b * g.a_bs;
    for (int q = 0; q < s.c.count; ++q)
      if (s.c.s[q].p) s.c.s[q].p = static_cast<T*>(s.c.s[q].p) + b * g.c_bs;
    s.B = static_cast<const T*>(g.B) + sel * g.b_bs;
    if (g.Bp) s.Bp = static_cast<const T*>(g.Bp) + sel * g.bp_bs;
    if (g.Bq) s.Bq = static_cast<const T*>(g.Bq) + sel * g.bq_bs;
    if (int rc = launch_gemm<T>(s, stream)) return rc;
  }
  return AA_OK;
}

template <>
int launch_gemm<float>(const GemmArgs& g, hipStream_t stream) {
  if (g.M == 0) return AA_OK;
  if (int rc = check_args(g)) return rc;
  const bool v1_only = g.opt_v1 != 0;
  if (g.force_kernel == 3 || g.act_kind != AA_ACT_SILU) {
    if (g.batch > 1) return launch_gemm_each<float>(g, stream);
    dim3 grid((unsigned)((g.M + GV_BM - 1) / GV_BM), (unsigned)((g.N + GV_BN - 1) / GV_BN));
    size_t smem = sizeof(float) * (GV_BK * GV_LDA + GV_BK * GV_BN);
    hipLaunchKernelGGL(gemm_valu_kernel<float>, grid, dim3(256), smem, stream, g);
  } else if (g.Bp && !v1_only && seglist_frag_ok_host(g.a)) {
    dim3 grid((unsigned)((g.M + 127) / 128));
    // 16-B epilogue accesses need every C/Z segment to be 4-column granular and 16-B aligned
    int vec_ok = 1;
    for (int s = 0; s < g.c.count; ++s) {
      if ((g.c.s[s].n & 3) || (g.c.s[s].ld & 3) || (reinterpret_cast<uintptr_t>(g.c.s[s].p) & 15)) vec_ok = 0;
      if (g.has_z && ((g.z.s[s].ld & 3) || (reinterpret_cast<uintptr_t>(g.z.s[s].p) & 15))) vec_ok = 0;
      if (g.has_add && ((g.add.s[s].ld & 3) || (reinterpret_cast<uintptr_t>(g.add.s[s].p) & 15))) vec_ok = 0;
    }
    const int KC = (g.K + 31) / 32;
    const bool direct_epi = g.opt_lds_epilogue == 0;  // default: direct (the LDS-transposed variant measured slower)
    bool seg32 = true;
    for (int q = 0; q < g.a.count; ++q) seg32 = seg32 && (g.a.s[q].n % 32 == 0);
    if (g.Bq && g.force_kernel != 1 && seg32) {
      const u32x4* Wq = static_cast<const u32x4*>(g.Bq);
      const bool lds = vec_ok && !direct_epi;
      if (g.batch > 1) {
        if (lds) return launch_gemm_each<float>(g, stream);
        grid.z = unsigned(g.batch);  // (the direct-epilogue kernel has the batched form)
      }
      const size_t smem = lds ? sizeof(float) * 4 * 32 * EP_LD : 0;
#define AA_LAUNCH_BF16(KCR)                                                                                   \
  if (lds)                                                                                                    \
    hipLaunchKernelGGL((gemm_bf16x3_kernel<KCR, true>), grid, dim3(256), smem, stream, g, Wq, vec_ok);        \
  else                                                                                                        \
    hipLaunchKernelGGL((gemm_bf16x3_kernel<KCR, false>), grid, dim3(256), smem, stream, g, Wq, vec_ok);
      if (KC <= 2) {
        AA_LAUNCH_BF16(2)
      } else if (KC <= 4) {
        AA_LAUNCH_BF16(4)
      } else {
        AA_LAUNCH_BF16(0)
      }
#undef AA_LAUNCH_BF16
    } else if (g.batch > 1)
      return launch_gemm_each<float>(g, stream);
    else if (KC <= 2)
      hipLaunchKernelGGL(gemm_mfma_f32_v3_kernel<2>, grid, dim3(256), 0, stream, g, vec_ok);
    else if (KC <= 4)
      hipLaunchKernelGGL(gemm_mfma_f32_v3_kernel<4>, grid, dim3(256), 0, stream, g, vec_ok);
    else
      hipLaunchKernelGGL(gemm_mfma_f32_v3_kernel<0>, grid, dim3(256), 0, stream, g, vec_ok);
  } else {
    if (g.batch > 1) return launch_gemm_each<float>(g, stream);
    dim3 grid((unsigned)((g.M + GM_BM - 1) / GM_BM), (unsigned)((g.N + GM_BN - 1) / GM_BN));
    size_t smem = sizeof(float) * (GM_BK * GM_LDA + GM_BK * GM_BN);
    hipLaunchKernelGGL(gemm_mfma_f32_kernel, grid, dim3(256), smem, stream, g);
  }
  AA_CHECK_HIP(hipGetLastError());
  return AA_OK;
}

template <>
int launch_gemm<double>(const GemmArgs& g, hipStream_t stream) {
  if (g.M == 0) return AA_OK;
  if (int rc = check_args(g)) return rc;
  dim3 grid((unsigned)((g.M + GV_BM - 1) / GV_BM), (unsigned)((g.N + GV_BN - 1) / GV_BN));
  size_t smem = sizeof(double) * (GV_BK * GV_LDA + GV_BK * GV_BN);
  bool pipe_ok = (g.N % 2) == 0 && (reinterpret_cast<uintptr_t>(g.B) & 15) == 0;
  for (int s2 = 0; s2 < g.a.count; ++s2)
    pipe_ok = pipe_ok && (g.a.s[s2].n % 16) == 0 && (g.a.s[s2].ld % 2) == 0 && (reinterpret_cast<uintptr_t>(g.a.s[s2].p) & 15) == 0;
  bool epilogue_reads = g.has_z || g.has_add;  // operands the epilogue has to FETCH (z, add, accumulated-into C)
  for (int s2 = 0; s2 < g.c.count; ++s2) {
    pipe_ok = pipe_ok && (g.c.s[s2].n % 16) == 0;
    epilogue_reads = epilogue_reads || g.c_accum[s2] != 0;
  }
  const bool rows_ok = pipe_ok && g.opt_f64_column_loop == 0 && g.opt_f64_rows != 2 && (g.K % 16) == 0 &&
                       (g.opt_f64_rows == 1 ? (g.N <= 128 || g.K <= 128) : (g.N <= 128 || (g.K <= 128 && !epilogue_reads)));
  if (g.batch > 1 && (g.force_kernel == 3 || g.act_kind != AA_ACT_SILU || !rows_ok)) return launch_gemm_each<double>(g, stream);  // (only the row-resident kernels have the batched form)
  if (g.force_kernel == 3 || g.act_kind != AA_ACT_SILU) {
    hipLaunchKernelGGL(gemm_valu_kernel<double>, grid, dim3(256), smem, stream, g);
  } else if (pipe_ok && g.opt_f64_column_loop == 0 && g.opt_f64_rows != 2 && (g.K % 16) == 0 &&
             (g.opt_f64_rows == 1 ? (g.N <= 128 || g.K <= 128) : (g.N <= 128 || (g.K <= 128 && !epilogue_reads)))) {
    // row-resident kernels: every operand row is read from HBM once.  Measured at C5 (1.7 M rows; profiles/archive/r03_*_stages_c5.log,
    // after the epilogue loads were batched and the operand activation deferred): the accumulator-resident form (N <= 128)
    // wins everywhere, 54-58 vs 45-52 TFLOP/s for K > 128 and 5-10 % on 128 x 128 layers with or without z / add operands;
    // the operand-resident form (N > 128, K <= 128) wins 5-8 % on plain layers and loses 25-30 % where the epilogue fetches
    // z / add operands or accumulates into C (its two waves per SIMD overlap the silu' arithmetic of one pass with the next pass's MFMAs worse
    // than the staged kernel's three) -- those keep the staged kernel unless forced (aa_plan_options.f64_rows = 1).
    dim3 gridr((unsigned)((g.M + 127) / 128), 1, g.batch > 1 ? unsigned(g.batch) : 1u);
    if (g.N <= 128) {
      const size_t smemr = sizeof(double) * 2 * 16 * (128 + 4);
      hipLaunchKernelGGL(gemm_f64_rows_kernel<false>, gridr, dim3(256), smemr, stream, g);
    } else {
      const size_t smemr = sizeof(double) * 2 * 16 * (64 + 4);
      hipLaunchKernelGGL(gemm_f64_rows_kernel<true>, gridr, dim3(256), smemr, stream, g);
    }
  } else if (pipe_ok) {
    dim3 grid6((unsigned)((g.M + G6_BM - 1) / G6_BM), (unsigned)((g.N + G6_BN - 1) / G6_BN));
    // enough row tiles to fill the chip on their own: one workgroup per row tile looping over the column tiles
    // (aa_plan_options.f64_column_loop: 1 never, 2 always -- tests)
    if (g.opt_f64_column_loop == 2 || (g.opt_f64_column_loop == 0 && grid6.x >= 2048)) grid6.y = 1;
    const size_t smem6 = sizeof(double) * 2 * (G6_BK * G6_LDA + G6_BK * G6_BN);
    hipLaunchKernelGGL(gemm_mfma_f64_pipe_kernel, grid6, dim3(256), smem6, stream, g);
  } else {
    hipLaunchKernelGGL(gemm_mfma_f64_kernel, grid, dim3(256), smem, stream, g);
  }
  AA_CHECK_HIP(hipGetLastError());
  return AA_OK;
}

}  // namespace aa

// ---------------------------------------------------------------------------------------------
// aa_linear_forward: one bias-free linear layer y = x @ W along the edges with W a DEVICE matrix that changes every call (the
// training path: the optimiser has just updated it).  Two small launches put W into the fragment orders of gemm_pack_b /
// gemm_pack_bf16x3 (same arithmetic as the host packers: three truncated bf16 levels), then the rows run through the bf16x3 kernel
// of the inference pipeline.
// ---------------------------------------------------------------------------------------------
namespace aa {
namespace {
// B[k][n] = W[k * ldk + n * ldn] (any strides: W or its transpose without a copy)
__global__ __launch_bounds__(256) void pack_b_device_kernel(const float* __restrict__ W, int64_t ldk, int64_t ldn, int K, int N, float* __restrict__ bp,
                                                            unsigned* __restrict__ bq) {
  const int NT = (N + 31) / 32, KC = (K + 31) / 32;
  const int item = blockIdx.x * 256 + threadIdx.x;  // (nt, kc, lane, half, q): 8 (half, q) per lane
  if (item >= NT * KC * 64 * 8) return;
  const int q = item & 3, half = (item >> 2) & 1, lane = (item >> 3) & 63, tile = item >> 9;
  const int kc = tile % KC, nt = tile / KC;
  const int n = nt * 32 + (lane & 31);
  unsigned h[3][2];
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const int sidx = half * 8 + q * 2 + e;
    const int k = kc * 32 + 8 * (sidx >> 2) + 4 * (lane >> 5) + (sidx & 3);  // accumulator order
    const float x = (k < K && n < N) ? W[k * ldk + n * ldn] : 0.f;
    const unsigned u0 = __builtin_bit_cast(unsigned, x) & 0xFFFF0000u;
    const float r = x - __builtin_bit_cast(float, u0);
    const unsigned u1 = __builtin_bit_cast(unsigned, r) & 0xFFFF0000u;
    const float r2 = r - __builtin_bit_cast(float, u1);
    h[0][e] = u0;
    h[1][e] = u1;
    h[2][e] = __builtin_bit_cast(unsigned, r2) & 0xFFFF0000u;
    // the fp32 fragment order (gemm_pack_b): element s of the lane = row kc * 32 + (lane >> 5) * 16 + s
    const int s2 = half * 8 + q * 2 + e, k2 = kc * 32 + (lane >> 5) * 16 + s2;
    bp[(size_t(tile) * 64 + lane) * 16 + s2] = (k2 < K && n < N) ? W[k2 * ldk + n * ldn] : 0.f;
  }
  unsigned* o = bq + size_t(tile) * 64 * 24 + size_t(lane) * 4;
#pragma unroll
  for (int lv = 0; lv < 3; ++lv) o[size_t(lv * 2 + half) * 64 * 4 + q] = (h[lv][0] >> 16) | h[lv][1];
}
}  // namespace
}  // namespace aa

extern "C" size_t aa_linear_forward_workspace_bytes(int K, int N) {
  if (K < 1 || N < 1) return 0;
  return (aa::gemm_packed_elems(K, N) + aa::gemm_bf16x3_words(K, N)) * 4;
}

extern "C" int aa_linear_forward(int64_t E, int K, int N, const float* x, int64_t ldx, const float* W, int64_t ldk, int64_t ldn, void* workspace,
                                 size_t workspace_bytes, float* out, int64_t ldo, aa_stream stream) {
  using namespace aa;
  AA_REQUIRE(E >= 0 && K >= 32 && N >= 32 && (K % 32) == 0 && (N % 32) == 0, "aa_linear_forward: K and N must be multiples of 32");
  if (E == 0) return AA_OK;
  AA_REQUIRE(x && W && out && workspace, "aa_linear_forward: null argument");
  AA_REQUIRE(workspace_bytes >= aa_linear_forward_workspace_bytes(K, N), "aa_linear_forward: workspace too small");
  AA_REQUIRE(ldx >= K && ldo >= N && (ldx % 4) == 0 && (ldo % 4) == 0 && ldx <= INT32_MAX && ldo <= INT32_MAX,
             "aa_linear_forward: row strides must be multiples of 4 elements and at least the row length");
  AA_REQUIRE(reinterpret_cast<uintptr_t>(x) % 16 == 0 && reinterpret_cast<uintptr_t>(out) % 16 == 0 && reinterpret_cast<uintptr_t>(workspace) % 16 == 0,
             "aa_linear_forward: x, out and the workspace must be 16-byte aligned");
  hipStream_t s = static_cast<hipStream_t>(stream);
  float* bp = static_cast<float*>(workspace);
  unsigned* bq = reinterpret_cast<unsigned*>(bp + gemm_packed_elems(K, N));
  const int items = ((N + 31) / 32) * ((K + 31) / 32) * 64 * 8;
  hipLaunchKernelGGL(pack_b_device_kernel, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, s, W, ldk, ldn, K, N, bp, bq);
  AA_CHECK_HIP(hipGetLastError());
  GemmArgs g{};
  g.M = E;
  g.K = K;
  g.N = N;
  g.a = SegList{1, {Seg{const_cast<float*>(x), int(ldx), K}}};
  g.c = SegList{1, {Seg{out, int(ldo), N}}};
  g.B = nullptr;  // (only the VALU kernel reads the plain matrix, and this shape never takes it)
  g.Bp = bp;
  g.Bq = bq;
  g.act_kind = AA_ACT_SILU;
  return launch_gemm<float>(g, s);
}

// ---------------------------------------------------------------------------------------------
// test hook (include/allegro_amd.h, "debug entry points"): one plain linear layer C = A @ W through a chosen fp32
// GEMM kernel, so that the split-precision arithmetic can be bounded directly against an fp64 product
// ---------------------------------------------------------------------------------------------
extern "C" int aa_debug_gemm_f32(int kernel, int64_t M, int K, int N, const float* A_dev, const float* W_host,
                                 float* C_dev, aa_stream stream) {
  using namespace aa;
  AA_REQUIRE(A_dev && W_host && C_dev && M >= 0 && K > 0 && N > 0, "aa_debug_gemm_f32: bad argument");
  AA_REQUIRE(kernel >= 0 && kernel <= 3, "aa_debug_gemm_f32: kernel must be 0 (bf16x3), 1 (fp32 MFMA), 2 (fused chain), 3 (VALU)");
  AA_REQUIRE((K % 32) == 0 && (N % 32) == 0, "aa_debug_gemm_f32: K and N must be multiples of 32");
  hipStream_t s = static_cast<hipStream_t>(stream);
  const size_t nw = size_t(K) * N, np = gemm_packed_elems(K, N), nq = gemm_bf16x3_words(K, N);
  std::vector<double> wd(W_host, W_host + nw), wpd(np);
  gemm_pack_b(wd.data(), K, N, wpd.data());
  std::vector<float> wp(wpd.begin(), wpd.end());
  std::vector<unsigned> wq(nq);
  gemm_pack_bf16x3(W_host, K, N, wq.data());
  void *dW = nullptr, *dWp = nullptr, *dWq = nullptr;
  AA_CHECK_HIP(hipMalloc(&dW, nw * 4));
  AA_CHECK_HIP(hipMalloc(&dWp, np * 4));
  AA_CHECK_HIP(hipMalloc(&dWq, nq * 4));
  auto cleanup = [&]() {
    (void)hipFree(dW);
    (void)hipFree(dWp);
    (void)hipFree(dWq);
  };
  int rc = AA_OK;
  if (hipMemcpy(dW, W_host, nw * 4, hipMemcpyHostToDevice) != hipSuccess || hipMemcpy(dWp, wp.data(), np * 4, hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy(dWq, wq.data(), nq * 4, hipMemcpyHostToDevice) != hipSuccess) {
    cleanup();
    return fail(AA_ERR_HIP, "aa_debug_gemm_f32: upload failed");
  }
  SegList a{1, {Seg{const_cast<float*>(A_dev), K, K}}}, c{1, {Seg{C_dev, N, N}}};
  if (kernel == 2) {
    ChainArgs ca{};
    ca.M = M;
    ca.nlayers = 1;
    ca.L[0].g.M = M;
    ca.L[0].g.K = K;
    ca.L[0].g.N = N;
    ca.L[0].g.a = a;
    ca.L[0].g.Bq = dWq;
    ca.L[0].g.c = c;
    ca.L[0].keep_tile = -1;
    rc = launch_gemm_chain(ca, s);
  } else {
    GemmArgs g{};
    g.M = M;
    g.K = K;
    g.N = N;
    g.a = a;
    g.B = dW;
    g.Bp = dWp;
    g.Bq = dWq;
    g.c = c;
    g.force_kernel = kernel;
    rc = launch_gemm<float>(g, s);
  }
  if (rc == AA_OK && hipStreamSynchronize(s) != hipSuccess) rc = fail(AA_ERR_HIP, "aa_debug_gemm_f32: kernel failed");
  cleanup();
  return rc;
}
