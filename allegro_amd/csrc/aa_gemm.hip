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

namespace aa {

typedef float v16f __attribute__((ext_vector_type(16)));

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

template <typename T>
__device__ __forceinline__ void seg_store(const GemmArgs& g, int64_t row, int col, T v) {
  int c = col;
#pragma unroll
  for (int s = 0; s < 3; ++s) {
    if (s < g.c.count) {
      if (c < g.c.s[s].n) {
        if (g.has_z) {
          T z = static_cast<const T*>(g.z.s[s].p)[row * g.z.s[s].ld + c];
          v *= dsilu(z);
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
        if (g.act_a) v = silu(v);
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
        if (g.act_a) v = silu(v);
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

static int check_args(const GemmArgs& g) {
  int ka = 0, nc = 0;
  for (int s = 0; s < g.a.count; ++s) ka += g.a.s[s].n;
  for (int s = 0; s < g.c.count; ++s) nc += g.c.s[s].n;
  if (ka != g.K || nc != g.N) return fail(AA_ERR_INVALID, "gemm: segment widths do not sum to K/N");
  if (g.has_z) {
    if (g.z.count != g.c.count) return fail(AA_ERR_INVALID, "gemm: z/c segment mismatch");
    for (int s = 0; s < g.c.count; ++s)
      if (g.z.s[s].n != g.c.s[s].n) return fail(AA_ERR_INVALID, "gemm: z/c segment mismatch");
  }
  return AA_OK;
}

static bool force_valu() {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("AA_GEMM_VALU");
    v = (e && e[0] == '1') ? 1 : 0;
  }
  return v == 1;
}

template <>
int launch_gemm<float>(const GemmArgs& g, hipStream_t stream) {
  if (g.M == 0) return AA_OK;
  if (int rc = check_args(g)) return rc;
  if (force_valu()) {
    dim3 grid((unsigned)((g.M + GV_BM - 1) / GV_BM), (unsigned)((g.N + GV_BN - 1) / GV_BN));
    size_t smem = sizeof(float) * (GV_BK * GV_LDA + GV_BK * GV_BN);
    hipLaunchKernelGGL(gemm_valu_kernel<float>, grid, dim3(256), smem, stream, g);
  } else {
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
  hipLaunchKernelGGL(gemm_valu_kernel<double>, grid, dim3(256), smem, stream, g);
  AA_CHECK_HIP(hipGetLastError());
  return AA_OK;
}

}  // namespace aa
