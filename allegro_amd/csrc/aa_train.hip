// Training-path kernels that the inference pipeline has no use for.
//
// The reference trains every ScalarMLPFunction through autograd (allegro/nn/_allegro.py:192-213): per linear layer the
// backward needs d W = x^T g, a product whose REDUCTION runs over the edges (K = E ~ 10^5..10^7, output 64..512 wide).  Library
// GEMMs pick a tile for the tiny output and walk the whole reduction in one or two workgroups (measured on MI355X: 0.73 ms for
// [64 x 298 144] @ [298 144 x 64], 3 TFLOP/s, 210 GB/s -- profiles/r05_v6_train_c3_kernel_stats.txt); here the edges are cut
// into slabs, one wave per slab and 64 x 64 output block, exact-fp32 / fp64 products on the matrix cores
// (v_mfma_f32_32x32x2_f32 / v_mfma_f64_16x16x4_f64: no split-precision needed, the kernel is HBM-bound), and the slabs are
// summed in a fixed order by a second small kernel: bit-reproducible, no atomics.
//
// The other half of this file is MakeWeightedChannels (allegro/nn/_strided/_channels.py:44-63) as a bilinear form
// B(sh, w)[e,c,i] = sh[e,i] w[e,c,r(i)] with its two partial contractions -- three kernels that are closed under differentiation
// (allegro_amd/ops.py: weighted_channels), each ONE pass over the [E,u,D] tensor instead of the expand / cat / mul / sum chains
// of eager autograd.
#include "aa_common.h"

namespace aa {
namespace {

typedef float v16f_t __attribute__((ext_vector_type(16)));
typedef double v4d_t __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------------------------------------------------
// d W = x^T g
// ---------------------------------------------------------------------------------------------------------------------
struct WgradArgs {
  int64_t E;
  int K, N;
  const void* x;
  const void* g;
  int64_t ldx, ldg;
  void* partial;  // [slabs][K][N]
  int64_t rows_per_slab;
  int slabs;
};

// fp32: a wave owns a 64 x 64 block of the output = 2 x 2 tiles of v_mfma_f32_32x32x2_f32 (A[i = l & 31][kk = l >> 5] = x[e + kk][k0 + i],
// B[kk][j = l & 31] = g[e + kk][n0 + j]: lanes 0-31 read 128 contiguous bytes of row e, lanes 32-63 of row e + 1)
__global__ __launch_bounds__(256) void wgrad_f32_kernel(WgradArgs a) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int slab = blockIdx.x * 4 + wv;
  if (slab >= a.slabs) return;
  const int k0 = blockIdx.y * 64, n0 = blockIdx.z * 64;
  const int i = lane & 31, kk = lane >> 5;
  const float* x = static_cast<const float*>(a.x);
  const float* g = static_cast<const float*>(a.g);
  const int64_t e0 = int64_t(slab) * a.rows_per_slab;
  const int64_t e1 = e0 + a.rows_per_slab < a.E ? e0 + a.rows_per_slab : a.E;
  // columns beyond K / N read a clamped address and contribute zero
  int kc[2], nc[2];
  float km[2], nm[2];
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    const int k = k0 + 32 * b + i, n = n0 + 32 * b + i;
    kc[b] = k < a.K ? k : a.K - 1;
    nc[b] = n < a.N ? n : a.N - 1;
    km[b] = k < a.K ? 1.f : 0.f;
    nm[b] = n < a.N ? 1.f : 0.f;
  }
  v16f_t acc[2][2];
#pragma unroll
  for (int p = 0; p < 2; ++p)
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[p][q][r] = 0.f;
  constexpr int UN = 8;  // edge pairs in flight
  for (int64_t e = e0; e < e1; e += 2 * UN) {
    float xv[UN][2], gv[UN][2];
#pragma unroll
    for (int s = 0; s < UN; ++s) {
      const int64_t row = e + 2 * s + kk;
      const int64_t rc = row < e1 ? row : e1 - 1;
      const float m = row < e1 ? 1.f : 0.f;
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        xv[s][b] = x[rc * a.ldx + kc[b]] * (m * km[b]);
        gv[s][b] = g[rc * a.ldg + nc[b]] * nm[b];
      }
    }
#pragma unroll
    for (int s = 0; s < UN; ++s)
#pragma unroll
      for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int q = 0; q < 2; ++q) acc[p][q] = __builtin_amdgcn_mfma_f32_32x32x2f32(xv[s][p], gv[s][q], acc[p][q], 0, 0, 0);
  }
  // D[row = (r & 3) + 8 (r >> 2) + 4 (l >> 5)][col = l & 31]
  float* out = static_cast<float*>(a.partial) + int64_t(slab) * a.K * a.N;
#pragma unroll
  for (int p = 0; p < 2; ++p)
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int n = n0 + 32 * q + i;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int k = k0 + 32 * p + (r & 3) + 8 * (r >> 2) + 4 * kk;
        if (k < a.K && n < a.N) out[int64_t(k) * a.N + n] = acc[p][q][r];
      }
    }
}

// fp64: 4 x 4 tiles of v_mfma_f64_16x16x4_f64 (A[i = l & 15][kk = l >> 4], B[kk][j = l & 15], D[4 r + (l >> 4)][l & 15]): four rows per step
__global__ __launch_bounds__(256) void wgrad_f64_kernel(WgradArgs a) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int slab = blockIdx.x * 4 + wv;
  if (slab >= a.slabs) return;
  const int k0 = blockIdx.y * 64, n0 = blockIdx.z * 64;
  const int i = lane & 15, kk = lane >> 4;
  const double* x = static_cast<const double*>(a.x);
  const double* g = static_cast<const double*>(a.g);
  const int64_t e0 = int64_t(slab) * a.rows_per_slab;
  const int64_t e1 = e0 + a.rows_per_slab < a.E ? e0 + a.rows_per_slab : a.E;
  int kc[4], nc[4];
  double km[4], nm[4];
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    const int k = k0 + 16 * b + i, n = n0 + 16 * b + i;
    kc[b] = k < a.K ? k : a.K - 1;
    nc[b] = n < a.N ? n : a.N - 1;
    km[b] = k < a.K ? 1.0 : 0.0;
    nm[b] = n < a.N ? 1.0 : 0.0;
  }
  v4d_t acc[4][4];
#pragma unroll
  for (int p = 0; p < 4; ++p)
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[p][q][r] = 0.0;
  constexpr int UN = 2;
  for (int64_t e = e0; e < e1; e += 4 * UN) {
    double xv[UN][4], gv[UN][4];
#pragma unroll
    for (int s = 0; s < UN; ++s) {
      const int64_t row = e + 4 * s + kk;
      const int64_t rc = row < e1 ? row : e1 - 1;
      const double m = row < e1 ? 1.0 : 0.0;
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        xv[s][b] = x[rc * a.ldx + kc[b]] * (m * km[b]);
        gv[s][b] = g[rc * a.ldg + nc[b]] * nm[b];
      }
    }
#pragma unroll
    for (int s = 0; s < UN; ++s)
#pragma unroll
      for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[p][q] = __builtin_amdgcn_mfma_f64_16x16x4f64(xv[s][p], gv[s][q], acc[p][q], 0, 0, 0);
  }
  double* out = static_cast<double*>(a.partial) + int64_t(slab) * a.K * a.N;
#pragma unroll
  for (int p = 0; p < 4; ++p)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int n = n0 + 16 * q + i;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int k = k0 + 16 * p + 4 * r + kk;
        if (k < a.K && n < a.N) out[int64_t(k) * a.N + n] = acc[p][q][r];
      }
    }
}

// out[k][n] = sum over slabs, in slab order
template <typename T>
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const T* partial, int slabs, int64_t KN, T* out) {
  const int64_t idx = int64_t(blockIdx.x) * 256 + threadIdx.x;
  if (idx >= KN) return;
  T s = T(0);
  for (int b = 0; b < slabs; ++b) s += partial[int64_t(b) * KN + idx];
  out[idx] = s;
}

int wgrad_slabs(int64_t E, int K, int N) {
  // ~1024 waves over the whole launch, at least 256 rows per slab (a slab's partial result is K x N elements of traffic)
  const int64_t blocks = int64_t((K + 63) / 64) * ((N + 63) / 64);
  int64_t s = std::max<int64_t>(16, 1024 / blocks);
  s = std::min<int64_t>(s, std::max<int64_t>(1, (E + 255) / 256));
  return int(s);
}

// ---------------------------------------------------------------------------------------------------------------------
// weighted channels: B(sh, w)[e,c,i] = sh[e,i] w[e,c,r(i)]  (R == 1: one weight per channel, shared by all irreps)
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int irrep_of(int i) { return i < 1 ? 0 : (i < 4 ? 1 : (i < 9 ? 2 : 3)); }

template <typename T>
__global__ __launch_bounds__(256) void wc_forward_kernel(int64_t total, int u, int D, int R, const T* sh, const T* w, T* out) {
  const int64_t idx = int64_t(blockIdx.x) * 256 + threadIdx.x;  // (e, c, i)
  if (idx >= total) return;
  const int i = int(idx % D);
  const int64_t ec = idx / D;
  const int64_t e = ec / u;
  out[idx] = sh[e * D + i] * w[ec * R + (R == 1 ? 0 : irrep_of(i))];
}

// gw[e,c,r] = sum_{i in r} g[e,c,i] sh[e,i]
template <typename T>
__global__ __launch_bounds__(256) void wc_grad_w_kernel(int64_t EC, int u, int D, int R, const T* g, const T* sh, T* gw) {
  const int64_t ec = int64_t(blockIdx.x) * 256 + threadIdx.x;
  if (ec >= EC) return;
  const int64_t e = ec / u;
  const T* gr = g + ec * D;
  const T* y = sh + e * D;
  T acc[4] = {T(0), T(0), T(0), T(0)};
  for (int i = 0; i < D; ++i) acc[R == 1 ? 0 : irrep_of(i)] += gr[i] * y[i];
  for (int r = 0; r < R; ++r) gw[ec * R + r] = acc[r];
}

// gsh[e,i] = sum_c g[e,c,i] w[e,c,r(i)]: one wave per edge, lanes over channels, D wave sums
template <typename T>
__global__ __launch_bounds__(256) void wc_grad_sh_kernel(int64_t E, int u, int D, int R, const T* g, const T* w, T* gsh) {
  const int lane = threadIdx.x & 63;
  const int64_t e = int64_t(blockIdx.x) * 4 + (threadIdx.x >> 6);
  if (e >= E) return;
  T acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = T(0);
  for (int c = lane; c < u; c += 64) {
    const T* gr = g + (e * u + c) * D;
    const T* wr = w + (e * u + c) * R;
#pragma unroll
    for (int i = 0; i < 16; ++i)
      if (i < D) acc[i] += gr[i] * wr[R == 1 ? 0 : irrep_of(i)];
  }
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    if (i < D) {
      T v = acc[i];
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
      if (lane == 0) gsh[e * D + i] = v;
    }
  }
}

}  // namespace
}  // namespace aa

extern "C" size_t aa_linear_wgrad_workspace_bytes(aa_dtype dtype, int64_t E, int K, int N) {
  if (E < 0 || K < 1 || N < 1) return 0;
  return size_t(aa::wgrad_slabs(E, K, N)) * size_t(K) * size_t(N) * (dtype == AA_F32 ? 4 : 8);
}

extern "C" int aa_linear_wgrad(aa_dtype dtype, int64_t E, int K, int N, const void* x, int64_t ldx, const void* g, int64_t ldg,
                               void* workspace, size_t workspace_bytes, void* out, aa_stream stream) {
  AA_REQUIRE(K >= 1 && N >= 1 && E >= 0 && out, "aa_linear_wgrad: bad shape");
  AA_REQUIRE(ldx >= K && ldg >= N, "aa_linear_wgrad: row strides shorter than the rows");
  hipStream_t s = static_cast<hipStream_t>(stream);
  const size_t esize = dtype == AA_F32 ? 4 : 8;
  if (E == 0) {
    AA_CHECK_HIP(hipMemsetAsync(out, 0, size_t(K) * N * esize, s));
    return AA_OK;
  }
  AA_REQUIRE(x && g && workspace, "aa_linear_wgrad: null argument");
  AA_REQUIRE(workspace_bytes >= aa_linear_wgrad_workspace_bytes(dtype, E, K, N), "aa_linear_wgrad: workspace too small");
  aa::WgradArgs a{};
  a.E = E;
  a.K = K;
  a.N = N;
  a.x = x;
  a.g = g;
  a.ldx = ldx;
  a.ldg = ldg;
  a.partial = workspace;
  a.slabs = aa::wgrad_slabs(E, K, N);
  const int64_t rows = (E + a.slabs - 1) / a.slabs;
  a.rows_per_slab = (rows + 3) / 4 * 4;
  a.slabs = int((E + a.rows_per_slab - 1) / a.rows_per_slab);
  dim3 grid((unsigned)((a.slabs + 3) / 4), (unsigned)((K + 63) / 64), (unsigned)((N + 63) / 64));
  const int64_t KN = int64_t(K) * N;
  if (dtype == AA_F32) {
    hipLaunchKernelGGL(aa::wgrad_f32_kernel, grid, dim3(256), 0, s, a);
    hipLaunchKernelGGL(aa::wgrad_reduce_kernel<float>, dim3((unsigned)((KN + 255) / 256)), dim3(256), 0, s, static_cast<const float*>(workspace),
                       a.slabs, KN, static_cast<float*>(out));
  } else {
    hipLaunchKernelGGL(aa::wgrad_f64_kernel, grid, dim3(256), 0, s, a);
    hipLaunchKernelGGL(aa::wgrad_reduce_kernel<double>, dim3((unsigned)((KN + 255) / 256)), dim3(256), 0, s, static_cast<const double*>(workspace),
                       a.slabs, KN, static_cast<double*>(out));
  }
  AA_CHECK_HIP(hipGetLastError());
  return AA_OK;
}

template <typename T>
static int wc_launch(int which, int64_t E, int u, int D, int R, const void* p0, const void* p1, void* out, hipStream_t s) {
  if (E == 0) return AA_OK;
  const int64_t EC = E * u;
  if (which == 0) {
    const int64_t total = EC * D;
    hipLaunchKernelGGL(aa::wc_forward_kernel<T>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, total, u, D, R, static_cast<const T*>(p0),
                       static_cast<const T*>(p1), static_cast<T*>(out));
  } else if (which == 1) {
    hipLaunchKernelGGL(aa::wc_grad_w_kernel<T>, dim3((unsigned)((EC + 255) / 256)), dim3(256), 0, s, EC, u, D, R, static_cast<const T*>(p0),
                       static_cast<const T*>(p1), static_cast<T*>(out));
  } else {
    hipLaunchKernelGGL(aa::wc_grad_sh_kernel<T>, dim3((unsigned)((E + 3) / 4)), dim3(256), 0, s, E, u, D, R, static_cast<const T*>(p0),
                       static_cast<const T*>(p1), static_cast<T*>(out));
  }
  AA_CHECK_HIP(hipGetLastError());
  return AA_OK;
}

extern "C" int aa_weighted_channels(aa_dtype dtype, int which, int64_t E, int u, int l_max, int shared, const void* a, const void* b, void* out,
                                    aa_stream stream) {
  AA_REQUIRE(which >= 0 && which <= 2 && E >= 0 && u >= 1 && l_max >= 0 && l_max <= 3, "aa_weighted_channels: bad argument");
  AA_REQUIRE(E == 0 || (a && b && out), "aa_weighted_channels: null argument");
  AA_REQUIRE(E * int64_t(u) * 16 < (int64_t(1) << 40), "aa_weighted_channels: too large");
  const int D = (l_max + 1) * (l_max + 1), R = shared ? 1 : l_max + 1;
  hipStream_t s = static_cast<hipStream_t>(stream);
  return dtype == AA_F32 ? wc_launch<float>(which, E, u, D, R, a, b, out, s) : wc_launch<double>(which, E, u, D, R, a, b, out, s);
}
