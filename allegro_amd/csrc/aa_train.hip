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

// The four waves of a workgroup take four consecutive slabs of rows and the same 64 x 64 output block; their accumulators are
// combined through LDS in a fixed order ((w0 + w2) + (w1 + w3)) and ONE partial block per workgroup goes to the workspace.
//
// fp32: 2 x 2 tiles of v_mfma_f32_32x32x2_f32.  Lane l = (i = l & 31, kk = l >> 5) supplies A[i][kk] and B[kk][j = i]: one float2 of
// row e + kk per operand (VEC: 512 contiguous bytes per load instruction) whose two elements feed two MFMAs, so MFMA (p, q) sees
// the column sets {k0 + 2 i + p} x {n0 + 2 j + q}; without VEC (odd widths / strides) single floats and the sets {k0 + 32 p + i}.
template <typename T, int NACC>
__device__ __forceinline__ void wgrad_block_reduce(T* acc, int wv, int lane, T* lds) {
  // acc: NACC values per lane.  Rounds: waves 2,3 -> LDS, waves 0,1 add; wave 1 -> LDS, wave 0 adds.
#pragma unroll
  for (int round = 0; round < 2; ++round) {
    const int writers_lo = round == 0 ? 2 : 1, nw = round == 0 ? 2 : 1;
    __syncthreads();
    if (wv >= writers_lo && wv < writers_lo + nw) {
      T* d = lds + size_t(wv - writers_lo) * 64 * NACC;
#pragma unroll
      for (int r = 0; r < NACC; ++r) d[r * 64 + lane] = acc[r];
    }
    __syncthreads();
    if (wv < nw) {
      const T* d = lds + size_t(wv) * 64 * NACC;
#pragma unroll
      for (int r = 0; r < NACC; ++r) acc[r] += d[r * 64 + lane];
    }
  }
}

template <bool VEC>
__global__ __launch_bounds__(256) void wgrad_f32_kernel(WgradArgs a) {
  typedef float f2 __attribute__((ext_vector_type(2)));
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int slab = blockIdx.x * 4 + wv;
  const int k0 = blockIdx.y * 64, n0 = blockIdx.z * 64;
  const int i = lane & 31, kk = lane >> 5;
  const float* x = static_cast<const float*>(a.x);
  const float* g = static_cast<const float*>(a.g);
  const int64_t e0 = int64_t(slab) * a.rows_per_slab;
  const int64_t e1 = slab >= a.slabs ? e0 : (e0 + a.rows_per_slab < a.E ? e0 + a.rows_per_slab : a.E);
  // columns beyond K / N: clamped address, zero contribution
  int kc[2], nc[2];
  float km[2], nm[2];
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    const int k = VEC ? k0 + 2 * i + b : k0 + 32 * b + i, n = VEC ? n0 + 2 * i + b : n0 + 32 * b + i;
    kc[b] = k < a.K ? k : a.K - 1;
    nc[b] = n < a.N ? n : a.N - 1;
    km[b] = k < a.K ? 1.f : 0.f;
    nm[b] = n < a.N ? 1.f : 0.f;
  }
  if (VEC) {  // (pairs never straddle the end: widths are even on this path) a pair beyond the end re-reads the last valid pair
    kc[0] = kc[0] & ~1;
    nc[0] = nc[0] & ~1;
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
      if (VEC) {
        const f2 xx = *reinterpret_cast<const f2*>(x + rc * a.ldx + kc[0]);
        const f2 gg = *reinterpret_cast<const f2*>(g + rc * a.ldg + nc[0]);
        xv[s][0] = xx[0] * (m * km[0]);
        xv[s][1] = xx[1] * (m * km[1]);
        gv[s][0] = gg[0] * nm[0];
        gv[s][1] = gg[1] * nm[1];
      } else {
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          xv[s][b] = x[rc * a.ldx + kc[b]] * (m * km[b]);
          gv[s][b] = g[rc * a.ldg + nc[b]] * nm[b];
        }
      }
    }
#pragma unroll
    for (int s = 0; s < UN; ++s)
#pragma unroll
      for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int q = 0; q < 2; ++q) acc[p][q] = __builtin_amdgcn_mfma_f32_32x32x2f32(xv[s][p], gv[s][q], acc[p][q], 0, 0, 0);
  }
  float flat[64];
#pragma unroll
  for (int p = 0; p < 2; ++p)
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int r = 0; r < 16; ++r) flat[(p * 2 + q) * 16 + r] = acc[p][q][r];
  wgrad_block_reduce<float, 64>(flat, wv, lane, reinterpret_cast<float*>(aa_smem));
  if (wv != 0) return;
  // D[row = (r & 3) + 8 (r >> 2) + 4 (l >> 5)][col = l & 31]
  float* out = static_cast<float*>(a.partial) + int64_t(blockIdx.x) * a.K * a.N;
#pragma unroll
  for (int p = 0; p < 2; ++p)
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int n = VEC ? n0 + 2 * i + q : n0 + 32 * q + i;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ir = (r & 3) + 8 * (r >> 2) + 4 * kk;
        const int k = VEC ? k0 + 2 * ir + p : k0 + 32 * p + ir;
        if (k < a.K && n < a.N) out[int64_t(k) * a.N + n] = flat[(p * 2 + q) * 16 + r];
      }
    }
}

// fp64: 4 x 4 tiles of v_mfma_f64_16x16x4_f64 (A[i = l & 15][kk = l >> 4], B[kk][j = l & 15], D[4 r + (l >> 4)][l & 15]): four rows per
// step; VEC: one double2 per 32-column half, columns {k0 + 32 h + 2 i + b} for tile 2 h + b
template <bool VEC>
__global__ __launch_bounds__(256) void wgrad_f64_kernel(WgradArgs a) {
  typedef double d2 __attribute__((ext_vector_type(2)));
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int slab = blockIdx.x * 4 + wv;
  const int k0 = blockIdx.y * 64, n0 = blockIdx.z * 64;
  const int i = lane & 15, kk = lane >> 4;
  const double* x = static_cast<const double*>(a.x);
  const double* g = static_cast<const double*>(a.g);
  const int64_t e0 = int64_t(slab) * a.rows_per_slab;
  const int64_t e1 = slab >= a.slabs ? e0 : (e0 + a.rows_per_slab < a.E ? e0 + a.rows_per_slab : a.E);
  auto col = [&](int base, int b) { return VEC ? base + 32 * (b >> 1) + 2 * i + (b & 1) : base + 16 * b + i; };
  int kc[4], nc[4];
  double km[4], nm[4];
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    const int k = col(k0, b), n = col(n0, b);
    kc[b] = k < a.K ? k : a.K - 1;
    nc[b] = n < a.N ? n : a.N - 1;
    km[b] = k < a.K ? 1.0 : 0.0;
    nm[b] = n < a.N ? 1.0 : 0.0;
  }
  if (VEC) {
    kc[0] &= ~1; kc[2] &= ~1; nc[0] &= ~1; nc[2] &= ~1;
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
      if (VEC) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const d2 xx = *reinterpret_cast<const d2*>(x + rc * a.ldx + kc[2 * h]);
          const d2 gg = *reinterpret_cast<const d2*>(g + rc * a.ldg + nc[2 * h]);
          xv[s][2 * h] = xx[0] * (m * km[2 * h]);
          xv[s][2 * h + 1] = xx[1] * (m * km[2 * h + 1]);
          gv[s][2 * h] = gg[0] * nm[2 * h];
          gv[s][2 * h + 1] = gg[1] * nm[2 * h + 1];
        }
      } else {
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          xv[s][b] = x[rc * a.ldx + kc[b]] * (m * km[b]);
          gv[s][b] = g[rc * a.ldg + nc[b]] * nm[b];
        }
      }
    }
#pragma unroll
    for (int s = 0; s < UN; ++s)
#pragma unroll
      for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[p][q] = __builtin_amdgcn_mfma_f64_16x16x4f64(xv[s][p], gv[s][q], acc[p][q], 0, 0, 0);
  }
  double flat[64];
#pragma unroll
  for (int p = 0; p < 4; ++p)
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int r = 0; r < 4; ++r) flat[(p * 4 + q) * 4 + r] = acc[p][q][r];
  wgrad_block_reduce<double, 64>(flat, wv, lane, reinterpret_cast<double*>(aa_smem));
  if (wv != 0) return;
  double* out = static_cast<double*>(a.partial) + int64_t(blockIdx.x) * a.K * a.N;
#pragma unroll
  for (int p = 0; p < 4; ++p)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int n = col(n0, q);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int ir = 4 * r + kk;
        const int k = VEC ? k0 + 32 * (p >> 1) + 2 * ir + (p & 1) : k0 + 16 * p + ir;
        if (k < a.K && n < a.N) out[int64_t(k) * a.N + n] = flat[(p * 4 + q) * 4 + r];
      }
    }
}

// out[k][n] = sum of the workgroups' partial blocks, in order: 64 outputs x 4 interleaved partial sums per workgroup, combined
// ((s0 + s1) + (s2 + s3)) through LDS
template <typename T>
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const T* partial, int parts, int64_t KN, T* out) {
  T* lds = reinterpret_cast<T*>(aa_smem);
  const int o = threadIdx.x & 63, sl = threadIdx.x >> 6;
  const int64_t idx = int64_t(blockIdx.x) * 64 + o;
  T s = T(0);
  if (idx < KN)
    for (int b = sl; b < parts; b += 4) s += partial[int64_t(b) * KN + idx];
  lds[threadIdx.x] = s;
  __syncthreads();
  if (sl == 0 && idx < KN) out[idx] = (lds[o] + lds[64 + o]) + (lds[128 + o] + lds[192 + o]);
}

// rows per slab (one wave): enough slabs to fill the chip several times over for every output-block count, at least 128 rows
int64_t wgrad_rows_per_slab(int64_t E) {
  int64_t rows = (E + 4095) / 4096;
  rows = std::max<int64_t>(rows, 128);
  return (rows + 15) / 16 * 16;
}
int wgrad_parts(int64_t E) {  // workgroups along the rows = partial blocks in the workspace
  const int64_t rows = wgrad_rows_per_slab(E);
  const int64_t slabs = (E + rows - 1) / rows;
  return int(std::max<int64_t>(1, (slabs + 3) / 4));
}

// ---------------------------------------------------------------------------------------------------------------------
// weighted channels: B(sh, w)[e,c,i] = sh[e,i] w[e,c,r(i)]  (R == 1: one weight per channel, shared by all irreps)
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int irrep_of(int i) { return i < 1 ? 0 : (i < 4 ? 1 : (i < 9 ? 2 : 3)); }

// One workgroup = 4 consecutive edges; rows of u * D elements are walked with compile-time D (no 64-bit divisions per element)
template <typename T, int D>
__global__ __launch_bounds__(256) void wc_forward_kernel(int64_t E, int u, int R, const T* sh, const T* w, T* out) {
  const int row = u * D;
  for (int q = 0; q < 4; ++q) {
    const int64_t e = int64_t(blockIdx.x) * 4 + q;
    if (e >= E) return;
    const T* y = sh + e * D;
    const T* wr = w + e * int64_t(u) * R;
    T* o = out + e * int64_t(row);
    for (int t = threadIdx.x; t < row; t += 256) {
      const int c = t / D, i = t - c * D;
      o[t] = y[i] * wr[c * R + (R == 1 ? 0 : irrep_of(i))];
    }
  }
}

// gw[e,c,r] = sum_{i in r} g[e,c,i] sh[e,i]
template <typename T, int D>
__global__ __launch_bounds__(256) void wc_grad_w_kernel(int64_t EC, int u, int R, const T* g, const T* sh, T* gw) {
  const int64_t ec = int64_t(blockIdx.x) * 256 + threadIdx.x;
  if (ec >= EC) return;
  const int64_t e = ec / u;
  const T* gr = g + ec * D;
  const T* y = sh + e * D;
  T acc[4] = {T(0), T(0), T(0), T(0)};
#pragma unroll
  for (int i = 0; i < D; ++i) acc[R == 1 ? 0 : irrep_of(i)] += gr[i] * y[i];
#pragma unroll
  for (int r = 0; r < 4; ++r)
    if (r < R) gw[ec * R + r] = acc[r];
}

// gsh[e,i] = sum_c g[e,c,i] w[e,c,r(i)]: one wave per edge, lanes over channels, D wave sums
template <typename T, int D>
__global__ __launch_bounds__(256) void wc_grad_sh_kernel(int64_t E, int u, int R, const T* g, const T* w, T* gsh) {
  const int lane = threadIdx.x & 63;
  const int64_t e = int64_t(blockIdx.x) * 4 + (threadIdx.x >> 6);
  if (e >= E) return;
  T acc[D];
#pragma unroll
  for (int i = 0; i < D; ++i) acc[i] = T(0);
  for (int c = lane; c < u; c += 64) {
    const T* gr = g + (e * u + c) * D;
    const T* wr = w + (e * u + c) * R;
    T wv[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) wv[r] = r < R ? wr[r] : T(0);
#pragma unroll
    for (int i = 0; i < D; ++i) acc[i] += gr[i] * wv[R == 1 ? 0 : irrep_of(i)];
  }
#pragma unroll
  for (int i = 0; i < D; ++i) {
    T v = acc[i];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
    if (lane == 0) gsh[e * D + i] = v;
  }
}

}  // namespace
}  // namespace aa

extern "C" size_t aa_linear_wgrad_workspace_bytes(aa_dtype dtype, int64_t E, int K, int N) {
  if (E < 0 || K < 1 || N < 1) return 0;
  return size_t(aa::wgrad_parts(E)) * size_t(K) * size_t(N) * (dtype == AA_F32 ? 4 : 8);
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
  a.rows_per_slab = aa::wgrad_rows_per_slab(E);
  a.slabs = int((E + a.rows_per_slab - 1) / a.rows_per_slab);
  const int parts = aa::wgrad_parts(E);
  dim3 grid((unsigned)parts, (unsigned)((K + 63) / 64), (unsigned)((N + 63) / 64));
  const int64_t KN = int64_t(K) * N;
  // two-element loads need even widths and strides and 2-element-aligned bases
  const bool vec = (K % 2 == 0) && (N % 2 == 0) && (ldx % 2 == 0) && (ldg % 2 == 0) && (reinterpret_cast<uintptr_t>(x) % (2 * esize) == 0) &&
                   (reinterpret_cast<uintptr_t>(g) % (2 * esize) == 0);
  const size_t lds = 2 * 64 * 64 * esize;  // two waves' accumulators
  if (dtype == AA_F32) {
    if (vec)
      hipLaunchKernelGGL(aa::wgrad_f32_kernel<true>, grid, dim3(256), lds, s, a);
    else
      hipLaunchKernelGGL(aa::wgrad_f32_kernel<false>, grid, dim3(256), lds, s, a);
    hipLaunchKernelGGL(aa::wgrad_reduce_kernel<float>, dim3((unsigned)((KN + 63) / 64)), dim3(256), 256 * sizeof(float), s,
                       static_cast<const float*>(workspace), parts, KN, static_cast<float*>(out));
  } else {
    if (vec)
      hipLaunchKernelGGL(aa::wgrad_f64_kernel<true>, grid, dim3(256), lds, s, a);
    else
      hipLaunchKernelGGL(aa::wgrad_f64_kernel<false>, grid, dim3(256), lds, s, a);
    hipLaunchKernelGGL(aa::wgrad_reduce_kernel<double>, dim3((unsigned)((KN + 63) / 64)), dim3(256), 256 * sizeof(double), s,
                       static_cast<const double*>(workspace), parts, KN, static_cast<double*>(out));
  }
  AA_CHECK_HIP(hipGetLastError());
  return AA_OK;
}

template <typename T, int D>
static int wc_launch_d(int which, int64_t E, int u, int R, const void* p0, const void* p1, void* out, hipStream_t s) {
  const int64_t EC = E * u;
  if (which == 0) {
    hipLaunchKernelGGL((aa::wc_forward_kernel<T, D>), dim3((unsigned)((E + 3) / 4)), dim3(256), 0, s, E, u, R, static_cast<const T*>(p0),
                       static_cast<const T*>(p1), static_cast<T*>(out));
  } else if (which == 1) {
    hipLaunchKernelGGL((aa::wc_grad_w_kernel<T, D>), dim3((unsigned)((EC + 255) / 256)), dim3(256), 0, s, EC, u, R, static_cast<const T*>(p0),
                       static_cast<const T*>(p1), static_cast<T*>(out));
  } else {
    hipLaunchKernelGGL((aa::wc_grad_sh_kernel<T, D>), dim3((unsigned)((E + 3) / 4)), dim3(256), 0, s, E, u, R, static_cast<const T*>(p0),
                       static_cast<const T*>(p1), static_cast<T*>(out));
  }
  AA_CHECK_HIP(hipGetLastError());
  return AA_OK;
}

template <typename T>
static int wc_launch(int which, int64_t E, int u, int D, int R, const void* p0, const void* p1, void* out, hipStream_t s) {
  if (E == 0) return AA_OK;
  switch (D) {
    case 1: return wc_launch_d<T, 1>(which, E, u, R, p0, p1, out, s);
    case 4: return wc_launch_d<T, 4>(which, E, u, R, p0, p1, out, s);
    case 9: return wc_launch_d<T, 9>(which, E, u, R, p0, p1, out, s);
    default: return wc_launch_d<T, 16>(which, E, u, R, p0, p1, out, s);
  }
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
