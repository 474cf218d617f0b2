// Training-path kernels that the inference pipeline has no use for.
//
// The reference trains every ScalarMLPFunction through autograd (allegro/nn/_allegro.py:192-213): per linear layer the
// backward needs d W = x^T g, a product whose REDUCTION runs over the edges (K = E ~ 10^5..10^7, output 64..512 wide).  Library
// GEMMs pick a tile for the tiny output and walk the whole reduction in one or two workgroups (measured on MI355X: 0.73 ms for
// [64 x 298 144] @ [298 144 x 64], 3 TFLOP/s, 210 GB/s, and 16 ms in fp64 -- profiles/r05_v14_wgrad_bench.md); here the edges are
// cut into slabs, one workgroup per slab and 64..128-wide output block, rows staged through LDS with full-line loads, exact-fp32 /
// fp64 products on the matrix cores (v_mfma_f32_32x32x2_f32 / v_mfma_f64_16x16x4_f64: no split-precision needed, the kernel is
// HBM-bound), and the slabs are summed in a fixed order by a second small kernel: bit-reproducible, no atomics.
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

// A workgroup (4 waves, 2 x 2) owns a KB x NB block of the output and a slab of rows.  Rows are staged through LDS in groups of RS
// with full-line 16-byte loads (each row of x / g is read once per workgroup, whatever the MFMA operand layout wants), and every
// wave multiplies ITS (KB/2) x (NB/2) quarter: fp32 v_mfma_f32_32x32x2_f32 (A[i = l & 31][kk = l >> 5] = xs[2 s + kk][k + i],
// B[kk][j] = gs[2 s + kk][n + j]), fp64 v_mfma_f64_16x16x4_f64 (A[i = l & 15][kk = l >> 4], four rows per step).  One partial
// block per workgroup goes to the workspace; a second kernel sums the partial blocks in order.
template <typename T>
struct WgTile;
template <>
struct WgTile<float> {
  static constexpr int TS = 32, RE = 2, NR = 16;  // tile side, rows per MFMA, accumulator registers per tile
  typedef v16f_t acc_t;
  static __device__ __forceinline__ acc_t mma(float a, float b, acc_t c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }
  static __device__ __forceinline__ int out_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }
};
template <>
struct WgTile<double> {
  static constexpr int TS = 16, RE = 4, NR = 4;
  typedef v4d_t acc_t;
  static __device__ __forceinline__ acc_t mma(double a, double b, acc_t c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
  static __device__ __forceinline__ int out_row(int r, int lane) { return 4 * r + (lane >> 4); }
};

template <typename T, int KB, int NB, bool VEC>
__global__ __launch_bounds__(256) void wgrad_kernel(WgradArgs a) {
  typedef WgTile<T> W;
  constexpr int TS = W::TS, RE = W::RE, RS = sizeof(T) == 8 ? 16 : 32;  // RS rows per stage (<= 34 KB of LDS for every block shape)
  constexpr int TK = KB / 2 / TS, TN = NB / 2 / TS;      // tiles per wave
  constexpr int LDX = KB + 4, LDG = NB + 4;              // (+4: rows 2 s + kk of the two lane halves land in different banks)
  constexpr int VW = 16 / sizeof(T);                     // elements per 16-byte load
  T* xs = reinterpret_cast<T*>(aa_smem);                 // [RS][LDX]
  T* gs = xs + RS * LDX;                                 // [RS][LDG]
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int k0 = blockIdx.y * KB, n0 = blockIdx.z * NB;
  const int wk = (wv >> 1) * (KB / 2), wn = (wv & 1) * (NB / 2);  // this wave's quarter
  const int ti = lane & (TS - 1), kk = lane / TS;
  const T* x = static_cast<const T*>(a.x);
  const T* g = static_cast<const T*>(a.g);
  const int64_t e0 = int64_t(blockIdx.x) * a.rows_per_slab;
  const int64_t e1 = e0 + a.rows_per_slab < a.E ? e0 + a.rows_per_slab : a.E;
  typename W::acc_t acc[TK][TN];
#pragma unroll
  for (int p = 0; p < TK; ++p)
#pragma unroll
    for (int q = 0; q < TN; ++q)
#pragma unroll
      for (int r = 0; r < W::NR; ++r) acc[p][q][r] = T(0);
  // staging: thread t moves elements t, t + 256, ... of the RS x (KB | NB) tile; rows beyond the slab and columns beyond K / N are zero
  auto stage = [&](const T* src, int64_t ld, int c0, int width, int CB, int LD, T* dst, int64_t eb) {
    if (VEC) {
      typedef T vec_t __attribute__((ext_vector_type(VW)));
      const int per_row = CB / VW;
      for (int idx = tid; idx < RS * per_row; idx += 256) {
        const int r = idx / per_row, c = (idx - r * per_row) * VW;
        vec_t v;
#pragma unroll
        for (int q = 0; q < VW; ++q) v[q] = T(0);
        if (eb + r < e1 && c0 + c < width) v = *reinterpret_cast<const vec_t*>(src + (eb + r) * ld + c0 + c);  // (width % VW == 0 on this path)
#pragma unroll
        for (int q = 0; q < VW; ++q) dst[r * LD + c + q] = v[q];
      }
    } else {
      for (int idx = tid; idx < RS * CB; idx += 256) {
        const int r = idx / CB, c = idx - r * CB;
        dst[r * LD + c] = (eb + r < e1 && c0 + c < width) ? src[(eb + r) * ld + c0 + c] : T(0);
      }
    }
  };
  for (int64_t eb = e0; eb < e1; eb += RS) {
    __syncthreads();  // (the previous stage has been consumed)
    stage(x, a.ldx, k0, a.K, KB, LDX, xs, eb);
    stage(g, a.ldg, n0, a.N, NB, LDG, gs, eb);
    __syncthreads();
#pragma unroll
    for (int s = 0; s < RS / RE; ++s) {
      T av[TK], bv[TN];
#pragma unroll
      for (int p = 0; p < TK; ++p) av[p] = xs[(RE * s + kk) * LDX + wk + TS * p + ti];
#pragma unroll
      for (int q = 0; q < TN; ++q) bv[q] = gs[(RE * s + kk) * LDG + wn + TS * q + ti];
#pragma unroll
      for (int p = 0; p < TK; ++p)
#pragma unroll
        for (int q = 0; q < TN; ++q) acc[p][q] = W::mma(av[p], bv[q], acc[p][q]);
    }
  }
  T* out = static_cast<T*>(a.partial) + int64_t(blockIdx.x) * a.K * a.N;
#pragma unroll
  for (int p = 0; p < TK; ++p)
#pragma unroll
    for (int q = 0; q < TN; ++q) {
      const int n = n0 + wn + TS * q + ti;
#pragma unroll
      for (int r = 0; r < W::NR; ++r) {
        const int k = k0 + wk + TS * p + W::out_row(r, lane);
        if (k < a.K && n < a.N) out[int64_t(k) * a.N + n] = acc[p][q][r];
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

// rows per workgroup: ~1024 workgroups along the rows (several per CU for every output-block count), at least 256 rows each
int64_t wgrad_rows_per_slab(int64_t E) {
  int64_t rows = (E + 1023) / 1024;
  rows = std::max<int64_t>(rows, 256);
  return (rows + 31) / 32 * 32;
}
int wgrad_parts(int64_t E) {  // workgroups along the rows = partial blocks in the workspace
  const int64_t rows = wgrad_rows_per_slab(E);
  return int(std::max<int64_t>(1, (E + rows - 1) / rows));
}

template <typename T, int KB, int NB>
int wgrad_launch(const WgradArgs& a, bool vec, int parts, hipStream_t s) {
  dim3 grid((unsigned)parts, (unsigned)((a.K + KB - 1) / KB), (unsigned)((a.N + NB - 1) / NB));
  const size_t lds = sizeof(T) * (sizeof(T) == 8 ? 16 : 32) * size_t(KB + 4 + NB + 4);
  if (vec)
    hipLaunchKernelGGL((wgrad_kernel<T, KB, NB, true>), grid, dim3(256), lds, s, a);
  else
    hipLaunchKernelGGL((wgrad_kernel<T, KB, NB, false>), grid, dim3(256), lds, s, a);
  return AA_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// weighted channels: B(sh, w)[e,c,i] = sh[e,i] w[e,c,r(i)]  (R == 1: one weight per channel, shared by all irreps)
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int irrep_of(int i) { return i < 1 ? 0 : (i < 4 ? 1 : (i < 9 ? 2 : 3)); }

// One workgroup = 4 consecutive edges; rows of u * D elements are walked with compile-time D (no 64-bit divisions per element)
template <typename T, int D>
__global__ __launch_bounds__(256) void wc_forward_kernel(int64_t E, int u, int R, const T* sh, const T* w, T* out) {
  // (write-bound: 2.15 TB/s of stores at C3 size; a 16-byte-store variant measured the same 319 us, tools/wc_probe.py)
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
  const int parts = aa::wgrad_parts(E);
  a.slabs = parts;
  const int64_t KN = int64_t(K) * N;
  // 16-byte loads need widths and strides that are multiples of the vector and 16-byte-aligned bases
  const int vw = int(16 / esize);
  const bool vec = (K % vw == 0) && (N % vw == 0) && (ldx % vw == 0) && (ldg % vw == 0) && (reinterpret_cast<uintptr_t>(x) % 16 == 0) &&
                   (reinterpret_cast<uintptr_t>(g) % 16 == 0);
  // block shape: 128-wide where the dimension is wider than 64 (each operand is then re-read half as often)
  const bool k128 = K > 64, n128 = N > 64;
  int rc;
  if (dtype == AA_F32) {
    rc = k128 ? (n128 ? aa::wgrad_launch<float, 128, 128>(a, vec, parts, s) : aa::wgrad_launch<float, 128, 64>(a, vec, parts, s))
              : (n128 ? aa::wgrad_launch<float, 64, 128>(a, vec, parts, s) : aa::wgrad_launch<float, 64, 64>(a, vec, parts, s));
    hipLaunchKernelGGL(aa::wgrad_reduce_kernel<float>, dim3((unsigned)((KN + 63) / 64)), dim3(256), 256 * sizeof(float), s,
                       static_cast<const float*>(workspace), parts, KN, static_cast<float*>(out));
  } else {
    rc = k128 ? (n128 ? aa::wgrad_launch<double, 128, 128>(a, vec, parts, s) : aa::wgrad_launch<double, 128, 64>(a, vec, parts, s))
              : (n128 ? aa::wgrad_launch<double, 64, 128>(a, vec, parts, s) : aa::wgrad_launch<double, 64, 64>(a, vec, parts, s));
    hipLaunchKernelGGL(aa::wgrad_reduce_kernel<double>, dim3((unsigned)((KN + 63) / 64)), dim3(256), 256 * sizeof(double), s,
                       static_cast<const double*>(workspace), parts, KN, static_cast<double*>(out));
  }
  (void)rc;
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
