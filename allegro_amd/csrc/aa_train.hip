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
#include "aa_wave.h"

namespace aa {
namespace {

template <typename T>
struct Pk16;  // 16 bytes of T
template <>
struct Pk16<float> {
  typedef float type __attribute__((ext_vector_type(4)));
};
template <>
struct Pk16<double> {
  typedef double type __attribute__((ext_vector_type(2)));
};


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

// Slabs: the workgroups of ONE launch should fill the chip exactly once -- 256 CUs x the workgroups a CU holds (8 by waves, fewer by
// the LDS of the block shape).  The first rule (~1024 slabs of >= 256 rows whatever the block count) launched 1864 workgroups for
// [3e5 x 192]^T [3e5 x 64] on 1536 slots: a second round at 21 % occupancy.  At least 64 rows per slab, at most 4096 slabs.
constexpr int kChipCUs = 256;
template <typename T>
int wgrad_block_shape(int K, int N, int* kb, int* nb) {  // workgroups per slab; block widths
  *kb = K > 64 ? 128 : 64;
  *nb = N > 64 ? 128 : 64;
  return ((K + *kb - 1) / *kb) * ((N + *nb - 1) / *nb);
}
template <typename T>
int64_t wgrad_rows_per_slab(int64_t E, int K, int N) {
  int kb, nb;
  const int per_slab = wgrad_block_shape<T>(K, N, &kb, &nb);
  const size_t lds = sizeof(T) * (sizeof(T) == 8 ? 16 : 32) * size_t(kb + 4 + nb + 4);
  const int per_cu = int(std::min<size_t>(8, std::max<size_t>(1, (160 * 1024) / lds)));
  const int64_t slabs = std::max<int64_t>(1, std::min<int64_t>(4096, int64_t(kChipCUs) * per_cu / per_slab));
  int64_t rows = std::max<int64_t>((E + slabs - 1) / slabs, 64);
  return (rows + 31) / 32 * 32;
}
template <typename T>
int wgrad_parts(int64_t E, int K, int N) {  // workgroups along the rows = partial blocks in the workspace
  const int64_t rows = wgrad_rows_per_slab<T>(E, K, N);
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

// Forward, slot form (rows of u D <= 256 kWcSlots elements).  A workgroup takes G consecutive edges: their sh rows (G D contiguous
// elements) and weight rows (G u R contiguous elements) are copied to LDS with coalesced loads, then every thread walks the G output
// rows through its slots -- thread `tid` owns the elements tid, tid + 256, ... of EVERY row; their (component, weight index) pairs are
// computed once and kept in registers -- with two LDS reads, a multiply and a coalesced store per element.  TWO: out = sh (x) w +
// sh2 (x) w2 in the same pass (the gradient of the fused pair of contractions with respect to their common operand: one store
// stream instead of two stores, two loads and a third store of an addition).
// History at C3 size (687 MB of stores), all forms reading the operands per element straight from global memory: 4 edges one after
// the other with a load-multiply-store chain per element 316 us; batched loads 242 us; incrementally carried indices 244 us; slots
// in registers with all loads of two edges in flight 292 us (TWO: 428 us) -- the time follows the NUMBER of per-lane gather loads
// (each lane of a load hits one of 9 / 24 scattered words: the address unit serialises them), ~70 us per gather per element.
constexpr int kWcSlots = 8, kWcMaxGroup = 16, kWcLdsBytes = 32768;
template <typename T, int D, bool TWO, int NS>  // NS: slots per thread, 256 NS >= u D
__global__ __launch_bounds__(256) void wc_forward_slots_kernel(int64_t E, int u, int R, int G, const T* __restrict__ sh,
                                                               const T* __restrict__ w, int64_t ldw, const T* __restrict__ sh2,
                                                               const T* __restrict__ w2, int64_t ldw2, T* __restrict__ out) {
  T* ly = reinterpret_cast<T*>(aa_smem);  // [G][D] | [G][u R] (| the same for the second term)
  const int row = u * D, wrow = u * R;
  T* lw = ly + G * D;
  T* ly2 = lw + G * wrow;
  T* lw2 = ly2 + G * D;
  const int64_t e0 = int64_t(blockIdx.x) * G;
  const int n = int(E - e0 < G ? E - e0 : int64_t(G));
  for (int k = threadIdx.x; k < n * D; k += 256) {
    ly[k] = sh[e0 * D + k];
    if (TWO) ly2[k] = sh2[e0 * D + k];
  }
  for (int k = threadIdx.x; k < n * wrow; k += 256) {  // (weight rows may be a column block of a wider matrix: row stride ldw)
    const int j = k / wrow, c = k - j * wrow;
    lw[k] = w[(e0 + j) * ldw + c];
    if (TWO) lw2[k] = w2[(e0 + j) * ldw2 + c];
  }
  int oy[NS], ow[NS];
#pragma unroll
  for (int m = 0; m < NS; ++m) {
    const int t = int(threadIdx.x) + 256 * m, c = t / D, i = t - c * D;
    // (slots past the end of the row read element 0 and store nothing)
    oy[m] = t < row ? i : 0;
    ow[m] = t < row ? c * R + (R == 1 ? 0 : irrep_of(i)) : 0;
  }
  __syncthreads();
  for (int j = 0; j < n; ++j) {
    T* o = out + (e0 + j) * row;
#pragma unroll
    for (int m = 0; m < NS; ++m) {
      const int t = int(threadIdx.x) + 256 * m;
      T v = ly[j * D + oy[m]] * lw[j * wrow + ow[m]];
      if (TWO) v += ly2[j * D + oy[m]] * lw2[j * wrow + ow[m]];
      if (t < row) o[t] = v;
    }
  }
}

// Forward, general form (any row length): one workgroup = 8 consecutive edges = one contiguous span of outputs, four elements per
// thread and batch, the operand loads of a batch in front of its stores.
constexpr int kWcEdges = 8;
template <typename T, int D, bool TWO>
__global__ __launch_bounds__(256) void wc_forward_kernel(int64_t E, int u, int R, const T* __restrict__ sh, const T* __restrict__ w,
                                                         int64_t ldw, const T* __restrict__ sh2, const T* __restrict__ w2, int64_t ldw2,
                                                         T* __restrict__ out) {
  const int64_t e0 = int64_t(blockIdx.x) * kWcEdges;
  const int row = u * D;
  const int total = int(E - e0 < kWcEdges ? E - e0 : int64_t(kWcEdges)) * row;
  const T* y = sh + e0 * D;
  const T* wr = w + e0 * ldw;
  const T* y2 = TWO ? sh2 + e0 * D : nullptr;
  const T* wr2 = TWO ? w2 + e0 * ldw2 : nullptr;
  T* o = out + e0 * int64_t(row);
  for (int t0 = threadIdx.x; t0 < total; t0 += 1024) {
    T v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int t = t0 + k * 256;
      v[k] = T(0);
      if (t < total) {
        const int cg = t / D, i = t - cg * D;  // (channel of the span, component)
        const int q = cg / u, iy = q * D + i, iw = (cg - q * u) * R + (R == 1 ? 0 : irrep_of(i));
        v[k] = y[iy] * wr[q * ldw + iw];
        if (TWO) v[k] += y2[iy] * wr2[q * ldw2 + iw];
      }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (t0 + k * 256 < total) o[t0 + k * 256] = v[k];
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
__global__ __launch_bounds__(256) void wc_grad_sh_kernel(int64_t E, int u, int R, const T* g, const T* w, int64_t ldw, T* gsh) {
  const int lane = threadIdx.x & 63;
  const int64_t e = int64_t(blockIdx.x) * 4 + (threadIdx.x >> 6);
  if (e >= E) return;
  T acc[D];
#pragma unroll
  for (int i = 0; i < D; ++i) acc[i] = T(0);
  for (int c = lane; c < u; c += 64) {
    const T* gr = g + (e * u + c) * D;
    const T* wr = w + e * ldw + c * R;
    T wv[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) wv[r] = r < R ? wr[r] : T(0);
#pragma unroll
    for (int i = 0; i < D; ++i) acc[i] += gr[i] * wv[R == 1 ? 0 : irrep_of(i)];
  }
  wave_sum_store<T, D>(acc, gsh + e * D, true, false);
}

// Both contractions of one [E,u,D] operand in ONE pass over it: gsh[e,i] = sum_c t[e,c,i] w[e,c,r(i)] and
// gw[e,c,r] = sum_{i in r} t[e,c,i] sh[e,i] (every first derivative of a weighted-channel tensor needs the pair, and so does every
// derivative of the pair itself, with other second operands).  One wave per edge, lanes over channels: the weight-side sums are
// lane-private, the harmonics-side sums D wave reductions.  The rows of 64 channels (64 D contiguous elements) are fetched with
// coalesced loads into a wave-private LDS patch and read back channel-per-lane at an odd stride (no bank conflicts); reading the D
// components of a lane's channel straight from global memory is one 64-way gather per component (229 us per call at C3 size).
template <int D>
__device__ __forceinline__ int wc_patch_index(int q) {  // element q of a 64-channel row block -> patch word (odd channel stride)
  if (D & 1) return q;
  constexpr int kShift = D == 4 ? 2 : (D == 16 ? 4 : (D == 2 ? 1 : 3));
  return (q >> kShift) * (D + 1) + (q & (D - 1));
}
// (A first staged form -- one edge per wave and workgroup barriers around the patch -- was slower than the gathers, 356 us: a wave's
//  whole life was one 2.3-KB row and two barriers.  This one: kWcPairEdges edges per wave, wave-level ordering only, the next row block's
//  loads issued before the current one is consumed.)
constexpr int kWcPairEdges = 4;
template <typename T, int D>
__global__ __launch_bounds__(256) void wc_grad_pair_kernel(int64_t E, int u, int R, const T* __restrict__ t, const T* __restrict__ sh,
                                                           const T* __restrict__ w, int64_t ldw, T* __restrict__ gsh, T* __restrict__ gw) {
  constexpr int DP = D | 1;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  T* patch = reinterpret_cast<T*>(aa_smem) + wave * 64 * DP;
  const int64_t e_first = (int64_t(blockIdx.x) * 4 + wave) * kWcPairEdges;
  if (e_first >= E) return;
  const int n_e = int(E - e_first < kWcPairEdges ? E - e_first : int64_t(kWcPairEdges));
  const int chunks = (u + 63) / 64, items = n_e * chunks;
  // item = (edge of the wave, block of 64 channels); its 64 D contiguous elements as D coalesced loads (element lane + 64 m)
  auto fetch = [&](int it, T* r) {
    const int c0 = (it % chunks) * 64, nc = u - c0 < 64 ? u - c0 : 64;
    const T* src = t + ((e_first + it / chunks) * u + c0) * D;
#pragma unroll
    for (int m = 0; m < D; ++m) r[m] = m * 64 + lane < nc * D ? src[m * 64 + lane] : T(0);
  };
  T r[D], acc[D];
#pragma unroll
  for (int i = 0; i < D; ++i) acc[i] = T(0);
  fetch(0, r);
  for (int it = 0; it < items; ++it) {
    const int64_t e = e_first + it / chunks;
    const int c = (it % chunks) * 64 + lane;
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int m = 0; m < D; ++m) patch[wc_patch_index<D>(m * 64 + lane)] = r[m];
    __builtin_amdgcn_wave_barrier();
    T x[D];
#pragma unroll
    for (int i = 0; i < D; ++i) x[i] = patch[lane * DP + i];
    if (it + 1 < items) fetch(it + 1, r);
    if (c < u) {
      const T* wr = w + e * ldw + c * R;
      T wv[4], s4[4] = {T(0), T(0), T(0), T(0)};
#pragma unroll
      for (int q = 0; q < 4; ++q) wv[q] = q < R ? wr[q] : T(0);
#pragma unroll
      for (int i = 0; i < D; ++i) {
        acc[i] += x[i] * wv[R == 1 ? 0 : irrep_of(i)];
        s4[R == 1 ? 0 : irrep_of(i)] += x[i] * sh[e * D + i];
      }
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (q < R) gw[(e * u + c) * R + q] = s4[q];
    }
    if (it % chunks == chunks - 1) {  // the edge is complete: D wave sums (reduce-scatter butterfly, aa_wave.h)
      wave_sum_store<T, D>(acc, gsh + e * D, true, false);
#pragma unroll
      for (int i = 0; i < D; ++i) acc[i] = T(0);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// SiLU and its derivatives as ONE elementwise family: A_k(x, g) = g f^(k)(x), f = x sigmoid(x).  d A_k / dx = A_{k+1}(x, g .),
// d A_k / dg = A_k(x, .): every derivative of a member is a member, so the hidden activations of the scalar MLPs cost one launch
// in the energy, one in the forces and one (the pair form: both gradients from one read of x, g, h) in the gradient of a force loss --
// autograd's own SiLU double-backward is a dozen elementwise launches per site.
// ---------------------------------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ T silu_order(T x, int k) {
  const T s = sigmoid_(x), s1 = s * (T(1) - s), m = T(1) - T(2) * s;
  switch (k) {
    case 0: return x * s;
    case 1: return s * (T(1) + x * (T(1) - s));
    case 2: return s1 * (T(2) + x * m);
    default: return s1 * (m * (T(2) + x * m) + m - T(2) * x * s1);
  }
}
// The same family for the other two nonlinearities the reference offers (allegro_models.py:49-60), evaluated in double (these MLPs
// are not on the fp32 fast paths; an elementwise pass is bandwidth-bound either way):
//   mish: f = x t, t = tanh(softplus x); with s = sigmoid x, w = 1 - t^2, q = 1 - s - 2 t s:
//         t1 = t' = w s, t2 = t'' = w s q, t3 = w s (q^2 + q'), q' = -s (1 - s) - 2 (w s^2 + t s (1 - s));  f^(k) = k t_(k-1) + x t_k
//   gelu (erf form): f = x Phi; f' = Phi + x phi, f'' = phi (2 - x^2), f^(3) = x phi (x^2 - 4)
template <typename T>
__device__ __forceinline__ T act_order(int act, T x, int k) {
  if (act == AA_ACT_SILU) return silu_order(x, k);
  const double xd = double(x);
  if (act == AA_ACT_MISH) {
    const double sp = xd > 30.0 ? xd : log1p(exp(xd));
    const double t = tanh(sp), s = 1.0 / (1.0 + exp(-xd)), w = 1.0 - t * t;
    if (k == 0) return T(xd * t);
    const double t1 = w * s;
    if (k == 1) return T(t + xd * t1);
    const double q = 1.0 - s - 2.0 * t * s, t2 = t1 * q;
    if (k == 2) return T(2.0 * t1 + xd * t2);
    const double q1 = -s * (1.0 - s) - 2.0 * (w * s * s + t * s * (1.0 - s));
    const double t3 = t1 * (q * q + q1);
    return T(3.0 * t2 + xd * t3);
  }
  const double phi = 0.39894228040143267794 * exp(-0.5 * xd * xd);
  if (k == 0) return T(0.5 * xd * (1.0 + erf(xd * 0.70710678118654752440)));
  if (k == 1) return T(0.5 * (1.0 + erf(xd * 0.70710678118654752440)) + xd * phi);
  if (k == 2) return T(phi * (2.0 - xd * xd));
  return T(xd * phi * (xd * xd - 4.0));
}
// h == nullptr: out0 = g f^(k)(x) (g == nullptr: f^(k)(x));  else: out0 = g h f^(k+1)(x), out1 = h f^(k)(x)
template <typename T>
__global__ __launch_bounds__(256) void silu_family_kernel(int64_t n, int k, const T* __restrict__ x, const T* __restrict__ g,
                                                          const T* __restrict__ h, T* __restrict__ out0, T* __restrict__ out1, int act) {
  constexpr int V = 16 / sizeof(T);
  const int64_t i0 = (int64_t(blockIdx.x) * 256 + threadIdx.x) * V;
  if (i0 >= n) return;
  T xv[V], gv[V], hv[V];
  const bool full = i0 + V <= n;
  if (full) {
    using Vec = typename Pk16<T>::type;
    *reinterpret_cast<Vec*>(xv) = *reinterpret_cast<const Vec*>(x + i0);
    if (g) *reinterpret_cast<Vec*>(gv) = *reinterpret_cast<const Vec*>(g + i0);
    if (h) *reinterpret_cast<Vec*>(hv) = *reinterpret_cast<const Vec*>(h + i0);
  } else {
    for (int j = 0; j < V; ++j) {
      const bool ok = i0 + j < n;
      xv[j] = ok ? x[i0 + j] : T(0);
      gv[j] = ok && g ? g[i0 + j] : T(0);
      hv[j] = ok && h ? h[i0 + j] : T(0);
    }
  }
  T a[V], b[V];
#pragma unroll
  for (int j = 0; j < V; ++j) {
    const T gj = g ? gv[j] : T(1);
    if (h) {
      a[j] = gj * hv[j] * act_order(act, xv[j], k + 1);
      b[j] = hv[j] * act_order(act, xv[j], k);
    } else {
      a[j] = gj * act_order(act, xv[j], k);
      b[j] = T(0);
    }
  }
  if (full) {
    using Vec = typename Pk16<T>::type;
    *reinterpret_cast<Vec*>(out0 + i0) = *reinterpret_cast<const Vec*>(a);
    if (h) *reinterpret_cast<Vec*>(out1 + i0) = *reinterpret_cast<const Vec*>(b);
  } else {
    for (int j = 0; j < V && i0 + j < n; ++j) {
      out0[i0 + j] = a[j];
      if (h) out1[i0 + j] = b[j];
    }
  }
}

// out[r, i] = a[r, i] + (i == 0 ? s[r] : 0) over rows of D elements (a == nullptr: zeros): the scalar (l = 0) components of a tensor
// feature are ALSO an input of the next latent MLP (`features[:, :, 0]`, _allegro.py:275-283), so the feature's gradient is the sum of
// the tensor-product gradient and the MLP's, padded -- one pass instead of a zero fill, a strided copy and an addition.
template <typename T, int DC>  // DC: the row length when it is one of the usual ones, 0: any (run-time division)
__global__ __launch_bounds__(256) void scalar_column_kernel(int64_t n, int d_any, const T* __restrict__ a, const T* __restrict__ s,
                                                            T* __restrict__ out) {
  // four consecutive elements per thread (one 16-byte access each way in fp32), one division per thread: the (row, component) of
  // the following elements are carried along
  const int D = DC ? DC : d_any;
  const int64_t t0 = (int64_t(blockIdx.x) * 256 + threadIdx.x) * 4;
  if (t0 >= n) return;
  int64_t r = t0 / D;
  int i = int(t0 - r * D);
  T v[4];
  const bool full = t0 + 4 <= n;
  if (a) {
    if (full && sizeof(T) == 4) {
      *reinterpret_cast<typename Pk16<float>::type*>(v) = *reinterpret_cast<const typename Pk16<float>::type*>(a + t0);
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = t0 + j < n ? a[t0 + j] : T(0);
    }
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = T(0);
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    if (i == 0 && t0 + j < n) v[j] += s[r];
    if (++i == D) {
      i = 0;
      ++r;
    }
  }
  if (full && sizeof(T) == 4) {
    *reinterpret_cast<typename Pk16<float>::type*>(out + t0) = *reinterpret_cast<const typename Pk16<float>::type*>(v);
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (t0 + j < n) out[t0 + j] = v[j];
  }
}

// out[e, off_j + c] = x_j[e, c]: the per-edge scalar features of all layers side by side (the input of every latent MLP and of the
// readout, _allegro.py:275-283) -- and the gradient of a column split.  Four columns per thread where widths, strides and bases allow
// 16-byte accesses.
constexpr int kCatMax = 8;
struct CatArgs {
  const void* x[kCatMax];
  int64_t ld[kCatMax];
  int width[kCatMax], off[kCatMax + 1];
  int n, total;
  int64_t E, ldo;
  void* out;
};
template <typename T, int V>
__global__ __launch_bounds__(256) void concat_columns_kernel(CatArgs a) {
  const int per_row = a.total / V;
  const int64_t t = int64_t(blockIdx.x) * 256 + threadIdx.x;
  if (t >= a.E * per_row) return;
  const int64_t e = t / per_row;
  const int c = int(t - e * per_row) * V;
  int j = 0;
#pragma unroll
  for (int q = 1; q < kCatMax; ++q)
    if (q < a.n && c >= a.off[q]) j = q;
  const T* src = static_cast<const T*>(a.x[j]) + e * a.ld[j] + (c - a.off[j]);
  T* dst = static_cast<T*>(a.out) + e * a.ldo + c;
  if (V == 1) {
    *dst = *src;
  } else {
    using Vec = typename Pk16<T>::type;
    *reinterpret_cast<Vec*>(dst) = *reinterpret_cast<const Vec*>(src);
  }
}

}  // namespace
}  // namespace aa

extern "C" size_t aa_linear_wgrad_workspace_bytes(aa_dtype dtype, int64_t E, int K, int N) {
  if (E < 0 || K < 1 || N < 1) return 0;
  const int parts = dtype == AA_F32 ? aa::wgrad_parts<float>(E, K, N) : aa::wgrad_parts<double>(E, K, N);
  return size_t(parts) * size_t(K) * size_t(N) * (dtype == AA_F32 ? 4 : 8);
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
  a.rows_per_slab = dtype == AA_F32 ? aa::wgrad_rows_per_slab<float>(E, K, N) : aa::wgrad_rows_per_slab<double>(E, K, N);
  const int parts = dtype == AA_F32 ? aa::wgrad_parts<float>(E, K, N) : aa::wgrad_parts<double>(E, K, N);
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
    if (rc == AA_OK) rc = aa::launch_column_sum<float>(static_cast<float*>(workspace), parts, KN, static_cast<float*>(out), s);
  } else {
    rc = k128 ? (n128 ? aa::wgrad_launch<double, 128, 128>(a, vec, parts, s) : aa::wgrad_launch<double, 128, 64>(a, vec, parts, s))
              : (n128 ? aa::wgrad_launch<double, 64, 128>(a, vec, parts, s) : aa::wgrad_launch<double, 64, 64>(a, vec, parts, s));
    if (rc == AA_OK) rc = aa::launch_column_sum<double>(static_cast<double*>(workspace), parts, KN, static_cast<double*>(out), s);
  }
  if (rc != AA_OK) return rc;
  AA_CHECK_HIP(hipGetLastError());
  return AA_OK;
}

template <typename T, int D, bool TWO, int NS>
static void wc_launch_slots(int64_t E, int u, int R, const T* sh, const T* w, int64_t ldw, const T* sh2, const T* w2, int64_t ldw2, T* out,
                            hipStream_t s) {
  const size_t per_edge = sizeof(T) * size_t(u * R + D) * (TWO ? 2 : 1);
  const int G = int(std::max<size_t>(1, std::min<size_t>(aa::kWcMaxGroup, aa::kWcLdsBytes / per_edge)));
  hipLaunchKernelGGL((aa::wc_forward_slots_kernel<T, D, TWO, NS>), dim3((unsigned)((E + G - 1) / G)), dim3(256), per_edge * G, s, E, u, R, G, sh, w,
                     ldw, sh2, w2, ldw2, out);
}
template <typename T, int D, bool TWO>
static void wc_launch_forward(int64_t E, int u, int R, const T* sh, const T* w, int64_t ldw, const T* sh2, const T* w2, int64_t ldw2, T* out,
                              hipStream_t s) {
  const int ns = (u * D + 255) / 256;
  if (ns <= 1) wc_launch_slots<T, D, TWO, 1>(E, u, R, sh, w, ldw, sh2, w2, ldw2, out, s);
  else if (ns == 2) wc_launch_slots<T, D, TWO, 2>(E, u, R, sh, w, ldw, sh2, w2, ldw2, out, s);
  else if (ns == 3) wc_launch_slots<T, D, TWO, 3>(E, u, R, sh, w, ldw, sh2, w2, ldw2, out, s);
  else if (ns == 4) wc_launch_slots<T, D, TWO, 4>(E, u, R, sh, w, ldw, sh2, w2, ldw2, out, s);
  else if (ns == 5) wc_launch_slots<T, D, TWO, 5>(E, u, R, sh, w, ldw, sh2, w2, ldw2, out, s);
  else if (ns <= aa::kWcSlots) wc_launch_slots<T, D, TWO, aa::kWcSlots>(E, u, R, sh, w, ldw, sh2, w2, ldw2, out, s);
  else
    hipLaunchKernelGGL((aa::wc_forward_kernel<T, D, TWO>), dim3((unsigned)((E + aa::kWcEdges - 1) / aa::kWcEdges)), dim3(256), 0, s, E, u, R, sh,
                       w, ldw, sh2, w2, ldw2, out);
}

// which 0..2: the three single forms; 3: out = p0 (x) p1 + p2 (x) p3; 4: the pair (out, out2) = (t . w, t . sh) with t = p0, sh = p1, w = p2.
// ldw / ldw2: row strides of the weight operands (which 0: p1, 2: p1, 3: p1 and p3, 4: p2)
template <typename T, int D>
static int wc_launch_d(int which, int64_t E, int u, int R, const void* p0, const void* p1, const void* p2, const void* p3, int64_t ldw, int64_t ldw2,
                       void* out, void* out2, hipStream_t s) {
  const int64_t EC = E * u;
  const T *a = static_cast<const T*>(p0), *b = static_cast<const T*>(p1), *c = static_cast<const T*>(p2), *d = static_cast<const T*>(p3);
  if (which == 0) {
    wc_launch_forward<T, D, false>(E, u, R, a, b, ldw, nullptr, nullptr, 0, static_cast<T*>(out), s);
  } else if (which == 3) {
    wc_launch_forward<T, D, true>(E, u, R, a, b, ldw, c, d, ldw2, static_cast<T*>(out), s);
  } else if (which == 1) {
    hipLaunchKernelGGL((aa::wc_grad_w_kernel<T, D>), dim3((unsigned)((EC + 255) / 256)), dim3(256), 0, s, EC, u, R, a, b, static_cast<T*>(out));
  } else if (which == 2) {
    hipLaunchKernelGGL((aa::wc_grad_sh_kernel<T, D>), dim3((unsigned)((E + 3) / 4)), dim3(256), 0, s, E, u, R, a, b, ldw, static_cast<T*>(out));
  } else {
    hipLaunchKernelGGL((aa::wc_grad_pair_kernel<T, D>), dim3((unsigned)((E + 4 * aa::kWcPairEdges - 1) / (4 * aa::kWcPairEdges))), dim3(256),
                       sizeof(T) * 4 * 64 * (D | 1), s, E, u, R, a, b, c, ldw, static_cast<T*>(out), static_cast<T*>(out2));
  }
  AA_CHECK_HIP(hipGetLastError());
  return AA_OK;
}

template <typename T>
static int wc_launch(int which, int64_t E, int u, int D, int R, const void* p0, const void* p1, const void* p2, const void* p3, int64_t ldw,
                     int64_t ldw2, void* out, void* out2, hipStream_t s) {
  if (E == 0) return AA_OK;
  switch (D) {
    case 1: return wc_launch_d<T, 1>(which, E, u, R, p0, p1, p2, p3, ldw, ldw2, out, out2, s);
    case 4: return wc_launch_d<T, 4>(which, E, u, R, p0, p1, p2, p3, ldw, ldw2, out, out2, s);
    case 9: return wc_launch_d<T, 9>(which, E, u, R, p0, p1, p2, p3, ldw, ldw2, out, out2, s);
    default: return wc_launch_d<T, 16>(which, E, u, R, p0, p1, p2, p3, ldw, ldw2, out, out2, s);
  }
}

static int wc_check(const char* what, int64_t E, int u, int l_max, int R, int64_t ldw, int64_t ldw2) {
  if (!(E >= 0 && u >= 1 && l_max >= 0 && l_max <= 3)) return aa::fail(AA_ERR_INVALID, std::string(what) + ": bad argument");
  if (!(E * int64_t(u) * 16 < (int64_t(1) << 40))) return aa::fail(AA_ERR_INVALID, std::string(what) + ": too large");
  if (ldw < int64_t(u) * R || ldw2 < int64_t(u) * R) return aa::fail(AA_ERR_INVALID, std::string(what) + ": weight row stride shorter than u R");
  return AA_OK;
}

extern "C" int aa_weighted_channels(aa_dtype dtype, int which, int64_t E, int u, int l_max, int shared, const void* a, const void* b, int64_t ldw,
                                    void* out, aa_stream stream) {
  AA_REQUIRE(which >= 0 && which <= 2, "aa_weighted_channels: bad argument");
  const int D = (l_max + 1) * (l_max + 1), R = shared ? 1 : l_max + 1;
  if (which == 1) ldw = int64_t(u) * R;  // (no weight operand)
  if (int rc = wc_check("aa_weighted_channels", E, u, l_max, R, ldw, ldw)) return rc;
  AA_REQUIRE(E == 0 || (a && b && out), "aa_weighted_channels: null argument");
  hipStream_t s = static_cast<hipStream_t>(stream);
  return dtype == AA_F32 ? wc_launch<float>(which, E, u, D, R, a, b, nullptr, nullptr, ldw, 0, out, nullptr, s)
                         : wc_launch<double>(which, E, u, D, R, a, b, nullptr, nullptr, ldw, 0, out, nullptr, s);
}

extern "C" int aa_weighted_channels_sum(aa_dtype dtype, int64_t E, int u, int l_max, int shared, const void* sh, const void* w, int64_t ldw,
                                        const void* sh2, const void* w2, int64_t ldw2, void* out, aa_stream stream) {
  const int D = (l_max + 1) * (l_max + 1), R = shared ? 1 : l_max + 1;
  if (int rc = wc_check("aa_weighted_channels_sum", E, u, l_max, R, ldw, ldw2)) return rc;
  AA_REQUIRE(E == 0 || (sh && w && sh2 && w2 && out), "aa_weighted_channels_sum: null argument");
  hipStream_t s = static_cast<hipStream_t>(stream);
  return dtype == AA_F32 ? wc_launch<float>(3, E, u, D, R, sh, w, sh2, w2, ldw, ldw2, out, nullptr, s)
                         : wc_launch<double>(3, E, u, D, R, sh, w, sh2, w2, ldw, ldw2, out, nullptr, s);
}

extern "C" int aa_weighted_channels_pair(aa_dtype dtype, int64_t E, int u, int l_max, int shared, const void* t, const void* sh, const void* w,
                                         int64_t ldw, void* out_sh, void* out_w, aa_stream stream) {
  const int D = (l_max + 1) * (l_max + 1), R = shared ? 1 : l_max + 1;
  if (int rc = wc_check("aa_weighted_channels_pair", E, u, l_max, R, ldw, ldw)) return rc;
  AA_REQUIRE(E == 0 || (t && sh && w && out_sh && out_w), "aa_weighted_channels_pair: null argument");
  hipStream_t s = static_cast<hipStream_t>(stream);
  return dtype == AA_F32 ? wc_launch<float>(4, E, u, D, R, t, sh, w, nullptr, ldw, ldw, out_sh, out_w, s)
                         : wc_launch<double>(4, E, u, D, R, t, sh, w, nullptr, ldw, ldw, out_sh, out_w, s);
}

static int silu_launch(const char* what, aa_dtype dtype, int order, int64_t n, const void* x, const void* g, const void* h, void* out0, void* out1,
                       aa_stream stream, int act = aa::AA_ACT_SILU) {
  if (!(n >= 0 && order >= 0 && order + (h ? 1 : 0) <= 3)) return aa::fail(AA_ERR_INVALID, std::string(what) + ": bad argument (derivative orders 0..3)");
  if (!(act == aa::AA_ACT_SILU || act == aa::AA_ACT_MISH || act == aa::AA_ACT_GELU)) return aa::fail(AA_ERR_INVALID, std::string(what) + ": activation 0 (silu), 1 (mish) or 2 (gelu)");
  if (n == 0) return AA_OK;
  if (!(x && out0 && (!h || (g && out1)))) return aa::fail(AA_ERR_INVALID, std::string(what) + ": null argument");
  for (const void* p : {x, g, h, static_cast<const void*>(out0), static_cast<const void*>(out1)})
    if (reinterpret_cast<uintptr_t>(p) % 16 != 0) return aa::fail(AA_ERR_INVALID, std::string(what) + ": pointers must be 16-byte aligned");
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int64_t per_block = 256 * (dtype == AA_F32 ? 4 : 2);
  const dim3 grid((unsigned)((n + per_block - 1) / per_block));
  if (dtype == AA_F32)
    hipLaunchKernelGGL(aa::silu_family_kernel<float>, grid, dim3(256), 0, s, n, order, static_cast<const float*>(x), static_cast<const float*>(g),
                       static_cast<const float*>(h), static_cast<float*>(out0), static_cast<float*>(out1), act);
  else
    hipLaunchKernelGGL(aa::silu_family_kernel<double>, grid, dim3(256), 0, s, n, order, static_cast<const double*>(x), static_cast<const double*>(g),
                       static_cast<const double*>(h), static_cast<double*>(out0), static_cast<double*>(out1), act);
  AA_CHECK_HIP(hipGetLastError());
  return AA_OK;
}

extern "C" int aa_silu_derivative(aa_dtype dtype, int order, int64_t n, const void* x, const void* g, void* out, aa_stream stream) {
  return silu_launch("aa_silu_derivative", dtype, order, n, x, g, nullptr, out, nullptr, stream);
}

extern "C" int aa_silu_derivative_pair(aa_dtype dtype, int order, int64_t n, const void* x, const void* g, const void* h, void* out_x, void* out_g,
                                       aa_stream stream) {
  if (!h && n > 0) return aa::fail(AA_ERR_INVALID, "aa_silu_derivative_pair: null argument");
  return silu_launch("aa_silu_derivative_pair", dtype, order, n, x, g, h, out_x, out_g, stream);
}

extern "C" int aa_act_derivative(aa_dtype dtype, int act, int order, int64_t n, const void* x, const void* g, void* out, aa_stream stream) {
  return silu_launch("aa_act_derivative", dtype, order, n, x, g, nullptr, out, nullptr, stream, act);
}

extern "C" int aa_act_derivative_pair(aa_dtype dtype, int act, int order, int64_t n, const void* x, const void* g, const void* h, void* out_x,
                                      void* out_g, aa_stream stream) {
  if (!h && n > 0) return aa::fail(AA_ERR_INVALID, "aa_act_derivative_pair: null argument");
  return silu_launch("aa_act_derivative_pair", dtype, order, n, x, g, h, out_x, out_g, stream, act);
}

extern "C" int aa_scalar_column(aa_dtype dtype, int64_t rows, int D, const void* a, const void* s, void* out, aa_stream stream) {
  AA_REQUIRE(rows >= 0 && D >= 1 && D <= 4096, "aa_scalar_column: bad argument");
  if (rows == 0) return AA_OK;
  AA_REQUIRE(s && out, "aa_scalar_column: null argument");
  AA_REQUIRE(reinterpret_cast<uintptr_t>(a) % 16 == 0 && reinterpret_cast<uintptr_t>(out) % 16 == 0, "aa_scalar_column: a and out must be 16-byte aligned");
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int64_t n = rows * D;
  const dim3 grid((unsigned)((n + 1023) / 1024));
#define AA_SC(T, DD)                                                                                                                         \
  hipLaunchKernelGGL((aa::scalar_column_kernel<T, DD>), grid, dim3(256), 0, st, n, D, static_cast<const T*>(a), static_cast<const T*>(s), \
                     static_cast<T*>(out))
  if (dtype == AA_F32) {
    if (D == 4) AA_SC(float, 4); else if (D == 9) AA_SC(float, 9); else if (D == 16) AA_SC(float, 16); else AA_SC(float, 0);
  } else {
    if (D == 4) AA_SC(double, 4); else if (D == 9) AA_SC(double, 9); else if (D == 16) AA_SC(double, 16); else AA_SC(double, 0);
  }
#undef AA_SC
  AA_CHECK_HIP(hipGetLastError());
  return AA_OK;
}

extern "C" int aa_concat_columns(aa_dtype dtype, int64_t E, int n, const void* const* xs, const int64_t* ldx, const int* widths, void* out, int64_t ldo,
                                 aa_stream stream) {
  AA_REQUIRE(E >= 0 && n >= 1 && n <= aa::kCatMax && xs && ldx && widths, "aa_concat_columns: 1..8 inputs");
  aa::CatArgs a{};
  const int esize = dtype == AA_F32 ? 4 : 8, vec = 16 / esize;
  int total = 0;
  bool v_ok = (ldo % vec == 0) && reinterpret_cast<uintptr_t>(out) % 16 == 0;
  for (int j = 0; j < n; ++j) {
    AA_REQUIRE(widths[j] >= 1 && ldx[j] >= widths[j] && (E == 0 || xs[j]), "aa_concat_columns: bad input");
    a.x[j] = xs[j];
    a.ld[j] = ldx[j];
    a.width[j] = widths[j];
    a.off[j] = total;
    total += widths[j];
    v_ok = v_ok && (widths[j] % vec == 0) && (ldx[j] % vec == 0) && reinterpret_cast<uintptr_t>(xs[j]) % 16 == 0;
  }
  a.off[n] = total;
  AA_REQUIRE(ldo >= total, "aa_concat_columns: output row stride shorter than the row");
  if (E == 0) return AA_OK;
  AA_REQUIRE(out, "aa_concat_columns: null output");
  a.n = n;
  a.total = total;
  a.E = E;
  a.ldo = ldo;
  a.out = out;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int64_t items = E * (v_ok ? total / vec : total);
  const dim3 grid((unsigned)((items + 255) / 256));
  if (dtype == AA_F32) {
    if (v_ok) hipLaunchKernelGGL((aa::concat_columns_kernel<float, 4>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((aa::concat_columns_kernel<float, 1>), grid, dim3(256), 0, s, a);
  } else {
    if (v_ok) hipLaunchKernelGGL((aa::concat_columns_kernel<double, 2>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((aa::concat_columns_kernel<double, 1>), grid, dim3(256), 0, s, a);
  }
  AA_CHECK_HIP(hipGetLastError());
  return AA_OK;
}
