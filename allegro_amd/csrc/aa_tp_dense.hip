// Specialised kernels of the OPERATOR seam (aa_tp_forward / aa_tp_backward: `Contracter.forward` of the reference,
// allegro/nn/_strided/_contract.py:185-251, and its input gradients) for the layer signatures of standard Allegro
// stacks: compile-time Clebsch-Gordan code (aa_cg_gen.h), operands in registers, lane = channel, one wave per
// (center atom, 64-channel slice).  Same math as the table-driven kernels of aa_tp.hip, which remain the path for
// arbitrary irreps, channel counts that are not multiples of 64, and fp64 at l_max = 3.
//
// The seam's tensors keep the reference's strided layout [E, u, d] (d minor): an edge's 64-channel slice is ONE
// contiguous run of 64 d elements.  A wave moves it with d fully coalesced 256-B accesses (element lane + 64 t) and turns
// it into the lane = channel view (d values per lane) through a wave-private LDS patch with odd row stride -- no
// uncoalesced access, no inter-wave traffic, no atomics; segment sums in a fixed order (bit-reproducible).
// This is what `enable_HipContracter` runs in eval mode and -- through the segmented differentiable form of
// allegro_amd/ops.py -- in training mode; the general kernels process one (edge, channel) pair per thread out of LDS
// tables and are ~6x off their bandwidth bound (bench.py --mode train-op).
#include "aa_cg_gen.h"
#include "aa_common.h"

namespace aa {

namespace {

constexpr int odd(int d) { return d | 1; }

// the wave's [64][D] run at `run` (wave-uniform) -> x[D] of this lane's channel
template <typename T, int D>
__device__ __forceinline__ void run_load(const T* run, T* sS, int lane, T* x) {
  constexpr int DP = odd(D);
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int t = 0; t < D; ++t) {
    const int q = lane + 64 * t;
    sS[(q / D) * DP + q % D] = run[q];
  }
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int i = 0; i < D; ++i) x[i] = sS[lane * DP + i];
}
// x[D] of this lane's channel -> the run in registers (r[t] = element lane + 64 t), ready for coalesced stores
template <typename T, int D>
__device__ __forceinline__ void run_stage(const T* x, T* sS, int lane, T* r) {
  constexpr int DP = odd(D);
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int i = 0; i < D; ++i) sS[lane * DP + i] = x[i];
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int t = 0; t < D; ++t) {
    const int q = lane + 64 * t;
    r[t] = sS[(q / D) * DP + q % D];
  }
}
template <typename T, int D>
__device__ __forceinline__ void run_store(T* run, const T* r, int lane) {
#pragma unroll
  for (int t = 0; t < D; ++t) run[lane + 64 * t] = r[t];
}

template <class Sig>
constexpr int patch_cols() {
  int m = odd(Sig::D1);
  if (odd(Sig::D2) > m) m = odd(Sig::D2);
  if (odd(Sig::DOUT) > m) m = odd(Sig::DOUT);
  return m;
}

struct WaveSlot {
  int64_t atom;
  int ch0;  // first channel of the wave's slice
  bool valid;
};
__device__ __forceinline__ WaveSlot wave_slot(int u, int64_t N) {
  const int wave = threadIdx.x >> 6, wpa = u >> 6;
  WaveSlot w;
  w.atom = int64_t(blockIdx.x) * (4 / wpa) + wave / wpa;
  w.ch0 = (wave % wpa) * 64;
  w.valid = w.atom < N;
  return w;
}

}  // namespace

template <class Sig, typename T>
__global__ __launch_bounds__(256) void tp_dense_fwd_kernel(TpDenseArgs a) {
  constexpr int D1 = Sig::D1, D2 = Sig::D2, DOUT = Sig::DOUT, P = Sig::P;
  const int u = a.u, lane = threadIdx.x & 63;
  const WaveSlot ws = wave_slot(u, a.N);
  T* sS = reinterpret_cast<T*>(aa_smem) + size_t(threadIdx.x >> 6) * 64 * patch_cols<Sig>();
  if (!ws.valid) return;  // (whole waves: no collective below spans waves)
  const int beg = __builtin_amdgcn_readfirstlane(a.rowptr[ws.atom]), end = __builtin_amdgcn_readfirstlane(a.rowptr[ws.atom + 1]);
  const int ch = ws.ch0 + lane;
  const T* X1 = static_cast<const T*>(a.x1);
  const T* X2 = static_cast<const T*>(a.x2);
  auto edge_of = [&](int s) -> int64_t { return a.eids ? a.eids[s] : s; };
  // ---- scale + segment sum of the env operand  (_contract.py:195-204); x2s is also an output (the reverse reads it)
  T x2s[D2];
#pragma unroll
  for (int j = 0; j < D2; ++j) x2s[j] = T(0);
  for (int s = beg; s < end; ++s) {
    T x[D2];
    run_load<T, D2>(X2 + (edge_of(s) * u + ws.ch0) * D2, sS, lane, x);
#pragma unroll
    for (int j = 0; j < D2; ++j) x2s[j] += x[j];
  }
  const T sf = T(a.sf);
#pragma unroll
  for (int j = 0; j < D2; ++j) x2s[j] *= sf;
  {
    T r[D2];
    run_stage<T, D2>(x2s, sS, lane, r);
    run_store<T, D2>(static_cast<T*>(a.x2s) + (ws.atom * u + ws.ch0) * D2, r, lane);
  }
  T w[P];
  {
    const T* W = static_cast<const T*>(a.weights);
#pragma unroll
    for (int p = 0; p < P; ++p) w[p] = a.coupling ? W[ch * P + p] : W[p];
  }
  // ---- contraction per edge with the gathered sum  (_contract.py:205-251)
  for (int s = beg; s < end; ++s) {
    const int64_t e = edge_of(s);
    T x1[D1], out[DOUT], r[DOUT];
    run_load<T, D1>(X1 + (e * u + ws.ch0) * D1, sS, lane, x1);
    Sig::template fwd<T>(x1, x2s, w, out);
    run_stage<T, DOUT>(out, sS, lane, r);
    run_store<T, DOUT>(static_cast<T*>(a.out) + (e * u + ws.ch0) * DOUT, r, lane);
  }
}

template <class Sig, typename T, bool G1, bool G2>
__global__ __launch_bounds__(256) void tp_dense_bwd_kernel(TpDenseArgs a) {
  constexpr int D1 = Sig::D1, D2 = Sig::D2, DOUT = Sig::DOUT, P = Sig::P;
  const int u = a.u, lane = threadIdx.x & 63;
  const WaveSlot ws = wave_slot(u, a.N);
  T* sS = reinterpret_cast<T*>(aa_smem) + size_t(threadIdx.x >> 6) * 64 * patch_cols<Sig>();
  if (!ws.valid) return;
  const int beg = __builtin_amdgcn_readfirstlane(a.rowptr[ws.atom]), end = __builtin_amdgcn_readfirstlane(a.rowptr[ws.atom + 1]);
  const int ch = ws.ch0 + lane;
  const T* X1 = static_cast<const T*>(a.x1);
  const T* GO = static_cast<const T*>(a.gout);
  auto edge_of = [&](int s) -> int64_t { return a.eids ? a.eids[s] : s; };
  T x2s[D2], g2acc[D2];
  if constexpr (G1) run_load<T, D2>(static_cast<const T*>(a.x2s) + (ws.atom * u + ws.ch0) * D2, sS, lane, x2s);
#pragma unroll
  for (int j = 0; j < D2; ++j) g2acc[j] = T(0);
  T w[P];
  {
    const T* W = static_cast<const T*>(a.weights);
#pragma unroll
    for (int p = 0; p < P; ++p) w[p] = a.coupling ? W[ch * P + p] : W[p];
  }
  // ---- per edge: gradient of x1 (G1), accumulation of the gradient of the gathered sum (G2); a gradient the caller
  //      did not ask for costs neither its operand stream nor its stores
  for (int s = beg; s < end; ++s) {
    const int64_t e = edge_of(s);
    T go[DOUT];
    run_load<T, DOUT>(GO + (e * u + ws.ch0) * DOUT, sS, lane, go);
    if constexpr (G2) {
      T x1[D1], g2[D2];
      run_load<T, D1>(X1 + (e * u + ws.ch0) * D1, sS, lane, x1);
      Sig::template bx2<T>(go, x1, w, g2);
#pragma unroll
      for (int j = 0; j < D2; ++j) g2acc[j] += g2[j];
    }
    if constexpr (G1) {
      T g1[D1], r[D1];
      Sig::template bx1<T>(go, x2s, w, g1);
      run_stage<T, D1>(g1, sS, lane, r);
      run_store<T, D1>(static_cast<T*>(a.gx1) + (e * u + ws.ch0) * D1, r, lane);
    }
  }
  // ---- adjoint of scale + segment sum + gather: every edge of the segment receives the scaled accumulated gradient
  if constexpr (G2) {
    const T sf = T(a.sf);
#pragma unroll
    for (int j = 0; j < D2; ++j) g2acc[j] *= sf;
    T r2[D2];
    run_stage<T, D2>(g2acc, sS, lane, r2);
    for (int s = beg; s < end; ++s) run_store<T, D2>(static_cast<T*>(a.gx2) + (edge_of(s) * u + ws.ch0) * D2, r2, lane);
  }
}

// Path-weight gradient: wave = (slot of consecutive center atoms, 64-channel slice); lane-private accumulators over the
// slot's edges in CSR order, one partial [u][P] slab per slot, summed in slot order by the general path's reduce kernel
// (aa_tp.hip) -- deterministic, no atomics.  (_contract.py:205-251 differentiated w.r.t. `weights`.)
template <class Sig, typename T>
__global__ __launch_bounds__(256) void tp_dense_wgrad_kernel(TpLayerWgradArgs a, int u, int nslots, int atoms_per_slot) {
  constexpr int D1 = Sig::D1, D2 = Sig::D2, DOUT = Sig::DOUT, P = Sig::P;
  const int lane = threadIdx.x & 63, wpa = u >> 6;
  const int gwave = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int slot = gwave / wpa, ch0 = (gwave % wpa) * 64;
  T* sS = reinterpret_cast<T*>(aa_smem) + size_t(threadIdx.x >> 6) * 64 * patch_cols<Sig>();
  if (slot >= nslots) return;  // (whole waves)
  const int64_t n0 = int64_t(slot) * atoms_per_slot;
  const int64_t n1 = n0 + atoms_per_slot < a.N ? n0 + atoms_per_slot : a.N;
  const T* X1 = static_cast<const T*>(a.x1);
  const T* GO = static_cast<const T*>(a.gout);
  T acc[P];
#pragma unroll
  for (int p = 0; p < P; ++p) acc[p] = T(0);
  for (int64_t n = n0; n < n1; ++n) {
    const int beg = __builtin_amdgcn_readfirstlane(a.rowptr[n]), end = __builtin_amdgcn_readfirstlane(a.rowptr[n + 1]);
    if (beg >= end) continue;
    T x2s[D2];
    run_load<T, D2>(static_cast<const T*>(a.x2s) + (n * u + ch0) * D2, sS, lane, x2s);
    for (int s = beg; s < end; ++s) {
      const int64_t e = a.eids ? a.eids[s] : s;
      T go[DOUT], x1[D1];
      run_load<T, DOUT>(GO + (e * u + ch0) * DOUT, sS, lane, go);
      run_load<T, D1>(X1 + (e * u + ch0) * D1, sS, lane, x1);
      Sig::template bw<T>(go, x1, x2s, acc);
    }
  }
  T* part = static_cast<T*>(a.partial) + (int64_t(slot) * u + ch0 + lane) * P;
#pragma unroll
  for (int p = 0; p < P; ++p) part[p] = acc[p];
}

bool tp_dense_supported(int sig, int u, int dtype) {
  if (sig < 0 || sig >= cg::kNumSigs || u < 64 || u > 256 || (u & 63)) return false;
  if (256 % u != 0) return false;  // 64, 128, 256: whole atoms per workgroup
  // (fp64 at l_max = 3: the straight-line code of the 353- / 611-term signatures does not fit the register file)
  return !(dtype == AA_F64 && cg::kSigs[sig].lmax >= 3);
}

template <typename T>
int launch_tp_dense(int sig, bool backward, const TpDenseArgs& a, hipStream_t stream) {
  if (a.N == 0 || a.E == 0) {
    if (!backward && a.N > 0) AA_CHECK_HIP(hipMemsetAsync(a.x2s, 0, sizeof(T) * size_t(a.N) * a.u * cg::kSigs[sig].d2, stream));
    return AA_OK;
  }
  const int apb = 4 / (a.u >> 6);
  dim3 grid((unsigned)((a.N + apb - 1) / apb));
  switch (sig) {
#define AA_CASE(ID, SIG)                                                                                       \
  case ID: {                                                                                                   \
    if constexpr (sizeof(T) == 8 && cg::SIG::LMAX >= 3) {                                                      \
      return fail(AA_ERR_INVALID, "tp dense: fp64 at l_max = 3 runs the general kernels");                     \
    } else {                                                                                                   \
      const size_t smem = sizeof(T) * 4 * 64 * patch_cols<cg::SIG>();                                          \
      if (backward && a.gx1 && a.gx2)                                                                          \
        hipLaunchKernelGGL((tp_dense_bwd_kernel<cg::SIG, T, true, true>), grid, dim3(256), smem, stream, a);   \
      else if (backward && a.gx1)                                                                              \
        hipLaunchKernelGGL((tp_dense_bwd_kernel<cg::SIG, T, true, false>), grid, dim3(256), smem, stream, a);  \
      else if (backward)                                                                                       \
        hipLaunchKernelGGL((tp_dense_bwd_kernel<cg::SIG, T, false, true>), grid, dim3(256), smem, stream, a);  \
      else                                                                                                     \
        hipLaunchKernelGGL((tp_dense_fwd_kernel<cg::SIG, T>), grid, dim3(256), smem, stream, a);               \
    }                                                                                                          \
    break;                                                                                                     \
  }
    AA_FOREACH_SIG(AA_CASE)
#undef AA_CASE
    default:
      return fail(AA_ERR_INVALID, "tp dense: unknown signature");
  }
  AA_CHECK_HIP(hipGetLastError());
  return AA_OK;
}
template <typename T>
int launch_tp_dense_wgrad(int sig, int u, int coupling, const TpLayerWgradArgs& a, hipStream_t stream) {
  const int P = cg::kSigs[sig].num_paths;
  if (a.N == 0 || a.E == 0) {
    AA_CHECK_HIP(hipMemsetAsync(a.gw, 0, sizeof(T) * size_t(coupling ? u : 1) * P, stream));
    return AA_OK;
  }
  const int nslots = tp_wgrad_slots(a.N, kDenseWgradSlots);
  const int aps = int((a.N + nslots - 1) / nslots);
  const int waves = nslots * (u >> 6);
  dim3 grid((unsigned)((waves + 3) / 4));
  switch (sig) {
#define AA_CASE(ID, SIG)                                                                                          \
  case ID: {                                                                                                      \
    if constexpr (sizeof(T) == 8 && cg::SIG::LMAX >= 3) {                                                         \
      return fail(AA_ERR_INVALID, "tp dense: fp64 at l_max = 3 runs the general kernels");                        \
    } else {                                                                                                      \
      const size_t smem = sizeof(T) * 4 * 64 * patch_cols<cg::SIG>();                                             \
      hipLaunchKernelGGL((tp_dense_wgrad_kernel<cg::SIG, T>), grid, dim3(256), smem, stream, a, u, nslots, aps);  \
    }                                                                                                             \
    break;                                                                                                        \
  }
    AA_FOREACH_SIG(AA_CASE)
#undef AA_CASE
    default:
      return fail(AA_ERR_INVALID, "tp dense: unknown signature");
  }
  AA_CHECK_HIP(hipGetLastError());
  return launch_tp_wgrad_reduce<T>(a.partial, nslots, u, P, coupling, a.gw, stream);
}
template int launch_tp_dense_wgrad<float>(int, int, int, const TpLayerWgradArgs&, hipStream_t);
template int launch_tp_dense_wgrad<double>(int, int, int, const TpLayerWgradArgs&, hipStream_t);
template int launch_tp_dense<float>(int, bool, const TpDenseArgs&, hipStream_t);
template int launch_tp_dense<double>(int, bool, const TpDenseArgs&, hipStream_t);

}  // namespace aa
