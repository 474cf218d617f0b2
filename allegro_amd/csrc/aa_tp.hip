// Strided Clebsch-Gordan tensor-product layer of the Allegro hot path (gfx950), general
// (table-driven) form: any irreps, any multiplicity, both weight modes, fp32/fp64.
//
// One workgroup owns one center atom's edge segment (edges are center-sorted), which fuses what the
// reference does in three full passes over [E,u,d] (allegro/nn/_strided/_contract.py:195-205:
// scale, scatter-sum by center, index_select with the SAME index) with the contraction (:213-251):
//   phase 1  x2s[ch,j] = f * sum_{e in segment} x2[e,ch,j]        -> LDS (+ saved for the backward)
//   phase 2  out[e,ch,k] = sum_paths W[ch,p] * sum_nz c_nz x1[e,ch,i] x2s[ch,j]
// x1/x2 may be "implicit" weighted spherical harmonics  sh[e,i] * w[e,ch,irrep(i)]
// (MakeWeightedChannels, allegro/nn/_strided/_channels.py:44-57) so that [E,u,D] env tensors are
// never written to HBM.  The sparse non-zeros of w3j replace the dense [z,u,i,j,k] product of
// _contract.py:236-241 (186 KB/edge at l_max=2).
//
// Lane mapping: thread <-> (edge, channel) pair; per-thread operand rows live in LDS with an odd row
// stride (conflict-free ds_read_b32), the CG table is wave-uniform (scalar loads).
#include <algorithm>
#include <tuple>

#include "aa_common.h"

namespace aa {

__device__ __forceinline__ float entry_val(const TpEntry& e, float) { return e.val_f; }
__device__ __forceinline__ double entry_val(const TpEntry& e, double) { return e.val_d; }

__host__ __device__ inline int odd_pad(int d) { return d | 1; }

template <typename T>
__device__ __forceinline__ T load_operand(const TpOperand& op, int64_t e, int ch, int i, int u, int d, int R) {
  if (op.dense) return static_cast<const T*>(op.dense)[(e * u + ch) * d + i];
  return static_cast<const T*>(op.sh)[e * op.ld_sh + i] * static_cast<const T*>(op.w)[e * op.ldw + ch * R + sh_l_of(i)];
}

// sum `v` over the channels of one edge into LDS slot *dst (dst is per (edge, component)).
template <typename T>
__device__ __forceinline__ void channel_reduce_add(T v, T* dst, bool wave_uniform, bool active) {
  if (wave_uniform) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    if ((threadIdx.x & 63) == 0 && active) atomicAdd(dst, v);
  } else {
    if (active) atomicAdd(dst, v);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void tp_layer_fwd_kernel(TpLayerDev L, TpLayerFwdArgs a) {
  const int u = L.mul, d1 = L.d1, d2 = L.d2, dout = L.dout, P = L.num_paths;
  const int d1p = odd_pad(d1), d2p = odd_pad(d2);
  const int R2 = a.x2.dense ? 0 : sh_l_of(d2 - 1) + 1;
  const int R1 = a.x1.dense ? 0 : sh_l_of(d1 - 1) + 1;
  T* sX2 = reinterpret_cast<T*>(aa_smem);  // [u][d2p]
  T* sX1 = sX2 + u * d2p;                  // [256][d1p]
  const int tid = threadIdx.x;
  const int64_t n = blockIdx.x;
  const int beg = a.rowptr[n], end = a.rowptr[n + 1];
  const int deg = end - beg;
  const T sf = T(a.scatter_factor);

  // phase 1: scaled segment sum of the env operand
  for (int idx = tid; idx < u * d2; idx += 256) {
    int ch = idx / d2, j = idx % d2;
    T acc = T(0);
    for (int s = beg; s < end; ++s) {
      int64_t e = a.eids ? a.eids[s] : s;
      acc += load_operand<T>(a.x2, e, ch, j, u, d2, R2);
    }
    acc *= sf;
    sX2[ch * d2p + j] = acc;
    static_cast<T*>(a.x2s)[(n * u + ch) * d2 + j] = acc;
  }
  __syncthreads();

  // phase 2: contraction per (edge, channel)
  const T* W = static_cast<const T*>(a.weights);
  const int npairs = deg * u;
  for (int q0 = 0; q0 < npairs; q0 += 256) {
    int q = q0 + tid;
    if (q >= npairs) continue;  // no collectives below
    int el = q / u, ch = q % u;
    int64_t e = a.eids ? a.eids[beg + el] : (beg + el);
    T* x1 = sX1 + tid * d1p;
    for (int i = 0; i < d1; ++i) x1[i] = load_operand<T>(a.x1, e, ch, i, u, d1, R1);
    const T* x2 = sX2 + ch * d2p;
    T cur = T(0);
    for (int gi = 0; gi < L.fwd.num_groups; ++gi) {
      TpGroup grp = L.fwd.groups[gi];
      T acc = T(0);
      for (int nz = grp.begin; nz < grp.end; ++nz) {
        TpEntry en = L.fwd.entries[nz];
        acc += entry_val(en, T(0)) * x1[en.a] * x2[en.b];
      }
      if (grp.end > grp.begin) {
        cur += (L.coupling ? W[ch * P + grp.path] : W[grp.path]) * acc;
      }
      bool last = (gi + 1 == L.fwd.num_groups) || (L.fwd.groups[gi + 1].out_idx != grp.out_idx);
      if (last) {
        if (a.out) static_cast<T*>(a.out)[(e * u + ch) * dout + grp.out_idx] = cur;
        if (a.scal && grp.out_idx == 0) static_cast<T*>(a.scal)[e * a.ld_scal + ch] = cur;
        cur = T(0);
      }
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void tp_layer_bwd_kernel(TpLayerDev L, TpLayerBwdArgs a, int EC) {
  const int u = L.mul, d1 = L.d1, d2 = L.d2, dout = L.dout, P = L.num_paths;
  const int d1p = odd_pad(d1), d2p = odd_pad(d2), dop = odd_pad(dout);
  const int R1 = a.x1.dense ? 0 : sh_l_of(d1 - 1) + 1;
  const int R2 = a.x2.dense ? 0 : sh_l_of(d2 - 1) + 1;
  const int Dsh = d1 > d2 ? d1 : d2;  // row width of the sh-gradient staging
  T* sX2 = reinterpret_cast<T*>(aa_smem);  // [u][d2p]   x2s of this atom
  T* sG2 = sX2 + u * d2p;                  // [u][d2p]   grad wrt x2s (accumulated over the segment)
  T* sX1 = sG2 + u * d2p;                  // [256][d1p]
  T* sGo = sX1 + 256 * d1p;                // [256][dop]
  T* sGY = sGo + 256 * dop;                // [EC][Dsh]
  const int tid = threadIdx.x;
  const int64_t n = blockIdx.x;
  const int beg = a.rowptr[n], end = a.rowptr[n + 1];
  const int deg = end - beg;
  const T sf = T(a.scatter_factor);
  const bool wave_uniform = (u % 64) == 0;
  const T* W = static_cast<const T*>(a.weights);

  // a gradient the caller did not ask for is skipped together with the operand only it reads (aa_tp_backward with a
  // null gx1 / gx2: the single partial contractions of the training path, allegro_amd/ops.py)
  const bool need1 = a.g1.dense || a.g1.gw, need2 = a.g2.dense || a.g2.gw;
  for (int idx = tid; idx < u * d2; idx += 256) {
    int ch = idx / d2, j = idx % d2;
    sX2[ch * d2p + j] = need1 ? static_cast<const T*>(a.x2s)[(n * u + ch) * d2 + j] : T(0);
    sG2[ch * d2p + j] = T(0);
  }
  __syncthreads();

  const int npairs = deg * u;
  const bool need_gsh1 = (!a.x1.dense) && a.g1.gsh;
  for (int q0 = 0; q0 < npairs; q0 += 256) {
    if (need_gsh1) {
      for (int idx = tid; idx < EC * Dsh; idx += 256) sGY[idx] = T(0);
      __syncthreads();
    }
    int q = q0 + tid;
    bool active = q < npairs;
    int el = active ? q / u : 0, ch = active ? q % u : 0;
    int el0 = q0 / u;
    int64_t e = active ? (a.eids ? a.eids[beg + el] : (beg + el)) : 0;
    T* x1 = sX1 + tid * d1p;
    T* go = sGo + tid * dop;
    if (active) {
      for (int i = 0; i < d1; ++i) x1[i] = (a.x1.dense || a.x1.sh) ? load_operand<T>(a.x1, e, ch, i, u, d1, R1) : T(0);  // (absent: only g1 was asked for)
      for (int k = 0; k < dout; ++k) go[k] = a.gout ? static_cast<const T*>(a.gout)[(e * u + ch) * dout + k] : T(0);
      if (a.gscal) go[0] += static_cast<const T*>(a.gscal)[e * a.ld_gscal + ch];
    } else {
      for (int i = 0; i < d1; ++i) x1[i] = T(0);
      for (int k = 0; k < dout; ++k) go[k] = T(0);
    }
    const T* x2 = sX2 + ch * d2p;
    // ---- grad wrt x1: g1[i] = sum W * c * gout[k] * x2s[j]
    if (need1) {
      T cur = T(0), gw_acc = T(0);
      int r_cur = 0;
      for (int gi = 0; gi < L.bx1.num_groups; ++gi) {
        TpGroup grp = L.bx1.groups[gi];
        T acc = T(0);
        for (int nz = grp.begin; nz < grp.end; ++nz) {
          TpEntry en = L.bx1.entries[nz];
          acc += entry_val(en, T(0)) * go[en.a] * x2[en.b];
        }
        if (grp.end > grp.begin) cur += (L.coupling ? W[ch * P + grp.path] : W[grp.path]) * acc;
        bool last = (gi + 1 == L.bx1.num_groups) || (L.bx1.groups[gi + 1].out_idx != grp.out_idx);
        if (last) {
          int i = grp.out_idx;
          if (a.g1.dense) {
            if (active) static_cast<T*>(a.g1.dense)[(e * u + ch) * d1 + i] = cur;
          } else if (a.g1.gw) {
            // implicit x1 = sh[e,i] * w[e,ch,r(i)]
            int r = sh_l_of(i);
            if (r != r_cur) {
              if (active) static_cast<T*>(a.g1.gw)[e * a.g1.ldgw + ch * R1 + r_cur] = gw_acc;
              gw_acc = T(0);
              r_cur = r;
            }
            T shv = active ? static_cast<const T*>(a.x1.sh)[e * a.x1.ld_sh + i] : T(0);
            T wv = active ? static_cast<const T*>(a.x1.w)[e * a.x1.ldw + ch * R1 + r] : T(0);
            gw_acc += cur * shv;
            if (need_gsh1) channel_reduce_add<T>(cur * wv, sGY + (el - el0) * Dsh + i, wave_uniform, active);
          }
          cur = T(0);
        }
      }
      if (!a.g1.dense && a.g1.gw && active) static_cast<T*>(a.g1.gw)[e * a.g1.ldgw + ch * R1 + r_cur] = gw_acc;
    }
    // ---- grad wrt x2s: g2[j] = sum W * c * gout[k] * x1[i], summed over the segment
    if (need2) {
      T cur = T(0);
      for (int gi = 0; gi < L.bx2.num_groups; ++gi) {
        TpGroup grp = L.bx2.groups[gi];
        T acc = T(0);
        for (int nz = grp.begin; nz < grp.end; ++nz) {
          TpEntry en = L.bx2.entries[nz];
          acc += entry_val(en, T(0)) * go[en.a] * x1[en.b];
        }
        if (grp.end > grp.begin) cur += (L.coupling ? W[ch * P + grp.path] : W[grp.path]) * acc;
        bool last = (gi + 1 == L.bx2.num_groups) || (L.bx2.groups[gi + 1].out_idx != grp.out_idx);
        if (last) {
          if (active) atomicAdd(&sG2[ch * d2p + grp.out_idx], cur);
          cur = T(0);
        }
      }
    }
    if (need_gsh1) {
      __syncthreads();
      int ne = (q0 + 256 < npairs ? q0 + 256 : npairs);
      int el1 = (ne - 1) / u;  // last edge touched in this chunk
      for (int idx = tid; idx < (el1 - el0 + 1) * d1; idx += 256) {
        int le = idx / d1, i = idx % d1;
        int64_t ee = a.eids ? a.eids[beg + el0 + le] : (beg + el0 + le);
        static_cast<T*>(a.g1.gsh)[ee * a.g1.ld_gsh + i] += sGY[le * Dsh + i];
      }
      __syncthreads();
    }
  }
  __syncthreads();

  // ---- phase C: adjoint of (scale, segment-sum, gather): every edge of the segment receives f * g_x2s
  if (a.g2.dense) {
    for (int idx = tid; idx < deg * u * d2; idx += 256) {
      int el = idx / (u * d2), rem = idx % (u * d2);
      int ch = rem / d2, j = rem % d2;
      int64_t e = a.eids ? a.eids[beg + el] : (beg + el);
      static_cast<T*>(a.g2.dense)[(e * u + ch) * d2 + j] = sf * sG2[ch * d2p + j];
    }
  } else if (a.g2.gw) {
    for (int q0 = 0; q0 < npairs; q0 += 256) {
      if (a.g2.gsh) {
        for (int idx = tid; idx < EC * Dsh; idx += 256) sGY[idx] = T(0);
        __syncthreads();
      }
      int q = q0 + tid;
      bool active = q < npairs;
      int el = active ? q / u : 0, ch = active ? q % u : 0;
      int el0 = q0 / u;
      int64_t e = active ? (a.eids ? a.eids[beg + el] : (beg + el)) : 0;
      const T* shp = static_cast<const T*>(a.x2.sh) + e * a.x2.ld_sh;
      const T* wp = static_cast<const T*>(a.x2.w) + e * a.x2.ldw + ch * R2;
      int j = 0;
      for (int r = 0; r < R2; ++r) {
        T acc = T(0);
        T wv = active ? wp[r] : T(0);
        for (int m = 0; m < 2 * r + 1; ++m, ++j) {
          T g = sf * sG2[ch * d2p + j];
          acc += (active ? shp[j] : T(0)) * g;
          if (a.g2.gsh) channel_reduce_add<T>(wv * g, sGY + (el - el0) * Dsh + j, wave_uniform, active);
        }
        if (active) static_cast<T*>(a.g2.gw)[e * a.g2.ldgw + ch * R2 + r] = acc;
      }
      if (a.g2.gsh) {
        __syncthreads();
        int ne = (q0 + 256 < npairs ? q0 + 256 : npairs);
        int el1 = (ne - 1) / u;
        for (int idx = tid; idx < (el1 - el0 + 1) * d2; idx += 256) {
          int le = idx / d2, jj = idx % d2;
          int64_t ee = a.eids ? a.eids[beg + el0 + le] : (beg + el0 + le);
          static_cast<T*>(a.g2.gsh)[ee * a.g2.ld_gsh + jj] += sGY[le * Dsh + jj];
        }
        __syncthreads();
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Path-weight gradient (training through the operator seam; the reference's eager Contracter gets it from
// autograd through `weights`, _contract.py:172-177,219):
//   gw[ch,p] = sum_e go[e,ch,k] * sum_{nz in (k,p)} c_nz x1[e,ch,i] x2s[center(e),ch,j]      (coupled)
//   gw[p]    = sum_ch of the above                                                            (uncoupled)
// Deterministic: every workgroup owns a contiguous range of center atoms and writes one partial [u][P] slab
// (thread-private accumulators, fixed reduction order inside the block); a second kernel sums the slabs in
// block order.  No atomics.
// ---------------------------------------------------------------------------------------------
constexpr int kWgThreads = 128;

template <typename T>
__global__ __launch_bounds__(kWgThreads) void tp_layer_wgrad_kernel(TpLayerDev L, TpLayerWgradArgs a, int atoms_per_block) {
  const int u = L.mul, d1 = L.d1, d2 = L.d2, dout = L.dout, P = L.num_paths;
  const int d1p = odd_pad(d1), d2p = odd_pad(d2), dop = odd_pad(dout), Pp = odd_pad(P);
  T* sX2 = reinterpret_cast<T*>(aa_smem);  // [u][d2p]
  T* sX1 = sX2 + u * d2p;                  // [TPB][d1p]
  T* sGo = sX1 + kWgThreads * d1p;         // [TPB][dop]
  T* sW = sGo + kWgThreads * dop;          // [TPB][Pp] thread-private accumulators
  const int tid = threadIdx.x;
  const int64_t n0 = int64_t(blockIdx.x) * atoms_per_block;
  const int64_t n1 = n0 + atoms_per_block < a.N ? n0 + atoms_per_block : a.N;
  T* part = static_cast<T*>(a.partial) + int64_t(blockIdx.x) * u * P;
  const int cw = u < kWgThreads ? u : kWgThreads;  // channels handled per pass
  const int epb = kWgThreads / cw;                 // edges in flight per pass
  const bool lane_on = tid < epb * cw;
  for (int cb = 0; cb < u; cb += cw) {
    const int ch = cb + tid % cw;
    const bool ch_ok = lane_on && ch < u;
    for (int p = 0; p < P; ++p) sW[tid * Pp + p] = T(0);
    for (int64_t n = n0; n < n1; ++n) {
      const int beg = a.rowptr[n], end = a.rowptr[n + 1];
      if (beg >= end) continue;  // (uniform over the block)
      __syncthreads();
      for (int idx = tid; idx < u * d2; idx += kWgThreads)
        sX2[(idx / d2) * d2p + idx % d2] = static_cast<const T*>(a.x2s)[n * u * d2 + idx];
      __syncthreads();
      for (int s0 = beg; s0 < end; s0 += epb) {
        const int s = s0 + tid / cw;
        if (!ch_ok || s >= end) continue;  // no collectives below
        const int64_t e = a.eids ? a.eids[s] : s;
        T* x1 = sX1 + tid * d1p;
        T* go = sGo + tid * dop;
        for (int i = 0; i < d1; ++i) x1[i] = static_cast<const T*>(a.x1)[(e * u + ch) * d1 + i];
        for (int k = 0; k < dout; ++k) go[k] = static_cast<const T*>(a.gout)[(e * u + ch) * dout + k];
        const T* x2 = sX2 + ch * d2p;
        for (int gi = 0; gi < L.fwd.num_groups; ++gi) {
          const TpGroup grp = L.fwd.groups[gi];
          T acc = T(0);
          for (int nz = grp.begin; nz < grp.end; ++nz) {
            const TpEntry en = L.fwd.entries[nz];
            acc += entry_val(en, T(0)) * x1[en.a] * x2[en.b];
          }
          if (grp.end > grp.begin) sW[tid * Pp + grp.path] += go[grp.out_idx] * acc;
        }
      }
    }
    __syncthreads();
    // fixed-order sum over the epb threads that share a channel
    if (tid < cw && cb + tid < u) {
      for (int p = 0; p < P; ++p) {
        T v = T(0);
        for (int m = 0; m < epb; ++m) v += sW[(tid + m * cw) * Pp + p];
        part[(cb + tid) * P + p] = v;
      }
    }
    __syncthreads();
  }
}

// Sum of the rows of a [rows, M] slab array, deterministic (the order depends on rows and M only) and parallel along BOTH axes: the
// first launch adds the rows in chunks of 64 (workgroup (x, y): columns 64 x .. 64 x + 63 of the rows 64 y .. 64 y + 63; thread
// (column, q) rows q, q + 4, ... with four independent partial sums so that sixteen loads are in flight per thread, combined
// ((a0 + a1) + (a2 + a3)), the four q through LDS in order) and leaves each chunk's sum IN the chunk's first row -- nobody else reads
// that row; the second launch adds the chunk sums the same way.  (The round-4 form walked all rows of 64 columns in one workgroup:
// 11 workgroups x 512 dependent loads for a path-weight gradient at 10^4 atoms, 183 us for 23 MB.)
template <typename T>
__global__ __launch_bounds__(256) void column_sum_kernel(const T* src, int rows, int64_t row_stride, int chunk, int64_t M, T* dst,
                                                         int64_t dst_row_stride) {
  T* lds = reinterpret_cast<T*>(aa_smem);  // [4][64]
  const int c = threadIdx.x & 63, q = threadIdx.x >> 6;
  const int64_t col = int64_t(blockIdx.x) * 64 + c;
  const int r0 = blockIdx.y * chunk, r1 = rows < r0 + chunk ? rows : r0 + chunk;
  T a0 = T(0), a1 = T(0), a2 = T(0), a3 = T(0);
  if (col < M) {
    const T* p = src + col;
    int r = r0 + q;
    for (; r + 12 < r1; r += 16) {
      a0 += p[int64_t(r) * row_stride];
      a1 += p[int64_t(r + 4) * row_stride];
      a2 += p[int64_t(r + 8) * row_stride];
      a3 += p[int64_t(r + 12) * row_stride];
    }
    for (; r < r1; r += 4) a0 += p[int64_t(r) * row_stride];
  }
  lds[threadIdx.x] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  if (q == 0 && col < M) dst[int64_t(blockIdx.y) * dst_row_stride + col] = (lds[c] + lds[64 + c]) + (lds[128 + c] + lds[192 + c]);
}

template <typename T>
int launch_column_sum(T* slabs, int rows, int64_t M, T* dst, hipStream_t stream) {
  constexpr int kChunk = 64;
  const unsigned bx = (unsigned)((M + 63) / 64);
  int left = rows;
  int64_t stride = M;
  if (rows > kChunk) {
    left = (rows + kChunk - 1) / kChunk;
    hipLaunchKernelGGL(column_sum_kernel<T>, dim3(bx, (unsigned)left), dim3(256), sizeof(T) * 256, stream, slabs, rows, M, kChunk, M, slabs,
                       int64_t(kChunk) * M);
    stride = int64_t(kChunk) * M;
  }
  hipLaunchKernelGGL(column_sum_kernel<T>, dim3(bx, 1), dim3(256), sizeof(T) * 256, stream, slabs, left, stride, left, M, dst, int64_t(0));
  AA_CHECK_HIP(hipGetLastError());
  return AA_OK;
}
template int launch_column_sum<float>(float*, int, int64_t, float*, hipStream_t);
template int launch_column_sum<double>(double*, int, int64_t, double*, hipStream_t);
// uncoupled path weights: gw[p] = sum over channels (in channel order) of the per-channel sums
template <typename T>
__global__ __launch_bounds__(64) void tp_layer_wgrad_channels_kernel(const T* __restrict__ per_channel, int u, int P, T* __restrict__ gw) {
  for (int p = threadIdx.x; p < P; p += 64) {
    T v = T(0);
    for (int ch = 0; ch < u; ++ch) v += per_channel[ch * P + p];
    gw[p] = v;
  }
}

int tp_wgrad_slots(int64_t N, int cap) { return int(std::min<int64_t>(std::max<int64_t>(N, 1), cap)); }
static int wgrad_blocks(int64_t N) { return tp_wgrad_slots(N, 1024); }

// (+ one slab: the per-channel sums of the uncoupled form between the two reduce kernels)
size_t tp_layer_wgrad_workspace_elems(const TpLayerDev& L, int64_t N) { return size_t(wgrad_blocks(N) + 1) * L.mul * L.num_paths; }

template <typename T>
int launch_tp_wgrad_reduce(const void* partial, int nslots, int u, int P, int coupling, void* gw, hipStream_t stream) {
  const int nout = u * P;
  const T* part = static_cast<const T*>(partial);
  T* per_channel = coupling ? static_cast<T*>(gw) : const_cast<T*>(part) + size_t(nslots) * nout;  // the extra slab
  if (int rc = launch_column_sum<T>(const_cast<T*>(part), nslots, nout, per_channel, stream)) return rc;
  if (!coupling) {
    hipLaunchKernelGGL(tp_layer_wgrad_channels_kernel<T>, dim3(1), dim3(64), 0, stream, per_channel, u, P, static_cast<T*>(gw));
    AA_CHECK_HIP(hipGetLastError());
  }
  return AA_OK;
}
template int launch_tp_wgrad_reduce<float>(const void*, int, int, int, int, void*, hipStream_t);
template int launch_tp_wgrad_reduce<double>(const void*, int, int, int, int, void*, hipStream_t);

template <typename T>
int launch_tp_layer_wgrad(const TpLayerDev& L, const TpLayerWgradArgs& a, hipStream_t stream) {
  const int nb = wgrad_blocks(a.N);
  if (a.N == 0 || a.E == 0) {
    AA_CHECK_HIP(hipMemsetAsync(a.gw, 0, sizeof(T) * size_t(L.coupling ? L.mul : 1) * L.num_paths, stream));
    return AA_OK;
  }
  const int apb = int((a.N + nb - 1) / nb);
  size_t smem = sizeof(T) * (size_t(L.mul) * odd_pad(L.d2) +
                             size_t(kWgThreads) * (odd_pad(L.d1) + odd_pad(L.dout) + odd_pad(L.num_paths)));
  AA_REQUIRE(smem <= 160 * 1024, "tp wgrad: LDS budget exceeded");
  AA_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&tp_layer_wgrad_kernel<T>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
  hipLaunchKernelGGL(tp_layer_wgrad_kernel<T>, dim3((unsigned)nb), dim3(kWgThreads), smem, stream, L, a, apb);
  AA_CHECK_HIP(hipGetLastError());
  return launch_tp_wgrad_reduce<T>(a.partial, nb, L.mul, L.num_paths, L.coupling, a.gw, stream);
}

// ---------------------------------------------------------------------------------------------
// host side: tables
// ---------------------------------------------------------------------------------------------
static int upload_table(const std::vector<std::tuple<int, int, int, int, double>>& ents_in /*out,path,a,b,val*/,
                        int num_out, TpTable* tab, std::vector<void*>* owned) {
  auto ents = ents_in;
  std::sort(ents.begin(), ents.end());
  std::vector<TpGroup> groups;
  std::vector<TpEntry> entries;
  size_t pos = 0;
  for (int c = 0; c < num_out; ++c) {
    bool any = false;
    while (pos < ents.size() && std::get<0>(ents[pos]) == c) {
      int p = std::get<1>(ents[pos]);
      TpGroup g{c, p, (int)entries.size(), 0};
      while (pos < ents.size() && std::get<0>(ents[pos]) == c && std::get<1>(ents[pos]) == p) {
        TpEntry en;
        en.a = std::get<2>(ents[pos]);
        en.b = std::get<3>(ents[pos]);
        en.val_d = std::get<4>(ents[pos]);
        en.val_f = (float)en.val_d;
        entries.push_back(en);
        ++pos;
      }
      g.end = (int)entries.size();
      groups.push_back(g);
      any = true;
    }
    if (!any) groups.push_back(TpGroup{c, 0, (int)entries.size(), (int)entries.size()});  // zero output
  }
  void *dg = nullptr, *de = nullptr;
  AA_CHECK_HIP(hipMalloc(&dg, sizeof(TpGroup) * std::max<size_t>(groups.size(), 1)));
  owned->push_back(dg);
  AA_CHECK_HIP(hipMalloc(&de, sizeof(TpEntry) * std::max<size_t>(entries.size(), 1)));
  owned->push_back(de);
  AA_CHECK_HIP(hipMemcpy(dg, groups.data(), sizeof(TpGroup) * groups.size(), hipMemcpyHostToDevice));
  AA_CHECK_HIP(hipMemcpy(de, entries.data(), sizeof(TpEntry) * entries.size(), hipMemcpyHostToDevice));
  tab->num_groups = (int)groups.size();
  tab->groups = static_cast<const TpGroup*>(dg);
  tab->entries = static_cast<const TpEntry*>(de);
  return AA_OK;
}

int build_tp_layer(const aa_tp_desc& d, TpLayerDev* out, std::vector<void*>* owned) {
  AA_REQUIRE(d.mul > 0 && d.d1 > 0 && d.d2 > 0 && d.dout > 0 && d.num_paths > 0 && d.nnz > 0, "tp desc: bad dims");
  std::vector<std::tuple<int, int, int, int, double>> f, b1, b2;
  for (int n = 0; n < d.nnz; ++n) {
    int i = d.nz_i[n], j = d.nz_j[n], k = d.nz_k[n], p = d.nz_path[n];
    AA_REQUIRE(i >= 0 && i < d.d1 && j >= 0 && j < d.d2 && k >= 0 && k < d.dout && p >= 0 && p < d.num_paths,
               "tp desc: index out of range");
    double v = d.nz_val[n];
    f.emplace_back(k, p, i, j, v);
    b1.emplace_back(i, p, k, j, v);
    b2.emplace_back(j, p, k, i, v);
  }
  out->mul = d.mul;
  out->d1 = d.d1;
  out->d2 = d.d2;
  out->dout = d.dout;
  out->num_paths = d.num_paths;
  out->coupling = d.coupling;
  if (int rc = upload_table(f, d.dout, &out->fwd, owned)) return rc;
  if (int rc = upload_table(b1, d.d1, &out->bx1, owned)) return rc;
  if (int rc = upload_table(b2, d.d2, &out->bx2, owned)) return rc;
  return AA_OK;
}

template <typename T>
int launch_tp_layer_fwd(const TpLayerDev& L, const TpLayerFwdArgs& a, hipStream_t stream) {
  if (a.N == 0) return AA_OK;
  size_t smem = sizeof(T) * (size_t(L.mul) * odd_pad(L.d2) + 256 * size_t(odd_pad(L.d1)));
  AA_REQUIRE(smem <= 160 * 1024, "tp fwd: LDS budget exceeded (mul*d2 too large)");
  AA_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&tp_layer_fwd_kernel<T>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
  hipLaunchKernelGGL(tp_layer_fwd_kernel<T>, dim3((unsigned)a.N), dim3(256), smem, stream, L, a);
  AA_CHECK_HIP(hipGetLastError());
  return AA_OK;
}

template <typename T>
int launch_tp_layer_bwd(const TpLayerDev& L, const TpLayerBwdArgs& a, hipStream_t stream) {
  if (a.N == 0) return AA_OK;
  int EC = (256 + L.mul - 1) / L.mul + 1;
  int Dsh = std::max(L.d1, L.d2);
  size_t smem = sizeof(T) * (2 * size_t(L.mul) * odd_pad(L.d2) + 256 * size_t(odd_pad(L.d1)) +
                             256 * size_t(odd_pad(L.dout)) + size_t(EC) * Dsh);
  AA_REQUIRE(smem <= 160 * 1024, "tp bwd: LDS budget exceeded");
  AA_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&tp_layer_bwd_kernel<T>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
  hipLaunchKernelGGL(tp_layer_bwd_kernel<T>, dim3((unsigned)a.N), dim3(256), smem, stream, L, a, EC);
  AA_CHECK_HIP(hipGetLastError());
  return AA_OK;
}

// ---------------------------------------------------------------------------------------------
// scale + segment sum alone (_contract.py:195-204): out[n] = scale * sum_{s in segment n} x[eid(s)], rows of `row`
// contiguous elements.  One workgroup per segment, thread = element, edges in CSR order (deterministic, no atomics);
// four rows in flight per thread.
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void segment_sum_kernel(const T* __restrict__ x, const int32_t* __restrict__ rowptr,
                                                          const int32_t* __restrict__ eids, int64_t row, T scale, T* __restrict__ out) {
  const int64_t n = blockIdx.x;
  const int beg = rowptr[n], end = rowptr[n + 1];
  for (int64_t q = threadIdx.x; q < row; q += 256) {
    T acc = T(0);
    int s = beg;
    for (; s + 4 <= end; s += 4) {
      T v[4];
#pragma unroll
      for (int m = 0; m < 4; ++m) v[m] = x[int64_t(eids ? eids[s + m] : s + m) * row + q];
#pragma unroll
      for (int m = 0; m < 4; ++m) acc += v[m];
    }
    for (; s < end; ++s) acc += x[int64_t(eids ? eids[s] : s) * row + q];
    out[n * row + q] = acc * scale;
  }
}

template <typename T>
int launch_segment_sum(int64_t N, int64_t row, const void* x, const int32_t* rowptr, const int32_t* eids, double scale, void* out,
                       hipStream_t stream) {
  if (N == 0 || row == 0) return AA_OK;
  hipLaunchKernelGGL(segment_sum_kernel<T>, dim3((unsigned)N), dim3(256), 0, stream, static_cast<const T*>(x), rowptr, eids, row,
                     T(scale), static_cast<T*>(out));
  AA_CHECK_HIP(hipGetLastError());
  return AA_OK;
}
template int launch_segment_sum<float>(int64_t, int64_t, const void*, const int32_t*, const int32_t*, double, void*, hipStream_t);
template int launch_segment_sum<double>(int64_t, int64_t, const void*, const int32_t*, const int32_t*, double, void*, hipStream_t);

template int launch_tp_layer_fwd<float>(const TpLayerDev&, const TpLayerFwdArgs&, hipStream_t);
template int launch_tp_layer_fwd<double>(const TpLayerDev&, const TpLayerFwdArgs&, hipStream_t);
template int launch_tp_layer_bwd<float>(const TpLayerDev&, const TpLayerBwdArgs&, hipStream_t);
template int launch_tp_layer_bwd<double>(const TpLayerDev&, const TpLayerBwdArgs&, hipStream_t);
template int launch_tp_layer_wgrad<float>(const TpLayerDev&, const TpLayerWgradArgs&, hipStream_t);
template int launch_tp_layer_wgrad<double>(const TpLayerDev&, const TpLayerWgradArgs&, hipStream_t);

}  // namespace aa
