// Building blocks of the fused per-atom-tile kernels (aa_fused.hip: forward; aa_fused_bwd.hip: reverse tail): the
// workgroup-shared weight pipeline (12-KB steps, L2 -> registers -> LDS, double buffered, one raw barrier per step),
// one linear layer on a wave's 32-edge tile in the accumulator layout, tiles parked in LDS, per-atom moments through the
// wave-private transposing patch.  `Args` is the kernel's argument block: anything with `const void* wstep[][2]`.
#pragma once
#include <type_traits>

#include "aa_cg_gen.h"
#include "aa_wave.h"
#include "aa_common.h"
#include "aa_geom.h"
#include "aa_mfma.h"

namespace aa {
namespace {

constexpr int kLdA = 68;   // row stride (floats) of the [32 edges][64 features] patch: 16-B aligned rows, conflict-light
constexpr int kLdT = kTileLdT;  // row stride of the [32][32] store-transpose patch (as in the chain kernel)
constexpr int kLdY = 16;   // row stride of sY [32 edges][D <= 16] and sM [64 k][D]
constexpr int kOffB = 32 * kLdT;          // sB [D][64] sits behind the store patch inside the wave region
constexpr int kWaveRegion = 32 * kLdA;    // floats: max(sA, sT + sB, sM + sB)
static_assert(kOffB + 16 * 64 <= kWaveRegion, "per-atom vectors must fit behind the store patch");

// ---- weight pipeline --------------------------------------------------------------------------------------------
// The kernel's program is a fixed sequence of NS "steps"; step S consumes one 12-KB block of weights (a tile pair x
// 32-deep chunk of a linear layer in bf16x3 fragments, or 16 rows of an env-weight matrix) from LDS buffer S & 1.
// Block S + 2 is requested from L2 at the START of step S into one of two register sets and lands in LDS at the END
// of step S + 1, i.e. every load has two full steps to arrive (the kernel runs one wave per SIMD: nothing else hides
// L2 latency).  Barriers are raw s_barrier with LDS-only fences, so that they do not drain the loads in flight
// (__syncthreads() carries a vmcnt(0)).  All indices are compile-time: the whole program is unrolled.
// element of `base` at a 32-bit byte offset (see pipe_load)
template <class T>
__device__ __forceinline__ T* at_bytes(T* base, unsigned byte_off) {
  using B = std::conditional_t<std::is_const_v<T>, const char, char>;
  return reinterpret_cast<T*>(reinterpret_cast<B*>(base) + byte_off);  // (a byte GEP with a zero-extended 32-bit index: the saddr + voffset pattern)
}

struct FusedPipe {
  u32x4 ra[3], rb[3];
  u32x4* wbuf;  // [2][kWStep]
  int tid, lane;
  bool stager;  // this wave takes part in staging the weight blocks (the first four waves of the workgroup: 256 x 48 B = 12 KB)
  int zero;     // 0, opaque to the optimizer and refreshed once per trip of the persistent loop (see pipe_load)
};


// (every load is wave-uniform base + lane offset: the bases stay in SGPRs and all loads of the program share ONE
//  offset register -- with per-lane base selects the compiler hoists ~140 loop-invariant 64-bit address pairs out of
//  the persistent loop and spills them)
//  What remains hoisted -- the block addresses of the ~46 steps -- costs two or three scratch reloads per step.)
//  `zero` (an SGPR the optimizer cannot see through, set anew in every trip of the persistent loop) is added to the step
//  index so that the two block addresses of a step are scalar loads from the argument block INSIDE the loop -- with a
//  constant index they are loop-invariant, all ~90 address pairs are hoisted, and what does not fit the scalar file is
//  spilled and reloaded from scratch two or three times per step.
template <class Args>
__device__ __forceinline__ void pipe_load(const Args& A, int tid, int t, u32x4* r, int zero = 0) {
  const u32x4* s0 = static_cast<const u32x4*>(A.wstep[t + zero][0]);
  const u32x4* s1 = static_cast<const u32x4*>(A.wstep[t + zero][1]);
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const u32x4* mid = wv < 2 ? s0 + 256 : s1 - 128;  // elements 256..383 of the first half | 0..127 of the second
  // (a 32-bit BYTE offset: base + zero-extended offset is ONE saddr + voffset load; an element index -- signed or not -- is scaled in
  //  64 bits and costs a v_lshl_add_u64 per load)
  unsigned ob = unsigned(tid) * 16u;
  opaque_vector(ob);  // (keeps the zero-extension next to the load: hoisted out of the loop as a 64-bit pair it defeats the saddr pattern)
  r[0] = *at_bytes(s0, ob);
  r[1] = *at_bytes(mid, ob);
  r[2] = *at_bytes(s1 + 128, ob);
}
__device__ __forceinline__ void pipe_store(u32x4* wbuf, int b, int tid, const u32x4* r) {
  u32x4* d = wbuf + b * kWStep;
  d[tid] = r[0];
  d[256 + tid] = r[1];
  d[512 + tid] = r[2];
}
template <int S, int NS, class Args>
__device__ __forceinline__ void pipe_issue(const Args& A, FusedPipe& p) {
  if (!p.stager) return;
  if constexpr ((S & 1) == 0)
    pipe_load(A, p.tid, (S + 2) % NS, p.ra, p.zero);
  else
    pipe_load(A, p.tid, (S + 2) % NS, p.rb, p.zero);
}
template <int S>
__device__ __forceinline__ void pipe_commit(FusedPipe& p) {
  if (p.stager) {
    if constexpr (((S + 1) & 1) == 0)
      pipe_store(p.wbuf, 0, p.tid, p.ra);
    else
      pipe_store(p.wbuf, 1, p.tid, p.rb);
  }
  lds_barrier();
  // one scheduling region per step: without it the fully unrolled program is treated as one region and later steps'
  // operand splits / LDS reads are hoisted far ahead (hundreds of spilled registers)
  __builtin_amdgcn_sched_barrier(0);
}

// The same pipeline staged by EIGHT waves (512 threads x 24 B = 12 KB; aa_fused8.hip): thread t moves element t, threads 0..255
// also element 512 + t.  Wave-uniform bases as above: elements 0..383 are the first 6-KB half, 384..767 the second.
struct FusedPipe8 {
  u32x4 ra[2], rb[2];
  u32x4* wbuf;  // [2][kWStep]
  int tid, lane;
  int wv;    // wave of the workgroup (scalar)
  int zero;  // (see FusedPipe)
};
template <class Args>
__device__ __forceinline__ void pipe_load8(const Args& A, const FusedPipe8& p, int t, u32x4* r) {
  const u32x4* s0 = static_cast<const u32x4*>(A.wstep[t + p.zero][0]);
  const u32x4* s1 = static_cast<const u32x4*>(A.wstep[t + p.zero][1]);
  const u32x4* b0 = p.wv < 6 ? s0 : s1 - 384;
  unsigned ob = unsigned(p.tid) * 16u;
  opaque_vector(ob);  // (keeps the zero-extension next to the load: hoisted out of the loop as a 64-bit pair it defeats the saddr pattern)
  r[0] = *at_bytes(b0, ob);
  if (p.wv < 4) r[1] = *at_bytes(s1 + 128, ob);
}
__device__ __forceinline__ void pipe_store8(const FusedPipe8& p, int b, const u32x4* r) {
  u32x4* d = p.wbuf + b * kWStep;
  d[p.tid] = r[0];
  if (p.wv < 4) d[512 + p.tid] = r[1];
}
template <int S, int NS, class Args>
__device__ __forceinline__ void pipe_issue(const Args& A, FusedPipe8& p) {
  if constexpr ((S & 1) == 0)
    pipe_load8(A, p, (S + 2) % NS, p.ra);
  else
    pipe_load8(A, p, (S + 2) % NS, p.rb);
}
template <int S>
__device__ __forceinline__ void pipe_commit(FusedPipe8& p) {
  if constexpr (((S + 1) & 1) == 0)
    pipe_store8(p, 0, p.ra);
  else
    pipe_store8(p, 1, p.rb);
  lds_barrier();
  __builtin_amdgcn_sched_barrier(0);  // (one scheduling region per step: see pipe_commit above)
}

// 24 MFMAs of one step (6 cross products x 2 k halves x 2 tiles); weight levels read from LDS just in time
__device__ __forceinline__ void fused_mma_step(const u32x4* wb, int lane, const XSplit& x, v16f& acc0, v16f& acc1) {
  const u32x4* w = wb + lane;
#define AA_W(T_, Q_) w[((T_)*6 + (Q_)) * 64]
  {
    const u32x4 a0 = AA_W(0, 4), b0 = AA_W(1, 4), a1 = AA_W(0, 5), b1 = AA_W(1, 5);  // level 3
    acc0 = mma_bf16(a0, x.l1[0], acc0);
    acc1 = mma_bf16(b0, x.l1[0], acc1);
    acc0 = mma_bf16(a1, x.l1[1], acc0);
    acc1 = mma_bf16(b1, x.l1[1], acc1);
  }
  {
    const u32x4 a0 = AA_W(0, 2), b0 = AA_W(1, 2), a1 = AA_W(0, 3), b1 = AA_W(1, 3);  // level 2
    acc0 = mma_bf16(a0, x.l2[0], acc0);
    acc1 = mma_bf16(b0, x.l2[0], acc1);
    acc0 = mma_bf16(a1, x.l2[1], acc0);
    acc1 = mma_bf16(b1, x.l2[1], acc1);
    acc0 = mma_bf16(a0, x.l1[0], acc0);
    acc1 = mma_bf16(b0, x.l1[0], acc1);
    acc0 = mma_bf16(a1, x.l1[1], acc0);
    acc1 = mma_bf16(b1, x.l1[1], acc1);
  }
  {
    const u32x4 a0 = AA_W(0, 0), b0 = AA_W(1, 0), a1 = AA_W(0, 1), b1 = AA_W(1, 1);  // level 1
    acc0 = mma_bf16(a0, x.l3[0], acc0);
    acc1 = mma_bf16(b0, x.l3[0], acc1);
    acc0 = mma_bf16(a1, x.l3[1], acc0);
    acc1 = mma_bf16(b1, x.l3[1], acc1);
    acc0 = mma_bf16(a0, x.l2[0], acc0);
    acc1 = mma_bf16(b0, x.l2[0], acc1);
    acc0 = mma_bf16(a1, x.l2[1], acc0);
    acc1 = mma_bf16(b1, x.l2[1], acc1);
    acc0 = mma_bf16(a0, x.l1[0], acc0);
    acc1 = mma_bf16(b0, x.l1[0], acc1);
    acc0 = mma_bf16(a1, x.l1[1], acc0);
    acc1 = mma_bf16(b1, x.l1[1], acc1);
  }
#undef AA_W
}

// One linear layer on the wave's tile, steps S0 .. S0 + KC * NT / 2 - 1 of the program: KC 32-deep operand chunks
// (op(kc) -> the v16f tile that is chunk kc), NT output tiles in pairs (epi(pair, acc0, acc1) after each pair).
// Operand splits are software-pipelined: chunk kc + 1 is split while the MFMAs of chunk kc execute (layers with few
// chunks and several pairs split all chunks once up front).
// an operand chunk is either a tile in accumulator layout (split here) or an XSplit that was split when it was produced
// (tiles that feed several layers -- the two-body scalars, lat0 -- are split ONCE and held)
__device__ __forceinline__ void to_xsplit(const v16f& t, XSplit& x) { xsplit_from_acc(t, x); }
__device__ __forceinline__ void to_xsplit(const XSplit& t, XSplit& x) { x = t; }

template <int S0, int NS, int KC, int NT, class Args, class Pipe, class OpF, class EpiF>
__device__ __forceinline__ void fused_layer(const Args& A, Pipe& p, OpF&& op, EpiF&& epi) {
  static_assert(NT % 2 == 0, "output tiles come in pairs");
  constexpr bool PRE = NT > 2;
  constexpr bool PIPE = true;
  XSplit xs[PRE ? KC : 2];
  if constexpr (PRE) {
    static_for<0, KC>([&](auto kc) { to_xsplit(op(kc), xs[kc]); });
  } else if constexpr (PIPE) {
    to_xsplit(op(std::integral_constant<int, 0>{}), xs[0]);
  }
  static_for<0, NT / 2>([&](auto ntp) {
    v16f acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      acc0[r] = 0.f;
      acc1[r] = 0.f;
    }
    static_for<0, KC>([&](auto kcc) {
      constexpr int kc = decltype(kcc)::value;
      constexpr int S = S0 + decltype(ntp)::value * KC + kc;
      pipe_issue<S, NS>(A, p);
      if constexpr (!PRE && !PIPE) to_xsplit(op(kcc), xs[kc & 1]);
      fused_mma_step(p.wbuf + (S & 1) * kWStep, p.lane, xs[PRE ? kc : (kc & 1)], acc0, acc1);
      if constexpr (!PRE && PIPE && kc + 1 < KC) {
        to_xsplit(op(std::integral_constant<int, kc + 1>{}), xs[(kc + 1) & 1]);
        // (measured, round 4: asking the scheduler to interleave this split with the step's MFMAs -- sched_group_barrier
        //  "1 MFMA, 4 VALU" x 24 -- does produce that pattern in the ISA and is 1 % SLOWER on MI355X; left to the compiler)
      }
      pipe_commit<S>(p);
    });
    epi(ntp, acc0, acc1);
  });
}

// A tile pair parked in LDS in accumulator layout ([q][lane] 16-B cells: conflict-free b128 accesses).  The two-body
// scalars and lat0 are operands of three / two later layers; parking them frees 64 registers per lane for the whole
// second half of the kernel (the kernel runs one wave per SIMD, LDS is plentiful).
constexpr int kFusedOcc = 1;  // one workgroup per CU: the kernel needs the whole register file and most of the LDS
__device__ __forceinline__ void park_tile(float* slot, const v16f& t, int lane) {
#pragma unroll
  for (int q = 0; q < 4; ++q) *reinterpret_cast<v4f*>(slot + (q * 64 + lane) * 4) = v4f{t[4 * q], t[4 * q + 1], t[4 * q + 2], t[4 * q + 3]};
}
__device__ __forceinline__ v16f fetch_tile(const float* slot, int lane) {
  v16f t;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const v4f v = *reinterpret_cast<const v4f*>(slot + (q * 64 + lane) * 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) t[4 * q + i] = v[i];
  }
  return t;
}
constexpr int kTileFloats = 64 * 16;  // one parked 32-feature tile

template <bool ACT>
__device__ __forceinline__ void keep_tile(const v16f& acc, v16f& k) {
#pragma unroll
  for (int r = 0; r < 16; ++r) k[r] = ACT ? silu(acc[r]) : acc[r];
}

// M[j] (lane = k) = sum over the tile's rows of Y[e][j] * a[e][k]: the two tiles go to the LDS patch in [e][k] order,
// every lane then walks its column.  Rows beyond the segment carry Y = 0 (sY), so they drop out.
template <int D>
__device__ __forceinline__ void tile_moments(float* sA, const float* sY, const v16f& t0, const v16f& t1, int lane, float* M) {
  const int el = lane & 31, hh = lane >> 5;
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    *reinterpret_cast<v4f*>(sA + el * kLdA + 8 * q + 4 * hh) = v4f{t0[4 * q], t0[4 * q + 1], t0[4 * q + 2], t0[4 * q + 3]};
    *reinterpret_cast<v4f*>(sA + el * kLdA + 32 + 8 * q + 4 * hh) = v4f{t1[4 * q], t1[4 * q + 1], t1[4 * q + 2], t1[4 * q + 3]};
  }
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int j = 0; j < D; ++j) M[j] = 0.f;
#pragma unroll 4
  for (int e = 0; e < 32; ++e) {
    const float a = sA[e * kLdA + lane];
    float y[16];
#pragma unroll
    for (int q = 0; q < (D + 3) / 4; ++q) {
      const v4f yy = *reinterpret_cast<const v4f*>(sY + e * kLdY + 4 * q);
#pragma unroll
      for (int i = 0; i < 4; ++i) y[4 * q + i] = yy[i];
    }
#pragma unroll
    for (int j = 0; j < D; ++j) M[j] += y[j] * a;
  }
  __builtin_amdgcn_wave_barrier();
}

// x2s[j] (lane = channel) = f * sum_k M[j][k] * Wk[k][r(j)][ch].  M is handed over through sM [k][D]; the env-weight
// matrix Wk [64][R][64] arrives through the weight pipeline as 4 blocks of 16 rows (steps S0 .. S0 + 3).
template <int S0, int NS, int D, int R, int LDY = kLdY, class Args, class Pipe>
__device__ __forceinline__ void project_moments(const Args& A, Pipe& p, float* sM, const float* M, float sf, float* x2s) {
  const int lane = p.lane;
#pragma unroll
  for (int q = 0; q < (D + 3) / 4; ++q) {
    v4f mm;
#pragma unroll
    for (int i = 0; i < 4; ++i) mm[i] = 4 * q + i < D ? M[4 * q + i] : 0.f;
    *reinterpret_cast<v4f*>(sM + lane * LDY + 4 * q) = mm;
  }
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int j = 0; j < D; ++j) x2s[j] = 0.f;
  static_for<0, 4>([&](auto cc) {
    constexpr int c = decltype(cc)::value;
    constexpr int S = S0 + c;
    pipe_issue<S, NS>(A, p);
    const float* wf = reinterpret_cast<const float*>(p.wbuf + (S & 1) * kWStep) + lane;
    // rows are fetched one row ahead of the nine FMAs that consume them (two operand sets): fetched and consumed in the same
    // breath, every row waited out a full LDS latency -- 128 exposed waits per tile, ~15 % of the kernel (ISA: `ds_read x5;
    // s_waitcnt lgkmcnt(0); 9 x v_fma` per row)
    constexpr int Q = (D + 3) / 4;
    auto fetch = [&](int kk, float* w, v4f* m) {
#pragma unroll
      for (int r = 0; r < R; ++r) w[r] = wf[(kk * R + r) * 64];
#pragma unroll
      for (int q = 0; q < Q; ++q) m[q] = *reinterpret_cast<const v4f*>(sM + (16 * c + kk) * LDY + 4 * q);
    };
    auto consume = [&](const float* w, const v4f* m) {
#pragma unroll
      for (int j = 0; j < D; ++j) x2s[j] += m[j >> 2][j & 3] * w[r_of<0>(j)];
    };
    float wa[R], wb[R];
    v4f ma[Q], mb[Q];
    fetch(0, wa, ma);
#pragma unroll
    for (int kk = 0; kk < 16; kk += 2) {
      fetch(kk + 1, wb, mb);
      consume(wa, ma);
      if (kk + 2 < 16) fetch(kk + 2, wa, ma);
      consume(wb, mb);
      // (at 256 registers the rows of a whole block are still gathered at the front and spilled -- sched_barrier binds the machine
      //  scheduler only; an anchor is a memory barrier to every pass)
      if constexpr (std::is_same_v<Pipe, FusedPipe8>) {
#pragma unroll
        for (int j = 0; j < D; ++j) anchor(x2s[j]);  // (every accumulation chain: an unpinned one is sunk below the block with its operands live)
      }
      __builtin_amdgcn_sched_barrier(0);  // at most two rows' LDS operands in flight beyond the ones being consumed
    }
    pipe_commit<S>(p);
  });
#pragma unroll
  for (int j = 0; j < D; ++j) x2s[j] *= sf;
}

// The same projection on the matrix cores (kProjMfma).  The moments go through sM [k][kLdY] into an operand tile pair in
// accumulator layout whose 32 columns are the components j (columns >= D carry zeros) and whose features are k; irrep r's 64x64
// env-weight matrix is an ordinary bf16x3 layer (steps S0 + 2 r, S0 + 2 r + 1) whose output columns j in irrep r are kept; the
// result returns to the lane = channel view through sX [D][64] behind sM.
template <int S0, int NS, int D, int R, int LDY = kLdY, class Args, class Pipe>
__device__ __forceinline__ void project_moments_mfma(const Args& A, Pipe& p, float* sM, const float* M, float sf, float* x2s, float* sX = nullptr) {
  const int lane = p.lane, el = lane & 31, hh = lane >> 5;
  if (sX == nullptr) sX = sM + 64 * LDY;  // (result patch [D][64]: behind sM, or where the caller has room)
  static_assert(LDY != kLdY || 64 * kLdY + 16 * 64 <= kWaveRegion, "moments + result patch must fit the wave region");
#pragma unroll
  for (int q = 0; q < (D + 3) / 4; ++q) {
    v4f mm;
#pragma unroll
    for (int i = 0; i < 4; ++i) mm[i] = 4 * q + i < D ? M[4 * q + i] : 0.f;
    *reinterpret_cast<v4f*>(sM + lane * LDY + 4 * q) = mm;
  }
#pragma unroll
  for (int q = (D + 3) / 4; q < LDY / 4; ++q) *reinterpret_cast<v4f*>(sM + lane * LDY + 4 * q) = v4f{0.f, 0.f, 0.f, 0.f};
  __builtin_amdgcn_wave_barrier();
  XSplit xs[2];
  {
    v16f t0, t1;
    const int col = el < LDY ? el : LDY - 1;  // (columns beyond the patch: any value, never read back)
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const int k = 8 * (s >> 2) + 4 * hh + (s & 3);
      t0[s] = sM[k * LDY + col];
      t1[s] = sM[(32 + k) * LDY + col];
    }
    xsplit_from_acc(t0, xs[0]);
    xsplit_from_acc(t1, xs[1]);
  }
  static_for<0, R>([&](auto rr) {
    constexpr int r = decltype(rr)::value;
    v16f acc0, acc1;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      acc0[q] = 0.f;
      acc1[q] = 0.f;
    }
    static_for<0, 2>([&](auto kcc) {
      constexpr int kc = decltype(kcc)::value;
      constexpr int S = S0 + 2 * r + kc;
      pipe_issue<S, NS>(A, p);
      fused_mma_step(p.wbuf + (S & 1) * kWStep, p.lane, xs[kc], acc0, acc1);
      pipe_commit<S>(p);
    });
    if (el >= r * r && el < (r + 1) * (r + 1)) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        *reinterpret_cast<v4f*>(sX + el * 64 + 8 * q + 4 * hh) = v4f{acc0[4 * q], acc0[4 * q + 1], acc0[4 * q + 2], acc0[4 * q + 3]};
        *reinterpret_cast<v4f*>(sX + el * 64 + 32 + 8 * q + 4 * hh) = v4f{acc1[4 * q], acc1[4 * q + 1], acc1[4 * q + 2], acc1[4 * q + 3]};
      }
    }
  });
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int j = 0; j < D; ++j) x2s[j] = sX[j * 64 + lane] * sf;
  __builtin_amdgcn_wave_barrier();
}

}  // namespace
}  // namespace aa
