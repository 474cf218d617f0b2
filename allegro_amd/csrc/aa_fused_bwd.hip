// Fused per-atom-tile REVERSE TAIL of the standard 2-layer Allegro stack (gfx950, fp32 via bf16x3 MFMA): everything the
// reverse pass does after the layer-0 scalar gradients exist, in one launch, one wave per center atom's edge tile.
//
// The reference reaches the forces by autograd through allegro/nn/_allegro.py:237-301 and
// allegro/nn/_strided/_contract.py:185-251; the staged reverse pass of this library (aa_model.hip: Runner::backward) ends
// with three launches that communicate through HBM:
//
//   tp_mom_bwd_first   layer-0 tensor product reverse: g_w0 [E,R*64], g_aenv [E,64], dE/dY slots        (2.7 KB/edge)
//   gemm chain "B1"    [g_two_body | g_w0] @ G0^T + g_aenv -> scalar_embed_mlp reverse -> 8 basis sums   (1.5 KB/edge)
//   edge_backward      dE/dY + basis sums -> dE/dr_e                                                     (0.2 KB/edge)
//
// Here one wave owns one center atom's <= 32 edges (as in the fused forward, aa_fused.hip) and keeps all of that on chip:
// 2.1 KB/edge are read (w0, the two scalar-gradient rows, EDGE_EMBEDDING, the two-body gradient, one pre-activation,
// geometry) and 16 B/edge written (dE/dr_e).
//
// Layouts.  Every per-edge operand is loaded ONCE, straight into the MFMA accumulator layout of the swapped-operand GEMM
// (lane = (edge el = lane & 31, half hh = lane >> 5); a 32-feature tile = 16 registers, register 4q + i holds feature
// 8q + 4hh + i) -- which is also the B operand of the linear layers.  In that layout a sum over the 64 channels of an edge
// is 32 lane-local terms + ONE half exchange (v_permlane32_swap), so the per-edge channel sums (dE/dY, the 8 basis sums)
// need no 64-lane butterflies and no LDS.  Only the per-ATOM sums over edges (Q_l[a][ch] = sum_e g_l[e,ch] x1[e,a,ch],
// tp_mom_bwd_first_kernel in aa_tp_spec.hip) change view: the products go through the wave-private transposing patch
// and every lane (= channel) walks the 32 rows, exactly as the forward's moments (tile_moments).  The Clebsch-Gordan
// contractions then run once per atom in the lane = channel view (Sig0::bx2), the env-weight projection of their result
// (GM = d x2s0 . Wenv0^T) takes its weights from the workgroup's weight pipeline like the forward's project_moments, and
// the layer-0 x1-weight gradient g_w0 is never formed as an array: each 32-channel chunk of it is built in registers as
// the B operand of the first linear layer,  g_w0[e][r][ch] = sum_{a in r} Y[e][a] (g1[e][ch] B1[a][ch] + g0[e][ch] B0[a][ch]).
//
// Eligibility = the fused forward's (2 layers, 64 channels / features / hidden widths, fp32, l_max <= 2, every center atom
// <= 32 edges) with at most 2 species (two-body table of the reverse: aa_model_plan.embed_fused).
#include "aa_fused_tile.h"

namespace aa {

namespace {

#ifndef AA_TAIL_WAVES
#define AA_TAIL_WAVES 8
#endif
constexpr int kTailWaves = AA_TAIL_WAVES;  // waves (= atoms) per workgroup: 8 = two per SIMD share one 24-KB weight double buffer
constexpr int kTailD = 9;      // l_max <= 2
// wave-private LDS region (floats): sY [32][kLdY] | patch [32][kLdA] (later: sG [64][kLdY] at 0, sGM [D][64] behind it) | sB0 [D][64] | sB1 [D][64]
constexpr int kTailOffP = 32 * kLdY, kTailOffGM = kTailOffP + 64 * kLdY, kTailOffB0 = kTailOffP + 32 * kLdA,
              kTailOffB1 = kTailOffB0 + kTailD * 64, kTailWaveFloats = kTailOffB1 + kTailD * 64;
static_assert(64 * kLdY + kTailD * 64 <= 32 * kLdA, "sG and sGM live inside the patch region");

// rows [beg, beg + 32) x 32 features of a row-major array -> one tile in accumulator layout (zero beyond the segment).
// `tile0` = the array's column block at row `beg` (wave-uniform: it stays in scalar registers), `off` = the lane's
// el * ld + 4 * hh (one 32-bit register per leading dimension, shared by every array with that row stride)
__device__ __forceinline__ v16f ld_tile(const float* tile0, int off, bool ok) {
  v16f t;
#pragma unroll
  for (int r = 0; r < 16; ++r) t[r] = 0.f;
  if (ok) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const v4f v = *reinterpret_cast<const v4f*>(tile0 + off + 8 * q);
#pragma unroll
      for (int i = 0; i < 4; ++i) t[4 * q + i] = v[i];
    }
  }
  return t;
}

// dE/dY of the x1 path, irrep RR:  gy[a] += sum_ch (t1[e][ch] B1[a][ch] + t0[e][ch] B0[a][ch]),  t_l = g_l * w0_RR
// (tile pairs: *a = channels 0..31, *b = 32..63; b1 / b0: the lane's view of sB1 / sB0, offset by 4 * hh)
template <int RR>
__device__ __forceinline__ void tile_gy_x1(const float* b1, const float* b0, const v16f& t1a, const v16f& t1b, const v16f& t0a,
                                           const v16f& t0b, float* gy) {
  constexpr int a0 = RR * RR, na = 2 * RR + 1;
#pragma unroll
  for (int g = 0; g < 8; ++g) {
    const int q = g & 3, off = 32 * (g >> 2) + 8 * q;
#pragma unroll
    for (int a = 0; a < na; ++a) {
      const v4f c1 = *reinterpret_cast<const v4f*>(b1 + (a0 + a) * 64 + off);
      const v4f c0 = *reinterpret_cast<const v4f*>(b0 + (a0 + a) * 64 + off);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float x1 = g < 4 ? t1a[4 * q + i] : t1b[4 * q + i];
        const float x0 = g < 4 ? t0a[4 * q + i] : t0b[4 * q + i];
        gy[a0 + a] += x1 * c1[i] + x0 * c0[i];
      }
    }
    // gy is consumed at the very end of the tile's program: unanchored, the optimizer sinks these sums below the MFMA
    // phases and keeps every product tile alive until then (aa::anchor)
#pragma unroll
    for (int a = 0; a < na; ++a) anchor(gy[a0 + a]);
    __builtin_amdgcn_sched_barrier(0);
  }
}

// one 32-channel chunk (irrep RR, half H) of the layer-0 x1-weight gradient, in accumulator layout = operand layout:
//   g_w0[e][RR][ch] = g1[e][ch] * sum_{a in RR} Y[e][a] B1[a][ch]  +  g0[e][ch] * sum_a Y[e][a] B0[a][ch]
template <int RR, int H>
__device__ __forceinline__ v16f tile_gw0(const float* b1, const float* b0, const float* Y, const v16f& g1x, const v16f& g0x) {
  constexpr int a0 = RR * RR, na = 2 * RR + 1;
  v16f out;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    v4f p1 = {0.f, 0.f, 0.f, 0.f}, p0 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int a = 0; a < na; ++a) {
      const v4f c1 = *reinterpret_cast<const v4f*>(b1 + (a0 + a) * 64 + 32 * H + 8 * q);
      const v4f c0 = *reinterpret_cast<const v4f*>(b0 + (a0 + a) * 64 + 32 * H + 8 * q);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        p1[i] += Y[a0 + a] * c1[i];
        p0[i] += Y[a0 + a] * c0[i];
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) out[4 * q + i] = g1x[4 * q + i] * p1[i] + g0x[4 * q + i] * p0[i];
  }
  return out;
}

// Q[a] (lane = channel) += sum over the tile's rows of Y[e][a] * t[e][ch] for the components a of irrep RR: the tile pair
// goes to the patch in [e][ch] order, every lane walks its column (rows beyond the segment hold zeros)
template <int RR>
__device__ __forceinline__ void tile_q_accumulate(float* sP, const float* sY, const v16f& ta, const v16f& tb, int lane, float* Q) {
  constexpr int a0 = RR * RR, na = 2 * RR + 1;
  const int el = lane & 31, hh = lane >> 5;
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    *reinterpret_cast<v4f*>(sP + el * kLdA + 8 * q + 4 * hh) = v4f{ta[4 * q], ta[4 * q + 1], ta[4 * q + 2], ta[4 * q + 3]};
    *reinterpret_cast<v4f*>(sP + el * kLdA + 32 + 8 * q + 4 * hh) = v4f{tb[4 * q], tb[4 * q + 1], tb[4 * q + 2], tb[4 * q + 3]};
  }
  __builtin_amdgcn_wave_barrier();
#pragma unroll 4
  for (int e = 0; e < 32; ++e) {
    const float t = sP[e * kLdA + lane];
    float y[12];
    // the 16-B cells of sY row e that cover components [a0, a0 + na): cell 0 | cell 0 | cells 1, 2
    constexpr int c_lo = a0 / 4, c_hi = (a0 + na - 1) / 4;
#pragma unroll
    for (int c = c_lo; c <= c_hi; ++c) {
      const v4f yy = *reinterpret_cast<const v4f*>(sY + e * kLdY + 4 * c);
#pragma unroll
      for (int i = 0; i < 4; ++i) y[4 * c + i] = yy[i];
    }
#pragma unroll
    for (int a = 0; a < na; ++a) Q[a0 + a] += y[a0 + a] * t;
  }
  __builtin_amdgcn_wave_barrier();
}

// ---- weight pipeline of this kernel: the forward's (aa_fused_tile.h) with distance 1 and all eight waves staging -- block
// S + 1 is requested from L2 at the START of step S (two / one 16-B loads per lane) and lands in LDS buffer (S + 1) & 1 at
// its END, one raw barrier per step.  8 instead of 24 registers: two waves per SIMD hide the L2 latency that the forward,
// at one wave per SIMD, covers with a distance-2 pipeline.
struct TailPipe {
  u32x4 r0, r1, r2;
  u32x4* wbuf;  // [2][kWStep]
  int tid, lane, wv;
};
template <class Args>
__device__ __forceinline__ void tail_pipe_load(const Args& A, TailPipe& p, int t) {
  const u32x4* s0 = static_cast<const u32x4*>(A.wstep[t][0]);  // elements 0..383 of the 768-element step
  const u32x4* s1 = static_cast<const u32x4*>(A.wstep[t][1]);  // elements 384..767
  if constexpr (kTailWaves == 8) {
    const u32x4* b0 = p.wv < 6 ? s0 : s1 - 384;                // (wave-uniform bases: the lane offset is shared)
    p.r0 = b0[p.tid];
    if (p.wv < 4) p.r1 = (s1 + 128)[p.tid];
  } else {
    const u32x4* mid = p.wv < 2 ? s0 + 256 : s1 - 128;
    p.r0 = s0[p.tid];
    p.r1 = mid[p.tid];
    p.r2 = (s1 + 128)[p.tid];
  }
}
__device__ __forceinline__ void tail_pipe_store(TailPipe& p, int b) {
  u32x4* d = p.wbuf + b * kWStep;
  d[p.tid] = p.r0;
  if constexpr (kTailWaves == 8) {
    if (p.wv < 4) d[512 + p.tid] = p.r1;
  } else {
    d[256 + p.tid] = p.r1;
    d[512 + p.tid] = p.r2;
  }
}
template <int S, int NS, class Args>
__device__ __forceinline__ void tail_issue(const Args& A, TailPipe& p) {
  tail_pipe_load(A, p, (S + 1) % NS);
}
template <int S>
__device__ __forceinline__ void tail_commit(TailPipe& p) {
  tail_pipe_store(p, (S + 1) & 1);
  lds_barrier();
  __builtin_amdgcn_sched_barrier(0);  // one scheduling region per step
}
// one 64-output linear layer on the wave's tile: KC 32-deep operand chunks op(kc), steps S0 .. S0 + KC - 1; epi(acc0, acc1)
// pre(kc) runs at the start of step kc, before the step's operand is built: the place to REQUEST global operands of later
// steps (this kernel is bound by exposed memory latency, not by issue: every operand is requested at least a step ahead)
template <int S0, int NS, int KC, class Args, class PreF, class OpF, class EpiF>
__device__ __forceinline__ void tail_layer(const Args& A, TailPipe& p, PreF&& pre, OpF&& op, EpiF&& epi) {
  v16f acc0, acc1;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    acc0[r] = 0.f;
    acc1[r] = 0.f;
  }
  static_for<0, KC>([&](auto kcc) {
    constexpr int S = S0 + decltype(kcc)::value;
    tail_issue<S, NS>(A, p);
    pre(kcc);
    XSplit xs;
    {
      const v16f t = op(kcc);
      xsplit_from_acc(t, xs);
    }
    fused_mma_step(p.wbuf + (S & 1) * kWStep, p.lane, xs, acc0, acc1);
    tail_commit<S>(p);
  });
  epi(acc0, acc1);
}

// GM[j] (lane = k) = sum_ch g[j][ch] * W[ch][r(j)][k]: the forward's project_moments with the roles of the two indices
// exchanged (g is handed over through sG [ch][kLdY]; the matrix [64 ch][R][64 k] arrives as 4 blocks of 16 channels through
// the weight pipeline, steps S0 .. S0 + 3).  Rolled loops: this kernel runs two waves per SIMD on half the register file.
template <int S0, int NS, int D, int R, class Args>
__device__ __forceinline__ void project_gm(const Args& A, TailPipe& p, float* sG, const float* g, float* gm) {
  const int lane = p.lane;
#pragma unroll
  for (int q = 0; q < (D + 3) / 4; ++q) {
    v4f mm;
#pragma unroll
    for (int i = 0; i < 4; ++i) mm[i] = 4 * q + i < D ? g[4 * q + i] : 0.f;
    *reinterpret_cast<v4f*>(sG + lane * kLdY + 4 * q) = mm;
  }
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int j = 0; j < D; ++j) gm[j] = 0.f;
  static_for<0, 4>([&](auto cc) {
    constexpr int c = decltype(cc)::value;
    constexpr int S = S0 + c;
    tail_issue<S, NS>(A, p);
    const float* wf = reinterpret_cast<const float*>(p.wbuf + (S & 1) * kWStep) + lane;
#pragma unroll 2
    for (int kk = 0; kk < 16; ++kk) {
      float w[R], m[12];
#pragma unroll
      for (int r = 0; r < R; ++r) w[r] = wf[(kk * R + r) * 64];
#pragma unroll
      for (int q = 0; q < (D + 3) / 4; ++q) {
        const v4f mm = *reinterpret_cast<const v4f*>(sG + (16 * c + kk) * kLdY + 4 * q);
#pragma unroll
        for (int t = 0; t < 4; ++t) m[4 * q + t] = mm[t];
      }
#pragma unroll
      for (int j = 0; j < D; ++j) gm[j] += m[j] * w[r_of<0>(j)];
    }
    tail_commit<S>(p);
  });
}

}  // namespace

// Optional phase timing (build with AA_BUILD_DEFINES=-DAA_TAIL_TIMING): wave 0 of the middle workgroup stamps the shader
// clock at every phase boundary of its first tile into a small global buffer that the launcher prints.
#ifdef AA_TAIL_TIMING
__device__ unsigned long long g_tail_ticks[32];
#define AA_TTICK(i)                                                                   \
  if (blockIdx.x == gridDim.x / 2 && threadIdx.x == 0 && it == 1) g_tail_ticks[i] = __builtin_readcyclecounter();
#else
#define AA_TTICK(i)
#endif

template <class Sig0, class Sig1>
__global__ __launch_bounds__(64 * kTailWaves, 1) void fused_bwd_tail_kernel(FusedTailArgs A) {
  constexpr int D = Sig0::D2, R = Sig0::LMAX + 1;
  static_assert(Sig0::D1 == D && Sig0::DOUT == D && Sig1::D1 == D && Sig1::DOUT == 1 && D <= kTailD, "standard 2-layer stack, l_max <= 2");
  // the program: GM (4 env-weight blocks) | R6: [g_two_body | g_w0] @ G0^T (2 + 2R chunks) | R7: embed layer 1 reverse | R8: layer 0 reverse
  constexpr int S_GM = 0, S_R6 = 4, S_R7 = S_R6 + 2 + 2 * R, S_R8 = S_R7 + 2, NS = S_R8 + 2;
  static_assert(NS % 2 == 0 && NS <= kFusedMaxSteps, "the two LDS buffers alternate consistently across iterations");
  u32x4* wbuf = reinterpret_cast<u32x4*>(aa_smem);
  float* sRm = reinterpret_cast<float*>(wbuf + 2 * kWStep);  // [16: T*T <= 4 used] 1 / r_max per type pair, then [8] Bessel roots at 16
  float* sTab = sRm + 32;                                    // [T*T][8][64] two-body table
  const int ntab = A.num_types * A.num_types * 512;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, hh = lane >> 5, el = lane & 31;
  float* sW = sTab + ntab + wv * kTailWaveFloats;
  float* sY = sW;
  float* sP = sW + kTailOffP;
  float* sGM = sW + kTailOffGM;
  float* sB0 = sW + kTailOffB0;
  float* sB1 = sW + kTailOffB1;
  if (tid < A.num_types * A.num_types) sRm[tid] = A.rmax_recip[tid];
  if (tid >= 16 && tid < 24) sRm[tid] = A.embed_kind == 0 ? A.bessel_w[tid - 16] : 0.f;
  for (int i = tid; i < ntab; i += 64 * kTailWaves) sTab[i] = A.emb_tab[i];
  TailPipe p;
  p.wbuf = wbuf;
  p.tid = tid;
  p.lane = lane;
  p.wv = __builtin_amdgcn_readfirstlane(wv);
  tail_pipe_load(A, p, 0);
  tail_pipe_store(p, 0);
  // (the path weights are re-read per tile -- L1 / L2 hits -- instead of living in 14 registers across the MFMA phases)
  auto load_wp = [&](float* wp0, float* wp1) {
#pragma unroll
    for (int q = 0; q < Sig0::P; ++q) wp0[q] = A.coupling ? A.tpw0[lane * Sig0::P + q] : A.tpw0[q];
#pragma unroll
    for (int q = 0; q < Sig1::P; ++q) wp1[q] = A.coupling ? A.tpw1[lane * Sig1::P + q] : A.tpw1[q];
  };
  lds_barrier();  // tables + first weight step staged
  const int64_t ngroups = (A.atom_end - A.atom0 + kTailWaves - 1) / kTailWaves;
  for (int64_t it = 0; blockIdx.x + it * gridDim.x < ngroups; ++it) {
    // (the staging loads' lane offset is made opaque per iteration: otherwise the 64-bit addresses of all 16 weight blocks are
    //  hoisted out of this loop as loop invariants and live in -- spilled -- registers across the whole tile program)
    asm volatile("" : "+v"(p.tid));
    AA_TTICK(0)
    const int64_t atom = A.atom0 + (int64_t(blockIdx.x) + it * gridDim.x) * kTailWaves + wv;
    const bool atom_ok = atom < A.atom_end;
    int beg = 0, cnt = 0;
    if (atom_ok) {
      beg = A.rowptr[atom];
      cnt = A.rowptr[atom + 1] - beg;
    }
    beg = __builtin_amdgcn_readfirstlane(beg);
    cnt = __builtin_amdgcn_readfirstlane(cnt);
    const bool row_ok = el < cnt;
    const int64_t row0 = beg;                 // (wave-uniform: array bases at this row stay in scalar registers)
    const int64_t e = row0 + el;
    const int off64 = el * 64 + 4 * hh, offw = el * (64 * R) + 4 * hh, offg = el * A.ld_gtb + 4 * hh;
    // ---- REQUEST everything the first phases read, then compute: unit vector + length and the neighbor (forward), the
    //      atom's x2s blocks, the two scalar-gradient rows, the first irrep of w0
    v4f vv4 = {1.f, 0.f, 0.f, 1.f};
    int nbr = 0;
    if (row_ok) {
      vv4 = *reinterpret_cast<const v4f*>(A.vec + 4 * e);
      nbr = A.nbr[e];
    }
    float x2s0[D], x2s1[D];
#pragma unroll
    for (int j = 0; j < D; ++j) {
      x2s0[j] = atom_ok ? A.x2s0[(atom * D + j) * 64 + lane] : 0.f;
      x2s1[j] = atom_ok ? A.x2s1[(atom * D + j) * 64 + lane] : 0.f;
    }
    const float* gs0 = A.gscal0 + row0 * 64;
    const float* gs1 = A.gscal1 + row0 * 64;
    const float* w0t = A.w0 + row0 * (64 * R);
    const v16f g0a = ld_tile(gs0, off64, row_ok), g0b = ld_tile(gs0 + 32, off64, row_ok);
    const v16f g1a = ld_tile(gs1, off64, row_ok), g1b = ld_tile(gs1 + 32, off64, row_ok);
    v16f wq[2][2];  // w0 tiles of the current / the next irrep
    wq[0][0] = ld_tile(w0t, offw, row_ok);
    wq[0][1] = ld_tile(w0t + 32, offw, row_ok);
    AA_TTICK(1)
    // ---- geometry of the lane's edge; harmonics re-evaluated from the stored unit vector
    float Y[D];
    const float nx = vv4[0], ny = vv4[1], nz = vv4[2], rr = vv4[3];
    int pair = 0;
    if (row_ok) pair = A.types[atom] * A.num_types + A.types[nbr];
    {
      float Yf[16];
      sh_eval<float>(Sig0::LMAX, nx, ny, nz, Yf);
#pragma unroll
      for (int m = 0; m < D; ++m) Y[m] = row_ok ? Yf[m] : 0.f;
      if (hh == 0) {
#pragma unroll
        for (int m = 0; m < kLdY; ++m) sY[el * kLdY + m] = m < D ? Y[m] : 0.f;
      }
    }
    // ---- per-atom vectors (lane = channel): v = dSig1/dtf1 (x2s1), B1 = Sig0^T_x1(v, x2s0), B0 = Sig0^T_x1(e_0, x2s0)
    {
      float wp0[Sig0::P], wp1[Sig1::P], vv[D];
      load_wp(wp0, wp1);
      float B0[D], B1[D], e0[D], one[1] = {1.f};
#pragma unroll
      for (int j = 0; j < D; ++j) e0[j] = j == 0 ? 1.f : 0.f;
      Sig1::template bx1<float>(one, x2s1, wp1, vv);
      Sig0::template bx1<float>(vv, x2s0, wp0, B1);
      Sig0::template bx1<float>(e0, x2s0, wp0, B0);
#pragma unroll
      for (int a = 0; a < D; ++a) {
        sB0[a * 64 + lane] = B0[a];
        sB1[a * 64 + lane] = B1[a];
      }
      __builtin_amdgcn_wave_barrier();
    }
    AA_TTICK(2)
    AA_TTICK(3)
    // ---- per irrep: t_l = g_l * w0_r  ->  dE/dY (x1 path, per edge)  and  Q_l (per atom); the next irrep's w0 tiles are in flight
    float gy[D], Q1[D], Q0[D];
#pragma unroll
    for (int j = 0; j < D; ++j) {
      gy[j] = 0.f;
      Q1[j] = 0.f;
      Q0[j] = 0.f;
    }
    static_for<0, R>([&](auto rr_) {
      constexpr int r = decltype(rr_)::value;
      if constexpr (r + 1 < R) {
        wq[(r + 1) & 1][0] = ld_tile(w0t + (r + 1) * 64, offw, row_ok);
        wq[(r + 1) & 1][1] = ld_tile(w0t + (r + 1) * 64 + 32, offw, row_ok);
      }
      const v16f& wa = wq[r & 1][0];
      const v16f& wb = wq[r & 1][1];
      v16f t1a, t1b, t0a, t0b;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        t1a[i] = g1a[i] * wa[i];
        t1b[i] = g1b[i] * wb[i];
        t0a[i] = g0a[i] * wa[i];
        t0b[i] = g0b[i] * wb[i];
      }
      tile_gy_x1<r>(sB1 + 4 * hh, sB0 + 4 * hh, t1a, t1b, t0a, t0b, gy);
      tile_q_accumulate<r>(sP, sY, t1a, t1b, lane, Q1);
      tile_q_accumulate<r>(sP, sY, t0a, t0b, lane, Q0);
      __builtin_amdgcn_sched_barrier(0);
    });
    AA_TTICK(4)
    // ---- REQUEST the two-body gradient tiles (first two operand chunks of R6): in flight during the per-atom phase
    const float* gtb = A.g_tb + row0 * A.ld_gtb;
    v16f tb0 = ld_tile(gtb, offg, row_ok), tb1 = ld_tile(gtb + 32, offg, row_ok);
    // ---- per atom: d x2s0 = Sig0^T_x2(v, Q1) + Sig0^T_x2(e_0, Q0), scaled; GM[j][k] = sum_ch d x2s0[j][ch] Wenv0^T[r(j)][ch][k]
    float gm[D];
    {
      float wp0[Sig0::P], wp1[Sig1::P], vv[D], x21[D], one[1] = {1.f};
      load_wp(wp0, wp1);
#pragma unroll
      for (int j = 0; j < D; ++j) x21[j] = atom_ok ? A.x2s1[(atom * D + j) * 64 + lane] : 0.f;
      Sig1::template bx1<float>(one, x21, wp1, vv);  // (v again: 9 registers less across the edge phase)
      float ga[D], gb[D], e0[D], g2[D];
#pragma unroll
      for (int j = 0; j < D; ++j) e0[j] = j == 0 ? 1.f : 0.f;
      Sig0::template bx2<float>(vv, Q1, wp0, ga);
      Sig0::template bx2<float>(e0, Q0, wp0, gb);
#pragma unroll
      for (int j = 0; j < D; ++j) g2[j] = (ga[j] + gb[j]) * A.sf;
    AA_TTICK(5)
      project_gm<S_GM, NS, D, R>(A, p, sP, g2, gm);
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int j = 0; j < D; ++j) sGM[j * 64 + lane] = gm[j];
      __builtin_amdgcn_wave_barrier();
    }
    AA_TTICK(6)
    // ---- R6: g_emb = [g_two_body | g_w0] @ G0^T + g_aenv,  g_aenv[e][k] = sum_j Y[e][j] GM[j][k];  dE/dY of the env path rides
    //      along.  Operand chunks in the order  two-body (2) | channel half 0 of every irrep | channel half 1 of every irrep
    //      (the weight program follows, fused_bwd_tail_chunk_order): each half's two gradient tiles are read again -- L2 hits,
    //      this wave read them a moment ago -- once, a step or more ahead of their first use, instead of being held in 64
    //      registers across the per-atom phases.
    v16f k0, k1;
    v16f gh1[2], gh0[2], ea, eb;  // [half]: gradient tiles of the channel half being consumed
    tail_layer<S_R6, NS, 2 + 2 * R>(
        A, p,
        [&](auto kc) {
          constexpr int k = decltype(kc)::value;
          if constexpr (k == 0) {
            gh1[0] = ld_tile(gs1, off64, row_ok);
            gh0[0] = ld_tile(gs0, off64, row_ok);
          } else if constexpr (k == R + 1) {  // (the last half-0 chunk is step R + 1)
            gh1[1] = ld_tile(gs1 + 32, off64, row_ok);
            gh0[1] = ld_tile(gs0 + 32, off64, row_ok);
          } else if constexpr (k == 2 * R + 1) {
            ea = ld_tile(A.emb + row0 * 64, off64, row_ok);
            eb = ld_tile(A.emb + row0 * 64 + 32, off64, row_ok);
          }
        },
        [&](auto kc) -> v16f {
          constexpr int k = decltype(kc)::value;
          if constexpr (k == 0) {
            return tb0;
          } else if constexpr (k == 1) {
            return tb1;
          } else {
            constexpr int h = (k - 2) / R, r = (k - 2) % R;
            return tile_gw0<r, h>(sB1 + 4 * hh, sB0 + 4 * hh, Y, gh1[h], gh0[h]);
          }
        },
        [&](const v16f& a0, const v16f& a1) {
          k0 = a0;
          k1 = a1;
          const float* gmv = sGM + 4 * hh;
#pragma unroll
          for (int j = 0; j < D; ++j) {
            float s = 0.f;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const v4f c0 = *reinterpret_cast<const v4f*>(gmv + j * 64 + 8 * q);
              const v4f c1 = *reinterpret_cast<const v4f*>(gmv + j * 64 + 32 + 8 * q);
#pragma unroll
              for (int i = 0; i < 4; ++i) {
                k0[4 * q + i] += Y[j] * c0[i];
                k1[4 * q + i] += Y[j] * c1[i];
                s += ea[4 * q + i] * c0[i] + eb[4 * q + i] * c1[i];
              }
            }
            gy[j] += s;
            anchor(gy[j]);
            __builtin_amdgcn_sched_barrier(0);
          }
        });
    // ---- REQUEST the pre-activation tiles of R7's epilogue and the layer-1 dE/dY slot of the final sum
    const v16f za = ld_tile(A.se_h + row0 * 64, off64, row_ok), zb = ld_tile(A.se_h + row0 * 64 + 32, off64, row_ok);
    float ge1[D];
#pragma unroll
    for (int j = 0; j < D; ++j) ge1[j] = (A.gsh_env1 && row_ok && hh == 0) ? A.gsh_env1[e * D + j] : 0.f;
    AA_TTICK(7)
    // ---- R7: g_h = (g_emb @ W_e1^T) * silu'(h_e)
    tail_layer<S_R7, NS, 2>(
        A, p, [](auto) {}, [&](auto kc) -> const v16f& { if constexpr (decltype(kc)::value == 0) return k0; else return k1; },
        [&](const v16f& a0, const v16f& a1) {
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            k0[i] = a0[i] * dsilu(za[i]);
            k1[i] = a1[i] * dsilu(zb[i]);
          }
        });
    AA_TTICK(8)
    // ---- R8: g_emb0 = g_h @ W_e0^T, contracted with the two-body table of the lane's type pair to the 8 basis sums
    float tsum[8];
    tail_layer<S_R8, NS, 2>(
        A, p, [](auto) {}, [&](auto kc) -> const v16f& { if constexpr (decltype(kc)::value == 0) return k0; else return k1; },
        [&](const v16f& a0, const v16f& a1) {
          const float* tb = sTab + pair * 512 + 4 * hh;
#pragma unroll
          for (int n = 0; n < 8; ++n) {
            float s = 0.f;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const v4f t0 = *reinterpret_cast<const v4f*>(tb + n * 64 + 8 * q);
              const v4f t1 = *reinterpret_cast<const v4f*>(tb + n * 64 + 32 + 8 * q);
#pragma unroll
              for (int i = 0; i < 4; ++i) s += a0[4 * q + i] * t0[i] + a1[4 * q + i] * t1[i];
            }
            tsum[n] = s;
          }
        });
    AA_TTICK(9)
    // ---- both lane halves of a row: own + partner
    {
      float o8[8];
#pragma unroll
      for (int n = 0; n < 8; ++n) o8[n] = tsum[n];
      permlane32_swap4(tsum, o8);
      permlane32_swap4(tsum + 4, o8 + 4);
#pragma unroll
      for (int n = 0; n < 8; ++n) tsum[n] += o8[n];
      float og[12], gg[12];
#pragma unroll
      for (int j = 0; j < 12; ++j) {
        gg[j] = j < D ? gy[j] : 0.f;
        og[j] = gg[j];
      }
      permlane32_swap4(gg, og);
      permlane32_swap4(gg + 4, og + 4);
      permlane32_swap4(gg + 8, og + 8);
#pragma unroll
      for (int j = 0; j < D; ++j) gy[j] = gg[j] + og[j];
    }
    if (row_ok && hh == 0) {
      // dE/dY of the layer-1 env path comes from tp_mom_bwd_last
#pragma unroll
      for (int j = 0; j < D; ++j) gy[j] += ge1[j];
      if (A.dvec) {
        // ---- edge_backward (aa_edge.hip): chain rule to the edge vector
        const float x = rr * sRm[pair];
        float dEdx = 0.f;
        if (A.embed_kind == 1) {
#pragma unroll
          for (int n = 0; n < 8; ++n) {
            float bv, dbv;
            spline_basis_and_grad<float>(x, n, 8, A.spline_span, bv, dbv);
            dEdx += tsum[n] * dbv;
          }
        } else {
          float f, df;
          cutoff_and_grad<float>(x, A.poly_p, f, df);
#pragma unroll
          for (int n = 0; n < 8; ++n) {
            const float w = sRm[16 + n];
            const float s = aa_sin(w * x), c = aa_cos(w * x);
            const float bv = s / x;
            const float dbv = (w * c * x - s) / (x * x);
            dEdx += tsum[n] * (dbv * f + bv * df);
          }
        }
        const float dEdr = dEdx * sRm[pair];
        float gx, gyv, gz;
        sh_grad<float>(Sig0::LMAX, nx, ny, nz, gy, gx, gyv, gz);
        const float dot = gx * nx + gyv * ny + gz * nz;
        const float inv = 1.f / rr;
        *reinterpret_cast<v4f*>(A.dvec + 4 * e) = v4f{dEdr * nx + (gx - dot * nx) * inv, dEdr * ny + (gyv - dot * ny) * inv,
                                                      dEdr * nz + (gz - dot * nz) * inv, 0.f};
      } else {
#pragma unroll
        for (int j = 0; j < D; ++j) A.gsh_out[e * D + j] = gy[j];
        *reinterpret_cast<v4f*>(A.trev + 8 * e) = v4f{tsum[0], tsum[1], tsum[2], tsum[3]};
        *reinterpret_cast<v4f*>(A.trev + 8 * e + 4) = v4f{tsum[4], tsum[5], tsum[6], tsum[7]};
      }
    }
    AA_TTICK(10)
  }
}

size_t fused_bwd_tail_lds_bytes(int num_types) {
  return sizeof(u32x4) * 2 * kWStep + sizeof(float) * (32 + size_t(num_types) * num_types * 512 + size_t(kTailWaves) * kTailWaveFloats);
}

int fused_bwd_tail_num_steps(int R) { return 4 + (2 + 2 * R) + 2 + 2; }

// 32-deep operand chunk of the first linear layer (index into [two-body (2) | w0 irrep-major (2 R)]) consumed at its step i
int fused_bwd_tail_chunk_order(int R, int i) { return i < 2 ? i : 2 + 2 * ((i - 2) % R) + (i - 2) / R; }

int launch_fused_bwd_tail(int pair, const FusedTailArgs& a, hipStream_t stream) {
  if (a.atom_end <= a.atom0) return AA_OK;
  if (a.num_types < 1 || a.num_types > 2) return fail(AA_ERR_INVALID, "fused reverse tail: 1..2 species");
  const size_t smem = fused_bwd_tail_lds_bytes(a.num_types);
  if (smem > 160 * 1024) return fail(AA_ERR_INVALID, "fused reverse tail: LDS budget exceeded");
  static int num_cu = 0;
  if (num_cu == 0) {
    int dev = 0, n = 0;
    AA_CHECK_HIP(hipGetDevice(&dev));
    AA_CHECK_HIP(hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev));
    num_cu = n > 0 ? n : 256;
  }
  const int64_t ngroups = (a.atom_end - a.atom0 + kTailWaves - 1) / kTailWaves;
  dim3 grid((unsigned)std::min<int64_t>(ngroups, num_cu));
#define AA_TAIL_LAUNCH(S0_, S1_)                                                                                     \
  {                                                                                                                  \
    const void* fn = (const void*)fused_bwd_tail_kernel<cg::S0_, cg::S1_>;                                           \
    AA_CHECK_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, int(smem)));                    \
    hipLaunchKernelGGL((fused_bwd_tail_kernel<cg::S0_, cg::S1_>), grid, dim3(64 * kTailWaves), smem, stream, a);     \
  }
  if (pair == 0)
    AA_TAIL_LAUNCH(Sig1, Sig0)
  else if (pair == 1)
    AA_TAIL_LAUNCH(Sig5, Sig4)
  else
    return fail(AA_ERR_INVALID, "fused reverse tail: unsupported signature pair");
#undef AA_TAIL_LAUNCH
  AA_CHECK_HIP(hipGetLastError());
#ifdef AA_TAIL_TIMING
  {
    static int calls = 0;
    if (++calls == 8) {  // a warm call
      unsigned long long t[32];
      AA_CHECK_HIP(hipStreamSynchronize(stream));
      AA_CHECK_HIP(hipMemcpyFromSymbol(t, HIP_SYMBOL(g_tail_ticks), sizeof(t)));
      static const char* nm[10] = {"geometry", "per-atom CG 1", "g tile loads", "irrep loop", "per-atom CG 2", "GM", "R6 + epilogue", "R7", "R8", "edge math"};
      for (int i = 0; i < 10; ++i) fprintf(stderr, "[tail timing] %-16s %8llu cycles\n", nm[i], t[i + 1] - t[i]);
      fprintf(stderr, "[tail timing] %-16s %8llu cycles\n", "total", t[10] - t[0]);
    }
  }
#endif
  return AA_OK;
}

}  // namespace aa
