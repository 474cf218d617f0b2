// The fused per-atom-tile forward of aa_fused.hip at TWO waves per SIMD (round 6).
//
// Why.  The one-tile kernel of aa_fused.hip needs the whole register file (424 VGPRs) and runs one wave per SIMD.  Its
// instruction stream is MFMA-dense weight steps (24 MFMAs and ~50 other instructions between two barriers) alternating with
// MFMA-free epilogues and per-atom phases (300-1000 vector / LDS / scalar instructions each): ~11 000 instructions per tile of
// which 576 are MFMAs.  A single in-order wave issues one instruction every ~4 cycles and can hide at most ~5 of them behind a
// 32-cycle MFMA, so the tile time is the SUM of its phases (85 k cycles; matrix pipe 0.21, vector issue 0.25, waiting 0.32:
// profiles/r05_v48_bound_table_c4.md).  Two waves per SIMD let one wave's vector phase issue beside the other's waits and
// MFMAs -- provided the kernel fits 256 registers and half the LDS per wave WITHOUT spills (the round-2 / round-3 attempts at
// two waves per SIMD spilled 150-430 registers and lost; HISTORY.md sections 9.2, 9.4).
//
// How it fits.  One workgroup of EIGHT waves per CU (8 atoms in flight per CU instead of 4) walking the weight program in lock
// step, so the 24 KB of double-buffered weight steps are shared by eight tiles (half the L2 -> LDS weight traffic and half the
// staging work per tile).  Per wave 16.6 KB of LDS and <= 256 registers:
//   * the two-body scalars (three consumers) are parked raw in LDS (8 KB), the latent-0 activation (two consumers) stays in 32
//     registers; nothing is held pre-split (KEEP) and w0 is not held (HOLD): 192 registers of the one-tile form are gone;
//   * layer 1's tensor-track scalars take w0 from the rows this wave stored for the reverse pass a few thousand cycles earlier
//     (L2-resident; 24 16-byte loads per lane) instead of recomputing it with 6 MFMA steps or holding it in 96 registers;
//   * latent 1 and the readout hidden layer share their first four operand chunks [two-body | lat0]: the merged phase walks
//     those chunks ONCE (one fetch + one bf16x3 split per chunk instead of two) into two accumulator pairs, after which the
//     parked tiles and lat0 are dead;
//   * the moments go through a HALF patch: one 32-feature tile at a time ([32][36] floats, the store-transpose patch itself),
//     lanes 0..31 / 32..63 walk rows 0..15 / 16..31 and one v_permlane32_swap per component joins the halves -- 4.6 KB instead
//     of the 8.7-KB [32][68] patch; harmonics rows with stride 12 instead of 16.
// Program (R irreps): Wenv0 (4 steps) | first stage (2 + 2 R) | latent 0 hidden (4) | Wenv1 (4) | latent 1: scal1 chunks (2) |
// merged [lat0, two-body] chunks x {latent 1, readout} (8) | readout: lat1 chunks (2) = 32 steps at l_max 2, 24 of them MFMA
// steps -- the count of the w0-holding one-tile form.
//
// Same inputs, outputs and masking rules as fused_fwd_kernel<..., TEAMS = false>; taken for one-species plans with every fold
// of DESIGN.md section 3.2 active and w0 stored (launch_fused_fwd, FusedFwdArgs::wide).  Semantics: allegro/nn/_allegro.py:237-301.
#include "aa_fused_tile.h"

namespace aa {

#ifdef AA_FUSED_TIMING
__device__ unsigned long long g_fused8_ticks[32];
#define AA_TICK8(i) \
  if (blockIdx.x == gridDim.x / 2 && threadIdx.x == 0) g_fused8_ticks[i] = __builtin_readcyclecounter();
#else
#define AA_TICK8(i)
#endif

namespace {

constexpr int kLdY8 = 12;                    // row stride of the harmonics rows [32][12] and of sM [64 k][12]
constexpr int kOffB8 = 32 * kLdT;            // sB [D <= 9][64] behind the store / moments patch
constexpr int kOffY8 = kOffB8 + 9 * 64;      // sY [32][12]
constexpr int kOffPark8 = kOffY8 + 32 * kLdY8;  // two parked 32-feature tiles (the two-body scalars)
constexpr int fused_wave_floats(int waves) { return kOffPark8 + (waves == 8 ? 2 : 1) * kTileFloats; }  // 16 640 / 12 544 B per wave
static_assert(64 * kLdY8 <= 32 * kLdT, "sM must fit the patch");

// M[j] (lane = k) = sum over the tile's rows of Y[e][j] * a[e][k], one 32-feature tile at a time through the [32][kLdT]
// patch: lane (k = lane & 31, half = lane >> 5) walks rows 16 half .. 16 half + 15 of its column; the halves meet through one
// v_permlane32_swap per component (a = [M0.lo | M1.lo], b = [M0.hi | M1.hi]; a + b: lanes < 32 feature k of tile 0, lanes >= 32
// feature k of tile 1 = feature `lane` of the pair).  Rows beyond the segment carry Y = 0.
template <int D>
__device__ __forceinline__ void tile_moments_half(float* sT, const float* sY, const v16f& t0, const v16f& t1, int lane, float* M) {
  const int el = lane & 31, hh = lane >> 5;
  float M0[D], M1[D];
#pragma unroll
  for (int j = 0; j < D; ++j) {
    M0[j] = 0.f;
    M1[j] = 0.f;
  }
  auto pass = [&](const v16f& t, float* Mx) {
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int q = 0; q < 4; ++q) *reinterpret_cast<v4f*>(sT + el * kLdT + 8 * q + 4 * hh) = v4f{t[4 * q], t[4 * q + 1], t[4 * q + 2], t[4 * q + 3]};
    __builtin_amdgcn_wave_barrier();
    const float* col = sT + (16 * hh) * kLdT + el;
    const float* yr = sY + (16 * hh) * kLdY8;
#pragma unroll 4
    for (int i = 0; i < 16; ++i) {
      const float a = col[i * kLdT];
      float y[12];
#pragma unroll
      for (int q = 0; q < (D + 3) / 4; ++q) {
        const v4f yy = *reinterpret_cast<const v4f*>(yr + i * kLdY8 + 4 * q);
#pragma unroll
        for (int c = 0; c < 4; ++c) y[4 * q + c] = yy[c];
      }
#pragma unroll
      for (int j = 0; j < D; ++j) Mx[j] += y[j] * a;
    }
    __builtin_amdgcn_wave_barrier();
  };
  pass(t0, M0);
  pass(t1, M1);
#pragma unroll
  for (int j = 0; j < D; ++j) {
    permlane32_swap(M0[j], M1[j]);
    M[j] = M0[j] + M1[j];
  }
}

// one weight step: the 24 MFMAs of chunk `x` into the accumulator pair, `between` (the split of the next chunk, a parked-tile
// fetch) behind them, then the commit (next block into LDS, barrier)
template <int S, int NS, class Args, class Pipe, class F>
__device__ __forceinline__ void fused_step8(const Args& A, Pipe& p, const XSplit& x, v16f& a0, v16f& a1, F&& between) {
  pipe_issue<S, NS>(A, p);
  fused_mma_step(p.wbuf + (S & 1) * kWStep, p.lane, x, a0, a1);
  between();
  pipe_commit<S>(p);
}

// Global addressing rule of this kernel: every access is a WAVE-UNIFORM base pointer (scalar registers, formed per tile from the
// uniform row / atom index) + an UNSIGNED 32-bit lane offset, so that it is one saddr + voffset instruction.  With per-lane 64-bit
// addresses the optimizer hoists dozens of loop-invariant address pairs out of the persistent loop and, at 256 registers, spills
// them: every use inside the loop then is a scratch reload (a vector-memory operation behind the in-order vmcnt).

// a 32-feature tile (accumulator layout) of rows [0, 32) of the row-major rows at `base` (row stride ld); rows beyond the segment: zeros
__device__ __forceinline__ v16f tile_load_rows8(const float* base, unsigned ld, int lane, bool row_ok) {
  unsigned el = lane & 31;
  const unsigned hh = lane >> 5;
  opaque_vector(el);  // (see tile_store_rows8)
  v16f t;
  const unsigned off = (el * ld + 4u * hh) * 4u;  // (bytes: see at_bytes)
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    v4f v = {0.f, 0.f, 0.f, 0.f};
    if (row_ok) v = *reinterpret_cast<const v4f*>(at_bytes(base, off + 32u * q));
#pragma unroll
    for (int i = 0; i < 4; ++i) t[4 * q + i] = v[i];
  }
  return t;
}

// tile_store_rows (aa_mfma.h) with that addressing: `base` = first row of the segment (uniform), cnt rows are written
__device__ __forceinline__ void tile_store_rows8(float* sT, const v16f& acc, float* base, int cnt, unsigned ld, int lane) {
  const int el = lane & 31, hh = lane >> 5;
  float* st = sT + el * kTileLdT + 4 * hh;
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int q = 0; q < 4; ++q) *reinterpret_cast<v4f*>(st + 8 * q) = v4f{acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]};
  __builtin_amdgcn_wave_barrier();
  const int pr = lane >> 3, pc = 4 * (lane & 7);
  v4f v[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) v[q] = *reinterpret_cast<const v4f*>(sT + (8 * q + pr) * kTileLdT + pc);
  __builtin_amdgcn_wave_barrier();
  // (the row offsets are RE-derived from an opaque copy of the lane's row at every call: as loop invariants -- two or three per row
  //  stride -- they are hoisted out of the persistent loop, spilled in its prologue and reloaded from scratch at every store)
  unsigned pro = unsigned(pr);
  opaque_vector(pro);
  const unsigned off = (pro * ld + unsigned(pc)) * 4u;  // (bytes)
#pragma unroll
  for (int q = 0; q < 4; ++q)
    if (pr + 8 * q < cnt) *reinterpret_cast<v4f*>(at_bytes(base, off + 32u * q * ld)) = v[q];
}

// tile_scal_accumulate (aa_mfma.h) with ONE cell buffer: s[e][ch] += w[e][r][ch] * sum_{a in irrep RR} Y[e][a] * B[a][ch].  The
// two-deep request ring of the one-tile form costs 4 (2 RR + 1) more registers; here the other wave of the SIMD covers the LDS latency.
template <int RR>
__device__ __forceinline__ void tile_scal_accumulate8(const float* bb, const float* Y, const v16f& w0a, const v16f& w0b, v16f& s0, v16f& s1) {
  constexpr int a0 = RR * RR, na = 2 * RR + 1;
#pragma unroll
  for (int g = 0; g < 8; ++g) {
    v4f T4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int a = 0; a < na; ++a) {
      const v4f b = *reinterpret_cast<const v4f*>(bb + (a0 + a) * 64 + 32 * (g >> 2) + 8 * (g & 3));
#pragma unroll
      for (int i = 0; i < 4; ++i) T4[i] += Y[a0 + a] * b[i];
    }
    const int q = g & 3;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (g < 4)
        s0[4 * q + i] += w0a[4 * q + i] * T4[i];
      else
        s1[4 * q + i] += w0b[4 * q + i] * T4[i];
    }
    if (g < 4)
      anchor(s0);
    else
      anchor(s1);
    __builtin_amdgcn_sched_barrier(0);
  }
}

struct TileIn8 {
  int beg, cnt;
  int j;
  float pi[3], pj[3], sv[3];
  int ti, tj;
};

}  // namespace

constexpr int fused_fwd8_steps(int R, bool pm, bool tail = false) { return (pm ? 2 * R : 4) + (2 + 2 * R) + 4 + (pm ? 2 * R : 4) + 2 + 8 + 2 + (tail ? 12 : 0); }

// WAVES = 8: one workgroup per CU, eight tiles in lock step (both two-body tiles parked in LDS: 16.6 KB per wave).
// WAVES = 4: TWO independent workgroups per CU (78 KB of LDS each: one two-body tile parked, the other in 16 registers), each with
//            its own weight pipeline -- the two waves of a SIMD drift apart, so one workgroup's MFMA steps run beside the other's
//            vector / LDS phases instead of both hitting the same pipe at the same time.
// TAIL: the readout-reverse chain of the reverse pass (Runner::backward "B3": d ro_h = factor scale w silu'(ro_h); d h1 = (d ro_h W_a)
//       silu'(h1); d EDGE_FEATURES[:, :128] = [d ro_h | d h1] W_b; d scal1 = d h1 W_c -- 12 more steps on the same weight fragments the
//       chain kernel uses) runs here, where its operands are in registers: the total energy is a plain sum, so the gradient seeds of
//       every edge-local layer are known at the end of the edge's own forward.
template <class Sig0, class Sig1, int WAVES, bool PM, bool TAIL = false>
__global__ __launch_bounds__(64 * WAVES, 2) void fused_fwd8_kernel(FusedFwdArgs A) {
  static_assert(WAVES == 8 || WAVES == 4, "workgroup forms");
  constexpr int NT = 64 * WAVES;                   // threads
  constexpr int NPARK = WAVES == 8 ? 2 : 1;        // parked two-body tiles
  constexpr int kWaveF = kOffPark8 + NPARK * kTileFloats;  // floats of LDS per wave
  constexpr int D = Sig0::D2, R = Sig0::LMAX + 1;
  static_assert(Sig0::D1 == D && Sig0::DOUT == D && Sig1::D1 == D && Sig1::DOUT == 1, "standard 2-layer stack");
  static_assert(D <= 9, "l_max <= 2");
  static_assert(kFoldEmbed && kFoldEmb1 && kFoldLatent, "the wide form exists for the folded program only");
  constexpr int kPS = PM ? 2 * R : 4;  // pipeline steps of one env projection
  constexpr int S_P0 = 0, S_L2 = kPS, S_L3 = S_L2 + 2 + 2 * R, S_P1 = S_L3 + 4, S_L6A = S_P1 + kPS, S_M = S_L6A + 2, S_L8K = S_M + 8, S_T = S_L8K + 2,
                NS = S_T + (TAIL ? 12 : 0);
  static_assert(NS % 2 == 0 && NS <= kFusedMaxSteps && NS == fused_fwd8_steps(R, PM, TAIL), "program length");
  u32x4* wbuf = reinterpret_cast<u32x4*>(aa_smem);
  float* sRo = reinterpret_cast<float*>(wbuf + 2 * kWStep);  // [64] last readout weights
  float* sRm = sRo + 64;                                     // [16] 1 / r_max per type pair, [8] Bessel roots at 16
  float* sTab = sRm + 32;                                    // [T*T][8][64] two-body table
  const int ntab = A.num_types * A.num_types * 512;
  const int tid = threadIdx.x, lane = tid & 63, hh = lane >> 5, el = lane & 31;
  const int wvs = __builtin_amdgcn_readfirstlane(tid >> 6);
  float* sW = sTab + ntab + wvs * kWaveF;  // patch (store transpose / moments half / sM)
  float* sBv = sW + kOffB8;
  float* sY = sW + kOffY8;
  float* sPark = sW + kOffPark8;
  for (int i = tid; i < 64; i += NT) sRo[i] = A.ro_w[i];
  if (tid < A.num_types * A.num_types) sRm[tid] = A.rmax_recip[tid];
  if (tid >= 16 && tid < 24) sRm[tid] = A.embed_kind == 0 ? A.bessel_w[tid - 16] : 0.f;
  for (int i = tid; i < ntab; i += NT) sTab[i] = A.emb_tab[i];
  std::conditional_t<WAVES == 8, FusedPipe8, FusedPipe> p;
  p.wbuf = wbuf;
  p.tid = tid;
  p.lane = lane;
  p.zero = 0;
  if constexpr (WAVES == 8) {
    p.wv = wvs;
    u32x4 r[2];
    pipe_load8(A, p, 0, r);
    pipe_store8(p, 0, r);
    pipe_load8(A, p, 1, p.rb);  // (step 1 lands in LDS at the end of step 0)
  } else {
    p.stager = true;
    u32x4 r[3];
    pipe_load(A, tid, 0, r);
    pipe_store(wbuf, 0, tid, r);
    pipe_load(A, tid, 1, p.rb);
  }
  const int64_t ngroups = (A.atom_end - A.atom0 + WAVES - 1) / WAVES;
  auto group_of = [&](int64_t it) { return int64_t(blockIdx.x) + it * gridDim.x; };
  auto atom_of = [&](int64_t it) -> int64_t { return A.atom0 + group_of(it) * WAVES + wvs; };
  auto load_rows = [&](int64_t atom, int& beg, int& cnt) {
    beg = 0;
    cnt = 0;
    if (atom < A.atom_end) {
      const int b0 = A.rowptr[atom], deg = A.rowptr[atom + 1] - b0;
      beg = b0;
      cnt = deg > 32 ? 32 : deg;
      // (the hint rules of fused_fwd_kernel: a long atom belongs to the team pass of the mixed form, or the hint was stale)
      if (A.skip_long && deg > 32 && deg <= kFusedMaxDegree) {
        cnt = -2;
      } else if (deg > 32) {
        cnt = -1;
        if (A.status && lane == 0) *reinterpret_cast<volatile int32_t*>(A.status) = deg;
        if (hh == 0 && el < deg) *reinterpret_cast<v4f*>((A.vec + 4 * int64_t(b0)) + 4u * unsigned(el)) = v4f{__builtin_nanf(""), __builtin_nanf(""), __builtin_nanf(""), __builtin_nanf("")};
      }
    }
  };
  auto load_nbr = [&](int beg, int cnt) { return el < cnt ? (A.nbr + beg)[unsigned(el)] : 0; };  // (beg: uniform)
  auto load_geo = [&](int64_t atom, TileIn8& t) {  // (atom, t.beg: uniform)
    if (el < t.cnt) {
      const float* pi = A.pos + 3 * atom;
      const unsigned oj = 3u * unsigned(t.j), oe = 3u * unsigned(el);
      const float* svb = A.shift_vec ? A.shift_vec + 3 * int64_t(__builtin_amdgcn_readfirstlane(t.beg)) : nullptr;
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        t.pi[q] = pi[q];
        t.pj[q] = A.pos[oj + q];
        t.sv[q] = svb ? svb[oe + q] : 0.f;
      }
      t.ti = A.types[atom];
      t.tj = A.types[unsigned(t.j)];
    }
  };
  auto load_pw = [&](const float* w, int P, int q) { return A.coupling ? w[unsigned(lane * P + q)] : w[q]; };
  int64_t a_cur = atom_of(0), a_nxt = atom_of(1), a_nn = atom_of(2);
  TileIn8 cur, nxt;
  int beg2 = 0, cnt2 = 0;
  load_rows(a_cur, cur.beg, cur.cnt);
  load_rows(a_nxt, nxt.beg, nxt.cnt);
  cur.j = load_nbr(cur.beg, cur.cnt);
  load_geo(a_cur, cur);
  lds_barrier();  // tables + first weight step staged
  for (int64_t it = 0; group_of(it) < ngroups; ++it) {
    opaque_scalar(p.zero);
    AA_TICK8(0)
    const int64_t atom = a_cur;
    const int64_t a_n3 = atom_of(it + 3);
    const int beg = __builtin_amdgcn_readfirstlane(cur.beg), cnt = __builtin_amdgcn_readfirstlane(cur.cnt);
    const bool atom_ok = atom < A.atom_end && cnt != -2;
    const bool row_ok = el < cnt;
    const int64_t row0 = beg;
    nxt.j = load_nbr(nxt.beg, nxt.cnt);
    load_rows(a_nn, beg2, cnt2);
    // ---- geometry of the lane's edge
    float Y[D], basis[8];
    int pair = 0;
    {
      float vx = 1.f, vy = 0.f, vz = 0.f;
      float x = 0.5f;
      if (row_ok) {
        vx = cur.pj[0] - cur.pi[0] + cur.sv[0];
        vy = cur.pj[1] - cur.pi[1] + cur.sv[1];
        vz = cur.pj[2] - cur.pi[2] + cur.sv[2];
        pair = cur.ti * A.num_types + cur.tj;
      }
      const float rr = aa_sqrt(vx * vx + vy * vy + vz * vz);
      const float inv = 1.f / rr;
      const float nx = vx * inv, ny = vy * inv, nz = vz * inv;
      if (row_ok) x = rr * sRm[pair];
      float Yf[16];
      sh_eval<float>(Sig0::LMAX, nx, ny, nz, Yf);
#pragma unroll
      for (int m = 0; m < D; ++m) Y[m] = row_ok ? Yf[m] : 0.f;
      if (row_ok && hh == 0) {
        *reinterpret_cast<v4f*>(at_bytes(A.vec + 4 * row0, 16u * unsigned(el))) = v4f{nx, ny, nz, rr};
        if (A.sh) {
          float* shb = A.sh + row0 * D;
#pragma unroll
          for (int m = 0; m < D; ++m) *at_bytes(shb, 4u * (unsigned(el) * D + m)) = Yf[m];
        }
      }
      __builtin_amdgcn_wave_barrier();  // (the previous tile's readers of sY are done: program order + wave-private region)
      if (hh == 0) {
#pragma unroll
        for (int q = 0; q < kLdY8 / 4; ++q) {
          v4f yy;
#pragma unroll
          for (int i = 0; i < 4; ++i) yy[i] = 4 * q + i < D ? Y[4 * q + i] : 0.f;
          *reinterpret_cast<v4f*>(sY + el * kLdY8 + 4 * q) = yy;
        }
      }
      if (A.embed_kind == 1) {
#pragma unroll
        for (int n = 0; n < 8; ++n) {
          float dbv;
          spline_basis_and_grad<float>(x, n, 8, A.spline_span, basis[n], dbv);
        }
      } else {
        float f, df;
        cutoff_and_grad<float>(x, A.poly_p, f, df);
        const float fx = f / x;
#pragma unroll
        for (int n = 0; n < 8; ++n) basis[n] = aa_sin(sRm[16 + n] * x) * fx;
      }
    }
    AA_TICK8(1)
    // ---- two-body table: pre-activation h of scalar_embed_mlp's hidden layer (kFoldEmbed); a_e = silu(h) stands in for the
    //      embedding (kFoldEmb1)
    v16f em0, em1;
    {
      const float* tb = sTab + pair * 512 + 4 * hh;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        em0[r] = 0.f;
        em1[r] = 0.f;
      }
#pragma unroll
      for (int n = 0; n < 8; ++n) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const v4f t0 = *reinterpret_cast<const v4f*>(tb + n * 64 + 8 * q);
          const v4f t1 = *reinterpret_cast<const v4f*>(tb + n * 64 + 32 + 8 * q);
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            em0[4 * q + i] += basis[n] * t0[i];
            em1[4 * q + i] += basis[n] * t1[i];
          }
        }
        // (one table row per region, accumulators pinned: unconstrained, all 64 row reads are gathered at the front -- 256 registers --
        //  and the arithmetic is sunk to its first use)
        anchor(em0);
        anchor(em1);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    tile_store_rows8(sW, em0, (A.se_h) + row0 * 64, cnt, 64, lane);
    tile_store_rows8(sW, em1, (A.se_h + 32) + row0 * 64, cnt, 64, lane);
    keep_tile<true>(em0, em0);
    keep_tile<true>(em1, em1);
    tile_store_rows8(sW, em0, (A.emb) + row0 * 64, cnt, 64, lane);
    tile_store_rows8(sW, em1, (A.emb + 32) + row0 * 64, cnt, 64, lane);
    AA_TICK8(2)
    // ---- per-atom part of layer 0: moments of a_e -> x2s0 -> B0
    float x2s0[D];
    {
      float wp0[Sig0::P];
#pragma unroll
      for (int q = 0; q < Sig0::P; ++q) wp0[q] = load_pw(A.tpw0, Sig0::P, q);
      float M[D];
      tile_moments_half<D>(sW, sY, em0, em1, lane, M);
      if constexpr (PM)
        project_moments_mfma<S_P0, NS, D, R, kLdY8>(A, p, sW, M, A.sf, x2s0, sBv);
      else
        project_moments<S_P0, NS, D, R, kLdY8>(A, p, sW, M, A.sf, x2s0);
      if (atom_ok) {
#pragma unroll
        for (int j = 0; j < D; ++j) *at_bytes(A.x2s0 + (atom * D + j) * 64, 4u * unsigned(lane)) = x2s0[j];
      }
      float e0[D], B0[D];
#pragma unroll
      for (int k = 0; k < D; ++k) e0[k] = k == 0 ? 1.f : 0.f;
      Sig0::template bx1<float>(e0, x2s0, wp0, B0);
#pragma unroll
      for (int a = 0; a < D; ++a) sBv[a * 64 + lane] = B0[a];
      __builtin_amdgcn_wave_barrier();
    }
    AA_TICK8(3)
    // ---- first stage: [two-body scalars | w0 irrep 0 | 1 | ...] = a_e @ W; layer-0 tensor-track scalars behind each irrep's pair
    // (the lane's harmonics row comes back from sY where it is used: 9 registers less across the MFMA and per-atom phases)
    auto load_Y = [&](float* Yv) {
#pragma unroll
      for (int q = 0; q < (D + 3) / 4; ++q) {
        const v4f yy = *reinterpret_cast<const v4f*>(sY + el * kLdY8 + 4 * q);
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (4 * q + i < D) Yv[4 * q + i] = yy[i];
      }
    };
    v16f sc0, sc1;
    v16f tbr;  // (WAVES == 4: the second two-body tile, in registers)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      sc0[r] = 0.f;
      sc1[r] = 0.f;
    }
    fused_layer<S_L2, NS, 2, 2 + 2 * R>(A, p,
                                        [&](auto kc) -> const v16f& { if constexpr (decltype(kc)::value == 0) return em0; else return em1; },
                                        [&](auto ntp, const v16f& a0, const v16f& a1) {
                                          constexpr int q = decltype(ntp)::value;
                                          if constexpr (q == 0) {
                                            park_tile(sPark, a0, lane);
                                            if constexpr (NPARK == 2)
                                              park_tile(sPark + kTileFloats, a1, lane);
                                            else
                                              tbr = a1;
                                          } else {
                                            tile_store_rows8(sW, a0, (A.w0 + (q - 1) * 64) + row0 * 64 * R, cnt, 64 * R, lane);
                                            tile_store_rows8(sW, a1, (A.w0 + (q - 1) * 64 + 32) + row0 * 64 * R, cnt, 64 * R, lane);
                                            float Yv[D];
                                            load_Y(Yv);
                                            tile_scal_accumulate8<q - 1>(sBv + 4 * hh, Yv, a0, a1, sc0, sc1);
                                          }
                                        });
    AA_TICK8(4)
    // ---- latent 0, hidden layer: [two-body | scal0] -> h (stored), a_0 = silu(h)
    v16f k0, k1;
    fused_layer<S_L3, NS, 4, 2>(A, p,
                                [&](auto kc) {
                                  constexpr int k = decltype(kc)::value;
                                  if constexpr (k < NPARK) {
                                    return v16f(fetch_tile(sPark + k * kTileFloats, lane));
                                  } else if constexpr (k < 2) {
                                    return tbr;
                                  } else if constexpr (k == 2) {
                                    return sc0;
                                  } else {
                                    return sc1;
                                  }
                                },
                                [&](auto, const v16f& a0, const v16f& a1) {
                                  tile_store_rows8(sW, a0, (A.lat_h0) + row0 * 64, cnt, 64, lane);
                                  tile_store_rows8(sW, a1, (A.lat_h0 + 32) + row0 * 64, cnt, 64, lane);
                                  keep_tile<true>(a0, k0);
                                  keep_tile<true>(a1, k1);
                                });
    AA_TICK8(5)
    // ---- per-atom part of layer 1: moments of a_0 -> x2s1 -> v = dSig1/dtf1 (x2s1) -> B1 = Sig0^T_x1(v, x2s0)
    {
      float wp0[Sig0::P], wp1[Sig1::P];
#pragma unroll
      for (int q = 0; q < Sig0::P; ++q) wp0[q] = load_pw(A.tpw0, Sig0::P, q);
#pragma unroll
      for (int q = 0; q < Sig1::P; ++q) wp1[q] = load_pw(A.tpw1, Sig1::P, q);
      float M[D], x2s1[D];
      tile_moments_half<D>(sW, sY, k0, k1, lane, M);
      if constexpr (PM)
        project_moments_mfma<S_P1, NS, D, R, kLdY8>(A, p, sW, M, A.sf, x2s1, sBv);
      else
        project_moments<S_P1, NS, D, R, kLdY8>(A, p, sW, M, A.sf, x2s1);
      if (atom_ok) {
#pragma unroll
        for (int j = 0; j < D; ++j) *at_bytes(A.x2s1 + (atom * D + j) * 64, 4u * unsigned(lane)) = x2s1[j];
      }
      float one[1] = {1.f}, v[D], B1[D], x2s0b[D];
#pragma unroll
      for (int j = 0; j < D; ++j) x2s0b[j] = atom_ok ? *at_bytes(A.x2s0 + (atom * D + j) * 64, 4u * unsigned(lane)) : 0.f;  // (this wave's own store of layer 0)
      Sig1::template bx1<float>(one, x2s1, wp1, v);
      Sig0::template bx1<float>(v, x2s0b, wp0, B1);
#pragma unroll
      for (int a = 0; a < D; ++a) sBv[a * 64 + lane] = B1[a];
      __builtin_amdgcn_wave_barrier();
    }
    // inputs of the next tile (its neighbor ids arrived long ago): positions, shifts, types
    load_geo(a_nxt, nxt);
    AA_TICK8(6)
    // ---- layer-1 tensor-track scalars with B1; w0 from the rows stored in the first stage (this wave's own stores, L2-resident)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      sc0[r] = 0.f;
      sc1[r] = 0.f;
    }
    {
      v16f wa = tile_load_rows8(A.w0 + row0 * (64 * R), 64 * R, lane, row_ok), wb = tile_load_rows8(A.w0 + row0 * (64 * R) + 32, 64 * R, lane, row_ok);
      static_for<0, R>([&](auto rr) {
        constexpr int r = decltype(rr)::value;
        v16f na = wa, nb = wb;
        if constexpr (r + 1 < R) {
          na = tile_load_rows8(A.w0 + row0 * (64 * R) + (r + 1) * 64, 64 * R, lane, row_ok);
          nb = tile_load_rows8(A.w0 + row0 * (64 * R) + (r + 1) * 64 + 32, 64 * R, lane, row_ok);
        }
        float Yv[D];
        load_Y(Yv);
        tile_scal_accumulate8<r>(sBv + 4 * hh, Yv, wa, wb, sc0, sc1);
        wa = na;
        wb = nb;
      });
    }
    AA_TICK8(7)
    // ---- latent 1 hidden layer (acc6) and readout hidden layer (acc8): scal1 chunks into acc6; then the chunks both layers
    //      share -- lat0 (registers), two-body scalars (parked) -- each fetched and split ONCE for both; lat1 chunks into acc8
    v16f a60, a61, a80, a81;
    v16f dk0, dk1;  // (TAIL) silu'(latent-1 pre-activation)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      a60[r] = 0.f;
      a61[r] = 0.f;
      a80[r] = 0.f;
      a81[r] = 0.f;
    }
    {
      XSplit xs[2];
      v16f t;
      xsplit_from_acc(sc0, xs[0]);
      fused_step8<S_L6A + 0, NS>(A, p, xs[0], a60, a61, [&] { xsplit_from_acc(sc1, xs[1]); });
      fused_step8<S_L6A + 1, NS>(A, p, xs[1], a60, a61, [&] { xsplit_from_acc(k0, xs[0]); });
      fused_step8<S_M + 0, NS>(A, p, xs[0], a60, a61, [&] {});
      fused_step8<S_M + 1, NS>(A, p, xs[0], a80, a81, [&] { xsplit_from_acc(k1, xs[1]); });
      fused_step8<S_M + 2, NS>(A, p, xs[1], a60, a61, [&] { t = fetch_tile(sPark, lane); });
      fused_step8<S_M + 3, NS>(A, p, xs[1], a80, a81, [&] { xsplit_from_acc(t, xs[0]); });
      fused_step8<S_M + 4, NS>(A, p, xs[0], a60, a61, [&] {
        if constexpr (NPARK == 2) t = fetch_tile(sPark + kTileFloats, lane); else t = tbr;
      });
      fused_step8<S_M + 5, NS>(A, p, xs[0], a80, a81, [&] { xsplit_from_acc(t, xs[1]); });
      fused_step8<S_M + 6, NS>(A, p, xs[1], a60, a61, [&] {});
      fused_step8<S_M + 7, NS>(A, p, xs[1], a80, a81, [&] {});
      AA_TICK8(8)
      // latent 1: pre-activation stored, a_1 = silu(h) feeds the readout (its output layer is folded: kFoldLatent)
      if constexpr (TAIL) {  // (the reverse of this layer runs below: its pre-activation is needed there, not in HBM)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          dk0[r] = dsilu(a60[r]);
          dk1[r] = dsilu(a61[r]);
        }
      } else {
        tile_store_rows8(sW, a60, (A.lat_h1) + row0 * 64, cnt, 64, lane);
        tile_store_rows8(sW, a61, (A.lat_h1 + 32) + row0 * 64, cnt, 64, lane);
      }
      keep_tile<true>(a60, k0);
      keep_tile<true>(a61, k1);
      xsplit_from_acc(k0, xs[0]);
      fused_step8<S_L8K + 0, NS>(A, p, xs[0], a80, a81, [&] { xsplit_from_acc(k1, xs[1]); });
      fused_step8<S_L8K + 1, NS>(A, p, xs[1], a80, a81, [&] {});
    }
    AA_TICK8(9)
    // ---- readout: hidden pre-activation stored; last linear layer + edge sum
    {
      if constexpr (!TAIL) {
        tile_store_rows8(sW, a80, (A.ro_h) + row0 * 64, cnt, 64, lane);
        tile_store_rows8(sW, a81, (A.ro_h + 32) + row0 * 64, cnt, 64, lane);
      }
      float part = 0.f;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const v4f w0v = *reinterpret_cast<const v4f*>(sRo + 8 * q + 4 * hh);
        const v4f w1v = *reinterpret_cast<const v4f*>(sRo + 32 + 8 * q + 4 * hh);
#pragma unroll
        for (int i = 0; i < 4; ++i) part += silu(a80[4 * q + i]) * w0v[i] + silu(a81[4 * q + i]) * w1v[i];
      }
      float tot = row_ok ? part : 0.f;
#pragma unroll
      for (int m = 32; m >= 1; m >>= 1) tot += __shfl_xor(tot, m);
      if (atom_ok && lane == 0) {
        float en = tot * A.ro_factor;
        const int t = A.types[atom];
        if (A.scales) en *= A.scales[t];
        if (A.shifts) en += A.shifts[t];
        if (cnt < 0) en = __builtin_nanf("");  // (segment beyond the max_degree hint)
        A.atom_energy[atom] = en;
      }
    }
    if constexpr (TAIL) {
      // ---- readout-reverse chain (see TAIL): d ro_h in place of ro_h
      float rofac = A.ro_factor;
      if (A.scales && atom_ok) rofac *= A.scales[A.types[atom]];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const v4f w0v = *reinterpret_cast<const v4f*>(sRo + 8 * q + 4 * hh);
        const v4f w1v = *reinterpret_cast<const v4f*>(sRo + 32 + 8 * q + 4 * hh);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          a80[4 * q + i] = rofac * w0v[i] * dsilu(a80[4 * q + i]);
          a81[4 * q + i] = rofac * w1v[i] * dsilu(a81[4 * q + i]);
        }
      }
      XSplit xd[2], xh[2];
      xsplit_from_acc(a80, xd[0]);
      xsplit_from_acc(a81, xd[1]);
      v16f t0, t1;
      auto zero = [&] {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          t0[r] = 0.f;
          t1[r] = 0.f;
        }
      };
      zero();  // d h1 = (d ro_h @ W_a) * silu'(h1)
      fused_step8<S_T + 0, NS>(A, p, xd[0], t0, t1, [&] {});
      fused_step8<S_T + 1, NS>(A, p, xd[1], t0, t1, [&] {});
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        t0[r] *= dk0[r];
        t1[r] *= dk1[r];
      }
      xsplit_from_acc(t0, xh[0]);
      xsplit_from_acc(t1, xh[1]);
      static_for<0, 2>([&](auto pp) {  // d EDGE_FEATURES[:, 64 pp .. 64 pp + 63] = [d ro_h | d h1] @ W_b
        constexpr int pr = decltype(pp)::value;
        zero();
        fused_step8<S_T + 2 + 4 * pr + 0, NS>(A, p, xd[0], t0, t1, [&] {});
        fused_step8<S_T + 2 + 4 * pr + 1, NS>(A, p, xd[1], t0, t1, [&] {});
        fused_step8<S_T + 2 + 4 * pr + 2, NS>(A, p, xh[0], t0, t1, [&] {});
        fused_step8<S_T + 2 + 4 * pr + 3, NS>(A, p, xh[1], t0, t1, [&] {});
        tile_store_rows8(sW, t0, (A.g_fcat + 64 * pr) + row0 * A.ld_gfcat, cnt, unsigned(A.ld_gfcat), lane);
        tile_store_rows8(sW, t1, (A.g_fcat + 64 * pr + 32) + row0 * A.ld_gfcat, cnt, unsigned(A.ld_gfcat), lane);
      });
      zero();  // d scal1 = d h1 @ W_c
      fused_step8<S_T + 10, NS>(A, p, xh[0], t0, t1, [&] {});
      fused_step8<S_T + 11, NS>(A, p, xh[1], t0, t1, [&] {});
      tile_store_rows8(sW, t0, (A.g_scal1) + row0 * 64, cnt, 64, lane);
      tile_store_rows8(sW, t1, (A.g_scal1 + 32) + row0 * 64, cnt, 64, lane);
    }
    AA_TICK8(10)
    cur.beg = nxt.beg;
    cur.cnt = nxt.cnt;
    cur.j = nxt.j;
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      cur.pi[q] = nxt.pi[q];
      cur.pj[q] = nxt.pj[q];
      cur.sv[q] = nxt.sv[q];
    }
    cur.ti = nxt.ti;
    cur.tj = nxt.tj;
    nxt.beg = beg2;
    nxt.cnt = cnt2;
    a_cur = a_nxt;
    a_nxt = a_nn;
    a_nn = a_n3;
  }
}

size_t fused_fwd8_lds_bytes(int num_types, int waves) {
  return sizeof(u32x4) * 2 * kWStep + sizeof(float) * (64 + 32 + size_t(num_types) * num_types * 512 + size_t(waves) * fused_wave_floats(waves));
}
int fused_fwd8_num_steps(int R, bool proj_mfma, bool tail) { return fused_fwd8_steps(R, proj_mfma, tail); }

int launch_fused_fwd8(int pair, int waves, const FusedFwdArgs& a, hipStream_t stream) {
  if (a.atom_end <= a.atom0) return AA_OK;
  if (waves != 4 && waves != 8) return fail(AA_ERR_INVALID, "fused forward (wide): 4 or 8 waves per workgroup");
  const size_t smem = fused_fwd8_lds_bytes(a.num_types, waves);
  if (smem * (waves == 4 ? 2 : 1) > 160 * 1024) return fail(AA_ERR_INVALID, "fused forward (wide): LDS budget exceeded");
  if (!a.w0) return fail(AA_ERR_INVALID, "fused forward (wide): needs the w0 rows");
  static int num_cu = 0;
  if (num_cu == 0) {
    int dev = 0, n = 0;
    AA_CHECK_HIP(hipGetDevice(&dev));
    AA_CHECK_HIP(hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev));
    num_cu = n > 0 ? n : 256;
  }
  const int64_t ngroups = (a.atom_end - a.atom0 + waves - 1) / waves;
  dim3 grid((unsigned)std::min<int64_t>(ngroups, int64_t(num_cu) * ((waves == 4 && !a.wide_one_per_cu) ? 2 : 1)));
  // (the tail exists for the eight-wave form: with four-wave workgroups -- 24 instead of 16 staging registers, one two-body tile in
  //  registers -- it spills 186 registers and the kernel takes 5.1-5.3 instead of 4.3 ms at C4, profiles/r06_v19_*)
  if (a.tail && (waves != 8 || a.wide_proj_mfma || !a.g_fcat || !a.g_scal1)) return fail(AA_ERR_INVALID, "fused forward (wide): the reverse tail exists for the eight-wave form with vector projections");
#define AA_FUSED8_LAUNCHT(S0_, S1_)                                                                                       \
  {                                                                                                                        \
    const void* fn = (const void*)fused_fwd8_kernel<cg::S0_, cg::S1_, 8, false, true>;                                     \
    AA_CHECK_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, int(smem)));                          \
    hipLaunchKernelGGL((fused_fwd8_kernel<cg::S0_, cg::S1_, 8, false, true>), grid, dim3(512), smem, stream, a);           \
  }
#define AA_FUSED8_LAUNCH1(S0_, S1_, W_, P_)                                                                     \
  {                                                                                                           \
    const void* fn = (const void*)fused_fwd8_kernel<cg::S0_, cg::S1_, W_, P_>;                                \
    AA_CHECK_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, int(smem)));             \
    hipLaunchKernelGGL((fused_fwd8_kernel<cg::S0_, cg::S1_, W_, P_>), grid, dim3(64 * W_), smem, stream, a);  \
  }
#define AA_FUSED8_LAUNCH(S0_, S1_) \
  if (a.tail) AA_FUSED8_LAUNCHT(S0_, S1_) else if (waves == 8 && a.wide_proj_mfma) AA_FUSED8_LAUNCH1(S0_, S1_, 8, true) else if (waves == 8) AA_FUSED8_LAUNCH1(S0_, S1_, 8, false) \
  else if (a.wide_proj_mfma) AA_FUSED8_LAUNCH1(S0_, S1_, 4, true) else AA_FUSED8_LAUNCH1(S0_, S1_, 4, false)
  if (pair == 0) {
    AA_FUSED8_LAUNCH(Sig1, Sig0)
  } else if (pair == 1) {
    AA_FUSED8_LAUNCH(Sig5, Sig4)
  } else {
    return fail(AA_ERR_INVALID, "fused forward (wide): unsupported signature pair");
  }
#undef AA_FUSED8_LAUNCH1
#undef AA_FUSED8_LAUNCHT
#undef AA_FUSED8_LAUNCH
  AA_CHECK_HIP(hipGetLastError());
#ifdef AA_FUSED_TIMING
  {
    static int calls = 0;
    if (++calls == 8) {
      unsigned long long t[32];
      AA_CHECK_HIP(hipStreamSynchronize(stream));
      AA_CHECK_HIP(hipMemcpyFromSymbol(t, HIP_SYMBOL(g_fused8_ticks), sizeof(t)));
      static const char* nm[10] = {"geometry", "table+stores", "TPA0", "first stage", "latent0", "TPA1", "scal1", "lat1+ro merged", "lat1 epi + ro lat1", "readout epi"};
      for (int i = 0; i < 10; ++i) fprintf(stderr, "[fused8 timing] %-20s %8llu cycles\n", nm[i], t[i + 1] - t[i]);
      fprintf(stderr, "[fused8 timing] %-20s %8llu cycles\n", "total", t[10] - t[0]);
    }
  }
#endif
  return AA_OK;
}

}  // namespace aa
