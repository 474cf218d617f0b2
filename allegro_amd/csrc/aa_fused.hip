// Fused per-atom-tile kernels of the standard 2-layer Allegro stack (gfx950, fp32 via bf16x3 MFMA).
//
// Strict locality (the reference's own model test, tests/model/test_allegro.py:68-70) makes every dependency of the
// forward pass local to one center atom: its edge segment and per-atom sums over that segment.  One WAVE therefore
// owns one center atom's edge tile (<= 32 edges = one 32-column MFMA tile) and runs the WHOLE module chain of
// allegro/model/allegro_models.py:222-297 on it with every activation in registers / LDS:
//
//   geometry, spherical harmonics, Bessel x cutoff x type-pair embedding   (tensorembed.py:86-92, scalarembed.py:60-81)
//   scalar_embed_mlp  ->  env_embed_linear | first_layer projection         (allegro_models.py:173-183, _allegro.py:251)
//   layer 0: moments -> x2s0 -> per-atom vector B0 -> scal0 -> latent 0      (_channels.py:44-57, _contract.py:185-251)
//   layer 1: moments -> x2s1 -> per-atom vector B1 -> scal1 -> latent 1      (_allegro.py:263-294)
//   edge readout + EdgewiseReduce + PerTypeScaleShift                        (allegro_models.py:231-260, edgewise.py:40-60)
//
// What goes to HBM is only what the reverse pass reads again (pre-activations of the four hidden layers, the edge
// embedding, x2s per atom) -- the 9 stage boundaries of the staged pipeline (aa_model.hip) disappear.
//
// Layouts.  Per-edge activations live in the MFMA accumulator layout of the swapped-operand GEMM (aa_gemm.hip):
// lane = (edge el = lane & 31, half hh = lane >> 5); a 32-feature tile is 16 registers, register s holds feature
// 8 (s >> 2) + 4 hh + (s & 3).  That IS the B operand of the next layer (weights pre-permuted on the host), so linear
// layers chain with no data movement.  Per-atom quantities (moments M[j][k], x2s[j][ch], the Clebsch-Gordan vectors
// B[a][ch]) live in the transposed view lane = k / channel; the two views exchange through a wave-private LDS patch:
//   moments   M[j][k] = sum_e Y[e][j] a[e][k]      tile -> LDS [e][k] -> lane k walks the 32 rows
//   scalars   scal[e][ch] = sum_r w0[e][r][ch] * (sum_{a in r} Y[e][a] B[a][ch])      B from LDS (broadcast reads),
//             evaluated in the epilogue of the GEMM tile pair that produces w0[.][r][.] -- w0 is never stored for
//             the forward's own use and is recomputed (6 MFMA steps) for the second layer instead of being held in
//             96 registers or re-read from HBM.
// Weights stream L2 -> LDS once per workgroup (4 waves = 4 atoms share every 12-KB step, double buffered), exactly
// the staging of gemm_chain_bf16x3_kernel.
//
// Segments longer than one tile (TEAMS = true, aa_graph.max_degree in 33..128): an atom with up to 64 / 128 edges is
// owned by a TEAM of 2 / 4 waves of one workgroup, wave t of the team running edges [32 t, 32 t + 32) of the segment.
// Everything per edge is unchanged; the two per-atom sums (x2s of either layer, linear in the moments) and the atom's
// energy are completed across the team through LDS in team order (bit-reproducible), every member then derives the
// per-atom vectors B from the complete x2s.  Atoms are dealt to workgroups by class (4 / 2 / 1 tiles: one / two / four
// atoms per workgroup) from three lists that fused_classify_kernel fills with atomic counters -- which atom lands in which
// slot varies from run to run, no result depends on it.
//
// Limits: every center atom of the block has <= 128 edges (larger segments run the staged pipeline), u = S = all MLP
// widths = 64, embedding table path (<= 3 species, 8 basis functions), l_max <= 2, fp32.
#include "aa_fused_tile.h"

namespace aa {

// ------------------------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------------------------
// Optional phase timing (tools/fused_timing.sh builds with -DAA_FUSED_TIMING): wave 0 of the middle workgroup stamps the
// shader clock at every phase boundary into a small global buffer that the launcher prints.
#ifdef AA_FUSED_TIMING
__device__ unsigned long long g_fused_ticks[32];
#define AA_TICK(i)                                                                   \
  if (blockIdx.x == gridDim.x / 2 && threadIdx.x == 0) g_fused_ticks[i] = __builtin_readcyclecounter();
#else
#define AA_TICK(i)
#endif

// number of weight-pipeline steps of the program for R irreps: what the host builds (aa_model.hip: forward_fused) and what the
// kernel consumes -- asserted against each other at compile time (a silent mismatch shifts every later layer's weights)
constexpr int fused_fwd_steps(int R, bool hold) {
  const int ps = kProjMfma ? 2 * R : 4;  // one env projection
  return (kFoldEmbed ? (kFoldEmb1 ? 0 : 2) : 4) + ps + (2 + 2 * R) + 4 + ps + (kFoldLatent ? 0 : 2) + (hold ? 0 : 2 * R) + 6 + (kFoldLatent ? 0 : 2) + 6;
}

// x[D] (lane = channel) summed over the waves [first, first + tsize) of a team, in that order (every member ends up with the
// same bits); tsize is workgroup-uniform, so the barrier is too.  The exchange area is reused by the next call only after
// dozens of weight-pipeline barriers.
template <int D>
__device__ __forceinline__ void team_sum(float* sX, int wv, int first, int tsize, int lane, float* x) {
  if (tsize == 1) return;
#pragma unroll
  for (int j = 0; j < D; ++j) sX[(wv * 16 + j) * 64 + lane] = x[j];
  lds_barrier();
#pragma unroll
  for (int j = 0; j < D; ++j) x[j] = 0.f;
  for (int m = 0; m < tsize; ++m) {
#pragma unroll
    for (int j = 0; j < D; ++j) x[j] += sX[((first + m) * 16 + j) * 64 + lane];
  }
}

// per-lane geometry inputs of one tile, fetched one iteration ahead of their use
struct TileIn {
  int beg, cnt;   // edge segment of the wave's atom (cnt = 0: nothing to do)
  int j;          // neighbor of the lane's edge
  float pi[3], pj[3], sv[3];
  int ti, tj;
};

// team exchange area behind the parked tiles: x2s partials [4 waves][D <= 16][64] + energy partials [4]
constexpr int kTeamFloats = 4 * 16 * 64 + 4;

// KEEP: tile pairs that feed several later layers and are therefore split into their bf16 levels ONCE, where they are produced,
// and held in registers as MFMA operands (48 registers per pair) instead of being parked raw in LDS and split again by every
// layer that reads them: 0 none (LDS parking), 1 the two-body scalars (read by L3, L6, L8), 2 also lat0 (L6, L8).  The
// one-tile, w0-holding instantiation has the registers to spare (366 of 512 before).
template <class Sig0, class Sig1, bool HOLD, bool TEAMS, int KEEP = 0>
__global__ __launch_bounds__(256, kFusedOcc) void fused_fwd_kernel(FusedFwdArgs A) {
  constexpr int D = Sig0::D2, R = Sig0::LMAX + 1;
  static_assert(Sig0::D1 == D && Sig0::DOUT == D && Sig1::D1 == D && Sig1::DOUT == 1, "standard 2-layer stack");
  static_assert(D <= 16, "l_max <= 3");
  // the program: L0 L1 | Wenv0 | L2 | L3 | Wenv1 | L4 | (L5: w0 again, unless held) | L6 L7 L8
  constexpr int kPS = kProjMfma ? 2 * R : 4;  // pipeline steps of one env projection
  constexpr int S_L0 = 0, S_L1 = kFoldEmbed ? 0 : 2, S_P0 = S_L1 + (kFoldEmb1 ? 0 : 2), S_L2 = S_P0 + kPS, S_L3 = S_L2 + 2 + 2 * R, S_P1 = S_L3 + 4, S_L4 = S_P1 + kPS,
                S_L5 = S_L4 + (kFoldLatent ? 0 : 2), S_L6 = S_L5 + (HOLD ? 0 : 2 * R), S_L7 = S_L6 + 6, S_L8 = S_L7 + (kFoldLatent ? 0 : 2), NS = S_L8 + 6;
  static_assert(NS % 2 == 0 && NS <= kFusedMaxSteps, "the two LDS buffers alternate consistently across iterations");
  static_assert(NS == fused_fwd_steps(R, HOLD), "kernel and host disagree about the length of the weight program");
  u32x4* wbuf = reinterpret_cast<u32x4*>(aa_smem);
  float* sRo = reinterpret_cast<float*>(wbuf + 2 * kWStep);            // [64] last readout weights
  float* sRm = sRo + 64;                                               // [16: T*T <= 9 used] 1 / r_max per type pair, then [8] Bessel roots at 16
  float* sTab = sRm + 32;                                              // [T*T][8][64] two-body table
  const int ntab = A.num_types * A.num_types * 512;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, hh = lane >> 5, el = lane & 31;
  float* sW = sTab + ntab + wv * (kWaveRegion + 32 * kLdY);            // wave region: patch / per-atom vectors ...
  float* sY = sW + kWaveRegion;                                        // ... and the harmonics of the tile [32][kLdY]
  float* sBv = sW + kOffB;
  // parked tiles of this wave: two-body scalars (2 tiles) and lat0 (2 tiles)
  float* sPark = sTab + ntab + 4 * (kWaveRegion + 32 * kLdY) + wv * 4 * kTileFloats;
  float* sTeam = sTab + ntab + 4 * (kWaveRegion + 32 * kLdY) + 4 * 4 * kTileFloats;  // (TEAMS only) [4][D][64] | [4]
#define AA_PARK(i, t) park_tile(sPark + (i) * kTileFloats, t, lane)
#define AA_FETCH(i) fetch_tile(sPark + (i) * kTileFloats, lane)
  for (int i = tid; i < 64; i += 256) sRo[i] = A.ro_w[i];
  if (tid < A.num_types * A.num_types) sRm[tid] = A.rmax_recip[tid];
  if (tid >= 16 && tid < 24) sRm[tid] = A.embed_kind == 0 ? A.bessel_w[tid - 16] : 0.f;
  for (int i = tid; i < ntab; i += 256) sTab[i] = A.emb_tab[i];
  FusedPipe p;
  p.wbuf = wbuf;
  p.tid = tid;
  p.lane = lane;
  p.stager = true;
  p.zero = 0;
  {
    u32x4 r[3];
    pipe_load(A, tid, 0, r);
    pipe_store(wbuf, 0, tid, r);
    pipe_load(A, tid, 1, p.rb);  // (step 1 lands in LDS at the end of step 0)
  }
  // ---- persistent loop over groups of 4 waves' worth of tiles; the inputs of the next tile are fetched during the current one
  // TEAMS: groups [0, n4) hold one 4-tile atom each, [n4, n4 + g2) two 2-tile atoms, the rest four 1-tile atoms
  int n4 = 0, n2 = 0, n1 = 0;
  if constexpr (TEAMS) {
    n4 = __builtin_amdgcn_readfirstlane(A.tile_counts[0]);
    n2 = __builtin_amdgcn_readfirstlane(A.tile_counts[1]);
    n1 = __builtin_amdgcn_readfirstlane(A.tile_counts[2]);
  }
  const int64_t g2 = (int64_t(n2) + 1) / 2, g1 = (int64_t(n1) + 3) / 4;
  const int64_t ngroups = TEAMS ? int64_t(n4) + g2 + g1 : (A.atom_end - A.atom0 + 3) / 4;
  auto group_of = [&](int64_t it) { return int64_t(blockIdx.x) + it * gridDim.x; };
  // team geometry of the wave in group g: tiles per atom, this wave's tile.  Everything here is wave-uniform and kept in
  // scalar registers (the kernel has no vector register to spare).
  const int wvs = __builtin_amdgcn_readfirstlane(wv);
  auto team_size = [&](int64_t g) { return !TEAMS ? 1 : (g < n4 ? 4 : (g < n4 + g2 ? 2 : 1)); };
  auto tile_of = [&](int64_t g) { return !TEAMS ? 0 : (g < n4 ? wvs : (g < n4 + g2 ? (wvs & 1) : 0)); };
  auto atom_of = [&](int64_t it) -> int64_t {
    const int64_t g = group_of(it);
    if constexpr (!TEAMS) {
      return A.atom0 + g * 4 + wv;
    } else {
      int at = int(A.atom_end);
      if (g < ngroups) {
        if (g < n4) {
          at = A.tile_atoms[g];
        } else if (g < n4 + g2) {
          const int64_t i = (g - n4) * 2 + (wvs >> 1);
          if (i < n2) at = A.tile_atoms[A.tile_cap + i];
        } else {
          const int64_t i = (g - n4 - g2) * 4 + wvs;
          if (i < n1) at = A.tile_atoms[2 * A.tile_cap + i];
        }
      }
      return __builtin_amdgcn_readfirstlane(at);
    }
  };
  auto load_rows = [&](int64_t atom, int tile, int& beg, int& cnt) {
    beg = 0;
    cnt = 0;
    if (atom < A.atom_end) {
      const int b0 = A.rowptr[atom], deg = A.rowptr[atom + 1] - b0;
      beg = b0 + 32 * tile;
      cnt = deg - 32 * tile;
      cnt = cnt < 0 ? 0 : (cnt > 32 ? 32 : cnt);
      // A segment longer than this instantiation covers means the caller's aa_graph.max_degree hint was stale.  Never a
      // silent truncation: the atom is skipped (cnt < 0: every row masked), its energy becomes NaN and the plan's status
      // word makes the next aa_model_energy_forces / aa_model_check fail (the reference takes any segment length,
      // allegro/nn/_strided/_contract.py:195-205 -- the host then has to pass the true degree and gets the staged pipeline).
      if (!TEAMS && A.skip_long && deg > 32 && deg <= kFusedMaxDegree) {
        // mixed form (launch_fused_fwd): the few atoms with more than one tile of edges belong to the team pass that follows on
        // the same stream; this pass leaves every row and every per-atom result of theirs alone
        cnt = -2;
      } else if (deg > (TEAMS ? kFusedMaxDegree : 32)) {
        cnt = -1;
        if (A.status && lane == 0) *reinterpret_cast<volatile int32_t*>(A.status) = deg;
        // ... and the FORCES of this step must not look valid either (none of this atom's workspace rows are written,
        // the reverse pass would combine leftovers of earlier steps into finite numbers): the direction rows of this tile
        // become NaN -- edge_backward multiplies them into dE/dr of these edges, so the skipped atom's and its neighbours'
        // forces (and the virial) are NaN in the same step
        if (hh == 0 && 32 * tile + el < deg) *reinterpret_cast<v4f*>(A.vec + 4 * (int64_t(b0) + 32 * tile + el)) = v4f{__builtin_nanf(""), __builtin_nanf(""), __builtin_nanf(""), __builtin_nanf("")};
      }
    }
  };
  auto load_nbr = [&](int beg, int cnt) { return el < cnt ? A.nbr[int64_t(beg) + el] : 0; };
  auto load_geo = [&](int64_t atom, TileIn& t) {
    if (el < t.cnt) {
      const int64_t e = int64_t(t.beg) + el;
      const float* pi = A.pos + 3 * atom;
      const float* pj = A.pos + 3 * int64_t(t.j);
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        t.pi[q] = pi[q];
        t.pj[q] = pj[q];
        t.sv[q] = A.shift_vec ? A.shift_vec[3 * e + q] : 0.f;
      }
      t.ti = A.types[atom];
      t.tj = A.types[t.j];
    }
  };
  // atom ids travel three iterations ahead (TEAMS: they come from the class lists), row pointers two, neighbor ids and geometry one
  int64_t a_cur = atom_of(0), a_nxt = atom_of(1), a_nn = atom_of(2);
  TileIn cur, nxt;
  int beg2 = 0, cnt2 = 0;
  load_rows(a_cur, tile_of(group_of(0)), cur.beg, cur.cnt);
  load_rows(a_nxt, tile_of(group_of(1)), nxt.beg, nxt.cnt);
  cur.j = load_nbr(cur.beg, cur.cnt);
  load_geo(a_cur, cur);
  float wp0[Sig0::P], wp1[Sig1::P];
#pragma unroll
  for (int q = 0; q < Sig0::P; ++q) wp0[q] = A.coupling ? A.tpw0[lane * Sig0::P + q] : A.tpw0[q];
#pragma unroll
  for (int q = 0; q < Sig1::P; ++q) wp1[q] = A.coupling ? A.tpw1[lane * Sig1::P + q] : A.tpw1[q];
  lds_barrier();  // tables + first weight step staged
  for (int64_t it = 0; group_of(it) < ngroups; ++it) {
    opaque_scalar(p.zero);  // (see pipe_load)
    AA_TICK(0)
    const int64_t atom = a_cur;
    const int tsize = team_size(group_of(it)), tile = tile_of(group_of(it)), tfirst = wvs - tile;  // (tsize: workgroup-uniform)
    const bool leader = tile == 0;
    const int64_t a_n3 = atom_of(it + 3);
    const int beg = __builtin_amdgcn_readfirstlane(cur.beg), cnt = __builtin_amdgcn_readfirstlane(cur.cnt);
    const bool atom_ok = atom < A.atom_end && cnt != -2;  // (-2: an atom of the other pass of the mixed form)
    const bool row_ok = el < cnt;
    const int64_t row0 = beg;
    // prefetch: neighbor ids of the next tile, row pointers of the one after
    nxt.j = load_nbr(nxt.beg, nxt.cnt);
    load_rows(a_nn, tile_of(group_of(it + 2)), beg2, cnt2);
    // ---- geometry of the lane's edge (rows beyond the segment: a harmless dummy that is masked everywhere)
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
        const int64_t e = row0 + el;
        *reinterpret_cast<v4f*>(A.vec + 4 * e) = v4f{nx, ny, nz, rr};
        if (A.sh) {
#pragma unroll
          for (int m = 0; m < D; ++m) A.sh[e * D + m] = Yf[m];
        }
      }
      if (hh == 0) {
#pragma unroll
        for (int m = 0; m < kLdY; ++m) sY[el * kLdY + m] = m < D ? Y[m] : 0.f;
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
    AA_TICK(1)
    // ---- two-body embedding of the lane's 32 features: emb0[c] = sum_n basis[n] * tab[pair][n][c]
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
      }
    }
    v16f k0, k1, sc0, sc1;
    v16f w0t[HOLD ? 2 * R : 1];
    XSplit tb[KEEP >= 1 ? 2 : 1], l0x[KEEP >= 2 ? 2 : 1];  // held split tiles (see KEEP)
    AA_TICK(2)
    // ---- L0: scalar_embed_mlp layer 0 (pre-activation kept for the reverse pass).  Folded into the table (kFoldEmbed): em0 / em1
    //      ARE its pre-activation, no GEMM
    if constexpr (kFoldEmbed) {
      tile_store_rows(sW, em0, A.se_h, row0, cnt, 64, lane);
      tile_store_rows(sW, em1, A.se_h + 32, row0, cnt, 64, lane);
      keep_tile<true>(em0, k0);
      keep_tile<true>(em1, k1);
    } else {
      fused_layer<S_L0, NS, 2, 2>(A, p,
                                  [&](auto kc) -> const v16f& { if constexpr (decltype(kc)::value == 0) return em0; else return em1; },
                                  [&](auto, const v16f& a0, const v16f& a1) {
                                    tile_store_rows(sW, a0, A.se_h, row0, cnt, 64, lane);
                                    tile_store_rows(sW, a1, A.se_h + 32, row0, cnt, 64, lane);
                                    keep_tile<true>(a0, k0);
                                    keep_tile<true>(a1, k1);
                                  });
    }
    AA_TICK(3)
    // ---- L1: scalar_embed_mlp layer 1 -> EDGE_EMBEDDING.  Folded (kFoldEmb1): every consumer of the embedding is linear, so they
    //      take a_e = silu(h) (k0, k1) against W1-folded weights; a_e is stored in the embedding's slot for the reverse pass
    if constexpr (kFoldEmb1) {
      em0 = k0;
      em1 = k1;
      tile_store_rows(sW, em0, A.emb, row0, cnt, 64, lane);
      tile_store_rows(sW, em1, A.emb + 32, row0, cnt, 64, lane);
    } else {
      fused_layer<S_L1, NS, 2, 2>(A, p,
                                  [&](auto kc) -> const v16f& { if constexpr (decltype(kc)::value == 0) return k0; else return k1; },
                                  [&](auto, const v16f& a0, const v16f& a1) {
                                    tile_store_rows(sW, a0, A.emb, row0, cnt, 64, lane);
                                    tile_store_rows(sW, a1, A.emb + 32, row0, cnt, 64, lane);
                                    em0 = a0;
                                    em1 = a1;
                                  });
    }
    AA_TICK(4)
    // ---- per-atom part of layer 0: moments of the embedding -> x2s0 -> B0 = Sig0^T_x1(e_0, x2s0)
    float x2s0[D];
    {
      float M[D];
      tile_moments<D>(sW, sY, em0, em1, lane, M);
      if constexpr (kProjMfma)
        project_moments_mfma<S_P0, NS, D, R>(A, p, sW, M, A.sf, x2s0);
      else
        project_moments<S_P0, NS, D, R>(A, p, sW, M, A.sf, x2s0);
      if constexpr (TEAMS) team_sum<D>(sTeam, wvs, tfirst, tsize, lane, x2s0);
      if (atom_ok && leader) {
#pragma unroll
        for (int j = 0; j < D; ++j) A.x2s0[(atom * D + j) * 64 + lane] = x2s0[j];
      }
      float e0[D], B0[D];
#pragma unroll
      for (int k = 0; k < D; ++k) e0[k] = k == 0 ? 1.f : 0.f;
      Sig0::template bx1<float>(e0, x2s0, wp0, B0);
#pragma unroll
      for (int a = 0; a < D; ++a) sBv[a * 64 + lane] = B0[a];
      __builtin_amdgcn_wave_barrier();
    }
    AA_TICK(5)
    // ---- L2: [two-body scalars | w0 irrep 0 | irrep 1 | ...] = emb @ [first_proj[:, :S] | env_embed_linear]; the
    //          layer-0 tensor-track scalars are accumulated as each irrep's tile pair comes out
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
                                            if constexpr (KEEP >= 1) {
                                              xsplit_from_acc(a0, tb[0]);
                                              xsplit_from_acc(a1, tb[1]);
                                            } else {
                                              AA_PARK(0, a0);
                                              AA_PARK(1, a1);
                                            }
                                            if (A.fcat) {
                                              tile_store_rows(sW, a0, A.fcat, row0, cnt, 192, lane);
                                              tile_store_rows(sW, a1, A.fcat + 32, row0, cnt, 192, lane);
                                            }
                                          } else {
                                            if (A.w0) {
                                              tile_store_rows(sW, a0, A.w0 + (q - 1) * 64, row0, cnt, 64 * R, lane);
                                              tile_store_rows(sW, a1, A.w0 + (q - 1) * 64 + 32, row0, cnt, 64 * R, lane);
                                            }
                                            tile_scal_accumulate<q - 1>(sBv + 4 * hh, Y, a0, a1, sc0, sc1);
                                            if constexpr (HOLD) {
                                              w0t[2 * (q - 1)] = a0;
                                              w0t[2 * (q - 1) + 1] = a1;
                                            }
                                          }
                                        });
    AA_TICK(6)
    // ---- L3: latent 0, hidden layer: [two-body | scal0] -> h (pre-activation stored), a1 = silu(h)
    fused_layer<S_L3, NS, 4, 2>(A, p,
                                [&](auto kc) {
                                  constexpr int k = decltype(kc)::value;
                                  if constexpr (k < 2) {
                                    if constexpr (KEEP >= 1) return tb[k]; else return v16f(AA_FETCH(k));
                                  } else if constexpr (k == 2) {
                                    return sc0;
                                  } else {
                                    return sc1;
                                  }
                                },
                                [&](auto, const v16f& a0, const v16f& a1) {
                                  tile_store_rows(sW, a0, A.lat_h0, row0, cnt, 64, lane);
                                  tile_store_rows(sW, a1, A.lat_h0 + 32, row0, cnt, 64, lane);
                                  keep_tile<true>(a0, k0);
                                  keep_tile<true>(a1, k1);
                                });
    AA_TICK(7)
    // ---- per-atom part of layer 1: moments of a1 -> x2s1 -> v = dSig1/dtf1 (x2s1) -> B1 = Sig0^T_x1(v, x2s0)
    {
      float M[D], x2s1[D];
      tile_moments<D>(sW, sY, k0, k1, lane, M);
      if constexpr (kProjMfma)
        project_moments_mfma<S_P1, NS, D, R>(A, p, sW, M, A.sf, x2s1);
      else
        project_moments<S_P1, NS, D, R>(A, p, sW, M, A.sf, x2s1);
      if constexpr (TEAMS) team_sum<D>(sTeam, wvs, tfirst, tsize, lane, x2s1);
      if (atom_ok && leader) {
#pragma unroll
        for (int j = 0; j < D; ++j) A.x2s1[(atom * D + j) * 64 + lane] = x2s1[j];
      }
      float one[1] = {1.f}, v[D], B1[D];
      Sig1::template bx1<float>(one, x2s1, wp1, v);
      Sig0::template bx1<float>(v, x2s0, wp0, B1);
#pragma unroll
      for (int a = 0; a < D; ++a) sBv[a * 64 + lane] = B1[a];
      __builtin_amdgcn_wave_barrier();
    }
    AA_TICK(8)
    // ---- L4: latent 0, output layer -> lat0.  Folded (kFoldLatent): lat0 is never formed -- its consumers L6 / L8 take the hidden
    //      activation a1 = silu(h) (k0, k1) against Wout_0 @ their lat0 row blocks; a1 is what is held / parked instead
    if constexpr (kFoldLatent) {
      if constexpr (KEEP >= 2) {
        xsplit_from_acc(k0, l0x[0]);
        xsplit_from_acc(k1, l0x[1]);
      } else {
        AA_PARK(2, k0);
        AA_PARK(3, k1);
      }
    } else {
      fused_layer<S_L4, NS, 2, 2>(A, p,
                                  [&](auto kc) -> const v16f& { if constexpr (decltype(kc)::value == 0) return k0; else return k1; },
                                  [&](auto, const v16f& a0, const v16f& a1) {
                                    if constexpr (KEEP >= 2) {
                                      xsplit_from_acc(a0, l0x[0]);
                                      xsplit_from_acc(a1, l0x[1]);
                                    } else {
                                      AA_PARK(2, a0);
                                      AA_PARK(3, a1);
                                    }
                                    if (A.fcat) {
                                      tile_store_rows(sW, a0, A.fcat + 64, row0, cnt, 192, lane);
                                      tile_store_rows(sW, a1, A.fcat + 96, row0, cnt, 192, lane);
                                    }
                                  });
    }
    // inputs of the next tile (its neighbor ids arrived long ago): positions, shifts, types
    load_geo(a_nxt, nxt);
    AA_TICK(9)
    // ---- L5: layer-1 scalars with B1 -- from the held w0 tiles, or from w0 recomputed out of the embedding
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      sc0[r] = 0.f;
      sc1[r] = 0.f;
    }
    if constexpr (HOLD) {
      static_for<0, R>([&](auto rr) {
        tile_scal_accumulate<decltype(rr)::value>(sBv + 4 * hh, Y, w0t[2 * decltype(rr)::value], w0t[2 * decltype(rr)::value + 1], sc0, sc1);
      });
    } else {
      fused_layer<S_L5, NS, 2, 2 * R>(A, p,
                                      [&](auto kc) -> const v16f& { if constexpr (decltype(kc)::value == 0) return em0; else return em1; },
                                      [&](auto ntp, const v16f& a0, const v16f& a1) {
                                        tile_scal_accumulate<decltype(ntp)::value>(sBv + 4 * hh, Y, a0, a1, sc0, sc1);
                                      });
    }
    AA_TICK(10)
    // ---- L6: latent 1, hidden layer: [two-body | lat0 | scal1]
    fused_layer<S_L6, NS, 6, 2>(A, p,
                                [&](auto kc) {
                                  constexpr int k = decltype(kc)::value;
                                  if constexpr (k < 2) {
                                    if constexpr (KEEP >= 1) return tb[k]; else return v16f(AA_FETCH(k));
                                  } else if constexpr (k < 4) {
                                    if constexpr (KEEP >= 2) return l0x[k - 2]; else return v16f(AA_FETCH(k));
                                  } else if constexpr (k == 4) {
                                    return sc0;
                                  } else {
                                    return sc1;
                                  }
                                },
                                [&](auto, const v16f& a0, const v16f& a1) {
                                  tile_store_rows(sW, a0, A.lat_h1, row0, cnt, 64, lane);
                                  tile_store_rows(sW, a1, A.lat_h1 + 32, row0, cnt, 64, lane);
                                  keep_tile<true>(a0, k0);
                                  keep_tile<true>(a1, k1);
                                });
    AA_TICK(11)
    // ---- L7: latent 1, output layer -> lat1.  Folded (kFoldLatent): the readout takes silu(h) of latent 1 (k0, k1) directly
    if constexpr (!kFoldLatent) {
      fused_layer<S_L7, NS, 2, 2>(A, p,
                                  [&](auto kc) -> const v16f& { if constexpr (decltype(kc)::value == 0) return k0; else return k1; },
                                  [&](auto, const v16f& a0, const v16f& a1) {
                                    k0 = a0;
                                    k1 = a1;
                                    if (A.fcat) {
                                      tile_store_rows(sW, a0, A.fcat + 128, row0, cnt, 192, lane);
                                      tile_store_rows(sW, a1, A.fcat + 160, row0, cnt, 192, lane);
                                    }
                                  });
    }
    AA_TICK(12)
    // ---- L8: edge readout hidden layer on [two-body | lat0 | lat1]; last linear layer + edge sum in the epilogue
    fused_layer<S_L8, NS, 6, 2>(A, p,
                                [&](auto kc) {
                                  constexpr int k = decltype(kc)::value;
                                  if constexpr (k < 2) {
                                    if constexpr (KEEP >= 1) return tb[k]; else return v16f(AA_FETCH(k));
                                  } else if constexpr (k < 4) {
                                    if constexpr (KEEP >= 2) return l0x[k - 2]; else return v16f(AA_FETCH(k));
                                  } else if constexpr (k == 4) {
                                    return k0;
                                  } else {
                                    return k1;
                                  }
                                },
                                [&](auto, const v16f& a0, const v16f& a1) {
                                  tile_store_rows(sW, a0, A.ro_h, row0, cnt, 64, lane);
                                  tile_store_rows(sW, a1, A.ro_h + 32, row0, cnt, 64, lane);
                                  float part = 0.f;
#pragma unroll
                                  for (int q = 0; q < 4; ++q) {
                                    const v4f w0v = *reinterpret_cast<const v4f*>(sRo + 8 * q + 4 * hh);
                                    const v4f w1v = *reinterpret_cast<const v4f*>(sRo + 32 + 8 * q + 4 * hh);
#pragma unroll
                                    for (int i = 0; i < 4; ++i) part += silu(a0[4 * q + i]) * w0v[i] + silu(a1[4 * q + i]) * w1v[i];
                                  }
                                  // E_i = scale_t * factor * sum over the rows of the segment (each lane half holds half a row) + shift_t
                                  float tot = row_ok ? part : 0.f;
#pragma unroll
                                  for (int m = 32; m >= 1; m >>= 1) tot += __shfl_xor(tot, m);
                                  if constexpr (TEAMS) {
                                    if (tsize > 1) {  // (workgroup-uniform) the team's tiles, added in tile order
                                      float* sE = sTeam + 4 * 16 * 64;
                                      if (lane == 0) sE[wvs] = tot;
                                      lds_barrier();
                                      tot = 0.f;
                                      for (int m = 0; m < tsize; ++m) tot += sE[tfirst + m];
                                    }
                                  }
                                  if (atom_ok && leader && lane == 0) {
                                    float en = tot * A.ro_factor;
                                    const int t = A.types[atom];
                                    if (A.scales) en *= A.scales[t];
                                    if (A.shifts) en += A.shifts[t];
                                    if (cnt < 0) en = __builtin_nanf("");  // (segment beyond the max_degree hint: see load_rows)
                                    A.atom_energy[atom] = en;
                                  }
                                });
    AA_TICK(13)
    // rotate the prefetched inputs
    cur = nxt;
    nxt.beg = beg2;
    nxt.cnt = cnt2;
    a_cur = a_nxt;
    a_nxt = a_nn;
    a_nn = a_n3;
  }
}

// the class lists of the TEAMS form: atoms with 65..128 / 33..64 / 0..32 edges -> lists 0 / 1 / 2 ([3][cap]) through atomic
// counters (zeroed by the launcher).  Slot order is arbitrary; nothing depends on it.
__global__ __launch_bounds__(256) void fused_classify_kernel(int64_t a0, int64_t a1, const int32_t* rowptr, int64_t cap, int32_t* atoms,
                                                             int32_t* counts, int min_deg) {
  const int64_t n = a0 + int64_t(blockIdx.x) * 256 + threadIdx.x;
  if (n >= a1) return;
  const int deg = rowptr[n + 1] - rowptr[n];
  if (deg <= min_deg) return;  // (mixed form: only the atoms the one-tile pass skipped)
  const int cls = deg > 64 ? 0 : (deg > 32 ? 1 : 2);
  const int idx = atomicAdd(&counts[cls], 1);
  atoms[cls * cap + idx] = int32_t(n);
}

// atoms outside the block the fused kernel covers (other ranks' blocks of an atom partition): E_i = shift_t
__global__ __launch_bounds__(256) void fused_fill_energy_kernel(int64_t N, int64_t a0, int64_t a1, const int32_t* types,
                                                                const float* shifts, float* atom_energy) {
  const int64_t n = int64_t(blockIdx.x) * 256 + threadIdx.x;
  if (n < N && (n < a0 || n >= a1)) atom_energy[n] = shifts ? shifts[types[n]] : 0.f;
}

size_t fused_fwd_lds_bytes(int num_types, bool teams) {
  return sizeof(u32x4) * 2 * kWStep +
         sizeof(float) * (64 + 32 + size_t(num_types) * num_types * 512 + 4 * (kWaveRegion + 32 * kLdY) + (kFusedOcc == 1 ? 4 * 4 * kTileFloats : 0) +
                          (teams ? kTeamFloats : 0));
}

// number of weight-pipeline steps of the program for R irreps (see the kernel)
int fused_fwd_num_steps(int R, bool hold) { return fused_fwd_steps(R, hold); }

static int launch_fused_fwd_one(int pair, bool hold_w0, const FusedFwdArgs& a, hipStream_t stream);

// Mixed form (a.mixed, with the class lists set): most atoms have one tile of edges, a FEW have more (thermal disorder pushes a
// handful of Si atoms past 32 neighbours; profiles/r05_v23_md_loop_c4.json).  The team form costs every atom ~12 % (class lists,
// exchange area, 2 fewer tile parkings), the staged pipeline ~12 % of the step: instead the one-tile kernel runs over all atoms at
// full speed and SKIPS the long ones, and a second, small launch of the team form takes exactly those (its grid is the list).
int fused_num_cus() {
  static int cus = 0;
  if (cus == 0) {
    int dev = 0, n = 0;
    cus = (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) ? n : 256;
  }
  return cus;
}

int launch_fused_fwd(int pair, bool hold_w0, const FusedFwdArgs& a, hipStream_t stream, const FusedFwdArgs* wide, bool* ran_wide) {
  if (ran_wide) *ran_wide = false;
  const bool teams = a.tile_atoms != nullptr;
  // (boxes that leave a CU at most one workgroup run one wave per SIMD whatever the kernel: there the register-rich one-tile form --
  //  w0 held, operands kept split -- is the faster one: 64 atoms 44 vs 48 us, profiles/r06_v4_ab_c2_*)
  if (wide && wide->wide_waves > 0 && a.atom_end - a.atom0 <= int64_t(4) * fused_num_cus()) wide = nullptr;
  if (wide && (!teams || a.mixed)) {
    // the one-tile pass on the eight-wave form (two waves per SIMD); the team pass over the long atoms, if any, as below
    if (a.atom_end <= a.atom0) return AA_OK;
    if (a.N > 0 && (a.atom0 > 0 || a.atom_end < a.N) && !a.fill_done) {
      hipLaunchKernelGGL(fused_fill_energy_kernel, dim3((unsigned)((a.N + 255) / 256)), dim3(256), 0, stream, a.N, a.atom0, a.atom_end,
                         a.types, a.shifts, a.atom_energy);
    }
    FusedFwdArgs one = *wide;
    one.tile_atoms = nullptr;
    one.tile_counts = nullptr;
    one.tile_cap = 0;
    one.skip_long = teams ? 1 : 0;
    if (int rc = launch_fused_fwd8(pair, wide->wide_waves == -8 ? 8 : 4, one, stream)) return rc;
    if (ran_wide) *ran_wide = true;
    if (!teams) return AA_OK;
    FusedFwdArgs team = a;
    team.long_only = 1;
    team.fill_done = 1;
    return launch_fused_fwd_one(pair, hold_w0, team, stream);
  }
  if (!a.mixed || a.tile_atoms == nullptr) return launch_fused_fwd_one(pair, hold_w0, a, stream);
  FusedFwdArgs one = a;
  one.tile_atoms = nullptr;
  one.tile_counts = nullptr;
  one.tile_cap = 0;
  one.skip_long = 1;
  if (int rc = launch_fused_fwd_one(pair, hold_w0, one, stream)) return rc;
  FusedFwdArgs team = a;
  team.long_only = 1;
  team.fill_done = 1;
  return launch_fused_fwd_one(pair, hold_w0, team, stream);
}

static int launch_fused_fwd_one(int pair, bool hold_w0, const FusedFwdArgs& a, hipStream_t stream) {
  if (a.atom_end <= a.atom0) return AA_OK;
  const bool teams = a.tile_atoms != nullptr;
  const size_t smem = fused_fwd_lds_bytes(a.num_types, teams);
  if (smem > 160 * 1024) return fail(AA_ERR_INVALID, "fused forward: LDS budget exceeded");
  if (a.N > 0 && (a.atom0 > 0 || a.atom_end < a.N) && !a.fill_done) {
    hipLaunchKernelGGL(fused_fill_energy_kernel, dim3((unsigned)((a.N + 255) / 256)), dim3(256), 0, stream, a.N, a.atom0, a.atom_end,
                       a.types, a.shifts, a.atom_energy);
  }
  // persistent: one workgroup per CU (the kernel needs the whole register file and most of the LDS of a CU)
  static int num_cu = 0;
  if (num_cu == 0) {
    int dev = 0, n = 0;
    AA_CHECK_HIP(hipGetDevice(&dev));
    AA_CHECK_HIP(hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev));
    num_cu = n > 0 ? n : 256;
  }
  // TEAMS: the number of groups is only known on the device (class counters): at most one group per atom
  const int64_t ngroups = teams ? a.atom_end - a.atom0 : (a.atom_end - a.atom0 + 3) / 4;
  dim3 grid((unsigned)std::min<int64_t>(ngroups, int64_t(num_cu) * kFusedOcc));
  if (teams) {
    AA_CHECK_HIP(hipMemsetAsync(a.tile_counts, 0, 4 * sizeof(int32_t), stream));
    hipLaunchKernelGGL(fused_classify_kernel, dim3((unsigned)((a.atom_end - a.atom0 + 255) / 256)), dim3(256), 0, stream, a.atom0, a.atom_end,
                       a.rowptr, a.tile_cap, a.tile_atoms, a.tile_counts, a.long_only ? 32 : -1);
  }
#define AA_FUSED_LAUNCH1(S0_, S1_, H_, T_)                                                                     \
  {                                                                                                            \
    const void* fn = (const void*)fused_fwd_kernel<cg::S0_, cg::S1_, H_, T_>;                                  \
    AA_CHECK_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, int(smem)));              \
    hipLaunchKernelGGL((fused_fwd_kernel<cg::S0_, cg::S1_, H_, T_>), grid, dim3(256), smem, stream, a);        \
  }
#define AA_FUSED_LAUNCHK(S0_, S1_, K_)                                                                          \
  {                                                                                                            \
    const void* fn = (const void*)fused_fwd_kernel<cg::S0_, cg::S1_, true, false, K_>;                         \
    AA_CHECK_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, int(smem)));              \
    hipLaunchKernelGGL((fused_fwd_kernel<cg::S0_, cg::S1_, true, false, K_>), grid, dim3(256), smem, stream, a); \
  }
  // (the one-tile, w0-holding form keeps split tile pairs in registers: a.keep, see the kernel's KEEP)
#define AA_FUSED_LAUNCH(S0_, S1_, H_)                                                                            \
  if (teams) AA_FUSED_LAUNCH1(S0_, S1_, H_, true)                                                                \
  else if (H_ && a.keep == 1) AA_FUSED_LAUNCHK(S0_, S1_, 1)                                                      \
  else if (H_ && a.keep == 2) AA_FUSED_LAUNCHK(S0_, S1_, 2)                                                      \
  else AA_FUSED_LAUNCH1(S0_, S1_, H_, false)
  if (pair == 0) {
    if (hold_w0) AA_FUSED_LAUNCH(Sig1, Sig0, true) else AA_FUSED_LAUNCH(Sig1, Sig0, false)
  } else if (pair == 1) {
    if (hold_w0) AA_FUSED_LAUNCH(Sig5, Sig4, true) else AA_FUSED_LAUNCH(Sig5, Sig4, false)
  } else {
    return fail(AA_ERR_INVALID, "fused forward: unsupported signature pair");
  }
#undef AA_FUSED_LAUNCH
#undef AA_FUSED_LAUNCHK
#undef AA_FUSED_LAUNCH1
  AA_CHECK_HIP(hipGetLastError());
#ifdef AA_FUSED_TIMING
  {
    static int calls = 0;
    if (++calls == 8) {  // a warm call
      unsigned long long t[32];
      AA_CHECK_HIP(hipStreamSynchronize(stream));
      AA_CHECK_HIP(hipMemcpyFromSymbol(t, HIP_SYMBOL(g_fused_ticks), sizeof(t)));
      static const char* nm[13] = {"geometry", "emb0", "L0", "L1", "TPA0", "L2", "L3", "TPA1", "L4", "L5", "L6", "L7", "L8"};
      for (int i = 0; i < 13; ++i) fprintf(stderr, "[fused timing] %-16s %8llu cycles\n", nm[i], t[i + 1] - t[i]);
      fprintf(stderr, "[fused timing] %-16s %8llu cycles\n", "total", t[13] - t[0]);
    }
  }
#endif
  return AA_OK;
}

}  // namespace aa
