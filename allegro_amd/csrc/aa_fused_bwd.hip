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

constexpr int kTailWaves = 8;  // waves (= atoms) per workgroup: two per SIMD share one 24-KB weight double buffer
constexpr int kTailD = 9;      // l_max <= 2
// wave-private LDS region (floats): sY [32][kLdY] | patch [32][kLdA] (later: sG [64][kLdY] at 0, sGM [D][64] behind it) | sB0 [D][64] | sB1 [D][64]
constexpr int kTailOffP = 32 * kLdY, kTailOffGM = kTailOffP + 64 * kLdY, kTailOffB0 = kTailOffP + 32 * kLdA,
              kTailOffB1 = kTailOffB0 + kTailD * 64, kTailWaveFloats = kTailOffB1 + kTailD * 64;
static_assert(64 * kLdY + kTailD * 64 <= 32 * kLdA, "sG and sGM live inside the patch region");

// rows [row0, row0 + 32) x 32 features of a row-major [E, ld] array -> one tile in accumulator layout (zero beyond the segment)
__device__ __forceinline__ v16f ld_tile(const float* col0, int64_t row, int ld, int hh, bool ok) {
  v16f t;
#pragma unroll
  for (int r = 0; r < 16; ++r) t[r] = 0.f;
  if (ok) {
    const float* p = col0 + row * ld + 4 * hh;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const v4f v = *reinterpret_cast<const v4f*>(p + 8 * q);
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
    if ((g & 1) == 1) __builtin_amdgcn_sched_barrier(0);  // at most two groups' cells in flight
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

}  // namespace

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
  FusedPipe p;
  p.wbuf = wbuf;
  p.tid = tid;
  p.lane = lane;
  p.stager = wv < 4;
  if (p.stager) {
    u32x4 r[3];
    pipe_load(A, tid, 0, r);
    pipe_store(wbuf, 0, tid, r);
    pipe_load(A, tid, 1, p.rb);  // (step 1 lands in LDS at the end of step 0)
  }
  float wp0[Sig0::P], wp1[Sig1::P];
#pragma unroll
  for (int q = 0; q < Sig0::P; ++q) wp0[q] = A.coupling ? A.tpw0[lane * Sig0::P + q] : A.tpw0[q];
#pragma unroll
  for (int q = 0; q < Sig1::P; ++q) wp1[q] = A.coupling ? A.tpw1[lane * Sig1::P + q] : A.tpw1[q];
  lds_barrier();  // tables + first weight step staged
  const int64_t ngroups = (A.atom_end - A.atom0 + kTailWaves - 1) / kTailWaves;
  for (int64_t it = 0; blockIdx.x + it * gridDim.x < ngroups; ++it) {
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
    const int64_t e = int64_t(beg) + el;
    // ---- geometry of the lane's edge: unit vector + length as the forward left them; harmonics re-evaluated from it
    float Y[D], nx = 1.f, ny = 0.f, nz = 0.f, rr = 1.f;
    int pair = 0, nbr = 0;
    if (row_ok) {
      const v4f vv4 = *reinterpret_cast<const v4f*>(A.vec + 4 * e);
      nx = vv4[0];
      ny = vv4[1];
      nz = vv4[2];
      rr = vv4[3];
      nbr = A.nbr[e];
      pair = A.types[atom] * A.num_types + A.types[nbr];
    }
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
    float vv[D];
    {
      float x2s0[D], x2s1[D], B0[D], B1[D], e0[D], one[1] = {1.f};
#pragma unroll
      for (int j = 0; j < D; ++j) {
        x2s0[j] = atom_ok ? A.x2s0[(atom * D + j) * 64 + lane] : 0.f;
        x2s1[j] = atom_ok ? A.x2s1[(atom * D + j) * 64 + lane] : 0.f;
        e0[j] = j == 0 ? 1.f : 0.f;
      }
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
    // ---- the two scalar-gradient rows of the lane's edge (operands of every later phase)
    const v16f g0a = ld_tile(A.gscal0, e, 64, hh, row_ok), g0b = ld_tile(A.gscal0 + 32, e, 64, hh, row_ok);
    const v16f g1a = ld_tile(A.gscal1, e, 64, hh, row_ok), g1b = ld_tile(A.gscal1 + 32, e, 64, hh, row_ok);
    // ---- per irrep: t_l = g_l * w0_r  ->  dE/dY (x1 path, per edge)  and  Q_l (per atom)
    float gy[D], Q1[D], Q0[D];
#pragma unroll
    for (int j = 0; j < D; ++j) {
      gy[j] = 0.f;
      Q1[j] = 0.f;
      Q0[j] = 0.f;
    }
    static_for<0, R>([&](auto rr_) {
      constexpr int r = decltype(rr_)::value;
      const v16f wa = ld_tile(A.w0 + r * 64, e, 64 * R, hh, row_ok), wb = ld_tile(A.w0 + r * 64 + 32, e, 64 * R, hh, row_ok);
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
    // ---- per atom: d x2s0 = Sig0^T_x2(v, Q1) + Sig0^T_x2(e_0, Q0), scaled; GM[j][k] = sum_ch d x2s0[j][ch] Wenv0^T[r(j)][ch][k]
    float gm[D];
    {
      float ga[D], gb[D], e0[D], g2[D];
#pragma unroll
      for (int j = 0; j < D; ++j) e0[j] = j == 0 ? 1.f : 0.f;
      Sig0::template bx2<float>(vv, Q1, wp0, ga);
      Sig0::template bx2<float>(e0, Q0, wp0, gb);
#pragma unroll
      for (int j = 0; j < D; ++j) g2[j] = (ga[j] + gb[j]) * A.sf;
      project_moments<S_GM, NS, D, R>(A, p, sP, g2, 1.f, gm);
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int j = 0; j < D; ++j) sGM[j * 64 + lane] = gm[j];
      __builtin_amdgcn_wave_barrier();
    }
    // ---- R6: g_emb = [g_two_body | g_w0] @ G0^T + g_aenv,  g_aenv[e][k] = sum_j Y[e][j] GM[j][k];  dE/dY of the env path rides along
    v16f k0, k1;
    fused_layer<S_R6, NS, 2 + 2 * R, 2>(
        A, p,
        [&](auto kc) -> v16f {
          constexpr int k = decltype(kc)::value;
          if constexpr (k < 2) {
            return ld_tile(A.g_tb + 32 * k, e, A.ld_gtb, hh, row_ok);
          } else {
            constexpr int r = (k - 2) >> 1, h = (k - 2) & 1;
            if constexpr (h == 0) return tile_gw0<r, 0>(sB1 + 4 * hh, sB0 + 4 * hh, Y, g1a, g0a);
            return tile_gw0<r, 1>(sB1 + 4 * hh, sB0 + 4 * hh, Y, g1b, g0b);
          }
        },
        [&](auto, const v16f& a0, const v16f& a1) {
          const v16f ea = ld_tile(A.emb, e, 64, hh, row_ok), eb = ld_tile(A.emb + 32, e, 64, hh, row_ok);
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
            if ((j & 1) == 1) __builtin_amdgcn_sched_barrier(0);
          }
        });
    // ---- R7: g_h = (g_emb @ W_e1^T) * silu'(h_e)
    fused_layer<S_R7, NS, 2, 2>(
        A, p, [&](auto kc) -> const v16f& { if constexpr (decltype(kc)::value == 0) return k0; else return k1; },
        [&](auto, const v16f& a0, const v16f& a1) {
          const v16f za = ld_tile(A.se_h, e, 64, hh, row_ok), zb = ld_tile(A.se_h + 32, e, 64, hh, row_ok);
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            k0[i] = a0[i] * dsilu(za[i]);
            k1[i] = a1[i] * dsilu(zb[i]);
          }
        });
    // ---- R8: g_emb0 = g_h @ W_e0^T, contracted with the two-body table of the lane's type pair to the 8 basis sums
    float tsum[8];
    fused_layer<S_R8, NS, 2, 2>(
        A, p, [&](auto kc) -> const v16f& { if constexpr (decltype(kc)::value == 0) return k0; else return k1; },
        [&](auto, const v16f& a0, const v16f& a1) {
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
      if (A.gsh_env1) {
#pragma unroll
        for (int j = 0; j < D; ++j) gy[j] += A.gsh_env1[e * D + j];
      }
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
  }
}

size_t fused_bwd_tail_lds_bytes(int num_types) {
  return sizeof(u32x4) * 2 * kWStep + sizeof(float) * (32 + size_t(num_types) * num_types * 512 + size_t(kTailWaves) * kTailWaveFloats);
}

int fused_bwd_tail_num_steps(int R) { return 4 + (2 + 2 * R) + 2 + 2; }

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
  return AA_OK;
}

}  // namespace aa
