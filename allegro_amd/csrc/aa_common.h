// Shared declarations of the gfx950 Allegro hot-path library (internal; the public C ABI is
// include/allegro_amd.h).
#pragma once
#include <type_traits>
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "allegro_amd.h"

// Every kernel carves its scratch from this single dynamic-LDS symbol (keeping ONE __shared__
// object also avoids the ROCm 7.2 "second __shared__ object" vmcnt(0) trap, HIP guide §5 item 4a).
extern __shared__ unsigned char aa_smem[];

namespace aa {

void set_error(const std::string& msg);
int fail(int code, const std::string& msg);

#define AA_CHECK_HIP(expr)                                                                      \
  do {                                                                                          \
    hipError_t _e = (expr);                                                                     \
    if (_e != hipSuccess)                                                                       \
      return aa::fail(AA_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));           \
  } while (0)

#define AA_REQUIRE(cond, msg)                                       \
  do {                                                              \
    if (!(cond)) return aa::fail(AA_ERR_INVALID, std::string(msg)); \
  } while (0)

constexpr int kThreads = 256;

// ----------------------------------------------------------------------------------------------
// small device helpers
// ----------------------------------------------------------------------------------------------
// SiLU and its derivative.  fp64: exact forms.  fp32: v_exp_f32 + v_rcp_f32 (1-2 ulp each) instead of the IEEE
// division / accurate expf sequences, which cost ~15 VALU instructions per element (768 v_div_* in the first
// fused-chain kernel) -- the difference is ~1e-7 relative, far inside the reference's 5e-5 model tolerance.
// ---- cross-lane exchange primitives of gfx950 used by the wave reductions (VALU only, no LDS traffic).
// v_permlane32_swap / v_permlane16_swap exchange the upper half (odd 16-lane rows) of `a` with the lower half
// (even rows) of `b`.  They are issued as inline ISA: the compiler builtin of ROCm 7.2 returns the first result
// for both outputs.
#ifndef AA_HAVE_LANE_OPS  // (a build may pre-define these four primitives itself: the test-only CPU emulation does)
__device__ __forceinline__ void permlane32_swap(float& a, float& b) {
  asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ void permlane16_swap(float& a, float& b) {
  asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
}
// four independent swaps in one block (one pair of hazard nops for all of them)
__device__ __forceinline__ void permlane32_swap4(float* a, float* b) {
  asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %4\n\tv_permlane32_swap_b32 %1, %5\n\tv_permlane32_swap_b32 %2, %6\n\t"
      "v_permlane32_swap_b32 %3, %7\n\ts_nop 1"
      : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]));
}
__device__ __forceinline__ void permlane16_swap4(float* a, float* b) {
  asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %4\n\tv_permlane16_swap_b32 %1, %5\n\tv_permlane16_swap_b32 %2, %6\n\t"
      "v_permlane16_swap_b32 %3, %7\n\ts_nop 1"
      : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]));
}
// scheduling anchor: the value (a scalar or a register tuple) is final at this program point -- orders the arithmetic
// that produced it against the neighbouring anchors and memory operations (emits no instruction)
template <class V>
__device__ __forceinline__ void anchor(V& v) {
  asm volatile("" : "+v"(v)::"memory");
}
// a wave-uniform value the optimizer cannot see through from here on (stays in a scalar register; emits no instruction)
__device__ __forceinline__ void opaque_scalar(int& v) { asm volatile("" : "+s"(v)); }
// the same for a per-lane value: what is derived from it afterwards is recomputed where it is used, not hoisted out of a loop
__device__ __forceinline__ void opaque_vector(unsigned& v) { asm volatile("" : "+v"(v)); }
#endif
// element of a WAVE-UNIFORM `base` at a 32-bit byte offset: a byte GEP with a zero-extended index is the saddr + voffset addressing
// mode (one instruction); a per-lane 64-bit pointer costs a v_lshl_add_u64 per access
template <class T>
__device__ __forceinline__ T* lane_at(T* base, unsigned byte_off) {
  using B = std::conditional_t<std::is_const_v<T>, const char, char>;
  return reinterpret_cast<T*>(reinterpret_cast<B*>(base) + byte_off);
}
// DPP lane pattern applied to v (fused by the compiler into the consuming VALU op)
constexpr int kDppRowRor8 = 0x128, kDppRowRor4 = 0x124, kDppHalfMirror = 0x141, kDppQuad1032 = 0xB1, kDppQuad2301 = 0x4E;
template <int CTRL>
__device__ __forceinline__ float dpp_move(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}

// fp64 exp for the activations: argument reduction by ln 2 (two-part constant) + a degree-13 Taylor polynomial on
// |r| <= ln2/2 (remainder < 4e-18) + ldexp -- ~20 fp64 instructions instead of the ~150 of the generic libm exp with
// its special-case handling, < 1.5 ulp (tests/test_host_logic.py).  fp64 models evaluate SiLU / SiLU' for every
// element of every hidden layer in the GEMM operand staging, the epilogues and the moment kernels: with the libm
// sequence those were 5 VALU instructions per MFMA in the fp64 linear layers (profiles/archive/r02_v7_rocprofv3_c5_summary.txt).
__device__ __forceinline__ double aa_exp_f64(double x) {
  x = x > 709.0 ? 709.0 : (x < -745.0 ? -745.0 : x);
  const double k = rint(x * 1.44269504088896338700e+00);
  double r = fma(-k, 6.93147180369123816490e-01, x);
  r = fma(-k, 1.90821492927058770002e-10, r);
  double p = 1.6059043836821614599e-10;  // 1/13!
  p = fma(p, r, 2.0876756987868098979e-09);
  p = fma(p, r, 2.5052108385441718775e-08);
  p = fma(p, r, 2.7557319223985890653e-07);
  p = fma(p, r, 2.7557319223985892510e-06);
  p = fma(p, r, 2.4801587301587301566e-05);
  p = fma(p, r, 1.9841269841269841253e-04);
  p = fma(p, r, 1.3888888888888889419e-03);
  p = fma(p, r, 8.3333333333333332177e-03);
  p = fma(p, r, 4.1666666666666664354e-02);
  p = fma(p, r, 1.6666666666666665741e-01);
  p = fma(p, r, 0.5);
  p = fma(p, r, 1.0);
  p = fma(p, r, 1.0);
  return ldexp(p, int(k));
}
__device__ __forceinline__ double sigmoid_(double x) {
  const double d = 1.0 + aa_exp_f64(-x);
#if defined(__HIP_DEVICE_COMPILE__)
  double y = __builtin_amdgcn_rcp(d);  // v_rcp_f64 + two Newton steps: full double accuracy without the IEEE division sequence
  y = fma(fma(-d, y, 1.0), y, y);
  y = fma(fma(-d, y, 1.0), y, y);
  return y;
#else
  return 1.0 / d;
#endif
}
__device__ __forceinline__ float sigmoid_(float x) {
  return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}
template <typename T>
__device__ __forceinline__ T silu(T x) {
  return x * sigmoid_(x);
}
template <typename T>
__device__ __forceinline__ T dsilu(T x) {
  T s = sigmoid_(x);
  return s * (T(1) + x * (T(1) - s));
}
// Activation by kind (aa_model_config.act_kind): 0 silu, 1 mish, 2 gelu (erf), 3 identity.  Only the general (VALU)
// linear-layer kernels and the readout kernels take a kind; every fused fast path is SiLU-only and the plan falls
// back to the general kernels for the other nonlinearities of the reference (allegro_models.py:49-60).
enum { AA_ACT_SILU = 0, AA_ACT_MISH = 1, AA_ACT_GELU = 2, AA_ACT_NONE = 3 };
template <typename T>
__device__ __forceinline__ T act_apply(int kind, T x) {
  if (kind == AA_ACT_SILU) return silu(x);
  if (kind == AA_ACT_NONE) return x;
  const double xd = double(x);
  if (kind == AA_ACT_MISH) {
    const double sp = xd > 30.0 ? xd : log1p(exp(xd));  // softplus
    return T(xd * tanh(sp));
  }
  return T(0.5 * xd * (1.0 + erf(xd * 0.70710678118654752440)));
}
template <typename T>
__device__ __forceinline__ T act_grad(int kind, T x) {
  if (kind == AA_ACT_SILU) return dsilu(x);
  if (kind == AA_ACT_NONE) return T(1);
  const double xd = double(x);
  if (kind == AA_ACT_MISH) {
    const double sp = xd > 30.0 ? xd : log1p(exp(xd));
    const double t = tanh(sp), sg = 1.0 / (1.0 + exp(-xd));
    return T(t + xd * (1.0 - t * t) * sg);
  }
  const double cdf = 0.5 * (1.0 + erf(xd * 0.70710678118654752440));
  return T(cdf + xd * 0.39894228040143267794 * exp(-0.5 * xd * xd));
}
// fp32 sin / cos of the radial-basis arguments w x (|a| of a few tens; valid to |a| ~ 1e4): two-constant Cody-Waite
// reduction by pi/2 carried by FMAs + the degree-7 / degree-8 minimax polynomials of the BSD libm float kernels on
// [-pi/4, pi/4] -- ~25 instructions for both, max abs error 7e-8 against the double-precision functions (libm sinf: 6e-8).
// The generic sinf / cosf carry the Payne-Hanek reduction for huge arguments: ~245 instructions each, which made the 8
// Bessel functions a seventh of the fused forward's instruction stream and 28 % of the fused reverse tail's.
__device__ __forceinline__ void aa_sincos(float a, float& sn, float& cs) {
  const float k = __builtin_rintf(a * 0.63661977236758134f);
  float r = __builtin_fmaf(-k, 1.57079637050628662109375f, a);  // float(pi/2)
  r = __builtin_fmaf(-k, -4.37113900018624283e-8f, r);          // pi/2 - float(pi/2)
  const float z = r * r;
  float ps = __builtin_fmaf(z, 2.7183114939898219064e-6f, -1.98393348360966317347e-4f);
  ps = __builtin_fmaf(z, ps, 8.3333293858894631756e-3f);
  ps = __builtin_fmaf(z, ps, -1.66666666416265235595e-1f);
  const float s = __builtin_fmaf(r * z, ps, r);
  float pc = __builtin_fmaf(z, 2.43904487962774090654e-5f, -1.38867637746099294692e-3f);
  pc = __builtin_fmaf(z, pc, 4.16666233237390631894e-2f);
  pc = __builtin_fmaf(z, pc, -4.99999997251031003120e-1f);
  const float c = __builtin_fmaf(z, pc, 1.0f);
  const int q = int(k) & 3;
  sn = (q & 1) ? c : s;
  cs = (q & 1) ? s : c;
  if (q & 2) sn = -sn;
  if ((q + 1) & 2) cs = -cs;
}
__device__ __forceinline__ float aa_sin(float x) {
  float s, c;
  aa_sincos(x, s, c);
  return s;
}
__device__ __forceinline__ double aa_sin(double x) { return sin(x); }
__device__ __forceinline__ float aa_cos(float x) {
  float s, c;
  aa_sincos(x, s, c);
  return c;
}
__device__ __forceinline__ double aa_cos(double x) { return cos(x); }
// x^y: the polynomial-cutoff exponents are small integers (p = 6: x^5) -- square-and-multiply instead of powf's ~200 instructions
__device__ __forceinline__ float aa_pow(float x, float y) {
  const int n = int(y);
  if (float(n) == y && n >= 0 && n <= 32) {
    float r = 1.f, b = x;
    for (int m = n; m; m >>= 1) {
      if (m & 1) r *= b;
      b *= b;
    }
    return r;
  }
  return powf(x, y);
}
__device__ __forceinline__ double aa_pow(double x, double y) { return pow(x, y); }
__device__ __forceinline__ float aa_sqrt(float x) { return sqrtf(x); }
__device__ __forceinline__ double aa_sqrt(double x) { return sqrt(x); }

// irrep index r of SH component i (m ordered -l..l per l): l = floor(sqrt(i))
__device__ __forceinline__ int sh_l_of(int i) { return i < 1 ? 0 : (i < 4 ? 1 : (i < 9 ? 2 : 3)); }

// Column-segmented row-major matrix view: logical [M, sum n] split over up to 3 buffers
// (this is how torch.cat / torch.narrow of the reference never materialise here,
//  _allegro.py:253-258,278,284-294,300).
struct Seg {
  void* p;
  int ld;
  int n;
};
struct SegList {
  int count;
  Seg s[3];
};

// ----------------------------------------------------------------------------------------------
// GEMM  C_segs (=|+=) (act(A_segs)[M,K] @ B[K,N]) (* dsilu(Z_segs))
// ----------------------------------------------------------------------------------------------
struct GemmArgs {
  int64_t M;
  int K, N;
  SegList a;       // K split
  const void* B;   // [K,N] row-major
  const void* Bp;  // optional: B in MFMA fragment order [ceil(N/32)][ceil(K/32)][64 lanes][16] (f32 fast path)
  const void* Bq;  // optional: the same fragments split into 3 bf16 levels (bf16x3 path, aa_gemm.hip)
  SegList c;       // N split
  int c_accum[3];  // per C segment: 1 -> +=
  int has_z;       // multiply result by dsilu(z) (z split like c)
  SegList z;
  int act_a;       // apply silu to A on load ...
  int act_lo, act_hi;  // ... to columns [act_lo, act_hi) of A only (multiples of 32; act_hi == 0: every column) -- the folded first
                       // layers of the slot form read [two-body | pre-activations of earlier latents | tensor scalars]
  int has_add;     // result += add (before the dsilu(z) factor); split like c
  SegList add;
  int force_kernel;  // 0: automatic; 1: native fp32-input MFMA kernel; 3: VALU kernel (aa_debug_gemm_f32 / A-B tests)
  int act_kind;      // activation behind act_a / has_z (AA_ACT_*); anything but SiLU runs the general VALU kernel
  int opt_v1, opt_lds_epilogue, opt_f64_column_loop, opt_f64_rows;  // aa_plan_options pass-throughs (A/B switches)
  // Batched form (gridDim.z = batch <= 16 problems of the same shape in one launch; no z / add operands): problem b reads
  // A + b a_bs, writes C + b c_bs (elements) and multiplies by weight matrix number (bsel4 >> 4 b) & 15 of a set stored at uniform
  // strides (a packed word, not an array: a run-time-indexed kernel-argument array sends the WHOLE struct to scratch memory --
  // every linear layer of C5 ran 35-40 % slower with `unsigned char bsel[16]` here).
  // (the env projections of the operator kernels: one problem per spherical-harmonic component, one matrix per irrep)
  int batch;
  int64_t a_bs, c_bs, b_bs, bp_bs, bq_bs;
  unsigned long long bsel4;
};
template <typename T>
int launch_gemm(const GemmArgs& g, hipStream_t stream);

// Fused chain of up to 4 linear layers (fp32 via bf16x3): the 64-wide output of layer i stays in the MFMA
// accumulators and is consumed as (the trailing 64 inputs of) layer i+1.  Each layer is a GemmArgs (A segments
// may be empty) plus chaining flags.
struct ChainLayer {
  GemmArgs g;        // M, K (incl. 64 chained inputs if use_prev), N, a (global inputs), Bq, c/z/add (a C segment with
                     // p == nullptr is computed but not stored), act_a
  int use_prev;      // append the previous layer's kept 64 features as the last two k chunks
  int keep_tile;     // first of the two output tiles kept for the next layer, or -1
  int keep_act;      // silu on the kept values
  int a_mode;        // 1: A[e,k] = ro_factor * scale(type(center e)) * ro_w[k] * silu'(a[e,k])  (readout reverse)
                     // 2: A[e,k] = (a[e,k] + a2_add[e,k]) * silu'(a2_z[e,k]): the elementwise tail of a folded-away linear layer as the
                     //    operand transform of the next one (layers whose operand is split once: <= 2 k chunks, > 2 output tiles)
  const void* a2_add;  // [M, ld_a2add] (a_mode 2)
  const void* a2_z;    // [M, ld_a2z]
  int ld_a2add, ld_a2z;
  void* embrev_out;    // [M][8] or nullptr: out[e,n] = sum_c C[e,c] * emb_table[pair(e)][n][c] of this 64-wide layer -- the
                       // reverse of the two-body basis expansion folded into the epilogue (C itself need not be stored)
  void* edge_sum_out;  // [M] or nullptr: out[e] = sum_c silu(C[e,c]) * ro_w[c] of this (64-wide) layer -- the last linear
                       // readout layer folded into the epilogue, so the edge sum reads 4 B/edge instead of a row
  void* kept_out;      // [M, ld_kept] or nullptr: the kept 64 features AS KEPT (activated if keep_act) are also stored -- the hidden
  int ld_kept;         // activation a later launch consumes where the folded-away output layer's result used to be (staged forward)
};
struct ChainArgs {
  int64_t M;
  int nlayers;
  ChainLayer L[4];
  const void* ro_w;      // a_mode 1 extras
  double ro_factor;
  const void* ro_scales; // [T] or nullptr
  const int32_t* types;
  const int32_t* center;
  const int32_t* nbr;       // embrev_out extras
  const void* emb_table;    // [T*T][8][64]: type_embed(c | pair) * basis_linear[n][c]  (_edgeembed.py:70-84)
  int num_types;
};
int launch_gemm_chain(const ChainArgs& c, hipStream_t stream);  // fp32 only
// The one-layer latent-0 reverse chain with LDS-resident weights (aa_chain_res.hip): out = ((a + add) silu'(z)) @ W, K = 64, N = 128;
// columns 0..63 are ACCUMULATED into c0, columns 64..127 stored to c1.  Row strides in floats (multiples of 4).
struct ChainB2Args {
  int64_t M;
  const float* a;
  const float* add;
  const float* z;
  int lda, ldadd, ldz;
  const void* Wq;  // bf16x3 fragments of the [64, 128] matrix (gemm_pack_bf16x3)
  float* c0;
  float* c1;
  int ldc0, ldc1;
};
int launch_chain_b2_resident(const ChainB2Args& g, hipStream_t stream);
// element count of the fragment-ordered copy of a [K,N] matrix, and the host-side packer
size_t gemm_packed_elems(int K, int N);
void gemm_pack_b(const double* B, int K, int N, double* out);
size_t gemm_bf16x3_words(int K, int N);
void gemm_pack_bf16x3(const float* B, int K, int N, unsigned* out);

// ----------------------------------------------------------------------------------------------
// Sparse trilinear (Clebsch-Gordan) tables on device
// ----------------------------------------------------------------------------------------------
// One table computes  o[c] = sum_groups(c,p) W[.,p] * sum_{nz in group} val * A[a] * B[b]
struct TpGroup {
  int32_t out_idx;  // c
  int32_t path;     // p
  int32_t begin, end;
};
struct TpEntry {
  int32_t a, b;
  float val_f;
  double val_d;
};
struct TpTable {
  int32_t num_groups;
  const TpGroup* groups;   // device
  const TpEntry* entries;  // device
};
struct TpLayerDev {
  int32_t mul, d1, d2, dout, num_paths, coupling;
  TpTable fwd;   // out k ; A = x1[i],  B = x2[j]
  TpTable bx1;   // out i ; A = gout[k], B = x2[j]
  TpTable bx2;   // out j ; A = gout[k], B = x1[i]
};

// x1 / x2 operand sources of a TP layer
struct TpOperand {
  const void* dense;   // explicit [E,u,d] (or nullptr)
  const void* sh;      // implicit: sh[E,D] ...
  const void* w;       // ... times w[E, ldw] viewed as [u,R]  (MakeWeightedChannels, _channels.py:44-57)
  int ldw;
  int ld_sh;
};
struct TpOperandGrad {
  void* dense;     // explicit grad [E,u,d] (written) or nullptr
  void* gw;        // implicit: grad of w [E, ldgw] (written)
  int ldgw;
  void* gsh;       // grad of sh [E, ld_gsh] (accumulated, +=)
  int ld_gsh;
};

struct TpLayerFwdArgs {
  int64_t E, N;
  const int32_t* rowptr;
  const int32_t* eids;  // nullable
  TpOperand x1, x2;
  const void* weights;  // [u,p] or [p]
  double scatter_factor;
  void* x2s;            // [N,u,d2]
  void* out;            // [E,u,dout] or nullptr
  void* scal;           // [E, ld_scal] scalars (k=0) or nullptr
  int ld_scal;
};
struct TpLayerBwdArgs {
  int64_t E, N;
  const int32_t* rowptr;
  const int32_t* eids;
  TpOperand x1, x2;     // x2 only needed when implicit (for gsh)
  const void* weights;
  double scatter_factor;
  const void* x2s;      // saved [N,u,d2]
  const void* gout;     // [E,u,dout] or nullptr
  const void* gscal;    // [E, ld_gscal] added at k=0, or nullptr
  int ld_gscal;
  TpOperandGrad g1, g2;
};
template <typename T>
int launch_tp_layer_fwd(const TpLayerDev& L, const TpLayerFwdArgs& a, hipStream_t stream);
template <typename T>
int launch_tp_layer_bwd(const TpLayerDev& L, const TpLayerBwdArgs& a, hipStream_t stream);

// path-weight gradient of the operator (dense x1, saved x2s); `partial` is caller-provided scratch of
// tp_layer_wgrad_workspace_elems(L, N) elements
struct TpLayerWgradArgs {
  int64_t E, N;
  const int32_t* rowptr;
  const int32_t* eids;  // nullable
  const void* x1;       // [E,u,d1]
  const void* x2s;      // [N,u,d2] saved by the forward (scatter factor applied)
  const void* gout;     // [E,u,dout]
  void* partial;
  void* gw;             // [u,P] (coupled) or [P]
};
size_t tp_layer_wgrad_workspace_elems(const TpLayerDev& L, int64_t N);
template <typename T>
int launch_tp_layer_wgrad(const TpLayerDev& L, const TpLayerWgradArgs& a, hipStream_t stream);

// s_waitcnt vmcnt(0) (gfx9 encoding: expcnt / lgkmcnt fields at 'no wait'): all of the wave's global loads have landed.
// Used before a pipelined loop whose first trip would otherwise enter with prologue loads in flight: the compiler's
// waitcnt insertion merges that state with the back edge's conservatively and then waits for the NEWEST prefetches in
// front of every trip's first MFMA (seen in gemm_f64_rows_kernel: vmcnt(2) / vmcnt(0) ahead of MFMAs that read registers
// written by v_mov).
__device__ __forceinline__ void wait_vmem_all() { __builtin_amdgcn_s_waitcnt(0x0F70); }

template <typename T>
int launch_segment_sum(int64_t N, int64_t row, const void* x, const int32_t* rowptr, const int32_t* eids, double scale, void* out,
                       hipStream_t stream);
// slots (partial slabs) of the path-weight gradient, shared by the general and the dense kernels; the workspace holds
// nslots + 1 slabs of u * P elements (launch_tp_wgrad_reduce uses the last one)
int tp_wgrad_slots(int64_t N, int cap);
constexpr int kDenseWgradSlots = 8192;  // one wave per slot and 64-channel slice: enough waves to fill the chip at 10^5 atoms
template <typename T>
int launch_tp_wgrad_reduce(const void* partial, int nslots, int u, int P, int coupling, void* gw, hipStream_t stream);
// dst[m] = sum over rows of slabs[row][m], fixed order, two launches; OVERWRITES rows of `slabs` with chunk sums (aa_tp.hip)
template <typename T>
int launch_column_sum(T* slabs, int rows, int64_t M, T* dst, hipStream_t stream);

int build_tp_layer(const aa_tp_desc& d, TpLayerDev* out, std::vector<void*>* owned);

// ---- specialised (compile-time CG table) layer kernels, channel-minor layouts (aa_tp_spec.hip)
struct TpSpecFwdArgs {
  int64_t E, N;
  const int32_t* rowptr;
  int u;
  const void* sh;        // [E, ld_sh]
  int ld_sh;
  const void* w_x1;      // implicit x1 weights [E, ld_w1] laid out [R][u], or nullptr
  int ld_w1;
  const void* x1_dense;  // [E][D1][u] or nullptr
  const void* w_env;     // env weights [E, ld_we] laid out [R][u]
  int ld_we;
  const void* weights;   // [u,P] or [P]
  int coupling;
  double sf;
  void* x2s;             // [N][D2][u]
  void* out;             // [E][DOUT][u] or nullptr
  void* scal;            // [E, ld_scal] or nullptr
  int ld_scal;
};
struct TpSpecBwdArgs {
  int64_t E, N;
  const int32_t* rowptr;
  int u;
  const void* sh;
  int ld_sh;
  const void* w_x1;
  int ld_w1;
  const void* x1_dense;
  const void* w_env;
  int ld_we;
  const void* weights;
  int coupling;
  double sf;
  const void* x2s;       // saved [N][D2][u]
  const void* gout;      // [E][DOUT][u] or nullptr
  const void* gscal;     // [E, ld_gscal] or nullptr
  int ld_gscal;
  void* g_x1_dense;      // [E][D1][u] or nullptr
  void* g_w1;            // [E, ld_gw1] ([R][u]) written
  int ld_gw1;
  void* g_wenv;          // [E, ld_gwe] ([R][u]) written
  int ld_gwe;
  void* gsh_x1;          // [E, ld_gsh] grad of sh through the x1 operand (written; atomically added if u > 64)
  void* gsh_env;         // [E, ld_gsh] grad of sh through the env operand (written; atomically added if u > 64)
  int ld_gsh;
};
// 2-layer stacks ("chain"): the layer-1 kernels recompute tf1 = TP0(sh*w0, x2s0) per edge in registers and
// the layer-0 reverse kernel recomputes d_tf1 from (d_scal1, x2s1), so [E,u,D] tensors never touch HBM.
struct TpChainArgs {
  int64_t E, N;         // moments kernels: atoms [atom0, N) are processed (N = end of the owned block)
  int64_t atom0;
  const int32_t* rowptr;
  int u;
  const void* sh;
  int ld_sh;
  const void* w0;       // layer-0 x1 weights [E][R][u]
  int ld_w0;
  const void* wenv0;    // layer-0 env weights (reverse of layer 0 only)
  int ld_we0;
  const void* wenv1;    // layer-1 env weights
  int ld_we1;
  const void* weights0; // path weights [u,P0] / [P0]
  const void* weights1;
  int coupling;
  double sf;
  const void* x2s0;     // [N][D][u] (saved by the layer-0 forward)
  void* x2s1;           // [N][D][u] written by fwd_last, read by the reverse kernels
  void* scal1;          // fwd_last output [E, ld_scal]
  int ld_scal;
  const void* gscal0;   // [E, ld_gscal] (bwd_first)
  const void* gscal1;   // [E, ld_gscal]
  int ld_gscal;
  void* g_w0;           // bwd_first outputs
  int ld_gw0;
  void* g_wenv;         // env-weight grad of the layer being reversed
  int ld_gwe;
  void* gsh_x1;         // [E, ld_gsh]
  void* gsh_env;
  int ld_gsh;
};
int find_chain_pair(int sig0, int sig1);  // pair id or -1

// "Moments" variant of the chain kernels (u == 64): the env weights are a LINEAR image of a per-edge scalar
// vector a[e,:] (layer 0: the two-body embedding; layer l>0: silu of the previous latent hidden layer), and they
// enter only through the per-atom sum  x2s[n,j,ch] = f * sum_e sh[e,j] * (a[e,:] @ Wenv)[r(j),ch]
// (allegro/nn/_allegro.py:251-258,263,289-294 + allegro/nn/_strided/_contract.py:195-205).  Summing first,
//     M[n,j,k] = sum_e sh[e,j] * a[e,k],      x2s[n,j,ch] = f * sum_k M[n,j,k] * Wenv[k, r(j), ch],
// is exact by linearity and removes every [E, R*u] env tensor (and its gradient) from HBM.
struct TpMomArgs {
  TpChainArgs c;        // wenv0/wenv1/g_wenv are unused here
  const void* a0;       // layer-0 env input  [E, ld_a0]  (EDGE_EMBEDDING; no activation)
  int ld_a0, ka0;
  const void* a1;       // layer-1 env input  [E, ld_a1]  (pre-activation of latent 0's hidden layer; silu applied)
  int ld_a1, ka1;
  const void* wk0;      // Wenv of layer 0 as [ka][R][u]  (alpha folded)
  const void* wt0;      //                  and [R][u][ka]
  const void* wk1;
  const void* wt1;
  void* g_a;            // reverse kernels: grad wrt the env input of the layer being reversed [E, ld_ga]
  int ld_ga;
  int ka_lds;           // row stride of the wave-private moment patch in LDS (set by the launcher: max(ka0, ka1))
  int waves_per_block;  // 0 = 1 (aa_plan_options.moments_waves_per_block)
};
// Per-atom operator form of the tensor-product track for L <= 3 layers, u = 64*m (aa_tp_op.hip)
struct TpOpArgs {
  int64_t N, E;          // atoms [atom0, N) are processed
  int64_t atom0;
  const int32_t* rowptr;
  int u;
  const void* sh;        // [E, ld_sh]
  int ld_sh;
  const void* w0;        // [E][R][u] first-layer x1 weights
  int ld_w0;
  int coupling;
  double sf;             // 1/sqrt(avg_num_neighbors)
  const void* x2s[3];    // [N][D][u] per layer (written by that layer's forward)
  const void* tpw[3];    // path weights per layer
  const void* a;         // env input of the layer being processed [E, ld_a] (silu applied if act)
  int ld_a, ka, act;
  const void* wk;        // Wenv as [ka][R][u]
  const void* wt;        //      and [R][u][ka]
  void* scal;            // forward: [E, ld_scal]
  int ld_scal;
  const void* gscal[3];  // reverse: dE/dscal_m [E, ld_gscal] (m >= layer; all m for layer 0)
  int ld_gscal;
  void* q;               // [N][L][D][u] per-atom x1 moments Q_m (written for m = layer, read for m > layer)
  void* g_w0;            // layer 0 reverse: [E, ld_gw0]
  int ld_gw0;
  void* gsh_x1;          // layer 0 reverse: u/64 slots of [E, ld_gsh]
  void* gsh_env;         // ka/64 slots of [E, ld_gsh]
  int ld_gsh;
  void* g_a;             // grad wrt the env input [E, ld_ga]
  int ld_ga;
  int ka_lds;
  // split form (bvec != nullptr): the per-atom vectors travel through HBM between a register-heavy per-atom kernel
  // and lean, high-occupancy edge-streaming kernels
  void* bvec;            // [N][L][D1][u]  B_m(n) (forward: slot `layer`; layer-0 reverse: all L)
  void* gmbuf;           // [N][D][ka]     reverse: GM = d x2s . Wenv^T of the layer being reversed
  void* mbuf;            // [N][D][ka]     forward: moments M = sum_e Y[e] (x) act(a[e]) of the layer being evaluated
  int num_layers;
  // The two env projections -- x2s = f M Wenv (forward) and GM = f d x2s Wenv^T (reverse) -- are per-irrep matrix products over
  // all atoms.  proj_gemm: the caller runs them as batched linear-layer launches BETWEEN phase 1 and phase 2 of launch_tp_op
  // (split form only); the per-atom kernels then read x2s from / write d x2s to HBM instead of streaming the whole Wenv
  // (512 KB at C5) through every atom's wave.
  int proj_gemm;
  void* dx2s;            // [N][D][u]      reverse, proj_gemm: d x2s of the layer being reversed (unscaled)
  int env_mfma;          // fp64: the adjoint of the moments on the edges (tp_op_edge_env) on the f64 matrix cores
  int bvec_ready;        // reverse, split form: bvec already holds B_0 .. B_{L-1} of this step (the forward kernels of the same
                         // step wrote them): tp_op_bvecs_kernel, which recomputes exactly those vectors, is not launched
};
int find_op_chain(const int* sigs, int num_layers);  // chain id or -1
template <typename T>
int launch_tp_op(int chain, int layer, bool reverse, const TpOpArgs& a, hipStream_t stream, int phase = 0);  // phase: see TpOpArgs::proj_gemm

template <typename T>
int launch_tp_mom_fwd_first(int pair, const TpMomArgs& a, hipStream_t stream);
template <typename T>
int launch_tp_mom_fwd_last(int pair, const TpMomArgs& a, hipStream_t stream);
template <typename T>
int launch_tp_mom_bwd_last(int pair, const TpMomArgs& a, hipStream_t stream);
template <typename T>
int launch_tp_mom_bwd_first(int pair, const TpMomArgs& a, hipStream_t stream);
template <typename T>
int launch_tp_chain_fwd_last(int pair, const TpChainArgs& a, hipStream_t stream);
template <typename T>
int launch_tp_chain_bwd_last(int pair, const TpChainArgs& a, hipStream_t stream);
template <typename T>
int launch_tp_chain_bwd_first(int pair, const TpChainArgs& a, hipStream_t stream);

int find_spec_sig(const aa_tp_desc& d);  // signature id or -1

// operator seam on the generated signatures (aa_tp_dense.hip): the tensors of aa_tp_forward / aa_tp_backward in the
// reference's strided layout [E, u, d]
struct TpDenseArgs {
  int64_t E, N;
  const int32_t* rowptr;
  const int32_t* eids;   // stable sort permutation of an unsorted scatter index, or nullptr
  int u, coupling;
  double sf;             // scatter factor
  const void* x1;        // [E,u,d1]
  const void* x2;        // [E,u,d2]   (forward)
  const void* weights;   // [u,P] | [P]
  void* x2s;             // [N,u,d2]   forward: written; backward: read
  void* out;             // [E,u,dout] (forward)
  const void* gout;      // [E,u,dout] (backward)
  void* gx1;             // [E,u,d1]   (backward)
  void* gx2;             // [E,u,d2]   (backward)
};
bool tp_dense_supported(int sig, int u, int dtype);
template <typename T>
int launch_tp_dense(int sig, bool backward, const TpDenseArgs& a, hipStream_t stream);
template <typename T>
int launch_tp_dense_wgrad(int sig, int u, int coupling, const TpLayerWgradArgs& a, hipStream_t stream);
template <typename T>
int launch_tp_spec_fwd(int sig, const TpSpecFwdArgs& a, hipStream_t stream);
template <typename T>
int launch_tp_spec_bwd(int sig, const TpSpecBwdArgs& a, hipStream_t stream);

// ----------------------------------------------------------------------------------------------
// fused per-atom-tile kernels (aa_fused.hip): the whole forward of the standard 2-layer, 64-wide stack in ONE launch
// ----------------------------------------------------------------------------------------------
constexpr int kFusedMaxSteps = 56;
constexpr int kFusedKeepDefault = 2;
// Env projection of the fused forward, x2s[j][ch] = f sum_k M[j][k] Wenv[k][r(j)][ch], on the matrix cores: the per-atom moments
// become an operand tile whose 32 "edge" columns are the components j, one 64x64 bf16x3 layer per irrep (2 R MFMA steps instead
// of 4 env-weight steps of 288 FMAs each).  Built, parity-green, and measured 2.5 % SLOWER than the vector form on MI355X
// (same-box A/B, profiles/r04_v12_ab_c4_proj_mfma.txt: fused forward 4.01 -> 4.11 ms): 12 more MFMA steps per tile cost more
// than the 2 300 FMAs they replace.  -DAA_PROJ_MFMA builds it (A/B); the product uses the vector form.
#ifdef AA_PROJ_MFMA
constexpr bool kProjMfma = true;
#else
constexpr bool kProjMfma = false;
#endif
// The first layer of scalar_embed_mlp is LINEAR in the two-body embedding, and the embedding is linear in the 8 radial basis
// functions (emb0[c] = sum_n basis[n] tab[pair][n][c], scalarembed.py:60-81 / :157-175), so its pre-activation is
// h[k] = sum_n basis[n] T[pair][n][k] with T = tab[pair] @ W0 -- a table of the same size, folded at pack time.  The fused
// forward then has no 64x64 layer L0 (2 of its 32 MFMA steps and 2 tile splits) and the last reverse chain no layer W0^T
// (2 of 12 steps): d basis[n] = sum_k d_h[k] T[pair][n][k].  -DAA_NO_FOLD_EMBED builds the unfolded form (A/B).
#ifdef AA_NO_FOLD_EMBED
constexpr bool kFoldEmbed = false;
#else
constexpr bool kFoldEmbed = true;
#endif
// The same algebra one level up.  The OUTPUT layer of a latent ScalarMLPFunction is linear and its result only ever enters
// linear first layers (dense-net concat, _allegro.py:272-300: lat_l is a K-block of the next latent MLP and of edge_readout),
// so lat_l = a_l @ Wout_l never has to be formed in the forward pass: the consumers take the hidden activation a_l against
// Wout_l @ W_in[lat_l rows], folded at pack time (fp64, then rounded once).  The fused forward loses the two 64x64 output
// layers L4 and L7 (26 instead of 30 MFMA steps); the reverse pass only needs the stored pre-activations and is unchanged,
// except that the readout-reverse chain merges "d lat1 = d ro_h @ Wro[lat1]^T" and "@ Wout_1^T" into one 64x64 layer.
// -DAA_NO_FOLD_LATENT builds the unfolded form (A/B).
#ifdef AA_NO_FOLD_LATENT
constexpr bool kFoldLatent = false;
#else
constexpr bool kFoldLatent = true;
#endif
// The single-layer pipeline's version of all of the above for any depth ("slot form", see aa_model_plan::slot_form): run-time
// switch aa_plan_options.no_slot_form, build-time -DAA_NO_SLOT_FORM.
#ifdef AA_NO_SLOT_FORM
constexpr bool kSlotForm = false;
#else
constexpr bool kSlotForm = true;
#endif
// reverse side of the lat0 fold: Wout_0^T rides in the lat0 columns of the readout-reverse chain's second layer, and what is left
// of the output layer's reverse -- (d a_0 + d a_0 of the moments) x silu'(h) -- is the operand transform (a_mode 2) of the
// latent-0 reverse chain, which is then ONE layer (4 steps instead of 6).  -DAA_NO_FOLD_LAT0_REV: A/B.
#if defined(AA_NO_FOLD_LAT0_REV) || defined(AA_NO_FOLD_LATENT)
constexpr bool kFoldLat0Rev = false;
#else
constexpr bool kFoldLat0Rev = true;
#endif
// ... and at the front: EDGE_EMBEDDING = a_e @ W1 (the linear output layer of scalar_embed_mlp, a_e = silu(h)) only ever enters
// linear maps -- env_embed_linear / first_layer_env_embed_projection (tensorembed.py:88-89, _allegro.py:251-258) and, through
// the moments, the env weights of layer 0.  With W1 folded into all of them (first stage: W1 @ [proj | env]; env weights:
// W1 @ Wenv0) the fused forward has no layer L1 either (24 MFMA steps), works on a_e wherever it used the embedding and stores
// a_e in the embedding's slot; the reverse pass that FOLLOWS A FUSED FORWARD reads a_e there: tp_mom_bwd_first with the folded
// transposed env weights, and the last reverse chain is ONE 256 -> 64 layer ((W1 @ G0)^T, + d a_e of the moments, x silu'(h),
// contracted against the folded two-body table).  After a staged forward (true embedding stored) the unfolded reverse runs.
// -DAA_NO_FOLD_EMB1 builds without (A/B); the experimental reverse tail reads the true embedding: no fold there.
#if defined(AA_NO_FOLD_EMB1) || defined(AA_NO_FOLD_EMBED) || defined(AA_EXPERIMENTAL_TAIL)
constexpr bool kFoldEmb1 = false;
#else
constexpr bool kFoldEmb1 = true;
#endif  // FusedFwdArgs::keep when aa_plan_options.fused_keep_split is 0
constexpr int kFusedMaxDegree = 128;  // longest edge segment the fused forward takes: a team of four 32-edge tiles
constexpr int kFusedTailAtomsPerCu = 64;   // from this many atoms per CU on the fused forward runs as eight-wave workgroups with the readout-reverse chain in its tail
constexpr int kFusedTeamTilesSmall = 4096;  // up to this many tiles the team form is chosen regardless of how full the tiles are
struct FusedFwdArgs {
  int64_t N, atom0, atom_end;  // atoms [atom0, atom_end) are evaluated: one wave each (every one has <= 32 edges), or -- with
                               // tile_atoms -- teams of 1 / 2 / 4 waves (<= 128 edges)
  int32_t* tile_atoms;         // nullptr | [3][tile_cap] class lists (scratch, filled by fused_classify_kernel)
  int32_t* tile_counts;        // [4] class counters (scratch)
  int64_t tile_cap;
  const int32_t *rowptr, *nbr, *types;
  const float* pos;
  const float* shift_vec;  // [E,3] or nullptr
  int num_types, embed_kind, spline_span;
  float poly_p;
  const float *rmax_recip, *bessel_w;
  const float* emb_tab;    // [T*T][8][64]
  // the kernel's weight program: one 12-KB block per step, as two 6-KB halves (a tile pair x 32-deep chunk of a linear
  // layer's bf16x3 fragments, or 16 rows of an env-weight matrix [k][R][64]) -- built by fused_fwd_program()
  const void* wstep[kFusedMaxSteps][2];
  const float *tpw0, *tpw1;  // path weights
  int coupling;
  float sf;                  // 1/sqrt(avg_num_neighbors)
  const float* ro_w;         // [64] last readout layer
  float ro_factor;
  const float *scales, *shifts;
  // outputs (what the reverse pass reads again)
  float* vec;      // [E,4]
  float* sh;       // [E,D] or nullptr
  float* se_h;     // [E,64] pre-activation of scalar_embed_mlp's hidden layer
  float* emb;      // [E,64] EDGE_EMBEDDING
  float* w0;       // [E,R*64] or nullptr (the staged reverse kernels read it; the fused reverse recomputes it)
  float* lat_h0;   // [E,64]
  float* lat_h1;   // [E,64]
  float* ro_h;     // [E,64]
  float* fcat;     // [E,192] EDGE_FEATURES or nullptr
  float *x2s0, *x2s1;  // [N][D][64]
  float* atom_energy;  // [N]
  int keep;            // split tile pairs held in registers by the one-tile w0-holding form: 0 none, 1 two-body scalars, 2 + lat0
  int32_t* status;     // nullable, host-visible: set to the offending degree when a segment exceeds what the max_degree hint promised
  int mixed;           // with the class lists set: one-tile pass over all atoms (long ones skipped) + team pass over the long ones only
  int skip_long, long_only, fill_done;  // (set by launch_fused_fwd for the two passes of the mixed form)
  // two-waves-per-SIMD form with the readout-reverse chain in its tail (`tail` != 0; aa_fused8.hip): d EDGE_FEATURES[:, :128] and the
  // gradient of the layer-1 tensor-track scalars leave the forward kernel, the pre-activations of latent 1 / the readout do not
  float* g_fcat;
  float* g_scal1;
  int ld_gfcat, tail;
  int wide_one_per_cu; // two-waves-per-SIMD form, four-wave workgroups: ONE per CU (half the registers and LDS of a CU stay free for kernels of other streams)
  int wide_proj_mfma;  // two-waves-per-SIMD form: env projections as bf16x3 layers on the matrix cores (its program then has 2 R steps per projection)
  int wide_waves;      // two-waves-per-SIMD form (aa_fused8.hip): 4 = four-wave workgroups except on small boxes; -4 / -8: four / eight waves, forced
};
size_t fused_fwd_lds_bytes(int num_types, bool teams);  // dynamic LDS of the fused forward (aa_fused.hip); the CU has 160 KB
// reverse tail (aa_fused_bwd.hip): layer-0 tensor product reverse + first-stage / scalar_embed_mlp reverse + edge reverse
struct FusedTailArgs {
  int64_t N, atom0, atom_end;  // atoms [atom0, atom_end), one wave each; every one has <= 32 edges
  const int32_t *rowptr, *nbr, *types;
  int num_types, embed_kind, spline_span;
  float poly_p;
  const float *rmax_recip, *bessel_w;
  const float* emb_tab;      // [T*T][8][64]
  const void* wstep[kFusedMaxSteps][2];  // weight program, 12-KB blocks (see fused_bwd_tail_kernel)
  const float *tpw0, *tpw1;  // path weights
  int coupling;
  float sf;                  // 1/sqrt(avg_num_neighbors)
  // inputs
  const float* vec;          // [E,4] unit vector, length (forward)
  const float* w0;           // [E,R*64] (forward)
  const float* emb;          // [E,64] EDGE_EMBEDDING (forward)
  const float* se_h;         // [E,64] pre-activation of scalar_embed_mlp's hidden layer (forward)
  const float *x2s0, *x2s1;  // [N][D][64] (forward)
  const float *gscal0, *gscal1;  // [E,64] gradients of the tensor-track scalars of the two layers
  const float* g_tb;         // [E, ld_gtb] gradient of the two-body scalars (first 64 columns of d EDGE_FEATURES)
  int ld_gtb;
  const float* gsh_env1;     // [E,D] dE/dY of the layer-1 env path (tp_mom_bwd_last) or nullptr
  // outputs: dvec [E,4] = dE/dr_e (edge reverse fused), or -- dvec == nullptr -- the inputs of edge_backward
  float* dvec;
  float* trev;               // [E,8]
  float* gsh_out;            // [E,D]
};
int fused_bwd_tail_num_steps(int R);
int fused_bwd_tail_chunk_order(int R, int i);
int launch_fused_bwd_tail(int pair, const FusedTailArgs& a, hipStream_t stream);
int fused_fwd_num_steps(int R, bool hold_w0);
// `wide` (nullable): the same arguments with the weight program of the eight-wave form (aa_fused8.hip) -- it then takes the one-tile
// pass (all atoms, or all but the long ones of the mixed form); the team pass keeps `a`
int launch_fused_fwd(int pair, bool hold_w0, const FusedFwdArgs& a, hipStream_t stream, const FusedFwdArgs* wide = nullptr, bool* ran_wide = nullptr);
int fused_fwd8_num_steps(int R, bool proj_mfma, bool tail = false);
int fused_num_cus();  // CUs of the current device (the small-box rule of launch_fused_fwd)
size_t fused_fwd8_lds_bytes(int num_types, int waves);  // per workgroup of 8 waves (one per CU) or 4 waves (two per CU)
int launch_fused_fwd8(int pair, int waves, const FusedFwdArgs& a, hipStream_t stream);


// ----------------------------------------------------------------------------------------------
// edge prologue / epilogue / readout reduce
// ----------------------------------------------------------------------------------------------
struct EdgeGeomArgs {
  int64_t E, N;
  const int32_t *center, *nbr, *types;
  const void* pos;        // [N,3]
  const void* shift_vec;  // [E,3] or nullptr
  int num_types, num_bessels, l_max, S0;
  double poly_p;
  const void* rmax_recip;      // [T,T]
  const void* bessel_w;        // [B]
  const void* center_embed;    // [T,S0/2]
  const void* neighbor_embed;  // [T,S0/2]
  const void* basis_w;         // [B,S0] (alpha folded)
  int embed_kind, spline_span; // 1: per-class spline embedding (spline.py), num_bessels = num_splines
  const void* emb_tab;         // embed_kind 1: [T*T][B][S0] class weights, basis-major
  void* vec;                   // [E,4]  (unit vector xyz, r)
  void* sh;                    // [E,D]
  void* emb0;                  // [E,S0]
};
template <typename T>
int launch_edge_prologue(const EdgeGeomArgs& a, hipStream_t stream);

struct EdgeBwdArgs {
  EdgeGeomArgs g;
  const void* g_emb0;  // [E,S0]
  const void* g_sh;    // [num_gsh][E,D] slices, summed here
  int num_gsh;
  void* forces;        // [N,3] (pre-zeroed; accumulated with atomics)
  const void* t_in;    // [E,B] or nullptr: dE/d(Bessel x cutoff) already contracted by the producer (then g_emb0 is unused)
  void* dvec;          // [E,4] or nullptr: dE/dr_e per edge (always written when set; aa_model_virial reads it too)
  int gather;          // 1: forces are assembled by force_gather_kernel from dvec; 0: scattered here with atomics
};
// W[a][b] = sum_e d[e][a] * r_e[b]   (strain derivative of the energy; r_e = vec[e].xyz * vec[e].w)
struct VirialArgs {
  int64_t E;
  const void* dvec;  // [E,4]
  const void* vec;   // [E,4] unit vector, length
  double* partial;   // [kVirialBlocks][9] scratch
  void* out;         // [9] model dtype
};
constexpr int kVirialBlocks = 512;
template <typename T>
int launch_virial(const VirialArgs& a, hipStream_t stream);
// F[n] = sum_{e in seg(n)} d[e] - sum_{e: nbr(e) = n} d[e], gathered in a fixed order (deterministic)
struct ForceGatherArgs {
  int64_t N;
  const int32_t *rowptr, *t_rowptr, *t_perm;
  const void* dvec;  // [E,4]
  void* forces;      // [N,3]
};
template <typename T>
int launch_force_gather(const ForceGatherArgs& a, hipStream_t stream);
// verifies the aa_graph.atom_begin / atom_end promise (no edge segment outside the block): *status = -2 otherwise
int launch_graph_hint_check(const int32_t* rowptr, int64_t N, int64_t a0, int64_t a1, int32_t* status, void* atom_energy, void* forces,
                            int esize, hipStream_t stream);
template <typename T>
int launch_edge_backward(const EdgeBwdArgs& a, hipStream_t stream);

struct ReadoutArgs {
  int64_t E, N;
  const int32_t *rowptr, *center, *types;
  const void* h;   // [E,H] (pre-activation if act) , ld
  int ld, H, act;
  const void* w;   // [H] final linear (alpha folded)
  double factor;   // 1/sqrt(2 avg_nn)   (allegro_models.py:245)
  const void* scales;  // [T] or nullptr
  const void* shifts;  // [T] or nullptr
  void* atom_energy;   // [N]
  void* g_h;           // bwd: [E,H] written
  const void* edge_sum;  // [E] or nullptr: per-edge values already contracted with w (then h/w/act are unused)
  int act_kind;          // AA_ACT_* of the readout MLP (used when act != 0)
};
template <typename T>
int launch_readout_reduce(const ReadoutArgs& a, hipStream_t stream);
template <typename T>
int launch_readout_backward(const ReadoutArgs& a, hipStream_t stream);

}  // namespace aa
