/* allegro_amd.h -- C ABI of the MI355X (gfx950) Allegro hot path.
 *
 * Boundary rules: plain pointers and sizes only (no torch types); every tensor
 * buffer -- inputs, outputs, workspace, packed weights -- is owned by the CALLER and lives in
 * device memory (HBM); the library allocates only the small immutable CG tables owned by a plan
 * handle; all work is enqueued on the caller's hipStream_t; functions return 0 on success and a
 * negative code otherwise (never throw); aa_last_error() gives the message for the calling thread.
 *
 * The reference (mir-group/allegro) has NO native interface; its accelerator seams are Python
 * (SURVEY.md §8b).  Each entry point below names the reference interface it stands behind.
 */
#ifndef ALLEGRO_AMD_H
#define ALLEGRO_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* The library is built with -fvisibility=hidden: the declarations of this header are its ONLY exported symbols
 * (tests/test_lib_symbols.py); the pragma gives them default visibility where the library itself is compiled and is
 * harmless in a consumer. */
#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility push(default)
#endif

typedef void* aa_stream; /* hipStream_t */

typedef enum { AA_F32 = 0, AA_F64 = 1 } aa_dtype;

enum {
  AA_OK = 0,
  AA_ERR_INVALID = -1,   /* bad argument / unsupported configuration */
  AA_ERR_WORKSPACE = -2, /* caller workspace too small */
  AA_ERR_HIP = -3        /* a HIP runtime call failed */
};

#define AA_MAX_LAYERS 4
#define AA_MAX_MLP_LAYERS 4

const char* aa_last_error(void);
int aa_version(void);

/* ------------------------------------------------------------------------------------------
 * 1. Tensor-product operator  (seam B1/B2)
 *    replaces allegro/nn/_strided/_contract.py:185-251 (Contracter.forward / ._contract); the
 *    descriptor carries what Contracter.__init__ builds at :53-177.
 * ------------------------------------------------------------------------------------------ */
typedef struct {
  int32_t mul;            /* u: strided channel count                     (_contract.py:75)   */
  int32_t d1, d2, dout;   /* base dims of in1, in2, out                   (_contract.py:72-74)*/
  int32_t num_paths;      /* p                                            (_contract.py:76)   */
  int32_t coupling;       /* 1: weights [u,p] ("uuup"), 0: weights [p]    (_contract.py:172)  */
  int32_t nnz;            /* non-zeros of the (norm-folded) w3j buffer    (_contract.py:95-119)*/
  const int32_t* nz_i;    /* [nnz] index into in1  (for ij-diagonal w3j: j == i)               */
  const int32_t* nz_j;    /* [nnz] index into in2                                              */
  const int32_t* nz_k;    /* [nnz] index into out                                              */
  const int32_t* nz_path; /* [nnz] path of the entry                                           */
  const double* nz_val;   /* [nnz] w3j value (already x sqrt(2 l_out+1), _contract.py:110,115) */
} aa_tp_desc;

typedef struct aa_tp_plan aa_tp_plan;

/* host arrays in `desc` are copied; tables are uploaded to the current device */
int aa_tp_plan_create(const aa_tp_desc* desc, aa_dtype dtype, aa_tp_plan** out);
void aa_tp_plan_destroy(aa_tp_plan* plan);
/* A plan whose descriptor equals one of the generated signatures of standard Allegro layers (and whose channel count is
 * 64, 128 or 256) runs aa_tp_forward / aa_tp_backward on specialised kernels with compile-time Clebsch-Gordan code
 * (aa_tp_dense.hip); on != 0 keeps the table-driven general kernels instead (A/B measurements, tests). */
int aa_tp_plan_use_general_kernels(aa_tp_plan* plan, int on);
int aa_tp_plan_is_specialised(const aa_tp_plan* plan); /* 1: aa_tp_forward / aa_tp_backward run the specialised kernels */

/* Contracter.forward (_contract.py:185-211) on a center-sorted segment layout:
 *   x2s[n] = scatter_factor * sum_{s in [rowptr[n],rowptr[n+1])} x2[eid(s)]   (:195-204)
 *   out[e] = contract(x1[e], x2s[center(e)])                                  (:205-211)
 * eids (nullable): sorted position -> edge id, for callers whose `idxs` are not sorted
 * (tests/nn/test_contract_kernels.py:95-97).  x1:[E,u,d1] x2:[E,u,d2] out:[E,u,dout],
 * weights: [u,p] or [p] (device), x2s: [N,u,d2] caller buffer (kept for the backward). */
int aa_tp_forward(const aa_tp_plan* plan, int64_t E, int64_t N, const void* x1, const void* x2,
                  const void* weights, const int32_t* rowptr, const int32_t* eids,
                  double scatter_factor, void* x2s, void* out, aa_stream stream);

/* input gradients of the above: gx1:[E,u,d1], gx2:[E,u,d2].  Either output may be NULL: that gradient is not computed and
 * the operand only it reads may be NULL too (gx1 == NULL: x2s unused; gx2 == NULL: x1 unused) -- the single partial
 * contractions the training path differentiates through (allegro_amd/ops.py; the reference gets them from autograd
 * through the eager contraction, _contract.py:213-251). */
int aa_tp_backward(const aa_tp_plan* plan, int64_t E, int64_t N, const void* x1, const void* x2s,
                   const void* weights, const int32_t* rowptr, const int32_t* eids,
                   double scatter_factor, const void* gout, void* gx1, void* gx2, aa_stream stream);

/* the scale + scatter-sum of `Contracter.forward` alone (_contract.py:195-204) on the same segment layout:
 * out[n] = scale * sum_{s in [rowptr[n],rowptr[n+1])} x[eid(s)], rows of row_elems contiguous elements (u * d2);
 * deterministic (CSR order, no atomics).  The training path uses it to form x2s for the partial contractions. */
int aa_tp_segment_sum(aa_dtype dtype, int64_t E, int64_t N, int64_t row_elems, const void* x, const int32_t* rowptr,
                      const int32_t* eids, double scale, void* out, aa_stream stream);

/* path-weight gradient of the above (training): gweights has the shape of `weights` ([u,p] or [p]).  The
 * reference's eager Contracter and its cuEquivariance variant get it from autograd through the `weights`
 * Parameter (_contract.py:172-177,219; docs/guide/accelerations.rst:15-17); its Triton path omits it
 * (_flashallegro.py:660).  Deterministic (fixed summation order, no atomics).  `workspace`: caller scratch of
 * aa_tp_weights_workspace_bytes(plan, N) bytes. */
size_t aa_tp_weights_workspace_bytes(const aa_tp_plan* plan, int64_t N);
int aa_tp_backward_weights(const aa_tp_plan* plan, int64_t E, int64_t N, const void* x1, const void* x2s,
                           const int32_t* rowptr, const int32_t* eids, const void* gout, void* workspace,
                           size_t workspace_bytes, void* gweights, aa_stream stream);

/* ------------------------------------------------------------------------------------------
 * 1b. Training-mode building blocks around the operator.  The reference trains every
 *    ScalarMLPFunction and every environment weighting through autograd (allegro/nn/_allegro.py:192-213,
 *    _strided/_channels.py:44-63); these are the pieces of that graph whose library / eager forms are slow on this
 *    device, as plain functions the Python host wraps into differentiable ops (allegro_amd/ops.py).
 * ------------------------------------------------------------------------------------------ */
/* Weight gradient of one bias-free linear layer y = x @ W (nequip ScalarMLPFunction, EXT): out[K,N] = sum_e x[e,:]^T g[e,:],
 * x [E,K] with row stride ldx, g [E,N] with row stride ldg (elements), out dense [K,N].  The reduction over the edges is cut
 * into slabs summed in a fixed order (bit-reproducible, no atomics); products are exact fp32 / fp64 on the matrix cores.
 * `workspace`: caller scratch of aa_linear_wgrad_workspace_bytes() bytes. */
size_t aa_linear_wgrad_workspace_bytes(aa_dtype dtype, int64_t E, int K, int N);
int aa_linear_wgrad(aa_dtype dtype, int64_t E, int K, int N, const void* x, int64_t ldx, const void* g, int64_t ldg,
                    void* workspace, size_t workspace_bytes, void* out, aa_stream stream);
/* The products ALONG the edges of the same layers in training mode: out[E,N] = x[E,K] @ B, B[k][n] = W[k ldk + n ldn] a DEVICE fp32
 * matrix that may change between calls (W or its transpose, no copy); fp32 through the split-bf16 matrix-core kernel of the inference
 * pipeline (three exact levels, fp32 accumulation; bounded against fp64 in tests/test_gemm_accuracy.py).  K, N multiples of 32;
 * row strides multiples of 4 elements, x / out / workspace 16-byte aligned; anything else: AA_ERR_INVALID (callers keep a library GEMM
 * for those).  `workspace`: aa_linear_forward_workspace_bytes(K, N) bytes of scratch (W in fragment order). */
size_t aa_linear_forward_workspace_bytes(int K, int N);
int aa_linear_forward(int64_t E, int K, int N, const float* x, int64_t ldx, const float* W, int64_t ldk, int64_t ldn, void* workspace,
                      size_t workspace_bytes, float* out, int64_t ldo, aa_stream stream);
/* MakeWeightedChannels (_channels.py:44-63) as a bilinear form and its two partial contractions; sh [E,D], D = (l_max+1)^2,
 * w [E,u,R] with R = l_max+1 weights per channel (shared != 0: R = 1, `weight_individual_irreps=False`), t [E,u,D]:
 *   which 0:  out[E,u,D] = a=sh (x) b=w           out[e,c,i] = sh[e,i] w[e,c,r(i)]
 *   which 1:  out[E,u,R] = a=t  . b=sh            out[e,c,r] = sum_{i in r} t[e,c,i] sh[e,i]
 *   which 2:  out[E,D]   = a=t  . b=w             out[e,i]   = sum_c t[e,c,i] w[e,c,r(i)]
 * Each is one pass over the [E,u,D] operand; the three are closed under differentiation (any derivative of one is another
 * with operands substituted), which is what a force-matching loss needs.  `ldw`: row stride (elements, >= u R) of the weight operand
 * (which 0 and 2: b) -- the weights are usually a column block of a wider MLP output; ignored for which 1. */
int aa_weighted_channels(aa_dtype dtype, int which, int64_t E, int u, int l_max, int shared, const void* a, const void* b, int64_t ldw,
                         void* out, aa_stream stream);
/* The two combinations every derivative of the forms above asks for, each in one pass over the [E,u,D] tensor instead of two or
 * three:  _pair: out_sh[E,D] = t . w (which 2) AND out_w[E,u,R] = t . sh (which 1) from one read of t;
 *         _sum:  out[E,u,D] = sh (x) w + sh2 (x) w2 (the gradient of the pair with respect to t) with one store stream. */
int aa_weighted_channels_pair(aa_dtype dtype, int64_t E, int u, int l_max, int shared, const void* t, const void* sh, const void* w, int64_t ldw,
                              void* out_sh, void* out_w, aa_stream stream);
int aa_weighted_channels_sum(aa_dtype dtype, int64_t E, int u, int l_max, int shared, const void* sh, const void* w, int64_t ldw, const void* sh2,
                             const void* w2, int64_t ldw2, void* out, aa_stream stream);
/* out[rows, D] = a (NULL: zeros) with s[rows] added to component 0 of every row: the gradient of a tensor feature whose scalar components
 * also feed the next latent MLP (`features[:, :, 0]`, _allegro.py:275-283), in one pass.  a, out 16-byte aligned. */
int aa_scalar_column(aa_dtype dtype, int64_t rows, int D, const void* a, const void* s, void* out, aa_stream stream);
/* out[e, off_j + c] = x_j[e, c] for n <= 8 inputs of `widths[j]` columns and row strides `ldx[j]` (elements): the scalar features of all
 * layers side by side, the input of every latent MLP and of the readout (_allegro.py:275-283) -- and the gradient of a column split. */
int aa_concat_columns(aa_dtype dtype, int64_t E, int n, const void* const* xs, const int64_t* ldx, const int* widths, void* out, int64_t ldo,
                      aa_stream stream);
/* The hidden activation of the scalar MLPs (ScalarMLPFunction with SiLU; _allegro.py:192-213) and its derivatives, elementwise over n
 * values: A_k(x, g) = g f^(k)(x), f(x) = x sigmoid(x), k = `order` in 0..3; g may be NULL (= 1).  The family is closed under
 * differentiation: d A_k/dx . h = A_{k+1}(x, g h), d A_k/dg . h = A_k(x, h); `_pair` returns both from one pass:
 * out_x = g h f^(order+1)(x), out_g = h f^(order)(x) (order <= 2).  Pointers 16-byte aligned. */
int aa_silu_derivative(aa_dtype dtype, int order, int64_t n, const void* x, const void* g, void* out, aa_stream stream);
int aa_silu_derivative_pair(aa_dtype dtype, int order, int64_t n, const void* x, const void* g, const void* h, void* out_x, void* out_g,
                            aa_stream stream);
/* The same family for every nonlinearity the reference's MLPs offer (allegro_models.py:49-60): act = 0 silu, 1 mish (x tanh softplus x),
 * 2 gelu (erf form, torch's default); orders 0..3 as above. */
int aa_act_derivative(aa_dtype dtype, int act, int order, int64_t n, const void* x, const void* g, void* out, aa_stream stream);
int aa_act_derivative_pair(aa_dtype dtype, int act, int order, int64_t n, const void* x, const void* g, const void* h, void* out_x, void* out_g,
                           aa_stream stream);

/* ------------------------------------------------------------------------------------------
 * 2. Whole hot path: forward + forces  (seam B3 + ForceStressOutput)
 *    replaces the module chain of allegro/model/allegro_models.py:222-297 wrapped by
 *    ForceStressOutput (:101-103): two-body embedding -> SH tensor embed (tensorembed.py:85-96)
 *    -> Allegro_Module.forward (_allegro.py:237-301) -> edge readout -> EdgewiseReduce
 *    (edgewise.py:40-60) -> per-type scale/shift, and its reverse pass w.r.t. positions.
 * ------------------------------------------------------------------------------------------ */
typedef struct {
  int32_t dtype;           /* aa_dtype                                                        */
  int32_t num_types;
  int32_t num_bessels;     /* scalarembed.py:22                                               */
  double poly_p;           /* polynomial cutoff exponent, scalarembed.py:24                   */
  int32_t l_max;           /* 1..3                                                            */
  int32_t num_layers;      /* L, _allegro.py:24                                               */
  int32_t num_scalar;      /* S, _allegro.py:25                                               */
  int32_t num_tensor;      /* u, _allegro.py:26 (not a multiple of 64: evaluated as the zero-padded next multiple where
                            * that is the faster path -- same results; aa_plan_options.no_channel_padding)  */
  int32_t embed_dim;       /* S0 = radial_chemical_embed_dim, allegro_models.py:160-164        */
  int32_t embed_mlp_depth, embed_mlp_width;     /* allegro_models.py:175-176                  */
  int32_t latent_mlp_depth, latent_mlp_width;   /* allegro_models.py:205-206                  */
  int32_t readout_mlp_depth, readout_mlp_width; /* allegro_models.py:233-234                  */
  int32_t forward_weight_init;                  /* allegro_models.py:146                      */
  double avg_num_neighbors;                     /* _allegro.py:182; allegro_models.py:245     */
  double act_const;        /* normalize2mom constant of SiLU used by ScalarMLPFunction        */
  int32_t has_scales, has_shifts;               /* allegro_models.py:251-260                  */
  aa_tp_desc tps[AA_MAX_LAYERS];                /* one per layer, _allegro.py:172-183         */
  /* two-body radial/chemical embedding: 0 = TwoBodyBesselScalarEmbed (scalarembed.py:19-81),
   * 1 = TwoBodySplineScalarEmbed (scalarembed.py:84-175; spline.py): `num_bessels` is then num_splines */
  int32_t embed_kind;
  int32_t spline_span;     /* spline.py:27                                                     */
  /* MakeWeightedChannels of the Allegro layers (_allegro.py:83-86, _channels.py:29-31,60-63): 0 = one env weight per
   * (channel, irrep) [default, weight_individual_irreps=True], 1 = one per channel shared by all irreps */
  int32_t env_shared_weights;
  /* nonlinearity of scalar_embed_mlp / the latent MLPs / edge_readout (allegro_models.py:49-60,126-138):
   * 0 silu, 1 mish, 2 gelu (erf form), 3 none (no activation between the layers); act_consts[i] is the matching
   * normalize2mom constant of nequip's ScalarMLPFunction (1 for "none"; 0 = use act_const) */
  int32_t act_kind[3];
  double act_consts[3];
  /* nequip's BesselEdgeLengthEncoding (EXT; called from scalarembed.py:60-66) has been published in two forms:
   * 1 = stored roots n*pi, basis sin(w x) / x;  2 = stored roots n, basis sinc(x w) w = sin(pi w x) / (pi x) (the form
   * whose roots may be trainable).  The host states which one the state_dict follows.  0 = recognise it from the
   * stored values (roots within 1e-5 relative of n*pi or of n); aa_model_pack_weights then FAILS on roots that match neither
   * (trained roots) instead of guessing. */
  int32_t bessel_convention;
} aa_model_config;

/* Raw parameters in the reference's own state_dict layout, HOST memory, float64.
 * MLP weights are [in,out] row-major, one pointer per linear layer. */
typedef struct {
  const double* rmax_recip;      /* [T,T]   edge_norm.rmax_recip                              */
  const double* bessel_weights;  /* [B]     radial_chemical_embed.bessel_encode.bessel_weights*/
  const double* center_embed;    /* [T,S0/2] ...type_embed.center_embed.weight               */
  const double* neighbor_embed;  /* [T,S0/2] ...type_embed.neighbor_embed.weight             */
  const double* basis_linear;    /* [B,S0]  ...type_embed.basis_linear.mlp.0.weight          */
  const double* embed_mlp[AA_MAX_MLP_LAYERS];   /* scalar_embed_mlp.mlp.mlp.{i}.weight        */
  const double* env_embed_linear;               /* [S,W] tensor_embed.env_embed_linear        */
  const double* first_proj;                     /* [S,S+W] allegro.first_layer_env_embed_projection */
  const double* latent[AA_MAX_LAYERS][AA_MAX_MLP_LAYERS]; /* allegro.latents.{l}.mlp.{i}.weight */
  const double* tp_weights[AA_MAX_LAYERS];      /* allegro.tps.{l}.weights                    */
  const double* readout[AA_MAX_MLP_LAYERS];     /* edge_readout.mlp.mlp.{i}.weight            */
  const double* scales;          /* [T] or NULL                                               */
  const double* shifts;          /* [T] or NULL                                               */
  const double* spline_weights;  /* embed_kind 1: [T*T, S0, num_splines] radial_chemical_embed.spline.class_embed.weight
                                    (the Bessel/type-embedding pointers above are then unused and may be NULL) */
} aa_model_raw_weights;

typedef struct {
  int64_t num_atoms;       /* N (real + ghost rows of pos)                                     */
  int64_t num_edges;       /* E directed edges, sorted by center                               */
  const int32_t* center;   /* [E] edge_index[0]                                                */
  const int32_t* nbr;      /* [E] edge_index[1]                                                */
  const int32_t* rowptr;   /* [N+1] CSR over centers                                           */
  const int32_t* types;    /* [N]                                                              */
  const void* shift_vec;   /* [E,3] cartesian periodic shift (model dtype) or NULL (ghost layout,
                              allegro/_compile.py:28-63)                                       */
  /* optional transposed CSR (edge ids grouped by NEIGHBOR atom): with it the forces are gathered per atom in a
   * fixed order (bit-reproducible, no atomics); NULL/NULL: neighbor contributions use floating-point atomics */
  const int32_t* t_rowptr; /* [N+1] or NULL                                                    */
  const int32_t* t_perm;   /* [E] edge ids sorted by neighbor (stable), or NULL                */
  /* optional hint: every center atom that has edges lies in [atom_begin, atom_end) -- the owned block of an
   * atom-block partition.  Per-atom kernels are then launched over that range only.  0,0 = all atoms.
   * The caller guarantees it (allegro_amd.nn.PreparedGraph derives it from rowptr). */
  int64_t atom_begin, atom_end;
  /* optional hint: no center atom has more than max_degree edges (0 = unknown).  With 0 < max_degree <= 128 the standard
   * 2-layer 64-wide fp32 stack runs the fused per-atom-tile forward (one launch instead of seven, 2.1 instead of 7.3 KB
   * of HBM traffic per edge; up to 32: one wave per atom; 33..128: teams of 2 / 4 waves per atom, dealt from class lists
   * that one extra small launch builds -- chosen where it is faster: small systems (<= 4096 tiles) or nearly full tiles;
   * see aa_plan_options.fused_forward and DESIGN.md section 9.1); otherwise the staged pipeline.  The caller states it
   * (allegro_amd.nn.PreparedGraph derives it from rowptr) and the kernels CHECK it: the reference accepts any segment length
   * (_contract.py:195-205), so a segment longer than the hint is never silently truncated -- that atom's energy comes back
   * NaN, the plan's host-visible status word is set, and the next aa_model_energy_forces on the plan (or aa_model_check)
   * returns AA_ERR_INVALID naming the degree that was found. */
  int64_t max_degree;
} aa_graph;

typedef struct aa_model_plan aa_model_plan;

/* Kernel-selection switches for A/B measurements and tests.  The library never reads the environment: a host that
 * wants switches passes them here (allegro_amd/_lib.py maps the AA_* environment variables of the Python host onto
 * this struct).  All zero = the defaults the plan would choose itself. */
typedef struct {
  int32_t tp_generic;      /* table-driven tensor-product kernels instead of the compile-time-CG ones          */
  int32_t tp_no_chain;     /* no 2-layer chain kernels                                                          */
  int32_t tp_no_moments;   /* no moments / per-atom operator kernels                                            */
  int32_t tp_no_operator;  /* no per-atom operator kernels (aa_tp_op.hip)                                       */
  int32_t tp_force_operator; /* operator kernels also where the 2-layer u = 64 kernels apply                   */
  int32_t tp_operator_fused; /* operator kernels in fused (not split) form                                      */
  int32_t gemm_no_chain;   /* single-layer linear kernels instead of the fused chains                           */
  int32_t gemm_fp32_mfma;  /* native fp32-input MFMA instead of bf16x3                                          */
  int32_t gemm_valu;       /* VALU linear layers                                                                */
  int32_t gemm_v1;         /* first-generation fp32 MFMA kernel                                                 */
  int32_t gemm_lds_epilogue; /* LDS-transposed epilogue in the single-layer bf16x3 kernel                       */
  int32_t f64_column_loop; /* fp64 linear layers: 0 automatic, 1 never, 2 always walk all column tiles per workgroup */
  int32_t embed_no_fuse;   /* reverse pass: materialise d(two-body embedding)                                   */
  int32_t fused_forward;   /* fused per-atom-tile forward: 0 / 1 whenever aa_graph.max_degree allows; 3 never (staged pipeline);
                            * A/B: 2 / 4 = for every graph with segments <= 128 in the pure team / the mixed form */
  int32_t fused_recompute_w0; /* fused forward: recompute w0 for the second layer instead of holding it         */
  int32_t moments_waves_per_block; /* 0 = 1                                                                      */
  int32_t f64_rows;        /* fp64 linear layers, row-resident kernels (operand rows read once): 0 where measured faster, 1 wherever applicable, 2 off */
  int32_t no_channel_padding; /* stacks whose channel count is not a multiple of 64, or with single hidden layers narrower than
                               * 64, are normally evaluated zero-padded (same results, tuned kernels); 1: keep them narrow */
  int32_t fused_tail;      /* experimental builds only (AA_BUILD_EXPERIMENTAL=1): 1 = the fused per-atom-tile
                            * reverse tail (aa_fused_bwd.hip) where the fused forward runs, 2 = ... leaving the edge reverse to
                            * edge_backward; measured slower than the staged tail on MI355X, ignored by the product build      */
  int32_t fused_keep_split; /* fused forward, one-tile form: tile pairs that feed several layers are split into their bf16 levels once and
                             * held in registers: 0 = the default (2), 1 = none (parked raw in LDS, split by every reader), 2 = the two-body
                             * scalars, 3 = two-body scalars and lat0 */
  int32_t poison_workspace; /* debugging: every step first fills the whole workspace with 0xFF bytes (NaN in fp32 and fp64), so
                             * that a kernel reading a cell no earlier kernel of the SAME step wrote shows up as NaN       */
  int32_t no_slot_form;     /* operator-kernel plans (e.g. fp64, l_max 3, 3 layers): 1 = the unfolded single-layer pipeline instead of
                             * the slot form (output layers of scalar_embed_mlp / the latent MLPs folded into their consumers, reverse
                             * pass evaluated per dense-net slot) -- same results, A/B and tests                                   */
  int32_t op_proj_gemm;     /* operator-kernel plans: the env projections (x2s = f M Wenv and its reverse) as batched linear-layer launches
                             * over all atoms instead of inside the per-atom kernels: 0 = from 4096 atoms on, 1 = always, 2 = never       */
  int32_t op_env_vector;    /* operator-kernel plans, fp64: 1 = the adjoint of the moments on the edges in its vector form instead of on the
                             * f64 matrix cores (A/B, tests)                                                                       */
  int32_t staged_no_fold;   /* staged fp32 pipeline (graphs the fused forward does not take): 1 = forward chains with the reference's own layers
                             * instead of the folded ones (A/B, tests)                                                             */
  int32_t op_recompute_bvecs; /* operator-kernel plans: 1 = the layer-0 reverse recomputes the per-atom vectors B_l instead of reading
                               * the ones the forward kernels of the same step stored (A/B, tests)                                 */
  int32_t readout_two_pass; /* single-layer pipeline: 1 = d E / d (readout hidden layer) by its own kernel in the reverse pass instead of
                             * by the forward's energy reduction, which reads the same rows (A/B, tests)                          */
  int32_t tp_prefer_moments; /* 2-layer u = 64 stacks the fused chains do not cover (fp64; S or MLP widths of 128): 1 = the 2-layer moments
                              * kernels + single linear layers (the selection up to round 4) instead of the operator kernels (A/B, tests) */
  int32_t fused_narrow;     /* fused forward, one-tile pass, where the two-waves-per-SIMD form applies (aa_fused8.hip: one species, folded
                             * program): 0 = that form as two independent four-wave workgroups per CU, 2 = as one eight-wave workgroup per CU
                             * (lock step), 1 = the one-wave-per-SIMD kernel of rounds 2-5 (A/B, tests).  Boxes of at most 4 atoms per CU take the
                             * round 2-5 kernel under 0 (every CU holds at most one workgroup anyway); 3 = the four-wave form there too (tests); 5 = 3 with the env
                             * projections as bf16x3 layers on the matrix cores (A/B: measured slower) */
  int32_t chain_staged_weights; /* one-layer reverse chains: 1 = the per-workgroup weight staging of the general chain kernel also where the
                                 * persistent form with LDS-resident weights applies (round 6; A/B, tests)                               */
} aa_plan_options;

int aa_model_plan_create(const aa_model_config* cfg, aa_model_plan** out);
int aa_model_plan_create_with_options(const aa_model_config* cfg, const aa_plan_options* options, aa_model_plan** out);
void aa_model_plan_destroy(aa_model_plan* plan);
/* on != 0: aa_model_energy_forces captures its launch sequence into a hipGraph the first time it sees a set of
 * arguments (all pointers and sizes) and replays it with one hipGraphLaunch afterwards -- for launch-bound (small)
 * systems in MD loops whose buffers stay put.  The plan then carries mutable state: one caller thread per plan. */
int aa_model_plan_enable_graph(aa_model_plan* plan, int on);
/* Forward / reverse hand-over points for hosts that PIPELINE atom blocks of one frame on several streams (one plan + workspace per
 * block in flight): `wait_event` (a hipEvent_t, nullable) is waited for on the step's stream before the first launch of the forward,
 * `record_event` (nullable) is recorded after its last launch, before the reverse pass.  With block i's forward waiting for block
 * i-1's record_event the forwards of consecutive blocks run one after the other while each block's reverse pass overlaps the next
 * block's forward (allegro_amd.nn.PipelinedStep).  Persistent until set again; not captured by aa_model_plan_enable_graph. */
int aa_model_plan_set_forward_events(aa_model_plan* plan, void* wait_event, void* record_event);
/* on != 0: every step materialises the per-edge intermediates that aa_model_debug_tap exposes (the staged pipeline is
 * used; the fused kernels keep them on chip).  Parity tests only. */
int aa_model_plan_enable_taps(aa_model_plan* plan, int on);

/* size of the packed device weight blob, and packing (host fp64 -> device model dtype, with the
 * ScalarMLPFunction normalisation constants folded and the two linear maps of the first stage
 * fused); call again whenever the parameters change */
size_t aa_model_weights_bytes(const aa_model_plan* plan);
/* 64-bit digest of the packed blob's layout (every offset, the padded shapes, the column orders chosen by the
 * kernel-selection options).  A blob packed by one plan may be consumed by another plan only when the digests agree:
 * the exported op (csrc/torch_ops.cpp) stores it next to the hyper-parameters and refuses a mismatch. */
uint64_t aa_model_plan_layout_hash(const aa_model_plan* plan);
int aa_model_pack_weights(const aa_model_plan* plan, const aa_model_raw_weights* raw, void* dev_blob,
                          size_t blob_bytes, aa_stream stream);

size_t aa_model_workspace_bytes(const aa_model_plan* plan, int64_t num_atoms, int64_t num_edges,
                                int with_forces);

/* atom_energy: [N] (model dtype); forces: [N,3] or NULL (energy only).  forces are written
 * (not accumulated); ghost rows receive their own contributions (LAMMPS reverse-communicates). */
int aa_model_energy_forces(const aa_model_plan* plan, const void* dev_weights, const aa_graph* graph,
                           const void* pos, void* workspace, size_t workspace_bytes,
                           void* atom_energy, void* forces, aa_stream stream);

/* One-line JSON description of what the plan runs (which forward, which algebraic folds, MFMA steps the fused forward executes
 * per 32-edge tile against the step-equivalents of the reference's layers) -- for benchmark lines and bug reports.  Returns the
 * length written (excluding the terminator), or < 0. */
int aa_model_plan_describe(const aa_model_plan* plan, char* buf, size_t buf_bytes);

/* Synchronises `stream` and reports whether any step enqueued on this plan since the last report contradicted the graph
 * hints it was given (aa_graph.max_degree too small, center atoms with edges outside [atom_begin, atom_end)):
 * AA_ERR_INVALID + aa_last_error() then, AA_OK otherwise; the condition is cleared.  aa_model_energy_forces performs the
 * same test (without synchronising) on entry, so a host that never calls this still gets the error one step late. */
int aa_model_check(const aa_model_plan* plan, aa_stream stream);

/* same as aa_model_energy_forces, additionally timing every kernel launch of the pass with HIP events
 * recorded on `stream` (synchronises the stream before returning).  stage_names is a
 * [max_stages][32] char buffer; stage i is the i-th launch, named after its kernel.  stage_bytes /
 * stage_flops (each [max_stages], may be NULL) receive the launch's ALGORITHMIC work: every distinct
 * operand row it must read or write once, and 2*M*K*N per GEMM layer. */
int aa_model_energy_forces_profiled(const aa_model_plan* plan, const void* dev_weights, const aa_graph* graph,
                                    const void* pos, void* workspace, size_t workspace_bytes,
                                    void* atom_energy, void* forces, aa_stream stream, int max_stages,
                                    float* stage_ms, char* stage_names, int* num_stages, double* stage_bytes,
                                    double* stage_flops);

/* strain derivative of the total energy from the per-edge data the LAST aa_model_energy_forces call (with forces)
 * left in `workspace`:  W[a][b] = dE/d eps_ab = sum_e (dE/dr_e)_a (r_e)_b, 3x3 row-major, model dtype, device memory.
 * stress = W / volume (nequip ForceStressOutput convention); LAMMPS' virial is -W. */
int aa_model_virial(const aa_model_plan* plan, const aa_graph* graph, void* workspace, size_t workspace_bytes,
                    void* virial9, aa_stream stream);

/* debug/parity taps: copy an intermediate of the LAST call out of the workspace layout.
 * name in {"edge_attrs","edge_embedding","edge_features","emb0","vec"}; a "+f" suffix selects the workspace layout of
 * a step that computed forces (required for "dvec": dE/dr_e [E,4]); returns elements per edge or <0 */
int aa_model_debug_tap(const aa_model_plan* plan, const char* name, int64_t num_atoms, int64_t num_edges,
                       const void* workspace, const void** ptr, int64_t* ld);

/* ---------------------------------------------------------------------------------------------
 * 3. On-device neighbor list (cell list) -> the center-sorted CSR graph above, without leaving the GPU.
 *    In the reference stack the edge list is built on the host (nequip's neighbor-list transform, EXT) or handed
 *    over by LAMMPS (pair_allegro); conventions of with_edge_vectors_ (tensorembed.py:86):
 *        r_e = pos[nbr] - pos[center] + cell_shift_e @ cell,  |r_e| < r_cut,
 *    every image within r_cut is listed (cells smaller than 2 r_cut included); an atom pairs with itself only
 *    through a non-zero shift.  Edges come out sorted by center, neighbors in a reproducible order.
 *    Two phases because E is only known after counting:
 *      aa_nl_count  bins the atoms, counts, writes rowptr[N+1] (device) and returns E (synchronises the stream);
 *      aa_nl_fill   writes center/nbr [E] and (optionally) cell_shift int32 [E,3] and shift_vec [E,3] (model dtype)
 *                   from the state aa_nl_count left in `workspace` (same input, same workspace).
 * ------------------------------------------------------------------------------------------ */
typedef struct {
  int64_t num_atoms;
  const void* pos;         /* [N,3] device, `dtype`                                            */
  double cell[9];          /* lattice vectors as rows (pos = frac @ cell), host values          */
  int32_t pbc[3];          /* periodic along a, b, c                                           */
  int32_t dtype;           /* aa_dtype of pos / shift_vec                                      */
  double r_cut;
} aa_nl_input;
size_t aa_nl_workspace_bytes(int64_t num_atoms);
int aa_nl_count(const aa_nl_input* in, void* workspace, size_t workspace_bytes, int32_t* rowptr, int64_t* num_edges,
                aa_stream stream);
int aa_nl_fill(const aa_nl_input* in, void* workspace, size_t workspace_bytes, const int32_t* rowptr, int32_t* center,
               int32_t* nbr, int32_t* cell_shift, void* shift_vec, aa_stream stream);

/* Transposed CSR of a center-sorted edge list on the device -- `t_rowptr` [N+1], `t_perm` [E]: edge ids grouped by NEIGHBOUR atom in
 * ascending order (= the stable argsort of `nbr`), what aa_graph.t_rowptr / t_perm ask for so that forces are gathered per atom in a
 * fixed order -- and, with `hints3` != NULL (device int32[3]; needs `rowptr`), the three graph hints {atom_begin, atom_end,
 * max_degree} from the row pointers.  Counting sort by atomics followed by a per-atom sort of the groups: the result does not depend
 * on the order the atomics were served in.  A host without a tensor library (LAMMPS) gets deterministic forces with this one call
 * per neighbour list; `workspace`: aa_graph_transpose_workspace_bytes(num_atoms) bytes of device scratch. */
size_t aa_graph_transpose_workspace_bytes(int64_t num_atoms);
int aa_graph_transpose(int64_t num_atoms, int64_t num_edges, const int32_t* rowptr, const int32_t* nbr, int32_t* t_rowptr, int32_t* t_perm,
                       int32_t* hints3, void* workspace, size_t workspace_bytes, aa_stream stream);

/* 128-bit content fingerprint of a neighbour list as the pair_allegro contract hands it over (allegro/_compile.py:10-14):
 * edge_index int64 [2,E] (second row at edge_index + row_stride) and atom_types [N] (int64 or int32), both in device
 * memory; fp2: uint64[2] in device memory, overwritten.  Position-dependent (a permutation of the edges changes it) and
 * bit-reproducible.  For hosts that cache the CSR of a list by tensor identity: csrc/torch_ops.cpp compares it on the
 * device on every cache hit, so contents rewritten through a raw pointer are noticed (its outputs turn NaN, the next call fails). */
int aa_graph_fingerprint(const int64_t* edge_index, int64_t row_stride, int64_t num_edges, const void* atom_types,
                         int types_are_int64, int64_t num_atoms, uint64_t* fp2, aa_stream stream);

/* ---------------------------------------------------------------------------------------------
 * 5. Host-side model files: what a Python-free host (a LAMMPS pair style, the C driver of INTEGRATION.md section 3) passes
 *    to aa_model_plan_create / aa_model_pack_weights, read from ONE flat file that allegro_amd.export.write_host_model
 *    writes from a model holding a reference checkpoint (same state_dict keys as allegro.model.AllegroModel,
 *    allegro/model/allegro_models.py:112-300): hyper-parameters, the Clebsch-Gordan non-zeros of every layer
 *    (Contracter.__init__, _contract.py:95-119) and every parameter as float64 in the reference's own layout.
 *    Pure host code, no GPU needed.  The returned pointers live until aa_model_file_close.
 *    aa_model_file_from_words parses the int64 word list alone (the `config` argument of the exported dispatcher op,
 *    csrc/torch_ops.cpp; weights all NULL then).  File layout: csrc/aa_hostfile.hip.
 * ------------------------------------------------------------------------------------------ */
typedef struct aa_model_file aa_model_file;
int aa_model_file_open(const char* path, aa_model_file** out);
int aa_model_file_from_words(const int64_t* words, int64_t num_words, aa_model_file** out);
const aa_model_config* aa_model_file_config(const aa_model_file* file);
const aa_model_raw_weights* aa_model_file_weights(const aa_model_file* file);
uint64_t aa_model_file_layout_digest(const aa_model_file* file); /* aa_model_plan_layout_hash of the plan the exporter packed for; 0: none */
void aa_model_file_close(aa_model_file* file);

/* ---------------------------------------------------------------------------------------------
 * 4. Debug entry points (parity tests; not part of the drop-in surface).
 *    aa_debug_gemm_f32: C[M,N] = A[M,K] @ W[K,N] (A, C device fp32 row-major; W HOST fp32 row-major) through one of
 *    the fp32 linear-layer kernels of the scalar MLPs: 0 = bf16x3 split-precision MFMA, 1 = native fp32-input MFMA,
 *    2 = the fused-chain kernel (one layer), 3 = VALU.  K and N multiples of 32.  Allocates temporaries and
 *    synchronises the stream: test use only.
 * ------------------------------------------------------------------------------------------ */
int aa_debug_gemm_f32(int kernel, int64_t M, int K, int N, const float* A_dev, const float* W_host, float* C_dev,
                      aa_stream stream);

#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility pop
#endif

#ifdef __cplusplus
}
#endif
#endif /* ALLEGRO_AMD_H */
