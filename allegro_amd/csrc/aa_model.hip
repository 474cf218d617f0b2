// Orchestration of the whole Allegro hot path behind the C ABI (include/allegro_amd.h):
// forward (allegro/model/allegro_models.py:222-297 module chain) and the hand-written reverse pass
// w.r.t. positions (what ForceStressOutput's autograd does, allegro_models.py:101-103).
// All kernels are enqueued on the caller's stream; all tensors live in caller-owned HBM buffers.
#include <algorithm>

#include "aa_common.h"

namespace aa {
static thread_local std::string g_err;
void set_error(const std::string& msg) { g_err = msg; }
int fail(int code, const std::string& msg) {
  g_err = msg;
  return code;
}
}  // namespace aa

using namespace aa;

// ------------------------------------------------------------------------------------------------
// TP operator plan (seam B1/B2)
// ------------------------------------------------------------------------------------------------
struct aa_tp_plan {
  aa_dtype dtype;
  TpLayerDev dev;
  std::vector<void*> owned;
  int spec_sig = -1;           // generated signature of this layer (aa_cg_gen.h) or -1
  bool dense_spec = false;     // forward / input gradients on the specialised dense-operand kernels (aa_tp_dense.hip)
};

extern "C" const char* aa_last_error(void) { return g_err.c_str(); }
extern "C" int aa_version(void) { return 1; }

extern "C" int aa_tp_plan_create(const aa_tp_desc* desc, aa_dtype dtype, aa_tp_plan** out) {
  AA_REQUIRE(desc && out, "aa_tp_plan_create: null argument");
  AA_REQUIRE(dtype == AA_F32 || dtype == AA_F64, "aa_tp_plan_create: bad dtype");
  aa_tp_plan* p = new aa_tp_plan();
  p->dtype = dtype;
  int rc = build_tp_layer(*desc, &p->dev, &p->owned);
  if (rc) {
    for (void* q : p->owned) (void)hipFree(q);
    delete p;
    return rc;
  }
  p->spec_sig = find_spec_sig(*desc);
  p->dense_spec = tp_dense_supported(p->spec_sig, desc->mul, dtype);
  *out = p;
  return AA_OK;
}

extern "C" int aa_tp_plan_use_general_kernels(aa_tp_plan* plan, int on) {
  AA_REQUIRE(plan, "aa_tp_plan_use_general_kernels: null plan");
  plan->dense_spec = !on && tp_dense_supported(plan->spec_sig, plan->dev.mul, plan->dtype);
  return AA_OK;
}

extern "C" int aa_tp_plan_is_specialised(const aa_tp_plan* plan) { return plan && plan->dense_spec ? 1 : 0; }

extern "C" void aa_tp_plan_destroy(aa_tp_plan* plan) {
  if (!plan) return;
  for (void* q : plan->owned) (void)hipFree(q);
  delete plan;
}

extern "C" int aa_tp_forward(const aa_tp_plan* plan, int64_t E, int64_t N, const void* x1, const void* x2,
                             const void* weights, const int32_t* rowptr, const int32_t* eids, double scatter_factor,
                             void* x2s, void* out, aa_stream stream) {
  AA_REQUIRE(plan && x1 && x2 && weights && rowptr && x2s && out, "aa_tp_forward: null argument");
  if (plan->dense_spec) {
    TpDenseArgs d{};
    d.E = E;
    d.N = N;
    d.rowptr = rowptr;
    d.eids = eids;
    d.u = plan->dev.mul;
    d.coupling = plan->dev.coupling;
    d.sf = scatter_factor;
    d.x1 = x1;
    d.x2 = x2;
    d.weights = weights;
    d.x2s = x2s;
    d.out = out;
    hipStream_t st = static_cast<hipStream_t>(stream);
    return plan->dtype == AA_F32 ? launch_tp_dense<float>(plan->spec_sig, false, d, st) : launch_tp_dense<double>(plan->spec_sig, false, d, st);
  }
  TpLayerFwdArgs a{};
  a.E = E;
  a.N = N;
  a.rowptr = rowptr;
  a.eids = eids;
  a.x1.dense = x1;
  a.x2.dense = x2;
  a.weights = weights;
  a.scatter_factor = scatter_factor;
  a.x2s = x2s;
  a.out = out;
  hipStream_t s = static_cast<hipStream_t>(stream);
  return plan->dtype == AA_F32 ? launch_tp_layer_fwd<float>(plan->dev, a, s) : launch_tp_layer_fwd<double>(plan->dev, a, s);
}

extern "C" int aa_tp_backward(const aa_tp_plan* plan, int64_t E, int64_t N, const void* x1, const void* x2s,
                              const void* weights, const int32_t* rowptr, const int32_t* eids, double scatter_factor,
                              const void* gout, void* gx1, void* gx2, aa_stream stream) {
  AA_REQUIRE(plan && weights && rowptr && gout && (gx1 || gx2), "aa_tp_backward: null argument");
  AA_REQUIRE((!gx1 || x2s) && (!gx2 || x1), "aa_tp_backward: a requested gradient's operand is null");
  if (plan->dense_spec) {
    TpDenseArgs d{};
    d.E = E;
    d.N = N;
    d.rowptr = rowptr;
    d.eids = eids;
    d.u = plan->dev.mul;
    d.coupling = plan->dev.coupling;
    d.sf = scatter_factor;
    d.x1 = x1;
    d.weights = weights;
    d.x2s = const_cast<void*>(x2s);
    d.gout = gout;
    d.gx1 = gx1;
    d.gx2 = gx2;
    hipStream_t st = static_cast<hipStream_t>(stream);
    return plan->dtype == AA_F32 ? launch_tp_dense<float>(plan->spec_sig, true, d, st) : launch_tp_dense<double>(plan->spec_sig, true, d, st);
  }
  TpLayerBwdArgs a{};
  a.E = E;
  a.N = N;
  a.rowptr = rowptr;
  a.eids = eids;
  a.x1.dense = x1;
  a.weights = weights;
  a.scatter_factor = scatter_factor;
  a.x2s = x2s;
  a.gout = gout;
  a.g1.dense = gx1;
  a.g2.dense = gx2;
  hipStream_t s = static_cast<hipStream_t>(stream);
  return plan->dtype == AA_F32 ? launch_tp_layer_bwd<float>(plan->dev, a, s) : launch_tp_layer_bwd<double>(plan->dev, a, s);
}

extern "C" int aa_tp_segment_sum(aa_dtype dtype, int64_t E, int64_t N, int64_t row_elems, const void* x, const int32_t* rowptr,
                                 const int32_t* eids, double scale, void* out, aa_stream stream) {
  AA_REQUIRE(rowptr && (N == 0 || row_elems == 0 || out) && (E == 0 || x), "aa_tp_segment_sum: null argument");
  AA_REQUIRE(dtype == AA_F32 || dtype == AA_F64, "aa_tp_segment_sum: dtype");
  hipStream_t s = static_cast<hipStream_t>(stream);
  return dtype == AA_F32 ? launch_segment_sum<float>(N, row_elems, x, rowptr, eids, scale, out, s)
                         : launch_segment_sum<double>(N, row_elems, x, rowptr, eids, scale, out, s);
}

extern "C" size_t aa_tp_weights_workspace_bytes(const aa_tp_plan* plan, int64_t N) {
  if (!plan) return 0;
  const size_t elem = plan->dtype == AA_F32 ? 4 : 8;
  if (plan->dense_spec) return size_t(tp_wgrad_slots(N, kDenseWgradSlots) + 1) * plan->dev.mul * plan->dev.num_paths * elem;
  return tp_layer_wgrad_workspace_elems(plan->dev, N) * elem;
}

extern "C" int aa_tp_backward_weights(const aa_tp_plan* plan, int64_t E, int64_t N, const void* x1, const void* x2s,
                                      const int32_t* rowptr, const int32_t* eids, const void* gout, void* workspace,
                                      size_t workspace_bytes, void* gweights, aa_stream stream) {
  AA_REQUIRE(plan && rowptr && gweights, "aa_tp_backward_weights: null argument");
  AA_REQUIRE(E == 0 || (x1 && x2s && gout), "aa_tp_backward_weights: null tensor");
  AA_REQUIRE(workspace_bytes >= aa_tp_weights_workspace_bytes(plan, N) && (workspace || workspace_bytes == 0),
             "aa_tp_backward_weights: workspace too small");
  TpLayerWgradArgs a{};
  a.E = E;
  a.N = N;
  a.rowptr = rowptr;
  a.eids = eids;
  a.x1 = x1;
  a.x2s = x2s;
  a.gout = gout;
  a.partial = workspace;
  a.gw = gweights;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (plan->dense_spec)
    return plan->dtype == AA_F32 ? launch_tp_dense_wgrad<float>(plan->spec_sig, plan->dev.mul, plan->dev.coupling, a, s)
                                 : launch_tp_dense_wgrad<double>(plan->spec_sig, plan->dev.mul, plan->dev.coupling, a, s);
  return plan->dtype == AA_F32 ? launch_tp_layer_wgrad<float>(plan->dev, a, s) : launch_tp_layer_wgrad<double>(plan->dev, a, s);
}

// ------------------------------------------------------------------------------------------------
// model plan
// ------------------------------------------------------------------------------------------------
struct MlpLayout {
  std::vector<int> dims;        // d_0 .. d_n
  std::vector<size_t> w, wt;    // blob offsets (elements) of W_i [d_i,d_{i+1}] and its transpose
  std::vector<size_t> wp, wtp;  // the same two matrices in MFMA fragment order (aa_gemm.hip v3)
  std::vector<size_t> wq, wtq;  // ... and split into 3 bf16 levels (bf16x3 path; fp32 plans only)
};

// one linear layer's matrix in the three forms the single-layer kernels take (row-major, MFMA fragment order, bf16x3 levels)
struct GemmMat {
  size_t w = 0, wp = 0, wq = 0;
  int K = 0, N = 0;
};

// a set of equally shaped matrices at uniform strides (GemmArgs::batch)
struct GemmMatSet {
  size_t w = 0, wp = 0, wq = 0, w_bs = 0, wp_bs = 0, wq_bs = 0;
  int K = 0, N = 0, count = 0;
};

struct aa_model_plan {
  aa_model_config cfg;
  aa_plan_options opt{};
  int D, R, W, SL1;  // SH dim, irreps, env weight numel, S*(L+1)
  std::vector<std::vector<int32_t>> keep_i32;
  std::vector<std::vector<double>> keep_f64;
  std::vector<TpLayerDev> layers;
  std::vector<void*> owned;
  // weight blob layout (element offsets)
  size_t o_rmax, o_bessel, o_cemb, o_nemb, o_basis, o_g0, o_g0t, o_g0p, o_g0tp, o_g0q, o_g0tq, o_b3a_q, o_b3b_q, o_b3c_q, o_ro_last, o_scales, o_shifts, n_elems;
  size_t o_tpw[AA_MAX_LAYERS];
  MlpLayout embed, readout;          // readout: only the GEMM layers (all but the final ->1 layer)
  MlpLayout latent[AA_MAX_LAYERS];
  int ro_last_dim;                   // input dim of the final readout linear
  int spec_sig[AA_MAX_LAYERS];       // generated-signature id per layer, or -1
  bool use_spec;                     // all layers specialised -> channel-minor internal layouts
  int u_raw;                         // num_tensor_features of the model; cfg.num_tensor is the next multiple of 64 when the channels were padded
  int hid_raw[3];                    // hidden widths of scalar_embed_mlp / latent MLPs / edge_readout in the model (cfg holds the padded ones)
  int chain_pair;                    // >= 0: 2-layer stack on the chain kernels (no [E,u,D] tensors in HBM)
  bool env_mom;                      // env weights through per-atom moments: no [E,R*u] env tensors (TpMomArgs / TpOpArgs)
  int tp_op;                         // >= 0: signature chain of the per-atom operator kernels (aa_tp_op.hip; any L <= 3, u = 64 m)
  bool chain_gemm;                   // MLP chains fused into gemm_chain_bf16x3_kernel (hidden layers stay in registers)
  bool fused_fwd;                    // the whole forward as ONE per-atom-tile kernel when the graph allows (aa_fused.hip)
  bool fused_hold_w0;                // ... holding the w0 tiles in registers between the two layers (else: recomputed)
  bool fused_wide = false;           // ... its one-tile pass on the eight-wave form (two waves per SIMD, aa_fused8.hip)
  mutable bool taps = false;         // aa_model_plan_enable_taps: staged pipeline so that every tap is materialised
  bool embed_fused;                  // reverse pass: d(two-body embedding) [E,S0] never materialised, the last reverse chain
                                     // contracts it back to the 8 basis functions in its epilogue (embrev_out in aa_common.h)
  size_t o_embtab;                   // [T*T][8][64] type_embed(c | pair) * basis_linear[n][c]
  size_t o_embtab_h;                 // [T*T][8][64] the same table times the first scalar_embed_mlp layer (kFoldEmbed), or 0
  size_t o_lat1in_fq, o_ro0_fq;      // (kFoldLatent, 2-layer 64-wide stacks) bf16x3 copies of the first layers of latent 1 / edge_readout with the
                                     // latent output layers folded into their lat_l row blocks, or 0
  size_t o_b3af_q;                   // ... and of the merged first layer of the readout-reverse chain, or 0
  size_t o_b3bf_q;                   // ... and of its second layer with Wout_0^T folded into the lat0 columns (d a_0 instead of d lat0), or 0
  size_t o_g0fq, o_g0tfq;            // (kFoldEmb1) bf16x3 copies of W1 @ G0 [64, ng0] and of its transpose, or 0
  size_t o_wk0f, o_wt0f;             // (kFoldEmb1) W1 @ Wenv0 as [k][R][u] and [R][u][k], or 0
  size_t o_wkq[AA_MAX_LAYERS];       // (kProjMfma, fused forward) Wenv_l as R bf16x3 64x64 layers [r][k -> ch] (layer 0: the folded one), or 0
  // "Slot form" of the single-layer pipeline (operator-kernel plans: C5 and every standard stack off the tuned 2-layer shape).
  // The same algebra as kFoldEmb1 / kFoldLatent, for any depth: the output layer of scalar_embed_mlp and of every latent MLP is
  // folded into its consumers at pack time, so slot l + 1 of the dense-net buffer holds the latent's hidden PRE-ACTIVATION z_l
  // (consumers activate those columns on load, GemmArgs::act_lo / act_hi) and the layers "a -> lat_l" do not exist.  The reverse
  // is evaluated BY SLOT, not by consumer: d z_l = ([d readout hidden | d z_{l+1} .. d z_{L-1}] @ stack_l + d a_l of the moments)
  // * act'(z_l) -- one K-stacked layer with 128-wide output per latent (the accumulator-resident kernel's shape, every slot
  // written once) instead of wide accumulate-into-the-dense-net layers.
  bool slot_form;
  GemmMat s_g0f, s_g0ft;             //   W_last(embed) @ G0 [He, ng0] and its transpose
  GemmMat s_in[AA_MAX_LAYERS];       //   first layer of latent l with the row blocks of earlier latents folded  [S (l+1) + u, H]
  GemmMat s_ro0;                     //   first readout layer, folded                                             [SL1, Hr]
  GemmMat s_rs[AA_MAX_LAYERS];       //   reverse stack of slot l + 1: rows [readout | latent l+1 .. L-1] -> d a_l  [Hr + S (L-1-l), H]
  GemmMat s_rstb;                    //   reverse stack of slot 0 (two-body): rows [readout | latent 0 .. L-1]      [Hr + S L, S]
  GemmMat s_sct[AA_MAX_LAYERS];      //   d z_l -> d (tensor scalars of layer l)                                   [H, u]
  size_t o_s_wk0, o_s_wt0;           //   W_last(embed) @ Wenv0 as [k][R][u] and [R][u][k]
  // operator-kernel plans: the env projections as batched linear-layer launches (TpOpArgs::proj_gemm) -- per layer the R matrices
  // f Wenv_l[:, r, :] [ka, u] and their transposes [u, ka]; layer 0 also behind the output layer of scalar_embed_mlp (slot form)
  bool op_proj;
  GemmMatSet s_pr[AA_MAX_LAYERS], s_prt[AA_MAX_LAYERS], s_pr0f, s_prt0f;
  int ng0;                           // output width of the fused first-stage GEMM
  size_t o_wk[AA_MAX_LAYERS], o_wt[AA_MAX_LAYERS];  // Wenv of layer l as [ka][R][u] and [R][u][ka]
#ifdef AA_EXPERIMENTAL_TAIL
  size_t o_wtk[AA_MAX_LAYERS];                      // ... and [u][R][ka] (the fused reverse tail streams it in 16-channel blocks)
#endif
  bool fused_tail;                   // (experimental build only) reverse: layer-0 tensor product reverse + first-stage /
                                     // embed-MLP reverse + edge reverse as ONE per-atom-tile kernel (aa_fused_bwd.hip); measured slower than the staged tail
  size_t esize() const { return cfg.dtype == AA_F32 ? 4 : 8; }
  // optional hipGraph replay of the whole step (aa_model_plan_enable_graph): the launch sequence is captured once per
  // distinct argument set and replayed with one hipGraphLaunch -- for launch-bound (small) systems
  mutable void* ev_wait = nullptr;    // aa_model_plan_set_forward_events
  mutable void* ev_record = nullptr;
  struct StepGraph {
    bool enabled = false;
    hipStream_t cap_stream = nullptr;
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    struct Key {
      const void *weights, *pos, *ws, *e, *f, *center, *nbr, *rowptr, *types, *shift, *trow, *tperm;
      int64_t N, E, a0, a1;
      size_t wsb;
      int64_t max_degree;  // selects the fused forward: part of what was captured
      int64_t taps;        // aa_model_plan_enable_taps switches pipelines
      bool operator==(const Key& o) const { return std::memcmp(this, &o, sizeof(Key)) == 0; }
    } key{};
  };
  mutable StepGraph sg;
  // Host-visible status word (pinned host memory, written by kernels, read by the host without a synchronisation): a step
  // whose graph contradicts the caller's hints (aa_graph.max_degree, atom_begin / atom_end) sets it, the offending atoms'
  // energies become NaN, and the NEXT aa_model_energy_forces on the plan -- or aa_model_check, which synchronises first --
  // returns AA_ERR_INVALID.  > 0: degree of a center atom beyond max_degree; -2: edges outside [atom_begin, atom_end).
  int32_t* status = nullptr;
};

// reads and clears the status word; AA_OK when clean
static int consume_status(const aa_model_plan* plan, const char* who) {
  if (!plan->status) return AA_OK;
  // read-and-clear in ONE atomic exchange: a violation a kernel of another stream writes between a separate read and clear would be
  // lost.  (The word is per plan: with several host threads stepping one plan a violation is reported to whichever of them looks
  // first -- include/allegro_amd.h, aa_model_check -- so a host that needs attribution uses one plan per stepping thread.)
  if (*reinterpret_cast<volatile int32_t*>(plan->status) == 0) return AA_OK;
  const int32_t v = __atomic_exchange_n(plan->status, 0, __ATOMIC_ACQ_REL);
  if (v == 0) return AA_OK;
  char msg[256];
  if (v > 0)
    snprintf(msg, sizeof msg, "%s: a step on this plan met a center atom with %d edges, more than aa_graph.max_degree promised "
             "(its energy was set to NaN; pass the true maximum, or 0 for \"unknown\")", who, int(v));
  else
    snprintf(msg, sizeof msg, "%s: a step on this plan met edges whose center lies outside [aa_graph.atom_begin, atom_end)", who);
  return fail(AA_ERR_INVALID, msg);
}

static std::vector<int> mlp_dims(int in, int depth, int width, int out) {
  std::vector<int> d{in};
  for (int i = 0; i < depth; ++i) d.push_back(width);
  d.push_back(out);
  return d;
}

extern "C" int aa_model_plan_create(const aa_model_config* cfg, aa_model_plan** out) {
  return aa_model_plan_create_with_options(cfg, nullptr, out);
}

extern "C" int aa_model_plan_create_with_options(const aa_model_config* cfg_in, const aa_plan_options* options, aa_model_plan** out) {
  AA_REQUIRE(cfg_in && out, "aa_model_plan_create: null argument");
  const aa_plan_options opt = options ? *options : aa_plan_options{};
  // Channel padding.  A stack whose tensor-channel count is not a multiple of 64 (16 <= u < 256) is evaluated as the
  // next multiple-of-64 stack whose extra channels have zero weights: env_embed_linear / the env columns of first_proj and of the latent outputs / the scalar rows of
  // the latent inputs are zero-padded at pack time (aa_model_pack_weights), so the padded channels carry exact zeros
  // through every layer and the results are those of the narrow model -- which thereby runs the tuned 64-channel
  // kernels (moments or per-atom operator kernels, fused chains, fused forward) instead of the per-edge ones (BASELINE config 0, u = 32: 26 launches
  // and 0.40 ms per step on 64 atoms without, 15 launches with).  Only where the 64-channel stack takes that path.
  aa_model_config cfg_local = *cfg_in;
  const int u_raw = cfg_in->num_tensor;
  {
    const aa_model_config& q = *cfg_in;
    const bool silu = q.act_kind[0] == AA_ACT_SILU && q.act_kind[1] == AA_ACT_SILU && q.act_kind[2] == AA_ACT_SILU;
    const int u_pad = (u_raw + 63) / 64 * 64;  // 16..63 -> 64 (moments / operator kernels), 65..127 -> 128, ... (operator kernels)
    const bool pad = !opt.no_channel_padding && u_raw >= 16 && u_raw != u_pad && u_pad <= 256 && silu && (q.num_layers == 2 || q.num_layers == 3) &&
                     (q.num_scalar == 64 || q.num_scalar == 128) && q.latent_mlp_depth >= 1 &&
                     (q.latent_mlp_width == 64 || q.latent_mlp_width == 128 || (q.latent_mlp_depth == 1 && q.latent_mlp_width >= 8 && q.latent_mlp_width < 64)) &&
                     !opt.tp_generic && !opt.tp_no_chain && !opt.tp_no_moments;
    if (pad) {
      cfg_local.num_tensor = u_pad;
      for (int l = 0; l < q.num_layers && l < AA_MAX_LAYERS; ++l) cfg_local.tps[l].mul = u_pad;
    }
    // Hidden-width padding, same idea: a single hidden layer narrower than 64 (the reference's constructor default for
    // edge_readout is 32, allegro_models.py:137) is zero-padded to 64 -- silu(0) = 0 and there are no biases, so the
    // extra units stay exactly zero -- which is what lets the stack run the fused linear-layer chains.
    const bool hid = !opt.no_channel_padding && silu && (q.num_layers == 2 || q.num_layers == 3) && (q.num_scalar == 64 || q.num_scalar == 128) &&
                     !opt.tp_generic && !opt.tp_no_chain && !opt.tp_no_moments;
    if (hid && q.embed_mlp_depth == 1 && q.embed_mlp_width >= 8 && q.embed_mlp_width < 64) cfg_local.embed_mlp_width = 64;
    if (hid && q.latent_mlp_depth == 1 && q.latent_mlp_width >= 8 && q.latent_mlp_width < 64) cfg_local.latent_mlp_width = 64;
    if (hid && q.readout_mlp_depth == 1 && q.readout_mlp_width >= 8 && q.readout_mlp_width < 64) cfg_local.readout_mlp_width = 64;
  }
  const aa_model_config* cfg = &cfg_local;
  AA_REQUIRE(cfg->dtype == AA_F32 || cfg->dtype == AA_F64, "model: bad dtype");
  AA_REQUIRE(cfg->l_max >= 1 && cfg->l_max <= 3, "model: l_max must be 1..3");
  for (int i = 0; i < 3; ++i) AA_REQUIRE(cfg->act_kind[i] >= AA_ACT_SILU && cfg->act_kind[i] <= AA_ACT_NONE, "model: unknown nonlinearity");
  AA_REQUIRE(cfg->num_layers >= 1 && cfg->num_layers <= AA_MAX_LAYERS, "model: num_layers out of range");
  AA_REQUIRE(cfg->embed_mlp_depth + 1 <= AA_MAX_MLP_LAYERS && cfg->latent_mlp_depth + 1 <= AA_MAX_MLP_LAYERS &&
                 cfg->readout_mlp_depth + 1 <= AA_MAX_MLP_LAYERS,
             "model: MLP too deep");
  AA_REQUIRE(cfg->num_types > 0 && cfg->num_bessels > 0 && cfg->num_bessels <= 16 && cfg->embed_dim % 2 == 0,
             "model: bad embedding sizes");
  aa_model_plan* p = new aa_model_plan();
  p->cfg = *cfg;
  p->u_raw = u_raw;
  p->hid_raw[0] = cfg_in->embed_mlp_width;
  p->hid_raw[1] = cfg_in->latent_mlp_width;
  p->hid_raw[2] = cfg_in->readout_mlp_width;
  p->opt = opt;
  const int L = cfg->num_layers, S = cfg->num_scalar, u = cfg->num_tensor;
  p->R = cfg->l_max + 1;
  p->D = p->R * p->R;
  p->W = p->R * u;
  p->SL1 = S * (L + 1);
  auto bail = [&](int rc) {
    for (void* q : p->owned) (void)hipFree(q);
    delete p;
    return rc;
  };
  p->layers.resize(L);
  for (int l = 0; l < L; ++l) {
    const aa_tp_desc& d = cfg->tps[l];
    bool ok = d.mul == u && d.d2 == p->D && (l == 0 ? d.d1 == p->D : d.d1 == cfg->tps[l - 1].dout) &&
              (l == L - 1 ? d.dout >= 1 : true);
    if (!ok) return bail(fail(AA_ERR_INVALID, "model: tensor-product layer dims are inconsistent"));
    int rc = build_tp_layer(d, &p->layers[l], &p->owned);
    if (rc) return bail(rc);
    p->spec_sig[l] = find_spec_sig(d);
    // the plan must not keep pointers into the caller's descriptor arrays
    p->cfg.tps[l].nz_i = p->cfg.tps[l].nz_j = p->cfg.tps[l].nz_k = p->cfg.tps[l].nz_path = nullptr;
    p->cfg.tps[l].nz_val = nullptr;
  }
  {
    p->use_spec = !opt.tp_generic;
    for (int l = 0; l < L; ++l) p->use_spec = p->use_spec && p->spec_sig[l] >= 0;
    // fp64 at l_max=3 does not fit the register file (256 VGPR + 256 AGPR + 1.9 KB scratch per lane, and
    // wrong results on hardware in round 1): keep it on the general LDS-table kernels for now
    if (cfg->dtype == AA_F64 && cfg->l_max >= 3) p->use_spec = false;
    p->chain_pair = -1;
    if (p->use_spec && L == 2 && !opt.tp_no_chain) p->chain_pair = find_chain_pair(p->spec_sig[0], p->spec_sig[1]);
    const int Dsh = (cfg->l_max + 1) * (cfg->l_max + 1);
    // every fused fast path below has SiLU built in; the other nonlinearities run the general kernels
    const bool all_silu = cfg->act_kind[0] == AA_ACT_SILU && cfg->act_kind[1] == AA_ACT_SILU && cfg->act_kind[2] == AA_ACT_SILU;
    p->env_mom = all_silu && p->chain_pair >= 0 && u == 64 && (S == 64 || S == 128) && cfg->latent_mlp_depth >= 1 &&
                 (cfg->latent_mlp_width == 64 || cfg->latent_mlp_width == 128) &&
                 (cfg->dtype == AA_F32 ? 4 : 8) * 4 * Dsh * (std::max(S, cfg->latent_mlp_width) + 64 + 64) <= 160 * 1024 &&
                 !opt.tp_no_moments;
    // per-atom operator kernels: every standard stack the tuned 2-layer/u=64 kernels above do not cover (and, with
    // AA_TP_OP=1, those too); they need the channel-minor layouts of the specialised path but none of its kernels,
    // so fp64 at l_max = 3 is fine here
    p->tp_op = -1;
    bool sigs_ok = !opt.tp_generic;
    for (int l = 0; l < L; ++l) sigs_ok = sigs_ok && p->spec_sig[l] >= 0;
    // Where the 2-layer moments kernels apply but the fused chains do NOT (fp64; S or the MLP widths 128; deeper MLPs), the operator
    // kernels -- with the slot form of the linear layers where that applies -- are the faster pipeline: measured on the C3 box
    // (profiles/r05_v3_shape_map_c3.md against r05_v7 / r05_v8 with the operator path forced): fp32 u 64 / S 128 7.41 -> 3.96 ms,
    // fp64 u = S = 64 4.89 -> 4.14 ms, fp64 S 128 8.80 -> 6.82 ms; fp32 S 64 with 128-wide latents is a wash (3.70 vs 3.76 ms) and
    // keeps the moments kernels.
    const bool would_chain = cfg->dtype == AA_F32 && S == 64 && cfg->embed_mlp_depth == 1 && cfg->embed_mlp_width == 64 && cfg->latent_mlp_depth == 1 &&
                             cfg->latent_mlp_width == 64 && cfg->readout_mlp_depth == 1 && cfg->readout_mlp_width == 64 && cfg->embed_dim % 32 == 0 &&
                             !opt.gemm_no_chain && !opt.gemm_fp32_mfma && !opt.gemm_valu;
    const bool would_slot = kSlotForm && !opt.no_slot_form && cfg->latent_mlp_depth == 1 && cfg->latent_mlp_width == S && cfg->readout_mlp_depth >= 1 &&
                            cfg->embed_mlp_depth >= 1 && cfg->embed_mlp_width == S && (cfg->readout_mlp_width % 16) == 0;
    const bool prefer_op = p->env_mom && !would_chain && (would_slot || cfg->dtype == AA_F64) && !opt.tp_prefer_moments;
    if (all_silu && sigs_ok && L >= 2 && L <= 3 && (u % 64) == 0 && u <= 256 && (S == 64 || S == 128) && cfg->latent_mlp_depth >= 1 &&
        (cfg->latent_mlp_width == 64 || cfg->latent_mlp_width == 128) && !opt.tp_no_moments && !opt.tp_no_operator &&
        (!p->env_mom || opt.tp_force_operator || prefer_op)) {
      const int chain = find_op_chain(p->spec_sig, L);
      if (chain >= 0) {
        p->tp_op = chain;
        p->env_mom = true;
        p->use_spec = true;
      }
    }
  }
  // weight blob layout
  size_t o = 0;
  auto take = [&](size_t n) {
    size_t r = o;
    o += (n + 63) / 64 * 64;
    return r;
  };
  const int T = cfg->num_types, B = cfg->num_bessels, S0 = cfg->embed_dim;
  p->o_rmax = take(size_t(T) * T);
  p->o_bessel = take(B);
  p->o_cemb = take(size_t(T) * S0 / 2);
  p->o_nemb = take(size_t(T) * S0 / 2);
  p->o_basis = take(size_t(B) * S0);
  AA_REQUIRE(cfg->embed_kind == 0 || (cfg->embed_kind == 1 && cfg->spline_span >= 0 && cfg->spline_span <= B),
             "model: embed_kind must be 0 (Bessel) or 1 (spline, 0 <= span <= num_splines)");
  auto lay = [&](MlpLayout& m, const std::vector<int>& dims, int nlayers) {
    m.dims = dims;
    for (int i = 0; i < nlayers; ++i) {
      m.w.push_back(take(size_t(dims[i]) * dims[i + 1]));
      m.wt.push_back(take(size_t(dims[i]) * dims[i + 1]));
      m.wp.push_back(take(gemm_packed_elems(dims[i], dims[i + 1])));
      m.wtp.push_back(take(gemm_packed_elems(dims[i + 1], dims[i])));
      m.wq.push_back(take(gemm_bf16x3_words(dims[i], dims[i + 1])));
      m.wtq.push_back(take(gemm_bf16x3_words(dims[i + 1], dims[i])));
    }
  };
  lay(p->embed, mlp_dims(S0, cfg->embed_mlp_depth, cfg->embed_mlp_width, S), cfg->embed_mlp_depth + 1);
  {
    p->chain_gemm = p->env_mom && (p->tp_op < 0 || u == 64) && cfg->dtype == AA_F32 && S == 64 && cfg->embed_mlp_depth == 1 &&
                    cfg->embed_mlp_width == 64 && cfg->latent_mlp_depth == 1 && cfg->latent_mlp_width == 64 &&
                    cfg->readout_mlp_depth == 1 && cfg->readout_mlp_width == 64 && cfg->embed_dim % 32 == 0 &&
                    !opt.gemm_no_chain && !opt.gemm_fp32_mfma && !opt.gemm_valu;
  }
  p->ng0 = p->env_mom ? S + p->W : S + 2 * p->W;
  p->o_g0 = take(size_t(S) * p->ng0);
  p->o_g0t = take(size_t(S) * p->ng0);
  p->o_g0p = take(gemm_packed_elems(S, p->ng0));
  p->o_g0tp = take(gemm_packed_elems(p->ng0, S));
  p->o_g0q = take(gemm_bf16x3_words(S, p->ng0));
  p->o_g0tq = take(gemm_bf16x3_words(p->ng0, S));
  if (p->env_mom) {
    for (int l = 0; l < L; ++l) {
      const size_t ka = l == 0 ? S : cfg->latent_mlp_width;
      p->o_wk[l] = take(ka * p->W);
      p->o_wt[l] = take(ka * p->W);
#ifdef AA_EXPERIMENTAL_TAIL
      p->o_wtk[l] = take(ka * p->W);
#endif
    }
  }
  for (int l = 0; l < L; ++l) {
    int in = S * (l + 1) + u, outd = S + ((l < L - 1 && !p->env_mom) ? p->W : 0);
    lay(p->latent[l], mlp_dims(in, cfg->latent_mlp_depth, cfg->latent_mlp_width, outd), cfg->latent_mlp_depth + 1);
    p->o_tpw[l] = take(size_t(cfg->tps[l].coupling ? u : 1) * cfg->tps[l].num_paths);
  }
  {
    std::vector<int> rd = mlp_dims(p->SL1, cfg->readout_mlp_depth, cfg->readout_mlp_width, 1);
    lay(p->readout, rd, cfg->readout_mlp_depth);  // all but the last layer
    p->ro_last_dim = rd[rd.size() - 2];
    p->o_ro_last = take(p->ro_last_dim);
  }
  p->o_b3a_q = p->o_b3b_q = p->o_b3c_q = 0;
  p->o_lat1in_fq = p->o_ro0_fq = p->o_b3af_q = p->o_b3bf_q = 0;
  p->o_g0fq = p->o_g0tfq = p->o_wk0f = p->o_wt0f = 0;
  for (int l = 0; l < AA_MAX_LAYERS; ++l) p->o_wkq[l] = 0;
  if (p->chain_gemm && p->env_mom && L == 2 && u == 64)  // (kProjMfma, and the two-waves-per-SIMD fused forward: env projection on the matrix cores)
    for (int l = 0; l < L; ++l) p->o_wkq[l] = take(size_t(p->R) * gemm_bf16x3_words(64, 64));
  if (kFoldEmb1 && p->chain_gemm && p->env_mom && L == 2 && u == 64) {
    p->o_g0fq = take(gemm_bf16x3_words(64, p->ng0));
    p->o_g0tfq = take(gemm_bf16x3_words(p->ng0, 64));
    p->o_wk0f = take(size_t(64) * p->W);
    p->o_wt0f = take(size_t(64) * p->W);
  }
  if (kFoldLatent && p->chain_gemm && L == 2 && u == 64) {  // (chain_gemm: S = every MLP width = 64, one hidden layer each)
    p->o_lat1in_fq = take(gemm_bf16x3_words(2 * S + u, 64));
    p->o_ro0_fq = take(gemm_bf16x3_words(3 * S, 64));
    p->o_b3af_q = take(gemm_bf16x3_words(64, 64));
    if (kFoldLat0Rev) p->o_b3bf_q = take(gemm_bf16x3_words(128, S * L));
  }
  {
    // two-body table [T*T][B][S0] (type embedding x basis weights): the last reverse chain contracts against it (<= 2
    // species: its LDS copy must leave room for three workgroups per CU) and the fused forward evaluates the embedding
    // from it (<= 3 species: 18 KB of its LDS).  Part of the blob whenever the shapes allow -- not a function of the options.
    const bool tab_ok = p->chain_gemm && T <= 3 && B == 8 && S0 == 64;
    p->embed_fused = tab_ok && T <= 2 && !opt.embed_no_fuse;
    p->o_embtab = (tab_ok || cfg->embed_kind == 1) ? take(size_t(T) * T * B * S0) : 0;
    // (the fold needs a hidden layer of 64 behind the embedding: what the chains / the fused forward require anyway)
    p->o_embtab_h = (kFoldEmbed && tab_ok && cfg->embed_mlp_depth >= 1 && p->embed.dims.size() >= 2 && p->embed.dims[1] == 64) ? take(size_t(T) * T * B * 64) : 0;
  }
  if (p->chain_gemm) {
    // merged reverse chain "readout' o latent_{L-1}'" (see Runner::backward): the readout-reverse columns that feed
    // the last latent, and [readout-reverse columns of the earlier features (zero-padded) ; latent-reverse] stacked
    p->o_b3a_q = take(gemm_bf16x3_words(64, S));
    p->o_b3b_q = take(gemm_bf16x3_words(128, S * L));
    p->o_b3c_q = take(gemm_bf16x3_words(64, u));
  }
  {
    const int De = cfg->embed_mlp_depth, Hr = cfg->readout_mlp_width, H = cfg->latent_mlp_width;
    p->slot_form = kSlotForm && !opt.no_slot_form && !p->chain_gemm && p->tp_op >= 0 && p->env_mom && cfg->latent_mlp_depth == 1 && H == S &&
                   cfg->readout_mlp_depth >= 1 && De >= 1 && cfg->embed_mlp_width == S && (Hr % 16) == 0;
    p->o_s_wk0 = p->o_s_wt0 = 0;
    if (p->slot_form) {
      auto mat = [&](int K, int N) {
        GemmMat m;
        m.K = K;
        m.N = N;
        m.w = take(size_t(K) * N);
        m.wp = take(gemm_packed_elems(K, N));
        m.wq = take(gemm_bf16x3_words(K, N));
        return m;
      };
      p->s_g0f = mat(S, p->ng0);
      p->s_g0ft = mat(p->ng0, S);
      for (int l = 0; l < L; ++l) {
        p->s_in[l] = mat(S * (l + 1) + u, H);
        p->s_rs[l] = mat(Hr + S * (L - 1 - l), H);
        p->s_sct[l] = mat(H, u);
      }
      p->s_ro0 = mat(p->SL1, Hr);
      p->s_rstb = mat(Hr + S * L, S);
      p->o_s_wk0 = take(size_t(S) * p->W);
      p->o_s_wt0 = take(size_t(S) * p->W);
    }
  }
  p->op_proj = p->tp_op >= 0 && opt.op_proj_gemm != 2;
  if (p->op_proj) {
    auto mset = [&](int K, int N, int count) {
      GemmMatSet m;
      m.K = K;
      m.N = N;
      m.count = count;
      auto r64 = [](size_t n) { return (n + 63) / 64 * 64; };
      m.w_bs = r64(size_t(K) * N);
      m.wp_bs = r64(gemm_packed_elems(K, N));
      m.wq_bs = r64(gemm_bf16x3_words(K, N));
      m.w = take(m.w_bs * count);
      m.wp = take(m.wp_bs * count);
      m.wq = take(m.wq_bs * count);
      return m;
    };
    for (int l = 0; l < L; ++l) {
      const int ka = l == 0 ? S : cfg->latent_mlp_width;
      p->s_pr[l] = mset(ka, u, p->R);
      p->s_prt[l] = mset(u, ka, p->R);
    }
    if (p->slot_form) {
      p->s_pr0f = mset(S, u, p->R);
      p->s_prt0f = mset(u, S, p->R);
    }
  }
  p->o_scales = take(T);
  p->o_shifts = take(T);
  p->n_elems = o;
  {
    // fused per-atom-tile forward (aa_fused.hip): the standard 2-layer 64-wide fp32 stack with the
    // two-body table in LDS; same parity tests as the staged pipeline.  With the tensor-track scalars accumulated in
    // anchored program order (aa::anchor -- the kernel used to carry 70-350 spilled VGPRs) the 32-edge-tile form beats the
    // staged forward at every size on MI355X: 22-24 % of the step on 64-1000 atoms (one launch instead of seven), 9 % at
    // 4096, 4.5 % at 10 648, 0.5 % at 97 336 atoms, and it moves 2.1 instead of 7.3 KB/edge (profiles/archive/r02_v23_fused_sweep.log).
    // aa_plan_options.fused_forward: 0 / 1 = whenever the graph allows (automatic), 3 = never (staged pipeline); A/B: 2 = also for every
    // graph with segments <= 128, in the pure team form, 4 = the same in the mixed form.
    const bool eligible = p->chain_gemm && p->env_mom && p->tp_op < 0 && (p->chain_pair == 0 || p->chain_pair == 1) && L == 2 &&
                          u == 64 && S == 64 && T <= 3 && B == 8 && S0 == 64 && p->o_embtab != 0 && (!kFoldEmbed || p->o_embtab_h != 0) &&
                          (!kFoldLatent || p->o_lat1in_fq != 0) && (!kFoldEmb1 || p->o_g0fq != 0) && (!kProjMfma || p->o_wkq[0] != 0);
    p->fused_fwd = eligible && opt.fused_forward != 3;
#ifdef AA_EXPERIMENTAL_TAIL
    p->fused_tail = p->fused_fwd && p->embed_fused && (opt.fused_tail == 1 || opt.fused_tail == 2);
#else
    p->fused_tail = false;
#endif
    p->fused_hold_w0 = !opt.fused_recompute_w0;  // A/B: recompute w0 for the second layer instead of holding it
    // the eight-wave form needs the folded program (every fold active), a two-body table that leaves 16.6 KB of LDS per wave
    // (one species) and the w0 rows of the staged reverse pass (no fused tail)
    p->fused_wide = p->fused_fwd && kFoldEmbed && kFoldEmb1 && kFoldLatent && !kProjMfma && p->o_embtab_h != 0 && p->o_g0fq != 0 &&
                    p->o_lat1in_fq != 0 && opt.fused_narrow != 1 && !p->fused_tail && fused_fwd8_lds_bytes(cfg->num_types, 8) <= 160 * 1024;
  }
  {
    void* st = nullptr;
    if (hipHostMalloc(&st, 64, hipHostMallocDefault) != hipSuccess || !st) {
      aa_model_plan_destroy(p);
      return fail(AA_ERR_HIP, "aa_model_plan_create: cannot allocate the status word");
    }
    std::memset(st, 0, 64);
    p->status = static_cast<int32_t*>(st);
  }
  *out = p;
  return AA_OK;
}

extern "C" int aa_model_plan_describe(const aa_model_plan* p, char* buf, size_t n) {
  AA_REQUIRE(p && buf && n > 0, "aa_model_plan_describe: null argument");
  const bool fused = p->fused_fwd;
  // MFMA steps (one step = a 64-feature tile pair x a 32-deep chunk = 24 bf16 MFMAs per 32-edge tile) of the fused forward: executed
  // (after the folds) vs the step-equivalents of the reference's linear layers (what `roofline.achieved` prices)
  const int R = p->R;
  const int ref_steps = 2 + 2 + (2 + 2 * R) + 4 + 2 + 6 + 2 + 6;
  const int exec_steps = fused ? fused_fwd_num_steps(R, p->fused_hold_w0) - 8 /* env-projection steps: vector work */ : 0;
  const int k = snprintf(buf, n,
                         "{\"fused_forward\": %s, \"fold_embed_table\": %s, \"fold_embed_output\": %s, \"fold_latent_outputs\": %s, "
                         "\"fold_lat0_reverse\": %s, \"fused_mfma_steps_executed\": %d, \"fused_mfma_steps_reference\": %d, "
                         "\"chain_gemm\": %s, \"moments\": %s, \"operator_path\": %s, \"slot_form\": %s, \"fused_wide\": %s}",
                         fused ? "true" : "false", (fused && kFoldEmbed && p->o_embtab_h) ? "true" : "false",
                         (fused && kFoldEmb1 && p->o_g0fq) ? "true" : "false", (fused && kFoldLatent && p->o_lat1in_fq) ? "true" : "false",
                         (kFoldLat0Rev && p->o_b3bf_q) ? "true" : "false", exec_steps, fused ? ref_steps : 0, p->chain_gemm ? "true" : "false",
                         p->env_mom ? "true" : "false", p->tp_op >= 0 ? "true" : "false", p->slot_form ? "true" : "false",
                         p->fused_wide ? "true" : "false");
  return (k < 0 || size_t(k) >= n) ? fail(AA_ERR_INVALID, "aa_model_plan_describe: buffer too small") : k;
}

extern "C" int aa_model_check(const aa_model_plan* plan, aa_stream stream) {
  AA_REQUIRE(plan, "aa_model_check: null plan");
  AA_CHECK_HIP(hipStreamSynchronize(static_cast<hipStream_t>(stream)));
  return consume_status(plan, "aa_model_check");
}

extern "C" int aa_model_plan_set_forward_events(aa_model_plan* plan, void* wait_event, void* record_event) {
  AA_REQUIRE(plan, "aa_model_plan_set_forward_events: null plan");
  plan->ev_wait = wait_event;
  plan->ev_record = record_event;
  return AA_OK;
}

extern "C" int aa_model_plan_enable_graph(aa_model_plan* plan, int on) {
  AA_REQUIRE(plan, "aa_model_plan_enable_graph: null plan");
  aa_model_plan::StepGraph& g = plan->sg;
  if (g.exec) (void)hipGraphExecDestroy(g.exec);
  if (g.graph) (void)hipGraphDestroy(g.graph);
  g.exec = nullptr;
  g.graph = nullptr;
  g.enabled = on != 0;
  if (g.enabled && !g.cap_stream) AA_CHECK_HIP(hipStreamCreateWithFlags(&g.cap_stream, hipStreamNonBlocking));
  return AA_OK;
}

extern "C" int aa_model_plan_enable_taps(aa_model_plan* plan, int on) {
  AA_REQUIRE(plan, "aa_model_plan_enable_taps: null plan");
  plan->taps = on != 0;
  // a captured step belongs to the pipeline that was selected when it was captured
  aa_model_plan::StepGraph& g = plan->sg;
  if (g.exec) (void)hipGraphExecDestroy(g.exec);
  if (g.graph) (void)hipGraphDestroy(g.graph);
  g.exec = nullptr;
  g.graph = nullptr;
  return AA_OK;
}

extern "C" void aa_model_plan_destroy(aa_model_plan* plan) {
  if (!plan) return;
  if (plan->sg.exec) (void)hipGraphExecDestroy(plan->sg.exec);
  if (plan->sg.graph) (void)hipGraphDestroy(plan->sg.graph);
  if (plan->sg.cap_stream) (void)hipStreamDestroy(plan->sg.cap_stream);
  for (void* q : plan->owned) (void)hipFree(q);
  if (plan->status) (void)hipHostFree(plan->status);
  delete plan;
}

extern "C" size_t aa_model_weights_bytes(const aa_model_plan* plan) { return plan ? plan->n_elems * plan->esize() : 0; }

extern "C" uint64_t aa_model_plan_layout_hash(const aa_model_plan* p) {
  if (!p) return 0;
  uint64_t h = 1469598103934665603ull;  // FNV-1a over every quantity aa_model_pack_weights lays the blob out by
  auto mix = [&](uint64_t v) {
    for (int i = 0; i < 8; ++i) {
      h ^= (v >> (8 * i)) & 0xff;
      h *= 1099511628211ull;
    }
  };
  auto mixv = [&](const std::vector<size_t>& v) {
    mix(v.size());
    for (size_t x : v) mix(x);
  };
  auto mixm = [&](const MlpLayout& m) {
    mix(m.dims.size());
    for (int d : m.dims) mix(uint64_t(d));
    mixv(m.w); mixv(m.wt); mixv(m.wp); mixv(m.wtp); mixv(m.wq); mixv(m.wtq);
  };
  const aa_model_config& c = p->cfg;
  for (uint64_t v : {uint64_t(c.dtype), uint64_t(c.num_types), uint64_t(c.num_bessels), uint64_t(c.l_max), uint64_t(c.num_layers),
                     uint64_t(c.num_scalar), uint64_t(c.num_tensor), uint64_t(c.embed_dim), uint64_t(c.embed_mlp_width),
                     uint64_t(c.latent_mlp_width), uint64_t(c.readout_mlp_width), uint64_t(p->u_raw), uint64_t(p->use_spec),
                     uint64_t(p->env_mom), uint64_t(p->chain_gemm), uint64_t(p->chain_pair + 1), uint64_t(p->tp_op + 1), uint64_t(p->ng0),
                     uint64_t(p->n_elems), uint64_t(p->slot_form)})
    mix(v);
  if (p->slot_form) {
    auto mixm = [&](const GemmMat& m) { mix(m.w); mix(m.wp); mix(m.wq); };
    mixm(p->s_g0f); mixm(p->s_g0ft); mixm(p->s_ro0); mixm(p->s_rstb);
    for (int l = 0; l < c.num_layers; ++l) { mixm(p->s_in[l]); mixm(p->s_rs[l]); mixm(p->s_sct[l]); }
    mix(p->o_s_wk0); mix(p->o_s_wt0);
  }
  if (p->op_proj) {
    auto mixs = [&](const GemmMatSet& m) { mix(m.w); mix(m.wp); mix(m.wq); mix(m.w_bs); };
    for (int l = 0; l < c.num_layers; ++l) { mixs(p->s_pr[l]); mixs(p->s_prt[l]); }
    mixs(p->s_pr0f); mixs(p->s_prt0f);
  }
  for (size_t v : {p->o_rmax, p->o_bessel, p->o_cemb, p->o_nemb, p->o_basis, p->o_g0, p->o_g0t, p->o_g0p, p->o_g0tp, p->o_g0q, p->o_g0tq,
                   p->o_b3a_q, p->o_b3b_q, p->o_b3c_q, p->o_ro_last, p->o_scales, p->o_shifts, p->o_embtab, p->o_embtab_h, p->o_lat1in_fq, p->o_ro0_fq, p->o_b3af_q, p->o_b3bf_q, p->o_g0fq, p->o_g0tfq, p->o_wk0f, p->o_wt0f, p->o_wkq[0], p->o_wkq[1]})
    mix(v);
  for (int l = 0; l < c.num_layers; ++l) {
    mix(p->o_tpw[l]);
    mix(p->env_mom ? p->o_wk[l] : 0);
    mix(p->env_mom ? p->o_wt[l] : 0);
#ifdef AA_EXPERIMENTAL_TAIL
    mix(p->env_mom ? p->o_wtk[l] : 0);
#endif
    mixm(p->latent[l]);
  }
  mixm(p->embed);
  mixm(p->readout);
  return h ? h : 1;
}

// alpha_i of nequip ScalarMLPFunction: c_prev / sqrt(fan_in | fan_out)
// `which`: 0 scalar_embed_mlp, 1 latent MLPs, 2 edge_readout (their nonlinearities may differ); linear maps pass -1
static double act_const_of(const aa_model_config& c, int which) {
  if (which < 0) return 1.0;
  if (c.act_kind[which] == AA_ACT_NONE) return 1.0;
  return c.act_consts[which] > 0 ? c.act_consts[which] : c.act_const;
}
static double mlp_alpha(const aa_model_config& c, int layer, int din, int dout, int which = -1) {
  double norm = layer == 0 ? 1.0 : act_const_of(c, which);
  return norm / std::sqrt(double(c.forward_weight_init ? din : dout));
}

extern "C" int aa_model_pack_weights(const aa_model_plan* p, const aa_model_raw_weights* raw_in, void* dev_blob,
                                     size_t blob_bytes, aa_stream stream) {
  AA_REQUIRE(p && raw_in && dev_blob, "aa_model_pack_weights: null argument");
  AA_REQUIRE(blob_bytes >= aa_model_weights_bytes(p), "aa_model_pack_weights: blob too small");
  const aa_model_config& c = p->cfg;
  const int S = c.num_scalar, u = c.num_tensor, L = c.num_layers, T = c.num_types, B = c.num_bessels, S0 = c.embed_dim;
  const int W = p->W;
  // Channel-padded plan (aa_model_plan_create): the state_dict tensors carry u_raw channels; build zero-padded copies in
  // the reference's own layouts so that everything below indexes a u-channel model.  The ScalarMLPFunction constants
  // (alpha = c / sqrt(fan_in | fan_out)) keep following the TRUE layer shapes (the *_raw widths).
  const int u_raw = p->u_raw, Rp = p->R, dlat = c.latent_mlp_depth;
  const int We_in = (c.env_shared_weights != 0) ? u_raw : Rp * u_raw;   // env columns in the state_dict
  const int We_pad = (c.env_shared_weights != 0) ? u : W;
  const int dW = W - Rp * u_raw, dWe = We_pad - We_in, du = u - u_raw;  // how much wider the padded shapes are
  const int dHe = c.embed_mlp_width - p->hid_raw[0], dHl = c.latent_mlp_width - p->hid_raw[1], dHr = c.readout_mlp_width - p->hid_raw[2];
  aa_model_raw_weights raw_local = *raw_in;
  std::vector<std::vector<double>> pad_store;
  // zero-extend a row-major [r_raw][c_raw] matrix to [r_pad][c_pad] (every padded block is a suffix: both env layouts are
  // channel-major -- [u][R] or [u] -- and hidden units / scalar rows are appended at the end)
  auto pad2d = [&](const double* src, int r_raw, int c_raw, int r_pad, int c_pad) -> const double* {
    if (r_raw == r_pad && c_raw == c_pad) return src;
    pad_store.emplace_back(size_t(r_pad) * c_pad, 0.0);
    std::vector<double>& d = pad_store.back();
    for (int r = 0; r < r_raw; ++r)
      for (int q = 0; q < c_raw; ++q) d[size_t(r) * c_pad + q] = src[size_t(r) * c_raw + q];
    return d.data();
  };
  if (du || dHe || dHl || dHr) {
    AA_REQUIRE(raw_in->env_embed_linear && raw_in->first_proj, "pack: missing embedding weights");
    raw_local.env_embed_linear = pad2d(raw_in->env_embed_linear, S, Rp * u_raw, S, W);
    raw_local.first_proj = pad2d(raw_in->first_proj, S, S + We_in, S, S + We_pad);
    if (dHe) {  // (only single-hidden-layer MLPs are padded)
      AA_REQUIRE(raw_in->embed_mlp[0] && raw_in->embed_mlp[1], "pack: missing scalar_embed_mlp weights");
      raw_local.embed_mlp[0] = pad2d(raw_in->embed_mlp[0], S0, p->hid_raw[0], S0, c.embed_mlp_width);
      raw_local.embed_mlp[1] = pad2d(raw_in->embed_mlp[1], p->hid_raw[0], S, c.embed_mlp_width, S);
    }
    for (int l = 0; l < L; ++l) {
      AA_REQUIRE(raw_in->latent[l][0] && raw_in->latent[l][dlat] && raw_in->tp_weights[l], "pack: missing latent / tp weights");
      const int out_raw = S + (l < L - 1 ? We_in : 0), out_pad = S + (l < L - 1 ? We_pad : 0);
      if (dlat == 0) {
        raw_local.latent[l][0] = pad2d(raw_in->latent[l][0], S * (l + 1) + u_raw, out_raw, S * (l + 1) + u, out_pad);
      } else {
        // first layer: input rows [S (l + 1) | u scalars], output = hidden units; last layer: [hidden][S | env columns]
        raw_local.latent[l][0] = pad2d(raw_in->latent[l][0], S * (l + 1) + u_raw, dlat == 1 ? p->hid_raw[1] : c.latent_mlp_width,
                                       S * (l + 1) + u, c.latent_mlp_width);
        raw_local.latent[l][dlat] = pad2d(raw_in->latent[l][dlat], dlat == 1 ? p->hid_raw[1] : c.latent_mlp_width, out_raw, c.latent_mlp_width, out_pad);
      }
      if (c.tps[l].coupling) raw_local.tp_weights[l] = pad2d(raw_in->tp_weights[l], u_raw, c.tps[l].num_paths, u, c.tps[l].num_paths);
    }
    if (dHr) {
      AA_REQUIRE(raw_in->readout[0] && raw_in->readout[1], "pack: missing readout weights");
      raw_local.readout[0] = pad2d(raw_in->readout[0], p->SL1, p->hid_raw[2], p->SL1, c.readout_mlp_width);
      raw_local.readout[1] = pad2d(raw_in->readout[1], p->hid_raw[2], 1, c.readout_mlp_width, 1);
    }
  }
  const aa_model_raw_weights* raw = &raw_local;
  std::vector<double> h(p->n_elems, 0.0);
  auto copy = [&](size_t off, const double* src, size_t n, double scale) {
    for (size_t i = 0; i < n; ++i) h[off + i] = src[i] * scale;
  };
  AA_REQUIRE(raw->rmax_recip && raw->env_embed_linear && raw->first_proj, "pack: missing embedding weights");
  copy(p->o_rmax, raw->rmax_recip, size_t(T) * T, 1.0);
  if (c.embed_kind == 1) {
    // [class][c][s] (spline.py:69-71) -> basis-major [class][s][c]
    AA_REQUIRE(raw->spline_weights, "pack: missing spline weights");
    for (int cls = 0; cls < T * T; ++cls)
      for (int cc = 0; cc < S0; ++cc)
        for (int n = 0; n < B; ++n)
          h[p->o_embtab + (size_t(cls) * B + n) * S0 + cc] = raw->spline_weights[(size_t(cls) * S0 + cc) * B + n];
  } else {
    AA_REQUIRE(raw->bessel_weights && raw->center_embed && raw->neighbor_embed && raw->basis_linear,
               "pack: missing embedding weights");
    // Two published conventions of nequip's BesselEdgeLengthEncoding (EXT; which one a given nequip release uses
    // cannot be checked in this container -- DESIGN.md section 6): (a) bessel_weights = n pi, basis sin(w x) / x;
    // (b) bessel_weights = n (linspace(1, B, B)), basis sinc(x w) w = sin(pi w x) / (pi x).  Both are served by the
    // same kernels, sin(w' x) / x: (b) is recognised by its integer roots and packed as w' = pi w with the 1/pi
    // prefactor folded into the basis_linear rows.
    const double kPi = 3.14159265358979323846;
    bool sinc_form = c.bessel_convention == 2;
    if (c.bessel_convention == 0) {
      bool is_n = true, is_npi = true;
      for (int n = 0; n < B; ++n) {
        // (relative 1e-5: the roots of an fp32 state_dict are n*pi rounded to fp32)
        is_n = is_n && std::fabs(raw->bessel_weights[n] - double(n + 1)) < 1e-5 * double(n + 1);
        is_npi = is_npi && std::fabs(raw->bessel_weights[n] - double(n + 1) * kPi) < 1e-5 * double(n + 1) * kPi;
      }
      if (!is_n && !is_npi)
        return fail(AA_ERR_INVALID, "pack: bessel_weights are neither n nor n*pi (trained roots?): state aa_model_config.bessel_convention (1: sin(w x)/x, 2: sinc)");
      sinc_form = is_n;
    } else if (c.bessel_convention != 1 && c.bessel_convention != 2) {
      return fail(AA_ERR_INVALID, "pack: bessel_convention must be 0 (recognise), 1 (roots n*pi) or 2 (sinc form)");
    }
    copy(p->o_bessel, raw->bessel_weights, B, sinc_form ? kPi : 1.0);
    copy(p->o_cemb, raw->center_embed, size_t(T) * S0 / 2, 1.0);
    copy(p->o_nemb, raw->neighbor_embed, size_t(T) * S0 / 2, 1.0);
    copy(p->o_basis, raw->basis_linear, size_t(B) * S0, mlp_alpha(c, 0, B, S0) * (sinc_form ? 1.0 / kPi : 1.0));
  }
  // env-weight columns: reference layout [u][R] (_channels.py:46-51); the specialised kernels want [R][u].
  // With weight_individual_irreps=False the Allegro layers' env weights are [u] in the reference (_channels.py:29-31,
  // 60-63: one weight per channel for all irreps): the packed matrices replicate that column for every irrep, which
  // is the same linear map, so every kernel runs unchanged.  (The tensor-embedding weights w0 are always individual,
  // tensorembed.py:76-81.)
  const int Rr = p->R;
  const bool shared = c.env_shared_weights != 0;
  const int We = shared ? u : W;  // env-weight columns of first_proj / latent outputs in the state_dict
  auto w0_col = [&](int q) { return p->use_spec ? (q % u) * Rr + q / u : q; };  // packed col q <- reference col
  auto env_col = [&](int q) {
    const int ch = p->use_spec ? q % u : q / Rr, r = p->use_spec ? q / u : q % Rr;
    return shared ? ch : ch * Rr + r;
  };
  // pack an MLP; if env_off >= 0 the LAST layer's columns [env_off, env_off+W) are env weights
  // raw_last_width: true column count of the LAST layer in the state_dict (>= packed width when the env columns
  // are split off for the moments path); alpha always follows the reference's full layer shape
  // true_din_less / true_dout_less: how much narrower the reference's first-layer input / last-layer output is than the
  // (channel-padded) shapes packed here -- alpha follows the reference
  // hidden_less: by how much the (single) hidden layer was widened -- layer 0's fan_out and layer 1's fan_in are narrower
  // in the reference.  `total_layers`: layers of the whole MLP (the readout packs only its first ones here).
  auto pack_mlp = [&](const MlpLayout& m, const double* const* ws, int nlayers, int env_off, int which, int raw_last_width = -1,
                      int true_din_less = 0, int true_dout_less = 0, int hidden_less = 0, int total_layers = -1) -> bool {
    if (total_layers < 0) total_layers = nlayers;
    for (int i = 0; i < nlayers; ++i) {
      if (!ws[i]) return false;
      int din = m.dims[i], dout = m.dims[i + 1];
      int raw_w = (i == nlayers - 1 && raw_last_width > 0) ? raw_last_width : dout;
      double al = mlp_alpha(c, i, din - (i == 0 ? true_din_less : hidden_less),
                            raw_w - (i == total_layers - 1 ? true_dout_less : hidden_less), which);
      for (int r = 0; r < din; ++r)
        for (int q = 0; q < dout; ++q) {
          int src = q;
          if (i == nlayers - 1 && env_off >= 0 && q >= env_off && q < env_off + W) src = env_off + env_col(q - env_off);
          double v = ws[i][size_t(r) * raw_w + src] * al;
          h[m.w[i] + size_t(r) * dout + q] = v;
          h[m.wt[i] + size_t(q) * din + r] = v;
        }
      gemm_pack_b(&h[m.w[i]], din, dout, &h[m.wp[i]]);
      gemm_pack_b(&h[m.wt[i]], dout, din, &h[m.wtp[i]]);
    }
    return true;
  };
  AA_REQUIRE(pack_mlp(p->embed, raw->embed_mlp, c.embed_mlp_depth + 1, -1, 0, -1, 0, 0, dHe), "pack: missing scalar_embed_mlp weights");
  {
    // fused first stage: [ two_body (first_proj[:, :S]) | w0 (env_embed_linear) | env_w0 (first_proj[:, S:]) ]
    // (moments path: the env_w0 columns are not part of the GEMM; they become Wenv of layer 0 below)
    const int NG = p->ng0;
    double a_env = mlp_alpha(c, 0, S, W - dW), a_proj = mlp_alpha(c, 0, S, S + We - dWe);
    for (int r = 0; r < S; ++r)
      for (int q = 0; q < NG; ++q) {
        double v;
        if (q < S)
          v = raw->first_proj[size_t(r) * (S + We) + q] * a_proj;
        else if (q < S + W)
          v = raw->env_embed_linear[size_t(r) * W + w0_col(q - S)] * a_env;
        else
          v = raw->first_proj[size_t(r) * (S + We) + S + env_col(q - S - W)] * a_proj;
        h[p->o_g0 + size_t(r) * NG + q] = v;
        h[p->o_g0t + size_t(q) * S + r] = v;
      }
    gemm_pack_b(&h[p->o_g0], S, NG, &h[p->o_g0p]);
    gemm_pack_b(&h[p->o_g0t], NG, S, &h[p->o_g0tp]);
  }
  for (int l = 0; l < L; ++l) {
    AA_REQUIRE(pack_mlp(p->latent[l], raw->latent[l], c.latent_mlp_depth + 1, (l < L - 1 && !p->env_mom) ? S : -1, 1,
                        S + (l < L - 1 ? We : 0), du, l < L - 1 ? dWe : 0, dHl),
               "pack: missing latent weights");
    AA_REQUIRE(raw->tp_weights[l], "pack: missing tp weights");
    copy(p->o_tpw[l], raw->tp_weights[l], size_t(c.tps[l].coupling ? u : 1) * c.tps[l].num_paths, 1.0);
  }
  if (p->env_mom) {
    // Wenv_l[k][r][ch] (and its [r][ch][k] transpose) from the reference's [k][S + ch*R + r] columns
    auto fill = [&](int l, const double* rawm, int ka, int raw_w, double al) {
      for (int k = 0; k < ka; ++k)
        for (int r = 0; r < Rr; ++r)
          for (int ch = 0; ch < u; ++ch) {
            double v = rawm[size_t(k) * raw_w + S + (shared ? ch : ch * Rr + r)] * al;
            h[p->o_wk[l] + (size_t(k) * Rr + r) * u + ch] = v;
            h[p->o_wt[l] + (size_t(r) * u + ch) * ka + k] = v;
#ifdef AA_EXPERIMENTAL_TAIL
            h[p->o_wtk[l] + (size_t(ch) * Rr + r) * ka + k] = v;
#endif
          }
    };
    fill(0, raw->first_proj, S, S + We, mlp_alpha(c, 0, S, S + We - dWe));
    const int dl = c.latent_mlp_depth;  // index of a latent's last layer
    for (int l = 1; l < L; ++l)
      fill(l, raw->latent[l - 1][dl], c.latent_mlp_width, S + We, mlp_alpha(c, dl, c.latent_mlp_width - dHl, S + We - dWe, 1));
  }
  AA_REQUIRE(pack_mlp(p->readout, raw->readout, c.readout_mlp_depth, -1, 2, -1, 0, 0, dHr, c.readout_mlp_depth + 1), "pack: missing readout weights");
  {
    const double* wl = raw->readout[c.readout_mlp_depth];
    AA_REQUIRE(wl, "pack: missing readout weights");
    copy(p->o_ro_last, wl, p->ro_last_dim, mlp_alpha(c, c.readout_mlp_depth, p->ro_last_dim - (c.readout_mlp_depth > 0 ? dHr : 0), 1, 2));
  }
  if (c.has_scales) {
    AA_REQUIRE(raw->scales, "pack: missing scales");
    copy(p->o_scales, raw->scales, T, 1.0);
  }
  if (c.has_shifts) {
    AA_REQUIRE(raw->shifts, "pack: missing shifts");
    copy(p->o_shifts, raw->shifts, T, 1.0);
  }
  if (p->o_embtab && c.embed_kind == 0) {
    const int half = S0 / 2;
    for (int ti = 0; ti < T; ++ti)
      for (int tj = 0; tj < T; ++tj)
        for (int n = 0; n < B; ++n)
          for (int cc = 0; cc < S0; ++cc) {
            const double te = cc < half ? h[p->o_cemb + size_t(ti) * half + cc] : h[p->o_nemb + size_t(tj) * half + (cc - half)];
            h[p->o_embtab + ((size_t(ti) * T + tj) * B + n) * S0 + cc] = te * h[p->o_basis + size_t(n) * S0 + cc];
          }
  }
  if (p->o_wk0f) {
    // (kFoldEmb1) env weights of layer 0 behind the output layer of scalar_embed_mlp: Wenv0'[k][r][ch] = sum_m W1[k][m] Wenv0[m][r][ch]
    const double* w1 = &h[p->embed.w[1]];  // [64, S]
    const int Rr_ = p->R, uu = c.num_tensor;
    for (int k = 0; k < 64; ++k)
      for (int r = 0; r < Rr_; ++r)
        for (int ch = 0; ch < uu; ++ch) {
          double v = 0.0;
          for (int m = 0; m < S; ++m) v += w1[size_t(k) * S + m] * h[p->o_wk[0] + (size_t(m) * Rr_ + r) * uu + ch];
          h[p->o_wk0f + (size_t(k) * Rr_ + r) * uu + ch] = v;
          h[p->o_wt0f + (size_t(r) * uu + ch) * 64 + k] = v;
        }
  }
  if (p->o_embtab_h) {
    // T[pair][n][k] = sum_c tab[pair][n][c] * W0[c][k]  (W0: the packed first layer of scalar_embed_mlp, normalisation folded)
    const int H = p->embed.dims[1];
    for (int cls = 0; cls < T * T; ++cls)
      for (int n = 0; n < B; ++n)
        for (int k = 0; k < H; ++k) {
          double acc = 0.0;
          for (int cc = 0; cc < S0; ++cc) acc += h[p->o_embtab + (size_t(cls) * B + n) * S0 + cc] * h[p->embed.w[0] + size_t(cc) * H + k];
          h[p->o_embtab_h + (size_t(cls) * B + n) * H + k] = acc;
        }
  }
  std::vector<const GemmMat*> slot_mats;  // (fp32 plans: bf16x3 copies are split off the rounded matrices below)
  if (p->slot_form) {
    // slot form: every matrix below is a product / regrouping of the NORMALISED matrices packed above, formed in fp64
    const int De = c.embed_mlp_depth, Hr = c.readout_mlp_width, H = c.latent_mlp_width, NG = p->ng0, SL1 = p->SL1;
    auto put = [&](const GemmMat& m, const std::vector<double>& d) {
      std::copy(d.begin(), d.end(), h.begin() + m.w);
      gemm_pack_b(&h[m.w], m.K, m.N, &h[m.wp]);
      slot_mats.push_back(&m);
    };
    const double* w1 = &h[p->embed.w[De]];  // [S(=He), S] output layer of scalar_embed_mlp
    {
      std::vector<double> gf(size_t(S) * NG), gft(size_t(NG) * S);
      for (int k = 0; k < S; ++k)
        for (int q = 0; q < NG; ++q) {
          double v = 0.0;
          for (int m = 0; m < S; ++m) v += w1[size_t(k) * S + m] * h[p->o_g0 + size_t(m) * NG + q];
          gf[size_t(k) * NG + q] = v;
          gft[size_t(q) * S + k] = v;
        }
      put(p->s_g0f, gf);
      put(p->s_g0ft, gft);
      for (int k = 0; k < S; ++k)
        for (int r = 0; r < Rr; ++r)
          for (int ch = 0; ch < u; ++ch) {
            double v = 0.0;
            for (int m = 0; m < S; ++m) v += w1[size_t(k) * S + m] * h[p->o_wk[0] + (size_t(m) * Rr + r) * u + ch];
            h[p->o_s_wk0 + (size_t(k) * Rr + r) * u + ch] = v;
            h[p->o_s_wt0 + (size_t(r) * u + ch) * S + k] = v;
          }
    }
    // F(j -> consumer)[m][n] = sum_q Wout_j[m][q] Wc[S (j+1) + q][n]: the consumer's row block of lat_j behind latent j's output layer
    auto fold_block = [&](int j, const double* wc, int Nc, std::vector<double>& out /* [H][Nc] */) {
      const double* wo = &h[p->latent[j].w[1]];  // [H, S]
      out.assign(size_t(H) * Nc, 0.0);
      for (int m = 0; m < H; ++m)
        for (int q = 0; q < S; ++q) {
          const double a = wo[size_t(m) * S + q];
          const double* row = wc + size_t(S * (j + 1) + q) * Nc;
          for (int n = 0; n < Nc; ++n) out[size_t(m) * Nc + n] += a * row[n];
        }
    };
    std::vector<double> blk;
    // forward: folded first layers
    for (int l = 0; l < L; ++l) {
      const int K = S * (l + 1) + u;
      const double* win = &h[p->latent[l].w[0]];  // [K, H]
      std::vector<double> f(win, win + size_t(K) * H);
      for (int j = 0; j < l; ++j) {
        fold_block(j, win, H, blk);
        std::copy(blk.begin(), blk.end(), f.begin() + size_t(S) * (j + 1) * H);
      }
      put(p->s_in[l], f);
      std::vector<double> sc(size_t(H) * u);
      for (int k = 0; k < H; ++k)
        for (int n = 0; n < u; ++n) sc[size_t(k) * u + n] = win[size_t(S * (l + 1) + n) * H + k];
      put(p->s_sct[l], sc);
    }
    const double* wro = &h[p->readout.w[0]];  // [SL1, Hr]
    {
      std::vector<double> f(wro, wro + size_t(SL1) * Hr);
      for (int j = 0; j < L; ++j) {
        fold_block(j, wro, Hr, blk);
        std::copy(blk.begin(), blk.end(), f.begin() + size_t(S) * (j + 1) * Hr);
      }
      put(p->s_ro0, f);
    }
    // reverse, by slot: rows follow the operand [d readout hidden | d z_{l+1} .. d z_{L-1}], columns the hidden units of latent l
    for (int l = 0; l < L; ++l) {
      std::vector<double> rs(size_t(Hr + S * (L - 1 - l)) * H);
      fold_block(l, wro, Hr, blk);
      for (int k = 0; k < Hr; ++k)
        for (int m = 0; m < H; ++m) rs[size_t(k) * H + m] = blk[size_t(m) * Hr + k];
      for (int l2 = l + 1; l2 < L; ++l2) {
        fold_block(l, &h[p->latent[l2].w[0]], H, blk);
        for (int k = 0; k < H; ++k)
          for (int m = 0; m < H; ++m) rs[size_t(Hr + S * (l2 - l - 1) + k) * H + m] = blk[size_t(m) * H + k];
      }
      put(p->s_rs[l], rs);
    }
    {
      std::vector<double> tb(size_t(Hr + S * L) * S);
      for (int k = 0; k < Hr; ++k)
        for (int n = 0; n < S; ++n) tb[size_t(k) * S + n] = wro[size_t(n) * Hr + k];
      for (int l2 = 0; l2 < L; ++l2)
        for (int k = 0; k < H; ++k)
          for (int n = 0; n < S; ++n) tb[size_t(Hr + S * l2 + k) * S + n] = h[p->latent[l2].w[0] + size_t(n) * H + k];
      put(p->s_rstb, tb);
    }
  }
  struct SetRef { const GemmMatSet* m; };
  std::vector<SetRef> proj_sets;
  if (p->op_proj) {
    const double sf = 1.0 / std::sqrt(c.avg_num_neighbors);
    auto put_set = [&](const GemmMatSet& fw, const GemmMatSet& bw, size_t wk_off, int ka) {
      for (int r = 0; r < Rr; ++r) {
        for (int k = 0; k < ka; ++k)
          for (int ch = 0; ch < u; ++ch) {
            const double v = sf * h[wk_off + (size_t(k) * Rr + r) * u + ch];
            h[fw.w + r * fw.w_bs + size_t(k) * u + ch] = v;
            h[bw.w + r * bw.w_bs + size_t(ch) * ka + k] = v;
          }
        gemm_pack_b(&h[fw.w + r * fw.w_bs], ka, u, &h[fw.wp + r * fw.wp_bs]);
        gemm_pack_b(&h[bw.w + r * bw.w_bs], u, ka, &h[bw.wp + r * bw.wp_bs]);
      }
      proj_sets.push_back({&fw});
      proj_sets.push_back({&bw});
    };
    for (int l = 0; l < L; ++l) put_set(p->s_pr[l], p->s_prt[l], p->o_wk[l], l == 0 ? S : c.latent_mlp_width);
    if (p->slot_form) put_set(p->s_pr0f, p->s_prt0f, p->o_s_wk0, S);
  }
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (c.dtype == AA_F64) {
    AA_CHECK_HIP(hipMemcpyAsync(dev_blob, h.data(), h.size() * 8, hipMemcpyHostToDevice, s));
  } else {
    std::vector<float> hf(h.begin(), h.end());
    // bf16x3 copies are derived from the ROUNDED fp32 matrices (bit patterns stored in float slots)
    auto splitw = [&](size_t w_off, int K, int N, size_t q_off) {
      gemm_pack_bf16x3(&hf[w_off], K, N, reinterpret_cast<unsigned*>(&hf[q_off]));
    };
    auto split_mlp = [&](const MlpLayout& m, int nlayers) {
      for (int i = 0; i < nlayers; ++i) {
        splitw(m.w[i], m.dims[i], m.dims[i + 1], m.wq[i]);
        splitw(m.wt[i], m.dims[i + 1], m.dims[i], m.wtq[i]);
      }
    };
    split_mlp(p->embed, c.embed_mlp_depth + 1);
    splitw(p->o_g0, S, p->ng0, p->o_g0q);
    splitw(p->o_g0t, p->ng0, S, p->o_g0tq);
    for (int l = 0; l < L; ++l) split_mlp(p->latent[l], c.latent_mlp_depth + 1);
    split_mlp(p->readout, c.readout_mlp_depth);
    for (const GemmMat* m : slot_mats) splitw(m->w, m->K, m->N, m->wq);
    for (const SetRef& sr : proj_sets)
      for (int r = 0; r < sr.m->count; ++r) splitw(sr.m->w + r * sr.m->w_bs, sr.m->K, sr.m->N, sr.m->wq + r * sr.m->wq_bs);
    if (p->chain_gemm) {
      const int SL = S * L, SL1 = p->SL1, N2 = SL + c.num_tensor;
      const float* rt = &hf[p->readout.wt[0]];          // [64, SL1]  (transposed first readout layer)
      const float* lt = &hf[p->latent[L - 1].wt[0]];    // [64, N2]   (transposed first layer of the last latent)
      // a: readout-reverse columns feeding the last latent; b: [readout-reverse ; latent-reverse] columns of the
      // earlier features (d_fcat[:, :SL]); c: latent-reverse columns of the tensor scalars (d_scal)
      std::vector<float> a(size_t(64) * S), b(size_t(128) * SL), cmat(size_t(64) * c.num_tensor);
      for (int k = 0; k < 64; ++k) {
        for (int n = 0; n < S; ++n) a[size_t(k) * S + n] = rt[size_t(k) * SL1 + SL + n];
        for (int n = 0; n < SL; ++n) {
          b[size_t(k) * SL + n] = rt[size_t(k) * SL1 + n];
          b[size_t(64 + k) * SL + n] = lt[size_t(k) * N2 + n];
        }
        for (int n = 0; n < c.num_tensor; ++n) cmat[size_t(k) * c.num_tensor + n] = lt[size_t(k) * N2 + SL + n];
      }
      gemm_pack_bf16x3(a.data(), 64, S, reinterpret_cast<unsigned*>(&hf[p->o_b3a_q]));
      gemm_pack_bf16x3(b.data(), 128, SL, reinterpret_cast<unsigned*>(&hf[p->o_b3b_q]));
      gemm_pack_bf16x3(cmat.data(), 64, c.num_tensor, reinterpret_cast<unsigned*>(&hf[p->o_b3c_q]));
    }
    for (int l = 0; l < L && l < 2; ++l) {
      if (!p->o_wkq[l]) continue;
      // (kProjMfma) env weights of layer l as one 64x64 bf16x3 layer per irrep: W_r[k][ch] = Wenv_l[k][r][ch]
      const size_t src = (l == 0 && p->o_wk0f) ? p->o_wk0f : p->o_wk[l];
      std::vector<float> wr(size_t(64) * 64);
      for (int r = 0; r < p->R; ++r) {
        for (int k = 0; k < 64; ++k)
          for (int ch = 0; ch < 64; ++ch) wr[size_t(k) * 64 + ch] = float(h[src + (size_t(k) * p->R + r) * 64 + ch]);
        gemm_pack_bf16x3(wr.data(), 64, 64, reinterpret_cast<unsigned*>(&hf[p->o_wkq[l] + size_t(r) * gemm_bf16x3_words(64, 64)]));
      }
    }
    if (p->o_g0fq) {
      // (kFoldEmb1) first stage behind the output layer of scalar_embed_mlp: W1 @ G0 and its transpose
      const int NG = p->ng0;
      std::vector<float> gf(size_t(64) * NG), gft(size_t(NG) * 64);
      for (int k = 0; k < 64; ++k)
        for (int q = 0; q < NG; ++q) {
          double v = 0.0;
          for (int m = 0; m < S; ++m) v += h[p->embed.w[1] + size_t(k) * S + m] * h[p->o_g0 + size_t(m) * NG + q];
          gf[size_t(k) * NG + q] = float(v);
          gft[size_t(q) * 64 + k] = float(v);
        }
      gemm_pack_bf16x3(gf.data(), 64, NG, reinterpret_cast<unsigned*>(&hf[p->o_g0fq]));
      gemm_pack_bf16x3(gft.data(), NG, 64, reinterpret_cast<unsigned*>(&hf[p->o_g0tfq]));
    }
    if (p->o_lat1in_fq) {
      // folded first layers (kFoldLatent): row block "lat_l" of a consumer's first layer <- Wout_l @ that block, in fp64 from the
      // normalised matrices, rounded once.  Row order of the consumers: [two-body | lat0 | scal1] (latent 1), [two-body | lat0 | lat1] (readout)
      const double* wo0 = &h[p->latent[0].w[1]];  // [64, 64] output layer of latent 0
      const double* wo1 = &h[p->latent[1].w[1]];  // [64, 64] output layer of latent 1
      auto folded = [&](const double* win, int K, const double* const* fold /* per 64-row block: Wout or nullptr */) {
        std::vector<float> out(size_t(K) * 64);
        for (int blk = 0; blk < K / 64; ++blk)
          for (int r = 0; r < 64; ++r)
            for (int n = 0; n < 64; ++n) {
              double v = 0.0;
              if (fold[blk]) {
                for (int m = 0; m < 64; ++m) v += fold[blk][size_t(r) * 64 + m] * win[size_t(blk * 64 + m) * 64 + n];
              } else {
                v = win[size_t(blk * 64 + r) * 64 + n];
              }
              out[size_t(blk * 64 + r) * 64 + n] = float(v);
            }
        return out;
      };
      const double* f1[3] = {nullptr, wo0, nullptr};
      const double* f2[3] = {nullptr, wo0, wo1};
      const std::vector<float> l1 = folded(&h[p->latent[1].w[0]], 2 * S + c.num_tensor, f1);
      const std::vector<float> r0 = folded(&h[p->readout.w[0]], 3 * S, f2);
      gemm_pack_bf16x3(l1.data(), 2 * S + c.num_tensor, 64, reinterpret_cast<unsigned*>(&hf[p->o_lat1in_fq]));
      gemm_pack_bf16x3(r0.data(), 3 * S, 64, reinterpret_cast<unsigned*>(&hf[p->o_ro0_fq]));
      // readout-reverse chain: d a1 = d ro_h @ (Wout_1 @ Wro[lat1 rows])^T, one layer instead of two
      std::vector<float> af(size_t(64) * 64);
      for (int k = 0; k < 64; ++k)       // readout hidden unit
        for (int m = 0; m < 64; ++m) {   // hidden unit of latent 1
          double v = 0.0;
          for (int n = 0; n < 64; ++n) v += wo1[size_t(m) * 64 + n] * h[p->readout.w[0] + size_t(2 * S + n) * 64 + k];
          af[size_t(k) * 64 + m] = float(v);
        }
      gemm_pack_bf16x3(af.data(), 64, 64, reinterpret_cast<unsigned*>(&hf[p->o_b3af_q]));
      // second layer of that chain, [readout' ; latent-1'] -> (d two-body | d lat0): with Wout_0^T folded into the lat0 columns it
      // yields d a_0 (before the moments' share and silu'), and the latent-0 reverse chain needs no output-layer reverse
      {
        const int SL = S * L;
        const float* rt = &hf[p->readout.wt[0]];
        const float* lt = &hf[p->latent[L - 1].wt[0]];
        const int N2 = SL + c.num_tensor, SL1_ = p->SL1;
        if (p->o_b3bf_q) {
        std::vector<double> b(size_t(128) * SL);
        for (int k = 0; k < 64; ++k)
          for (int n = 0; n < SL; ++n) {
            b[size_t(k) * SL + n] = rt[size_t(k) * SL1_ + n];
            b[size_t(64 + k) * SL + n] = lt[size_t(k) * N2 + n];
          }
        std::vector<float> bf(size_t(128) * SL);
        for (int k = 0; k < 128; ++k)
          for (int n = 0; n < SL; ++n) {
            double v = b[size_t(k) * SL + n];
            if (n >= S) {  // lat0 column block -> hidden unit m = n - S of latent 0
              v = 0.0;
              for (int q = 0; q < 64; ++q) v += b[size_t(k) * SL + S + q] * wo0[size_t(n - S) * 64 + q];
            }
            bf[size_t(k) * SL + n] = float(v);
          }
        gemm_pack_bf16x3(bf.data(), 128, SL, reinterpret_cast<unsigned*>(&hf[p->o_b3bf_q]));
        }
      }
    }
    AA_CHECK_HIP(hipMemcpyAsync(dev_blob, hf.data(), hf.size() * 4, hipMemcpyHostToDevice, s));
  }
  AA_CHECK_HIP(hipStreamSynchronize(s));  // host staging vectors die at return
  return AA_OK;
}

// ------------------------------------------------------------------------------------------------
// workspace layout
// ------------------------------------------------------------------------------------------------
struct Workspace {
  size_t vec, sh, emb0, emb, w0, fcat, g_fcat, g_envw, g_w0, g_emb, g_emb0, g_sh, g_ro_last, g_aenv, e_edge, q_op, bvec_op, gm_op, mom_op, dx2s_op, trev, dvec, vir_part, tiles;
  size_t g_scal[AA_MAX_LAYERS];
  size_t se_h[AA_MAX_MLP_LAYERS], g_se_h[AA_MAX_MLP_LAYERS];
  size_t envw[AA_MAX_LAYERS], x2s[AA_MAX_LAYERS], tf[AA_MAX_LAYERS], scal[AA_MAX_LAYERS];
  size_t lat_h[AA_MAX_LAYERS][AA_MAX_MLP_LAYERS], g_lat_h[AA_MAX_MLP_LAYERS];
  size_t ro_h[AA_MAX_MLP_LAYERS], g_ro_h[AA_MAX_MLP_LAYERS];
  size_t g_tf[2];
  size_t total;
};

// [E,D] slots that edge_backward sums into d sh: general kernels 1 (accumulated); specialised L+1; operator kernels one
// per 64-channel slice for the x1 path plus one per 64-wide block of every layer's env input
static int num_gsh_slots(const aa_model_plan* p) {
  const aa_model_config& c = p->cfg;
  if (p->tp_op >= 0) {
    int n = c.num_tensor / 64;
    for (int l = 0; l < c.num_layers; ++l) n += (l == 0 ? c.num_scalar : c.latent_mlp_width) / 64;
    return n;
  }
  return p->use_spec ? c.num_layers + 1 : 1;
}

static Workspace layout_workspace(const aa_model_plan* p, int64_t N, int64_t E, int with_forces) {
  Workspace w{};
  const aa_model_config& c = p->cfg;
  const size_t es = p->esize();
  size_t o = 0;
  auto take = [&](size_t elems) {
    size_t r = o;
    o += (elems * es + 255) / 256 * 256;
    return r;
  };
  const int S = c.num_scalar, u = c.num_tensor, L = c.num_layers, S0 = c.embed_dim;
  const size_t Ez = size_t(E), Nz = size_t(N);
  w.vec = take(Ez * 4);
  w.sh = take(Ez * p->D);
  w.emb0 = take(Ez * S0);
  for (int i = 0; i < c.embed_mlp_depth; ++i) w.se_h[i] = take(Ez * c.embed_mlp_width);
  w.emb = take(Ez * S);
  w.w0 = take(Ez * p->W);
  w.fcat = take(Ez * p->SL1);
  size_t dmax = 1;
  for (int l = 0; l < L; ++l) {
    if (!p->env_mom) w.envw[l] = take(Ez * p->W);
    w.x2s[l] = take(Nz * u * p->D);
    if (l < L - 1 && p->chain_pair < 0 && !p->env_mom) {
      w.tf[l] = take(Ez * u * c.tps[l].dout);
      dmax = std::max(dmax, size_t(c.tps[l].dout));
    }
    w.scal[l] = take(Ez * u);
    for (int i = 0; i < c.latent_mlp_depth; ++i) w.lat_h[l][i] = take(Ez * c.latent_mlp_width);
  }
  for (int i = 0; i < c.readout_mlp_depth; ++i) w.ro_h[i] = take(Ez * c.readout_mlp_width);
  if (p->chain_gemm) w.e_edge = take(Ez);
  if (p->embed_fused) w.trev = take(Ez * 8);
  if (p->fused_fwd) w.tiles = take(((3 * Nz + 8) * sizeof(int32_t) + es - 1) / es);  // class lists + counters of the team form
  if (with_forces) {
    w.dvec = take(Ez * 4);
    w.vir_part = take((size_t(kVirialBlocks) * 9 * sizeof(double) + es - 1) / es);
  }
  if (p->tp_op >= 0) {
    w.bvec_op = take(Nz * L * p->D * u);
    w.mom_op = take(Nz * p->D * size_t(std::max(c.num_scalar, c.latent_mlp_width)));
  }
  if (with_forces) {
    w.g_fcat = take(Ez * p->SL1);
    for (int i = 0; i < c.readout_mlp_depth; ++i) w.g_ro_h[i] = take(Ez * c.readout_mlp_width);
    for (int i = 0; i < c.latent_mlp_depth; ++i) w.g_lat_h[i] = take(Ez * c.latent_mlp_width);
    for (int l = 0; l < L; ++l) w.g_scal[l] = take(Ez * u);
    if (!p->env_mom) w.g_envw = take(Ez * p->W);
    if (p->env_mom) w.g_aenv = take(Ez * size_t(std::max(S, c.latent_mlp_width)));
    w.g_w0 = take(Ez * p->W);
    if (L > 1 && p->chain_pair < 0) {
      w.g_tf[0] = take(Ez * u * dmax);
      w.g_tf[1] = L > 2 ? take(Ez * u * dmax) : w.g_tf[0];
    }
    w.g_emb = take(Ez * S);
    for (int i = 0; i < c.embed_mlp_depth; ++i) w.g_se_h[i] = take(Ez * c.embed_mlp_width);
    w.g_emb0 = take(Ez * S0);
    w.g_sh = take(Ez * p->D * num_gsh_slots(p));  // slot 0: x1 path of layer 0; slot l+1: env path of layer l (see num_gsh_slots)
    if (p->tp_op >= 0) {
      w.q_op = take(Nz * L * p->D * u);
      w.gm_op = take(Nz * p->D * size_t(std::max(c.num_scalar, c.latent_mlp_width)));
      if (p->op_proj) w.dx2s_op = take(Nz * p->D * u);
    }
  }
  w.total = o;
  return w;
}

extern "C" size_t aa_model_workspace_bytes(const aa_model_plan* plan, int64_t N, int64_t E, int with_forces) {
  if (!plan) return 0;
  return layout_workspace(plan, N, E, with_forces).total;
}

// ------------------------------------------------------------------------------------------------
// pipeline
// ------------------------------------------------------------------------------------------------
namespace {

inline Seg seg(void* p, int ld, int n) { return Seg{p, ld, n}; }

// optional per-stage timing with HIP events recorded on the launch stream
struct StageProfile {
  std::vector<std::string> names;
  std::vector<hipEvent_t> events;
  std::vector<double> bytes, flops;  // algorithmic HBM bytes / flops of the launch that ended at this mark
  int mark(const char* name, hipStream_t s, double by, double fl) {
    hipEvent_t e;
    AA_CHECK_HIP(hipEventCreate(&e));
    AA_CHECK_HIP(hipEventRecord(e, s));
    names.emplace_back(name);
    events.push_back(e);
    bytes.push_back(by);
    flops.push_back(fl);
    return AA_OK;
  }
  void clear() {
    for (hipEvent_t e : events) (void)hipEventDestroy(e);
    events.clear();
    names.clear();
    bytes.clear();
    flops.clear();
  }
};

template <typename T>
struct Runner {
  const aa_model_plan* p;
  const T* wts;
  char* ws;
  Workspace w;
  int64_t E, N;
  hipStream_t stream;
  StageProfile* prof = nullptr;
  bool want_forces = false;    // (set by run_model before forward())
  bool ro_grad_done = false;   // forward(): readout_reduce also wrote d E / d (last readout hidden layer) -- no readout_backward launch
  bool fwd_b3_done = false;    // forward_fused(): the readout-reverse chain ran in the tail of the fused forward (FusedFwdArgs::tail)

  // per_edge / per_atom: operand elements the launch must move (each distinct operand row once); see DESIGN.md §5
  int mark(const char* name, double per_edge = 0, double per_atom = 0, double flops = 0) {
    return prof ? prof->mark(name, stream, sizeof(T) * (per_edge * double(E) + per_atom * double(N)), flops) : AA_OK;
  }
  static double seg_elems(const SegList& s) {
    double n = 0;
    for (int i = 0; i < s.count; ++i)
      if (s.s[i].p) n += s.s[i].n;
    return n;
  }
  // A read + C written (+ accumulated segments re-read) + Z + add operands of one GEMM, elements per row
  static double gemm_row_elems(const GemmArgs& g, bool a_from_hbm) {
    double n = (a_from_hbm ? seg_elems(g.a) : 0) + seg_elems(g.c);
    for (int i = 0; i < g.c.count; ++i)
      if (g.c.s[i].p && g.c_accum[i]) n += g.c.s[i].n;
    if (g.has_z) n += seg_elems(g.z);
    if (g.has_add) n += seg_elems(g.add);
    return n;
  }

  T* buf(size_t off) const { return reinterpret_cast<T*>(ws + off); }
  const T* wt(size_t off) const { return wts + off; }

  int gemm(const SegList& a, int act_a, const GemmMat& m, const SegList& c, const SegList* z = nullptr, const SegList* add = nullptr, int act_lo = 0,
           int act_hi = 0) {
    return gemm(a, act_a, wt(m.w), wt(m.wp), wt(m.wq), m.K, m.N, c, nullptr, z, add, act_lo, act_hi);
  }
  int gemm(const SegList& a, int act_a, const T* B, const T* Bp, const T* Bq, int K, int Nn, const SegList& c,
           const int* accum, const SegList* z, const SegList* add = nullptr, int act_lo = 0, int act_hi = 0) {
    GemmArgs g{};
    g.act_lo = act_lo;
    g.act_hi = act_hi;
    g.M = E;
    g.K = K;
    g.N = Nn;
    g.a = a;
    g.B = B;
    g.Bp = Bp;
    g.Bq = sizeof(T) == 4 ? Bq : nullptr;
    g.c = c;
    for (int i = 0; i < 3; ++i) g.c_accum[i] = accum ? accum[i] : 0;
    g.has_z = z ? 1 : 0;
    if (z) g.z = *z;
    g.act_a = act_a;
    g.act_kind = act_now;
    g.force_kernel = p->opt.gemm_valu ? 3 : (p->opt.gemm_fp32_mfma ? 1 : 0);
    g.opt_v1 = p->opt.gemm_v1;
    g.opt_lds_epilogue = p->opt.gemm_lds_epilogue;
    g.opt_f64_column_loop = p->opt.f64_column_loop;
    g.opt_f64_rows = p->opt.f64_rows;
    g.has_add = add ? 1 : 0;
    if (add) g.add = *add;
    if (int rc = launch_gemm<T>(g, stream)) return rc;
    if (!prof) return AA_OK;
    char nm[32];
    snprintf(nm, sizeof(nm), "gemm_%dx%d", K, Nn);
    return mark(nm, gemm_row_elems(g, true), 0, 2.0 * double(E) * K * Nn);
  }

  int act_now = AA_ACT_SILU;  // nonlinearity of the MLP whose layers are being launched (gemm() reads it)
  // forward of a ScalarMLPFunction: hidden pre-activations to h[i]; final linear output to `out`
  // which: 0 scalar_embed_mlp, 1 latent, 2 readout (selects the nonlinearity)
  int mlp_fwd(const MlpLayout& m, int nlayers, const SegList& in, const size_t* h, const SegList& out, int which) {
    act_now = p->cfg.act_kind[which];
    SegList a = in;
    for (int i = 0; i < nlayers; ++i) {
      SegList c;
      if (i < nlayers - 1) {
        c.count = 1;
        c.s[0] = seg(buf(h[i]), m.dims[i + 1], m.dims[i + 1]);
      } else {
        c = out;
      }
      if (int rc = gemm(a, i > 0, wt(m.w[i]), wt(m.wp[i]), wt(m.wq[i]), m.dims[i], m.dims[i + 1], c, nullptr, nullptr)) return rc;
      a = c;
    }
    return AA_OK;
  }

  // reverse: g_out (grad of final output) -> g_in (with per-segment accumulate flags)
  // add_last: optional extra gradient wrt the ACTIVATED last hidden layer (moments path), added before silu'
  int mlp_bwd(const MlpLayout& m, int nlayers, const SegList& g_out, const size_t* h, const size_t* g_h,
              const SegList& g_in, const int* g_in_accum, int which, const SegList* add_last = nullptr) {
    act_now = p->cfg.act_kind[which];
    SegList a = g_out;
    for (int i = nlayers - 1; i >= 0; --i) {
      SegList c, z;
      const int* acc = nullptr;
      const SegList* zp = nullptr;
      if (i > 0) {
        c.count = 1;
        c.s[0] = seg(buf(g_h[i - 1]), m.dims[i], m.dims[i]);
        z.count = 1;
        z.s[0] = seg(buf(h[i - 1]), m.dims[i], m.dims[i]);
        zp = &z;
      } else {
        c = g_in;
        acc = g_in_accum;
      }
      const SegList* addp = (add_last && i == nlayers - 1 && i > 0) ? add_last : nullptr;
      if (int rc = gemm(a, 0, wt(m.wt[i]), wt(m.wtp[i]), wt(m.wtq[i]), m.dims[i + 1], m.dims[i], c, acc, zp, addp)) return rc;
      a = c;
    }
    return AA_OK;
  }

  EdgeGeomArgs geom(const aa_graph* g, const void* pos) const {
    const aa_model_config& c = p->cfg;
    EdgeGeomArgs a{};
    a.E = E;
    a.N = N;
    a.center = g->center;
    a.nbr = g->nbr;
    a.types = g->types;
    a.pos = pos;
    a.shift_vec = g->shift_vec;
    a.num_types = c.num_types;
    a.num_bessels = c.num_bessels;
    a.l_max = c.l_max;
    a.S0 = c.embed_dim;
    a.poly_p = c.poly_p;
    a.rmax_recip = wt(p->o_rmax);
    a.bessel_w = wt(p->o_bessel);
    a.center_embed = wt(p->o_cemb);
    a.neighbor_embed = wt(p->o_nemb);
    a.basis_w = wt(p->o_basis);
    a.embed_kind = c.embed_kind;
    a.spline_span = c.spline_span;
    a.emb_tab = p->o_embtab ? wt(p->o_embtab) : nullptr;
    a.vec = buf(w.vec);
    a.sh = buf(w.sh);
    a.emb0 = buf(w.emb0);
    return a;
  }

  ReadoutArgs readout_args(const aa_graph* g, void* atom_energy) const {
    const aa_model_config& c = p->cfg;
    ReadoutArgs r{};
    r.E = E;
    r.N = N;
    r.rowptr = g->rowptr;
    r.center = g->center;
    r.types = g->types;
    if (c.readout_mlp_depth > 0) {
      r.h = buf(w.ro_h[c.readout_mlp_depth - 1]);
      r.ld = c.readout_mlp_width;
      r.H = c.readout_mlp_width;
      r.act = 1;
    } else {
      r.h = buf(w.fcat);
      r.ld = p->SL1;
      r.H = p->SL1;
      r.act = 0;
    }
    r.w = wt(p->o_ro_last);
    r.act_kind = c.act_kind[2];
    r.factor = 1.0 / std::sqrt(2.0 * c.avg_num_neighbors);
    r.scales = c.has_scales ? wt(p->o_scales) : nullptr;
    r.shifts = c.has_shifts ? wt(p->o_shifts) : nullptr;
    r.atom_energy = atom_energy;
    r.edge_sum = (p->chain_gemm && atom_energy) ? buf(w.e_edge) : nullptr;  // written by the forward readout chain
    return r;
  }

  // owned-block hint of the graph (per-atom kernels of the fast paths only visit these atoms)
  int64_t atom_begin(const aa_graph* g) const { return g->atom_end > g->atom_begin ? g->atom_begin : 0; }
  int64_t atom_end(const aa_graph* g) const { return g->atom_end > g->atom_begin ? g->atom_end : N; }

  TpChainArgs chain_args(const aa_graph* g) const {
    const aa_model_config& c = p->cfg;
    TpChainArgs a{};
    a.E = E;
    a.N = N;
    a.rowptr = g->rowptr;
    a.u = c.num_tensor;
    a.sh = buf(w.sh);
    a.ld_sh = p->D;
    a.w0 = buf(w.w0);
    a.ld_w0 = p->W;
    a.wenv0 = buf(w.envw[0]);
    a.ld_we0 = p->W;
    a.wenv1 = buf(w.envw[1]);
    a.ld_we1 = p->W;
    a.weights0 = wt(p->o_tpw[0]);
    a.weights1 = wt(p->o_tpw[1]);
    a.coupling = c.tps[0].coupling;
    a.sf = 1.0 / std::sqrt(c.avg_num_neighbors);
    a.x2s0 = buf(w.x2s[0]);
    a.x2s1 = buf(w.x2s[1]);
    a.scal1 = buf(w.scal[1]);
    a.ld_scal = c.num_tensor;
    a.ld_gscal = c.num_tensor;
    a.ld_gw0 = p->W;
    a.ld_gwe = p->W;
    a.ld_gsh = p->D;
    return a;
  }

  // ---- fused GEMM chains (plan->chain_gemm) -------------------------------------------------------------
  static ChainLayer chain_layer(int64_t M, const SegList& a, int act_a, const void* Bq, int K, int Nn, const SegList& c,
                                const int* accum, const SegList* z, const SegList* add, int use_prev, int keep_tile,
                                int keep_act) {
    ChainLayer L{};
    L.g.M = M;
    L.g.K = K;
    L.g.N = Nn;
    L.g.a = a;
    L.g.Bq = Bq;
    L.g.c = c;
    for (int i = 0; i < 3; ++i) L.g.c_accum[i] = accum ? accum[i] : 0;
    L.g.has_z = z ? 1 : 0;
    if (z) L.g.z = *z;
    L.g.has_add = add ? 1 : 0;
    if (add) L.g.add = *add;
    L.g.act_a = act_a;
    L.use_prev = use_prev;
    L.keep_tile = keep_tile;
    L.keep_act = keep_act;
    return L;
  }
  int run_chain(ChainArgs& ca, const char* tag) {
    ca.M = E;
    if (int rc = launch_gemm_chain(ca, stream)) return rc;
    if (!prof) return AA_OK;
    char nm[32];
    int o = snprintf(nm, sizeof(nm), "gc");
    double elems = 0, fl = 0;
    for (int i = 0; i < ca.nlayers; ++i) {
      if (o < 28) o += snprintf(nm + o, sizeof(nm) - o, "_%dx%d", ca.L[i].g.K, ca.L[i].g.N);
      elems += gemm_row_elems(ca.L[i].g, true);  // a is an empty list for use_prev layers (kept tile in registers)
      if (ca.L[i].a_mode == 2) elems += 2.0 * ca.L[i].g.a.s[0].n;  // (+ the add and z rows of the operand transform)
      fl += 2.0 * double(E) * ca.L[i].g.K * ca.L[i].g.N;
    }
    (void)tag;
    return mark(nm, elems, 0, fl);
  }

  TpMomArgs mom_args(const aa_graph* g) const {
    const aa_model_config& c = p->cfg;
    TpMomArgs m{};
    m.c = chain_args(g);
    m.c.atom0 = atom_begin(g);
    m.c.N = atom_end(g);
    m.c.wenv0 = m.c.wenv1 = nullptr;
    m.a0 = buf(w.emb);
    m.ld_a0 = c.num_scalar;
    m.ka0 = c.num_scalar;
    m.a1 = buf(w.lat_h[0][c.latent_mlp_depth - 1]);
    m.ld_a1 = c.latent_mlp_width;
    m.ka1 = c.latent_mlp_width;
    m.wk0 = wt(p->o_wk[0]);
    m.wt0 = wt(p->o_wt[0]);
    m.wk1 = wt(p->o_wk[1]);
    m.wt1 = wt(p->o_wt[1]);
    m.waves_per_block = p->opt.moments_waves_per_block;
    return m;
  }

  // slot form of the single-layer pipeline (aa_model_plan::slot_form); the debug taps name tensors it never forms
  bool use_slot() const { return p->slot_form && !p->taps; }

  TpOpArgs op_args(const aa_graph* g, int l) const {
    const aa_model_config& c = p->cfg;
    TpOpArgs o{};
    o.N = atom_end(g);
    o.atom0 = atom_begin(g);
    o.E = E;
    o.rowptr = g->rowptr;
    o.u = c.num_tensor;
    o.sh = buf(w.sh);
    o.ld_sh = p->D;
    o.w0 = buf(w.w0);
    o.ld_w0 = p->W;
    o.coupling = c.tps[0].coupling;
    o.sf = 1.0 / std::sqrt(c.avg_num_neighbors);
    for (int m = 0; m < c.num_layers && m < 3; ++m) {
      o.x2s[m] = buf(w.x2s[m]);
      o.tpw[m] = wt(p->o_tpw[m]);
    }
    if (l == 0) {
      o.a = buf(w.emb);
      o.ld_a = o.ka = c.num_scalar;
      o.act = 0;
    } else {
      o.a = buf(w.lat_h[l - 1][c.latent_mlp_depth - 1]);
      o.ld_a = o.ka = c.latent_mlp_width;
      o.act = 1;
    }
    o.wk = wt(p->o_wk[l]);
    o.wt = wt(p->o_wt[l]);
    if (use_slot()) {
      // env inputs are hidden pre-activations everywhere: scalar_embed_mlp's (env weights behind its output layer) for layer 0,
      // slot l of the dense-net buffer (= z_{l-1}) afterwards
      if (l == 0) {
        o.a = buf(w.se_h[c.embed_mlp_depth - 1]);
        o.ld_a = o.ka = c.embed_mlp_width;
        o.act = 1;
        o.wk = wt(p->o_s_wk0);
        o.wt = wt(p->o_s_wt0);
      } else {
        o.a = buf(w.fcat) + size_t(c.num_scalar) * l;
        o.ld_a = p->SL1;
      }
    }
    o.ld_scal = c.num_tensor;
    o.ld_gscal = c.num_tensor;
    o.q = buf(w.q_op);
    o.num_layers = c.num_layers;
    {
      // split form (default): per-atom vectors through HBM, edge loops in lean kernels (tp_operator_fused: fused form)
      const bool nosplit = p->opt.tp_operator_fused != 0;
      o.bvec = (!nosplit && w.bvec_op) ? buf(w.bvec_op) : nullptr;
      o.gmbuf = (!nosplit && w.bvec_op && w.gm_op) ? buf(w.gm_op) : nullptr;
      o.mbuf = (!nosplit && w.mom_op) ? buf(w.mom_op) : nullptr;
    }
    o.ld_gw0 = p->W;
    o.ld_gsh = p->D;
    o.ka_lds = std::max(c.num_scalar, c.latent_mlp_width);
    o.env_mfma = (sizeof(T) == 8 && !p->opt.op_env_vector && (o.ld_a % 2) == 0) ? 1 : 0;
    return o;
  }

  // env projections of the operator kernels as batched linear-layer launches (aa_model_plan::op_proj): where there are enough
  // atoms to fill the chip with 128-row tiles
  bool use_proj(const aa_graph* g) const {
    if (!(p->op_proj && w.bvec_op && w.mom_op && !p->opt.tp_operator_fused)) return false;
    if (p->opt.op_proj_gemm == 1) return true;
    return atom_end(g) - atom_begin(g) >= 4096;
  }
  // one problem per spherical-harmonic component j (rows: atoms), weight matrix number r(j)
  int proj_gemm(const TpOpArgs& o, const GemmMatSet& ms, const T* a_base, int64_t a_ld, int64_t a_bs, T* c_base, int64_t c_ld, int64_t c_bs) {
    GemmArgs gm{};
    gm.M = o.N - o.atom0;
    gm.K = ms.K;
    gm.N = ms.N;
    gm.a = SegList{1, {seg(const_cast<T*>(a_base) + o.atom0 * a_ld, int(a_ld), ms.K)}};
    gm.c = SegList{1, {seg(c_base + o.atom0 * c_ld, int(c_ld), ms.N)}};
    gm.B = wt(ms.w);
    gm.Bp = wt(ms.wp);
    gm.Bq = sizeof(T) == 4 ? wt(ms.wq) : nullptr;
    gm.act_kind = AA_ACT_SILU;
    gm.force_kernel = p->opt.gemm_valu ? 3 : (p->opt.gemm_fp32_mfma ? 1 : 0);
    gm.opt_v1 = p->opt.gemm_v1;
    gm.opt_lds_epilogue = p->opt.gemm_lds_epilogue;
    gm.opt_f64_column_loop = p->opt.f64_column_loop;
    gm.opt_f64_rows = p->opt.f64_rows;
    gm.batch = p->D;
    gm.a_bs = a_bs;
    gm.c_bs = c_bs;
    gm.b_bs = int64_t(ms.w_bs);
    gm.bp_bs = int64_t(ms.wp_bs);
    gm.bq_bs = int64_t(ms.wq_bs);
    for (int j = 0, r = 0; j < p->D; ++j) {
      if (j >= (r + 1) * (r + 1)) ++r;
      gm.bsel4 |= static_cast<unsigned long long>(r) << (4 * j);
    }
    if (int rc = launch_gemm<T>(gm, stream)) return rc;
    return mark("op_proj_gemm", 0, double(p->D) * (ms.K + ms.N), 2.0 * double(gm.M) * p->D * ms.K * ms.N);
  }
  // forward of tensor-product layer l on the operator kernels
  int run_op_fwd(const aa_graph* g, int l) {
    const int u = p->cfg.num_tensor;
    TpOpArgs o = op_args(g, l);
    o.scal = buf(w.scal[l]);
    if (use_proj(g)) {
      o.proj_gemm = 1;
      if (int rc = launch_tp_op<T>(p->tp_op, l, false, o, stream, 1)) return rc;
      if (int rc = mark("tp_op_moments", p->D + o.ka)) return rc;
      const GemmMatSet& ms = (l == 0 && use_slot()) ? p->s_pr0f : p->s_pr[l];
      if (int rc = proj_gemm(o, ms, buf(w.mom_op), int64_t(p->D) * o.ka, o.ka, buf(w.x2s[l]), int64_t(p->D) * u, u)) return rc;
      if (int rc = launch_tp_op<T>(p->tp_op, l, false, o, stream, 2)) return rc;
      return mark("tp_op_fwd", p->W + u, double(l + 1) * p->D * u);
    }
    if (int rc = launch_tp_op<T>(p->tp_op, l, false, o, stream)) return rc;
    return mark("tp_op_fwd", p->D + o.ka + p->W + u, double(l + 1) * p->D * u);
  }
  // reverse of tensor-product layer l on the operator kernels: d scal_m -> d w0 / d Y (layer 0), d (env input) -> g_aenv
  int run_op_bwd(const aa_graph* g, int l) {
    const aa_model_config& c = p->cfg;
    const int S = c.num_scalar, u = c.num_tensor, L = c.num_layers, W = p->W;
    TpOpArgs o = op_args(g, l);
    for (int m = 0; m < L; ++m) o.gscal[m] = buf(w.g_scal[m]);
    o.g_w0 = buf(w.g_w0);
    o.gsh_x1 = buf(w.g_sh);
    size_t slot = size_t(u / 64);
    for (int m = 0; m < l; ++m) slot += size_t(m == 0 ? S : c.latent_mlp_width) / 64;
    o.gsh_env = buf(w.g_sh) + slot * size_t(E) * p->D;
    o.g_a = buf(w.g_aenv);
    o.ld_ga = o.ka;
    // (split form: every forward kernel of this step stored its B_l next to the others -- the vectors the layer-0 reverse needs)
    o.bvec_ready = (o.bvec && !p->opt.op_recompute_bvecs) ? 1 : 0;
    const double elems = p->D + W + u + 2 * o.ka + p->D + (l == 0 ? W + double(L - 1) * u + p->D : 0);
    if (use_proj(g) && w.dx2s_op) {
      o.proj_gemm = 1;
      o.dx2s = buf(w.dx2s_op);
      if (int rc = launch_tp_op<T>(p->tp_op, l, true, o, stream, 1)) return rc;
      if (int rc = mark("tp_op_bwd", elems - 2 * o.ka - p->D, double(L) * p->D * u)) return rc;
      const GemmMatSet& ms = (l == 0 && use_slot()) ? p->s_prt0f : p->s_prt[l];
      if (int rc = proj_gemm(o, ms, buf(w.dx2s_op), int64_t(p->D) * u, u, buf(w.gm_op), int64_t(p->D) * o.ka, o.ka)) return rc;
      if (int rc = launch_tp_op<T>(p->tp_op, l, true, o, stream, 2)) return rc;
      return mark("tp_op_edge_env", 2 * o.ka + p->D);
    }
    if (int rc = launch_tp_op<T>(p->tp_op, l, true, o, stream)) return rc;
    return mark("tp_op_bwd", elems, double(L) * p->D * u);
  }

  TpOperand implicit(size_t w_off) const {
    TpOperand o{};
    o.sh = buf(w.sh);
    o.ld_sh = p->D;
    o.w = buf(w_off);
    o.ldw = p->W;
    return o;
  }

  // the whole forward in one launch (aa_fused.hip): every center atom's edge segment fits one 32-row MFMA tile
  bool use_fused_fwd(const aa_graph* g) const {
    if (!(sizeof(T) == 4 && p->fused_fwd && !p->taps && g->max_degree > 0 && g->max_degree <= kFusedMaxDegree)) return false;
    // (three species + the team exchange area exceed the 160 KB of LDS: such graphs run the staged pipeline)
    if (fused_fwd_lds_bytes(p->cfg.num_types, g->max_degree > 32) > size_t(160) * 1024) return false;
    if (g->max_degree <= 32) return true;  // one full-ish tile per atom: faster than the staged forward at every size
    // Team form (2 / 4 tiles per atom): a tile costs the same whether 32 or 12 of its rows carry an edge, so it pays where
    // the step is latency-bound (few tiles: one launch instead of seven) or the tiles are nearly full.  Measured on Si boxes
    // at r_max 6 / 7 (44 / 73 edges per atom, profiles/archive/r03_p_*): 216-512 atoms 17-27 % faster than the staged step, 1 728
    // atoms -2 % / +8 %, 10 648 atoms +10 % / +18 % slower (69 % / 57 % of the tile rows in use).
    // (num_edges IS the edge count of the active block: the atom-block hint promises that every center with edges lies in
    // [atom_begin, atom_end) -- verified on the device by graph_hint_check_kernel -- so rowptr[atom_end] - rowptr[atom_begin],
    // which the host does not hold, equals g->num_edges)
    const int64_t n_active = g->atom_end > g->atom_begin ? g->atom_end - g->atom_begin : g->num_atoms;
    const int64_t tiles = n_active * (g->max_degree <= 64 ? 2 : 4);  // (upper bound: every atom at the class of the longest segment)
    if (p->opt.fused_forward == 2 || p->opt.fused_forward == 4) return true;  // (A/B: the team / mixed form whenever segments fit it)
    // Round 5: a box whose AVERAGE atom fills most of one tile but whose longest segment is a little over 32 (thermal disorder: a
    // handful of Si atoms with 33..37 neighbours after 50 fs at 300 K) takes the MIXED form -- one-tile kernel over all atoms, the
    // team kernel over the long ones only (launch_fused_fwd) -- instead of giving up the fused forward for the whole box
    // (tools/md_loop.py at C4: staged 11.3 ms, team form for every atom 12.2 ms per step; profiles/r05_v2*_md_loop_*).
    const double fill = double(g->num_edges) / (32.0 * double(std::max<int64_t>(n_active, 1)));
    if (fill >= 0.6 && fill <= 1.15) return true;
    return tiles <= kFusedTeamTilesSmall || double(g->num_edges) >= 0.85 * 32.0 * double(tiles);
  }
  // the reverse tail in one launch (aa_fused_bwd.hip): same eligibility as the fused forward + the two-body table of the reverse
  bool use_fused_tail(const aa_graph* g) const { return use_fused_fwd(g) && p->fused_tail; }
  // the staged forward chains in the same folded form as the fused forward (kFoldEmb1, kFoldLatent): the hidden activations a_e /
  // a_0 are stored where the embedding / lat_0 used to be (ChainLayer::kept_out), consumers run on the folded matrices.  What
  // graphs with long segments -- dense systems, too large for the team form -- run.
  bool fold_staged() const {
    return kFoldEmb1 && kFoldLatent && p->chain_gemm && p->tp_op < 0 && p->o_g0fq && p->o_lat1in_fq && p->o_wk0f && p->cfg.num_layers == 2 && !p->taps &&
           !p->opt.staged_no_fold;
  }
  // did the forward of this step leave a_e in the embedding's slot (folded first stage / env weights in the reverse)?
  bool folded_fwd(const aa_graph* g) const { return use_fused_fwd(g) || fold_staged(); }
#ifdef AA_EXPERIMENTAL_TAIL
  int backward_fused_tail(const aa_graph* g, void* forces) {
    const aa_model_config& c = p->cfg;
    const int u = c.num_tensor;
    FusedTailArgs a{};
    a.N = N;
    a.atom0 = atom_begin(g);
    a.atom_end = atom_end(g);
    a.rowptr = g->rowptr;
    a.nbr = g->nbr;
    a.types = g->types;
    a.num_types = c.num_types;
    a.embed_kind = c.embed_kind;
    a.spline_span = c.spline_span;
    a.poly_p = float(c.poly_p);
    auto wf = [&](size_t off) { return reinterpret_cast<const float*>(wt(off)); };
    auto bf = [&](size_t off) { return reinterpret_cast<float*>(buf(off)); };
    a.rmax_recip = wf(p->o_rmax);
    a.bessel_w = wf(p->o_bessel);
    a.emb_tab = wf(p->o_embtab);
    int ns = 0;
    {
      const float* Wk = wf(p->o_wtk[0]);  // GM: 4 blocks of 16 channels x R x 64
      for (int cblk = 0; cblk < 4; ++cblk) {
        a.wstep[ns][0] = Wk + size_t(cblk) * 16 * p->R * 64;
        a.wstep[ns][1] = Wk + size_t(cblk) * 16 * p->R * 64 + 1536;
        ++ns;
      }
    }
    auto add_layer = [&](const float* Wq, int KC, bool tail_order) {  // one tile pair (64 outputs), KC 32-deep chunks
      for (int i = 0; i < KC; ++i) {
        const int kc = tail_order ? fused_bwd_tail_chunk_order(p->R, i) : i;  // (the kernel consumes the w0 chunks half-major)
        a.wstep[ns][0] = Wq + size_t(kc) * 64 * 24;
        a.wstep[ns][1] = Wq + (size_t(KC) + kc) * 64 * 24;
        ++ns;
      }
    };
    add_layer(wf(p->o_g0tq), 2 + 2 * p->R, true);
    add_layer(wf(p->embed.wtq[1]), 2, false);
    add_layer(wf(p->embed.wtq[0]), 2, false);
    if (ns != fused_bwd_tail_num_steps(p->R) || ns > kFusedMaxSteps) return fail(AA_ERR_INVALID, "fused reverse tail: program length mismatch");
    a.tpw0 = wf(p->o_tpw[0]);
    a.tpw1 = wf(p->o_tpw[1]);
    a.coupling = c.tps[0].coupling;
    a.sf = float(1.0 / std::sqrt(c.avg_num_neighbors));
    a.vec = bf(w.vec);
    a.w0 = bf(w.w0);
    a.emb = bf(w.emb);
    a.se_h = bf(w.se_h[0]);
    a.x2s0 = bf(w.x2s[0]);
    a.x2s1 = bf(w.x2s[1]);
    a.gscal0 = bf(w.g_scal[0]);
    a.gscal1 = bf(w.g_scal[1]);
    a.g_tb = bf(w.g_fcat);
    a.ld_gtb = p->SL1;
    a.gsh_env1 = bf(w.g_sh) + size_t(2) * size_t(E) * p->D;  // slot of the layer-1 env path (tp_mom_bwd_last)
    const bool gather = g->t_rowptr && g->t_perm;
    const bool fuse_edge = gather && p->opt.fused_tail != 2;
    if (fuse_edge) {
      a.dvec = bf(w.dvec);
    } else {
      a.trev = bf(w.trev);
      a.gsh_out = bf(w.g_sh);  // slot 0 carries the complete dE/dY
    }
    if (int rc = launch_fused_bwd_tail(p->chain_pair, a, stream)) return rc;
    // algorithmic traffic per edge: neighbor id, unit vector, w0, embedding, one pre-activation, three gradient rows, the
    // layer-1 dE/dY slot in; dE/dr_e (or the 8 basis sums + dE/dY) out.  Per atom: two x2s blocks, row pointer.
    const double per_edge = 1 + 4 + p->W + 64 + 64 + 2.0 * u + 64 + p->D + (fuse_edge ? 4 : 8 + p->D);
    const double fl = 2.0 * double(E) * (double(p->ng0) * 64 + 64.0 * 64 + 64.0 * c.embed_dim);
    if (int rc = mark("fused_bwd_tail", per_edge, 2.0 * p->D * u + 1, fl)) return rc;
    if (!fuse_edge) {
      EdgeBwdArgs eb{};
      eb.g = geom(g, nullptr);
      eb.g_emb0 = nullptr;
      eb.g_sh = buf(w.g_sh);
      eb.num_gsh = 1;
      eb.forces = forces;
      eb.t_in = buf(w.trev);
      eb.dvec = buf(w.dvec);
      eb.gather = gather ? 1 : 0;
      if (int rc = launch_edge_backward<T>(eb, stream)) return rc;
      if (int rc = mark("edge_backward", 8.0 / sizeof(T) + 4 + c.num_bessels + double(p->D) + (gather ? 4 : 6))) return rc;
    }
    if (gather) {
      ForceGatherArgs fg{N, g->rowptr, g->t_rowptr, g->t_perm, buf(w.dvec), forces};
      if (int rc = launch_force_gather<T>(fg, stream)) return rc;
      return mark("force_gather", 8.0 + 4.0 / sizeof(T), 3 + 8.0 / sizeof(T));
    }
    return AA_OK;
  }
#else
  int backward_fused_tail(const aa_graph*, void*) { return fail(AA_ERR_INVALID, "fused reverse tail: not part of this build"); }
#endif

  int forward_fused(const aa_graph* g, const void* pos, void* atom_energy) {
    const aa_model_config& c = p->cfg;
    const int S = c.num_scalar, u = c.num_tensor;
    FusedFwdArgs a{};
    a.N = N;
    a.atom0 = atom_begin(g);
    a.atom_end = atom_end(g);
    a.rowptr = g->rowptr;
    a.nbr = g->nbr;
    a.types = g->types;
    a.pos = static_cast<const float*>(pos);
    a.shift_vec = static_cast<const float*>(g->shift_vec);
    // segments of more than one 32-edge tile: teams of waves, dealt from class lists in the workspace
    a.tile_atoms = nullptr;
    a.tile_counts = nullptr;
    a.tile_cap = 0;
    if (g->max_degree > 32) {
      a.tile_counts = reinterpret_cast<int32_t*>(buf(w.tiles));
      a.tile_atoms = a.tile_counts + 8;
      a.tile_cap = g->num_atoms;
      // mixed form (one-tile pass over all atoms + team pass over the long ones) where long atoms are the minority -- boxes whose
      // average atom fits one tile; dense boxes that took the team path because they are small or their tiles are full run the pure
      // team form: there the one-tile launch would only skip.  2 / 4: the team / mixed form for every graph (A/B)
      const int64_t n_act = g->atom_end > g->atom_begin ? g->atom_end - g->atom_begin : g->num_atoms;
      const double fill = double(g->num_edges) / (32.0 * double(std::max<int64_t>(n_act, 1)));
      a.mixed = p->opt.fused_forward == 2 ? 0 : (p->opt.fused_forward == 4 || fill <= 1.15) ? 1 : 0;
    }
    a.num_types = c.num_types;
    a.embed_kind = c.embed_kind;
    a.spline_span = c.spline_span;
    a.poly_p = float(c.poly_p);
    auto wf = [&](size_t off) { return reinterpret_cast<const float*>(wt(off)); };
    auto bf = [&](size_t off) { return reinterpret_cast<float*>(buf(off)); };
    a.rmax_recip = wf(p->o_rmax);
    a.bessel_w = wf(p->o_bessel);
    a.emb_tab = wf(kFoldEmbed ? p->o_embtab_h : p->o_embtab);  // (folded: the table yields the first layer's pre-activation)
    // the weight program of the kernel (see fused_fwd_kernel): 12-KB blocks in execution order
    int ns = 0;
    FusedFwdArgs* prog = &a;  // (the program under construction: `a`, later the eight-wave form's copy)
    auto add_layer = [&](const float* Wq, int KC, int tile0, int ntiles) {
      for (int t = tile0; t < tile0 + ntiles; t += 2)
        for (int kc = 0; kc < KC; ++kc) {
          prog->wstep[ns][0] = Wq + (size_t(t) * KC + kc) * 64 * 24;
          prog->wstep[ns][1] = Wq + (size_t(t + 1) * KC + kc) * 64 * 24;
          ++ns;
        }
    };
    auto add_env = [&](const float* Wk) {
      for (int cblk = 0; cblk < 4; ++cblk) {
        prog->wstep[ns][0] = Wk + size_t(cblk) * 16 * p->R * 64;
        prog->wstep[ns][1] = Wk + size_t(cblk) * 16 * p->R * 64 + 1536;  // (blocks are loaded as 2 x 6 KB; 16 R 256 B are used)
        ++ns;
      }
    };
    const bool hold = p->fused_hold_w0;
    const bool folde = kFoldEmb1 && p->o_g0fq != 0;  // (see kFoldEmb1: no layer L1; first stage and env weights behind W1)
    if (!kFoldEmbed) add_layer(wf(p->embed.wq[0]), 2, 0, 2);
    if (!folde) add_layer(wf(p->embed.wq[1]), 2, 0, 2);
    auto add_proj = [&](int l, const float* Wk) {  // one env projection: R bf16x3 64x64 layers (kProjMfma) or 4 blocks of env-weight rows
      if (kProjMfma && p->o_wkq[l]) {
        for (int r = 0; r < p->R; ++r) add_layer(wf(p->o_wkq[l]) + size_t(r) * gemm_bf16x3_words(64, 64), 2, 0, 2);
      } else {
        add_env(Wk);
      }
    };
    add_proj(0, wf(folde ? p->o_wk0f : p->o_wk[0]));
    add_layer(wf(folde ? p->o_g0fq : p->o_g0q), 2, 0, 2 + 2 * p->R);
    add_layer(wf(p->latent[0].wq[0]), 4, 0, 2);
    add_proj(1, wf(p->o_wk[1]));
    const bool foldl = kFoldLatent && p->o_lat1in_fq != 0;  // (see kFoldLatent: no output layers L4 / L7, folded consumers)
    if (!foldl) add_layer(wf(p->latent[0].wq[1]), 2, 0, 2);
    if (!hold) add_layer(wf(folde ? p->o_g0fq : p->o_g0q), 2, 2, 2 * p->R);  // the w0 columns of the first-stage matrix again
    add_layer(wf(foldl ? p->o_lat1in_fq : p->latent[1].wq[0]), 6, 0, 2);
    if (!foldl) add_layer(wf(p->latent[1].wq[1]), 2, 0, 2);
    add_layer(wf(foldl ? p->o_ro0_fq : p->readout.wq[0]), 6, 0, 2);
    if (ns != fused_fwd_num_steps(p->R, hold) || ns > kFusedMaxSteps) return fail(AA_ERR_INVALID, "fused forward: program length mismatch");
    a.tpw0 = wf(p->o_tpw[0]);
    a.tpw1 = wf(p->o_tpw[1]);
    a.coupling = c.tps[0].coupling;
    a.sf = float(1.0 / std::sqrt(c.avg_num_neighbors));
    a.ro_w = wf(p->o_ro_last);
    a.ro_factor = float(1.0 / std::sqrt(2.0 * c.avg_num_neighbors));
    a.scales = c.has_scales ? wf(p->o_scales) : nullptr;
    a.shifts = c.has_shifts ? wf(p->o_shifts) : nullptr;
    a.vec = bf(w.vec);
    a.sh = bf(w.sh);
    a.se_h = bf(w.se_h[0]);
    a.emb = bf(w.emb);
    a.w0 = bf(w.w0);
    a.lat_h0 = bf(w.lat_h[0][0]);
    a.lat_h1 = bf(w.lat_h[1][0]);
    a.ro_h = bf(w.ro_h[0]);
    a.fcat = nullptr;
    a.x2s0 = bf(w.x2s[0]);
    a.x2s1 = bf(w.x2s[1]);
    a.atom_energy = static_cast<float*>(atom_energy);
    a.status = p->status;
    a.keep = p->opt.fused_keep_split == 0 ? kFusedKeepDefault : p->opt.fused_keep_split - 1;
    // the eight-wave form of the one-tile pass (aa_fused8.hip): its own step order -- Wenv0 | first stage | latent 0 | Wenv1 |
    // latent 1: scal1 chunks | [lat0, two-body] chunks x {latent 1, readout} | readout: lat1 chunks
    FusedFwdArgs a8{};
    const bool wide = p->fused_wide && folde && foldl && a.w0 != nullptr && (a.tile_atoms == nullptr || a.mixed);
    if (wide) {
      a8 = a;
      a8.wide_waves = p->opt.fused_narrow == 2 ? -8 : (p->opt.fused_narrow == 3 ? -4 : 4);  // (negative: forced, also on small boxes)
      a8.wide_one_per_cu = p->opt.fused_narrow == 6 ? 1 : 0;
      ns = 0;
      prog = &a8;
      auto add_step = [&](const float* Wq, int KC, int kc) {  // one step: the tile pair (0, 1) x chunk kc of a KC-chunk layer
        prog->wstep[ns][0] = Wq + (size_t(0) * KC + kc) * 64 * 24;
        prog->wstep[ns][1] = Wq + (size_t(1) * KC + kc) * 64 * 24;
        ++ns;
      };
      // env projections: 4 blocks of env-weight rows (vector form), or -- fused_narrow 5, A/B -- on the matrix cores (R bf16x3 64x64
      // layers, 2 steps each): the vector form is a quarter of the kernel's issue slots and 40 % of its LDS instructions while the
      // matrix pipe idles, and the matrix form still measured 11 % slower (3.50 vs 3.16 ms at C4, profiles/r06_v6_ab_c4_*)
      a8.wide_proj_mfma = (p->o_wkq[0] && p->o_wkq[1] && p->opt.fused_narrow == 5) ? 1 : 0;  // (A/B only: measured 11 % slower, HISTORY.md)
      auto add_proj8 = [&](int l, const float* Wk) {
        if (a8.wide_proj_mfma) {
          for (int r = 0; r < p->R; ++r) add_layer(wf(p->o_wkq[l]) + size_t(r) * gemm_bf16x3_words(64, 64), 2, 0, 2);
        } else {
          add_env(Wk);
        }
      };
      add_proj8(0, wf(p->o_wk0f));
      add_layer(wf(p->o_g0fq), 2, 0, 2 + 2 * p->R);
      add_layer(wf(p->latent[0].wq[0]), 4, 0, 2);
      add_proj8(1, wf(p->o_wk[1]));
      add_step(wf(p->o_lat1in_fq), 6, 4);
      add_step(wf(p->o_lat1in_fq), 6, 5);
      for (int kc : {2, 3, 0, 1}) {
        add_step(wf(p->o_lat1in_fq), 6, kc);
        add_step(wf(p->o_ro0_fq), 6, kc);
      }
      add_step(wf(p->o_ro0_fq), 6, 4);
      add_step(wf(p->o_ro0_fq), 6, 5);
      // the readout-reverse chain (Runner::backward "B3", folded form) in the forward's tail: eight-wave form, vector projections, no
      // team pass (its atoms would miss it), forces requested.  Taken from kFusedTailAtomsPerCu atoms per CU on (there the eight-wave
      // form is as fast as two four-wave workgroups and the chain's 0.75 ms of streaming disappear: C4 -0.45 ms same box,
      // profiles/r06_v19_*); fused_narrow 2 forces it, 3 / 7 keep it off.
      const bool tail_ok = want_forces && a.tile_atoms == nullptr && !a8.wide_proj_mfma && !a8.wide_one_per_cu && p->chain_gemm && kFoldLatent &&
                           p->o_b3af_q && p->o_b3c_q && c.num_layers == 2 && p->opt.fused_narrow != 7 && p->opt.fused_narrow != 3;
      if (tail_ok && (a8.wide_waves == -8 || (a8.wide_waves == 4 && a.atom_end - a.atom0 >= int64_t(kFusedTailAtomsPerCu) * fused_num_cus()))) {
        a8.wide_waves = -8;
        a8.tail = 1;
      }
      if (a8.tail) {
        add_layer(wf(p->o_b3af_q), 2, 0, 2);
        add_layer(wf(p->o_b3bf_q ? p->o_b3bf_q : p->o_b3b_q), 4, 0, 4);
        add_layer(wf(p->o_b3c_q), 2, 0, 2);
        a8.g_fcat = bf(w.g_fcat);
        a8.ld_gfcat = p->SL1;
        a8.g_scal1 = bf(w.g_scal[1]);
      }
      if (ns != fused_fwd8_num_steps(p->R, a8.wide_proj_mfma != 0, a8.tail != 0)) return fail(AA_ERR_INVALID, "fused forward (wide): program length mismatch");
    }
    if (int rc = mark("begin")) return rc;
    bool ran_wide = false;
    if (int rc = launch_fused_fwd(p->chain_pair, hold, a, stream, wide ? &a8 : nullptr, &ran_wide)) return rc;
    fwd_b3_done = ran_wide && a8.tail;
    if (wide && a8.tail && !ran_wide) return fail(AA_ERR_INVALID, "fused forward: the tail was planned for a launch that took the one-wave kernel");
    // algorithmic traffic: neighbor id + shift in; unit vector, harmonics, five 64-wide rows and w0 out per edge;
    // position, two x2s blocks, energy, row pointer per atom.  Flops: the linear layers of the forward (w0 counted once).
    const double per_edge = 1 + (g->shift_vec ? 3 : 0) + 3 + 4 + p->D + 5 * 64 + p->W;
    const double per_atom = 3 + 2.0 * p->D * u + 1 + 1;
    double fl = 2.0 * double(E) * (2.0 * 64 * 64 + 64.0 * p->ng0 + double(S + u) * 64 + 64.0 * S + double(2 * S + u) * 64 + 64.0 * S + 3.0 * S * 64);
    if (fwd_b3_done) {  // (+ the readout-reverse chain: 192 gradient columns out instead of two 64-wide pre-activation rows)
      fl += 2.0 * double(E) * (64.0 * 64 + 128.0 * 128 + 64.0 * 64);
      return mark("fused_fwd", per_edge + 64, per_atom, fl);
    }
    return mark("fused_fwd", per_edge, per_atom, fl);
  }

  int forward(const aa_graph* g, const void* pos, void* atom_energy) {
    const aa_model_config& c = p->cfg;
    const int S = c.num_scalar, u = c.num_tensor, L = c.num_layers, W = p->W, SL1 = p->SL1;
    if (use_fused_fwd(g)) return forward_fused(g, pos, atom_energy);
    // 1-2: geometry, SH, radial-chemical embedding
    if (int rc = mark("begin")) return rc;
    const double idx2 = 8.0 / sizeof(T);  // center + nbr ids, in elements
    if (int rc = launch_edge_prologue<T>(geom(g, pos), stream)) return rc;
    if (int rc = mark("edge_prologue", idx2 + 6 + (g->shift_vec ? 3 : 0) + 4 + p->D + c.embed_dim)) return rc;
    const SegList none{0, {}};
    const bool slot = use_slot();
    const bool fstaged = fold_staged();
    if (p->chain_gemm) {
      // 3 + 4 + 5a as ONE kernel: emb0 -> h_e -> emb -> [two_body | w0]; hidden layers stay in registers
      ChainArgs ca{};
      ca.nlayers = 3;
      SegList in{1, {seg(buf(w.emb0), c.embed_dim, c.embed_dim)}};
      SegList c0{1, {seg(buf(w.se_h[0]), 64, 64)}};
      SegList c1{1, {seg(buf(w.emb), S, S)}};
      SegList c2{2, {seg(buf(w.fcat), SL1, S), seg(buf(w.w0), W, W)}};
      ca.L[0] = chain_layer(E, in, 0, wt(p->embed.wq[0]), c.embed_dim, 64, c0, nullptr, nullptr, nullptr, 0, 0, 1);
      ca.L[1] = chain_layer(E, none, 0, wt(p->embed.wq[1]), 64, S, c1, nullptr, nullptr, nullptr, 1, 0, 0);
      ca.L[2] = chain_layer(E, none, 0, wt(p->o_g0q), 64, p->ng0, c2, nullptr, nullptr, nullptr, 1, -1, 0);
      if (fstaged) {
        // folded: a_e = silu(h_e) is stored where the embedding used to be; [two_body | w0] = a_e @ (W1 G0)
        ca.nlayers = 2;
        ca.L[0].kept_out = buf(w.emb);
        ca.L[0].ld_kept = S;
        ca.L[1] = chain_layer(E, none, 0, wt(p->o_g0fq), 64, p->ng0, c2, nullptr, nullptr, nullptr, 1, -1, 0);
      }
      if (int rc = run_chain(ca, "F1")) return rc;
    } else if (slot) {
      // 3: hidden layers of scalar_embed_mlp; 4 + 5a on the activated last hidden layer (output layer folded into G0)
      const int De = c.embed_mlp_depth, He = c.embed_mlp_width;
      SegList in{1, {seg(buf(w.emb0), c.embed_dim, c.embed_dim)}};
      SegList hid{1, {seg(buf(w.se_h[De - 1]), He, He)}};
      if (int rc = mlp_fwd(p->embed, De, in, w.se_h, hid, 0)) return rc;
      SegList out{2, {seg(buf(w.fcat), SL1, S), seg(buf(w.w0), W, W)}};
      act_now = c.act_kind[0];
      if (int rc = gemm(hid, 1, p->s_g0f, out)) return rc;
    } else {
    // 3: scalar_embed_mlp
    {
      SegList in{1, {seg(buf(w.emb0), c.embed_dim, c.embed_dim)}};
      SegList out{1, {seg(buf(w.emb), S, S)}};
      if (int rc = mlp_fwd(p->embed, c.embed_mlp_depth + 1, in, w.se_h, out, 0)) return rc;
    }
    // 4+5a: env_embed_linear and first_layer_env_embed_projection as ONE GEMM (tensorembed.py:89, _allegro.py:251)
    {
      SegList in{1, {seg(buf(w.emb), S, S)}};
      SegList out{3, {seg(buf(w.fcat), SL1, S), seg(buf(w.w0), W, W), seg(buf(w.envw[0]), W, W)}};
      if (p->env_mom) out.count = 2;
      act_now = AA_ACT_SILU;  // (a plain linear map: no activation involved)
      if (int rc = gemm(in, 0, wt(p->o_g0), wt(p->o_g0p), wt(p->o_g0q), S, p->ng0, out, nullptr, nullptr)) return rc;
    }
    }
    // 5: layers
    const double sfac = 1.0 / std::sqrt(c.avg_num_neighbors);
    for (int l = 0; l < L; ++l) {
      if (p->tp_op >= 0) {
        if (int rc = run_op_fwd(g, l)) return rc;
      } else if (p->env_mom) {
        TpMomArgs m = mom_args(g);
        if (l == 0) {
          m.c.scal1 = buf(w.scal[0]);  // the first-layer kernel writes its scalars through this field
          if (fstaged) m.wk0 = wt(p->o_wk0f);  // (its env input is a_e: env weights behind the output layer of scalar_embed_mlp)
          if (int rc = launch_tp_mom_fwd_first<T>(p->chain_pair, m, stream)) return rc;
          if (int rc = mark("tp_mom_fwd_first", p->D + m.ka0 + W + u, double(p->D) * u)) return rc;
        } else {
          if (int rc = launch_tp_mom_fwd_last<T>(p->chain_pair, m, stream)) return rc;
          if (int rc = mark("tp_mom_fwd_last", p->D + m.ka1 + W + u, 2.0 * p->D * u)) return rc;
        }
      } else if (p->chain_pair >= 0 && l == 1) {
        if (int rc = launch_tp_chain_fwd_last<T>(p->chain_pair, chain_args(g), stream)) return rc;
        if (int rc = mark("tp_chain_fwd_last", 2 * W + p->D + u, 2.0 * p->D * u)) return rc;
      } else if (p->use_spec) {
        TpSpecFwdArgs a{};
        a.E = E;
        a.N = N;
        a.rowptr = g->rowptr;
        a.u = u;
        a.sh = buf(w.sh);
        a.ld_sh = p->D;
        if (l == 0) {
          a.w_x1 = buf(w.w0);
          a.ld_w1 = W;
        } else {
          a.x1_dense = buf(w.tf[l - 1]);
        }
        a.w_env = buf(w.envw[l]);
        a.ld_we = W;
        a.weights = wt(p->o_tpw[l]);
        a.coupling = c.tps[l].coupling;
        a.sf = sfac;
        a.x2s = buf(w.x2s[l]);
        a.out = (l < L - 1 && p->chain_pair < 0) ? buf(w.tf[l]) : nullptr;
        a.scal = buf(w.scal[l]);
        a.ld_scal = u;
        if (int rc = launch_tp_spec_fwd<T>(p->spec_sig[l], a, stream)) return rc;
        if (int rc = mark("tp_spec_fwd", 2 * W + p->D + u, 2.0 * p->D * u)) return rc;
      } else {
      TpLayerFwdArgs a{};
      a.E = E;
      a.N = N;
      a.rowptr = g->rowptr;
      if (l == 0)
        a.x1 = implicit(w.w0);
      else
        a.x1.dense = buf(w.tf[l - 1]);
      a.x2 = implicit(w.envw[l]);
      a.weights = wt(p->o_tpw[l]);
      a.scatter_factor = sfac;
      a.x2s = buf(w.x2s[l]);
      a.out = l < L - 1 ? buf(w.tf[l]) : nullptr;
      a.scal = buf(w.scal[l]);
      a.ld_scal = u;
      if (int rc = launch_tp_layer_fwd<T>(p->layers[l], a, stream)) return rc;
      if (int rc = mark("tp_layer_fwd", 2 * W + p->D + u * p->D + u, 2.0 * p->D * u)) return rc;
      }
      if (p->chain_gemm) {
        ChainArgs ca{};
        SegList in{2, {seg(buf(w.fcat), SL1, S * (l + 1)), seg(buf(w.scal[l]), u, u)}};
        SegList ch{1, {seg(buf(w.lat_h[l][0]), 64, 64)}};
        SegList cl{1, {seg(buf(w.fcat) + S * (l + 1), SL1, S)}};
        ca.L[0] = chain_layer(E, in, 0, wt(p->latent[l].wq[0]), S * (l + 1) + u, 64, ch, nullptr, nullptr, nullptr, 0, 0, 1);
        if (fstaged && l < L - 1) {
          // folded: no output layer; a_l = silu(z_l) goes where lat_l used to be
          ca.nlayers = 1;
          ca.L[0].kept_out = buf(w.fcat) + S * (l + 1);
          ca.L[0].ld_kept = SL1;
        } else if (fstaged) {
          // folded: latent 1 and the readout on [two-body | a_0 | ...] with the output layers folded into their row blocks; a_1 stays in registers
          ca.nlayers = 2;
          SegList fin{1, {seg(buf(w.fcat), SL1, S * L)}};
          SegList cr{1, {seg(buf(w.ro_h[0]), 64, 64)}};
          ca.L[0] = chain_layer(E, in, 0, wt(p->o_lat1in_fq), S * (l + 1) + u, 64, ch, nullptr, nullptr, nullptr, 0, 0, 1);
          ca.L[1] = chain_layer(E, fin, 0, wt(p->o_ro0_fq), S * L + 64, 64, cr, nullptr, nullptr, nullptr, 1, -1, 0);
          ca.L[1].edge_sum_out = buf(w.e_edge);
          ca.ro_w = wt(p->o_ro_last);
        } else if (l < L - 1) {
          ca.nlayers = 2;
          ca.L[1] = chain_layer(E, none, 0, wt(p->latent[l].wq[1]), 64, S, cl, nullptr, nullptr, nullptr, 1, -1, 0);
        } else {
          // last latent + the readout's GEMM layer: lat_{L-1} stays in registers as the tail of the readout input
          ca.nlayers = 3;
          SegList fin{1, {seg(buf(w.fcat), SL1, S * L)}};
          SegList cr{1, {seg(buf(w.ro_h[0]), 64, 64)}};
          ca.L[1] = chain_layer(E, none, 0, wt(p->latent[l].wq[1]), 64, S, cl, nullptr, nullptr, nullptr, 1, 0, 0);
          ca.L[2] = chain_layer(E, fin, 0, wt(p->readout.wq[0]), S * L + 64, 64, cr, nullptr, nullptr, nullptr, 1, -1, 0);
          ca.L[2].edge_sum_out = buf(w.e_edge);  // last linear readout layer folded into the epilogue
          ca.ro_w = wt(p->o_ro_last);
        }
        if (int rc = run_chain(ca, "F2")) return rc;
        continue;
      }
      if (slot) {
        // hidden pre-activation z_l straight into slot l + 1; the earlier slots are activated on load (their output layers are
        // folded into this layer's row blocks)
        SegList in{2, {seg(buf(w.fcat), SL1, S * (l + 1)), seg(buf(w.scal[l]), u, u)}};
        SegList zl{1, {seg(buf(w.fcat) + S * (l + 1), SL1, S)}};
        act_now = c.act_kind[1];
        if (int rc = gemm(in, l > 0, p->s_in[l], zl, nullptr, nullptr, S, S * (l + 1))) return rc;
        continue;
      }
      SegList in{2, {seg(buf(w.fcat), SL1, S * (l + 1)), seg(buf(w.scal[l]), u, u)}};
      SegList out;
      out.count = (l < L - 1 && !p->env_mom) ? 2 : 1;
      out.s[0] = seg(buf(w.fcat) + S * (l + 1), SL1, S);
      if (l < L - 1 && !p->env_mom) out.s[1] = seg(buf(w.envw[l + 1]), W, W);
      if (int rc = mlp_fwd(p->latent[l], c.latent_mlp_depth + 1, in, w.lat_h[l], out, 1)) return rc;
    }
    // 6: edge readout GEMM layers, 7-8: last linear + edge sum + per-type scale/shift
    if (c.readout_mlp_depth > 0 && !p->chain_gemm) {
      act_now = c.act_kind[2];
      SegList a{1, {seg(buf(w.fcat), SL1, SL1)}};
      for (int i = 0; i < c.readout_mlp_depth; ++i) {
        SegList cs{1, {seg(buf(w.ro_h[i]), c.readout_mlp_width, c.readout_mlp_width)}};
        if (slot && i == 0) {
          act_now = c.act_kind[1];  // (the activated columns are the latents' hidden layers)
          if (int rc = gemm(a, 1, p->s_ro0, cs, nullptr, nullptr, S, SL1)) return rc;
          act_now = c.act_kind[2];
        } else if (int rc = gemm(a, i > 0, wt(p->readout.w[i]), wt(p->readout.wp[i]), wt(p->readout.wq[i]), p->readout.dims[i], p->readout.dims[i + 1], cs,
                                 nullptr, nullptr)) {
          return rc;
        }
        a = cs;
      }
    }
    ReadoutArgs ra = readout_args(g, atom_energy);
    ro_grad_done = want_forces && !p->chain_gemm && c.readout_mlp_depth > 0 && !p->opt.readout_two_pass;
    if (ro_grad_done) ra.g_h = buf(w.g_ro_h[c.readout_mlp_depth - 1]);
    if (int rc = launch_readout_reduce<T>(ra, stream)) return rc;
    return mark("readout_reduce", (p->chain_gemm ? 1 : (c.readout_mlp_depth > 0 ? c.readout_mlp_width : SL1)) * (ro_grad_done ? 2.0 : 1.0), 1);
  }

  // geometry reverse + force assembly (the end of every reverse pass)
  int edge_tail(const aa_graph* g, const void* pos, void* forces) {
    const aa_model_config& c = p->cfg;
    const int num_gsh = num_gsh_slots(p);
    EdgeBwdArgs eb{};
    eb.g = geom(g, pos);
    eb.g_emb0 = buf(w.g_emb0);
    eb.g_sh = buf(w.g_sh);
    eb.num_gsh = num_gsh;
    eb.forces = forces;
    if (p->embed_fused) eb.t_in = buf(w.trev);
    const bool gather = g->t_rowptr && g->t_perm;
    eb.dvec = buf(w.dvec);
    eb.gather = gather ? 1 : 0;
    if (int rc = launch_edge_backward<T>(eb, stream)) return rc;
    if (int rc = mark("edge_backward", 8.0 / sizeof(T) + 4 + (p->embed_fused ? c.num_bessels : c.embed_dim) + double(num_gsh) * p->D + (gather ? 4 : 6))) return rc;
    if (gather) {
      // deterministic force assembly: per atom, own segment minus transposed segment, fixed order (no atomics)
      ForceGatherArgs fg{N, g->rowptr, g->t_rowptr, g->t_perm, buf(w.dvec), forces};
      if (int rc = launch_force_gather<T>(fg, stream)) return rc;
      return mark("force_gather", 8.0 + 4.0 / sizeof(T), 3 + 8.0 / sizeof(T));
    }
    return AA_OK;
  }

  // Reverse pass of the slot form (aa_model_plan::slot_form): per dense-net slot, top down.  Every layer here has a 128-wide (S)
  // output -- the accumulator-resident kernel's shape -- and writes its slot once.
  int backward_slot(const aa_graph* g, const void* pos, void* forces) {
    const aa_model_config& c = p->cfg;
    const int S = c.num_scalar, u = c.num_tensor, L = c.num_layers, W = p->W, SL1 = p->SL1;
    const int De = c.embed_mlp_depth, He = c.embed_mlp_width, Dr = c.readout_mlp_depth, Hr = c.readout_mlp_width, H = c.latent_mlp_width;
    if (!(g->t_rowptr && g->t_perm)) AA_CHECK_HIP(hipMemsetAsync(forces, 0, size_t(N) * 3 * sizeof(T), stream));
    if (int rc = mark("memset", 0, 3)) return rc;
    // readout: d (last hidden) from the energy, then its hidden layers down to the first one
    {
      ReadoutArgs r = readout_args(g, nullptr);
      r.g_h = buf(w.g_ro_h[Dr - 1]);
      act_now = c.act_kind[2];
      if (!ro_grad_done) {
        if (int rc = launch_readout_backward<T>(r, stream)) return rc;
        if (int rc = mark("readout_backward", 2.0 * Hr)) return rc;
      }
      for (int i = Dr - 1; i >= 1; --i) {
        SegList a{1, {seg(buf(w.g_ro_h[i]), Hr, Hr)}};
        SegList cs{1, {seg(buf(w.g_ro_h[i - 1]), Hr, Hr)}};
        SegList z{1, {seg(buf(w.ro_h[i - 1]), Hr, Hr)}};
        if (int rc = gemm(a, 0, wt(p->readout.wt[i]), wt(p->readout.wtp[i]), wt(p->readout.wtq[i]), p->readout.dims[i + 1], p->readout.dims[i], cs, nullptr, &z))
          return rc;
      }
    }
    for (int l = L - 1; l >= 0; --l) {
      // d z_l = ([d readout hidden | d z_{l+1} .. d z_{L-1}] @ stack_l + d a_l of the next layer's moments) * act'(z_l)
      SegList a{l < L - 1 ? 2 : 1, {seg(buf(w.g_ro_h[0]), Hr, Hr), seg(buf(w.g_fcat) + S * (l + 2), SL1, S * (L - 1 - l))}};
      SegList dz{1, {seg(buf(w.g_fcat) + S * (l + 1), SL1, S)}};
      SegList z{1, {seg(buf(w.fcat) + S * (l + 1), SL1, S)}};
      SegList ad{1, {seg(buf(w.g_aenv), H, H)}};
      act_now = c.act_kind[1];
      if (int rc = gemm(a, 0, p->s_rs[l], dz, &z, l < L - 1 ? &ad : nullptr)) return rc;
      SegList gs{1, {seg(buf(w.g_scal[l]), u, u)}};
      if (int rc = gemm(dz, 0, p->s_sct[l], gs)) return rc;
      // tensor-product layer reverse (per-atom operator kernels)
      if (int rc = run_op_bwd(g, l)) return rc;
    }
    // slot 0 (two-body scalars): every consumer's share in one layer
    {
      SegList a{2, {seg(buf(w.g_ro_h[0]), Hr, Hr), seg(buf(w.g_fcat) + S, SL1, S * L)}};
      SegList tb{1, {seg(buf(w.g_fcat), SL1, S)}};
      if (int rc = gemm(a, 0, p->s_rstb, tb)) return rc;
    }
    // first stage + output layer of scalar_embed_mlp: d h = ([d two-body | d w0] @ (W_last G0)^T + d a_e of the moments) * act'(h)
    {
      SegList a{2, {seg(buf(w.g_fcat), SL1, S), seg(buf(w.g_w0), W, W)}};
      SegList dh{1, {seg(buf(w.g_se_h[De - 1]), He, He)}};
      SegList z{1, {seg(buf(w.se_h[De - 1]), He, He)}};
      SegList ad{1, {seg(buf(w.g_aenv), He, He)}};
      act_now = c.act_kind[0];
      if (int rc = gemm(a, 0, p->s_g0ft, dh, &z, &ad)) return rc;
      SegList gi{1, {seg(buf(w.g_emb0), c.embed_dim, c.embed_dim)}};
      if (int rc = mlp_bwd(p->embed, De, dh, w.se_h, w.g_se_h, gi, nullptr, 0)) return rc;
    }
    return edge_tail(g, pos, forces);
  }

  int backward(const aa_graph* g, const void* pos, void* forces) {
    if (use_slot()) return backward_slot(g, pos, forces);
    const aa_model_config& c = p->cfg;
    const int S = c.num_scalar, u = c.num_tensor, L = c.num_layers, W = p->W, SL1 = p->SL1;
    // spec path with u <= 64 writes every g_sh slot with plain stores; otherwise slots are accumulated into
    const bool gsh_stores = (p->use_spec && u <= 64) || p->tp_op >= 0;
    const int num_gsh = num_gsh_slots(p);
    if (!gsh_stores) AA_CHECK_HIP(hipMemsetAsync(buf(w.g_sh), 0, size_t(E) * p->D * num_gsh * sizeof(T), stream));
    if (!(g->t_rowptr && g->t_perm)) AA_CHECK_HIP(hipMemsetAsync(forces, 0, size_t(N) * 3 * sizeof(T), stream));
    if (int rc = mark("memset", gsh_stores ? 0 : p->D * num_gsh, 3)) return rc;
    const SegList none{0, {}};
    if (p->chain_gemm && fwd_b3_done) {
      // (the fused forward of this step ran this chain in its tail: d EDGE_FEATURES[:, :S L] and d scal_{L-1} are in the workspace)
    } else if (p->chain_gemm) {
      // readout reverse + last latent reverse in ONE kernel; d_lat_{L-1} and d_h never leave registers
      ReadoutArgs r = readout_args(g, nullptr);
      ChainArgs ca{};
      ca.nlayers = 3;
      ca.ro_w = r.w;
      ca.ro_factor = r.factor;
      ca.ro_scales = r.scales;
      ca.types = g->types;
      ca.center = g->center;
      SegList in{1, {seg(buf(w.ro_h[0]), 64, 64)}};
      // d_fcat[:, :S*L] receives the readout-reverse AND the latent-reverse contribution: instead of storing the
      // first and accumulating the second (write + read + write), the last layer takes both operands -- the
      // transformed readout hidden rows again (L2-resident) and the chained d_hidden -- against the stacked
      // weights [readout' ; latent'] and stores each tile once.
      SegList cn{1, {seg(nullptr, 64, 64)}};
      SegList z1{1, {seg(buf(w.lat_h[L - 1][0]), 64, 64)}};
      SegList c2{1, {seg(buf(w.g_fcat), SL1, S * L)}};
      SegList c3{1, {seg(buf(w.g_scal[L - 1]), u, u)}};
      ca.nlayers = 4;
      ca.L[0] = chain_layer(E, in, 0, wt(p->o_b3a_q), 64, S, cn, nullptr, nullptr, nullptr, 0, 0, 0);
      ca.L[0].a_mode = 1;
      ca.L[1] = chain_layer(E, none, 0, wt(p->latent[L - 1].wtq[1]), S, 64, cn, nullptr, &z1, nullptr, 1, 0, 0);
      ca.L[2] = chain_layer(E, in, 0, wt(p->o_b3b_q), 128, S * L, c2, nullptr, nullptr, nullptr, 1, -1, 0);
      ca.L[2].a_mode = 1;
      ca.L[3] = chain_layer(E, none, 0, wt(p->o_b3c_q), 64, u, c3, nullptr, nullptr, nullptr, 1, -1, 0);
      if (kFoldLatent && p->o_b3af_q && L == 2) {
        // "d lat1 = d ro_h @ Wro[lat1]^T" and "d a1 = d lat1 @ Wout_1^T" as ONE 64x64 layer (folded at pack time): 12 instead of 14 steps
        ca.nlayers = 3;
        ca.L[0] = chain_layer(E, in, 0, wt(p->o_b3af_q), 64, 64, cn, nullptr, &z1, nullptr, 0, 0, 0);
        ca.L[0].a_mode = 1;
        ca.L[1] = chain_layer(E, in, 0, wt(p->o_b3bf_q ? p->o_b3bf_q : p->o_b3b_q), 128, S * L, c2, nullptr, nullptr, nullptr, 1, -1, 0);
        ca.L[1].a_mode = 1;
        ca.L[2] = chain_layer(E, none, 0, wt(p->o_b3c_q), 64, u, c3, nullptr, nullptr, nullptr, 1, -1, 0);
      }
      if (int rc = run_chain(ca, "B3")) return rc;
    } else
    // readout
    {
      ReadoutArgs r = readout_args(g, nullptr);
      if (c.readout_mlp_depth > 0) {
        r.g_h = buf(w.g_ro_h[c.readout_mlp_depth - 1]);
        act_now = c.act_kind[2];
        if (!ro_grad_done) {
          if (int rc = launch_readout_backward<T>(r, stream)) return rc;
          if (int rc = mark("readout_backward", 2.0 * (c.readout_mlp_depth > 0 ? c.readout_mlp_width : SL1))) return rc;
        }
        SegList a{1, {seg(r.g_h, c.readout_mlp_width, c.readout_mlp_width)}};
        for (int i = c.readout_mlp_depth - 1; i >= 0; --i) {
          SegList cs, z;
          const SegList* zp = nullptr;
          if (i > 0) {
            cs = SegList{1, {seg(buf(w.g_ro_h[i - 1]), c.readout_mlp_width, c.readout_mlp_width)}};
            z = SegList{1, {seg(buf(w.ro_h[i - 1]), c.readout_mlp_width, c.readout_mlp_width)}};
            zp = &z;
          } else {
            cs = SegList{1, {seg(buf(w.g_fcat), SL1, SL1)}};
          }
          if (int rc = gemm(a, 0, wt(p->readout.wt[i]), wt(p->readout.wtp[i]), wt(p->readout.wtq[i]), p->readout.dims[i + 1], p->readout.dims[i], cs,
                            nullptr, zp))
            return rc;
          a = cs;
        }
      } else {
        r.g_h = buf(w.g_fcat);
        if (int rc = launch_readout_backward<T>(r, stream)) return rc;
        if (int rc = mark("readout_backward", 2.0 * (c.readout_mlp_depth > 0 ? c.readout_mlp_width : SL1))) return rc;
      }
    }
    const double sfac = 1.0 / std::sqrt(c.avg_num_neighbors);
    for (int l = L - 1; l >= 0; --l) {
      // latent MLP reverse
      if (p->chain_gemm && l == L - 1) {
        // (already done inside the readout chain above)
      } else if (p->chain_gemm) {
        ChainArgs ca{};
        ca.nlayers = 2;
        SegList in{1, {seg(buf(w.g_fcat) + S * (l + 1), SL1, S)}};
        SegList cn{1, {seg(nullptr, 64, 64)}};
        SegList zz{1, {seg(buf(w.lat_h[l][0]), 64, 64)}};
        SegList ad{1, {seg(buf(w.g_aenv), 64, 64)}};
        SegList c1{2, {seg(buf(w.g_fcat), SL1, S * (l + 1)), seg(buf(w.g_scal[l]), u, u)}};
        int acc1[3] = {1, 0, 0};
        ca.L[0] = chain_layer(E, in, 0, wt(p->latent[l].wtq[1]), S, 64, cn, nullptr, &zz, &ad, 0, 0, 0);
        ca.L[1] = chain_layer(E, none, 0, wt(p->latent[l].wtq[0]), 64, S * (l + 1) + u, c1, acc1, nullptr, nullptr, 1, -1, 0);
        if (kFoldLat0Rev && p->o_b3bf_q && L == 2 && l == 0) {
          // the readout-reverse chain already applied Wout_0^T (folded into its lat0 columns): what is left of the output layer's
          // reverse is elementwise -- d h = (d a_0 + d a_0 of the moments) x silu'(h) -- and rides as the operand transform of the
          // first-layer reverse: ONE layer, 4 steps instead of 6
          ca.nlayers = 1;
          ca.L[0] = chain_layer(E, in, 0, wt(p->latent[l].wtq[0]), 64, S * (l + 1) + u, c1, acc1, nullptr, nullptr, 0, -1, 0);
          ca.L[0].a_mode = 2;
          ca.L[0].a2_add = buf(w.g_aenv);
          ca.L[0].ld_a2add = 64;
          ca.L[0].a2_z = buf(w.lat_h[l][0]);
          ca.L[0].ld_a2z = 64;
          if (sizeof(T) == 4 && S == 64 && u == 64 && !p->opt.chain_staged_weights) {
            // the same layer with its 48 KB of weights resident in LDS (aa_chain_res.hip): one persistent workgroup per CU, no
            // per-workgroup staging, no barriers in the row loop
            ChainB2Args b2{};
            b2.M = E;
            b2.a = reinterpret_cast<const float*>(buf(w.g_fcat)) + S * (l + 1);
            b2.lda = SL1;
            b2.add = reinterpret_cast<const float*>(buf(w.g_aenv));
            b2.ldadd = 64;
            b2.z = reinterpret_cast<const float*>(buf(w.lat_h[l][0]));
            b2.ldz = 64;
            b2.Wq = wt(p->latent[l].wtq[0]);
            b2.c0 = reinterpret_cast<float*>(buf(w.g_fcat));
            b2.ldc0 = SL1;
            b2.c1 = reinterpret_cast<float*>(buf(w.g_scal[l]));
            b2.ldc1 = u;
            if (int rc = launch_chain_b2_resident(b2, stream)) return rc;
            if (int rc = mark("gc_64x128", gemm_row_elems(ca.L[0].g, true) + 2.0 * 64, 0, 2.0 * double(E) * 64 * 128)) return rc;
            goto b2_done;
          }
        }
        if (int rc = run_chain(ca, "B2")) return rc;
      b2_done:;
      } else {
      SegList go;
      go.count = (l < L - 1 && !p->env_mom) ? 2 : 1;
      go.s[0] = seg(buf(w.g_fcat) + S * (l + 1), SL1, S);
      if (l < L - 1 && !p->env_mom) go.s[1] = seg(buf(w.g_envw), W, W);
      SegList aenv{1, {seg(p->env_mom ? buf(w.g_aenv) : nullptr, c.latent_mlp_width, c.latent_mlp_width)}};
      const SegList* addp = (p->env_mom && l < L - 1) ? &aenv : nullptr;
      SegList gi{2, {seg(buf(w.g_fcat), SL1, S * (l + 1)), seg(buf(w.g_scal[l]), u, u)}};
      int acc[3] = {1, 0, 0};
      if (int rc = mlp_bwd(p->latent[l], c.latent_mlp_depth + 1, go, w.lat_h[l], w.g_lat_h, gi, acc, 1, addp)) return rc;
      }
      // tensor-product layer reverse
      if (p->tp_op >= 0) {
        if (int rc = run_op_bwd(g, l)) return rc;
        continue;
      }
      if (p->env_mom) {
        TpMomArgs m = mom_args(g);
        m.c.gscal0 = buf(w.g_scal[0]);
        m.c.gscal1 = buf(w.g_scal[1]);
        m.c.g_w0 = buf(w.g_w0);
        m.c.gsh_x1 = buf(w.g_sh);
        m.c.gsh_env = buf(w.g_sh) + size_t(l + 1) * size_t(E) * p->D;
        m.g_a = buf(w.g_aenv);
        if (l == 1) {
          m.ld_ga = c.latent_mlp_width;
          if (int rc = launch_tp_mom_bwd_last<T>(p->chain_pair, m, stream)) return rc;
          if (int rc = mark("tp_mom_bwd_last", p->D + W + u + 2 * m.ka1 + p->D, double(p->D) * u)) return rc;
        } else if (use_fused_tail(g)) {
          // (the fused reverse tail below takes it from here: layer-0 tensor product reverse + first-stage / embed-MLP reverse + edge reverse)
        } else {
          m.ld_ga = S;
          // (after a fused forward the embedding's slot holds a_e = silu(h) of scalar_embed_mlp and the env weights are folded behind
          //  its output layer, kFoldEmb1: d a_e comes out instead of d emb)
          if (kFoldEmb1 && p->o_wt0f && folded_fwd(g)) m.wt0 = wt(p->o_wt0f);
          if (int rc = launch_tp_mom_bwd_first<T>(p->chain_pair, m, stream)) return rc;
          if (int rc = mark("tp_mom_bwd_first", p->D + 2 * W + 2 * u + 2 * m.ka0 + 2 * p->D, 2.0 * p->D * u)) return rc;
        }
        continue;
      }
      if (p->chain_pair >= 0) {
        TpChainArgs a = chain_args(g);
        a.gscal0 = buf(w.g_scal[0]);
        a.gscal1 = buf(w.g_scal[1]);
        a.g_w0 = buf(w.g_w0);
        a.g_wenv = buf(w.g_envw);
        a.gsh_x1 = buf(w.g_sh);
        a.gsh_env = buf(w.g_sh) + size_t(l + 1) * size_t(E) * p->D;
        if (l == 1) {
          if (int rc = launch_tp_chain_bwd_last<T>(p->chain_pair, a, stream)) return rc;
          if (int rc = mark("tp_chain_bwd_last", 3 * W + 2 * p->D + u, 2.0 * p->D * u)) return rc;
        } else {
          if (int rc = launch_tp_chain_bwd_first<T>(p->chain_pair, a, stream)) return rc;
          if (int rc = mark("tp_chain_bwd_first", 4 * W + 3 * p->D + 2 * u, 2.0 * p->D * u)) return rc;
        }
        continue;
      }
      if (p->use_spec) {
        TpSpecBwdArgs a{};
        a.E = E;
        a.N = N;
        a.rowptr = g->rowptr;
        a.u = u;
        a.sh = buf(w.sh);
        a.ld_sh = p->D;
        if (l == 0) {
          a.w_x1 = buf(w.w0);
          a.ld_w1 = W;
          a.g_w1 = buf(w.g_w0);
          a.ld_gw1 = W;
        } else {
          a.x1_dense = buf(w.tf[l - 1]);
          a.g_x1_dense = buf(w.g_tf[(l - 1) & 1]);
        }
        a.w_env = buf(w.envw[l]);
        a.ld_we = W;
        a.weights = wt(p->o_tpw[l]);
        a.coupling = c.tps[l].coupling;
        a.sf = sfac;
        a.x2s = buf(w.x2s[l]);
        a.gout = l < L - 1 ? buf(w.g_tf[l & 1]) : nullptr;
        a.gscal = buf(w.g_scal[l]);
        a.ld_gscal = u;
        a.g_wenv = buf(w.g_envw);
        a.ld_gwe = W;
        a.gsh_x1 = buf(w.g_sh);
        a.gsh_env = buf(w.g_sh) + size_t(l + 1) * size_t(E) * p->D;
        a.ld_gsh = p->D;
        if (int rc = launch_tp_spec_bwd<T>(p->spec_sig[l], a, stream)) return rc;
        if (int rc = mark("tp_spec_bwd", 4 * W + 3 * p->D + 2 * u * p->D + u, 2.0 * p->D * u)) return rc;
        continue;
      }
      TpLayerBwdArgs a{};
      a.E = E;
      a.N = N;
      a.rowptr = g->rowptr;
      if (l == 0)
        a.x1 = implicit(w.w0);
      else
        a.x1.dense = buf(w.tf[l - 1]);
      a.x2 = implicit(w.envw[l]);
      a.weights = wt(p->o_tpw[l]);
      a.scatter_factor = sfac;
      a.x2s = buf(w.x2s[l]);
      a.gout = l < L - 1 ? buf(w.g_tf[l & 1]) : nullptr;
      a.gscal = buf(w.g_scal[l]);
      a.ld_gscal = u;
      if (l == 0) {
        a.g1.gw = buf(w.g_w0);
        a.g1.ldgw = W;
        a.g1.gsh = buf(w.g_sh);
        a.g1.ld_gsh = p->D;
      } else {
        a.g1.dense = buf(w.g_tf[(l - 1) & 1]);
      }
      a.g2.gw = buf(w.g_envw);
      a.g2.ldgw = W;
      a.g2.gsh = buf(w.g_sh);
      a.g2.ld_gsh = p->D;
      if (int rc = launch_tp_layer_bwd<T>(p->layers[l], a, stream)) return rc;
      if (int rc = mark("tp_layer_bwd", 4 * W + 2 * p->D + 2 * u * p->D + u, 2.0 * p->D * u)) return rc;
    }
    if (use_fused_tail(g)) return backward_fused_tail(g, forces);
    if (p->chain_gemm) {
      // first-stage reverse + scalar_embed_mlp reverse in ONE kernel
      ChainArgs ca{};
      ca.nlayers = 3;
      SegList in{2, {seg(buf(w.g_fcat), SL1, S), seg(buf(w.g_w0), W, W)}};
      SegList cn{1, {seg(nullptr, 64, 64)}};
      SegList ad{1, {seg(buf(w.g_aenv), S, S)}};
      SegList zz{1, {seg(buf(w.se_h[0]), 64, 64)}};
      SegList ce{1, {seg(buf(w.g_emb0), c.embed_dim, c.embed_dim)}};
      ca.L[0] = chain_layer(E, in, 0, wt(p->o_g0tq), p->ng0, S, cn, nullptr, nullptr, &ad, 0, 0, 0);
      ca.L[1] = chain_layer(E, none, 0, wt(p->embed.wtq[1]), S, 64, cn, nullptr, &zz, nullptr, 1, 0, 0);
      ca.L[2] = chain_layer(E, none, 0, wt(p->embed.wtq[0]), 64, c.embed_dim, ce, nullptr, nullptr, nullptr, 1, -1, 0);
      if (kFoldEmb1 && p->o_g0tfq && folded_fwd(g)) {
        // everything in front of the hidden layer of scalar_embed_mlp folded (kFoldEmb1; the forward stored a_e, not the embedding):
        // d h = ((d[two-body | w0] @ (W1 G0)^T) + d a_e of the moments) x silu'(h) -- ONE 256 -> 64 layer, 8 steps -- ...
        ca.nlayers = 1;
        ca.L[0] = chain_layer(E, in, 0, wt(p->o_g0tfq), p->ng0, 64, cn, nullptr, &zz, &ad, 0, 0, 0);
        if (p->embed_fused && p->o_embtab_h) {
          // ... contracted against the folded two-body table in its epilogue
          ca.L[0].embrev_out = buf(w.trev);
          ca.emb_table = wt(p->o_embtab_h);
          ca.num_types = c.num_types;
          ca.types = g->types;
          ca.center = g->center;
          ca.nbr = g->nbr;
        } else {
          // ... (three species: no table in the chain's LDS) followed by W0^T -> d emb0 for the edge reverse
          ca.nlayers = 2;
          ca.L[1] = chain_layer(E, none, 0, wt(p->embed.wtq[0]), 64, c.embed_dim, ce, nullptr, nullptr, nullptr, 1, -1, 0);
        }
      } else if (p->embed_fused && p->o_embtab_h) {
        // folded table (kFoldEmbed): d_h, the output of the second layer, is contracted straight back to the 8 basis functions
        // with T = tab @ W0 -- the layer W0^T and d emb0 do not exist
        ca.nlayers = 2;
        ca.L[1].embrev_out = buf(w.trev);
        ca.emb_table = wt(p->o_embtab_h);
        ca.num_types = c.num_types;
        ca.types = g->types;
        ca.center = g->center;
        ca.nbr = g->nbr;
      } else if (p->embed_fused) {
        // d emb0 is contracted straight back to the 8 basis functions in the epilogue and never stored
        ca.L[2].g.c = SegList{1, {seg(nullptr, c.embed_dim, c.embed_dim)}};
        ca.L[2].embrev_out = buf(w.trev);
        ca.emb_table = wt(p->o_embtab);
        ca.num_types = c.num_types;
        ca.types = g->types;
        ca.center = g->center;
        ca.nbr = g->nbr;
      }
      if (int rc = run_chain(ca, "B1")) return rc;
    } else {
    // fused first stage reverse
    {
      SegList go{3, {seg(buf(w.g_fcat), SL1, S), seg(buf(w.g_w0), W, W), seg(buf(w.g_envw), W, W)}};
      if (p->env_mom) go.count = 2;
      SegList gi{1, {seg(buf(w.g_emb), S, S)}};
      SegList aenv{1, {seg(p->env_mom ? buf(w.g_aenv) : nullptr, S, S)}};
      act_now = AA_ACT_SILU;
      if (int rc = gemm(go, 0, wt(p->o_g0t), wt(p->o_g0tp), wt(p->o_g0tq), p->ng0, S, gi, nullptr, nullptr,
                        p->env_mom ? &aenv : nullptr))
        return rc;
    }
    // scalar_embed_mlp reverse
    {
      SegList go{1, {seg(buf(w.g_emb), S, S)}};
      SegList gi{1, {seg(buf(w.g_emb0), c.embed_dim, c.embed_dim)}};
      if (int rc = mlp_bwd(p->embed, c.embed_mlp_depth + 1, go, w.se_h, w.g_se_h, gi, nullptr, 0)) return rc;
    }
    }
    return edge_tail(g, pos, forces);
  }
};

template <typename T>
int run_model(const aa_model_plan* p, const void* dev_weights, const aa_graph* g, const void* pos, void* workspace,
              size_t ws_bytes, void* atom_energy, void* forces, hipStream_t stream, StageProfile* prof = nullptr) {
  Runner<T> r;
  r.prof = prof;
  r.p = p;
  r.wts = static_cast<const T*>(dev_weights);
  r.ws = static_cast<char*>(workspace);
  r.E = g->num_edges;
  r.N = g->num_atoms;
  r.stream = stream;
  r.w = layout_workspace(p, r.N, r.E, forces != nullptr);
  if (r.w.total > ws_bytes) return fail(AA_ERR_WORKSPACE, "aa_model_energy_forces: workspace too small");
  if (p->opt.poison_workspace) AA_CHECK_HIP(hipMemsetAsync(workspace, 0xFF, r.w.total, stream));  // debugging: NaN everywhere
  r.want_forces = forces != nullptr;
  if (p->ev_wait) AA_CHECK_HIP(hipStreamWaitEvent(stream, (hipEvent_t)(uintptr_t)(p->ev_wait), 0));
  if (int rc = r.forward(g, pos, atom_energy)) return rc;
  if (p->ev_record) AA_CHECK_HIP(hipEventRecord((hipEvent_t)(uintptr_t)(p->ev_record), stream));
  if (forces)
    if (int rc = r.backward(g, pos, forces)) return rc;
  // the atom-block hint is the caller's promise (per-atom kernels skip the rest): two row pointers verify it on the device, as the
  // LAST launch of the step, so that a broken promise also turns this step's energies and forces into NaN
  if (g->atom_end > g->atom_begin && (g->atom_begin > 0 || g->atom_end < g->num_atoms) && p->status)
    if (int rc = launch_graph_hint_check(g->rowptr, g->num_atoms, g->atom_begin, g->atom_end, p->status, atom_energy, forces, int(sizeof(T)), stream))
      return rc;
  return AA_OK;
}
}  // namespace

extern "C" int aa_model_energy_forces(const aa_model_plan* plan, const void* dev_weights, const aa_graph* graph,
                                      const void* pos, void* workspace, size_t workspace_bytes, void* atom_energy,
                                      void* forces, aa_stream stream) {
  AA_REQUIRE(plan && dev_weights && graph && pos && atom_energy, "aa_model_energy_forces: null argument");
  AA_REQUIRE(graph->num_atoms >= 0 && graph->num_edges >= 0 && graph->num_edges < (int64_t(1) << 31),
             "aa_model_energy_forces: graph too large for int32 edge ids");
  AA_REQUIRE(graph->num_edges == 0 || (graph->center && graph->nbr), "aa_model_energy_forces: null edge arrays");
  AA_REQUIRE(graph->rowptr && graph->types, "aa_model_energy_forces: null rowptr/types");
  AA_REQUIRE(graph->atom_begin >= 0 && graph->atom_end <= graph->num_atoms &&
                 (graph->atom_end >= graph->atom_begin || graph->atom_end == 0),
             "aa_model_energy_forces: atom_begin/atom_end out of range");
  AA_REQUIRE(workspace || workspace_bytes == 0, "aa_model_energy_forces: null workspace");
  if (int rc = consume_status(plan, "aa_model_energy_forces")) return rc;  // (an EARLIER step contradicted the graph hints)
  hipStream_t s = static_cast<hipStream_t>(stream);
  auto run = [&](hipStream_t st) {
    if (plan->cfg.dtype == AA_F32)
      return run_model<float>(plan, dev_weights, graph, pos, workspace, workspace_bytes, atom_energy, forces, st);
    return run_model<double>(plan, dev_weights, graph, pos, workspace, workspace_bytes, atom_energy, forces, st);
  };
  aa_model_plan::StepGraph& sg = plan->sg;
  if (!sg.enabled) return run(s);
  const aa_model_plan::StepGraph::Key key{dev_weights,    pos,           workspace,     atom_energy,      forces,          graph->center,
                                          graph->nbr,     graph->rowptr, graph->types,  graph->shift_vec, graph->t_rowptr, graph->t_perm,
                                          graph->num_atoms, graph->num_edges, graph->atom_begin, graph->atom_end,
                                          workspace_bytes, graph->max_degree, plan->taps ? 1 : 0};
  if (!sg.exec || !(sg.key == key)) {
    if (sg.exec) (void)hipGraphExecDestroy(sg.exec);
    if (sg.graph) (void)hipGraphDestroy(sg.graph);
    sg.exec = nullptr;
    sg.graph = nullptr;
    AA_CHECK_HIP(hipStreamBeginCapture(sg.cap_stream, hipStreamCaptureModeThreadLocal));
    const int rc = run(sg.cap_stream);
    const hipError_t ec = hipStreamEndCapture(sg.cap_stream, &sg.graph);
    if (rc != AA_OK) return rc;
    if (ec != hipSuccess) return fail(AA_ERR_HIP, "aa_model_energy_forces: graph capture failed");
    AA_CHECK_HIP(hipGraphInstantiate(&sg.exec, sg.graph, nullptr, nullptr, 0));
    sg.key = key;
  }
  AA_CHECK_HIP(hipGraphLaunch(sg.exec, s));
  return AA_OK;
}

extern "C" int aa_model_energy_forces_profiled(const aa_model_plan* plan, const void* dev_weights, const aa_graph* graph,
                                               const void* pos, void* workspace, size_t workspace_bytes,
                                               void* atom_energy, void* forces, aa_stream stream, int max_stages,
                                               float* stage_ms, char* stage_names /* [max_stages][32] */,
                                               int* num_stages, double* stage_bytes, double* stage_flops) {
  AA_REQUIRE(plan && dev_weights && graph && pos && atom_energy && stage_ms && stage_names && num_stages,
             "aa_model_energy_forces_profiled: null argument");
  hipStream_t s = static_cast<hipStream_t>(stream);
  StageProfile prof;
  int rc = plan->cfg.dtype == AA_F32
               ? run_model<float>(plan, dev_weights, graph, pos, workspace, workspace_bytes, atom_energy, forces, s, &prof)
               : run_model<double>(plan, dev_weights, graph, pos, workspace, workspace_bytes, atom_energy, forces, s, &prof);
  if (rc == AA_OK) {
    hipError_t e = hipStreamSynchronize(s);
    if (e != hipSuccess) rc = fail(AA_ERR_HIP, "profile: stream sync failed");
  }
  int n = 0;
  if (rc == AA_OK) {
    for (size_t i = 1; i < prof.events.size() && n < max_stages; ++i, ++n) {
      float ms = 0.f;
      (void)hipEventElapsedTime(&ms, prof.events[i - 1], prof.events[i]);
      stage_ms[n] = ms;
      snprintf(stage_names + 32 * n, 32, "%s", prof.names[i].c_str());
      if (stage_bytes) stage_bytes[n] = prof.bytes[i];
      if (stage_flops) stage_flops[n] = prof.flops[i];
    }
  }
  *num_stages = n;
  prof.clear();
  return rc;
}

extern "C" int aa_model_virial(const aa_model_plan* plan, const aa_graph* graph, void* workspace, size_t workspace_bytes,
                               void* virial9, aa_stream stream) {
  AA_REQUIRE(plan && graph && workspace && virial9, "aa_model_virial: null argument");
  const Workspace w = layout_workspace(plan, graph->num_atoms, graph->num_edges, 1);
  if (w.total > workspace_bytes) return fail(AA_ERR_WORKSPACE, "aa_model_virial: workspace too small (was it sized with forces?)");
  char* base = static_cast<char*>(workspace);
  VirialArgs a{graph->num_edges, base + w.dvec, base + w.vec, reinterpret_cast<double*>(base + w.vir_part), virial9};
  hipStream_t s = static_cast<hipStream_t>(stream);
  return plan->cfg.dtype == AA_F32 ? launch_virial<float>(a, s) : launch_virial<double>(a, s);
}

extern "C" int aa_model_debug_tap(const aa_model_plan* plan, const char* name, int64_t N, int64_t E, const void* workspace,
                                  const void** ptr, int64_t* ld) {
  AA_REQUIRE(plan && name && workspace && ptr && ld, "aa_model_debug_tap: null argument");
  std::string n(name);
  int with_forces = 0;
  if (n.size() > 2 && n.compare(n.size() - 2, 2, "+f") == 0) {  // layout of a step that computed forces
    with_forces = 1;
    n.resize(n.size() - 2);
  }
  Workspace w = layout_workspace(plan, N, E, with_forces);
  const char* base = static_cast<const char*>(workspace);
  if (n == "edge_attrs") {
    *ptr = base + w.sh;
    *ld = plan->D;
  } else if (n == "edge_embedding") {
    *ptr = base + w.emb;
    *ld = plan->cfg.num_scalar;
  } else if (n == "edge_features") {
    *ptr = base + w.fcat;
    *ld = plan->SL1;
  } else if (n == "emb0") {
    *ptr = base + w.emb0;
    *ld = plan->cfg.embed_dim;
  } else if (n == "vec") {  // [E,4] unit vector, length
    *ptr = base + w.vec;
    *ld = 4;
  } else if (n == "dvec" && with_forces) {  // [E,4] dE/dr_e
    *ptr = base + w.dvec;
    *ld = 4;
  } else {
    return fail(AA_ERR_INVALID, "aa_model_debug_tap: unknown tap");
  }
  return int(*ld);
}
