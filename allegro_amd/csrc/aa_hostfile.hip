// Host-side model files for Python-free hosts (LAMMPS pair styles, C drivers): everything aa_model_plan_create and
// aa_model_pack_weights need -- hyper-parameters, the Clebsch-Gordan non-zeros of every layer, the parameters in the
// reference's own state_dict layout as float64 -- in one flat little-endian file that allegro_amd.export.write_host_model
// writes from a HipAllegroModel (i.e. from a reference checkpoint: load_state_dict(reference.state_dict())).
// No device code: plain host parsing behind the C ABI (include/allegro_amd.h, section 5).
//
//   file   = "AAMODEL1" | int64 n_words | int64 words[n_words] | int64 n_tensors | tensor*
//   words  = the serialized aa_model_config of csrc/torch_ops.cpp / allegro_amd/export.py: serialize_config
//            [magic, dtype, num_types, num_bessels, l_max, num_layers, num_scalar, num_tensor, embed_dim, embed_mlp_depth,
//             embed_mlp_width, latent_mlp_depth, latent_mlp_width, readout_mlp_depth, readout_mlp_width, forward_weight_init,
//             has_scales, has_shifts, embed_kind, spline_span, bits(poly_p), bits(avg_num_neighbors), bits(act_const),
//             env_shared_weights, act_kind[0..2] (one byte each), bits(act_consts[0..2]), bessel_convention, layout digest]
//            then per layer [mul, d1, d2, dout, num_paths, coupling, nnz, i[nnz], j[nnz], k[nnz], path[nnz], bits(val)[nnz]]
//   tensor = int64 slot | int64 numel | float64 data[numel]      slot: position of the pointer in aa_model_raw_weights
#include <cstdio>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "aa_common.h"

namespace {
constexpr int64_t kWordsMagic = 0x414c4c4547524f32;  // "ALLEGRO2" (the config format of csrc/torch_ops.cpp)
constexpr int kHeaderWords = 30;
constexpr int kNumSlots = 5 + AA_MAX_MLP_LAYERS + 2 + AA_MAX_LAYERS * AA_MAX_MLP_LAYERS + AA_MAX_LAYERS + AA_MAX_MLP_LAYERS + 3;

double bits_to_double(int64_t b) {
  double d;
  std::memcpy(&d, &b, 8);
  return d;
}
}  // namespace

struct aa_model_file {
  aa_model_config cfg{};
  aa_model_raw_weights raw{};
  uint64_t layout_digest = 0;
  std::vector<std::vector<int32_t>> ints;
  std::vector<std::vector<double>> vals;
  std::vector<std::vector<double>> tensors;
};

// slot -> pointer field of aa_model_raw_weights (declaration order of include/allegro_amd.h)
static const double** raw_slot(aa_model_raw_weights& r, int slot) {
  int s = slot;
  if (s == 0) return &r.rmax_recip;
  if (s == 1) return &r.bessel_weights;
  if (s == 2) return &r.center_embed;
  if (s == 3) return &r.neighbor_embed;
  if (s == 4) return &r.basis_linear;
  s -= 5;
  if (s < AA_MAX_MLP_LAYERS) return &r.embed_mlp[s];
  s -= AA_MAX_MLP_LAYERS;
  if (s == 0) return &r.env_embed_linear;
  if (s == 1) return &r.first_proj;
  s -= 2;
  if (s < AA_MAX_LAYERS * AA_MAX_MLP_LAYERS) return &r.latent[s / AA_MAX_MLP_LAYERS][s % AA_MAX_MLP_LAYERS];
  s -= AA_MAX_LAYERS * AA_MAX_MLP_LAYERS;
  if (s < AA_MAX_LAYERS) return &r.tp_weights[s];
  s -= AA_MAX_LAYERS;
  if (s < AA_MAX_MLP_LAYERS) return &r.readout[s];
  s -= AA_MAX_MLP_LAYERS;
  if (s == 0) return &r.scales;
  if (s == 1) return &r.shifts;
  if (s == 2) return &r.spline_weights;
  return nullptr;
}

// Elements the reference's state_dict holds for `slot` under the hyper-parameters `c` (the shapes documented at
// aa_model_raw_weights in include/allegro_amd.h; allegro_amd/nn.py: _raw_tensors writes exactly these); 0 = the slot does not
// exist for this model.  aa_model_pack_weights derives every read from the same quantities, so a tensor of any other length is
// refused when the file is opened instead of being read past its end.
static int64_t expected_numel(const aa_model_config& c, int slot) {
  const int64_t S = c.num_scalar, u = c.num_tensor, L = c.num_layers, T = c.num_types, B = c.num_bessels, S0 = c.embed_dim;
  const int64_t R = c.l_max + 1, We = c.env_shared_weights ? u : R * u;
  auto mlp = [](int64_t din, int64_t hidden, int64_t dout, int depth, int i) -> int64_t {
    if (i < 0 || i > depth) return 0;
    return (i == 0 ? din : hidden) * (i == depth ? dout : hidden);
  };
  int s = slot;
  const bool spline = c.embed_kind == 1;
  if (s == 0) return T * T;
  if (s == 1) return spline ? 0 : B;
  if (s == 2 || s == 3) return spline ? 0 : T * (S0 / 2);
  if (s == 4) return spline ? 0 : B * S0;
  s -= 5;
  if (s < AA_MAX_MLP_LAYERS) return mlp(S0, c.embed_mlp_width, S, c.embed_mlp_depth, s);
  s -= AA_MAX_MLP_LAYERS;
  if (s == 0) return S * R * u;
  if (s == 1) return S * (S + We);
  s -= 2;
  if (s < AA_MAX_LAYERS * AA_MAX_MLP_LAYERS) {
    const int l = s / AA_MAX_MLP_LAYERS, i = s % AA_MAX_MLP_LAYERS;
    if (l >= L) return 0;
    return mlp(S * (l + 1) + u, c.latent_mlp_width, S + (l < L - 1 ? We : 0), c.latent_mlp_depth, i);
  }
  s -= AA_MAX_LAYERS * AA_MAX_MLP_LAYERS;
  if (s < AA_MAX_LAYERS) return s < L ? (c.tps[s].coupling ? u * int64_t(c.tps[s].num_paths) : int64_t(c.tps[s].num_paths)) : 0;
  s -= AA_MAX_LAYERS;
  if (s < AA_MAX_MLP_LAYERS) return mlp(S * (L + 1), c.readout_mlp_width, 1, c.readout_mlp_depth, s);
  s -= AA_MAX_MLP_LAYERS;
  if (s == 0) return c.has_scales ? T : 0;
  if (s == 1) return c.has_shifts ? T : 0;
  if (s == 2) return spline ? T * T * S0 * B : 0;
  return 0;
}

static int parse_words(const int64_t* w, int64_t n, aa_model_file* f) {
  if (!w || n < kHeaderWords || (w[0] >> 8) != (kWordsMagic >> 8)) return aa::fail(AA_ERR_INVALID, "model config: not a serialized aa_model_config");
  if (w[0] != kWordsMagic) return aa::fail(AA_ERR_INVALID, "model config: written by another version of allegro_amd (config format differs); re-export the model");
  aa_model_config& c = f->cfg;
  c.dtype = int32_t(w[1]);
  c.num_types = int32_t(w[2]);
  c.num_bessels = int32_t(w[3]);
  c.l_max = int32_t(w[4]);
  c.num_layers = int32_t(w[5]);
  c.num_scalar = int32_t(w[6]);
  c.num_tensor = int32_t(w[7]);
  c.embed_dim = int32_t(w[8]);
  c.embed_mlp_depth = int32_t(w[9]);
  c.embed_mlp_width = int32_t(w[10]);
  c.latent_mlp_depth = int32_t(w[11]);
  c.latent_mlp_width = int32_t(w[12]);
  c.readout_mlp_depth = int32_t(w[13]);
  c.readout_mlp_width = int32_t(w[14]);
  c.forward_weight_init = int32_t(w[15]);
  c.has_scales = int32_t(w[16]);
  c.has_shifts = int32_t(w[17]);
  c.embed_kind = int32_t(w[18]);
  c.spline_span = int32_t(w[19]);
  c.poly_p = bits_to_double(w[20]);
  c.avg_num_neighbors = bits_to_double(w[21]);
  c.act_const = bits_to_double(w[22]);
  c.env_shared_weights = int32_t(w[23]);
  for (int i = 0; i < 3; ++i) {
    c.act_kind[i] = int32_t((w[24] >> (8 * i)) & 0xff);
    c.act_consts[i] = bits_to_double(w[25 + i]);
  }
  c.bessel_convention = int32_t(w[28]);
  f->layout_digest = uint64_t(w[29]);
  if (c.num_layers < 1 || c.num_layers > AA_MAX_LAYERS) return aa::fail(AA_ERR_INVALID, "model config: bad layer count");
  if (c.num_types < 1 || c.num_bessels < 1 || c.l_max < 0 || c.l_max > 3 || c.num_scalar < 1 || c.num_tensor < 1 || c.embed_dim < 1 ||
      c.embed_mlp_depth < 0 || c.embed_mlp_depth >= AA_MAX_MLP_LAYERS || c.latent_mlp_depth < 0 || c.latent_mlp_depth >= AA_MAX_MLP_LAYERS ||
      c.readout_mlp_depth < 0 || c.readout_mlp_depth >= AA_MAX_MLP_LAYERS || c.embed_mlp_width < 0 || c.latent_mlp_width < 0 ||
      c.readout_mlp_width < 0 || c.num_types > 4096 || c.num_scalar > 65536 || c.num_tensor > 65536 || c.embed_dim > 65536 ||
      c.embed_mlp_width > 65536 || c.latent_mlp_width > 65536 || c.readout_mlp_width > 65536 || c.num_bessels > 65536)
    return aa::fail(AA_ERR_INVALID, "model config: hyper-parameter out of range");
  int64_t o = kHeaderWords;
  for (int l = 0; l < c.num_layers; ++l) {
    if (o + 7 > n) return aa::fail(AA_ERR_INVALID, "model config: truncated");
    aa_tp_desc& d = c.tps[l];
    d.mul = int32_t(w[o]);
    d.d1 = int32_t(w[o + 1]);
    d.d2 = int32_t(w[o + 2]);
    d.dout = int32_t(w[o + 3]);
    d.num_paths = int32_t(w[o + 4]);
    d.coupling = int32_t(w[o + 5]);
    d.nnz = int32_t(w[o + 6]);
    o += 7;
    if (d.nnz < 0 || o + 5 * int64_t(d.nnz) > n) return aa::fail(AA_ERR_INVALID, "model config: truncated");
    if (d.mul < 1 || d.d1 < 1 || d.d2 < 1 || d.dout < 1 || d.num_paths < 1) return aa::fail(AA_ERR_INVALID, "model config: bad tensor-product shape");
    const int64_t lim[4] = {d.d1, d.d2, d.dout, d.num_paths};  // every non-zero indexes inside its operand (the kernels trust them)
    for (int q = 0; q < 4; ++q)
      for (int t = 0; t < d.nnz; ++t)
        if (w[o + int64_t(q) * d.nnz + t] < 0 || w[o + int64_t(q) * d.nnz + t] >= lim[q])
          return aa::fail(AA_ERR_INVALID, "model config: Clebsch-Gordan index out of range");
    const int32_t** dst[4] = {&d.nz_i, &d.nz_j, &d.nz_k, &d.nz_path};
    for (int q = 0; q < 4; ++q) {
      f->ints.emplace_back(size_t(d.nnz));
      for (int t = 0; t < d.nnz; ++t) f->ints.back()[size_t(t)] = int32_t(w[o + t]);
      *dst[q] = f->ints.back().data();
      o += d.nnz;
    }
    f->vals.emplace_back(size_t(d.nnz));
    for (int t = 0; t < d.nnz; ++t) f->vals.back()[size_t(t)] = bits_to_double(w[o + t]);
    d.nz_val = f->vals.back().data();
    o += d.nnz;
  }
  return AA_OK;
}

extern "C" int aa_model_file_from_words(const int64_t* words, int64_t num_words, aa_model_file** out) {
  AA_REQUIRE(out, "aa_model_file_from_words: null output");
  auto f = std::make_unique<aa_model_file>();
  if (int rc = parse_words(words, num_words, f.get())) return rc;
  *out = f.release();
  return AA_OK;
}

extern "C" int aa_model_file_open(const char* path, aa_model_file** out) {
  AA_REQUIRE(path && out, "aa_model_file_open: null argument");
  FILE* fp = std::fopen(path, "rb");
  if (!fp) return aa::fail(AA_ERR_INVALID, "aa_model_file_open: cannot open the file");
  auto f = std::make_unique<aa_model_file>();
  auto bad = [&](const char* m) {
    std::fclose(fp);
    return aa::fail(AA_ERR_INVALID, m);
  };
  char magic[8];
  int64_t nw = 0, nt = 0;
  if (std::fread(magic, 1, 8, fp) != 8 || std::memcmp(magic, "AAMODEL1", 8) != 0) return bad("aa_model_file_open: not an allegro_amd host model file");
  if (std::fread(&nw, 8, 1, fp) != 1 || nw < kHeaderWords || nw > (int64_t(1) << 24)) return bad("aa_model_file_open: bad config length");
  std::vector<int64_t> words(size_t(nw), 0);
  if (std::fread(words.data(), 8, size_t(nw), fp) != size_t(nw)) return bad("aa_model_file_open: truncated config");
  if (int rc = parse_words(words.data(), nw, f.get())) {
    std::fclose(fp);
    return rc;
  }
  if (std::fread(&nt, 8, 1, fp) != 1 || nt < 0 || nt > kNumSlots) return bad("aa_model_file_open: bad tensor count");
  for (int64_t t = 0; t < nt; ++t) {
    int64_t slot = -1, numel = -1;
    if (std::fread(&slot, 8, 1, fp) != 1 || std::fread(&numel, 8, 1, fp) != 1) return bad("aa_model_file_open: truncated tensor header");
    if (slot < 0 || slot >= kNumSlots) return bad("aa_model_file_open: unknown tensor slot");  // (checked as int64: 2^32 + k is not slot k)
    const double** dst = raw_slot(f->raw, int(slot));
    if (!dst || *dst) return bad("aa_model_file_open: unknown or repeated tensor slot");
    const int64_t want = expected_numel(f->cfg, int(slot));
    if (want == 0 || numel != want) {  // before anything is allocated or read: a short tensor would be read past its end by pack
      std::fclose(fp);
      return aa::fail(AA_ERR_INVALID, "aa_model_file_open: tensor slot " + std::to_string(slot) + " has " + std::to_string(numel) +
                                          " elements, the model's hyper-parameters imply " + std::to_string(want));
    }
    f->tensors.emplace_back(size_t(std::max<int64_t>(numel, 1)));
    if (std::fread(f->tensors.back().data(), 8, size_t(numel), fp) != size_t(numel)) return bad("aa_model_file_open: truncated tensor");
    *dst = f->tensors.back().data();
  }
  std::fclose(fp);
  for (int slot = 0; slot < kNumSlots; ++slot)  // every tensor the hyper-parameters call for is there (pack dereferences them)
    if (expected_numel(f->cfg, slot) > 0 && !*raw_slot(f->raw, slot))
      return aa::fail(AA_ERR_INVALID, "aa_model_file_open: tensor slot " + std::to_string(slot) + " is missing");
  *out = f.release();
  return AA_OK;
}

extern "C" const aa_model_config* aa_model_file_config(const aa_model_file* f) { return f ? &f->cfg : nullptr; }
extern "C" const aa_model_raw_weights* aa_model_file_weights(const aa_model_file* f) { return f ? &f->raw : nullptr; }
extern "C" uint64_t aa_model_file_layout_digest(const aa_model_file* f) { return f ? f->layout_digest : 0; }
extern "C" void aa_model_file_close(aa_model_file* f) { delete f; }
