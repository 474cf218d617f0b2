// C++-registered dispatcher op for the whole step -- the form an AOTInductor package / a Python-free host
// (LAMMPS pair_allegro, which loads nequip-compile'd packages: allegro/_compile.py:10-14,68-74) can call:
//
//   allegro_amd_native::energy_forces(Tensor pos, Tensor edge_index, Tensor atom_types, Tensor? shift_vec,
//                                     int[] config, Tensor weights)
//       -> (Tensor atom_energy [N], Tensor forces [N,3], Tensor virial [1,3,3])
//
// Everything the op needs travels in its arguments, so it survives torch.export / AOTI packaging: `config` is an int
// list holding the serialized aa_model_config (hyper-parameters + the Clebsch-Gordan non-zeros of every layer,
// layout below, written by allegro_amd/export.py), `weights` the packed device blob of aa_model_pack_weights.
// Plans are cached per config content.  The op is a thin host wrapper over the C ABI (include/allegro_amd.h):
// it builds the center-sorted CSR view with ATen ops, takes the workspace from the caching allocator and
// launches on the current stream.  There is no CPU kernel: only a Meta (shape) kernel besides the GPU one.
#include <ATen/ATen.h>
#include <c10/core/DeviceGuard.h>
#include <c10/hip/HIPStream.h>
#include <torch/library.h>

#include <cstring>
#include <limits>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "allegro_amd.h"

namespace {

constexpr int64_t kMagic = 0x414c4c4547524f32;  // "ALLEGRO2" (format 2: blob without the 16x16x32 weight copies, layout digest mandatory)
constexpr int kHeader = 30;

struct PlanEntry {
  aa_model_plan* plan = nullptr;
  aa_model_file* file = nullptr;  // owns the parsed config (Clebsch-Gordan tables) the plan was created from
  ~PlanEntry() {
    if (plan) aa_model_plan_destroy(plan);
    if (file) aa_model_file_close(file);
  }
};

std::mutex g_mu;
std::map<std::string, std::unique_ptr<PlanEntry>> g_plans;

// `config`: the serialized aa_model_config written by allegro_amd/export.py: serialize_config and parsed by the C ABI
// (aa_model_file_from_words; word layout in csrc/aa_hostfile.hip).  A plan owns Clebsch-Gordan tables in the memory of the
// device that was current when it was created: the cache key carries the device index and the caller holds a device guard.
const PlanEntry& plan_for(at::IntArrayRef config, int device_index) {
  const int64_t* w = config.data();
  const int64_t n = int64_t(config.size());
  TORCH_CHECK(n >= kHeader && (w[0] >> 8) == (kMagic >> 8), "allegro_amd: not a serialized model config");
  TORCH_CHECK(w[0] == kMagic, "allegro_amd: this package was exported by another version of allegro_amd (config format '",
              char(w[0] & 0xff), "', this library reads '", char(kMagic & 0xff), "'): its weight blob has another layout; re-export the model");
  std::string key(reinterpret_cast<const char*>(w), size_t(n) * 8);
  key.append(reinterpret_cast<const char*>(&device_index), sizeof(device_index));
  std::lock_guard<std::mutex> lock(g_mu);
  auto it = g_plans.find(key);
  if (it != g_plans.end()) return *it->second;
  auto e = std::make_unique<PlanEntry>();
  int rc = aa_model_file_from_words(w, n, &e->file);
  TORCH_CHECK(rc == 0, "allegro_amd: bad model config (", rc, "): ", aa_last_error());
  rc = aa_model_plan_create(aa_model_file_config(e->file), &e->plan);
  TORCH_CHECK(rc == 0, "aa_model_plan_create failed (", rc, "): ", aa_last_error());
  // the weight blob travels with the config: it must have been packed for THIS plan's layout (default options)
  const uint64_t digest = aa_model_file_layout_digest(e->file);
  TORCH_CHECK(digest != 0, "allegro_amd: the config carries no blob-layout digest; re-export the model");
  TORCH_CHECK(digest == aa_model_plan_layout_hash(e->plan),
              "allegro_amd: the weight blob was packed for another plan layout (kernel-selection options differ); "
              "re-export the model (allegro_amd.export.ExportableAllegro packs for the default options)");
  return *(g_plans[key] = std::move(e));
}

// Center-sorted CSR view of an edge list, kept between calls.  An MD driver that holds its neighbour list for several
// steps hands the op the SAME `edge_index` / `atom_types` tensors each time: the sortedness check, the sort, both CSR
// builds and the degree reduction (two host syncs) are then paid once per list instead of once per step.  Keyed on
// identity + in-place version of the two tensors, which the entry keeps alive so that their addresses cannot be
// recycled (the scheme of HipAllegroModel._graph_for).  Identity + version do not see writes through data_ptr() /
// accessors / from_blob memory (a C++ host refilling a persistent tensor), so every HIT is also validated by content:
// the list's 128-bit fingerprint (aa_graph_fingerprint, one pass over the list, no host sync) is compared ON THE DEVICE
// with the one taken when the entry was built; a mismatch turns this call's outputs into NaN and raises on the next
// call (the flag travels through pinned host memory).  One entry per device: a driver that rebuilds its list every
// step simply replaces it.  The workspace is NOT part of the entry: it comes from the caching allocator per call, which
// is what keeps two host threads / streams evaluating the same list apart.
struct GraphEntry {
  at::Tensor key_ei, key_types;  // kept alive
  uint32_t ver_ei = 0, ver_types = 0;
  int64_t N = -1;
  bool permuted = false;
  at::Tensor perm, center, nbr, rowptr, trow, tperm, types;
  at::Tensor fp;          // int64 [2] device: content fingerprint at build time
  at::Tensor stale_host;  // bool [1] pinned host: set (asynchronously) by a hit whose fingerprint differed
  int64_t max_degree = 0;
};
std::map<int, GraphEntry> g_graphs;  // device index -> most recent graph (guarded by g_mu)

bool tensor_version(const at::Tensor& t, uint32_t* v) {
  try {
    *v = t._version();
    return true;
  } catch (...) {  // inference tensors carry no version counter: never cached
    return false;
  }
}

// fingerprint of the caller's tensors as they are (no copies): int64 edge_index with unit column stride, int32 / int64 types
bool fingerprintable(const at::Tensor& edge_index, const at::Tensor& atom_types) {
  return edge_index.scalar_type() == at::kLong && (edge_index.size(1) == 0 || edge_index.stride(1) == 1) &&
         (atom_types.scalar_type() == at::kLong || atom_types.scalar_type() == at::kInt) && atom_types.is_contiguous();
}
at::Tensor fingerprint(const at::Tensor& edge_index, const at::Tensor& atom_types, hipStream_t stream) {
  at::Tensor fp = at::empty({2}, edge_index.options().dtype(at::kLong));
  const int rc = aa_graph_fingerprint(edge_index.data_ptr<int64_t>(), edge_index.size(1) > 0 ? edge_index.stride(0) : 0, edge_index.size(1),
                                      atom_types.data_ptr(), atom_types.scalar_type() == at::kLong ? 1 : 0, atom_types.numel(),
                                      reinterpret_cast<uint64_t*>(fp.data_ptr<int64_t>()), stream);
  TORCH_CHECK(rc == 0, "aa_graph_fingerprint failed (", rc, "): ", aa_last_error());
  return fp;
}

GraphEntry build_graph(const at::Tensor& edge_index, const at::Tensor& atom_types, int64_t N, const at::TensorOptions& popt) {
  GraphEntry ge;
  ge.N = N;
  const int64_t E = edge_index.size(1);
  at::Tensor ei = edge_index.to(at::kLong);
  if (E > 1) {  // edges must be grouped by center (LAMMPS' i-major lists are); sort stably otherwise
    const bool sorted = at::all(ei[0].slice(0, 1) >= ei[0].slice(0, 0, E - 1)).item<bool>();
    if (!sorted) {
      ge.perm = at::argsort(ei[0], /*stable=*/true, 0, false);
      ge.permuted = true;
      ei = ei.index_select(1, ge.perm);
    }
  }
  auto i32 = popt.dtype(at::kInt);
  ge.center = ei[0].to(at::kInt).contiguous();
  ge.nbr = ei[1].to(at::kInt).contiguous();
  ge.rowptr = at::zeros({N + 1}, i32);
  ge.trow = at::zeros({N + 1}, i32);
  ge.rowptr.slice(0, 1).copy_(at::cumsum(at::bincount(ei[0], {}, N), 0));
  ge.trow.slice(0, 1).copy_(at::cumsum(at::bincount(ei[1], {}, N), 0));
  ge.tperm = at::argsort(ei[1], /*stable=*/true, 0, false).to(at::kInt).contiguous();
  ge.types = atom_types.reshape({-1}).to(at::kInt).contiguous();
  // (one more host read next to the sortedness check above: selects the fused per-atom-tile kernels)
  ge.max_degree = E > 0 ? (ge.rowptr.slice(0, 1) - ge.rowptr.slice(0, 0, N)).max().item<int64_t>() : 0;
  return ge;
}

std::tuple<at::Tensor, at::Tensor, at::Tensor> energy_forces_gpu(const at::Tensor& pos, const at::Tensor& edge_index,
                                                                 const at::Tensor& atom_types,
                                                                 const std::optional<at::Tensor>& shift_vec,
                                                                 at::IntArrayRef config, const at::Tensor& weights) {
  TORCH_CHECK(pos.is_cuda() && edge_index.is_cuda() && atom_types.is_cuda() && weights.is_cuda(),
              "allegro_amd::energy_forces: tensors must live on the GPU; there is no CPU fallback");
  TORCH_CHECK(edge_index.get_device() == pos.get_device() && atom_types.get_device() == pos.get_device() &&
                  weights.get_device() == pos.get_device(),
              "allegro_amd::energy_forces: all tensors must live on the same device");
  // plan tables, workspace and launches all belong to the device of `pos`, whatever the caller's current device is
  const c10::DeviceGuard device_guard(pos.device());
  const PlanEntry& pe = plan_for(config, int(pos.get_device()));
  TORCH_CHECK(pos.dim() == 2 && pos.size(1) == 3 && edge_index.dim() == 2 && edge_index.size(0) == 2, "bad shapes");
  const int64_t N = pos.size(0), E = edge_index.size(1);
  const int64_t dt = config[1];
  TORCH_CHECK(pos.scalar_type() == (dt == AA_F32 ? at::kFloat : at::kDouble), "positions must be in the model dtype");
  at::Tensor p = pos.contiguous();
  hipStream_t stream = c10::hip::getCurrentHIPStream(pos.get_device()).stream();
  // graph structure: from the cache when the caller hands the same (unmodified) tensors again
  GraphEntry ge;
  at::Tensor stale;  // defined on a cache hit: device bool, true when the list's contents changed behind the tensors
  {
    uint32_t ve = 0, vt = 0;
    const bool printable = fingerprintable(edge_index, atom_types);
    const bool keyed = tensor_version(edge_index, &ve) && tensor_version(atom_types, &vt) && printable;
    if (!printable)  // (such callers used to be cached before hits were validated by content; say what it costs, once)
      TORCH_WARN_ONCE("allegro_amd::energy_forces: edge_index is not a contiguous int64 [2,E] tensor (or atom_types not contiguous "
                      "int32/int64): the neighbour list cannot be fingerprinted, so its CSR is rebuilt (sort + two host reads) on "
                      "EVERY call instead of once per list.  Pass edge_index.long().contiguous() to get the cache back.");
    bool hit = false;
    if (keyed) {
      std::lock_guard<std::mutex> lock(g_mu);
      auto it = g_graphs.find(int(pos.get_device()));
      if (it != g_graphs.end() && it->second.key_ei.is_same(edge_index) && it->second.key_types.is_same(atom_types) &&
          it->second.ver_ei == ve && it->second.ver_types == vt && it->second.N == N) {
        if (*it->second.stale_host.data_ptr<bool>()) {  // an EARLIER hit found other contents behind these tensors
          g_graphs.erase(it);
          TORCH_CHECK(false, "allegro_amd::energy_forces: the contents of edge_index / atom_types changed although the tensors and "
                             "their versions did not (written through a raw pointer?): the previous call on this list returned "
                             "NaN.  Hand over a new tensor, or modify it with a tensor operation, whenever the list changes.");
        }
        ge = it->second;
        hit = true;
      }
    }
    if (hit) {
      stale = at::ne(fingerprint(edge_index, atom_types, stream), ge.fp).any();
      ge.stale_host.copy_(stale.reshape({1}), /*non_blocking=*/true);
    } else {
      ge = build_graph(edge_index, atom_types, N, pos.options());
      if (keyed) {
        ge.key_ei = edge_index;
        ge.key_types = atom_types;
        ge.ver_ei = ve;
        ge.ver_types = vt;
        ge.fp = fingerprint(edge_index, atom_types, stream);
        ge.stale_host = at::zeros({1}, at::TensorOptions().dtype(at::kBool).pinned_memory(true));
        std::lock_guard<std::mutex> lock(g_mu);
        g_graphs[int(pos.get_device())] = ge;
      }
    }
  }
  const size_t wsb = aa_model_workspace_bytes(pe.plan, N, E, 1);
  // per call from the caching allocator (stream-safe).  The request is rounded up to 1/16 of its leading power of two: the edge
  // count of an MD run creeps up and down with every neighbour list, and an exact-size request misses the cached block at every
  // new maximum -- a fresh multi-GB device allocation (~0.5 s for 30 GB on this stack) instead of a reuse.
  size_t wsr = wsb;
  {
    size_t p2 = 1;
    while (p2 * 2 <= wsb) p2 *= 2;
    const size_t q = std::max<size_t>(p2 / 16, size_t(1) << 20);
    wsr = (wsb + q - 1) / q * q;
  }
  at::Tensor ws = at::empty({int64_t(wsr)}, pos.options().dtype(at::kByte));
  // what depends on VALUES that change while the list stays put is rebuilt every call: the periodic shift vectors
  at::Tensor svc;
  if (shift_vec.has_value()) {
    svc = shift_vec->to(pos.scalar_type());
    if (ge.permuted) svc = svc.index_select(0, ge.perm);
    svc = svc.contiguous();
  }
  aa_graph g{};
  g.num_atoms = N;
  g.num_edges = E;
  g.center = ge.center.data_ptr<int32_t>();
  g.nbr = ge.nbr.data_ptr<int32_t>();
  g.rowptr = ge.rowptr.data_ptr<int32_t>();
  g.types = ge.types.data_ptr<int32_t>();
  g.shift_vec = svc.defined() ? svc.data_ptr() : nullptr;
  g.t_rowptr = ge.trow.data_ptr<int32_t>();
  g.t_perm = ge.tperm.data_ptr<int32_t>();
  g.max_degree = ge.max_degree;
  at::Tensor e_atom = at::empty({N}, pos.options()), forces = at::empty({N, 3}, pos.options());
  TORCH_CHECK(size_t(weights.numel()) * weights.element_size() >= aa_model_weights_bytes(pe.plan),
              "allegro_amd::energy_forces: weight blob too small for this config");
  const int rc = aa_model_energy_forces(pe.plan, weights.data_ptr(), &g, p.data_ptr(), ws.data_ptr(), wsb,
                                        e_atom.data_ptr(), forces.data_ptr(), stream);
  TORCH_CHECK(rc == 0, "aa_model_energy_forces failed (", rc, "): ", aa_last_error());
  // strain derivative W = sum_e dE/dr_e (x) r_e from the per-edge data the step left in the workspace; reported in
  // LAMMPS' / nequip's VIRIAL_KEY convention, virial = -dE/d(strain) (ForceStressOutput, EXT), shape [1,3,3]
  at::Tensor w9 = at::empty({9}, pos.options());
  const int rv = aa_model_virial(pe.plan, &g, ws.data_ptr(), wsb, w9.data_ptr(), stream);
  TORCH_CHECK(rv == 0, "aa_model_virial failed (", rv, "): ", aa_last_error());
  at::Tensor virial = w9.neg().reshape({1, 3, 3});
  if (stale.defined()) {  // (no host sync: a list whose contents changed under the cache yields NaN, never a plausible number)
    const at::Tensor nan = at::full({}, std::numeric_limits<double>::quiet_NaN(), pos.options());
    e_atom = at::where(stale, nan, e_atom);
    forces = at::where(stale, nan, forces);
    virial = at::where(stale, nan, virial);
  }
  return {e_atom, forces, virial};
}

std::tuple<at::Tensor, at::Tensor, at::Tensor> energy_forces_meta(const at::Tensor& pos, const at::Tensor& edge_index,
                                                                  const at::Tensor& atom_types,
                                                                  const std::optional<at::Tensor>& shift_vec,
                                                                  at::IntArrayRef config, const at::Tensor& weights) {
  return {pos.new_empty({pos.size(0)}), pos.new_empty({pos.size(0), 3}), pos.new_empty({1, 3, 3})};
}

}  // namespace

TORCH_LIBRARY(allegro_amd_native, m) {
  m.def(
      "energy_forces(Tensor pos, Tensor edge_index, Tensor atom_types, Tensor? shift_vec, int[] config, Tensor weights)"
      " -> (Tensor, Tensor, Tensor)");
}
TORCH_LIBRARY_IMPL(allegro_amd_native, CUDA, m) { m.impl("energy_forces", &energy_forces_gpu); }
TORCH_LIBRARY_IMPL(allegro_amd_native, Meta, m) { m.impl("energy_forces", &energy_forces_meta); }
