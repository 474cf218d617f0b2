// TEST HOST, C++ without Python: what LAMMPS' pair_allegro does with a `nequip-compile --mode aotinductor --target
// pair_allegro` package (reference: docs/guide/lammps.md:13-21; tensor contract allegro/_compile.py:10-14,68-74) --
// dlopen the op library, load the AOTInductor package, run it on [pos, edge_index, atom_types] with ghost atoms appended,
// read per-atom energies and forces INCLUDING the ghost rows LAMMPS reverse-communicates.
//
//   host_aoti <package.pt2> <frame file (see host_c99.c)> <path to liballegro_amd_torch.so | "none"> [tol]
//
// With "none" the op library is NOT loaded: loading / running the package must then fail with the dispatcher's
// missing-schema error, the failure mode the reference documents for its own accelerator ops
// (docs/guide/cuequivariance.md:91: "Could not find schema for ...").  Exit code 0 = behaved as expected.
#include <dlfcn.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <iostream>
#include <string>
#include <vector>

#include <ATen/ATen.h>
#include <torch/csrc/inductor/aoti_package/model_package_loader.h>

int main(int argc, char** argv) {
  if (argc < 4) {
    std::fprintf(stderr, "usage: %s package.pt2 frame.bin liballegro_amd_torch.so|none [tol]\n", argv[0]);
    return 2;
  }
  const double tol = argc > 4 ? std::atof(argv[4]) : 5e-5;
  const bool preload = std::strcmp(argv[3], "none") != 0;
  if (preload && !dlopen(argv[3], RTLD_NOW | RTLD_GLOBAL)) {  // registers allegro_amd_native::energy_forces with the dispatcher
    std::fprintf(stderr, "dlopen(%s): %s\n", argv[3], dlerror());
    return 2;
  }
  FILE* fp = std::fopen(argv[2], "rb");
  char magic[8];
  int64_t hdr[3];
  if (!fp || std::fread(magic, 1, 8, fp) != 8 || std::memcmp(magic, "AAFRAME1", 8) != 0 || std::fread(hdr, 8, 3, fp) != 3) {
    std::fprintf(stderr, "bad frame file\n");
    return 2;
  }
  const int64_t N = hdr[0], E = hdr[1], nlocal = hdr[2];
  std::vector<float> pos(3 * N), e_ref(N), f_ref(3 * N);
  std::vector<int32_t> center(E), nbr(E), types(N);
  if (std::fread(pos.data(), 4, 3 * N, fp) != size_t(3 * N) || std::fread(center.data(), 4, E, fp) != size_t(E) ||
      std::fread(nbr.data(), 4, E, fp) != size_t(E) || std::fread(types.data(), 4, N, fp) != size_t(N) ||
      std::fread(e_ref.data(), 4, N, fp) != size_t(N) || std::fread(f_ref.data(), 4, 3 * N, fp) != size_t(3 * N)) {
    std::fprintf(stderr, "truncated frame file\n");
    return 2;
  }
  std::fclose(fp);
  try {
    const at::Device dev(at::kCUDA, 0);
    at::Tensor t_pos = at::from_blob(pos.data(), {N, 3}, at::kFloat).to(dev);
    at::Tensor ei = at::empty({2, E}, at::kLong);
    for (int64_t e = 0; e < E; ++e) {
      ei[0][e] = int64_t(center[e]);
      ei[1][e] = int64_t(nbr[e]);
    }
    at::Tensor t_ei = ei.to(dev);
    at::Tensor t_types = at::from_blob(types.data(), {N}, at::kInt).to(at::kLong).to(dev);
    torch::inductor::AOTIModelPackageLoader loader(argv[1]);
    std::vector<at::Tensor> out = loader.run({t_pos, t_ei, t_types});  // LMP_OUTPUTS: E_i [N,1], E [1,1], F [N,3], virial [1,3,3]
    if (!preload) {
      std::printf("host_aoti: the package ran WITHOUT the op library: unexpected\n");
      return 1;
    }
    if (out.size() != 4) {
      std::printf("host_aoti: %zu outputs, expected the 4 LMP_OUTPUTS\n", out.size());
      return 1;
    }
    at::Tensor e = out[0].reshape({-1}).cpu(), f = out[2].cpu();
    at::Tensor er = at::from_blob(e_ref.data(), {N}, at::kFloat), fr = at::from_blob(f_ref.data(), {N, 3}, at::kFloat);
    const double de = (e - er).abs().max().item<double>(), df = (f - fr).abs().max().item<double>();
    const double se = std::max(1.0, er.abs().max().item<double>()), sf = std::max(1.0, fr.abs().max().item<double>());
    const double etot = out[1].item<double>(), esum = er.sum().item<double>();
    std::printf("host_aoti: N=%lld (local %lld) E=%lld  max|dE_i|=%.3e max|dF|=%.3e (incl. %lld ghost rows)  E_total=%.6f (reference %.6f)\n",
                (long long)N, (long long)nlocal, (long long)E, de, df, (long long)(N - nlocal), etot, esum);
    // a second step on the same tensors (an MD loop holding its neighbour list: the op's validated graph cache)
    std::vector<at::Tensor> out2 = loader.run({t_pos, t_ei, t_types});
    const bool same = at::equal(out2[2], out[2]) && at::equal(out2[0], out[0]);
    const bool ok = de <= tol * se && df <= tol * sf && std::fabs(etot - esum) <= 10 * tol * std::fabs(esum) && same && std::isfinite(df);
    std::printf(ok ? "host_aoti: OK\n" : "host_aoti: FAILED\n");
    return ok ? 0 : 1;
  } catch (const std::exception& ex) {
    const std::string msg = ex.what();
    std::printf("host_aoti: exception: %s\n", msg.substr(0, 600).c_str());
    if (!preload && (msg.find("Could not find schema for allegro_amd_native::energy_forces") != std::string::npos ||
                     msg.find("allegro_amd_native") != std::string::npos)) {
      std::printf("host_aoti: OK (missing op library reported as the dispatcher's schema error)\n");
      return 0;
    }
    return 1;
  }
}
