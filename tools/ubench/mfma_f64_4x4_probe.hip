// Layout probe for v_mfma_f64_4x4x4_4b_f64 (gfx950): one-hot A / B lanes -> which lane of D becomes 1.
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/mfma_f64_4x4_probe.hip -o tools/ubench/mfma_f64_4x4_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void probe(double* out) {
  const int la = blockIdx.x >> 6, lb = blockIdx.x & 63, lane = threadIdx.x;
  const double a = lane == la ? 1.0 : 0.0, b = lane == lb ? 1.0 : 0.0;
  const double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
  out[size_t(blockIdx.x) * 64 + lane] = d;
}
int main() {
  double* out;
  hipMalloc(&out, 4096 * 64 * 8);
  probe<<<4096, 64>>>(out);
  std::vector<double> h(4096 * 64);
  hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost);
  for (int la = 0; la < 64; ++la)
    for (int lb = 0; lb < 64; ++lb)
      for (int l = 0; l < 64; ++l)
        if (h[(size_t(la) * 64 + lb) * 64 + l] != 0.0) printf("%d %d %d %g\n", la, lb, l, h[(size_t(la) * 64 + lb) * 64 + l]);
  return 0;
}
