// One-hot map of v_mfma_f64_4x4x4_4b_f64 with cbsz / abid: which D lane is set by (A lane, B lane).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int CBSZ, int ABID>
__global__ void probe(double* out) {
  const int la = blockIdx.x >> 6, lb = blockIdx.x & 63, lane = threadIdx.x;
  const double a = lane == la ? 1.0 : 0.0, b = lane == lb ? 1.0 : 0.0;
  out[size_t(blockIdx.x) * 64 + lane] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, CBSZ, ABID, 0);
}
template <int CBSZ, int ABID>
void run(double* out) {
  probe<CBSZ, ABID><<<4096, 64>>>(out);
  std::vector<double> h(4096 * 64);
  hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost);
  int n = 0;
  for (int la = 0; la < 64; ++la)
    for (int lb = 0; lb < 64; ++lb)
      for (int l = 0; l < 64; ++l)
        if (h[(size_t(la) * 64 + lb) * 64 + l] != 0.0) { printf("%d %d %d %d %d\n", CBSZ, ABID, la, lb, l); ++n; }
  fprintf(stderr, "cbsz %d abid %d: %d nonzero\n", CBSZ, ABID, n);
}
int main() {
  double* out;
  hipMalloc(&out, 4096 * 64 * 8);
  run<0, 0>(out); run<1, 0>(out); run<1, 1>(out); run<2, 0>(out); run<2, 1>(out); run<2, 2>(out); run<2, 3>(out);
  return 0;
}
