// Micro-benchmark (gfx950): HBM write rate of the fp64 linear layers' result stores.  A wave owns 32 rows x 128 fp64 columns (32 KB,
// contiguous in memory when ld = 128):
//   tiles   the epilogue's own pattern (aa_gemm.hip: f64_tile_epilogue): 16 MFMA tiles x 8 stores, an instruction writes 16 lanes x 8 B
//           = 128 B of each of 4 rows (pieces 1 KB apart)
//   rows    the same bytes as whole rows: an instruction writes 64 lanes x 16 B = one 1-KB row
//   tilesnt / rowsnt   with the non-temporal hint
// 1.72e6 rows (C5), one workgroup of four waves per 128 rows, grid as in the kernels.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/f64_tile_stores.hip -o tools/ubench/f64_tile_stores.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v2d __attribute__((ext_vector_type(2)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <bool NT>
__global__ __launch_bounds__(256) void tiles_kernel(double* c, long M, double v) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, li = lane & 15, lg = lane >> 4;
  const long m_base = ((long)blockIdx.x * 4 + wv) * 32;
  for (int j = 0; j < 8; ++j)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const long r = m_base + 16 * (e >> 2) + 4 * (e & 3) + lg;
      if (r < M) {
        double* p = c + r * 128 + 16 * j + li;
        if (NT) __builtin_nontemporal_store(v + e, p); else *p = v + e;
      }
    }
}
template <bool NT>
__global__ __launch_bounds__(256) void rows_kernel(double* c, long M, double v) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const long m_base = ((long)blockIdx.x * 4 + wv) * 32;
#pragma unroll 8
  for (int r = 0; r < 32; ++r) {
    if (m_base + r < M) {
      v2d* p = reinterpret_cast<v2d*>(c + (m_base + r) * 128) + lane;
      const v2d x{v + r, v - r};
      if (NT) __builtin_nontemporal_store(x, p); else *p = x;
    }
  }
}

int main() {
  const long M = 1720000;
  double* c;
  CHECK(hipMalloc(&c, M * 128 * 8));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  const dim3 grid((unsigned)((M + 127) / 128));
  auto run = [&](const char* name, auto launch) {
    for (int i = 0; i < 3; ++i) launch();
    hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 20;
    printf("%-8s %7.3f ms  %6.2f TB/s\n", name, ms, M * 128 * 8.0 / ms * 1e-9);
    return 0;
  };
  run("tiles", [&] { hipLaunchKernelGGL(tiles_kernel<false>, grid, dim3(256), 0, 0, c, M, 1.0); });
  run("tilesnt", [&] { hipLaunchKernelGGL(tiles_kernel<true>, grid, dim3(256), 0, 0, c, M, 1.0); });
  run("rows", [&] { hipLaunchKernelGGL(rows_kernel<false>, grid, dim3(256), 0, 0, c, M, 1.0); });
  run("rowsnt", [&] { hipLaunchKernelGGL(rows_kernel<true>, grid, dim3(256), 0, 0, c, M, 1.0); });
  CHECK(hipGetLastError());
  return 0;
}
