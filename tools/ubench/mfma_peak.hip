// Micro-benchmark (gfx950): sustained matrix-core rate with nothing else going on -- the practical ceiling behind the
// nominal peaks (78.6 TFLOP/s fp64, 2.5 PFLOP/s bf16 at 2.4 GHz) that bench.py prices the linear layers against.
// Every wave runs ITER x ACC independent accumulator chains of v_mfma_f64_16x16x4_f64 / v_mfma_f32_32x32x16_bf16;
// occupancy 1..4 waves per SIMD through the workgroup count.  Also reports the shader clock the loop ran at
// (s_memrealtime is a fixed 100 MHz counter; clock64() counts shader cycles).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_peak.hip -o tools/ubench/mfma_peak.bin
#include <hip/hip_runtime.h>
#include <cstdio>

typedef double v4d __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <int ACC>
__global__ __launch_bounds__(256) void f64_kernel(double* out, int iters, unsigned long long* cyc) {
  v4d acc[ACC];
  for (int i = 0; i < ACC; ++i) acc[i] = v4d{0, 0, 0, 0};
  double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
  const unsigned long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < ACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  const unsigned long long t1 = clock64();
  double s = 0;
  for (int i = 0; i < ACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) *cyc = t1 - t0;
}

// the 4-block 4x4x4 form (one result per lane, 512 flop per instruction)
template <int ACC>
__global__ __launch_bounds__(256) void f64_4x4_kernel(double* out, int iters, unsigned long long* cyc) {
  double acc[ACC];
  for (int i = 0; i < ACC; ++i) acc[i] = 0.0;
  double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
  const unsigned long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < ACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[i], 0, 0, 0);
  }
  const unsigned long long t1 = clock64();
  double s = 0;
  for (int i = 0; i < ACC; ++i) s += acc[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) *cyc = t1 - t0;
}
// plain vector FMA for comparison (2 flop per lane per instruction)
template <int ACC>
__global__ __launch_bounds__(256) void f64_valu_kernel(double* out, int iters, unsigned long long* cyc) {
  double acc[ACC];
  for (int i = 0; i < ACC; ++i) acc[i] = threadIdx.x * 1e-3 + i;
  double a = 1.0 + threadIdx.x * 1e-9, b = 1e-12;
  const unsigned long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < ACC; ++i) acc[i] = __builtin_fma(acc[i], a, b);
  }
  const unsigned long long t1 = clock64();
  double s = 0;
  for (int i = 0; i < ACC; ++i) s += acc[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) *cyc = t1 - t0;
}

template <int ACC>
__global__ __launch_bounds__(256) void bf16_kernel(float* out, int iters, unsigned long long* cyc) {
  v16f acc[ACC];
  for (int i = 0; i < ACC; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  bf16x8 a, b;
  for (int r = 0; r < 8; ++r) {
    a[r] = (__bf16)(float)(threadIdx.x * 1e-3f + r);
    b[r] = (__bf16)(float)(1.f + r * 0.01f);
  }
  const unsigned long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < ACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
  }
  const unsigned long long t1 = clock64();
  float s = 0;
  for (int i = 0; i < ACC; ++i)
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) *cyc = t1 - t0;
}

int main() {
  double* out;
  unsigned long long* cyc;
  CHECK(hipMalloc(&out, 8 * 256 * 4096));
  CHECK(hipMalloc(&cyc, 8));
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int iters = 20000;
  for (int occ = 1; occ <= 4; ++occ) {
    const int grid = 256 * occ;  // 256 CUs x occ workgroups of 4 waves = occ waves per SIMD
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      f64_kernel<4><<<grid, 256>>>(out, iters, cyc);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      unsigned long long c;
      CHECK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
      if (rep == 1) {
        const double flop = double(grid) * 4 * iters * 4 * 2.0 * 16 * 16 * 4;
        printf("fp64 16x16x4   %d waves/SIMD : %8.3f ms  %7.1f TFLOP/s   %5.1f cycles/MFMA/SIMD (clock64)  => %4.2f GHz\n", occ, ms, flop / ms * 1e-9,
               double(c) / (double(iters) * 4 * occ), double(c) / (ms * 1e6));
      }
    }
  }
  for (int occ = 1; occ <= 4; ++occ) {
    const int grid = 256 * occ;
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      f64_4x4_kernel<8><<<grid, 256>>>(out, iters, cyc);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      if (rep == 1) printf("fp64 4x4x4(4b)  %d waves/SIMD : %8.3f ms  %7.1f TFLOP/s\n", occ, ms, double(grid) * 4 * iters * 8 * 512.0 / ms * 1e-9);
    }
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      f64_valu_kernel<8><<<grid, 256>>>(out, iters, cyc);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      if (rep == 1) printf("fp64 v_fma_f64  %d waves/SIMD : %8.3f ms  %7.1f TFLOP/s\n", occ, ms, double(grid) * 256 * iters * 8 * 2.0 / ms * 1e-9);
    }
  }
  for (int occ = 1; occ <= 4; ++occ) {
    const int grid = 256 * occ;
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      bf16_kernel<4><<<grid, 256>>>((float*)out, iters, cyc);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      unsigned long long c;
      CHECK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
      if (rep == 1) {
        const double flop = double(grid) * 4 * iters * 4 * 2.0 * 32 * 32 * 16;
        printf("bf16 32x32x16  %d waves/SIMD : %8.3f ms  %7.1f TFLOP/s   %5.1f cycles/MFMA/SIMD (clock64)  => %4.2f GHz\n", occ, ms, flop / ms * 1e-9,
               double(c) / (double(iters) * 4 * occ), double(c) / (ms * 1e6));
      }
    }
  }
  return 0;
}
