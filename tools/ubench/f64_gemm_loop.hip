// Where does the fp64 GEMM inner loop lose the matrix-core rate?  Variants of the 4-deep step of gemm_mfma_f64_pipe_kernel
// (2 A x 4 B operands, 8 x v_mfma_f64_16x16x4_f64), 3 workgroups of 4 waves per CU:
//   0 registers only, j-outer     1 registers only, i-outer (the kernel's order)     2 operands from LDS (6 ds_read_b64 / step)
//   3 as 2 + a workgroup barrier every 4 steps     4 as 3 + the LDS stores of the staging (no global traffic)
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/f64_gemm_loop.hip -o tools/ubench/f64_gemm_loop.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));
extern __shared__ unsigned char smem_raw[];
constexpr int LDA = 132, BN = 64, BK = 16;

template <int MODE>
__global__ __launch_bounds__(256, 3) void loop(const double* in, double* out, int steps) {
  double* smem = reinterpret_cast<double*>(smem_raw);
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  for (int i = tid; i < 2 * (BK * LDA + BK * BN); i += 256) smem[i] = in[i & 511];
  __syncthreads();
  v4d acc[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = v4d{0, 0, 0, 0};
  double a[2] = {in[lane], in[64 + lane]}, b[4] = {in[128 + lane], in[192 + lane], in[256 + lane], in[320 + lane]};
  v2d ra[4] = {{1, 2}, {3, 4}, {5, 6}, {7, 8}}, rb[2] = {{1, 2}, {3, 4}};
  const int ar = tid >> 1, ak = (tid & 1) * 8, bk = tid >> 4, bn = (tid & 15) * 4;
  int buf = 0;
  for (int st = 0; st < steps; ++st) {
    const double* a_ = smem + buf * (BK * LDA + BK * BN);
    const double* b_ = a_ + BK * LDA;
#pragma unroll
    for (int kk = 0; kk < BK; kk += 4) {
      if (MODE >= 2) {
        const int kr = kk + (lane >> 4);
#pragma unroll
        for (int i = 0; i < 2; ++i) a[i] = a_[kr * LDA + wv * 32 + i * 16 + (lane & 15)];
#pragma unroll
        for (int j = 0; j < 4; ++j) b[j] = b_[kr * BN + j * 16 + (lane & 15)];
      } else {
        asm volatile("" : "+v"(a[0]), "+v"(a[1]));
      }
      if (MODE == 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int i = 0; i < 2; ++i) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
      } else {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
      }
    }
    if (MODE >= 4) {
      double* as = smem + (buf ^ 1) * (BK * LDA + BK * BN);
      double* bs = as + BK * LDA;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        as[(ak + 2 * q) * LDA + ar] = ra[q][0];
        as[(ak + 2 * q + 1) * LDA + ar] = ra[q][1];
      }
#pragma unroll
      for (int q = 0; q < 2; ++q) *reinterpret_cast<v2d*>(bs + bk * BN + bn + 2 * q) = rb[q];
    }
    if (MODE >= 3) {
      __syncthreads();
      buf ^= 1;
    }
  }
  double s = 0;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
  out[blockIdx.x * 256 + tid] = s;
}

template <int MODE>
void run(const double* in, double* out) {
  const int steps = 4000, blocks = 256 * 3;
  const size_t smem = 2 * (BK * LDA + BK * BN) * 8;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  loop<MODE><<<blocks, 256, smem>>>(in, out, 50);
  hipEventRecord(e0);
  loop<MODE><<<blocks, 256, smem>>>(in, out, steps);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  printf("variant %d: %.3f ms  %.1f TFLOP/s\n", MODE, ms, double(blocks) * 4 * steps * 4 * 8 * 2048.0 / ms * 1e-9);
}
int main() {
  double *in, *out;
  hipMalloc(&in, 512 * 8);
  hipMalloc(&out, 1 << 24);
  double h[512];
  for (int i = 0; i < 512; ++i) h[i] = 0.5 + 0.001 * i;
  hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
  run<0>(in, out); run<1>(in, out); run<2>(in, out); run<3>(in, out); run<4>(in, out);
  return 0;
}
