// fp64 matrix cores on gfx950: v_mfma_f64_16x16x4_f64 sustains 46-47 TFLOP/s, the 4-block v_mfma_f64_4x4x4_4b_f64 70-76
// (profiles/archive/r02_v14_mfma_peak.log).  cbsz / abid are IGNORED by the 4-block f64 form (tools/ubench/mfma_f64_bcast_map.hip),
// so a 16x16x4 product needs the A blocks ROTATED against the B blocks: four issues with A rotated by 0/4/8/12 lanes inside
// each 16-lane row (DPP row_ror) give the sixteen 4x4 block products.  This probe checks the result against the 16x16x4
// instruction (incl. the de-rotation of the accumulator registers) and measures the rate in the register pattern of
// gemm_mfma_f64_pipe_kernel (2 A x 4 B operands per 4-deep step).
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/mfma_f64_bcast_probe.hip -o tools/ubench/mfma_f64_bcast_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cmath>
typedef double v4d __attribute__((ext_vector_type(4)));

template <int S>
__device__ __forceinline__ double rot(double x) {  // rotate the 16-lane rows right by 4 S lanes
  if constexpr (S == 0) return x;
  const unsigned long long u = __builtin_bit_cast(unsigned long long, x);
  const int lo = __builtin_amdgcn_update_dpp(0, int(u), 0x120 + 4 * S, 0xF, 0xF, false);
  const int hi = __builtin_amdgcn_update_dpp(0, int(u >> 32), 0x120 + 4 * S, 0xF, 0xF, false);
  return __builtin_bit_cast(double, (unsigned long long)(unsigned)lo | ((unsigned long long)(unsigned)hi << 32));
}
struct A4 { double r[4]; };
__device__ __forceinline__ A4 rotations(double a) { return A4{{a, rot<1>(a), rot<2>(a), rot<3>(a)}}; }
__device__ __forceinline__ void mma4b(const A4& a, double b, v4d& acc) {
  acc[0] = __builtin_amdgcn_mfma_f64_4x4x4f64(a.r[0], b, acc[0], 0, 0, 0);
  acc[1] = __builtin_amdgcn_mfma_f64_4x4x4f64(a.r[1], b, acc[1], 0, 0, 0);
  acc[2] = __builtin_amdgcn_mfma_f64_4x4x4f64(a.r[2], b, acc[2], 0, 0, 0);
  acc[3] = __builtin_amdgcn_mfma_f64_4x4x4f64(a.r[3], b, acc[3], 0, 0, 0);
}
// rotated accumulators -> the 16x16x4 register order (register r = rows 4 r + lane / 16); DIR = +-1: rotation sense
template <int DIR>
__device__ __forceinline__ v4d derotate(const v4d& acc, int lane) {
  const int blk = (lane >> 2) & 3;
  v4d o;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int s = (DIR * (blk - r)) & 3;  // block blk of issue s holds A block (blk - DIR s) & 3
    o[r] = s == 0 ? acc[0] : s == 1 ? acc[1] : s == 2 ? acc[2] : acc[3];
  }
  return o;
}

__global__ void check(const double* in, double* out) {
  const int lane = threadIdx.x;
  v4d ref = {0, 0, 0, 0}, acc = {0, 0, 0, 0};
  for (int t = 0; t < 3; ++t) {
    const double a = in[128 * t + lane], b = in[128 * t + 64 + lane];
    ref = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, ref, 0, 0, 0);
    mma4b(rotations(a), b, acc);
  }
  const v4d p = derotate<1>(acc, lane), m = derotate<-1>(acc, lane);
  for (int r = 0; r < 4; ++r) {
    out[r * 64 + lane] = ref[r];
    out[256 + r * 64 + lane] = p[r];
    out[512 + r * 64 + lane] = m[r];
  }
}

template <int MODE>
__global__ __launch_bounds__(256, 3) void peak(const double* in, double* out, int iters) {
  double a0 = in[threadIdx.x & 63], a1 = in[64 + (threadIdx.x & 63)];
  double b[4];
  for (int j = 0; j < 4; ++j) b[j] = in[128 + 16 * j + (threadIdx.x & 15)];
  v4d acc[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = v4d{0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
    asm volatile("" : "+v"(a0), "+v"(a1));  // fresh operands every step (no hoisting of the rotations)
    if (MODE == 0) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        acc[0][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b[j], acc[0][j], 0, 0, 0);
        acc[1][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b[j], acc[1][j], 0, 0, 0);
      }
    } else {
      const A4 r0 = rotations(a0), r1 = rotations(a1);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        mma4b(r0, b[j], acc[0][j]);
        mma4b(r1, b[j], acc[1][j]);
      }
    }
  }
  double s = 0;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main() {
  double *in, *out;
  hipMalloc(&in, 512 * 8);
  hipMalloc(&out, 1 << 24);
  std::vector<double> h(512);
  for (int i = 0; i < 512; ++i) h[i] = std::sin(1.0 + i) + 0.01 * i;
  hipMemcpy(in, h.data(), 512 * 8, hipMemcpyHostToDevice);
  check<<<1, 64>>>(in, out);
  std::vector<double> o(768);
  hipMemcpy(o.data(), out, 768 * 8, hipMemcpyDeviceToHost);
  for (int v = 0; v < 2; ++v) {
    double err = 0, mag = 0;
    for (int i = 0; i < 256; ++i) {
      err = std::fmax(err, std::fabs(o[i] - o[256 * (v + 1) + i]));
      mag = std::fmax(mag, std::fabs(o[i]));
    }
    printf("rotated 4-block form, de-rotation sense %+d vs 16x16x4: max |diff| %.3e (max |ref| %.3e) -> %s\n", v ? -1 : 1, err, mag,
           err <= 1e-13 * mag ? "EQUAL" : "DIFFERENT");
  }
  for (int waves = 1; waves <= 3; ++waves)
    for (int mode = 0; mode < 2; ++mode) {
      const int iters = 20000, blocks = 256 * waves;
      hipEvent_t e0, e1;
      hipEventCreate(&e0);
      hipEventCreate(&e1);
      if (mode == 0) peak<0><<<blocks, 256>>>(in, out, 100); else peak<1><<<blocks, 256>>>(in, out, 100);
      hipEventRecord(e0);
      if (mode == 0) peak<0><<<blocks, 256>>>(in, out, iters); else peak<1><<<blocks, 256>>>(in, out, iters);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms = 0;
      (void)hipEventElapsedTime(&ms, e0, e1);
      const double flops = double(blocks) * 4 * iters * 8 * 2048.0;
      printf("%s  %d waves/SIMD: %.3f ms  %.1f TFLOP/s\n", mode ? "4x4x4_4b x4 + DPP rotations" : "16x16x4                    ", waves, ms, flops / ms * 1e-9);
    }
  return 0;
}
