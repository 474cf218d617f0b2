// Micro-benchmark (gfx950): how many bytes must a wave keep in flight to stream [M,K] fp32 rows in the MFMA
// fragment pattern of the GEMM kernels (lane = row (l&31), k-half (l>>5); one "chunk" = 32 rows x 32 floats =
// 4 KB per wave, four 16-B pieces per lane)?  Compares a register prefetch ring with the LDS-DMA ring
// (global_load_lds_dwordx4: data lands in LDS without passing through VGPRs) at 2 waves/SIMD.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/dma_stream.hip -o gpurun_out/dma_stream && gpurun_out/dma_stream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float v4f __attribute__((ext_vector_type(4)));
extern __shared__ char smem[];

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__device__ __forceinline__ void dma16(const void* g, unsigned lds_base) {
  asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"(__builtin_amdgcn_readfirstlane(lds_base)) : "memory");
}
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// layout probe: out[lane*4..] = what lane reads back at lds_base + lane*16
__global__ void probe_kernel(const float* a, float* out) {
  const int lane = threadIdx.x;
  const unsigned base = (unsigned)(uintptr_t)(smem);
  dma16(a + lane * 4, base);
  wait_vm<0>();
  const v4f v = *reinterpret_cast<const v4f*>(smem + lane * 16);
  for (int e = 0; e < 4; ++e) out[lane * 4 + e] = v[e];
}

template <int DEPTH>
__global__ __launch_bounds__(256) void reg_kernel(const float* __restrict__ a, float* __restrict__ out, long M, int K) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const long gm = ((long)blockIdx.x * 4 + wv) * 32 + (lane & 31);
  const float* p = a + (gm < M ? gm : M - 1) * K + 4 * (lane >> 5);
  const int KC = K / 32;
  v4f r[DEPTH][4];
#pragma unroll
  for (int d = 0; d < DEPTH; ++d)
#pragma unroll
    for (int q = 0; q < 4; ++q) r[d][q] = *reinterpret_cast<const v4f*>(p + (d < KC ? d : KC - 1) * 32 + 8 * q);
  v4f acc = {0, 0, 0, 0};
  for (int kc = 0; kc < KC; kc += DEPTH) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      const int nk = kc + d + DEPTH < KC ? kc + d + DEPTH : KC - 1;
      v4f cur[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) cur[q] = r[d][q];
#pragma unroll
      for (int q = 0; q < 4; ++q) r[d][q] = *reinterpret_cast<const v4f*>(p + nk * 32 + 8 * q);
      if (kc + d < KC)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc += cur[q];
    }
  }
  if (gm < M) *reinterpret_cast<v4f*>(out + gm * 8 + 4 * (lane >> 5)) = acc;
}

template <int DEPTH>
__global__ __launch_bounds__(256) void dma_kernel(const float* __restrict__ a, float* __restrict__ out, long M, int K) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const long gm = ((long)blockIdx.x * 4 + wv) * 32 + (lane & 31);
  const float* p = a + (gm < M ? gm : M - 1) * K + 4 * (lane >> 5);
  const int KC = K / 32;
  char* ring = smem + wv * DEPTH * 4096;
  const unsigned ring_u = (unsigned)(uintptr_t)ring;
  auto issue = [&](int kc, int slot) {
    const float* s = p + (kc < KC ? kc : KC - 1) * 32;
#pragma unroll
    for (int q = 0; q < 4; ++q) dma16(s + 8 * q, ring_u + slot * 4096 + q * 1024);
  };
#pragma unroll
  for (int d = 0; d < DEPTH - 1; ++d) issue(d, d);
  v4f acc = {0, 0, 0, 0};
  int slot = 0, pslot = DEPTH - 1;
  for (int kc = 0; kc < KC; ++kc) {
    issue(kc + DEPTH - 1, pslot);            // DEPTH-1 chunks ahead
    wait_vm<4 * (DEPTH - 1)>();              // chunk kc has landed
    const char* s = ring + slot * 4096 + lane * 16;
#pragma unroll
    for (int q = 0; q < 4; ++q) acc += *reinterpret_cast<const v4f*>(s + q * 1024);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the slot is free before it is refilled
    pslot = slot;
    slot = slot + 1 == DEPTH ? 0 : slot + 1;
  }
  wait_vm<0>();
  if (gm < M) *reinterpret_cast<v4f*>(out + gm * 8 + 4 * (lane >> 5)) = acc;
}

template <typename F>
static float time_ms(F launch, int reps = 5) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  launch();
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int i = 0; i < reps; ++i) launch();
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  return ms / reps;
}

int main() {
  // ---- layout probe
  {
    float *a, *o;
    CHECK(hipMalloc(&a, 1024 * 4));
    CHECK(hipMalloc(&o, 256 * 4));
    std::vector<float> h(1024);
    for (int i = 0; i < 1024; ++i) h[i] = float(i);
    CHECK(hipMemcpy(a, h.data(), 4096, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(probe_kernel, dim3(1), dim3(64), 4096, 0, a, o);
    CHECK(hipDeviceSynchronize());
    std::vector<float> r(256);
    CHECK(hipMemcpy(r.data(), o, 1024, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int i = 0; i < 256; ++i) bad += r[i] != float(i);
    printf("probe: lane l's 16 B land at lds_base + 16*l : %s (out[0..7] = %g %g %g %g %g %g %g %g)\n", bad ? "NO" : "yes", r[0], r[1], r[2],
           r[3], r[4], r[5], r[6], r[7]);
  }
  const long M = 2725408;
  for (int K : {64, 192}) {
    float *a, *o;
    CHECK(hipMalloc(&a, size_t(M) * K * 4));
    CHECK(hipMalloc(&o, size_t(M) * 8 * 4));
    CHECK(hipMemset(a, 0, size_t(M) * K * 4));
    const dim3 grid((M + 127) / 128), block(256);
    const double gb = double(M) * K * 4 / 1e9;
    // occupancy pinned to 2 blocks per CU (2 waves/SIMD) with a 72 KB dynamic LDS allocation unless stated
#define RUN_REG(D, LDS)                                                                                                   \
  {                                                                                                                      \
    hipFuncSetAttribute((const void*)reg_kernel<D>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);            \
    float ms = time_ms([&] { hipLaunchKernelGGL(reg_kernel<D>, grid, block, LDS, 0, a, o, M, K); });                     \
    printf("K=%3d reg ring depth %d  lds/block %3d KB : %7.3f ms  %7.1f GB/s\n", K, D, LDS / 1024, ms, gb / ms * 1e3);  \
  }
#define RUN_DMA(D, LDS)                                                                                                   \
  {                                                                                                                      \
    hipFuncSetAttribute((const void*)dma_kernel<D>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);            \
    float ms = time_ms([&] { hipLaunchKernelGGL(dma_kernel<D>, grid, block, LDS, 0, a, o, M, K); });                     \
    printf("K=%3d DMA ring depth %d  lds/block %3d KB : %7.3f ms  %7.1f GB/s\n", K, D, LDS / 1024, ms, gb / ms * 1e3);  \
  }
    RUN_REG(1, 72 * 1024)
    RUN_REG(2, 72 * 1024)
    RUN_REG(3, 72 * 1024)
    RUN_REG(4, 72 * 1024)
    RUN_REG(6, 72 * 1024)
    RUN_REG(2, 36 * 1024)
    RUN_REG(2, 18 * 1024)
    RUN_DMA(2, 72 * 1024)
    RUN_DMA(3, 72 * 1024)
    RUN_DMA(4, 72 * 1024)
    RUN_DMA(5, 80 * 1024)
    RUN_DMA(2, 36 * 1024)
    RUN_DMA(2, 32 * 1024)
    hipFree(a);
    hipFree(o);
  }
  CHECK(hipDeviceSynchronize());
  return 0;
}
