// Micro-benchmark (gfx950): build the fused-GEMM step up from a pure fragment-pattern copy to find which
// ingredient costs the time at 2 waves/SIMD.  One wave = 32 rows; per 32-column chunk a lane reads four 16-B
// pieces of its row (the MFMA operand layout of aa_gemm.hip) and writes 16-B pieces in the accumulator layout.
//   variant 0: copy  [M,K] -> [M,K]
//   variant 1: + exact 3-way bf16 split of every chunk (VALU), results folded into the output
//   variant 2: + 24 bf16 MFMAs per chunk with weight fragments read from LDS (no barrier)
//   variant 3: + block barrier per chunk and cooperative re-staging of the 12 KB weight step through LDS
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
extern __shared__ char smem[];
__device__ __forceinline__ unsigned f2u(float x) { return __builtin_bit_cast(unsigned, x); }
__device__ __forceinline__ float u2f(unsigned x) { return __builtin_bit_cast(float, x); }
__device__ __forceinline__ void split3_pack(const v4f* a, u32x4* lv1, u32x4* lv2, u32x4* lv3) {
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    u32x4 o1, o2, o3;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      unsigned h1[2], h2[2], h3[2];
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int idx = half * 8 + q * 2 + e;
        const float x = a[idx >> 2][idx & 3];
        h1[e] = f2u(x) & 0xFFFF0000u;
        const float r = x - u2f(h1[e]);
        h2[e] = f2u(r) & 0xFFFF0000u;
        const float r2 = r - u2f(h2[e]);
        h3[e] = f2u(r2) & 0xFFFF0000u;
      }
      o1[q] = (h1[0] >> 16) | h1[1];
      o2[q] = (h2[0] >> 16) | h2[1];
      o3[q] = (h3[0] >> 16) | h3[1];
    }
    lv1[half] = o1; lv2[half] = o2; lv3[half] = o3;
  }
}
__device__ __forceinline__ v16f mma(const u32x4& w, const u32x4& x, v16f acc) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, w), __builtin_bit_cast(bf16x8, x), acc, 0, 0, 0);
}

template <int V, int NOUT_TILES>
__global__ __launch_bounds__(256) void step_kernel(const float* __restrict__ a, float* __restrict__ out, const u32x4* __restrict__ wq, long M,
                                                   int K, int ldo) {
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const long gm = ((long)blockIdx.x * 4 + wv) * 32 + (lane & 31);
  const long gmc = gm < M ? gm : M - 1;
  const int hh = lane >> 5;
  const float* p = a + gmc * K + 4 * hh;
  const int KC = K / 32;
  u32x4* wbuf = reinterpret_cast<u32x4*>(smem);
  if (V >= 2) {
    for (int i = tid; i < 768 * 2; i += 256) wbuf[i] = wq[i % 768];
    __syncthreads();
  }
  v16f acc0, acc1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
  v4f a0[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) a0[q] = *reinterpret_cast<const v4f*>(p + 8 * q);
  for (int kc = 0; kc < KC; ++kc) {
    u32x4 r[3];
    if (V == 3 || V == 4) {
      r[0] = wq[tid]; r[1] = wq[256 + tid]; r[2] = wq[512 + tid];
    }
    v4f a1[4];
    const int nk = kc + 1 < KC ? kc + 1 : kc;
    if (V == 4 || V == 5 || V == 7) {
#pragma unroll
      for (int q = 0; q < 4; ++q) a1[q] = a0[q] + v4f{1.f, 1.f, 1.f, 1.f};
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q) a1[q] = *reinterpret_cast<const v4f*>(p + nk * 32 + 8 * q);
    }
    if (V == 0) {
      if (gm < M) {
#pragma unroll
        for (int q = 0; q < 4; ++q) *reinterpret_cast<v4f*>(out + gm * ldo + kc * 32 + 8 * q + 4 * hh) = a0[q];
      }
    } else if (V == 7) {
      // full-line loads AND stores (8 lanes x 16 B per row, 8 rows per instruction): the reference streaming pattern
      const long row = ((long)blockIdx.x * 4 + wv) * 32 + (lane >> 3);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const long rr = row + 8 * q < M ? row + 8 * q : M - 1;
        const v4f x = *reinterpret_cast<const v4f*>(a + rr * K + kc * 32 + 4 * (lane & 7));
        if (row + 8 * q < M) *reinterpret_cast<v4f*>(out + (row + 8 * q) * ldo + kc * 32 + 4 * (lane & 7)) = x;
      }
    } else if (V == 6) {
      // same bytes, but every store instruction writes whole 128-B lines: 8 lanes x 16 B per row, 8 rows
      const long row = ((long)blockIdx.x * 4 + wv) * 32 + (lane >> 3);
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (row + 8 * q < M) *reinterpret_cast<v4f*>(out + (row + 8 * q) * ldo + kc * 32 + 4 * (lane & 7)) = a0[q];
    } else {
      u32x4 x1[2], x2[2], x3[2];
      split3_pack(a0, x1, x2, x3);
      if (V == 1) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            acc0[h * 4 + e] += u2f(x1[h][e] ^ x2[h][e]);
            acc1[h * 4 + e] += u2f(x3[h][e]);
          }
      } else {
        const u32x4* w = wbuf + (((V == 3 || V == 4) ? (kc & 1) : 0) * 768) + lane;
#define W_(T_, Q_) w[((T_)*6 + (Q_)) * 64]
        {
          const u32x4 p0 = W_(0, 4), q0 = W_(1, 4), p1 = W_(0, 5), q1 = W_(1, 5);
          acc0 = mma(p0, x1[0], acc0); acc1 = mma(q0, x1[0], acc1); acc0 = mma(p1, x1[1], acc0); acc1 = mma(q1, x1[1], acc1);
        }
        {
          const u32x4 p0 = W_(0, 2), q0 = W_(1, 2), p1 = W_(0, 3), q1 = W_(1, 3);
          acc0 = mma(p0, x2[0], acc0); acc1 = mma(q0, x2[0], acc1); acc0 = mma(p1, x2[1], acc0); acc1 = mma(q1, x2[1], acc1);
          acc0 = mma(p0, x1[0], acc0); acc1 = mma(q0, x1[0], acc1); acc0 = mma(p1, x1[1], acc0); acc1 = mma(q1, x1[1], acc1);
        }
        {
          const u32x4 p0 = W_(0, 0), q0 = W_(1, 0), p1 = W_(0, 1), q1 = W_(1, 1);
          acc0 = mma(p0, x3[0], acc0); acc1 = mma(q0, x3[0], acc1); acc0 = mma(p1, x3[1], acc0); acc1 = mma(q1, x3[1], acc1);
          acc0 = mma(p0, x2[0], acc0); acc1 = mma(q0, x2[0], acc1); acc0 = mma(p1, x2[1], acc0); acc1 = mma(q1, x2[1], acc1);
          acc0 = mma(p0, x1[0], acc0); acc1 = mma(q0, x1[0], acc1); acc0 = mma(p1, x1[1], acc0); acc1 = mma(q1, x1[1], acc1);
        }
#undef W_
        if (V == 3 || V == 4) {
          u32x4* d = wbuf + ((kc + 1) & 1) * 768;
          d[tid] = r[0]; d[256 + tid] = r[1]; d[512 + tid] = r[2];
          __syncthreads();
        }
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) a0[q] = a1[q];
  }
  if (V >= 1 && V != 6 && V != 7 && gm < M) {
    // NOUT_TILES tile pairs of output in the accumulator layout (the same accumulators re-stored: traffic only)
#pragma unroll
    for (int t = 0; t < NOUT_TILES; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        *reinterpret_cast<v4f*>(out + gm * ldo + t * 64 + 8 * q + 4 * hh) = v4f{acc0[4 * q], acc0[4 * q + 1], acc0[4 * q + 2], acc0[4 * q + 3]};
        *reinterpret_cast<v4f*>(out + gm * ldo + t * 64 + 32 + 8 * q + 4 * hh) = v4f{acc1[4 * q], acc1[4 * q + 1], acc1[4 * q + 2], acc1[4 * q + 3]};
      }
  }
}

template <typename F>
static float time_ms(F launch, int reps = 5) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  launch();
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  for (int i = 0; i < reps; ++i) launch();
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  return ms / reps;
}

int main() {
  const long M = 2725408;
  const int K = 192;
  float *a, *o;
  u32x4* wq;
  (void)hipMalloc(&a, size_t(M) * K * 4);
  (void)hipMalloc(&o, size_t(M) * 256 * 4);
  (void)hipMalloc(&wq, 768 * 16);
  (void)hipMemset(a, 0, size_t(M) * K * 4);
  (void)hipMemset(wq, 0, 768 * 16);
  const dim3 grid((M + 127) / 128), block(256);
#define RUN(V, NT, LDS, LDO, WHAT)                                                                                          \
  {                                                                                                                        \
    (void)hipFuncSetAttribute((const void*)step_kernel<V, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);   \
    float ms = time_ms([&] { hipLaunchKernelGGL((step_kernel<V, NT>), grid, block, LDS, 0, a, o, wq, M, K, LDO); });       \
    const double gb = double(M) * 4 * (K + ((V == 0 || V == 6 || V == 7) ? K : 64 * NT)) / 1e9;                                                 \
    printf("%-58s lds %3d KB: %7.3f ms  %7.1f GB/s (read K=%d + write %d cols)\n", WHAT, LDS / 1024, ms, gb / ms * 1e3, K, \
           (V == 0 || V == 6 || V == 7) ? K : 64 * NT);                                                                              \
  }
  RUN(0, 1, 72 * 1024, 192, "v0 copy in fragment pattern")
  RUN(0, 1, 36 * 1024, 192, "v0 copy in fragment pattern (4 waves/SIMD)")
  RUN(6, 1, 72 * 1024, 192, "v6 copy, fragment loads, FULL-LINE stores")
  RUN(6, 1, 36 * 1024, 192, "v6 copy, fragment loads, FULL-LINE stores (4 waves/SIMD)")
  RUN(7, 1, 72 * 1024, 192, "v7 copy, FULL-LINE loads and stores")
  RUN(7, 1, 36 * 1024, 192, "v7 copy, FULL-LINE loads and stores (4 waves/SIMD)")
  RUN(1, 1, 72 * 1024, 64, "v1 read + split3 (VALU) + write 64")
  RUN(1, 3, 72 * 1024, 192, "v1 read + split3 (VALU) + write 192")
  RUN(2, 1, 72 * 1024, 64, "v2 + 24 MFMA/chunk, W in LDS, no barrier, write 64")
  RUN(2, 3, 72 * 1024, 192, "v2 + 24 MFMA/chunk, W in LDS, no barrier, write 192")
  RUN(3, 1, 72 * 1024, 64, "v3 + W re-staged per chunk + barrier, write 64")
  RUN(3, 3, 72 * 1024, 192, "v3 + W re-staged per chunk + barrier, write 192")
  RUN(4, 1, 72 * 1024, 64, "v4 register operands (no A loads), W re-staged + barrier, w64")
  RUN(5, 1, 72 * 1024, 64, "v5 register operands, W resident, no barrier, write 64")
  RUN(3, 3, 48 * 1024, 192, "v3 ... 3 blocks/CU")
  RUN(3, 3, 36 * 1024, 192, "v3 ... 4 blocks/CU")
  RUN(2, 3, 36 * 1024, 192, "v2 ... 4 blocks/CU")
  (void)hipDeviceSynchronize();
  return 0;
}
