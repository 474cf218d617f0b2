// Micro-benchmark (gfx950): which HBM streaming rate do the access patterns of the step's kernels reach, as a
// function of occupancy (waves per SIMD, set through the LDS allocation), bytes in flight per wave (unroll) and
// load width?  Patterns:
//   copy4     float4 grid-stride copy, R read streams : 1 write stream                  (the guide's 6.3 TB/s case)
//   rowsd     one wave per "atom" of 28 edges: per edge four 256-B rows, one dword per lane (TP forward pattern),
//             one 256-B row written per edge
//   rows4     same bytes, but every load is 16 B per lane (four edges per instruction)
//   frag      GEMM fragment pattern: lane = (row l&31, half l>>5), 4 x 16 B of the lane's own row per 32-deep chunk,
//             one 64-wide row in, one out
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/hbm_stream.hip -o tools/ubench/hbm_stream.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float v4f __attribute__((ext_vector_type(4)));
extern __shared__ char smem[];
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <int R, int U, bool NT>
__global__ __launch_bounds__(256) void copy4_kernel(const v4f* __restrict__ a, v4f* __restrict__ out, long n4) {
  const long stride = (long)gridDim.x * 256 * U;
  for (long i = (long)blockIdx.x * 256 * U + threadIdx.x; i < n4; i += stride) {
    v4f acc[U];
#pragma unroll
    for (int u = 0; u < U; ++u) acc[u] = v4f{0, 0, 0, 0};
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const long j = i + u * 256;
        if (j < n4) acc[u] += NT ? __builtin_nontemporal_load(a + r * n4 + j) : a[r * n4 + j];
      }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long j = i + u * 256;
      if (j < n4) {
        if (NT) __builtin_nontemporal_store(acc[u], out + j);
        else out[j] = acc[u];
      }
    }
  }
}

// wave per atom, DEG edges per atom, NR rows of 64 floats read per edge (row-major [E, NR*64]), 64 floats written
template <int U, bool NT>
__global__ __launch_bounds__(256) void rowsd_kernel(const float* __restrict__ a, float* __restrict__ out, long natoms, int deg, int nr) {
  const int lane = threadIdx.x & 63;
  const long wstride = (long)gridDim.x * 4;
  for (long n = (long)blockIdx.x * 4 + (threadIdx.x >> 6); n < natoms; n += wstride) {
    const long e0 = n * deg;
    for (int e = 0; e < deg; e += U) {
      float v[U][4];
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float* p = a + ((e0 + e + u) * nr + (r < nr ? r : nr - 1)) * 64 + lane;
          v[u][r] = (e + u < deg) ? (NT ? __builtin_nontemporal_load(p) : *p) : 0.f;
        }
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (e + u < deg) {
          const float s = v[u][0] + v[u][1] + v[u][2] + v[u][3];
          if (NT) __builtin_nontemporal_store(s, out + (e0 + e + u) * 64 + lane);
          else out[(e0 + e + u) * 64 + lane] = s;
        }
    }
  }
}

// same traffic, 16 B per lane: one instruction covers 4 edges x 64 floats of one row block
template <int U, bool NT>
__global__ __launch_bounds__(256) void rows4_kernel(const float* __restrict__ a, float* __restrict__ out, long natoms, int deg, int nr) {
  const int lane = threadIdx.x & 63, sub = lane >> 4, c4 = (lane & 15) * 4;
  const long wstride = (long)gridDim.x * 4;
  for (long n = (long)blockIdx.x * 4 + (threadIdx.x >> 6); n < natoms; n += wstride) {
    const long e0 = n * deg;
    for (int e = 0; e < deg; e += 4 * U) {
      v4f v[U][4];
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int ee = e + 4 * u + sub;
          const v4f* p = reinterpret_cast<const v4f*>(a + ((e0 + ee) * nr + (r < nr ? r : nr - 1)) * 64 + c4);
          v[u][r] = ee < deg ? (NT ? __builtin_nontemporal_load(p) : *p) : v4f{0, 0, 0, 0};
        }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int ee = e + 4 * u + sub;
        if (ee < deg) {
          const v4f s = v[u][0] + v[u][1] + v[u][2] + v[u][3];
          v4f* q = reinterpret_cast<v4f*>(out + (e0 + ee) * 64 + c4);
          if (NT) __builtin_nontemporal_store(s, q);
          else *q = s;
        }
      }
    }
  }
}

// GEMM fragment pattern: per wave 32 rows x 64 floats in (2 chunks of 4 x 16 B per lane), same out
template <int U, bool NT>
__global__ __launch_bounds__(256) void frag_kernel(const float* __restrict__ a, float* __restrict__ out, long ntiles) {
  const int lane = threadIdx.x & 63;
  const long wstride = (long)gridDim.x * 4;
  for (long t = (long)blockIdx.x * 4 + (threadIdx.x >> 6); t < ntiles; t += wstride * U) {
    v4f v[U][8];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long tt = t + u * wstride;
      const float* p = a + (tt * 32 + (lane & 31)) * 64 + 4 * (lane >> 5);
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const v4f* pp = reinterpret_cast<const v4f*>(p + 8 * q);
        v[u][q] = tt < ntiles ? (NT ? __builtin_nontemporal_load(pp) : *pp) : v4f{0, 0, 0, 0};
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long tt = t + u * wstride;
      float* p = out + (tt * 32 + (lane & 31)) * 64 + 4 * (lane >> 5);
      if (tt < ntiles)
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          v4f* pp = reinterpret_cast<v4f*>(p + 8 * q);
          const v4f s = v[u][q] * 1.5f;
          if (NT) __builtin_nontemporal_store(s, pp);
          else *pp = s;
        }
    }
  }
}

struct Timer {
  hipEvent_t a, b;
  Timer() { (void)hipEventCreate(&a); (void)hipEventCreate(&b); }
  template <class F> float best(F f, int reps = 4) {
    float m = 1e30f;
    for (int i = 0; i < reps; ++i) {
      (void)hipEventRecord(a, 0);
      f();
      (void)hipEventRecord(b, 0);
      (void)hipEventSynchronize(b);
      float ms;
      (void)hipEventElapsedTime(&ms, a, b);
      if (i > 0 && ms < m) m = ms;
    }
    return m;
  }
};

int main() {
  const long E = 2725408 / 4 * 4;             // C4's edge count
  const int deg = 28;
  const long natoms = E / deg;
  float *a, *out;
  CHECK(hipMalloc(&a, size_t(E) * 256 * sizeof(float)));   // up to 4 rows of 64 floats per edge
  CHECK(hipMalloc(&out, size_t(E) * 64 * sizeof(float)));
  CHECK(hipMemset(a, 0, size_t(E) * 256 * sizeof(float)));
  CHECK(hipMemset(out, 0, size_t(E) * 64 * sizeof(float)));
  Timer tm;
  const int occs[] = {1, 2, 3, 4, 6, 8};
  for (int occ : occs) {
    const int lds = (160 * 1024) / occ - 1024;
    const int grid = 256 * occ;
    auto report = [&](const char* name, int U, bool nt, double bytes, float ms) {
      printf("%-8s occ=%d U=%d nt=%d  %8.1f us  %7.0f GB/s\n", name, occ, U, (int)nt, ms * 1e3, bytes / ms * 1e-6);
      fflush(stdout);
    };
    const long n4 = E * 16;                                  // one 64-float row per edge, in float4
#define RUN_COPY(R, U, NT)                                                                                             \
  {                                                                                                                    \
    CHECK(hipFuncSetAttribute((const void*)copy4_kernel<R, U, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));  \
    float ms = tm.best([&] { copy4_kernel<R, U, NT><<<grid, 256, lds, 0>>>((const v4f*)a, (v4f*)out, n4); });          \
    report("copy4_r" #R, U, NT, double(n4) * 16 * (R + 1), ms);                                                        \
  }
    RUN_COPY(1, 1, false) RUN_COPY(1, 2, false) RUN_COPY(1, 4, false) RUN_COPY(1, 4, true)
    RUN_COPY(4, 1, false) RUN_COPY(4, 2, false) RUN_COPY(4, 4, false) RUN_COPY(4, 2, true)
#define RUN_ROWS(K, U, NT)                                                                                             \
  {                                                                                                                    \
    CHECK(hipFuncSetAttribute((const void*)K<U, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));                \
    float ms = tm.best([&] { K<U, NT><<<grid, 256, lds, 0>>>(a, out, natoms, deg, 4); });                              \
    report(#K, U, NT, double(E) * 256 * 5, ms);                                                                        \
  }
    RUN_ROWS(rowsd_kernel, 1, false) RUN_ROWS(rowsd_kernel, 2, false) RUN_ROWS(rowsd_kernel, 4, false) RUN_ROWS(rowsd_kernel, 4, true)
    RUN_ROWS(rows4_kernel, 1, false) RUN_ROWS(rows4_kernel, 2, false) RUN_ROWS(rows4_kernel, 2, true)
#define RUN_FRAG(U, NT)                                                                                                \
  {                                                                                                                    \
    CHECK(hipFuncSetAttribute((const void*)frag_kernel<U, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));      \
    float ms = tm.best([&] { frag_kernel<U, NT><<<grid, 256, lds, 0>>>(a, out, E / 32); });                            \
    report("frag", U, NT, double(E / 32) * 32 * 256 * 2, ms);                                                          \
  }
    RUN_FRAG(1, false) RUN_FRAG(2, false) RUN_FRAG(2, true)
  }
  return 0;
}
