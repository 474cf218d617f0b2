// TEST INFRASTRUCTURE ONLY -- a tiny CPU emulation of the HIP execution model.
//
// Purpose: compile the UNMODIFIED product kernels (allegro_amd/csrc/*.hip) for the host so that
// indexing / table / segment / MFMA-fragment logic can be exercised by `pytest -m "not gpu"` in a
// container without a GPU.  It is NOT a product path: allegro_amd/_lib.py only ever loads the
// gfx950 library and fails loudly without it; this header is found only when tests/emu/build_emu.py
// puts tests/emu/include first on the include path.
//
// Model: one block at a time; every GPU thread is a ucontext fiber; __syncthreads() and the
// wave-level exchanges (__shfl*, MFMA) are rendezvous points among the fibers of a block / wave.
// MFMA fragment layouts follow /opt/skills/guides/cdna_hip_programming.md §3:
//   v_mfma_f32_32x32x2_f32: A[i=l&31][k=l>>5], B[k=l>>5][j=l&31], C/D col=l&31,
//   row=(r&3)+8*(r>>2)+4*(l>>5).
#pragma once
#include <ucontext.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __shared__
#define __launch_bounds__(...)
#define __restrict__

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct uint3_emu {
  unsigned x, y, z;
};
extern uint3_emu threadIdx, blockIdx;
extern dim3 blockDim, gridDim;
extern unsigned char aa_smem[];  // the single dynamic-LDS symbol used by all kernels

typedef void* hipStream_t;
typedef int hipError_t;
#define hipSuccess 0
#define hipMemcpyHostToDevice 1
#define hipMemcpyDeviceToHost 2
#define hipMemcpyDeviceToDevice 3
inline hipError_t hipMalloc(void** p, size_t n) { *p = malloc(n ? n : 1); return *p ? 0 : 2; }
inline hipError_t hipFree(void* p) { free(p); return 0; }
constexpr unsigned hipHostMallocDefault = 0;
inline hipError_t hipHostMalloc(void** p, size_t n, unsigned) { *p = malloc(n ? n : 1); return *p ? 0 : 2; }
inline hipError_t hipHostFree(void* p) { free(p); return 0; }
inline hipError_t hipMemcpy(void* d, const void* s, size_t n, int) { memcpy(d, s, n); return 0; }
inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, int, hipStream_t) { memcpy(d, s, n); return 0; }
inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { memset(d, v, n); return 0; }
inline hipError_t hipGetLastError() { return 0; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return 0; }
// hipGraph API: not emulated (aa_model_plan_enable_graph reports an error under the emulator)
typedef void* hipGraph_t;
typedef void* hipGraphExec_t;
enum { hipStreamNonBlocking = 1, hipStreamCaptureModeThreadLocal = 1 };
inline hipError_t hipStreamCreateWithFlags(hipStream_t*, unsigned) { return 801; }
inline hipError_t hipStreamDestroy(hipStream_t) { return 0; }
inline hipError_t hipStreamBeginCapture(hipStream_t, int) { return 801; }
inline hipError_t hipStreamEndCapture(hipStream_t, hipGraph_t*) { return 801; }
inline hipError_t hipGraphInstantiate(hipGraphExec_t*, hipGraph_t, void*, void*, size_t) { return 801; }
inline hipError_t hipGraphLaunch(hipGraphExec_t, hipStream_t) { return 801; }
inline hipError_t hipGraphDestroy(hipGraph_t) { return 0; }
inline hipError_t hipGraphExecDestroy(hipGraphExec_t) { return 0; }
inline const char* hipGetErrorString(hipError_t) { return "emu"; }
typedef int hipEvent_t;
inline hipError_t hipEventCreate(hipEvent_t* e) { *e = 0; return 0; }
inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return 0; }
inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return 0; }
inline hipError_t hipEventDestroy(hipEvent_t) { return 0; }
inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.f; return 0; }
inline hipError_t hipFuncSetAttribute(const void*, int, int) { return 0; }
inline hipError_t hipGetDevice(int* d) { *d = 0; return 0; }
constexpr int hipDeviceAttributeMultiprocessorCount = 0;
inline hipError_t hipDeviceGetAttribute(int* v, int, int) { *v = 3; return 0; }  // a few "CUs": persistent kernels loop
#define hipFuncAttributeMaxDynamicSharedMemorySize 8

namespace emu {
constexpr int WAVE = 64;
struct Fiber {
  ucontext_t ctx;
  char* stack = nullptr;
  bool done = false;
  int wait_kind = 0;  // 0 running, 1 block barrier, 2 wave rendezvous
};
struct State {
  ucontext_t sched;
  std::vector<Fiber> fibers;
  int cur = -1;
  int nthreads = 0;
  std::function<void()> body;
  // wave exchange scratch: per wave, per lane, up to 3 x 8 bytes
  std::vector<double> xchg;  // [nwaves][WAVE][4]
};
State& st();
void yield_block_barrier();
void wave_rendezvous();
void launch(dim3 grid, dim3 block, size_t smem, const std::function<void()>& body);
inline int lane() { return st().cur % WAVE; }
inline int wave() { return st().cur / WAVE; }
inline double* slot(int w, int l) { return &st().xchg[(size_t(w) * WAVE + l) * 4]; }
}  // namespace emu

#define hipLaunchKernelGGL(kern, grid, block, smem, stream, ...) \
  emu::launch(dim3(grid), dim3(block), (smem), [=]() { kern(__VA_ARGS__); })

inline void __syncthreads() { emu::yield_block_barrier(); }
inline void __threadfence_block() {}

template <typename T>
inline T __shfl(T v, int src, int width = 64) {
  static_assert(sizeof(T) <= 8, "shfl payload");
  int l = emu::lane(), w = emu::wave();
  memcpy(emu::slot(w, l), &v, sizeof(T));
  emu::wave_rendezvous();
  int base = (l / width) * width;
  T r;
  memcpy(&r, emu::slot(w, base + (src % width)), sizeof(T));
  emu::wave_rendezvous();
  return r;
}
template <typename T>
inline T __shfl_xor(T v, int mask, int width = 64) {
  int l = emu::lane();
  return __shfl(v, (l ^ mask) % width + 0 * width, width);
}
template <typename T>
inline T __shfl_down(T v, int delta, int width = 64) {
  int l = emu::lane() % width;
  int src = l + delta < width ? l + delta : l;
  return __shfl(v, src, width);
}

template <typename T>
inline T __shfl_up(T v, int delta, int width = 64) {
  int l = emu::lane() % width;
  int src = l - delta >= 0 ? l - delta : l;
  return __shfl(v, src, width);
}

// gfx950 lane-exchange primitives (aa_common.h issues them as inline ISA on the device)
#define AA_HAVE_LANE_OPS 1  // the permlane-swap primitives and the scheduling anchor of aa_common.h are supplied below (namespace aa)
// swap the upper half (odd `width`-lane rows) of a with the lower half (even rows) of b
inline void aa_emu_permlane_swap(float& a, float& b, int width) {
  const int l = emu::lane();
  const bool odd = (l / width) & 1;
  const float ga = __shfl(a, odd ? l : l + width);  // even rows read a of the partner odd row
  const float gb = __shfl(b, odd ? l - width : l);  // odd rows read b of the partner even row
  if (odd) a = gb; else b = ga;
}
namespace aa {
inline void permlane32_swap(float& a, float& b) { aa_emu_permlane_swap(a, b, 32); }
inline void permlane16_swap(float& a, float& b) { aa_emu_permlane_swap(a, b, 16); }
inline void permlane32_swap4(float* a, float* b) {
  for (int i = 0; i < 4; ++i) aa_emu_permlane_swap(a[i], b[i], 32);
}
inline void permlane16_swap4(float* a, float* b) {
  for (int i = 0; i < 4; ++i) aa_emu_permlane_swap(a[i], b[i], 16);
}
template <class V>
inline void anchor(V&) {}  // (a scheduling anchor on the device: no semantics)
inline void opaque_scalar(int&) {}
inline void opaque_vector(unsigned&) {}
}  // namespace aa
inline int __builtin_amdgcn_update_dpp(int, int v, int ctrl, int, int, bool) {
  const int l = emu::lane();
  int src = l;
  if (ctrl >= 0x121 && ctrl <= 0x12F) src = (l & ~15) | ((l + (ctrl - 0x120)) & 15);  // row_ror: lane i reads i+n (mod 16)
  else if (ctrl == 0x141) src = (l & ~7) | (7 - (l & 7));                             // row_half_mirror
  else if (ctrl >= 0 && ctrl <= 0xFF) src = (l & ~3) | ((ctrl >> (2 * (l & 3))) & 3);  // quad_perm
  else abort();
  return __shfl(v, src);
}

typedef double emu_v4d __attribute__((ext_vector_type(4)));
// v_mfma_f64_16x16x4_f64: A[i = l&15][k = l>>4], B[k = l>>4][j = l&15], D[i = 4*r + (l>>4)][j = l&15] (probed on gfx950)
inline emu_v4d __builtin_amdgcn_mfma_f64_16x16x4f64(double a, double b, emu_v4d c, int, int, int) {
  int l = emu::lane(), w = emu::wave();
  double ab[2] = {a, b};
  memcpy(emu::slot(w, l), ab, sizeof(ab));
  emu::wave_rendezvous();
  emu_v4d d = c;
  const int j = l & 15;
  for (int r = 0; r < 4; ++r) {
    const int i = 4 * r + (l >> 4);
    double acc = c[r];
    for (int k = 0; k < 4; ++k) {
      double av[2], bv[2];
      memcpy(av, emu::slot(w, i + 16 * k), sizeof(av));  // lane holding A[i][k]
      memcpy(bv, emu::slot(w, j + 16 * k), sizeof(bv));  // lane holding B[k][j]
      acc += av[0] * bv[1];
    }
    d[r] = acc;
  }
  emu::wave_rendezvous();
  return d;
}

typedef float emu_v16f __attribute__((ext_vector_type(16)));
inline emu_v16f __builtin_amdgcn_mfma_f32_32x32x2f32(float a, float b, emu_v16f c, int, int, int) {
  int l = emu::lane(), w = emu::wave();
  float ab[2] = {a, b};
  memcpy(emu::slot(w, l), ab, sizeof(ab));
  emu::wave_rendezvous();
  emu_v16f d = c;
  int j = l & 31;
  for (int r = 0; r < 16; ++r) {
    int i = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
    float acc = c[r];
    for (int k = 0; k < 2; ++k) {
      float av[2], bv[2];
      memcpy(av, emu::slot(w, i + 32 * k), sizeof(av));  // lane holding A[i][k]
      memcpy(bv, emu::slot(w, j + 32 * k), sizeof(bv));  // lane holding B[k][j]
      acc = std::fmaf(av[0], bv[1], acc);
    }
    d[r] = acc;
  }
  emu::wave_rendezvous();
  return d;
}

// v_mfma_f32_32x32x16_bf16: lane l holds A[i=l&31][k=8*(l>>5)+e] and B[k=8*(l>>5)+e][j=l&31], e=0..7
typedef __bf16 emu_bf16x8 __attribute__((ext_vector_type(8)));
inline emu_v16f __builtin_amdgcn_mfma_f32_32x32x16_bf16(emu_bf16x8 a, emu_bf16x8 b, emu_v16f c, int, int, int) {
  int l = emu::lane(), w = emu::wave();
  // 2 x 16 bytes per lane -> the 4-double slot
  memcpy(emu::slot(w, l), &a, 16);
  memcpy(reinterpret_cast<char*>(emu::slot(w, l)) + 16, &b, 16);
  emu::wave_rendezvous();
  emu_v16f d = c;
  int j = l & 31;
  auto bf = [](unsigned short h) { unsigned u = unsigned(h) << 16; float f; memcpy(&f, &u, 4); return f; };
  for (int r = 0; r < 16; ++r) {
    int i = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
    float acc = c[r];
    for (int h = 0; h < 2; ++h) {
      unsigned short av[8], bv[8];
      memcpy(av, emu::slot(w, i + 32 * h), 16);
      memcpy(bv, reinterpret_cast<char*>(emu::slot(w, j + 32 * h)) + 16, 16);
      for (int e = 0; e < 8; ++e) acc += bf(av[e]) * bf(bv[e]);
    }
    d[r] = acc;
  }
  emu::wave_rendezvous();
  return d;
}
// v_mfma_f32_16x16x32_bf16: lane l holds A[i=l&15][k=8*(l>>4)+e] and B[k=8*(l>>4)+e][j=l&15], e=0..7;
// D[i = 4*(l>>4) + r][j = l&15], r = 0..3
typedef float emu_v4f_mfma __attribute__((ext_vector_type(4)));
inline emu_v4f_mfma __builtin_amdgcn_mfma_f32_16x16x32_bf16(emu_bf16x8 a, emu_bf16x8 b, emu_v4f_mfma c, int, int, int) {
  int l = emu::lane(), w = emu::wave();
  memcpy(emu::slot(w, l), &a, 16);
  memcpy(reinterpret_cast<char*>(emu::slot(w, l)) + 16, &b, 16);
  emu::wave_rendezvous();
  emu_v4f_mfma d = c;
  const int j = l & 15;
  auto bf = [](unsigned short h) { unsigned u = unsigned(h) << 16; float f; memcpy(&f, &u, 4); return f; };
  for (int r = 0; r < 4; ++r) {
    const int i = 4 * (l >> 4) + r;
    float acc = c[r];
    for (int g = 0; g < 4; ++g) {
      unsigned short av[8], bv[8];
      memcpy(av, emu::slot(w, i + 16 * g), 16);                                      // lane holding A[i][8g..8g+7]
      memcpy(bv, reinterpret_cast<char*>(emu::slot(w, j + 16 * g)) + 16, 16);        // lane holding B[8g..8g+7][j]
      for (int e = 0; e < 8; ++e) acc += bf(av[e]) * bf(bv[e]);
    }
    d[r] = acc;
  }
  emu::wave_rendezvous();
  return d;
}
inline float __builtin_amdgcn_rcpf(float x) { return 1.0f / x; }
inline float __builtin_amdgcn_exp2f(float x) { return std::exp2(x); }
inline void __builtin_amdgcn_sched_barrier(int) {}
inline unsigned __builtin_amdgcn_perm(unsigned a, unsigned b, unsigned sel) {  // v_perm_b32: byte i of the result = byte sel[i] of {a, b} (0..3: b, 4..7: a)
  const unsigned long long ab = (static_cast<unsigned long long>(a) << 32) | b;
  unsigned r = 0;
  for (int i = 0; i < 4; ++i) {
    const unsigned k = (sel >> (8 * i)) & 0xFF;
    const unsigned byte = k < 8 ? unsigned((ab >> (8 * k)) & 0xFF) : (k == 0x0C ? 0x00u : 0xFFu);
    r |= byte << (8 * i);
  }
  return r;
}
inline void __builtin_amdgcn_s_waitcnt(int) {}
inline void __builtin_amdgcn_s_barrier() { emu::yield_block_barrier(); }
#define __builtin_amdgcn_fence(...) ((void)0)  // (address-space scoped fences around a raw s_barrier)
inline void __builtin_amdgcn_wave_barrier() { emu::wave_rendezvous(); }  // fibers of a wave run one after another here
inline int __builtin_amdgcn_readfirstlane(int v) { return v; }  // only used on wave-uniform values
inline float atomicAdd(float* p, float v) { float o = *p; *p = o + v; return o; }
inline double atomicAdd(double* p, double v) { double o = *p; *p = o + v; return o; }
inline int atomicAdd(int* p, int v) { int o = *p; *p = o + v; return o; }
inline unsigned long long atomicAdd(unsigned long long* p, unsigned long long v) { unsigned long long o = *p; *p = o + v; return o; }
inline float rsqrtf(float x) { return 1.0f / std::sqrt(x); }
inline double rsqrt(double x) { return 1.0 / std::sqrt(x); }
