// Probe of v_mfma_f32_16x16x32_bf16 on gfx950: which (row, col) of D does (lane, register) hold, and do the A / B
// operand slots (lane group g, element e) pair up as k = 8 g + e?   hipcc --offload-arch=gfx950 mfma16_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));
__global__ void k(const bf16x8* a, const bf16x8* b, v4f* out) {
  v4f c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[threadIdx.x], b[threadIdx.x], c, 0, 0, 0);
  out[threadIdx.x] = c;
}
static unsigned short bf(float x) { unsigned u; memcpy(&u, &x, 4); return (unsigned short)(u >> 16); }
int main() {
  unsigned short ha[64][8], hb[64][8];
  int bad = 0;
  for (int slot = 0; slot < 32; ++slot) {  // k slot = 8 g + e: only this slot is non-zero in A and B
    memset(ha, 0, sizeof(ha)); memset(hb, 0, sizeof(hb));
    const int g = slot / 8, e = slot % 8;
    for (int i = 0; i < 16; ++i) { ha[16 * g + i][e] = bf(float(i + 1)); hb[16 * g + i][e] = bf(float(16 * (i + 1))); }
    bf16x8 *da, *db; v4f* dout; v4f ho[64];
    hipMalloc(&da, sizeof(ha)); hipMalloc(&db, sizeof(hb)); hipMalloc(&dout, sizeof(ho));
    hipMemcpy(da, ha, sizeof(ha), hipMemcpyHostToDevice); hipMemcpy(db, hb, sizeof(hb), hipMemcpyHostToDevice);
    k<<<1, 64>>>(da, db, dout);
    hipMemcpy(ho, dout, sizeof(ho), hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) {
      const int want_i = 4 * (l >> 4) + r, want_j = l & 15;  // assumed layout
      const float want = float(want_i + 1) * float(16 * (want_j + 1));
      if (ho[l][r] != want) { if (bad < 8) printf("slot %d lane %d r %d: got %g want %g\n", slot, l, r, ho[l][r], want); ++bad; }
    }
    hipFree(da); hipFree(db); hipFree(dout);
  }
  printf(bad ? "MFMA16 LAYOUT MISMATCH (%d)\n" : "mfma16 layout as assumed: A[i=l&15][k=8(l>>4)+e], B[k][j=l&15], D[i=4(l>>4)+r][j=l&15] (%d mismatches)\n", bad);
  return bad != 0;
}
