/* TEST HOST, plain C99: the driver of INTEGRATION.md section 3 written out -- what a LAMMPS pair style does with
 * include/allegro_amd.h and no Python in the process.  The frame is the `pair_allegro` tensor contract of the reference
 * (allegro/_compile.py:10-14,28-63: positions with ghost atoms appended, no cell, edges grouped by center atom), the
 * expected energies / forces are the reference's own outputs on it (tests/golden/model_c2_ghost.npz).
 *
 *   host_c99 <model file written by allegro_amd.export.write_host_model> <frame file> [tolerance]
 *
 * frame file: "AAFRAME1" | int64 N, E, nlocal | f32 pos[N][3] | i32 center[E] | i32 nbr[E] | i32 types[N] |
 *             f32 e_ref[N] | f32 f_ref[N][3]
 * Build (tests/test_host_programs.py):  gcc -std=c99 -D__HIP_PLATFORM_AMD__ -I include -I /opt/rocm/include host_c99.c
 *                                       -L allegro_amd -lallegro_amd -L /opt/rocm/lib -lamdhip64 -lm                      */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <hip/hip_runtime_api.h>

#include "allegro_amd.h"

#define CHECK_AA(call)                                                                  \
  do {                                                                                  \
    int rc_ = (call);                                                                   \
    if (rc_ != AA_OK) {                                                                 \
      fprintf(stderr, "%s -> %d: %s\n", #call, rc_, aa_last_error());                   \
      return 2;                                                                         \
    }                                                                                   \
  } while (0)
#define CHECK_HIP(call)                                                                 \
  do {                                                                                  \
    hipError_t e_ = (call);                                                             \
    if (e_ != hipSuccess) {                                                             \
      fprintf(stderr, "%s -> %s\n", #call, hipGetErrorString(e_));                      \
      return 2;                                                                         \
    }                                                                                   \
  } while (0)

static void* dev_copy(const void* host, size_t bytes) {
  void* d = NULL;
  if (hipMalloc(&d, bytes ? bytes : 4) != hipSuccess) return NULL;
  if (bytes && hipMemcpy(d, host, bytes, hipMemcpyHostToDevice) != hipSuccess) return NULL;
  return d;
}

int main(int argc, char** argv) {
  if (argc < 3) {
    fprintf(stderr, "usage: %s model.aamodel frame.bin [tol]\n", argv[0]);
    return 2;
  }
  const double tol = argc > 3 ? atof(argv[3]) : 5e-5; /* tests/model/test_allegro.py:72-74 of the reference (fp32) */

  /* ---- the frame -------------------------------------------------------------------------------------------- */
  FILE* fp = fopen(argv[2], "rb");
  char magic[8];
  int64_t hdr[3];
  if (!fp || fread(magic, 1, 8, fp) != 8 || memcmp(magic, "AAFRAME1", 8) != 0 || fread(hdr, 8, 3, fp) != 3) {
    fprintf(stderr, "bad frame file\n");
    return 2;
  }
  const int64_t N = hdr[0], E = hdr[1], nlocal = hdr[2];
  float* pos = (float*)malloc(sizeof(float) * 3 * (size_t)N);
  int32_t* center = (int32_t*)malloc(sizeof(int32_t) * (size_t)E);
  int32_t* nbr = (int32_t*)malloc(sizeof(int32_t) * (size_t)E);
  int32_t* types = (int32_t*)malloc(sizeof(int32_t) * (size_t)N);
  float* e_ref = (float*)malloc(sizeof(float) * (size_t)N);
  float* f_ref = (float*)malloc(sizeof(float) * 3 * (size_t)N);
  if (fread(pos, 4, 3 * (size_t)N, fp) != 3 * (size_t)N || fread(center, 4, (size_t)E, fp) != (size_t)E ||
      fread(nbr, 4, (size_t)E, fp) != (size_t)E || fread(types, 4, (size_t)N, fp) != (size_t)N ||
      fread(e_ref, 4, (size_t)N, fp) != (size_t)N || fread(f_ref, 4, 3 * (size_t)N, fp) != 3 * (size_t)N) {
    fprintf(stderr, "truncated frame file\n");
    return 2;
  }
  fclose(fp);
  /* CSR over centers (LAMMPS' lists are i-major already) and the transposed CSR (edges grouped by neighbor, stable):
   * with it the forces are gathered in a fixed order -- bit-reproducible, no atomics */
  int32_t* rowptr = (int32_t*)calloc((size_t)N + 1, sizeof(int32_t));
  int32_t* trow = (int32_t*)calloc((size_t)N + 1, sizeof(int32_t));
  int32_t* tperm = (int32_t*)malloc(sizeof(int32_t) * (size_t)(E ? E : 1));
  int64_t max_degree = 0;
  for (int64_t e = 0; e < E; ++e) {
    if (e > 0 && center[e] < center[e - 1]) {
      fprintf(stderr, "edges must be grouped by center atom\n");
      return 2;
    }
    rowptr[center[e] + 1]++;
    trow[nbr[e] + 1]++;
  }
  for (int64_t n = 0; n < N; ++n) {
    if (rowptr[n + 1] > max_degree) max_degree = rowptr[n + 1];
    rowptr[n + 1] += rowptr[n];
    trow[n + 1] += trow[n];
  }
  {
    int32_t* cur = (int32_t*)malloc(sizeof(int32_t) * (size_t)(N + 1));
    memcpy(cur, trow, sizeof(int32_t) * (size_t)(N + 1));
    for (int64_t e = 0; e < E; ++e) tperm[cur[nbr[e]]++] = (int32_t)e;
    free(cur);
  }

  /* ---- the model: file -> plan -> packed device weights ------------------------------------------------------ */
  aa_model_file* mf = NULL;
  CHECK_AA(aa_model_file_open(argv[1], &mf));
  const aa_model_config* cfg = aa_model_file_config(mf);
  if (cfg->dtype != AA_F32) {
    fprintf(stderr, "this test host handles fp32 models\n");
    return 2;
  }
  aa_model_plan* plan = NULL;
  CHECK_AA(aa_model_plan_create(cfg, &plan));
  hipStream_t stream;
  CHECK_HIP(hipStreamCreate(&stream));
  const size_t wbytes = aa_model_weights_bytes(plan);
  void* blob = NULL;
  CHECK_HIP(hipMalloc(&blob, wbytes));
  CHECK_AA(aa_model_pack_weights(plan, aa_model_file_weights(mf), blob, wbytes, stream));
  const size_t ws_bytes = aa_model_workspace_bytes(plan, N, E, 1);
  void* ws = NULL;
  CHECK_HIP(hipMalloc(&ws, ws_bytes ? ws_bytes : 4));

  /* ---- one MD step ------------------------------------------------------------------------------------------- */
  void *d_pos = dev_copy(pos, sizeof(float) * 3 * (size_t)N), *d_center = dev_copy(center, sizeof(int32_t) * (size_t)E),
       *d_nbr = dev_copy(nbr, sizeof(int32_t) * (size_t)E), *d_rowptr = dev_copy(rowptr, sizeof(int32_t) * ((size_t)N + 1)),
       *d_types = dev_copy(types, sizeof(int32_t) * (size_t)N), *d_trow = dev_copy(trow, sizeof(int32_t) * ((size_t)N + 1)),
       *d_tperm = dev_copy(tperm, sizeof(int32_t) * (size_t)E);
  void *d_e = NULL, *d_f = NULL, *d_w9 = NULL;
  CHECK_HIP(hipMalloc(&d_e, sizeof(float) * (size_t)N));
  CHECK_HIP(hipMalloc(&d_f, sizeof(float) * 3 * (size_t)N));
  CHECK_HIP(hipMalloc(&d_w9, sizeof(float) * 9));
  if (!d_pos || !d_center || !d_nbr || !d_rowptr || !d_types || !d_trow || !d_tperm) {
    fprintf(stderr, "device allocation failed\n");
    return 2;
  }
  /* ---- the same transposed CSR and the graph hints from the library, on the device (what a host without a sort of its own calls once
   *      per neighbour list): must equal the host's counting sort entry by entry */
  {
    const size_t tb = aa_graph_transpose_workspace_bytes(N);
    void *d_tws = NULL, *d_trow2 = NULL, *d_tperm2 = NULL, *d_hints = NULL;
    CHECK_HIP(hipMalloc(&d_tws, tb ? tb : 4));
    CHECK_HIP(hipMalloc(&d_trow2, sizeof(int32_t) * ((size_t)N + 1)));
    CHECK_HIP(hipMalloc(&d_tperm2, sizeof(int32_t) * (size_t)(E ? E : 1)));
    CHECK_HIP(hipMalloc(&d_hints, sizeof(int32_t) * 3));
    CHECK_AA(aa_graph_transpose(N, E, (const int32_t*)d_rowptr, (const int32_t*)d_nbr, (int32_t*)d_trow2, (int32_t*)d_tperm2, (int32_t*)d_hints,
                                d_tws, tb, stream));
    int32_t* trow2 = (int32_t*)malloc(sizeof(int32_t) * ((size_t)N + 1));
    int32_t* tperm2 = (int32_t*)malloc(sizeof(int32_t) * (size_t)(E ? E : 1));
    int32_t hints[3];
    CHECK_HIP(hipStreamSynchronize(stream));
    CHECK_HIP(hipMemcpy(trow2, d_trow2, sizeof(int32_t) * ((size_t)N + 1), hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(tperm2, d_tperm2, sizeof(int32_t) * (size_t)E, hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(hints, d_hints, sizeof hints, hipMemcpyDeviceToHost));
    if (memcmp(trow2, trow, sizeof(int32_t) * ((size_t)N + 1)) != 0 || memcmp(tperm2, tperm, sizeof(int32_t) * (size_t)E) != 0 ||
        hints[2] != (int32_t)max_degree) {
      fprintf(stderr, "aa_graph_transpose disagrees with the host's counting sort (max_degree %d vs %d)\n", (int)hints[2], (int)max_degree);
      return 1;
    }
    printf("host_c99: aa_graph_transpose == host counting sort (%lld edges), hints [%d, %d) max_degree %d\n", (long long)E, (int)hints[0],
           (int)hints[1], (int)hints[2]);
    free(trow2);
    free(tperm2);
    CHECK_HIP(hipFree(d_tws));
    CHECK_HIP(hipFree(d_trow2));
    CHECK_HIP(hipFree(d_tperm2));
    CHECK_HIP(hipFree(d_hints));
  }
  aa_graph g;
  memset(&g, 0, sizeof g);
  g.num_atoms = N;
  g.num_edges = E;
  g.center = (const int32_t*)d_center;
  g.nbr = (const int32_t*)d_nbr;
  g.rowptr = (const int32_t*)d_rowptr;
  g.types = (const int32_t*)d_types;
  g.shift_vec = NULL; /* ghost layout: no cell */
  g.t_rowptr = (const int32_t*)d_trow;
  g.t_perm = (const int32_t*)d_tperm;
  g.atom_begin = 0;
  g.atom_end = nlocal; /* only local atoms are centers */
  g.max_degree = max_degree;
  CHECK_AA(aa_model_energy_forces(plan, blob, &g, d_pos, ws, ws_bytes, d_e, d_f, stream));
  CHECK_AA(aa_model_virial(plan, &g, ws, ws_bytes, d_w9, stream));
  CHECK_AA(aa_model_check(plan, stream)); /* synchronises; graph hints verified on the device */
  float* e_got = (float*)malloc(sizeof(float) * (size_t)N);
  float* f_got = (float*)malloc(sizeof(float) * 3 * (size_t)N);
  float w9[9];
  CHECK_HIP(hipMemcpy(e_got, d_e, sizeof(float) * (size_t)N, hipMemcpyDeviceToHost));
  CHECK_HIP(hipMemcpy(f_got, d_f, sizeof(float) * 3 * (size_t)N, hipMemcpyDeviceToHost));
  CHECK_HIP(hipMemcpy(w9, d_w9, sizeof w9, hipMemcpyDeviceToHost));
  double de = 0, df = 0, se = 1, sf = 1;
  for (int64_t n = 0; n < N; ++n) {
    if (fabs(e_got[n] - e_ref[n]) > de || isnan(e_got[n])) de = isnan(e_got[n]) ? 1e30 : fabs(e_got[n] - e_ref[n]);
    if (fabs(e_ref[n]) > se) se = fabs(e_ref[n]);
  }
  for (int64_t i = 0; i < 3 * N; ++i) {
    if (fabs(f_got[i] - f_ref[i]) > df || isnan(f_got[i])) df = isnan(f_got[i]) ? 1e30 : fabs(f_got[i] - f_ref[i]);
    if (fabs(f_ref[i]) > sf) sf = fabs(f_ref[i]);
  }
  printf("host_c99: N=%lld (local %lld) E=%lld max_degree=%lld  max|dE_i|=%.3e max|dF|=%.3e (incl. %lld ghost rows)  tol=%.1e x scale\n",
         (long long)N, (long long)nlocal, (long long)E, (long long)max_degree, de, df, (long long)(N - nlocal), tol);
  printf("host_c99: dE/d(strain) diag = %.6f %.6f %.6f\n", w9[0], w9[4], w9[8]);
  int bad = !(de <= tol * se) || !(df <= tol * sf);

  /* ---- a stale hint must fail loudly (never a plausible number from a truncated segment) -------------------------
   * The hint selects the tile class of the fused forward (<= 32 edges: one wave per atom).  A host that forgot to refresh
   * it after its list grew: every edge listed twice (2 x max_degree edges per atom), hint left at the old maximum. */
  if (max_degree > 16 && max_degree <= 32) {
    int32_t* c2 = (int32_t*)malloc(sizeof(int32_t) * 2 * (size_t)E);
    int32_t* n2 = (int32_t*)malloc(sizeof(int32_t) * 2 * (size_t)E);
    int32_t* r2 = (int32_t*)malloc(sizeof(int32_t) * ((size_t)N + 1));
    for (int64_t e = 0; e < E; ++e) {
      c2[2 * e] = c2[2 * e + 1] = center[e];
      n2[2 * e] = n2[2 * e + 1] = nbr[e];
    }
    for (int64_t n = 0; n <= N; ++n) r2[n] = 2 * rowptr[n];
    void *d_c2 = dev_copy(c2, sizeof(int32_t) * 2 * (size_t)E), *d_n2 = dev_copy(n2, sizeof(int32_t) * 2 * (size_t)E),
         *d_r2 = dev_copy(r2, sizeof(int32_t) * ((size_t)N + 1));
    const size_t ws2_bytes = aa_model_workspace_bytes(plan, N, 2 * E, 1);
    void* ws2 = NULL;
    CHECK_HIP(hipMalloc(&ws2, ws2_bytes));
    aa_graph g2 = g;
    g2.num_edges = 2 * E;
    g2.center = (const int32_t*)d_c2;
    g2.nbr = (const int32_t*)d_n2;
    g2.rowptr = (const int32_t*)d_r2;
    g2.t_rowptr = NULL; /* (neighbor contributions through atomics) */
    g2.t_perm = NULL;
    g2.max_degree = max_degree; /* stale: the list now has 2 x max_degree edges per atom */
    CHECK_AA(aa_model_energy_forces(plan, blob, &g2, d_pos, ws2, ws2_bytes, d_e, d_f, stream));
    const int rc = aa_model_check(plan, stream);
    CHECK_HIP(hipMemcpy(e_got, d_e, sizeof(float) * (size_t)N, hipMemcpyDeviceToHost));
    int n_nan = 0;
    for (int64_t n = 0; n < nlocal; ++n) n_nan += isnan(e_got[n]) ? 1 : 0;
    printf("host_c99: stale max_degree=%lld on a list with %lld edges per atom -> aa_model_check = %d (%s), %d of %lld local energies NaN\n",
           (long long)g2.max_degree, (long long)(2 * max_degree), rc, rc ? aa_last_error() : "ok", n_nan, (long long)nlocal);
    if (rc != AA_ERR_INVALID || n_nan == 0) bad = 1;
    /* with the true maximum the same list is evaluated (team form of the fused forward, or the staged pipeline) */
    g2.max_degree = 2 * max_degree;
    CHECK_AA(aa_model_energy_forces(plan, blob, &g2, d_pos, ws2, ws2_bytes, d_e, d_f, stream));
    CHECK_AA(aa_model_check(plan, stream));
    CHECK_HIP(hipMemcpy(e_got, d_e, sizeof(float) * (size_t)N, hipMemcpyDeviceToHost));
    for (int64_t n = 0; n < N; ++n)
      if (isnan(e_got[n])) bad = 1;
  }
  aa_model_plan_destroy(plan);
  aa_model_file_close(mf);
  printf(bad ? "host_c99: FAILED\n" : "host_c99: OK\n");
  return bad;
}
