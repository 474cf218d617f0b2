// On-device neighbor list: cell list -> center-sorted CSR edge list (center, nbr, rowptr, periodic shifts) -- the
// graph contract of the hot path (aa_graph), built without leaving the GPU.  In the reference
// stack the list comes from the host (nequip's neighbor-list transform, EXT) or from LAMMPS (pair_allegro); the
// conventions are those of `with_edge_vectors_` (called at allegro/nn/tensorembed.py:86):
//     r_e = pos[nbr] - pos[center] + cell_shift_e @ cell,   |r_e| < r_cut,   self-pairs only through a non-zero shift.
// Geometry is evaluated in double for either position dtype so that borderline pairs are classified the same way as
// by a float64 host list.  Deterministic: cells are filled with integer atomics and then sorted, neighbors of an
// atom are emitted in (cell offset, atom index) order.
#include "aa_common.h"

#include <algorithm>
#include <cmath>

using namespace aa;
namespace {

struct NlDev {
  int64_t N;
  const void* pos;
  double cell[9], inv[9];
  int nc[3], k[3], pbc[3];
  double rc2;
  int ncells;
  // workspace
  int* cell_count;  // [ncells+1]
  int* cell_start;  // [ncells+1]
  int* atom_cell;   // [N]
  int* atom_slot;   // [N]
  int* cell_atoms;  // [N]
  int* wrap;        // [N][3]
  double* wpos;     // [N][3] positions wrapped into the cell along periodic directions
  int* counts;      // [N+1]
  int* scan_tmp;    // [max(cells, atoms) / 4096 + 2]
};

template <typename T>
__global__ __launch_bounds__(256) void nl_bin_kernel(NlDev d) {
  const int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x;
  if (i >= d.N) return;
  const T* p = static_cast<const T*>(d.pos) + 3 * i;
  const double x = double(p[0]), y = double(p[1]), z = double(p[2]);
  double s[3];
  int w[3], c[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    s[a] = x * d.inv[0 * 3 + a] + y * d.inv[1 * 3 + a] + z * d.inv[2 * 3 + a];
    w[a] = 0;
    if (d.pbc[a]) {
      const double fl = floor(s[a]);
      w[a] = -int(fl);
      s[a] -= fl;
      if (s[a] >= 1.0) {  // rounding: -1e-17 - floor(.) == 1.0
        s[a] -= 1.0;
        w[a] -= 1;
      }
    }
    int ca = int(floor(s[a] * d.nc[a]));
    c[a] = ca < 0 ? 0 : (ca >= d.nc[a] ? d.nc[a] - 1 : ca);  // atoms outside a non-periodic box go to the border cells
  }
  d.wpos[3 * i + 0] = x + w[0] * d.cell[0] + w[1] * d.cell[3] + w[2] * d.cell[6];
  d.wpos[3 * i + 1] = y + w[0] * d.cell[1] + w[1] * d.cell[4] + w[2] * d.cell[7];
  d.wpos[3 * i + 2] = z + w[0] * d.cell[2] + w[1] * d.cell[5] + w[2] * d.cell[8];
  d.wrap[3 * i + 0] = w[0];
  d.wrap[3 * i + 1] = w[1];
  d.wrap[3 * i + 2] = w[2];
  const int cid = (c[0] * d.nc[1] + c[1]) * d.nc[2] + c[2];
  d.atom_cell[i] = cid;
  d.atom_slot[i] = atomicAdd(&d.cell_count[cid], 1);
}

// Exclusive prefix sums out[0..n] of in[0..n) (out[n] = total) in three launches: per-block totals (4096 items per
// workgroup), one workgroup scans the totals, every workgroup scans its own items from its offset.
constexpr int kScanItems = 16, kScanBlock = 256 * kScanItems;

__global__ __launch_bounds__(256) void nl_scan_totals_kernel(const int* in, int* block_tot, int64_t n) {
  int* sSum = reinterpret_cast<int*>(aa_smem);  // [256]
  const int tid = threadIdx.x;
  const int64_t lo = int64_t(blockIdx.x) * kScanBlock + int64_t(tid) * kScanItems;
  int acc = 0;
  for (int q = 0; q < kScanItems; ++q)
    if (lo + q < n) acc += in[lo + q];
  sSum[tid] = acc;
  __syncthreads();
  for (int st = 128; st >= 1; st >>= 1) {
    if (tid < st) sSum[tid] += sSum[tid + st];
    __syncthreads();
  }
  if (tid == 0) block_tot[blockIdx.x] = sSum[0];
}

// in place: block_tot[b] -> sum of block_tot[0..b); out_total = grand total.  One workgroup, chunk per thread.
__global__ __launch_bounds__(256) void nl_scan_offsets_kernel(int* block_tot, int nblocks, int* out_total) {
  int* sSum = reinterpret_cast<int*>(aa_smem);
  const int tid = threadIdx.x;
  const int chunk = (nblocks + 255) / 256;
  const int lo = std::min(nblocks, chunk * tid), hi = std::min(nblocks, lo + chunk);
  int acc = 0;
  for (int q = lo; q < hi; ++q) acc += block_tot[q];
  sSum[tid] = acc;
  __syncthreads();
  if (tid == 0) {
    int run = 0;
    for (int t = 0; t < 256; ++t) {
      const int v = sSum[t];
      sSum[t] = run;
      run += v;
    }
    *out_total = run;
  }
  __syncthreads();
  int run = sSum[tid];
  for (int q = lo; q < hi; ++q) {
    const int v = block_tot[q];
    block_tot[q] = run;
    run += v;
  }
}

__global__ __launch_bounds__(256) void nl_scan_apply_kernel(const int* in, const int* block_off, int* out, int64_t n) {
  int* sSum = reinterpret_cast<int*>(aa_smem);  // [256]
  const int tid = threadIdx.x;
  const int64_t lo = int64_t(blockIdx.x) * kScanBlock + int64_t(tid) * kScanItems;
  int v[kScanItems];
  int acc = 0;
#pragma unroll
  for (int q = 0; q < kScanItems; ++q) {
    v[q] = lo + q < n ? in[lo + q] : 0;
    acc += v[q];
  }
  sSum[tid] = acc;
  __syncthreads();
  if (tid == 0) {
    int run = block_off[blockIdx.x];
    for (int t = 0; t < 256; ++t) {
      const int x = sSum[t];
      sSum[t] = run;
      run += x;
    }
  }
  __syncthreads();
  int run = sSum[tid];
#pragma unroll
  for (int q = 0; q < kScanItems; ++q) {
    if (lo + q < n) out[lo + q] = run;
    run += v[q];
  }
}

// out[n] receives the total; `block_tot` is scratch of (n / 4096 + 1) ints
void launch_scan(const int* in, int* out, int64_t n, int* block_tot, hipStream_t s) {
  const int nb = int((n + kScanBlock - 1) / kScanBlock);
  if (nb > 0) hipLaunchKernelGGL(nl_scan_totals_kernel, dim3(nb), dim3(256), sizeof(int) * 256, s, in, block_tot, n);
  hipLaunchKernelGGL(nl_scan_offsets_kernel, dim3(1), dim3(256), sizeof(int) * 256, s, block_tot, nb, out + n);
  if (nb > 0) hipLaunchKernelGGL(nl_scan_apply_kernel, dim3(nb), dim3(256), sizeof(int) * 256, s, in, block_tot, out, n);
}

__global__ __launch_bounds__(256) void nl_cell_fill_kernel(NlDev d) {
  const int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x;
  if (i >= d.N) return;
  d.cell_atoms[d.cell_start[d.atom_cell[i]] + d.atom_slot[i]] = int(i);
}

// the slots came from atomics: sort every cell's (short) atom list so that the edge order is reproducible
__global__ __launch_bounds__(256) void nl_cell_sort_kernel(NlDev d) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= d.ncells) return;
  const int lo = d.cell_start[c], hi = d.cell_start[c + 1];
  for (int q = lo + 1; q < hi; ++q) {
    const int v = d.cell_atoms[q];
    int r = q - 1;
    while (r >= lo && d.cell_atoms[r] > v) {
      d.cell_atoms[r + 1] = d.cell_atoms[r];
      --r;
    }
    d.cell_atoms[r + 1] = v;
  }
}

template <typename T, bool FILL>
__global__ __launch_bounds__(256) void nl_pairs_kernel(NlDev d, const int32_t* rowptr, int32_t* center, int32_t* nbr,
                                                       int32_t* cell_shift, void* shift_vec) {
  const int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x;
  if (i >= d.N) return;
  const double xi = d.wpos[3 * i], yi = d.wpos[3 * i + 1], zi = d.wpos[3 * i + 2];
  const int cid = d.atom_cell[i];
  const int c2 = cid % d.nc[2], c1 = (cid / d.nc[2]) % d.nc[1], c0 = cid / (d.nc[2] * d.nc[1]);
  const int wi0 = d.wrap[3 * i], wi1 = d.wrap[3 * i + 1], wi2 = d.wrap[3 * i + 2];
  int cnt = 0;
  int64_t e = FILL ? rowptr[i] : 0;
  for (int o0 = -d.k[0]; o0 <= d.k[0]; ++o0) {
    int a0 = c0 + o0, n0 = 0;
    if (d.pbc[0]) {
      n0 = a0 >= 0 ? a0 / d.nc[0] : -((-a0 + d.nc[0] - 1) / d.nc[0]);
      a0 -= n0 * d.nc[0];
    } else if (a0 < 0 || a0 >= d.nc[0]) {
      continue;
    }
    for (int o1 = -d.k[1]; o1 <= d.k[1]; ++o1) {
      int a1 = c1 + o1, n1 = 0;
      if (d.pbc[1]) {
        n1 = a1 >= 0 ? a1 / d.nc[1] : -((-a1 + d.nc[1] - 1) / d.nc[1]);
        a1 -= n1 * d.nc[1];
      } else if (a1 < 0 || a1 >= d.nc[1]) {
        continue;
      }
      for (int o2 = -d.k[2]; o2 <= d.k[2]; ++o2) {
        int a2 = c2 + o2, n2 = 0;
        if (d.pbc[2]) {
          n2 = a2 >= 0 ? a2 / d.nc[2] : -((-a2 + d.nc[2] - 1) / d.nc[2]);
          a2 -= n2 * d.nc[2];
        } else if (a2 < 0 || a2 >= d.nc[2]) {
          continue;
        }
        const double sx = n0 * d.cell[0] + n1 * d.cell[3] + n2 * d.cell[6];
        const double sy = n0 * d.cell[1] + n1 * d.cell[4] + n2 * d.cell[7];
        const double sz = n0 * d.cell[2] + n1 * d.cell[5] + n2 * d.cell[8];
        const int cj = (a0 * d.nc[1] + a1) * d.nc[2] + a2;
        const bool home = n0 == 0 && n1 == 0 && n2 == 0;
        for (int q = d.cell_start[cj]; q < d.cell_start[cj + 1]; ++q) {
          const int j = d.cell_atoms[q];
          if (home && j == i) continue;
          const double dx = d.wpos[3 * int64_t(j)] + sx - xi, dy = d.wpos[3 * int64_t(j) + 1] + sy - yi,
                       dz = d.wpos[3 * int64_t(j) + 2] + sz - zi;
          if (dx * dx + dy * dy + dz * dz < d.rc2) {
            if constexpr (FILL) {
              center[e] = int32_t(i);
              nbr[e] = j;
              // undo the wrapping: r_e = pos[j] - pos[i] + S @ cell with S = n + w_j - w_i
              const int S0 = n0 + d.wrap[3 * int64_t(j)] - wi0, S1 = n1 + d.wrap[3 * int64_t(j) + 1] - wi1,
                        S2 = n2 + d.wrap[3 * int64_t(j) + 2] - wi2;
              if (cell_shift) {
                cell_shift[3 * e] = S0;
                cell_shift[3 * e + 1] = S1;
                cell_shift[3 * e + 2] = S2;
              }
              if (shift_vec) {
                T* sv = static_cast<T*>(shift_vec) + 3 * e;
                sv[0] = T(S0 * d.cell[0] + S1 * d.cell[3] + S2 * d.cell[6]);
                sv[1] = T(S0 * d.cell[1] + S1 * d.cell[4] + S2 * d.cell[7]);
                sv[2] = T(S0 * d.cell[2] + S1 * d.cell[5] + S2 * d.cell[8]);
              }
              ++e;
            } else {
              ++cnt;
            }
          }
        }
      }
    }
  }
  if constexpr (!FILL) d.counts[i] = cnt;
}

size_t align_up(size_t x) { return (x + 255) / 256 * 256; }

int64_t max_cells(int64_t N) { return std::max<int64_t>(64, 4 * N); }

int plan_nl(const aa_nl_input* in, void* ws, size_t ws_bytes, NlDev& d) {
  AA_REQUIRE(in && ws, "aa_nl: null argument");
  AA_REQUIRE(in->num_atoms >= 0 && in->num_atoms < (int64_t(1) << 31), "aa_nl: atom count out of range");
  AA_REQUIRE(in->dtype == AA_F32 || in->dtype == AA_F64, "aa_nl: bad dtype");
  AA_REQUIRE(in->r_cut > 0.0, "aa_nl: r_cut must be positive");
  AA_REQUIRE(in->pos || in->num_atoms == 0, "aa_nl: null positions");
  if (ws_bytes < aa_nl_workspace_bytes(in->num_atoms)) return fail(AA_ERR_WORKSPACE, "aa_nl: workspace too small");
  d.N = in->num_atoms;
  d.pos = in->pos;
  const double* c = in->cell;
  for (int q = 0; q < 9; ++q) d.cell[q] = c[q];
  const double det = c[0] * (c[4] * c[8] - c[5] * c[7]) - c[1] * (c[3] * c[8] - c[5] * c[6]) + c[2] * (c[3] * c[7] - c[4] * c[6]);
  AA_REQUIRE(std::fabs(det) > 1e-12, "aa_nl: singular cell");
  // inverse (pos = s @ cell  =>  s = pos @ inv)
  d.inv[0] = (c[4] * c[8] - c[5] * c[7]) / det;
  d.inv[1] = (c[2] * c[7] - c[1] * c[8]) / det;
  d.inv[2] = (c[1] * c[5] - c[2] * c[4]) / det;
  d.inv[3] = (c[5] * c[6] - c[3] * c[8]) / det;
  d.inv[4] = (c[0] * c[8] - c[2] * c[6]) / det;
  d.inv[5] = (c[2] * c[3] - c[0] * c[5]) / det;
  d.inv[6] = (c[3] * c[7] - c[4] * c[6]) / det;
  d.inv[7] = (c[1] * c[6] - c[0] * c[7]) / det;
  d.inv[8] = (c[0] * c[4] - c[1] * c[3]) / det;
  // perpendicular height of the cell along lattice direction a = 1 / |column a of inv|
  double h[3];
  for (int a = 0; a < 3; ++a) {
    const double n2 = d.inv[a] * d.inv[a] + d.inv[3 + a] * d.inv[3 + a] + d.inv[6 + a] * d.inv[6 + a];
    h[a] = 1.0 / std::sqrt(n2);
    d.pbc[a] = in->pbc[a] ? 1 : 0;
    d.nc[a] = std::max(1, int(std::min(1024.0, std::floor(h[a] / in->r_cut))));
  }
  while (int64_t(d.nc[0]) * d.nc[1] * d.nc[2] > max_cells(d.N)) {  // never more cells than ~4 per atom
    int a = 0;
    if (d.nc[1] > d.nc[a]) a = 1;
    if (d.nc[2] > d.nc[a]) a = 2;
    d.nc[a] = (d.nc[a] + 1) / 2;
  }
  for (int a = 0; a < 3; ++a) {
    const double width = h[a] / d.nc[a];
    d.k[a] = d.pbc[a] ? int(std::ceil(in->r_cut / width - 1e-12)) : (d.nc[a] > 1 ? 1 : 0);
    AA_REQUIRE(d.k[a] <= 64, "aa_nl: cell much smaller than r_cut (more than 64 images along one direction)");
  }
  d.rc2 = in->r_cut * in->r_cut;
  d.ncells = d.nc[0] * d.nc[1] * d.nc[2];
  char* base = static_cast<char*>(ws);
  size_t o = 0;
  auto take = [&](size_t bytes) {
    char* r = base + o;
    o += align_up(bytes);
    return r;
  };
  const size_t N = size_t(d.N), NC = size_t(max_cells(d.N));
  d.cell_count = reinterpret_cast<int*>(take(sizeof(int) * (NC + 1)));
  d.cell_start = reinterpret_cast<int*>(take(sizeof(int) * (NC + 1)));
  d.atom_cell = reinterpret_cast<int*>(take(sizeof(int) * N));
  d.atom_slot = reinterpret_cast<int*>(take(sizeof(int) * N));
  d.cell_atoms = reinterpret_cast<int*>(take(sizeof(int) * N));
  d.wrap = reinterpret_cast<int*>(take(sizeof(int) * 3 * N));
  d.wpos = reinterpret_cast<double*>(take(sizeof(double) * 3 * N));
  d.counts = reinterpret_cast<int*>(take(sizeof(int) * (N + 1)));
  d.scan_tmp = reinterpret_cast<int*>(take(sizeof(int) * (std::max(N, NC) / kScanBlock + 2)));
  return AA_OK;
}

// ---- transposed CSR of a center-sorted edge list (edges grouped by NEIGHBOUR atom, ascending edge id inside a group = the stable
// ---- argsort of nbr) and the three graph hints, without leaving the device: counting sort by atomics + a per-atom sort of the
// ---- (short) groups, which makes the result independent of the order the atomics were served in
// (a neighbour id outside [0, N) -- a malformed list -- is skipped by both passes alike: its edge appears in no group, nothing is
//  written out of bounds; the step itself reads pos[nbr] and is the caller's to keep valid)
__global__ __launch_bounds__(256) void gt_count_kernel(const int32_t* nbr, int64_t E, int64_t N, int* cnt) {
  const int64_t e = int64_t(blockIdx.x) * 256 + threadIdx.x;
  if (e >= E) return;
  const int j = nbr[e];
  if (j >= 0 && j < N) atomicAdd(&cnt[j], 1);
}
__global__ __launch_bounds__(256) void gt_place_kernel(const int32_t* nbr, int64_t E, int64_t N, const int32_t* t_rowptr, int* cursor,
                                                       int32_t* t_perm) {
  const int64_t e = int64_t(blockIdx.x) * 256 + threadIdx.x;
  if (e >= E) return;
  const int j = nbr[e];
  if (j >= 0 && j < N) t_perm[t_rowptr[j] + atomicAdd(&cursor[j], 1)] = int32_t(e);
}
// one wave per atom: every entry's final position is its RANK inside the group (edge ids are distinct), counted with wave shuffles
// from registers -- no data-dependent loop over memory, whatever order the atomics left the group in (groups of more than 256
// entries: one lane sorts in place)
__global__ __launch_bounds__(256) void gt_sort_kernel(int64_t N, const int32_t* t_rowptr, int32_t* t_perm) {
  const int lane = threadIdx.x & 63;
  const int64_t a = int64_t(blockIdx.x) * 4 + (threadIdx.x >> 6);
  if (a >= N) return;
  const int lo = __builtin_amdgcn_readfirstlane(t_rowptr[a]), d = __builtin_amdgcn_readfirstlane(t_rowptr[a + 1]) - lo;
  if (d <= 1) return;
  if (d > 256) {
    if (lane == 0) {
      for (int q = lo + 1; q < lo + d; ++q) {
        const int v = t_perm[q];
        int r = q - 1;
        while (r >= lo && t_perm[r] > v) {
          t_perm[r + 1] = t_perm[r];
          --r;
        }
        t_perm[r + 1] = v;
      }
    }
    return;
  }
  int x[4], rank[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int idx = lane + 64 * k;
    x[k] = idx < d ? t_perm[lo + idx] : 0x7fffffff;
    rank[k] = 0;
  }
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) {
    if (64 * kk < d) {  // (wave-uniform)
      const int n = d - 64 * kk < 64 ? d - 64 * kk : 64;
      for (int m = 0; m < n; ++m) {
        const int y = __shfl(x[kk], m);
#pragma unroll
        for (int k = 0; k < 4; ++k) rank[k] += y < x[k] ? 1 : 0;
      }
    }
  }
  __builtin_amdgcn_wave_barrier();  // (every entry is in a register before the first one is written back)
#pragma unroll
  for (int k = 0; k < 4; ++k)
    if (lane + 64 * k < d) t_perm[lo + rank[k]] = x[k];
}
// hints[0..2] = first atom with edges, one past the last atom with edges, largest segment (0, 0, 0 without edges)
__global__ __launch_bounds__(256) void gt_hints_kernel(int64_t N, const int32_t* rowptr, int32_t* hints) {
  int* s = reinterpret_cast<int*>(aa_smem);  // [3][256]
  int first = int(N), last = 0, dmax = 0;
  for (int64_t a = threadIdx.x; a < N; a += 256) {
    const int d = rowptr[a + 1] - rowptr[a];
    if (d > 0) {
      first = first < int(a) ? first : int(a);
      last = int(a) + 1;
      dmax = dmax > d ? dmax : d;
    }
  }
  s[threadIdx.x] = first;
  s[256 + threadIdx.x] = last;
  s[512 + threadIdx.x] = dmax;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int t = 1; t < 256; ++t) {
      first = first < s[t] ? first : s[t];
      last = last > s[256 + t] ? last : s[256 + t];
      dmax = dmax > s[512 + t] ? dmax : s[512 + t];
    }
    hints[0] = last > 0 ? first : 0;
    hints[1] = last;
    hints[2] = dmax;
  }
}

}  // namespace

extern "C" size_t aa_nl_workspace_bytes(int64_t num_atoms) {
  const size_t N = size_t(std::max<int64_t>(0, num_atoms)), NC = size_t(max_cells(num_atoms));
  return 2 * align_up(sizeof(int) * (NC + 1)) + 3 * align_up(sizeof(int) * N) + align_up(sizeof(int) * 3 * N) +
         align_up(sizeof(double) * 3 * N) + align_up(sizeof(int) * (N + 1)) +
         align_up(sizeof(int) * (std::max(N, NC) / kScanBlock + 2));
}

extern "C" int aa_nl_count(const aa_nl_input* in, void* workspace, size_t workspace_bytes, int32_t* rowptr,
                           int64_t* num_edges, aa_stream stream) {
  NlDev d{};
  if (int rc = plan_nl(in, workspace, workspace_bytes, d)) return rc;
  AA_REQUIRE(rowptr && num_edges, "aa_nl_count: null output");
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int nb = int((d.N + 255) / 256);
  AA_CHECK_HIP(hipMemsetAsync(d.cell_count, 0, sizeof(int) * (size_t(d.ncells) + 1), s));
  if (d.N > 0) {
    if (in->dtype == AA_F32)
      hipLaunchKernelGGL(nl_bin_kernel<float>, dim3(nb), dim3(256), 0, s, d);
    else
      hipLaunchKernelGGL(nl_bin_kernel<double>, dim3(nb), dim3(256), 0, s, d);
  }
  launch_scan(d.cell_count, d.cell_start, int64_t(d.ncells), d.scan_tmp, s);
  if (d.N > 0) {
    hipLaunchKernelGGL(nl_cell_fill_kernel, dim3(nb), dim3(256), 0, s, d);
    hipLaunchKernelGGL(nl_cell_sort_kernel, dim3((d.ncells + 255) / 256), dim3(256), 0, s, d);
    if (in->dtype == AA_F32)
      hipLaunchKernelGGL((nl_pairs_kernel<float, false>), dim3(nb), dim3(256), 0, s, d, nullptr, nullptr, nullptr, nullptr, nullptr);
    else
      hipLaunchKernelGGL((nl_pairs_kernel<double, false>), dim3(nb), dim3(256), 0, s, d, nullptr, nullptr, nullptr, nullptr, nullptr);
  }
  launch_scan(d.counts, rowptr, d.N, d.scan_tmp, s);
  AA_CHECK_HIP(hipGetLastError());
  int32_t total = 0;
  AA_CHECK_HIP(hipMemcpyAsync(&total, rowptr + d.N, sizeof(int32_t), hipMemcpyDeviceToHost, s));
  AA_CHECK_HIP(hipStreamSynchronize(s));
  AA_REQUIRE(total >= 0, "aa_nl_count: more than 2^31 edges");
  *num_edges = total;
  return AA_OK;
}

extern "C" int aa_nl_fill(const aa_nl_input* in, void* workspace, size_t workspace_bytes, const int32_t* rowptr,
                          int32_t* center, int32_t* nbr, int32_t* cell_shift, void* shift_vec, aa_stream stream) {
  NlDev d{};
  if (int rc = plan_nl(in, workspace, workspace_bytes, d)) return rc;
  AA_REQUIRE(rowptr && center && nbr, "aa_nl_fill: null output");
  if (d.N == 0) return AA_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int nb = int((d.N + 255) / 256);
  if (in->dtype == AA_F32)
    hipLaunchKernelGGL((nl_pairs_kernel<float, true>), dim3(nb), dim3(256), 0, s, d, rowptr, center, nbr, cell_shift, shift_vec);
  else
    hipLaunchKernelGGL((nl_pairs_kernel<double, true>), dim3(nb), dim3(256), 0, s, d, rowptr, center, nbr, cell_shift, shift_vec);
  AA_CHECK_HIP(hipGetLastError());
  return AA_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// content fingerprint of a neighbour list (aa_graph_fingerprint): lets a host that caches graph structure by tensor
// IDENTITY (csrc/torch_ops.cpp) notice that the contents behind an unchanged tensor were rewritten through a raw pointer
// ---------------------------------------------------------------------------------------------------------------------
extern "C" size_t aa_graph_transpose_workspace_bytes(int64_t num_atoms) {
  const size_t N = size_t(std::max<int64_t>(0, num_atoms));
  return 2 * align_up(sizeof(int) * (N + 1)) + align_up(sizeof(int) * (N / kScanBlock + 2));
}

extern "C" int aa_graph_transpose(int64_t num_atoms, int64_t num_edges, const int32_t* rowptr, const int32_t* nbr, int32_t* t_rowptr,
                                  int32_t* t_perm, int32_t* hints3, void* workspace, size_t workspace_bytes, aa_stream stream) {
  AA_REQUIRE(num_atoms >= 0 && num_edges >= 0 && num_edges < (int64_t(1) << 31) && num_atoms < (int64_t(1) << 31), "aa_graph_transpose: size out of range");
  AA_REQUIRE(t_rowptr && (num_edges == 0 || (nbr && t_perm)) && workspace, "aa_graph_transpose: null argument");
  AA_REQUIRE(workspace_bytes >= aa_graph_transpose_workspace_bytes(num_atoms), "aa_graph_transpose: workspace too small");
  AA_REQUIRE(!hints3 || rowptr, "aa_graph_transpose: the hints need the row pointers");
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int64_t N = num_atoms, E = num_edges;
  char* w = static_cast<char*>(workspace);
  int* cnt = reinterpret_cast<int*>(w);
  int* cursor = reinterpret_cast<int*>(w + align_up(sizeof(int) * (N + 1)));
  int* scan_tmp = reinterpret_cast<int*>(w + 2 * align_up(sizeof(int) * (N + 1)));
  AA_CHECK_HIP(hipMemsetAsync(w, 0, 2 * align_up(sizeof(int) * (N + 1)), s));
  if (E > 0) hipLaunchKernelGGL(gt_count_kernel, dim3((unsigned)((E + 255) / 256)), dim3(256), 0, s, nbr, E, N, cnt);
  launch_scan(cnt, t_rowptr, N, scan_tmp, s);
  if (E > 0) {
    hipLaunchKernelGGL(gt_place_kernel, dim3((unsigned)((E + 255) / 256)), dim3(256), 0, s, nbr, E, N, t_rowptr, cursor, t_perm);
    hipLaunchKernelGGL(gt_sort_kernel, dim3((unsigned)((N + 3) / 4)), dim3(256), 0, s, N, t_rowptr, t_perm);
  }
  if (hints3) hipLaunchKernelGGL(gt_hints_kernel, dim3(1), dim3(256), 3 * 256 * sizeof(int), s, N, rowptr, hints3);
  AA_CHECK_HIP(hipGetLastError());
  return AA_OK;
}

namespace aa {
namespace {
__device__ __forceinline__ unsigned long long fp_mix(unsigned long long x) {  // splitmix64 finaliser
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
// fp[0] += sum_k mix(center_k, nbr_k, k), fp[1] += sum_i mix(type_i, i): integer sums, hence independent of the order the
// workgroups arrive in (bit-reproducible), yet every term depends on its position, so a permutation changes the result
template <typename TT>
__global__ __launch_bounds__(256) void graph_fingerprint_kernel(const int64_t* e0, const int64_t* e1, int64_t E, const TT* types, int64_t N,
                                                                unsigned long long* fp) {
  unsigned long long h0 = 0, h1 = 0;
  const int64_t stride = int64_t(gridDim.x) * 256;
  for (int64_t k = int64_t(blockIdx.x) * 256 + threadIdx.x; k < E; k += stride)
    h0 += fp_mix(fp_mix((unsigned long long)e0[k] * 0xD6E8FEB86659FD93ull + (unsigned long long)e1[k]) ^ (unsigned long long)k);
  for (int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x; i < N; i += stride)
    h1 += fp_mix(((unsigned long long)(int64_t)types[i] << 40) ^ (unsigned long long)i);
  for (int m = 32; m >= 1; m >>= 1) {
    h0 += __shfl_xor(h0, m);
    h1 += __shfl_xor(h1, m);
  }
  if ((threadIdx.x & 63) == 0) {
    if (h0) atomicAdd(&fp[0], h0);
    if (h1) atomicAdd(&fp[1], h1);
  }
}
}  // namespace
}  // namespace aa

extern "C" int aa_graph_fingerprint(const int64_t* edge_index, int64_t row_stride, int64_t num_edges, const void* atom_types,
                                    int types_are_int64, int64_t num_atoms, uint64_t* fp2, aa_stream stream) {
  AA_REQUIRE(fp2 && num_edges >= 0 && num_atoms >= 0 && (num_edges == 0 || edge_index) && (num_atoms == 0 || atom_types),
             "aa_graph_fingerprint: bad argument");
  hipStream_t s = static_cast<hipStream_t>(stream);
  AA_CHECK_HIP(hipMemsetAsync(fp2, 0, 2 * sizeof(uint64_t), s));
  const int64_t work = std::max(num_edges, num_atoms);
  if (work == 0) return AA_OK;
  const unsigned nb = unsigned(std::min<int64_t>((work + 255) / 256, 2048));
  auto* out = reinterpret_cast<unsigned long long*>(fp2);
  if (types_are_int64)
    hipLaunchKernelGGL(aa::graph_fingerprint_kernel<int64_t>, dim3(nb), dim3(256), 0, s, edge_index, edge_index + row_stride, num_edges,
                       static_cast<const int64_t*>(atom_types), num_atoms, out);
  else
    hipLaunchKernelGGL(aa::graph_fingerprint_kernel<int32_t>, dim3(nb), dim3(256), 0, s, edge_index, edge_index + row_stride, num_edges,
                       static_cast<const int32_t*>(atom_types), num_atoms, out);
  AA_CHECK_HIP(hipGetLastError());
  return AA_OK;
}
