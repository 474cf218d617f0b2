"""Multi-GPU decomposition of the hot path: one process per GPU, atom-block sharding, one collective.

Strict locality (tests/model/test_allegro.py:68-70 of the reference): everything the
network computes for center atom i depends only on edges (i, .).  With center-sorted edges, rank r owns
a contiguous block of center atoms [a_r, a_{r+1}) and exactly the CSR edge range of those centers; the
forward needs NO communication.  The reverse pass deposits dE/dr_ij on both atoms of an edge, so
neighbor ("ghost") atoms owned by other ranks receive contributions: one all-reduce (RCCL over xGMI
when the backend is "nccl") of the [N,3] force array per step.  The reference has no collective at all
(SURVEY.md §2.3); its external analogue is LAMMPS reverse communication in pair_allegro.

Two layouts of a rank's share: `LocalShard` (compact local numbering: owned block + ghost atoms, everything the
rank holds is O(local); what bench.py --gpus N runs) and `local_graph` (global numbering with the owned-range hint;
kept for callers that already hold global-size arrays).
"""
from typing import List, Optional, Tuple

import numpy as np
import torch

from .nn import PreparedGraph


def partition_atoms(rowptr: np.ndarray, nparts: int) -> List[int]:
    """Cut points of contiguous atom blocks with (nearly) equal edge counts."""
    E = int(rowptr[-1])
    cuts = [0]
    for r in range(1, nparts):
        cuts.append(int(np.searchsorted(rowptr, E * r / nparts, side="left")))
    cuts.append(len(rowptr) - 1)
    return cuts


def local_graph(edge_index: np.ndarray, types: np.ndarray, num_atoms: int, shift_vec: Optional[np.ndarray],
                rank: int, world: int, device, dtype) -> Tuple[PreparedGraph, Tuple[int, int]]:
    """The rank's share of a center-sorted edge list (global atom numbering is kept)."""
    counts = np.bincount(edge_index[0], minlength=num_atoms)
    rowptr = np.zeros(num_atoms + 1, dtype=np.int64)
    np.cumsum(counts, out=rowptr[1:])
    cuts = partition_atoms(rowptr, world)
    a0, a1 = cuts[rank], cuts[rank + 1]
    e0, e1 = int(rowptr[a0]), int(rowptr[a1])
    ei = torch.tensor(edge_index[:, e0:e1], device=device)
    sv = None if shift_vec is None else torch.tensor(shift_vec[e0:e1], dtype=dtype, device=device)
    g = PreparedGraph(ei, torch.tensor(types, device=device), num_atoms, sv)
    return g, (a0, a1)


class LocalShard:
    """One rank's share of a center-sorted edge list in COMPACT local numbering -- everything the rank holds and
    computes is O(owned atoms + ghosts), not O(N):

      local atoms  = the owned block [a0, a1) (local ids 0 .. n_own-1, same order) followed by the ghost atoms: every
                     neighbor of an owned atom that another rank owns (sorted by global id) -- the ghost-atom concept
                     of the reference's `pair_allegro` layout (allegro/_compile.py:28-63) and of LAMMPS itself;
      local edges  = the CSR range of the owned centers with both endpoints renumbered; periodic shift vectors kept.

    `step()` gathers the local positions from the caller's position array into a persistent buffer, runs the whole hot
    path on the local arrays (graph, workspace and kernels see n_own + n_ghost atoms), and deposits this rank's
    contribution to the GLOBAL force array (`[N,3]`, non-zero on the local atoms only) and the energies of the owned
    atoms in ONE persistent buffer; `energy_forces_local()` below adds the one collective of the step (all-reduce of
    that buffer: 16 N bytes, 1.56 MB at C4).  Nothing is allocated per step."""

    def __init__(self, edge_index: np.ndarray, types: np.ndarray, num_atoms: int, shift_vec: Optional[np.ndarray],
                 rank: int, world: int, device, dtype, rowptr: Optional[np.ndarray] = None):
        if rowptr is None:
            rowptr = np.zeros(num_atoms + 1, dtype=np.int64)
            np.cumsum(np.bincount(edge_index[0], minlength=num_atoms), out=rowptr[1:])
        cuts = partition_atoms(rowptr, world)
        self.a0, self.a1 = cuts[rank], cuts[rank + 1]
        e0, e1 = int(rowptr[self.a0]), int(rowptr[self.a1])
        center, nbr = edge_index[0, e0:e1], edge_index[1, e0:e1]
        own = np.arange(self.a0, self.a1)
        outside = (nbr < self.a0) | (nbr >= self.a1)
        ghosts = np.unique(nbr[outside])
        local_ids = np.concatenate([own, ghosts])
        lookup = np.full(num_atoms, -1, dtype=np.int64)
        lookup[local_ids] = np.arange(local_ids.size)
        self.num_atoms_global, self.n_own, self.n_ghost = int(num_atoms), int(own.size), int(ghosts.size)
        self.local_ids = torch.tensor(local_ids, device=device)
        ei_local = torch.tensor(np.stack([center - self.a0, lookup[nbr]]), device=device)
        sv = None if shift_vec is None else torch.tensor(shift_vec[e0:e1], dtype=dtype, device=device)
        self.graph = PreparedGraph(ei_local, torch.tensor(types[local_ids], device=device), int(local_ids.size), sv)
        self._reduce = self._pos_local = None  # persistent step buffers (see _buffers)

    @property
    def num_local_atoms(self) -> int:
        return self.n_own + self.n_ghost

    def _buffers(self, pos: torch.Tensor):
        """Persistent per-shard buffers (allocated on the first step, reused afterwards -- no allocation in the MD loop):
        the gathered local positions and the ONE array the ranks all-reduce, `[N + ceil(N/3), 3]`: rows 0..N-1 are this
        rank's contribution to the global forces, the tail rows carry the per-atom energies (flattened: element
        3N + i = E_i), so energies ride in the same collective as the forces."""
        N = self.num_atoms_global
        if self._reduce is None or self._reduce.dtype != pos.dtype or self._reduce.device != pos.device:
            self._reduce = torch.zeros((N + (N + 2) // 3, 3), dtype=pos.dtype, device=pos.device)
            self._pos_local = torch.empty((self.num_local_atoms, 3), dtype=pos.dtype, device=pos.device)
        return self._reduce, self._pos_local

    def step(self, model, pos: torch.Tensor):
        """Runs the hot path on the local arrays and deposits the result in the shard's reduce buffer.  Returns
        (E_i [N] view: this rank's owned block filled in, zero elsewhere; forces [N,3] view: this rank's contribution,
        non-zero on its local atoms only).  Both are views of ONE buffer: `energy_forces_local` all-reduces it once."""
        buf, pos_local = self._buffers(pos)
        N = self.num_atoms_global
        torch.index_select(pos, 0, self.local_ids, out=pos_local)
        e_loc, f_loc = model.energy_forces(pos_local, self.graph)
        buf.zero_()  # (the previous step's all-reduce left every rank's rows in it)
        buf.index_copy_(0, self.local_ids, f_loc)  # local ids are unique: a plain scatter, deterministic
        e_all = buf.view(-1)[3 * N: 4 * N]
        e_all[self.a0: self.a1] = e_loc[: self.n_own]
        return e_all, buf[:N]


def energy_forces_local(model, pos: torch.Tensor, shard: LocalShard, group=None):
    """One step on this rank's compact shard + THE collective of the step: one all-reduce (sum) of the shard's reduce
    buffer -- ghost-atom force contributions and the owned-atom energies in one message of 16 N bytes (fp32: 1.56 MB at
    C4; RCCL over xGMI with backend "nccl").  Returns (E_i [N], forces [N,3]) summed over ranks, as views of the
    shard's persistent buffer (valid until its next step).  This is the function `bench.py --gpus N` times and the
    world-size-2/4/8 gloo tests drive (tests/test_dist_gloo.py)."""
    import torch.distributed as dist

    e, f = shard.step(model, pos)
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(shard._reduce, group=group)
    return e, f


def energy_forces_sharded(model, pos: torch.Tensor, graph: PreparedGraph, owned: Tuple[int, int], group=None):
    """One step on this rank's atom block + the force all-reduce.  Returns (E_i of owned atoms placed in a
    zero [N] vector and summed over ranks, forces [N,3] summed over ranks)."""
    import torch.distributed as dist

    e_atom, forces = model.energy_forces(pos, graph)
    a0, a1 = owned
    e_own = torch.zeros_like(e_atom)
    e_own[a0:a1] = e_atom[a0:a1]
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(forces, group=group)
        dist.all_reduce(e_own, group=group)
    return e_own, forces
