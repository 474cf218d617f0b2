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

    `step()` gathers the local positions from the caller's position array, runs the whole hot path on the local
    arrays (graph, workspace and kernels see n_own + n_ghost atoms), and returns this rank's contribution to the
    GLOBAL force array (`[N,3]`, non-zero on the local atoms only) plus the energies of the owned atoms;
    `energy_forces_local()` below adds the one collective (all-reduce of that array: 12 N bytes, 1.2 MB at C4)."""

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

    @property
    def num_local_atoms(self) -> int:
        return self.n_own + self.n_ghost

    def step(self, model, pos: torch.Tensor):
        """(energies of the owned atoms [n_own], contribution to the global forces [N,3])."""
        e_loc, f_loc = model.energy_forces(pos.index_select(0, self.local_ids), self.graph)
        f_glob = torch.zeros((self.num_atoms_global, 3), dtype=f_loc.dtype, device=f_loc.device)
        f_glob.index_copy_(0, self.local_ids, f_loc)  # local ids are unique: a plain scatter, deterministic
        return e_loc[: self.n_own], f_glob


def energy_forces_local(model, pos: torch.Tensor, shard: LocalShard, group=None):
    """One step on this rank's compact shard + the force all-reduce.  Returns (E_i [N] with the owned block filled
    in and summed over ranks, forces [N,3] summed over ranks)."""
    import torch.distributed as dist

    e_own, f = shard.step(model, pos)
    e = torch.zeros(shard.num_atoms_global, dtype=e_own.dtype, device=e_own.device)
    e[shard.a0:shard.a1] = e_own
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(f, group=group)  # ghost-atom force contributions: RCCL over xGMI with backend "nccl"
        dist.all_reduce(e, group=group)
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
