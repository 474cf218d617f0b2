"""Multi-GPU decomposition of the hot path: one process per GPU, atom-block sharding, one collective.

Strict locality (tests/model/test_allegro.py:68-70 of the reference): everything the
network computes for center atom i depends only on edges (i, .).  With center-sorted edges, rank r owns
a contiguous block of center atoms [a_r, a_{r+1}) and exactly the CSR edge range of those centers; the
forward needs NO communication.  The reverse pass deposits dE/dr_ij on both atoms of an edge, so
neighbor ("ghost") atoms owned by other ranks receive contributions: one all-reduce (RCCL over xGMI
when the backend is "nccl") of the [N,3] force array per step.  The reference has no collective at all
(SURVEY.md §2.3); its external analogue is LAMMPS reverse communication in pair_allegro.
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
