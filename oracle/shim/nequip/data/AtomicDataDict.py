"""ORACLE SHIM: key names + helpers of `nequip.data.AtomicDataDict` used by the reference."""
from typing import Dict

import torch

Type = Dict[str, torch.Tensor]

POSITIONS_KEY = "pos"
EDGE_INDEX_KEY = "edge_index"
ATOM_TYPE_KEY = "atom_types"
EDGE_TYPE_KEY = "edge_type"
EDGE_VECTORS_KEY = "edge_vectors"
EDGE_LENGTH_KEY = "edge_lengths"
NORM_LENGTH_KEY = "normed_edge_lengths"
EDGE_ATTRS_KEY = "edge_attrs"
EDGE_FEATURES_KEY = "edge_features"
EDGE_EMBEDDING_KEY = "edge_embedding"
EDGE_ENERGY_KEY = "edge_energy"
EDGE_CUTOFF_KEY = "edge_cutoff"
PER_ATOM_ENERGY_KEY = "atomic_energy"
TOTAL_ENERGY_KEY = "total_energy"
FORCE_KEY = "forces"
CELL_KEY = "cell"
EDGE_CELL_SHIFT_KEY = "edge_cell_shift"
BATCH_KEY = "batch"
NUM_NODES_KEY = "num_atoms"


def num_nodes(data: Type) -> int:
    return data[POSITIONS_KEY].shape[0]


def num_frames(data: Type) -> int:
    if NUM_NODES_KEY in data:
        return data[NUM_NODES_KEY].shape[0]
    return 1


def with_edge_vectors_(data: Type, with_lengths: bool = True) -> Type:
    """r_ij = pos[j] - pos[i] (+ shift @ cell); center = edge_index[0], neighbor = edge_index[1]
    (direction corroborated by allegro/_compile.py:41-43)."""
    if EDGE_VECTORS_KEY not in data:
        pos = data[POSITIONS_KEY]
        ei = data[EDGE_INDEX_KEY]
        vec = torch.index_select(pos, 0, ei[1]) - torch.index_select(pos, 0, ei[0])
        if CELL_KEY in data and EDGE_CELL_SHIFT_KEY in data:
            cell = data[CELL_KEY].view(3, 3)
            vec = vec + torch.mm(data[EDGE_CELL_SHIFT_KEY].to(vec.dtype), cell)
        data[EDGE_VECTORS_KEY] = vec
    if with_lengths and EDGE_LENGTH_KEY not in data:
        data[EDGE_LENGTH_KEY] = torch.linalg.norm(data[EDGE_VECTORS_KEY], dim=-1)
    return data
