"""Synthetic periodic boxes, neighbor lists and the center-sorted CSR edge layout.

Host-side (numpy) preparation of the inputs of the hot path.  The HIP kernels consume a
*center-sorted* directed edge list (SURVEY.md §8d "edges sorted by center, then neighbor"):
`center[E]`, `nbr[E]` int32 plus `rowptr[N+1]`; periodic images are carried either as a
per-edge cartesian shift vector (PBC layout) or as appended ghost atoms (the `pair_allegro`
layout the reference builds in allegro/_compile.py:28-63).
"""
from dataclasses import dataclass
from typing import Optional, Tuple

import numpy as np

SI_LATTICE_A = 5.431  # Angstrom, diamond cubic


@dataclass
class Graph:
    pos: np.ndarray  # [N,3] float64
    types: np.ndarray  # [N] int64
    edge_index: np.ndarray  # [2,E] int64, row0=center, row1=neighbor, sorted by (center, neighbor)
    cell: Optional[np.ndarray] = None  # [3,3] or None (ghost layout)
    cell_shift: Optional[np.ndarray] = None  # [E,3] integer shifts (float64) or None
    n_local: Optional[int] = None  # number of real (non-ghost) atoms

    @property
    def num_atoms(self):
        return self.pos.shape[0]

    @property
    def num_edges(self):
        return self.edge_index.shape[1]

    def shift_vec(self):
        """Cartesian per-edge shift (cell_shift @ cell), or None in the ghost layout."""
        if self.cell_shift is None:
            return None
        return self.cell_shift @ self.cell


def diamond_si(n_cells: int, jitter: float = 0.05, seed: int = 0) -> Tuple[np.ndarray, np.ndarray]:
    """n^3 conventional diamond cells (8 atoms each) with Gaussian jitter; returns (pos, cell)."""
    a = SI_LATTICE_A
    basis = np.array(
        [[0, 0, 0], [0, 0.5, 0.5], [0.5, 0, 0.5], [0.5, 0.5, 0],
         [0.25, 0.25, 0.25], [0.25, 0.75, 0.75], [0.75, 0.25, 0.75], [0.75, 0.75, 0.25]],
        dtype=np.float64,
    )
    g = np.arange(n_cells)
    cells = np.stack(np.meshgrid(g, g, g, indexing="ij"), -1).reshape(-1, 3)
    pos = (cells[:, None, :] + basis[None, :, :]).reshape(-1, 3) * a
    rng = np.random.default_rng(seed)
    pos = pos + rng.normal(0.0, jitter, size=pos.shape)
    cell = np.eye(3) * (a * n_cells)
    return pos, cell


def water_box(n_mol_side: int, box: float, seed: int = 0) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """Rigid H2O on a jittered simple-cubic molecular lattice. types: O=0, H=1."""
    rng = np.random.default_rng(seed)
    r_oh, ang = 0.9572, np.deg2rad(104.52)
    mol = np.array([[0, 0, 0], [r_oh, 0, 0], [r_oh * np.cos(ang), r_oh * np.sin(ang), 0]])
    g = (np.arange(n_mol_side) + 0.5) * (box / n_mol_side)
    centers = np.stack(np.meshgrid(g, g, g, indexing="ij"), -1).reshape(-1, 3)
    centers = centers + rng.normal(0, 0.15, size=centers.shape)
    q = rng.normal(size=(centers.shape[0], 4))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    w, x, y, z = q.T
    rot = np.stack(
        [1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w),
         2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w),
         2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], -1).reshape(-1, 3, 3)
    pos = (centers[:, None, :] + np.einsum("mij,aj->mai", rot, mol)).reshape(-1, 3)
    types = np.tile(np.array([0, 1, 1]), centers.shape[0])
    return pos, np.eye(3) * box, types


def neighbor_list_pbc(pos: np.ndarray, cell: np.ndarray, r_cut: float):
    """Directed edges (i->j and j->i) under orthorhombic PBC, minimum image (requires box > 2 r_cut).

    Returns edge_index [2,E] (sorted by center then neighbor) and integer cell_shift [E,3] such that
    r_ij = pos[j] - pos[i] + cell_shift @ cell.
    """
    from scipy.spatial import cKDTree

    box = np.diag(cell).copy()
    assert np.allclose(cell, np.diag(box)), "orthorhombic cells only"
    assert np.all(box > 2 * r_cut), "minimum-image neighbor list needs box > 2 r_cut"
    wrapped = np.mod(pos, box)
    wrap_shift = np.round((wrapped - pos) / box)  # integer image offset applied to each atom
    tree = cKDTree(wrapped, boxsize=box)
    pairs = tree.query_pairs(r_cut, output_type="ndarray")
    i = np.concatenate([pairs[:, 0], pairs[:, 1]])
    j = np.concatenate([pairs[:, 1], pairs[:, 0]])
    d = wrapped[j] - wrapped[i]
    s = -np.round(d / box)  # minimum image shift for wrapped coords
    # r_ij = wrapped[j]-wrapped[i] + s*box = pos[j]-pos[i] + (wrap_shift[j]-wrap_shift[i]+s)*box
    shift = s + wrap_shift[j] - wrap_shift[i]
    order = np.lexsort((j, i))
    ei = np.stack([i[order], j[order]]).astype(np.int64)
    return ei, shift[order].astype(np.float64)


def make_si_graph(n_cells: int, r_cut: float = 5.0, jitter: float = 0.05, seed: int = 0) -> Graph:
    pos, cell = diamond_si(n_cells, jitter, seed)
    ei, shift = neighbor_list_pbc(pos, cell, r_cut)
    return Graph(pos=pos, types=np.zeros(len(pos), dtype=np.int64), edge_index=ei, cell=cell,
                 cell_shift=shift, n_local=len(pos))


def make_water_graph(n_mol_side: int, box: float, r_cut: float = 5.0, seed: int = 0) -> Graph:
    pos, cell, types = water_box(n_mol_side, box, seed)
    ei, shift = neighbor_list_pbc(pos, cell, r_cut)
    return Graph(pos=pos, types=types.astype(np.int64), edge_index=ei, cell=cell, cell_shift=shift,
                 n_local=len(pos))


def to_ghost_layout(g: Graph) -> Graph:
    """PBC graph -> ghost-atom graph with the same tensor contract as the reference's `pair_allegro`
    target (allegro/_compile.py:28-63): one ghost per outside-cell edge (not deduplicated), ghosts
    appended after the real atoms, no cell / shifts.  Edges are then re-sorted by center."""
    assert g.cell_shift is not None
    outside = np.abs(g.cell_shift).sum(-1) != 0
    n = g.num_atoms
    ghost_src = g.edge_index[1][outside]
    ghost_pos = g.pos[ghost_src] + g.cell_shift[outside] @ g.cell
    ei = g.edge_index.copy()
    ei[1][outside] = np.arange(n, n + outside.sum())
    order = np.lexsort((ei[1], ei[0]))
    return Graph(pos=np.concatenate([g.pos, ghost_pos]), types=np.concatenate([g.types, g.types[ghost_src]]),
                 edge_index=ei[:, order], cell=None, cell_shift=None, n_local=n)


def csr_from_sorted_centers(center: np.ndarray, num_atoms: int) -> np.ndarray:
    """rowptr[N+1] of a center-sorted edge list."""
    counts = np.bincount(center, minlength=num_atoms)
    rowptr = np.zeros(num_atoms + 1, dtype=np.int64)
    np.cumsum(counts, out=rowptr[1:])
    return rowptr


def is_center_sorted(center: np.ndarray) -> bool:
    return bool(np.all(center[1:] >= center[:-1]))
