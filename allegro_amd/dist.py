"""Multi-GPU decomposition of the hot path: one process per GPU, atom-block sharding, one collective.

Strict locality (tests/model/test_allegro.py:68-70 of the reference): everything the
network computes for center atom i depends only on edges (i, .).  With center-sorted edges, rank r owns
a contiguous block of center atoms [a_r, a_{r+1}) and exactly the CSR edge range of those centers; the
forward needs NO communication.  The reverse pass deposits dE/dr_ij on both atoms of an edge, so
neighbor ("ghost") atoms owned by other ranks receive contributions: one all-reduce (RCCL over xGMI
when the backend is "nccl") of the [N,3] force array per step.  The reference has no collective at all
(SURVEY.md §2.3); its external analogue is LAMMPS reverse communication in pair_allegro.

`HaloShard` + `energy_forces_halo` (below, round 4) are the sharded-integrator form of the same decomposition -- what LAMMPS
itself does around pair_allegro: every rank keeps only ITS atoms' positions, ghost positions arrive by forward
communication and ghost forces leave by reverse communication (two `all_to_all_single` of the ghost rows only, sizes fixed
per neighbour list), energies stay local; nothing O(N) exists on a rank.  `bench.py --gpus N` times that path.

Two earlier layouts of a rank's share: `LocalShard` (compact local numbering: owned block + ghost atoms, everything the
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


# ----------------------------------------------------------------------------------------------------------------------
# Halo exchange: forward / reverse communication of ghost rows only (round 4)
# ----------------------------------------------------------------------------------------------------------------------
class HaloShard:
    """One rank's share of a frame for a SHARDED integrator: the rank holds the positions of its owned atoms only.

      atoms      numbered globally in a slab order (sorted along one lattice direction; `order` maps them back to the
                 caller's numbering); rank r owns the contiguous block [cuts[r], cuts[r+1]);
      local set  = owned block (local ids 0 .. n_own-1) + ghost atoms (every neighbour of an owned atom that another rank
                 owns, sorted by global id -- hence grouped by owner rank, in rank order);
      plan       recv_counts[p]: ghosts owned by rank p (rows this rank RECEIVES positions for and SENDS forces of);
                 send_counts[p] / send_idx: owned atoms that rank p holds as ghosts (rows this rank sends positions of and
                 receives forces for) -- exchanged ONCE when the shard is built (one all_to_all of the counts, one of the ids).

    Per step (`energy_forces_halo`): forward communication of ghost positions, the whole hot path on the local arrays,
    reverse communication of ghost forces -- each ONE `all_to_all_single` of ghost rows (12 B per row; C4 at 8 ranks:
    ~6 700 ghosts = 80 KB per rank instead of the 1.56 MB all-reduce of `LocalShard`) -- and a fixed-order accumulation into
    the owned rows (bit-reproducible).  Energies of the owned atoms stay on the rank.  This mirrors what LAMMPS does around
    `pair_allegro` (ghost layout allegro/_compile.py:28-63; `comm->forward_comm()` / `reverse_comm()`, README.md:45)."""

    def __init__(self, rank, world, cuts, local_gids, edge_index_local, types_local, shift_vec, device, dtype, order=None, group=None,
                 connect: bool = True):
        """`connect=False` (analysis on one GPU: `bench.py --emulate-shard / --shard-sweep`): the shard of rank `rank` of `world`
        without a process group -- no plan is exchanged and `energy_forces_halo` skips both communications (the caller fills the
        ghost positions once with `fill_local_positions`)."""
        import torch.distributed as dist

        self.rank, self.world, self.cuts, self.group = int(rank), int(world), [int(c) for c in cuts], group
        self.a0, self.a1 = self.cuts[rank], self.cuts[rank + 1]
        self.n_own = self.a1 - self.a0
        local_gids = torch.as_tensor(local_gids, dtype=torch.int64)
        self.n_ghost = int(local_gids.numel()) - self.n_own
        self.order = order  # slab order -> caller's atom ids (None: identity)
        self.local_gids = local_gids.to(device)
        ghosts = local_gids[self.n_own:].cpu()
        assert bool((ghosts[1:] > ghosts[:-1]).all()) if ghosts.numel() > 1 else True, "ghosts must be sorted by global id"
        owner = torch.searchsorted(torch.tensor(self.cuts[1:], dtype=torch.int64), ghosts, right=True)
        self.recv_counts = torch.bincount(owner, minlength=world).tolist()
        assert self.recv_counts[rank] == 0
        # tell every owner which of its atoms this rank needs (setup-time collectives)
        self.connected = bool(connect) and world > 1
        self.host_staged = False
        self._device = torch.device(device)
        if self.connected:
            assert dist.is_initialized(), "HaloShard with world > 1 needs an initialised process group"
            # gloo has no device all_to_all: rows are staged through host buffers (tests: two ranks on one GPU); "nccl" = RCCL takes
            # device rows as they are
            self.host_staged = self._device.type == "cuda" and dist.get_backend(group) == "gloo"
            cnt_in = torch.tensor(self.recv_counts, dtype=torch.int64, device=device)
            cnt_out = torch.empty(world, dtype=torch.int64, device=device)
            _all_to_all_rows(cnt_out, cnt_in, None, None, group, self.host_staged)
            send_counts = cnt_out.tolist()
            ids_out = torch.empty(int(sum(send_counts)), dtype=torch.int64, device=device)
            _all_to_all_rows(ids_out, self.local_gids[self.n_own:].contiguous(), send_counts, self.recv_counts, group, self.host_staged)
            self._install_plan(send_counts, ids_out)
        else:
            self._install_plan([0] * world, torch.empty(0, dtype=torch.int64, device=device))
        self.graph = PreparedGraph(edge_index_local.to(device), torch.as_tensor(types_local).to(device), self.n_own + self.n_ghost,
                                   None if shift_vec is None else shift_vec.to(device=device, dtype=dtype))
        self._bufs = None

    def _install_plan(self, send_counts, send_gids: torch.Tensor):
        """The reverse half of the communication plan: `send_counts[p]` owned atoms (global slab ids `send_gids`, grouped by the
        rank p that holds them as ghosts) -- what the setup collectives deliver, or what `InProcessHaloGroup` reads off the other
        shards directly.  Builds the fixed-order accumulation table of the received force rows: for every owned atom that is
        somebody's ghost the (few) rows of the receive buffer that belong to it, padded to the largest multiplicity with a row
        of zeros -- one gather + sum, no atomics, the same bits every run."""
        device = self._device
        self.send_counts = [int(c) for c in send_counts]
        self.send_idx = send_gids.to(device) - self.a0
        ns = int(self.send_idx.numel())
        assert ns == sum(self.send_counts)
        assert ns == 0 or (int(self.send_idx.min()) >= 0 and int(self.send_idx.max()) < self.n_own)
        if ns > 0:
            srt, perm = torch.sort(self.send_idx, stable=True)
            uniq, cnt = torch.unique_consecutive(srt, return_counts=True)
            mult = int(cnt.max())
            start = torch.cumsum(cnt, 0) - cnt
            slot = torch.arange(ns, device=device) - torch.repeat_interleave(start, cnt)
            rows = torch.full((uniq.numel(), mult), ns, dtype=torch.int64, device=device)  # ns = the zero row
            rows[torch.repeat_interleave(torch.arange(uniq.numel(), device=device), cnt), slot] = perm
            self._touched, self._rows = uniq, rows
        else:
            self._touched = self._rows = None
        self._bufs = None

    def owned_ids(self) -> torch.Tensor:
        """Caller's atom ids of the owned block, in the order `energy_forces_halo` expects `pos_own` and returns its results:
        `pos_own = pos_caller[shard.owned_ids()]`, `forces_caller[shard.owned_ids()] = f_own`."""
        ids = torch.arange(self.a0, self.a1, device=self._device)
        return ids if self.order is None else self.order[self.a0:self.a1]

    def pack_forward(self, pos_own: torch.Tensor):
        """Step phase 1: owned positions into the local array, the rows other ranks hold as ghosts into the send buffer.
        Returns (pos_loc [n_own + n_ghost, 3], send rows [ns, 3], receive rows of the reverse communication [ns + 1, 3])."""
        if pos_own.shape[0] != self.n_own:
            raise ValueError(f"pos_own has {pos_own.shape[0]} rows; this shard owns {self.n_own} atoms (slab order: shard.owned_ids())")
        pos_loc, send_f, recv_r = self._buffers(pos_own.dtype, pos_own.device)
        pos_loc[:self.n_own] = pos_own
        if self.send_idx.numel():
            torch.index_select(pos_own, 0, self.send_idx, out=send_f)
        return pos_loc, send_f, recv_r

    def accumulate_reverse(self, f_own: torch.Tensor, recv_r: torch.Tensor):
        """Step phase 3: adds the force rows received for owned atoms (`recv_r[:ns]`, row ns stays zero) in a fixed order
        (touched ids are unique: one add per row, fixed order inside the gathered sum)."""
        if self._rows is not None:
            f_own.index_add_(0, self._touched, recv_r[self._rows].sum(1))
        return f_own

    # -- construction ------------------------------------------------------------------------------------------------
    @classmethod
    def from_graph(cls, edge_index: np.ndarray, types: np.ndarray, num_atoms: int, shift_vec: Optional[np.ndarray], rank: int,
                   world: int, device, dtype, group=None, rowptr: Optional[np.ndarray] = None, connect: bool = True):
        """From a full center-sorted edge list on the host (tests, small systems): blocks of equal edge count."""
        if rowptr is None:
            rowptr = np.zeros(num_atoms + 1, dtype=np.int64)
            np.cumsum(np.bincount(edge_index[0], minlength=num_atoms), out=rowptr[1:])
        cuts = partition_atoms(rowptr, world)
        a0, a1 = cuts[rank], cuts[rank + 1]
        e0, e1 = int(rowptr[a0]), int(rowptr[a1])
        center, nbr = edge_index[0, e0:e1], edge_index[1, e0:e1]
        ghosts = np.unique(nbr[(nbr < a0) | (nbr >= a1)])
        local_gids = np.concatenate([np.arange(a0, a1), ghosts])
        lookup = np.full(num_atoms, -1, dtype=np.int64)
        lookup[local_gids] = np.arange(local_gids.size)
        ei = torch.tensor(np.stack([center - a0, lookup[nbr]]))
        sv = None if shift_vec is None else torch.tensor(shift_vec[e0:e1])
        return cls(rank, world, cuts, local_gids, ei, types[local_gids], sv, device, dtype, group=group, connect=connect)

    @classmethod
    def from_positions(cls, pos_all: torch.Tensor, types_all: torch.Tensor, cell, r_cut: float, rank: int, world: int, group=None,
                       axis: int = 0, lib=None, connect: bool = True):
        """From the frame itself: atoms are put in slab order along lattice direction `axis` (every rank computes the same
        permutation), cut into `world` blocks of equal atom count, and THIS rank builds the neighbour list only of its slab
        plus the halo within r_cut of it (device cell list, `allegro_amd.nn.neighbor_list`) -- no rank ever holds the full
        EDGE list, and everything the rank keeps afterwards is O(local).  The set-up itself is O(N) per rank: `pos_all` [N,3]
        (device, model dtype) and `types_all` are needed in full (as an MD code has them at start-up), and the slab sort
        (`frac`, `argsort`, `rank_of`, `lookup`: a few 8-byte words per atom) runs over all N atoms on every rank -- 97 336
        atoms at C4, i.e. ~5 MB of transient arrays; a domain-decomposed host hands over its slab instead: `from_owned`.  Full periodic
        `cell` (3x3, rows).  Results come back in SLAB order: `shard.owned_ids()` maps them to the caller's numbering."""
        from .nn import neighbor_list

        dev, dtype = pos_all.device, pos_all.dtype
        N = pos_all.shape[0]
        cell_t = torch.as_tensor(cell, dtype=torch.float64)
        frac = torch.linalg.solve(cell_t.T.to(dev), pos_all.double().T).T  # pos = frac @ cell
        fx = frac[:, axis] - torch.floor(frac[:, axis])
        order = torch.argsort(fx, stable=True)
        cuts = [(N * r) // world for r in range(world + 1)]
        a0, a1 = cuts[rank], cuts[rank + 1]
        own = order[a0:a1]
        # halo candidates: within r_cut of the slab along the axis (periodic); the slab's extent from its own atoms
        height = float(torch.linalg.det(cell_t).abs() / torch.linalg.norm(torch.linalg.cross(cell_t[(axis + 1) % 3], cell_t[(axis + 2) % 3])))
        margin = 1.02 * float(r_cut) / height
        lo, hi = float(fx[own].min()), float(fx[own].max())
        d_lo = torch.remainder(lo - fx, 1.0)   # distance below the slab (periodic)
        d_hi = torch.remainder(fx - hi, 1.0)   # distance above
        inside = torch.zeros(N, dtype=torch.bool, device=dev)
        inside[own] = True
        cand = (~inside) & ((d_lo < margin) | (d_hi < margin) | ((fx >= lo) & (fx <= hi)))
        halo = torch.nonzero(cand).reshape(-1)
        sub = torch.cat([own, halo])
        nl = neighbor_list(pos_all.index_select(0, sub).contiguous(), cell_t, True, r_cut, lib=lib)
        n_own = int(own.numel())
        e_own = int(nl.rowptr[n_own])  # edges are sorted by center: the owned centers' edges are a prefix
        c_loc = nl.edge_index[0, :e_own].long()
        n_sub = nl.edge_index[1, :e_own].long()
        # global (slab-order) id of every atom of the subset
        rank_of = torch.empty(N, dtype=torch.int64, device=dev)
        rank_of[order] = torch.arange(N, device=dev)
        gid_sub = rank_of[sub]
        is_ghost = n_sub >= n_own
        ghosts = torch.unique(gid_sub[n_sub[is_ghost]])  # sorted
        local_gids = torch.cat([torch.arange(a0, a1, device=dev), ghosts])
        lookup = torch.full((N,), -1, dtype=torch.int64, device=dev)
        lookup[local_gids] = torch.arange(local_gids.numel(), device=dev)
        ei = torch.stack([c_loc, lookup[gid_sub[n_sub]]])
        types_local = types_all.to(dev)[order[local_gids]]
        return cls(rank, world, cuts, local_gids.cpu(), ei, types_local, nl.shift_vec[:e_own], dev, dtype, order=order, group=group,
                   connect=connect)

    @classmethod
    def from_owned(cls, pos_own: torch.Tensor, types_own: torch.Tensor, cell, r_cut: float, rank: int, world: int, group=None, axis: int = 0,
                   bounds=None, lib=None):
        """From a DOMAIN-DECOMPOSED host: this rank passes only the atoms it owns -- those whose fractional coordinate along lattice
        direction `axis` lies in its slab [bounds[rank], bounds[rank+1]) (default: `world` slabs of equal width) -- and nothing here is
        O(N): the ranks exchange their atom counts (the global numbering is rank-major: id = first id of the owner + index on the owner),
        every rank sends each other rank the (position, type, index) rows of ITS atoms that lie within r_cut of that rank's slab (one
        all_to_all of counts, three of rows), builds the neighbour list of its slab + the received candidates, keeps the candidates its
        atoms actually see as ghosts, and the usual plan exchange follows (`__init__`).  Set-up traffic per rank: the halo candidates
        only.  What `from_positions` does from a replicated frame, this does from slabs: what LAMMPS' domain decomposition hands to
        `pair_allegro`.  `pos_own` / results stay in the rank's own atom order (`owned_ids()` = the global rank-major ids)."""
        import torch.distributed as dist

        from .nn import neighbor_list

        dev, dtype = pos_own.device, pos_own.dtype
        n_own = int(pos_own.shape[0])
        cell_t = torch.as_tensor(cell, dtype=torch.float64)
        if bounds is None:
            bounds = [q / world for q in range(world + 1)]
        bounds = [float(b) for b in bounds]
        if len(bounds) != world + 1 or abs(bounds[0]) > 1e-12 or abs(bounds[-1] - 1.0) > 1e-12 or any(b1 <= b0 for b0, b1 in zip(bounds, bounds[1:])):
            raise ValueError("HaloShard.from_owned: bounds must rise from 0 to 1 in world + 1 steps")
        frac = torch.linalg.solve(cell_t.T.to(dev), pos_own.double().T).T if n_own else torch.zeros((0, 3), dtype=torch.float64, device=dev)
        fx = frac[:, axis] - torch.floor(frac[:, axis])
        # (a rank-local defect must not leave the other ranks blocked in the first collective: the verdict travels WITH the counts --
        #  a rank whose atoms lie outside its slab announces -1 atoms, and every rank raises after the exchange)
        local_ok = not n_own or bool(((fx >= bounds[rank] - 1e-9) & (fx < bounds[rank + 1] + 1e-9)).all())
        multi = world > 1
        if multi:
            assert dist.is_initialized(), "HaloShard.from_owned with world > 1 needs an initialised process group"
        host_staged = multi and dev.type == "cuda" and dist.get_backend(group) == "gloo"
        # global numbering: rank-major
        counts = torch.zeros(world, dtype=torch.int64, device=dev)
        if multi:
            mine = torch.full((world,), n_own if local_ok else -1, dtype=torch.int64, device=dev)
            _all_to_all_rows(counts, mine, None, None, group, host_staged)  # (counts[p] = atoms of rank p)
        else:
            counts[0] = n_own if local_ok else -1
        bad = [p for p, c in enumerate(counts.tolist()) if c < 0]
        if bad:
            raise ValueError(f"HaloShard.from_owned: rank(s) {bad} were handed atoms outside their slab along axis {axis} "
                             f"(this rank: {rank}, slab [{bounds[rank]}, {bounds[rank + 1]}))")
        cuts = [0] + torch.cumsum(counts, 0).tolist()
        # candidates for every other rank: my atoms within r_cut (with a margin) of its slab, periodic along the axis
        height = float(torch.linalg.det(cell_t).abs() / torch.linalg.norm(torch.linalg.cross(cell_t[(axis + 1) % 3], cell_t[(axis + 2) % 3])))
        margin = 1.02 * float(r_cut) / height
        send_ids = []
        for q in range(world):
            if q == rank or n_own == 0:
                send_ids.append(torch.empty(0, dtype=torch.int64, device=dev))
                continue
            lo, hi = bounds[q], bounds[q + 1]
            near = ((fx >= lo) & (fx <= hi)) | (torch.remainder(lo - fx, 1.0) < margin) | (torch.remainder(fx - hi, 1.0) < margin)
            send_ids.append(torch.nonzero(near).reshape(-1))
        n_send = [int(v.numel()) for v in send_ids]
        n_recv = [0] * world
        if multi:
            t_in = torch.tensor(n_send, dtype=torch.int64, device=dev)
            t_out = torch.empty(world, dtype=torch.int64, device=dev)
            _all_to_all_rows(t_out, t_in, None, None, group, host_staged)
            n_recv = t_out.tolist()
        sel = torch.cat(send_ids) if multi else torch.empty(0, dtype=torch.int64, device=dev)
        tot = int(sum(n_recv))
        cand_pos = torch.empty((tot, 3), dtype=dtype, device=dev)
        cand_meta = torch.empty((tot, 2), dtype=torch.int64, device=dev)  # (type, index on the owner)
        if multi:
            _all_to_all_rows(cand_pos, pos_own.index_select(0, sel).contiguous(), n_recv, n_send, group, host_staged)
            meta = torch.stack([types_own.to(dev).long().index_select(0, sel), sel], dim=1).contiguous()
            _all_to_all_rows(cand_meta, meta, n_recv, n_send, group, host_staged)
        owner = torch.repeat_interleave(torch.arange(world, device=dev), torch.tensor(n_recv, dtype=torch.int64, device=dev))
        cand_gid = torch.tensor(cuts[:-1], dtype=torch.int64, device=dev)[owner] + cand_meta[:, 1]  # rising: grouped by owner, rising index
        # this rank's list: centers = own atoms, neighbours among own + candidates
        nl = neighbor_list(torch.cat([pos_own, cand_pos]).contiguous(), cell_t, True, r_cut, lib=lib)
        e_own = int(nl.rowptr[n_own])
        c_loc = nl.edge_index[0, :e_own].long()
        n_sub = nl.edge_index[1, :e_own].long()
        seen = torch.unique(n_sub[n_sub >= n_own])  # sorted: the candidates that are ghosts
        lookup = torch.full((n_own + tot,), -1, dtype=torch.int64, device=dev)
        lookup[:n_own] = torch.arange(n_own, device=dev)
        lookup[seen] = n_own + torch.arange(seen.numel(), device=dev)
        local_gids = torch.cat([torch.arange(cuts[rank], cuts[rank + 1], device=dev), cand_gid[seen - n_own]])
        types_local = torch.cat([types_own.to(dev).long(), cand_meta[seen - n_own, 0]])
        ei = torch.stack([c_loc, lookup[n_sub]])
        return cls(rank, world, cuts, local_gids.cpu(), ei, types_local, nl.shift_vec[:e_own], dev, dtype, group=group, connect=True)

    def fill_local_positions(self, pos_all: torch.Tensor):
        """Owned + ghost positions straight from a full position array in the CALLER's numbering (start-up, or `connect=False`)."""
        gids = self.local_gids if self.order is None else self.order[self.local_gids]
        pos_loc = self._buffers(pos_all.dtype, pos_all.device)[0]
        torch.index_select(pos_all, 0, gids, out=pos_loc)
        return pos_loc

    # -- step --------------------------------------------------------------------------------------------------------
    def _buffers(self, dtype, device):
        if self._bufs is None or self._bufs[0].dtype != dtype:
            ns = int(self.send_idx.numel())
            self._bufs = (torch.empty((self.n_own + self.n_ghost, 3), dtype=dtype, device=device),  # local positions
                          torch.empty((ns, 3), dtype=dtype, device=device),                            # rows sent forward
                          torch.zeros((ns + 1, 3), dtype=dtype, device=device))                        # rows received in reverse (+ zero row)
        return self._bufs


def energy_forces_halo(model, pos_own: torch.Tensor, shard: HaloShard):
    """One step of a sharded MD code on this rank: forward communication (ghost positions), the hot path on the compact local
    arrays, reverse communication (ghost forces).  `pos_own` [n_own,3]: the positions of the rank's OWN atoms (slab order).
    Returns (E_i [n_own], forces [n_own,3]) of the owned atoms, complete (every contribution of every rank included).
    Collectives per step: two `all_to_all_single` (RCCL over xGMI with backend "nccl"), ghost rows only; none with one rank."""
    import torch.distributed as dist

    pos_loc, send_f, recv_r = shard.pack_forward(pos_own)
    n_own = shard.n_own
    multi = shard.connected
    if multi:
        _all_to_all_rows(pos_loc[n_own:], send_f, shard.recv_counts, shard.send_counts, shard.group, shard.host_staged)
    e_loc, f_loc = model.energy_forces(pos_loc, shard.graph)
    f_own = f_loc[:n_own]
    if multi:
        ns = recv_r.shape[0] - 1
        _all_to_all_rows(recv_r[:ns], f_loc[n_own:].contiguous(), shard.send_counts, shard.recv_counts, shard.group, shard.host_staged)
        shard.accumulate_reverse(f_own, recv_r)
    return e_loc[:n_own], f_own


def _all_to_all_rows(out: torch.Tensor, inp: torch.Tensor, out_splits, in_splits, group, host_staged: bool):
    """`dist.all_to_all_single` of rows.  `host_staged` (device tensors over a backend without device all_to_all, i.e. gloo in
    the two-ranks-on-one-GPU test): the rows go through host copies -- `.cpu()` waits for the producing kernels on the current
    stream, `copy_` enqueues the upload on it, so the ordering around the collective is the one the RCCL path has."""
    import torch.distributed as dist

    if not host_staged:
        dist.all_to_all_single(out, inp, out_splits, in_splits, group=group)
        return
    inp_h = inp.cpu()
    out_h = torch.empty(out.shape, dtype=out.dtype)
    dist.all_to_all_single(out_h, inp_h, out_splits, in_splits, group=group)
    out.copy_(out_h)


class InProcessHaloGroup:
    """ALL `world` HaloShards of one frame in ONE process on one device (tests and analysis on a one-GPU box): the same shards,
    buffers, plan tables, pack / accumulate code and kernels as a `world`-rank job; the two communications of a step are slice
    copies between the shards' buffers in rank order, and the plan the setup collectives would deliver is read off the other
    shards' ghost lists.  Not a product path: a real job runs one process per GPU through `energy_forces_halo`."""

    def __init__(self, shards: List[HaloShard]):
        self.shards = shards
        W = len(shards)
        assert all(s.world == W and s.rank == r and not s.connected for r, s in enumerate(shards))
        for p, sp in enumerate(shards):
            counts, ids = [], []
            for q, sq in enumerate(shards):  # rows rank q receives from p = q's ghosts owned by p (sorted by global id)
                gq = sq.local_gids[sq.n_own:]
                mine = gq[(gq >= sp.a0) & (gq < sp.a1)]
                assert int(mine.numel()) == sq.recv_counts[p]
                counts.append(int(mine.numel()))
                ids.append(mine)
            sp._install_plan(counts, torch.cat(ids))

    @classmethod
    def from_positions(cls, pos_all, types_all, cell, r_cut, world, lib=None, axis: int = 0):
        return cls([HaloShard.from_positions(pos_all, types_all, cell, r_cut, r, world, axis=axis, lib=lib, connect=False) for r in range(world)])

    def _exchange(self, outs, inps, forward: bool):
        """out[q] block p <- inp[p] block q; block sizes: forward (positions) p sends send_counts[q] rows, reverse (forces) p sends
        recv_counts[q] rows."""
        W = len(self.shards)
        for q, sq in enumerate(self.shards):
            o = 0
            for p, sp in enumerate(self.shards):
                cnt = sp.send_counts if forward else sp.recv_counts
                n = cnt[q]
                start = sum(cnt[:q])
                outs[q][o:o + n] = inps[p][start:start + n]
                o += n
            assert o == outs[q].shape[0]

    def step(self, model, pos_own_list):
        """One step of every rank: [(E_i [n_own], forces [n_own, 3])] in rank order (complete, reverse communication included)."""
        packed = [s.pack_forward(p) for s, p in zip(self.shards, pos_own_list)]
        self._exchange([pk[0][s.n_own:] for pk, s in zip(packed, self.shards)], [pk[1] for pk in packed], forward=True)
        res = []
        for s, pk in zip(self.shards, packed):
            e_loc, f_loc = model.energy_forces(pk[0], s.graph)
            res.append((e_loc.clone(), f_loc.clone()))  # (the model's output buffers are reused by the next shard's step)
        self._exchange([pk[2][:pk[2].shape[0] - 1] for pk in packed], [f[s.n_own:] for (_, f), s in zip(res, self.shards)], forward=False)
        out = []
        for s, pk, (e_loc, f_loc) in zip(self.shards, packed, res):
            f_own = f_loc[:s.n_own]
            s.accumulate_reverse(f_own, pk[2])
            out.append((e_loc[:s.n_own], f_own))
        return out
