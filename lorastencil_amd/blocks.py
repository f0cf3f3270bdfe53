"""2-D block decomposition of the 2D shapes over a Py x Px process grid: ghost zones on four sides, the
reference's time-step loop (2d/gpu.cu:544-546) per block.

A block's local array is the same padded layout as everything else -- ``[pad 4][G ghost][own][G ghost][pad 4]`` in BOTH
dimensions (no ghost zone on a side that is the global edge: there the pad IS the global halo and keeps the reference's
halo semantics) -- so the block kernels are the single-GPU kernels (``lora_plan_stepk`` & co. on the local extents) and
ghost cells are ordinary interior cells of the local problem.  As in the slab drivers (slab.py / csrc/slab.cpp) the ghost
zones are ``G = radius x applications-per-launch x E`` deep and refreshed every E launches; a launch sweeps the whole
local grid, and what it computes within ``radius x applications`` cells of a ghost zone's outer rim is garbage that the
next refresh overwrites before it can reach an own cell.

The exchange has two phases so that the corners arrive without diagonal messages: first the column ghost zones (E / W:
``G`` columns of the OWN rows -- strided, packed and unpacked by ``lora_copy_block_f64``), then the row ghost zones (N / S:
``G`` rows over the FULL local width, the column ghosts just received included -- contiguous, sent in place).

Two transports behind one driver (``BlockSet``): without a process group it holds EVERY block of the grid in one process
on one device and moves ghost zones with device copies (the loopback rehearsal the tests compare bit for bit with the
undivided grid); with ``distributed=True`` it holds the one block of this rank (rank = iy * Px + ix) and the ghost zones
travel as ``torch.distributed`` P2P messages -- the packed column strips, then whole padded rows in place (RCCL or gloo).

What this does NOT have (slab.py does): boundary-first overlap, deferred waits, the Dirichlet / periodic options, 1D / 3D.
It exists to answer SURVEY 8f-4 with a tested implementation; DESIGN section 6 says why slabs stay the default.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, Sequence

import numpy as np

from . import _lib, ops
from .slab import next_depth

try:  # torch is plumbing: device memory, streams, torch.distributed
    import torch
    import torch.distributed as dist
except Exception:  # pragma: no cover
    torch = None
    dist = None

HALO = 4    # pad of the 2D layouts, either side, both dimensions
RADIUS = 3  # every 2D shape of the reference has radius 3


def _split(n: int, parts: int, k: int) -> tuple[int, int]:
    """[begin, end) of part k of n cells in `parts` nearly equal parts, every boundary at an even index (16-byte rows)."""
    base = (n // parts) & ~1
    if base < 2:
        raise ValueError(f"cannot cut {n} cells into {parts} blocks of at least two")
    b = k * base
    e = n if k == parts - 1 else (k + 1) * base
    return b, e


@dataclass
class BlockLayout:
    global_dims: tuple[int, int]
    grid: tuple[int, int]
    coords: tuple[int, int]
    ghost: int

    def __post_init__(self):
        (m, n), (py, px), (iy, ix) = self.global_dims, self.grid, self.coords
        self.r0, self.r1 = _split(m, py, iy)
        self.c0, self.c1 = _split(n, px, ix)
        g = self.ghost
        self.gt = g if iy > 0 else 0
        self.gb = g if iy < py - 1 else 0
        self.gl = g if ix > 0 else 0
        self.gr = g if ix < px - 1 else 0
        self.own = (self.r1 - self.r0, self.c1 - self.c0)
        self.local_dims = (self.gt + self.own[0] + self.gb, self.gl + self.own[1] + self.gr)
        self.padded = (self.local_dims[0] + 2 * HALO, self.local_dims[1] + 2 * HALO)

    # padded local indices of the first own row / column
    @property
    def row0(self) -> int:
        return HALO + self.gt

    @property
    def col0(self) -> int:
        return HALO + self.gl


class HipBlockStepper:
    """The HIP engine on one block's local array."""

    def __init__(self, shape, layout: BlockLayout, weights=None, options=None):
        self.plan = ops.Plan(shape, layout.local_dims)
        if weights is not None:
            self.plan.set_weights(weights)
        for k, v in (options or {}).items():
            self.plan.set_option(k, int(v))
        self.stream = int(torch.cuda.current_stream().cuda_stream)

    @property
    def apps_per_launch(self) -> int:
        return self.plan.get_option("steps_per_launch")

    def set_apps(self, apps: int) -> None:
        """the launch depth the driver chose for EVERY block (from the global grid), whatever this block's size resolves"""
        self.plan.set_option("steps_per_launch", int(apps))

    def step(self, src, dst):
        self.plan.step(src.data_ptr(), dst.data_ptr(), stream=self.stream)

    def step2(self, src, dst):
        self.plan.step2(src.data_ptr(), dst.data_ptr(), stream=self.stream)

    def stepk(self, src, dst):
        self.plan.stepk(src.data_ptr(), dst.data_ptr(), stream=self.stream)

    def stepn(self, napps: int, src, dst):
        self.plan.stepn_region(napps, src.data_ptr(), dst.data_ptr(), 0, self.plan.dims[0], stream=self.stream)

    def copy_block(self, dst, dst_off: int, dst_ld: int, src, src_off: int, src_ld: int, rows: int, cols: int):
        """rows x cols elements between two strided arrays (element offsets / leading dimensions): the pack / unpack kernel"""
        _lib.check(_lib.lib().lora_copy_block_f64(dst.data_ptr() + 8 * dst_off, dst_ld, src.data_ptr() + 8 * src_off, src_ld,
                                                 rows, cols, self.stream), "lora_copy_block_f64")


class Block:
    """One block: its two buffers, the halo-ring bookkeeping of the fused path, its launches."""

    def __init__(self, shape, layout: BlockLayout, stepper, device):
        self.shape, self.lay, self.stepper, self.device = shape, layout, stepper, device
        self.buf = [torch.zeros(layout.padded, dtype=torch.float64, device=device) for _ in range(2)]
        self.cur = 0
        self.ring = ["input", "zero"]  # what the pad ring of each physical buffer holds

    def load_global(self, a: np.ndarray) -> None:
        """buffer 0 <- this block's own cells + ghost zones + pads of the global padded input, buffer 1 <- 0"""
        lay = self.lay
        r_lo, c_lo = lay.r0 - lay.gt, lay.c0 - lay.gl  # interior index of local interior cell (0, 0)
        piece = a[r_lo:r_lo + lay.padded[0], c_lo:c_lo + lay.padded[1]]  # (global padded index = interior index + HALO)
        self.buf[0].copy_(torch.from_numpy(np.ascontiguousarray(piece)))
        self.buf[1].zero_()
        self.cur = 0
        self.ring = ["input", "zero"]

    def set_ring(self, b: int, what: str, src: int) -> None:
        """pad ring of physical buffer b: 'input' (copy of buffer src's) or 'zero' -- only its global-edge parts are ever
        read for a valid result; the rest borders ghost cells that are refreshed before they matter"""
        if self.ring[b] == what:
            return
        t, s = self.buf[b], self.buf[src]
        for idx in ((slice(0, HALO),), (slice(-HALO, None),), (slice(None), slice(0, HALO)), (slice(None), slice(-HALO, None))):
            if what == "zero":
                t[idx].zero_()
            else:
                assert self.ring[src] == "input"  # even time levels carry the input's ring, and only they are copied from
                t[idx].copy_(s[idx])
        self.ring[b] = what

    def launch(self, napps: int, apps: int, steps_done: int) -> None:
        src_i, dst_i = self.cur, 1 - self.cur
        fused = napps > 1
        even = steps_done % 2 == 0
        # fused launches need the level-0 ring in both buffers; a single sweep from an even level writes the odd level,
        # whose ring is 0, from an odd level an even one (ring = input)
        self.set_ring(dst_i, "input" if (fused or not even) else "zero", src_i)
        src, dst = self.buf[src_i], self.buf[dst_i]
        if napps == apps and apps > 1:
            self.stepper.stepk(src, dst)
        elif napps == 2:
            self.stepper.step2(src, dst)
        elif napps > 1:
            self.stepper.stepn(napps, src, dst)
        else:
            self.stepper.step(src, dst)
        self.cur = dst_i

    def own_piece(self):
        """(row, column, array): own cells (and the global-edge pads this block holds) of the current buffer and where they
        go in the global padded array"""
        lay = self.lay
        t = self.buf[self.cur].cpu().numpy()
        (m, n), (py, px), (iy, ix) = lay.global_dims, lay.grid, lay.coords
        lr0 = lay.row0 - (HALO if iy == 0 else 0)
        lr1 = lay.row0 + lay.own[0] + (HALO if iy == py - 1 else 0)
        lc0 = lay.col0 - (HALO if ix == 0 else 0)
        lc1 = lay.col0 + lay.own[1] + (HALO if ix == px - 1 else 0)
        gr0 = lay.r0 + HALO - (HALO if iy == 0 else 0)
        gc0 = lay.c0 + HALO - (HALO if ix == 0 else 0)
        return gr0, gc0, np.ascontiguousarray(t[lr0:lr1, lc0:lc1])


class BlockSet:
    """Every block of a Py x Px grid in ONE process on one device; ghost zones move by device copies (loopback)."""

    def __init__(self, shape, global_dims: Sequence[int], grid: Sequence[int], device=None, weights=None, options=None,
                 exchange_every: int = 2, stepper_factory: Callable | None = None, distributed: bool = False, group=None):
        if ops.ndim(shape) != 2:
            raise ValueError("block decomposition is implemented for the 2D shapes")
        self.shape = shape
        self.global_dims = (int(global_dims[0]), int(global_dims[1]))
        self.grid = (int(grid[0]), int(grid[1]))
        self.device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        self.distributed, self.group = bool(distributed), group
        if self.distributed:
            if dist.get_world_size(group) != self.grid[0] * self.grid[1]:
                raise ValueError("one rank per block: world size must be Py x Px")
            self.rank = dist.get_rank(group)
        make = stepper_factory or (lambda lay: HipBlockStepper(shape, lay, weights=weights, options=options))
        # applications per launch as the engine resolves them on the GLOBAL grid (rank-independent, like the slab drivers),
        # then the ghost depth, then the real thing -- with that depth forced on every block's plan
        probe = make(BlockLayout(self.global_dims, (1, 1), (0, 0), 0))
        self.apps = int(probe.apps_per_launch)
        del probe
        self.need = RADIUS * self.apps
        thinnest = min(min(BlockLayout(self.global_dims, self.grid, (iy, ix), 0).own)
                       for iy in range(self.grid[0]) for ix in range(self.grid[1]))
        e = max(1, min(int(exchange_every), thinnest // self.need))
        if thinnest < self.need:
            raise ValueError("blocks are thinner than radius x applications per launch")
        self.exchange_every = e
        self.ghost = self.need * e
        self.blocks = {}
        agree = True
        for iy in range(self.grid[0]):
            for ix in range(self.grid[1]):
                if self.distributed and iy * self.grid[1] + ix != self.rank:
                    continue
                lay = BlockLayout(self.global_dims, self.grid, (iy, ix), self.ghost)
                st = make(lay)
                if hasattr(st, "set_apps"):
                    st.set_apps(self.apps)
                agree = agree and int(st.apps_per_launch) == self.apps
                self.blocks[(iy, ix)] = Block(shape, lay, st, self.device)
        if self.distributed:  # every rank raises together, or none does (a lone raise leaves the others in an exchange)
            dev = self.device if dist.get_backend(group) == "nccl" else torch.device("cpu")
            verdict = torch.tensor([1 if agree else 0], dtype=torch.int32, device=dev)
            dist.all_reduce(verdict, op=dist.ReduceOp.MIN, group=group)
            agree = bool(int(verdict.item()))
        if not agree:
            raise ValueError("a block's plan cannot run the launch depth chosen for the grid")
        self.valid = self.ghost
        self.steps_done = 0
        self.exchanges = 0

    def load(self, a: np.ndarray) -> None:
        for b in self.blocks.values():
            b.load_global(a)
        self.valid = self.ghost  # ghost zones were cut out of the same global array
        self.steps_done = 0

    # ---- ghost exchange (two phases: columns of the own rows, then rows over the full local width) ----------------
    def exchange(self) -> None:
        if self.distributed:
            self._exchange_p2p()
        else:
            self._exchange_loopback()
        self.valid = self.ghost
        self.exchanges += 1

    def _exchange_loopback(self) -> None:
        g = self.ghost
        py, px = self.grid
        for (iy, ix), b in self.blocks.items():  # phase 1: E / W
            if ix + 1 < px:
                e = self.blocks[(iy, ix + 1)]
                rows = b.lay.own[0]
                bt, et = b.buf[b.cur], e.buf[e.cur]
                bld, eld = b.lay.padded[1], e.lay.padded[1]
                # my last g own columns -> its left ghost zone; its first g own columns -> my right ghost zone
                b.stepper.copy_block(et, e.lay.row0 * eld + e.lay.col0 - g, eld, bt, b.lay.row0 * bld + b.lay.col0 + b.lay.own[1] - g, bld, rows, g)
                b.stepper.copy_block(bt, b.lay.row0 * bld + b.lay.col0 + b.lay.own[1], bld, et, e.lay.row0 * eld + e.lay.col0, eld, rows, g)
        for (iy, ix), b in self.blocks.items():  # phase 2: N / S, full local width (interior columns incl. column ghosts)
            if iy + 1 < py:
                s = self.blocks[(iy + 1, ix)]
                bt, st = b.buf[b.cur], s.buf[s.cur]
                w = b.lay.local_dims[1]
                assert w == s.lay.local_dims[1]
                br, sr = b.lay.row0 + b.lay.own[0], s.lay.row0
                st[sr - g:sr, HALO:HALO + w].copy_(bt[br - g:br, HALO:HALO + w])  # my last g own rows -> its top ghost zone
                bt[br:br + g, HALO:HALO + w].copy_(st[sr:sr + g, HALO:HALO + w])  # its first g own rows -> my bottom ghost zone

    def _exchange_p2p(self) -> None:
        """One block per rank.  Phase 1: the E / W strips (own rows x g columns) are packed into contiguous buffers by the
        copy kernel, exchanged, and unpacked into the ghost columns; phase 2: the N / S strips are g whole padded rows --
        contiguous, sent and received in place (their pad columns carry the same global halo on both sides)."""
        g = self.ghost
        py, px = self.grid
        ((iy, ix), b), = self.blocks.items()
        t, lay, ld = b.buf[b.cur], b.lay, b.lay.padded[1]
        rows = lay.own[0]
        if t.is_cuda and dist.get_backend(self.group) != "nccl":
            torch.cuda.current_stream(t.device).synchronize()  # gloo reads device tensors from a CPU thread, unordered
        west = iy * px + ix - 1 if ix > 0 else None
        east = iy * px + ix + 1 if ix + 1 < px else None
        north = (iy - 1) * px + ix if iy > 0 else None
        south = (iy + 1) * px + ix if iy + 1 < py else None
        sends, recvs, ops_l = {}, {}, []
        if not hasattr(self, "_pack"):  # contiguous strips for the column ghost zones, made once
            self._pack = {side: (torch.empty((rows, g), dtype=t.dtype, device=t.device),
                                 torch.empty((rows, g), dtype=t.dtype, device=t.device))
                          for side, peer in (("w", west), ("e", east)) if peer is not None}
        for side, peer, src_col in (("w", west, lay.col0), ("e", east, lay.col0 + lay.own[1] - g)):
            if peer is None:
                continue
            sends[side], recvs[side] = self._pack[side]
            b.stepper.copy_block(sends[side], 0, g, t, lay.row0 * ld + src_col, ld, rows, g)  # pack
        if t.is_cuda and dist.get_backend(self.group) != "nccl" and sends:
            torch.cuda.current_stream(t.device).synchronize()
        # a fixed order per pair (lower rank's view): backends match the k-th send to a peer with its k-th receive from us
        for side, peer in (("w", west), ("e", east)):
            if peer is not None:
                ops_l.append(dist.P2POp(dist.isend, sends[side], peer, group=self.group))
                ops_l.append(dist.P2POp(dist.irecv, recvs[side], peer, group=self.group))
        for w in (dist.batch_isend_irecv(ops_l) if ops_l else []):
            w.wait()
        if west is not None:
            b.stepper.copy_block(t, lay.row0 * ld + lay.col0 - g, ld, recvs["w"], 0, g, rows, g)  # unpack
        if east is not None:
            b.stepper.copy_block(t, lay.row0 * ld + lay.col0 + lay.own[1], ld, recvs["e"], 0, g, rows, g)
        if t.is_cuda and dist.get_backend(self.group) != "nccl":
            torch.cuda.current_stream(t.device).synchronize()
        br = lay.row0 + lay.own[0]
        ops_l = []
        if north is not None:
            ops_l.append(dist.P2POp(dist.isend, t[lay.row0:lay.row0 + g], north, group=self.group))
            ops_l.append(dist.P2POp(dist.irecv, t[lay.row0 - g:lay.row0], north, group=self.group))
        if south is not None:
            ops_l.append(dist.P2POp(dist.isend, t[br - g:br], south, group=self.group))
            ops_l.append(dist.P2POp(dist.irecv, t[br:br + g], south, group=self.group))
        for w in (dist.batch_isend_irecv(ops_l) if ops_l else []):
            w.wait()

    def _launch(self, napps: int) -> None:
        need = RADIUS * napps
        if self.valid < need:
            self.exchange()
        for b in self.blocks.values():
            b.launch(napps, self.apps, self.steps_done)
        self.valid -= need
        self.steps_done += napps

    def run(self, times: int) -> None:
        """`times` applications: launches of the engine's depth at even time levels, then the tail launches and single
        sweeps of the slab drivers' rule (slab.next_depth)"""
        t = 0
        tails = (2, 4) if all(hasattr(b.stepper, "stepn") for b in self.blocks.values()) else (2,)
        while t < times:
            d = next_depth(2, self.apps, times - t, self.steps_done % 2 == 0, True, tails)
            self._launch(d)
            t += d

    def store(self, out: np.ndarray | None = None, dst_rank: int = 0):
        """the global padded result (own cells of every block, global-edge pads from the rim blocks); distributed: on
        ``dst_rank`` only (None elsewhere) -- every rank contributes the piece it owns, not a global-sized array"""
        if self.device.type == "cuda":
            torch.cuda.synchronize(self.device)
        pieces = [b.own_piece() for b in self.blocks.values()]
        if self.distributed:
            gathered = [None] * dist.get_world_size(self.group) if self.rank == dst_rank else None
            dist.gather_object(pieces, gathered, dst=dst_rank, group=self.group)
            if self.rank != dst_rank:
                return None
            pieces = [p for rank_pieces in gathered for p in rank_pieces]
        full = out if out is not None else np.zeros((self.global_dims[0] + 2 * HALO, self.global_dims[1] + 2 * HALO))
        for r0, c0, a in pieces:
            full[r0:r0 + a.shape[0], c0:c0 + a.shape[1]] = a
        return full
