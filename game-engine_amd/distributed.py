"""Partitioning of the render path across the GPUs of one node, one process per GPU.

The reference is a single-tab, single-GPU program and has nothing to mirror here; the three
workload kinds shard as follows:
  * stateless cinematics: whole frames, frame k -> rank k mod N, no collective;
  * memory/feedback cinematics: horizontal bands; the feedback read is the same pixel of the
    previous frame (game-compiler/src/codegen/memory.rs:38-45 samples at v_uv), so the halo radius
    of the reference semantics is 0 and no rows are exchanged; halo_exchange() exists for spatial
    filters with a radius > 0;
  * particles: contiguous index ranges; every rank integrates its range against all positions and
    the new positions are all-gathered after each step (velocities stay local).
All collectives go through torch.distributed (backend "nccl" = RCCL over xGMI on the GPU box,
"gloo" in CPU tests); tensors may live on either device.
"""
import ctypes

import numpy as np


def frames_for_rank(nframes, rank, world):
    """Global frame indices rendered by `rank` (frame k -> rank k mod world)."""
    return list(range(rank, nframes, world))


def band_for_rank(height, rank, world, align=8):
    """(row0, rows) of the horizontal band owned by `rank`; bands are multiples of `align` rows
    except the last one (the kernel itself works row by row, so any split is valid), and may be
    empty when height < world * align."""
    per = ((height + world - 1) // world + align - 1) // align * align
    row0 = min(height, rank * per)
    return row0, max(0, min(height, row0 + per) - row0)


def particle_range(n, rank, world):
    """(first, count) of the contiguous particle index range owned by `rank`."""
    per = (n + world - 1) // world
    first = min(n, rank * per)
    return first, max(0, min(n, first + per) - first)


def gather_ranges(full, ranges, dist, group=None):
    """Makes `full` (1-D tensor, identical length on every rank) consistent: rank r's range
    ranges[r] = (first, count) is broadcast from r. Equal ranges use one all-gather."""
    world = len(ranges)
    counts = {c for _, c in ranges}
    rank = dist.get_rank(group)
    if len(counts) == 1 and ranges[0][1] * world == full.numel():
        first, count = ranges[rank]
        dist.all_gather_into_tensor(full, full[first:first + count].clone(), group=group)
        return
    for r, (first, count) in enumerate(ranges):
        if count:
            dist.broadcast(full[first:first + count], src=dist.get_global_rank(group, r) if group is not None else r, group=group)


def halo_exchange(band, radius, rank, world, dist, group=None):
    """band: (rows + 2*radius, width, ...) tensor whose first and last `radius` rows are halo
    slots. Sends the first/last `radius` owned rows to the neighbours above/below and receives
    theirs (edge ranks keep their outer halo untouched). radius 0 is a no-op."""
    if radius == 0 or world == 1:
        return
    ops = []
    top_own = band[radius:2 * radius].contiguous()
    bot_own = band[-2 * radius:-radius].contiguous()
    top_halo = band[:radius]
    bot_halo = band[-radius:]
    recv_top = top_halo.clone()
    recv_bot = bot_halo.clone()
    if rank > 0:
        ops.append(dist.P2POp(dist.isend, top_own, rank - 1, group))
        ops.append(dist.P2POp(dist.irecv, recv_top, rank - 1, group))
    if rank < world - 1:
        ops.append(dist.P2POp(dist.isend, bot_own, rank + 1, group))
        ops.append(dist.P2POp(dist.irecv, recv_bot, rank + 1, group))
    for req in dist.batch_isend_irecv(ops):
        req.wait()
    if rank > 0:
        top_halo.copy_(recv_top)
    if rank < world - 1:
        bot_halo.copy_(recv_bot)


class _DevArray:
    """Exposes a raw device pointer through __cuda_array_interface__ so torch can wrap it."""

    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f4", "data": (int(ptr), False), "version": 2}


class PositionExchange:
    """All-gather of particle positions between steps, in place on the renderer's device arrays."""

    def __init__(self, renderer, n, first, count, torch):
        import torch.distributed as dist
        self.r, self.n, self.torch, self.dist = renderer, n, torch, dist
        world = dist.get_world_size()
        self.ranges = [particle_range(n, k, world) for k in range(world)]
        assert self.ranges[dist.get_rank()] == (first, count)

    def exchange(self):
        self.r.sync()                                  # the step ran on the renderer's own stream
        x_ptr, y_ptr = self.r.particles_positions()    # buffers the next step will read
        for ptr in (x_ptr, y_ptr):
            t = self.torch.as_tensor(_DevArray(ptr, self.n), device="cuda")
            gather_ranges(t, self.ranges, self.dist)
        self.torch.cuda.current_stream().synchronize()
