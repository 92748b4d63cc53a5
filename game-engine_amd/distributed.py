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


def gather_ranges(full, ranges, dist, group=None, scratch=None):
    """Makes `full` (1-D tensor, identical length on every rank) consistent: afterwards rank r's range
    ranges[r] = (first, count) holds rank r's values everywhere. One all-gather whatever the split: the ranges of
    particle_range() are `per` long except the last ones, so every rank sends a `per`-long buffer (its values, then
    padding) and the gathered world * per values are the full array followed by padding.
    scratch: optional (send, recv) tensors of per and world * per elements to reuse between calls."""
    world = len(ranges)
    rank = dist.get_rank(group)
    per = max(c for _, c in ranges)
    n = full.numel()
    if any(f != min(n, r * per) for r, (f, _) in enumerate(ranges)) or sum(c for _, c in ranges) != n:
        raise ValueError("gather_ranges: ranges must be the contiguous split of particle_range()")
    first, count = ranges[rank]
    if per * world == n:                       # even split: gather straight into the array
        send = scratch[0] if scratch else full.new_empty(per)
        send.copy_(full[first:first + count])
        dist.all_gather_into_tensor(full, send, group=group)
        return
    send, recv = scratch if scratch else (full.new_zeros(per), full.new_empty(per * world))
    send[:count].copy_(full[first:first + count])
    dist.all_gather_into_tensor(recv, send, group=group)
    full.copy_(recv[:n])


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


def exchange_raw_halo(frame, row0, rows, radius, rank, world, dist, group=None):
    """Halo rows of a band-sharded frame filter. frame: (h, w) tensor holding this rank's rows [row0, row0 + rows) of
    the raw frame (CPU or GPU; the other rows are scratch). Sends the band's first / last `radius` rows to the ranks
    above / below and receives theirs into the rows just outside the band, as one batch of point-to-point operations
    (over xGMI with the nccl backend). Rows beyond the frame's edges are not exchanged: the filter clamps there."""
    if radius == 0 or world == 1:
        return
    if rows < radius:
        raise ValueError("exchange_raw_halo: a band must be at least `radius` rows tall")
    h = frame.shape[0]
    ops, landing = [], []
    if rank > 0 and row0 > 0:
        up = frame[row0:row0 + radius].contiguous()
        recv = frame.new_empty((min(radius, row0),) + tuple(frame.shape[1:]))
        ops += [dist.P2POp(dist.isend, up, rank - 1, group), dist.P2POp(dist.irecv, recv, rank - 1, group)]
        landing.append((row0 - recv.shape[0], recv))
    if rank < world - 1 and row0 + rows < h:
        down = frame[row0 + rows - radius:row0 + rows].contiguous()
        recv = frame.new_empty((min(radius, h - row0 - rows),) + tuple(frame.shape[1:]))
        ops += [dist.P2POp(dist.isend, down, rank + 1, group), dist.P2POp(dist.irecv, recv, rank + 1, group)]
        landing.append((row0 + rows, recv))
    if not ops:
        return
    for req in dist.batch_isend_irecv(ops):
        req.wait()
    for first, recv in landing:
        # the rank above sent its LAST rows, the rank below its FIRST rows: either way they are consecutive frame rows
        frame[first:first + recv.shape[0]].copy_(recv)


class _DevArray:
    """Exposes a raw device pointer through __cuda_array_interface__ so torch can wrap it."""

    def __init__(self, ptr, shape, typestr="<f4"):
        shape = (shape,) if isinstance(shape, int) else tuple(shape)
        self.__cuda_array_interface__ = {"shape": shape, "typestr": typestr, "data": (int(ptr), False), "version": 2}


class BandHaloExchange:
    """Per-frame halo exchange for a band-sharded cinematic rendered with the frame filter (Renderer.set_blur):
    between render_raw_band() and blur_band() every rank trades `radius` raw rows with its neighbours, device to
    device, in place in the renderers' raw frames and queued on the renderer's own stream."""

    def __init__(self, renderer, row0, rows, radius, torch, group=None):
        import torch.distributed as dist
        self.r, self.row0, self.rows, self.radius, self.torch, self.dist, self.group = renderer, row0, rows, radius, torch, dist, group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self.host = dist.get_backend(group) == "gloo"     # CPU rehearsals of the control flow: staged through host memory
        self.stream = torch.cuda.ExternalStream(renderer.stream_ptr())

    def exchange(self):
        torch = self.torch
        with torch.cuda.stream(self.stream):
            frame = torch.as_tensor(_DevArray(self.r.raw_rows_ptr(0), (self.r.h, self.r.w), "<i4"), device="cuda")
            if not self.host:
                exchange_raw_halo(frame, self.row0, self.rows, self.radius, self.rank, self.world, self.dist, self.group)
                return
            lo, hi = max(0, self.row0 - self.radius), min(self.r.h, self.row0 + self.rows + self.radius)
            window = frame[lo:hi].cpu()
            exchange_raw_halo(window, self.row0 - lo, self.rows, self.radius, self.rank, self.world, self.dist, self.group)
            frame[lo:hi].copy_(window)


class PositionExchange:
    """All-gather of particle positions between steps, in place on the renderer's device arrays.

    One collective per step carries x and y together: every rank sends [x of its range | y of its range] (padded to
    the longest range) and unpacks the gathered blocks into both position arrays. Everything is queued on the
    renderer's own HIP stream (wrapped as a torch ExternalStream): the pack waits for the step kernel in stream order,
    the collective is ordered against that stream by the process group's own events, and the next step's kernel waits
    for the unpack - no host synchronisation between steps. finish() waits for the last exchange."""

    def __init__(self, renderer, n, first, count, torch, group=None):
        import torch.distributed as dist
        self.r, self.n, self.torch, self.dist, self.group = renderer, n, torch, dist, group
        self.world = dist.get_world_size(group)
        self.ranges = [particle_range(n, k, self.world) for k in range(self.world)]
        assert self.ranges[dist.get_rank(group)] == (first, count)
        self.first, self.count = first, count
        self.per = max(c for _, c in self.ranges)
        # gloo (CPU rehearsals of the control flow only) moves host tensors: that path waits for the step, stages
        # through host memory and is not meant to be fast
        self.host = dist.get_backend(group) == "gloo"
        self.stream = torch.cuda.ExternalStream(renderer.stream_ptr())
        with torch.cuda.stream(self.stream):
            self.send = torch.zeros(2 * self.per, dtype=torch.float32, device="cuda")
            self.recv = torch.empty(self.world * 2 * self.per, dtype=torch.float32, device="cuda")

    def exchange(self):
        torch = self.torch
        x_ptr, y_ptr = self.r.particles_positions()    # buffers the next step will read
        with torch.cuda.stream(self.stream):
            x = torch.as_tensor(_DevArray(x_ptr, self.n), device="cuda")
            y = torch.as_tensor(_DevArray(y_ptr, self.n), device="cuda")
            f, c, per = self.first, self.count, self.per
            self.send[:c].copy_(x[f:f + c])
            self.send[per:per + c].copy_(y[f:f + c])
            if self.host:
                send, recv = self.send.cpu(), torch.empty(self.recv.shape, dtype=torch.float32)
                self.dist.all_gather_into_tensor(recv, send, group=self.group)
                self.recv.copy_(recv)
            else:
                self.dist.all_gather_into_tensor(self.recv, self.send, group=self.group)
            blocks = self.recv.view(self.world, 2, per)
            x.copy_(blocks[:, 0, :].reshape(-1)[:self.n])
            y.copy_(blocks[:, 1, :].reshape(-1)[:self.n])

    def finish(self):
        self.stream.synchronize()
