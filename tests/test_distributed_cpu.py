"""The N > 1 paths on CPU: two gloo ranks, the oracle standing in for the GPU kernels. Checks the
partition rules and the exchange steps (band stitching, halo rows, particle all-gather)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import C2B_NOISE, C4_TRAILS, C5_GRAVITY, ROOT, Oracle, load_product


def test_partition_rules(game):
    from game_engine_amd import distributed as D
    assert D.frames_for_rank(10, 1, 4) == [1, 5, 9]
    assert sorted(sum((D.frames_for_rank(1000, r, 8) for r in range(8)), [])) == list(range(1000))
    for h, world in ((4320, 8), (2160, 4), (117, 2), (10, 4), (8, 8), (4321, 3)):
        bands = [D.band_for_rank(h, r, world) for r in range(world)]
        assert sum(n for _, n in bands) == h
        pos = 0
        for r0, n in bands:
            assert r0 == pos or n == 0
            assert r0 % 8 == 0 or n == 0
            pos += n
    assert D.band_for_rank(4320, 3, 8) == (1632, 544)
    for n, world in ((1000, 3), (16, 8), (5, 8), (1 << 20, 8)):
        rs = [D.particle_range(n, r, world) for r in range(world)]
        assert sum(c for _, c in rs) == n and rs[0][0] == 0
        assert all(rs[i][0] + rs[i][1] == rs[i + 1][0] or rs[i + 1][1] == 0 for i in range(world - 1))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, outdir):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    game = load_product()
    oracle = Oracle()
    from game_engine_amd import distributed as D

    # 1. feedback cinematic, band-sharded, frame-serial: each rank keeps only its band
    cin = game.compile(C4_TRAILS).cinematics[0]
    w, h, n = 40, 22, 5
    rows = cin.uniform_rows([k / 60 for k in range(n)], w, h, audio={"bass": [0.2 * k for k in range(n)]})
    r0, nr = D.band_for_rank(h, rank, world)
    band = oracle.render_sequence(cin.blob, w, h, rows, True, row0=r0, rows=nr, threads=1)
    np.save(os.path.join(outdir, f"band{rank}.npy"), band)

    # 2. stateless cinematic, frame-sharded k mod N
    cin2 = game.compile(C2B_NOISE).cinematics[0]
    rows2 = cin2.uniform_rows([k / 10 for k in range(5)], 24, 14)
    mine = D.frames_for_rank(5, rank, world)
    np.save(os.path.join(outdir, f"frames{rank}.npy"), np.stack([oracle.render(cin2.blob, 24, 14, rows2[k], threads=1) for k in mine]))

    # 3. particles: index ranges + all-gather of positions every step (uneven ranges: 101 / 2)
    cin3 = game.compile(C5_GRAVITY).cinematics[0]
    m = 101
    ranges = [D.particle_range(m, k, world) for k in range(world)]
    first, count = ranges[rank]
    state = oracle.seed_particles(m, seed=7)
    t = 0.0
    for _ in range(4):
        stepped, t2 = oracle.gravity(cin3.blob, state, 1 / 60, 1, sim_time=t, threads=1)   # whole set; keep own rows only
        state[first:first + count] = stepped[first:first + count]
        t = t2
        for col in (0, 1):
            full = torch.from_numpy(np.ascontiguousarray(state[:, col]))
            D.gather_ranges(full, ranges, dist)
            state[:, col] = full.numpy()
    np.save(os.path.join(outdir, f"particles{rank}.npy"), state[:, :2])

    # 4. even ranges take the all-gather path
    full = torch.arange(8, dtype=torch.float32) * (1 if rank == 0 else -1)
    D.gather_ranges(full, [D.particle_range(8, k, world) for k in range(world)], dist)
    assert full.tolist() == [0.0, 1.0, 2.0, 3.0, -4.0, -5.0, -6.0, -7.0]

    # 5. halo rows for a radius-2 spatial filter
    rad, own = 2, 3
    b = torch.full((own + 2 * rad, 4), float(rank + 1))
    b[:rad] = -1.0
    b[-rad:] = -1.0
    D.halo_exchange(b, rad, rank, world, dist)
    if rank == 0:
        assert b[-rad:].eq(2.0).all() and b[:rad].eq(-1.0).all()
    else:
        assert b[:rad].eq(1.0).all() and b[-rad:].eq(-1.0).all()
    D.halo_exchange(b, 0, rank, world, dist)      # radius 0: nothing moves
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding_matches_single_process(tmp_path, game, oracle):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    cin = game.compile(C4_TRAILS).cinematics[0]
    w, h, n = 40, 22, 5
    rows = cin.uniform_rows([k / 60 for k in range(n)], w, h, audio={"bass": [0.2 * k for k in range(n)]})
    want = oracle.render_sequence(cin.blob, w, h, rows, True, threads=1)
    got = np.concatenate([np.load(tmp_path / f"band{r}.npy") for r in range(world)], axis=1)
    assert np.array_equal(got, want)

    cin2 = game.compile(C2B_NOISE).cinematics[0]
    rows2 = cin2.uniform_rows([k / 10 for k in range(5)], 24, 14)
    parts = [np.load(tmp_path / f"frames{r}.npy") for r in range(world)]
    for k in range(5):
        assert np.array_equal(parts[k % world][k // world], oracle.render(cin2.blob, 24, 14, rows2[k], threads=1))

    cin3 = game.compile(C5_GRAVITY).cinematics[0]
    whole, _ = oracle.gravity(cin3.blob, oracle.seed_particles(101, seed=7), 1 / 60, 4, threads=1)
    for r in range(world):
        assert np.array_equal(np.load(tmp_path / f"particles{r}.npy"), whole[:, :2])
