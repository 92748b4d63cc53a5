"""GPU tests of the pieces the multi-GPU particle path is made of, within one GPU's limits: the position exchange
running end to end over RCCL with one rank, stream ordering of particle uploads against queued steps."""
import os
import socket

import numpy as np
import pytest

from conftest import C5_GRAVITY

pytestmark = pytest.mark.gpu


def test_reseed_after_queued_steps_is_ordered(game, oracle):
    """particles_step only queues work; re-initialising the same particle count right after it must not let the
    uploads overtake the kernels still running (the arrays are reused in place)."""
    cin = game.compile(C5_GRAVITY).cinematics[0]
    n = 32768                                   # one step runs for many milliseconds: long enough to overtake
    r = cin.renderer(16, 16)
    r.particles_seed(n)
    r.particles_step(1 / 60, 4)                 # queued, not waited for
    m = 2048
    p0 = oracle.seed_particles(m, seed=5)
    big = oracle.seed_particles(n, seed=5)
    r.particles_init(big[:, :2], big[:, 2:])    # same n: no reallocation, the arrays are overwritten in place
    r.particles_step(1 / 60, 1)
    pos, vel = r.particles_read()
    r.close()
    r2 = cin.renderer(16, 16)                   # reference run: a fresh context that never overlapped anything
    r2.particles_init(big[:, :2], big[:, 2:])
    r2.particles_step(1 / 60, 1)
    pos2, vel2 = r2.particles_read()
    r2.close()
    assert np.array_equal(pos, pos2) and np.array_equal(vel, vel2)
    # and the small case against the oracle, through the same sequence of calls
    r3 = cin.renderer(16, 16)
    r3.particles_seed(m)
    r3.particles_step(1 / 60, 50)
    r3.particles_init(p0[:, :2], p0[:, 2:])
    r3.particles_step(1 / 60, 2)
    pos3, vel3 = r3.particles_read()
    r3.close()
    want, _ = oracle.gravity(cin.blob, p0, 1 / 60, 2)
    assert np.array_equal(pos3, want[:, :2]) and np.array_equal(vel3, want[:, 2:])


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_position_exchange_over_rccl_single_rank(game, oracle):
    """PositionExchange end to end on the GPU with the nccl (= RCCL) backend and one rank: the renderer's hipMalloc'd
    position arrays wrapped as torch tensors, packed, all-gathered on the renderer's own stream and unpacked in place.
    With one rank the exchange must leave the simulation exactly where a run without it is (and equal to the oracle)."""
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", WORLD_SIZE="1")
    torch.cuda.set_device(0)
    dist.init_process_group(backend="nccl", device_id=torch.device("cuda", 0))
    try:
        cin = game.compile(C5_GRAVITY).cinematics[0]
        n = 3001                                # not a multiple of anything
        p0 = oracle.seed_particles(n, seed=11)
        r = cin.renderer(16, 16)
        r.particles_init(p0[:, :2], p0[:, 2:])
        ex = game.distributed.PositionExchange(r, n, 0, n, torch)
        for _ in range(5):
            r.particles_step(1 / 60, 1, first=0, count=n)
            ex.exchange()                       # nothing waits on the host between steps
        ex.finish()
        pos, vel = r.particles_read()
        r.close()
        want, _ = oracle.gravity(cin.blob, p0, 1 / 60, 5)
        assert np.array_equal(pos, want[:, :2]) and np.array_equal(vel, want[:, 2:])
        # the generic helper on device tensors: even and uneven (padded) splits degenerate to a copy with one rank
        full = torch.arange(10, dtype=torch.float32, device="cuda")
        game.distributed.gather_ranges(full, [(0, 10)], dist)
        assert full.tolist() == list(range(10))
    finally:
        dist.destroy_process_group()


def test_shutdown_releases_tables_and_later_contexts_rebuild_them(game, oracle):
    from conftest import C2B_NOISE
    cin = game.compile(C2B_NOISE).cinematics[0]        # uses the lattice, feature-point and simplex tables
    w, h = 96, 54
    rows = cin.uniform_rows([0.25], w, h)
    want = oracle.render(cin.blob, w, h, rows[0])
    for _ in range(2):
        r = cin.renderer(w, h)
        assert np.array_equal(r.render(rows)[0], want)
        r.close()
        game.shutdown()
    game.shutdown()                                    # nothing left: still fine
    # while a context exists the tables stay: its kernels read them through pointers it holds
    r = cin.renderer(w, h)
    with pytest.raises(game.GameError, match="still exist"):
        game.shutdown()
    assert np.array_equal(r.render(rows)[0], want)
    r.close()
    game.shutdown()
