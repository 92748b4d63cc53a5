"""The frame filter (game_hip_set_blur): an LDS-tiled separable binomial blur of every rendered RGBA8 frame, and the
halo rows band sharding needs for it. The reference has no neighbourhood filter, so the checker is the filter's own
integer definition in numpy (np_restatement.blur_rgba8) applied to the oracle's frames."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import C2_FBM, C4_TRAILS, ROOT, Oracle, load_product
from np_restatement import blur_rgba8


def test_blur_definition_basics():
    rng = np.random.default_rng(1)
    f = rng.integers(0, 256, (19, 23, 4), dtype=np.uint8)
    flat = np.full((9, 11, 4), 77, dtype=np.uint8)
    for r in (1, 2, 3, 4):
        assert np.array_equal(blur_rgba8(flat, r), flat)              # weights sum to 4^r: a constant frame is a fixed point
        b = blur_rgba8(f, r)
        assert b.shape == f.shape and b.min() >= f.min() and b.max() <= f.max()
    one = np.zeros((9, 9, 4), dtype=np.uint8)
    one[4, 4] = 255
    assert blur_rgba8(one, 1)[3:6, 3:6, 0].tolist() == [[16, 32, 16], [32, 64, 32], [16, 32, 16]]   # (1 2 1)/4 twice, rounded half up


def _oracle_blurred_sequence(oracle, cin, w, h, rows, radius):
    frames, prev = [], None
    for r in rows:
        raw = oracle.render(cin.blob, w, h, r, prev=prev if cin.uses_memory else None)
        prev = blur_rgba8(raw, radius)
        frames.append(prev)
    return np.stack(frames)


@pytest.mark.gpu
@pytest.mark.parametrize("radius", [1, 2, 3, 4])
@pytest.mark.parametrize("name,src,w,h", [("fbm", C2_FBM, 203, 77), ("trails", C4_TRAILS, 128, 96), ("tiny", C4_TRAILS, 5, 3)])
def test_gpu_blur_matches_definition(game, oracle, radius, name, src, w, h):
    cin = game.compile(src).cinematics[0]
    rows = cin.uniform_rows([k / 30 for k in range(4)], w, h, audio={"bass": [0.1 * k for k in range(4)]})
    r = cin.renderer(w, h)
    r.set_blur(radius)
    got = r.render(rows)
    r.set_blur(0)                                   # off again: plain frames
    r.reset_memory()
    plain = r.render(rows)
    r.close()
    assert np.array_equal(got, _oracle_blurred_sequence(oracle, cin, w, h, rows, radius))
    assert np.array_equal(plain, oracle.render_sequence(cin.blob, w, h, rows, cin.uses_memory))


@pytest.mark.gpu
@pytest.mark.parametrize("radius,cuts", [(4, [0, 40, 96]), (2, [0, 8, 64, 96]), (3, [0, 33, 96])])
def test_gpu_bands_with_device_side_halo_equal_one_band(game, radius, cuts):
    """Two or three contexts on the one GPU play the ranks of a band-sharded feedback cinematic: every frame each
    renders its raw band, the raw halo rows move device to device between the contexts' raw frames, each blurs its
    band. Stitched, the bands are the single-context frames byte for byte, frame after frame (the blurred frame is the
    next frame's memory, so an error in a halo row would spread)."""
    import torch
    from game_engine_amd.distributed import _DevArray
    cin = game.compile(C4_TRAILS).cinematics[0]
    w, h, n = 160, 96, 6
    rows = cin.uniform_rows([k / 30 for k in range(n)], w, h, audio={"bass": [0.15 * k for k in range(n)]})
    whole = cin.renderer(w, h)
    whole.set_blur(radius)
    want = whole.render(rows)
    whole.close()
    bands = list(zip(cuts[:-1], cuts[1:]))
    ctxs = [cin.renderer(w, h) for _ in bands]
    for c in ctxs:
        c.set_blur(radius)
    raws = [torch.as_tensor(_DevArray(c.raw_rows_ptr(0), (h, w), "<i4"), device="cuda") for c in ctxs]
    got = np.zeros_like(want)
    for k in range(n):
        for c, (a, b) in zip(ctxs, bands):
            c.render_raw_band(rows[k], a, b - a)
        for c in ctxs:
            c.sync()
        for i, (a, b) in enumerate(bands):           # what exchange_raw_halo does between ranks, as device copies
            if i > 0:
                raws[i][a - radius:a].copy_(raws[i - 1][a - radius:a])
            if i + 1 < len(bands):
                raws[i][b:b + radius].copy_(raws[i + 1][b:b + radius])
        torch.cuda.synchronize()
        for c, (a, b) in zip(ctxs, bands):
            got[k, a:b] = c.blur_band(a, b - a)
    for c in ctxs:
        c.close()
    assert np.array_equal(got, want)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _halo_worker(rank, world, port, outdir):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    game = load_product()
    oracle = Oracle()
    D = game.distributed
    cin = game.compile(C4_TRAILS).cinematics[0]
    w, h, n, radius = 48, 37, 4, 3
    rows = cin.uniform_rows([k / 30 for k in range(n)], w, h, audio={"bass": [0.2 * k for k in range(n)]})
    r0, nr = D.band_for_rank(h, rank, world)
    prev_band, out = None, []
    for k in range(n):
        raw = np.zeros((h, w, 4), dtype=np.uint8)
        raw[r0:r0 + nr] = oracle.render(cin.blob, w, h, rows[k], prev=prev_band, row0=r0, rows=nr, threads=1)
        t = torch.from_numpy(raw.view(np.int32).reshape(h, w))
        D.exchange_raw_halo(t, r0, nr, radius, rank, world, dist)
        lo, hi = max(0, r0 - radius), min(h, r0 + nr + radius)
        # blur the window that holds the band and its halo; rows of the window that are not frame edges are cut off again
        win = blur_rgba8(np.pad(raw[lo:hi], ((radius if lo > 0 else 0, radius if hi < h else 0), (0, 0), (0, 0)), mode="edge"), radius)
        win = win[(radius if lo > 0 else 0):win.shape[0] - (radius if hi < h else 0)]
        prev_band = win[r0 - lo:r0 - lo + nr]
        out.append(prev_band)
    np.save(os.path.join(outdir, f"blur{rank}.npy"), np.stack(out))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_halo_exchange_matches_single_process(tmp_path, game, oracle):
    import torch.multiprocessing as mp
    world = 2
    mp.spawn(_halo_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    cin = game.compile(C4_TRAILS).cinematics[0]
    w, h, n, radius = 48, 37, 4, 3
    rows = cin.uniform_rows([k / 30 for k in range(n)], w, h, audio={"bass": [0.2 * k for k in range(n)]})
    want = _oracle_blurred_sequence(oracle, cin, w, h, rows, radius)
    got = np.concatenate([np.load(tmp_path / f"blur{r}.npy") for r in range(world)], axis=1)
    assert np.array_equal(got, want)
