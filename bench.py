#!/usr/bin/env python3
"""Benchmark of the GPU render path.

    python bench.py --gpus N --steps K --warmup W [--config c2]

Default workload (config c2) is the benchmark the metric is quoted on: the fbm showcase cinematic
`domain_warp(0.1, 3.0) | fbm(3.0, octaves: 6, ...) | glow(1.5) | tint(...)` at 3840x2160, rendered
as synthetic frame batches with u_time = k/60. One step = one batch of --batch frames rendered
into the HBM frame ring (no read-back inside the timed region). With N > 1 (launched by
torch.distributed.run, one rank per GPU) frames are sharded k mod N with no collective on the data
path; per-GPU work is fixed, so scaling is weak. Rank 0 prints one JSON line.

Other configs (c1, c2b, c3, c4, c5) exercise the remaining benchmark cinematics; c4 shards
horizontal bands, c5 shards particle indices with an all-gather of positions per step.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import conftest as T  # noqa: E402  (fixtures for the benchmark cinematics + loaders)

PEAK_FP32_TFLOPS = 157.3          # MI355X vector FP32 peak: 256 CU x 4 SIMD x 32 lanes x 2 (fma) x 2.4 GHz
PEAK_HBM_GBS = 8000.0
REHEARSE = os.environ.get("GAME_BENCH_REHEARSE") == "1"   # not a measurement mode: see dist_setup
SIMDS, CLOCK_HZ = 1024, 2.4e9     # 256 CUs x 4 SIMDs; one wave64 VALU instruction per 2 cycles per SIMD at best


def measured_counters(key, frames_per_launch):
    """HBM traffic and VALU instruction counts per launch from the committed rocprofv3 PMC summary
    (profiles/r*_traffic.json, written from the output of tools/profile_bench.sh). The counters are
    collected in separate profiler runs of this same command, so they are read back here instead of
    being measured inside the timed run; None when the profile does not cover this launch shape."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")), reverse=True):
        try:
            rec = json.load(open(path)).get(key)
        except (OSError, ValueError):
            continue
        if rec and rec.get("frames_per_launch") == frames_per_launch:
            return rec
    return None

# Algorithmic work per pixel (flop convention: add/sub/mul/div/min/max/abs/floor/fract/compare/select
# = 1, sin/cos/sqrt/atan = 1, mul+add pair = 2), summed over the stages of each cinematic.
CONFIGS = {
    "c1": dict(src=T.C1_HELLO, w=1024, h=1024, flops_px=47, bytes_px=4, bound="hbm", batch=200,
               workload="examples/001-hello.game circle|glow|tint, 1024x1024"),
    "c2": dict(src=T.C2_FBM, w=3840, h=2160, flops_px=1006, bytes_px=4, bound="valu",
               workload="fbm showcase: domain_warp(0.1,3.0)|fbm(3.0,octaves:6,0.5,2.0)|glow(1.5)|tint, 3840x2160, frame-sharded"),
    "c2b": dict(src=T.C2B_NOISE, w=3840, h=2160, flops_px=900, bytes_px=4, bound="valu",
                workload="examples/tutorial-05-noise.game (fbm4 + simplex + voronoi, 3 layers), 3840x2160"),
    "c3": dict(src=T.C3_SHOWCASE, w=3840, h=2160, flops_px=95, bytes_px=4, bound="hbm",
               workload="examples/007-showcase.game spectrum|bloom|chromatic|vignette|grain, 3840x2160"),
    "c4": dict(src=T.C4_TRAILS, w=7680, h=4320, flops_px=59, bytes_px=8, bound="hbm",
               workload="examples/003-memory-trails.game feedback memory, 7680x4320, frame-serial, band-sharded"),
}


def dist_setup(n_gpus):
    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if REHEARSE:
        local = 0       # rehearsal of the N > 1 control flow on a one-GPU box: every rank shares GPU 0, gloo rendezvous
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local)
        if REHEARSE:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local))
    if world != n_gpus:
        raise SystemExit(f"--gpus {n_gpus} but WORLD_SIZE={world}: launch N>1 with torch.distributed.run --nproc-per-node N")
    return torch, rank, world, local


def barrier(torch, world):
    torch.cuda.synchronize()
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
    torch.cuda.synchronize()


def max_over_ranks(torch, world, value):
    if world == 1:
        return value
    import torch.distributed as dist
    t = torch.tensor([value], dtype=torch.float64, device="cpu" if REHEARSE else "cuda")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def cpu_baseline_frames(game, oracle, cin, cfg, seconds_budget=20.0):
    """Times the oracle (CPU port of the same path) on whole frames of the same workload."""
    threads = min(16, os.cpu_count() or 1)
    w, h = cfg["w"], cfg["h"]
    rows = cin.uniform_rows([0.0], w, h)
    probe_rows = max(8, threads)
    t0 = time.time()
    oracle.render(cin.blob, w, h, rows[0], row0=0, rows=probe_rows, threads=threads)
    per_row = (time.time() - t0) / probe_rows
    nrows = int(max(threads, seconds_budget / max(per_row, 1e-9)))
    done, t0 = 0, time.time()
    while done < nrows:
        n = min(h, nrows - done)
        oracle.render(cin.blob, w, h, rows[0], row0=0, rows=n, threads=threads)
        done += n
    dt = time.time() - t0
    return {"value": round(done * w / dt / 1e6, 3), "unit": "Mpixel/s", "cores": threads, "kind": "port",
            "sample": f"{done} rows of {w} px of the same cinematic ({done * w / 1e6:.1f} Mpixel, {dt:.1f} s, {threads} threads)"}


def run_frames(args, torch, rank, world, local, key):
    cfg = CONFIGS[key]
    game = T.load_product()
    cin = game.compile(cfg["src"]).cinematics[0]
    w, h = cfg["w"], cfg["h"]
    banded = cin.uses_memory
    batch = args.batch if args.batch else cfg.get("batch", 25)
    if banded:
        # horizontal bands, multiples of 8 rows; every rank renders every frame of its band. The
        # feedback read is the same pixel of the previous frame, so no halo rows are exchanged.
        from game_engine_amd import distributed
        row0, nrows = distributed.band_for_rank(h, rank, world)
    else:
        row0, nrows = 0, h
    r = cin.renderer(w, h, device=local)
    r.set_ring(2 if banded else batch)      # stateless: one launch per step covers the whole batch

    def rows_for(step):
        if banded:
            ks = [step * batch + i for i in range(batch)]
        else:
            ks = [(step * batch + i) * world + rank for i in range(batch)]           # frame k -> GPU k mod N
        bass = [0.5 + 0.5 * np.sin(2 * np.pi * 0.5 * k / 60.0) for k in ks]
        audio = {"bass": bass} if key == "c4" else ({"bass": 0.8, "mid": 0.6, "treble": 0.4, "energy": 0.6} if key == "c3" else None)
        return cin.uniform_rows([k / 60.0 for k in ks], w, h, audio=audio)

    tables = [rows_for(s) for s in range(args.warmup + args.steps)]

    def step(s):
        if banded:
            r.render_band(tables[s], row0, nrows, readback=False)
        else:
            r.render(tables[s], readback=False)

    for s in range(args.warmup):
        step(s)
    r.sync()
    kernel_ms, launches = 0.0, 0
    barrier(torch, world)
    t0 = time.perf_counter()
    if args.events:
        r.timing_begin()                    # one HIP event pair per step on the render stream; nothing waits in between
    for s in range(args.warmup, args.warmup + args.steps):
        step(s)
    r.sync()
    barrier(torch, world)
    dt = max_over_ranks(torch, world, time.perf_counter() - t0)
    if args.events:
        kernel_ms, launches = r.timing_end()

    frames_total = args.steps * batch * (1 if banded else world)
    pixels_total = frames_total * w * h
    value = pixels_total / dt / 1e6
    out = {
        "metric": "Mpixel/s at 3840x2160 (fbm showcase)" if key == "c2" else f"Mpixel/s ({key})",
        "value": round(value, 1), "unit": "Mpixel/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "strong" if banded else "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": cfg["workload"], "frames_per_step": batch, "width": w, "height": h,
                   "parallelism": f"bands{world}" if banded else f"frames{world}", "output": "RGBA8 frames in an HBM ring"},
    }
    if rank == 0:
        if args.events and launches:
            px_launch = (nrows if banded else h) * w * (1 if banded else batch)
            sec = kernel_ms / launches / 1e3
            pmc = measured_counters(key, 1 if banded else batch) if world == 1 else None
            traffic = round(pmc["traffic_bytes_per_launch"]) if pmc else None
            if cfg["bound"] == "valu":
                ach = px_launch * cfg["flops_px"] / sec / 1e12
                out["roofline"] = {"bound": "valu", "achieved": round(ach, 2), "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s",
                                   "frac": round(ach / PEAK_FP32_TFLOPS, 4), "traffic": traffic,
                                   "kernel": "game_frame", "avg_launch_ms": round(sec * 1e3, 4), "pixels_per_launch": px_launch,
                                   "flops_per_pixel": cfg["flops_px"]}
            else:
                ach = px_launch * cfg["bytes_px"] / sec / 1e9
                out["roofline"] = {"bound": "hbm", "achieved": round(ach, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                   "frac": round(ach / PEAK_HBM_GBS, 4), "traffic": traffic,
                                   "kernel": "game_frame", "avg_launch_ms": round(sec * 1e3, 4), "pixels_per_launch": px_launch,
                                   "bytes_per_pixel": cfg["bytes_px"]}
            if cfg["bound"] == "valu" and not os.environ.get("GAME_HIP_NO_LUT"):
                out["roofline"]["note"] = ("lattice hashes / voronoi points are read from per-device tables, so the algorithmic flop rate "
                                           "can exceed the arithmetic peak; valu_issue_frac is the share of VALU issue slots actually used; "
                                           "arithmetic_only is the same kernel built without the tables")
                if world == 1 and not args.no_cpu:   # comparison legs are skipped together (--no-cpu: profiler runs)
                    # the same workload with every hash evaluated arithmetically (no tables), a few steps, outside the main timing
                    os.environ["GAME_HIP_NO_LUT"] = "1"
                    r2 = cin.renderer(w, h, device=local)
                    os.environ.pop("GAME_HIP_NO_LUT")
                    r2.set_ring(batch)
                    for s2 in range(3):
                        r2.render(tables[s2 % len(tables)], readback=False)
                    r2.sync()
                    ms2, n2 = 0.0, 10
                    for s2 in range(n2):
                        r2.render(tables[s2 % len(tables)], readback=False)
                        ms2 += r2.last_kernel_ms()[0]
                    r2.close()
                    sec2 = ms2 / n2 / 1e3
                    ach2 = px_launch * cfg["flops_px"] / sec2 / 1e12
                    out["roofline"]["arithmetic_only"] = {"value": round(px_launch / sec2 / 1e6, 1), "unit": "Mpixel/s", "achieved": round(ach2, 2),
                                                          "frac": round(ach2 / PEAK_FP32_TFLOPS, 4), "avg_launch_ms": round(sec2 * 1e3, 4)}
            if pmc:
                # share of the VALU issue slots (one wave64 instruction per 2 cycles per SIMD) this launch used
                out["roofline"]["valu_issue_frac"] = round(pmc["waves_per_launch"] * pmc["valu_instructions_per_wave"] * 2.0 / (SIMDS * CLOCK_HZ * sec), 4)
                out["roofline"]["valu_instructions_per_wave"] = round(pmc["valu_instructions_per_wave"], 1)
        if world == 1 and not args.no_cpu:
            oracle = T.Oracle()
            out["cpu_baseline"] = cpu_baseline_frames(game, oracle, cin, cfg, args.cpu_seconds)
            # parity spot check outside the timed region: a band of the last frame against the oracle
            last = tables[-1][-1:]
            r.reset_memory()
            got = r.render_band(last, h // 2 - 2, 4)[0]
            want = oracle.render(cin.blob, w, h, last[0], row0=h // 2 - 2, rows=4)
            out["parity"] = "bit-exact" if (not banded and np.array_equal(got, want)) else ("n/a (stateful)" if banded else "MISMATCH")
    r.close()
    return out


def run_particles(args, torch, rank, world, local):
    game = T.load_product()
    cin = game.compile(T.C5_GRAVITY).cinematics[0]
    n = args.particles
    from game_engine_amd import distributed
    first, count = distributed.particle_range(n, rank, world)
    r = cin.renderer(16, 16, device=local)
    r.particles_seed(n)
    gather = None
    if world > 1:
        from game_engine_amd import distributed
        gather = distributed.PositionExchange(r, n, first, count, torch)

    def step():
        r.particles_step(1 / 60.0, 1, first=first, count=count)
        if gather:
            gather.exchange()

    for _ in range(args.warmup):
        step()
    r.sync()
    kernel_ms, launches = 0.0, 0
    barrier(torch, world)
    t0 = time.perf_counter()
    if args.events:
        r.timing_begin()
    for _ in range(args.steps):
        step()
    r.sync()
    barrier(torch, world)
    dt = max_over_ranks(torch, world, time.perf_counter() - t0)
    if args.events:
        kernel_ms, launches = r.timing_end()
    pairs = float(n) * (n - 1) * args.steps
    out = {"metric": "Mpair/s (gravity all-pairs)", "value": round(pairs / dt / 1e6, 1), "unit": "Mpair/s", "n_gpus": world,
           "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True,
           "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": f"examples/010-gravity-particles.game, {n} particles, all-pairs, dt=1/60", "particles": n,
                      "parallelism": f"index{world}"}}
    if rank == 0 and args.events and launches:
        sec = kernel_ms / launches / 1e3
        ach = (20.0 * count * (n - 1) + 12.0 * count) / sec / 1e12
        out["roofline"] = {"bound": "valu", "achieved": round(ach, 2), "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s",
                           "frac": round(ach / PEAK_FP32_TFLOPS, 4), "traffic": None, "kernel": "game_gravity",
                           "avg_launch_ms": round(sec * 1e3, 4)}
    if rank == 0 and world == 1 and not args.no_cpu:
        oracle = T.Oracle()
        threads = min(16, os.cpu_count() or 1)
        m = min(n, 16384)
        p = oracle.seed_particles(m)
        t0 = time.time()
        oracle.gravity(cin.blob, p, 1 / 60.0, 1, threads=threads)
        dtc = time.time() - t0
        out["cpu_baseline"] = {"value": round(m * (m - 1) / dtc / 1e6, 2), "unit": "Mpair/s", "cores": threads, "kind": "port",
                               "sample": f"one step of {m} particles ({dtc:.1f} s, {threads} threads)"}
    r.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="c2", choices=sorted(CONFIGS) + ["c5"])
    ap.add_argument("--batch", type=int, default=0, help="frames per step (per GPU); default 25 (200 for the 1024x1024 config)")
    ap.add_argument("--particles", type=int, default=1048576)
    ap.add_argument("--no-events", dest="events", action="store_false", help="skip per-step HIP event timing")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    args = ap.parse_args()
    torch, rank, world, local = dist_setup(args.gpus)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (torch.cuda.is_available() is False)")
    torch.cuda.set_device(local)
    out = run_particles(args, torch, rank, world, local) if args.config == "c5" else run_frames(args, torch, rank, world, local, args.config)
    if rank == 0:
        if REHEARSE:
            out["data"] = "synthetic (REHEARSAL: all ranks on one GPU, not a measurement)"
        print(json.dumps(out), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
