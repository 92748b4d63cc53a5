#!/usr/bin/env python3
"""Benchmark of the GPU render path.

    python bench.py --gpus N --steps K --warmup W [--config c2]

Default workload (config c2) is the benchmark the metric is quoted on: the fbm showcase cinematic
`domain_warp(0.1, 3.0) | fbm(3.0, octaves: 6, ...) | glow(1.5) | tint(...)` at 3840x2160, 1000 frames with
u_time = k/60. One step = the whole job on each GPU: 1000 frames rendered into the HBM frame ring in ring-sized
launches of 25 frames (no read-back inside the timed region). With N > 1 there is one process per GPU; frames are
sharded k mod N with no collective on the data path and every GPU renders 1000 frames of an N x 1000 frame job, so
scaling is weak. `python bench.py --gpus N` starts its N rank processes itself (before anything touches a GPU) unless
a launcher (torch.distributed.run) already did. Rank 0 prints one JSON line.

Other configs (c1, c2b, c3, c4, c5) exercise the remaining benchmark cinematics; c4 shards horizontal bands, c5 shards
particle indices with an all-gather of positions per step. At N = 1 the default run also times each of them briefly
after the headline and reports them under "other_configs" (among them one step of the particle workload at its full
size, 2^24 particles: about 80 s).

    --scaling strong   stateless configs: the config's own job (c2: 1000 frames) is split k mod N over the GPUs instead
                       of every GPU rendering a whole job; value = job pixels / wall time
    --blur R           c4 with the frame filter on: per frame every rank renders its raw band, trades R raw rows with the
                       ranks above and below (RCCL point-to-point, device to device), then filters its band
A config whose data path needs RCCL (c5, c4 with --blur) exits non-zero when RCCL cannot start; only the rehearsal mode
(GAME_BENCH_REHEARSE=1: all ranks on one GPU, gloo) runs it over host memory, and says so in the line.
"""
import argparse
import glob
import importlib.util
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))

PEAK_FP32_TFLOPS = 157.3          # MI355X vector FP32 peak: 256 CU x 4 SIMD x 32 lanes x 2 (fma) x 2.4 GHz
PEAK_HBM_GBS = 8000.0
# The chip raises its clock only under sustained load: measured with GRBM_GUI_ACTIVE per dispatch (tools/clock_probe.sh),
# a frame kernel starts at 1.75 - 2.1 GHz after the GPU has idled (a kernel compile is enough) and reaches its steady
# 2.3 - 2.4 GHz after 35 - 85 ms of back-to-back launches. Every timed frame leg therefore first runs its own step,
# untimed, for this many seconds (on top of the --warmup steps), so that what is timed is the steady state; the line
# says so. (The particle legs need none: one step is 0.3 s of one kernel.)
CLOCK_WARMUP_S = float(os.environ.get("GAME_BENCH_CLOCK_WARMUP_S", "0.3"))   # 0 for counter passes, which need no steady clock
REHEARSE = os.environ.get("GAME_BENCH_REHEARSE") == "1"   # not a measurement mode: see dist_setup
CONTROL_BACKEND = "nccl"    # what carries the barrier and the max over ranks at N > 1 (dist_setup may fall back to gloo)
SIMDS, CLOCK_HZ = 1024, 2.4e9     # 256 CUs x 4 SIMDs; one wave64 VALU instruction per 2 cycles per SIMD at best


def _load(name, path, package=False):
    if name in sys.modules:
        return sys.modules[name]
    spec = importlib.util.spec_from_file_location(name, path, submodule_search_locations=[os.path.dirname(path)] if package else None)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def load_product():
    """The product binding (game-engine_amd/__init__.py -> libgame_hip.so); raises when the library is missing."""
    return _load("game_engine_amd", os.path.join(ROOT, "game-engine_amd", "__init__.py"), package=True)


def load_oracle():
    """The CPU restatement, only for the cpu_baseline leg and the parity spot check (never inside a timed GPU region)."""
    return _load("game_oracle_binding", os.path.join(ROOT, "oracle", "binding.py")).Oracle()


def measured_counters(key, frames_per_launch, kernel_hash):
    """HBM traffic and VALU instruction counts per launch from the committed rocprofv3 PMC summary
    (profiles/r*_traffic.json, written from the output of tools/profile_bench.sh). The counters are collected in
    separate profiler runs of this same command, so they are read back here instead of being measured inside the
    timed run. The record carries the hash of the kernel it was measured on; a record for another kernel build or
    another launch shape is ignored (None), so a stale profile never decorates a new kernel."""
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")), reverse=True):
        try:
            rec = json.load(open(path)).get(key)
        except (OSError, ValueError):
            continue
        if rec and rec.get("frames_per_launch") == frames_per_launch and rec.get("kernel_hash") == kernel_hash:
            rec["file"] = os.path.relpath(path, ROOT)
            return rec
    return None


def spawn_ranks(n, argv):
    """`--gpus N` without a launcher: start N rank processes (fresh interpreters, nothing here has touched a GPU),
    one per GPU, rendezvous on 127.0.0.1; rank 0 prints the JSON line. Returns the worst exit code."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for rank in range(n):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, "-u", os.path.abspath(__file__)] + argv, env=env))
    worst = 0
    pending = list(procs)
    while pending:
        for p in list(pending):
            rc = p.poll()
            if rc is None:
                continue
            pending.remove(p)
            if rc != 0:
                worst = worst or rc
                for q in pending:       # a rank failed: the others would wait in a barrier for ever
                    q.terminate()
        time.sleep(0.05)
    return worst


def dist_setup(n_gpus, needs_rccl=False):
    """needs_rccl: the config's data path runs collectives between the GPUs (particle positions, halo rows); a gloo
    fallback would carry them through host memory and still print a value, so RCCL failing is an error there."""
    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if REHEARSE:
        local = 0       # rehearsal of the N > 1 control flow on a one-GPU box: every rank shares GPU 0, gloo rendezvous
    if world != n_gpus:
        raise SystemExit(f"--gpus {n_gpus} but WORLD_SIZE={world}")
    if world > 1:
        import torch.distributed as dist
        if local >= torch.cuda.device_count():
            raise SystemExit(f"rank {rank}: GPU {local} does not exist ({torch.cuda.device_count()} visible)")
        torch.cuda.set_device(local)
        if REHEARSE:
            dist.init_process_group(backend="gloo")
        else:
            try:
                dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local))
                probe = torch.ones(1, device="cuda")
                dist.all_reduce(probe)          # the communicator works before anything is timed
                torch.cuda.synchronize()
                if int(probe.item()) != world:
                    raise RuntimeError(f"all_reduce over {world} ranks returned {probe.item()}")
            except Exception as e:              # RCCL unusable on this node: the frame-sharded configs need only a
                global CONTROL_BACKEND          # barrier and a max over ranks, which gloo provides as well
                if needs_rccl:
                    raise SystemExit(f"rank {rank}: RCCL start-up failed ({type(e).__name__}: {e}) and this config exchanges data between "
                                     "the GPUs: no measurement (GAME_BENCH_REHEARSE=1 rehearses the control flow over gloo)")
                print(f"rank {rank}: RCCL start-up failed ({type(e).__name__}: {e}); control collectives over gloo", file=sys.stderr, flush=True)
                if dist.is_initialized():
                    dist.destroy_process_group()
                dist.init_process_group(backend="gloo")
                CONTROL_BACKEND = "gloo"
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
    t = torch.tensor([value], dtype=torch.float64, device="cpu" if (REHEARSE or CONTROL_BACKEND == "gloo") else "cuda")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def clock_warmup(step, sync, torch, world):
    """Runs `step` untimed for about CLOCK_WARMUP_S seconds. The number of repetitions is derived from the slowest
    rank's first step, so every rank runs the same number (a step may hold collectives)."""
    step()                                  # the first step also pays for loading the code object
    sync()
    t0 = time.perf_counter()
    step()
    sync()
    one = max_over_ranks(torch, world, time.perf_counter() - t0)
    reps = int(min(5000, max(0.0, CLOCK_WARMUP_S - one) / max(one, 1e-6)))
    t0 = time.perf_counter()
    for _ in range(reps):
        step()
    sync()
    return round(one + time.perf_counter() - t0, 3)


def cpu_baseline_frames(oracle, cin, cfg, rows, seconds_budget=20.0):
    """Times the oracle (CPU port of the same path) on whole rows of the same workload."""
    threads = min(16, os.cpu_count() or 1)
    w, h = cfg["w"], cfg["h"]
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


def valu_roofline(pmc, sec, algorithmic_flops):
    """Roofline record of a VALU-bound kernel. `frac` is the share of VALU issue slots the launch used: executed
    wave64 VALU instructions (PMC SQ_INSTS_VALU per launch) x 2 cycles / (SIMDs x clock x launch time); `achieved`
    is the same thing in the peak's unit, every issued instruction counted as a full-rate 64-lane fma
    (128 flop): achieved / peak = frac. The algorithmic rate (operators of the emitted GLSL by the flop convention,
    whether executed or read from a lattice table) is reported beside it and may exceed the arithmetic peak."""
    alg = algorithmic_flops / sec / 1e12
    out = {"bound": "valu", "achieved": None, "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s", "frac": None, "traffic": None,
           "frac_definition": "VALU issue-slot occupancy: SQ_INSTS_VALU x 2 cycles / (1024 SIMDs x clock_hz_assumed x launch time)",
           "clock_hz_assumed": CLOCK_HZ, "algorithmic_tflops": round(alg, 2), "algorithmic_over_peak": round(alg / PEAK_FP32_TFLOPS, 4)}
    if pmc:
        insts = pmc["waves_per_launch"] * pmc["valu_instructions_per_wave"]
        frac = insts * 2.0 / (SIMDS * CLOCK_HZ * sec)
        out.update({"achieved": round(insts * 128.0 / sec / 1e12, 2), "frac": round(frac, 4),
                    "traffic": round(pmc["traffic_bytes_per_launch"]), "valu_instructions_per_launch": round(insts),
                    "valu_instructions_per_wave": round(pmc["valu_instructions_per_wave"], 1), "counters_from": pmc["file"],
                    "counters_note": "PMC counters come from separate rocprofv3 runs on another box, matched to this kernel build by hash; "
                                     "the launch time is this run's"})
        # what the hardware counted as FP32 operations (an fma = 2, a move or a compare = 0), over the vector peak
        if pmc.get("hw_counted_flops_per_wave_lane"):
            flops = pmc["hw_counted_flops_per_wave_lane"] * pmc["waves_per_launch"] * 64.0
            out["executed_tflops"] = round(flops / sec / 1e12, 2)
            out["frac_executed_flops"] = round(flops / sec / 1e12 / PEAK_FP32_TFLOPS, 4)
        # the same issue-slot share with the cycle count taken from the counters too (GRBM_GUI_ACTIVE, summed over the
        # 8 XCDs by the profiler) instead of launch time x an assumed 2.4 GHz: no clock assumption, both from the profiled run
        if pmc.get("gui_active_cycles_per_launch"):
            out["frac_issue_counted_cycles"] = round(insts * 2.0 / (SIMDS * pmc["gui_active_cycles_per_launch"] / 8.0), 4)
    else:
        out["note"] = "no committed PMC record matches this kernel build and launch shape: frac and traffic not available"
    return out


def time_frames(args, game, torch, rank, world, local, key, steps, warmup, extras):
    """Times `steps` passes of workload `key` on this rank's GPU; returns the bench record (roofline on rank 0)."""
    W = game.workloads
    cfg = W.CONFIGS[key]
    cin = game.compile(cfg["src"]).cinematics[0]
    w, h = cfg["w"], cfg["h"]
    banded = cin.uses_memory
    blur = args.blur if banded else 0
    strong = args.scaling == "strong" and not banded
    frames = args.frames if args.frames else cfg["frames"]          # frames per step: on this GPU (weak), of the whole job (strong)
    if banded:
        # horizontal bands, multiples of 8 rows; every rank renders every frame of its band. The
        # feedback read is the same pixel of the previous frame, so no halo rows are exchanged (unless --blur).
        row0, nrows = game.distributed.band_for_rank(h, rank, world)
        ks = list(range(frames))
    elif strong:
        row0, nrows = 0, h
        ks = game.distributed.frames_for_rank(frames, rank, world)   # the job's frame k -> GPU k mod N
    else:
        row0, nrows = 0, h
        ks = [i * world + rank for i in range(frames)]               # frame k -> GPU k mod N of an N x frames job
    batch = max(1, min(args.batch if args.batch else cfg["batch"], len(ks)))   # frames per launch (stateless cinematics)
    t_create = time.perf_counter()
    r = cin.renderer(w, h, device=local)
    create_ms = (time.perf_counter() - t_create) * 1e3
    compile_ms, tables_ms = r.setup_ms()
    r.set_ring(2 if banded else batch)
    table = W.frame_rows(cin, key, ks, w, h)
    halo = None
    if blur:
        r.set_blur(blur)
        if world > 1:
            halo = game.distributed.BandHaloExchange(r, row0, nrows, blur, torch)

    def step():
        if blur and world > 1:
            # two-phase frames: raw band, `blur` raw rows traded with the ranks above and below on the render stream, filter
            for k in range(len(ks)):
                r.render_raw_band(table[k], row0, nrows)
                halo.exchange()
                r.blur_band(row0, nrows, readback=False)
        elif banded:
            r.render_band(table, row0, nrows, readback=False)
        elif len(ks):
            r.render(table, readback=False)

    clock_s = clock_warmup(step, r.sync, torch, world) if len(ks) else 0.0
    for _ in range(warmup):
        step()
    r.sync()
    kernel_ms, launches = 0.0, 0
    barrier(torch, world)
    t0 = time.perf_counter()
    if args.events:
        r.timing_begin()                    # one HIP event pair per step on the render stream; nothing waits in between
    for _ in range(steps):
        step()
    r.sync()
    barrier(torch, world)
    dt = max_over_ranks(torch, world, time.perf_counter() - t0)
    if args.events:
        kernel_ms, launches = r.timing_end()

    job_frames = frames if (banded or strong) else frames * world     # frames of the whole job, per step
    value = steps * job_frames * w * h / dt / 1e6
    out = {
        "metric": "Mpixel/s at 3840x2160 (fbm showcase)" if key == "c2" else f"Mpixel/s ({key})",
        "value": round(value, 1), "unit": "Mpixel/s", "n_gpus": world, "steps": steps, "warmup": warmup,
        "ms_per_step": round(dt / steps * 1e3, 4), "higher_is_better": True, "scaling": "strong" if (banded or strong) else "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": cfg["workload"] + (f", every frame blurred (binomial, radius {blur}) before it is stored and fed back" if blur else ""),
                   "frames_per_step": len(ks), "frames_per_launch": 1 if banded else batch,
                   "job_frames_per_step": job_frames, "width": w, "height": h,
                   "parallelism": (f"bands{world}" + (f"+halo{blur}" if blur else "")) if banded else f"frames{world}",
                   "output": "RGBA8 frames in an HBM ring"},
        "timed_region_s": round(dt, 4),
        "clock_warmup_s": clock_s,       # untimed repetitions of the step ahead of the warm-up steps (see CLOCK_WARMUP_S)
    }
    if rank == 0 and args.events and launches and not blur:
        per_launch_frames = 1 if banded else batch
        px_launch = (nrows if banded else h) * w * per_launch_frames
        sec = kernel_ms / launches / 1e3
        # the counters describe one launch of this kernel build; a band launch (N > 1, memory cinematics) is another shape
        pmc = measured_counters(key, per_launch_frames, r.kernel_hash) if (world == 1 or not banded) else None
        ach = px_launch * cfg["bytes_px"] / sec / 1e9
        if cfg["bound"] == "valu":
            roof = valu_roofline(pmc, sec, px_launch * cfg["flops_px"])
            roof["flops_per_pixel"] = cfg["flops_px"]
            if not cfg.get("tables"):
                # no lattice tables: every operator of the flop convention is executed, so the convention's rate over the
                # vector peak is the fraction; the issue-slot share and the store rate against HBM ride along
                roof["valu_issue_frac"] = roof["frac"]
                roof.update({"achieved": roof["algorithmic_tflops"], "frac": roof["algorithmic_over_peak"],
                             "frac_definition": f"{cfg['flops_px']} flop per pixel (operators of the emitted GLSL) over the FP32 vector peak"})
            roof.update({"hbm_gbs": round(ach, 1), "hbm_frac": round(ach / PEAK_HBM_GBS, 4), "bytes_per_pixel": cfg["bytes_px"]})
        else:
            roof = {"bound": "hbm", "achieved": round(ach, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(ach / PEAK_HBM_GBS, 4),
                    "traffic": round(pmc["traffic_bytes_per_launch"]) if pmc else None, "bytes_per_pixel": cfg["bytes_px"]}
            if pmc:
                insts = pmc["waves_per_launch"] * pmc["valu_instructions_per_wave"]
                roof["valu_issue_frac"] = round(insts * 2.0 / (SIMDS * CLOCK_HZ * sec), 4)
        if pmc:
            roof["valu_instructions_per_pixel"] = round(pmc["waves_per_launch"] * pmc["valu_instructions_per_wave"] * 64.0 / px_launch, 1)
            if pmc.get("salu_instructions_per_wave") is not None:
                roof["salu_instructions_per_pixel"] = round(pmc["waves_per_launch"] * pmc["salu_instructions_per_wave"] * 64.0 / px_launch, 1)
        roof.update({"kernel": "game_frame", "kernel_hash": r.kernel_hash, "avg_launch_ms": round(sec * 1e3, 4),
                     "pixels_per_launch": px_launch, "launches_timed": launches})
        out["roofline"] = roof
    if rank == 0 and blur:
        # both kernels of a frame inside the wall-clock span: 16 B per pixel (raw band written and read back, previous
        # frame read, filtered frame written)
        gbs = steps * frames * w * h * 16 / dt / 1e9
        out["roofline"] = {"bound": "hbm", "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS * world, "unit": "GB/s",
                           "frac": round(gbs / (PEAK_HBM_GBS * world), 4), "traffic": None, "bytes_per_pixel": 16,
                           "kernel": "game_frame + blur_kernel" + (" + halo rows over RCCL" if world > 1 else ""),
                           "note": "from the wall clock of the timed region (the two-phase band calls carry no event pairs)"}
        out["setup"] = {"context_create_ms": round(create_ms, 1), "kernel_compile_ms": round(compile_ms, 1), "table_build_ms": round(tables_ms, 1)}
    if rank == 0 and world == 1 and extras and not args.no_cpu and not blur:   # comparison legs (skipped together: profiler runs)
        if cfg.get("tables") and not os.environ.get("GAME_HIP_NO_LUT") and "roofline" in out:
            # the same workload with every hash evaluated arithmetically (no tables), a few launches, outside the main timing
            os.environ["GAME_HIP_NO_LUT"] = "1"
            r2 = cin.renderer(w, h, device=local)
            os.environ.pop("GAME_HIP_NO_LUT")
            r2.set_ring(batch)
            r2.render(table[:batch], readback=False)
            r2.sync()
            r2.render(table[:10 * batch], readback=False)
            ms2, n2 = r2.last_kernel_ms()
            r2.close()
            sec2 = ms2 / n2 / 1e3
            ach2 = px_launch * cfg["flops_px"] / sec2 / 1e12
            out["roofline"]["arithmetic_only"] = {"value": round(px_launch / sec2 / 1e6, 1), "unit": "Mpixel/s", "algorithmic_tflops": round(ach2, 2),
                                                  "algorithmic_over_peak": round(ach2 / PEAK_FP32_TFLOPS, 4), "avg_launch_ms": round(sec2 * 1e3, 4),
                                                  "note": "same kernel built without the lattice tables (every hash computed)"}
        if not banded:
            # the same path with every frame also delivered to page-locked host memory (never the headline value)
            nf = min(8, batch)
            host = game.pinned_array((nf, h, w, 4))
            r.render(table[:nf], out=host)
            t1 = time.perf_counter()
            r.render(table[:nf], out=host)
            out["pcie_inclusive_mpixel_s"] = round(nf * w * h / (time.perf_counter() - t1) / 1e6, 1)
            del host
        if True:
            oracle = load_oracle()
            out["cpu_baseline"] = cpu_baseline_frames(oracle, cin, cfg, table, args.cpu_seconds)
            # parity spot check outside the timed region: a band of the last frame against the oracle
            last = table[-1:]
            r.reset_memory()
            got = r.render_band(last, h // 2 - 2, 4)[0]
            want = oracle.render(cin.blob, w, h, last[0], row0=h // 2 - 2, rows=4)
            out["parity"] = "bit-exact" if (not banded and np.array_equal(got, want)) else ("n/a (stateful)" if banded else "MISMATCH")
    r.close()
    return out


def sync_with_heartbeat(r, label, every=20.0):
    """r.sync() for work that takes minutes: a line on stderr every `every` seconds while the GPU works, so that
    whoever watches the run sees it is alive (the wait itself is the library's stream synchronisation)."""
    import threading
    done = threading.Event()

    def beat():
        t0 = time.time()
        while not done.wait(every):
            print(f"[bench] {label}: GPU busy, {time.time() - t0:.0f} s", file=sys.stderr, flush=True)

    th = threading.Thread(target=beat, daemon=True)
    th.start()
    try:
        r.sync()
    finally:
        done.set()
        th.join()


def time_particles(args, game, torch, rank, world, local, n, steps, warmup, extras, fast=False):
    cin = game.compile(game.workloads.C5_GRAVITY).cinematics[0]
    first, count = game.distributed.particle_range(n, rank, world)
    r = cin.renderer(16, 16, device=local, gravity_fast=fast)
    r.particles_seed(n)
    gather = game.distributed.PositionExchange(r, n, first, count, torch) if world > 1 else None

    def step():
        r.particles_step(1 / 60.0, 1, first=first, count=count)
        if gather:
            gather.exchange()

    for _ in range(warmup):
        step()
    r.sync()
    kernel_ms, launches = 0.0, 0
    barrier(torch, world)
    t0 = time.perf_counter()
    if args.events:
        r.timing_begin()
    for _ in range(steps):
        step()
    if gather:
        gather.finish()
    sync_with_heartbeat(r, f"gravity step, {n} particles")
    barrier(torch, world)
    dt = max_over_ranks(torch, world, time.perf_counter() - t0)
    if args.events:
        kernel_ms, launches = r.timing_end()
    pairs = float(n) * (n - 1) * steps
    out = {"metric": "Mpair/s (gravity all-pairs)", "value": round(pairs / dt / 1e6, 1), "unit": "Mpair/s", "n_gpus": world,
           "steps": steps, "warmup": warmup, "ms_per_step": round(dt / steps * 1e3, 4), "higher_is_better": True,
           "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": f"examples/010-gravity-particles.game, {n} particles, all-pairs, dt=1/60" +
                                  (" (fast arithmetic: not a parity mode)" if fast else ""), "particles": n,
                      "parallelism": f"index{world}"},
           "timed_region_s": round(dt, 4)}
    if rank == 0 and args.events and launches:
        sec = kernel_ms / steps / 1e3          # device time of one step of this rank's range (one launch per 2^20 particles)
        flops = 20.0 * count * (n - 1) + 12.0 * count
        pmc = measured_counters("c5", n, r.kernel_hash) if world == 1 and not fast else None
        roof = valu_roofline(pmc, sec, flops)
        # for the particle kernel the flop convention (20 per pair) is below the arithmetic peak by construction, so
        # it stays the headline fraction; the issue-slot occupancy is reported beside it
        if roof["frac"] is not None:
            roof["valu_issue_frac"] = roof["frac"]
        roof.update({"achieved": roof["algorithmic_tflops"], "frac": roof["algorithmic_over_peak"],
                     "frac_definition": "20 flop per pair + 12 per particle, over the FP32 vector peak",
                     "kernel": "game_gravity", "kernel_hash": r.kernel_hash, "avg_step_ms": round(sec * 1e3, 4),
                     "launches_per_step": launches // steps, "flops_per_pair": 20})
        roof.pop("note", None)
        out["roofline"] = roof
    if rank == 0 and world == 1 and extras and not args.no_cpu:
        oracle = load_oracle()
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


def time_filtered_feedback(game, local, radius, frames):
    """The 8K feedback cinematic with the frame filter on: fused kernel into the raw frame + blur into the ring slot,
    per frame. Frames stay in HBM; both kernels are inside the timed span."""
    W = game.workloads
    cfg = W.CONFIGS["c4"]
    cin = game.compile(cfg["src"]).cinematics[0]
    w, h = cfg["w"], cfg["h"]
    r = cin.renderer(w, h, device=local)
    r.set_blur(radius)
    rows = W.frame_rows(cin, "c4", range(frames), w, h)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < CLOCK_WARMUP_S:       # untimed, to the steady clock (one process, no collective)
        r.render(rows[:25], readback=False)
        r.sync()
    t0 = time.perf_counter()
    r.render(rows, readback=False)
    r.sync()
    dt = time.perf_counter() - t0
    ms, launches = r.last_kernel_ms()
    r.close()
    return {"value": round(frames * w * h / dt / 1e6, 1), "unit": "Mpixel/s", "steps": 1, "ms_per_step": round(dt * 1e3, 4),
            "timed_region_s": round(dt, 4),
            "workload": f"{cfg['workload']}, every frame blurred (binomial, radius {radius}) before it is stored and fed back",
            "roofline": {"bound": "hbm", "achieved": round(frames * w * h * 16 / (ms / 1e3) / 1e9, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                         "frac": round(frames * w * h * 16 / (ms / 1e3) / 1e9 / PEAK_HBM_GBS, 4), "traffic": None, "bytes_per_pixel": 16,
                         "kernel": "game_frame + blur_kernel", "avg_launch_ms": round(ms / launches, 4), "launches_timed": launches}}


def brief(rec):
    """The part of a bench record kept per entry of "other_configs"."""
    keep = {k: rec[k] for k in ("value", "unit", "steps", "ms_per_step", "timed_region_s", "clock_warmup_s") if k in rec}
    keep["workload"] = rec["config"]["workload"]
    if "roofline" in rec:
        keep["roofline"] = {k: v for k, v in rec["roofline"].items() if k not in ("frac_definition", "note", "arithmetic_only")}
    return keep


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="c2", choices=["c1", "c2", "c2b", "c3", "c3b", "c4", "c5"])
    ap.add_argument("--frames", type=int, default=0, help="frames per step on each GPU; default = the whole job of the config (c2: 1000)")
    ap.add_argument("--batch", type=int, default=0, help="frames per launch; default 25 (200 for the 1024x1024 config)")
    ap.add_argument("--particles", type=int, default=1048576)
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="stateless configs at N > 1: weak = every GPU renders a whole job (N x frames in all), strong = one job split k mod N")
    ap.add_argument("--blur", type=int, default=0, help="c4 only: frame filter radius 1..4; at N > 1 the bands trade halo rows over RCCL")
    ap.add_argument("--no-events", dest="events", action="store_false", help="skip per-step HIP event timing")
    ap.add_argument("--no-cpu", action="store_true", help="skip the comparison legs (CPU baseline, table-free build, host delivery)")
    ap.add_argument("--no-others", action="store_true", help="skip the brief runs of the other configs after the headline")
    ap.add_argument("--no-16m", action="store_true", help="skip the 2^24-particle step (about 80 s) among the other configs")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))
    if args.blur and args.config != "c4":
        raise SystemExit("--blur applies to --config c4 (the feedback cinematic)")
    torch, rank, world, local = dist_setup(args.gpus, needs_rccl=args.config == "c5" or (args.config == "c4" and args.blur > 0))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (torch.cuda.is_available() is False)")
    torch.cuda.set_device(local)
    game = load_product()
    if args.config == "c5":
        out = time_particles(args, game, torch, rank, world, local, args.particles, args.steps, args.warmup, True)
    else:
        out = time_frames(args, game, torch, rank, world, local, args.config, args.steps, args.warmup, True)
    if rank == 0 and world == 1 and args.config == "c2" and not args.no_others and not args.frames and not args.batch:
        others = {}
        for key, steps in (("c1", 100), ("c2b", 10), ("c3", 50), ("c3b", 50), ("c4", 5)):
            others[key] = brief(time_frames(args, game, torch, rank, world, local, key, steps, 1, False))
        others["c5"] = brief(time_particles(args, game, torch, rank, world, local, 1048576, 3, 1, False))
        others["c5_fast"] = brief(time_particles(args, game, torch, rank, world, local, 1048576, 3, 1, False, fast=True))
        others["c4_blur4"] = time_filtered_feedback(game, local, 4, 100)
        if not args.no_16m:
            # the particle workload at its full size: one step of 2^24 particles (2.8e14 pair evaluations, about 80 s), no warm-up step
            others["c5_16m"] = brief(time_particles(args, game, torch, rank, world, local, 1 << 24, 1, 0, False))
        out["other_configs"] = others
    if rank == 0:
        if world > 1:
            out["control_backend"] = "gloo" if REHEARSE else CONTROL_BACKEND
        if REHEARSE:
            out["data"] = "synthetic (REHEARSAL: all ranks on one GPU, not a measurement)"
        print(json.dumps(out), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
