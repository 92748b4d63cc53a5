#!/usr/bin/env python3
"""Builds profiles/<round>_traffic.json (read back by bench.py for roofline.traffic and the VALU issue
share) from the per-config summaries written by tools/summarize_profile.py.

    python tools/make_traffic_json.py r01 c2 c3 c4 ...

HBM traffic per launch = 2 x FETCH_SIZE + WRITE_SIZE (rocprofv3 reports KiB; on gfx950 FETCH_SIZE counts
128-byte read requests as 64 bytes, WRITE_SIZE is exact for the store widths used here)."""
import json, sys

rnd, configs = sys.argv[1], sys.argv[2:]
out = {}
for c in configs:
    d = json.load(open(f"profiles/{rnd}_{c}.json"))
    kname = "game_gravity" if c == "c5" else "game_frame"
    # the launch shape the run is made of: the one with the most dispatches (comparison legs add a few others)
    res = max((r for r in d["kernel_resources"] if r["name"] == kname), key=lambda r: r["dispatches"])
    k = max((v for kk, v in d["pmc_per_dispatch"].items() if kk.startswith(kname)), key=lambda v: v["SQ_WAVES"]["dispatches"])
    fetch, write = k["FETCH_SIZE"]["avg"] * 1024, k["WRITE_SIZE"]["avg"] * 1024
    line = d["bench_lines"]["trace_stdout.txt"]
    # what bench.py matches a record by: the launch shape (frames per launch; particles for the gravity kernel) and the
    # hash of the kernel build the counters were taken on
    shape = line["config"]["particles"] if c == "c5" else res["grid"][2]
    out[c] = {"kernel": kname, "frames_per_launch": shape, "kernel_hash": line["roofline"]["kernel_hash"],
              "fetch_size_bytes_raw": fetch, "write_size_bytes": write,
              "fetch_correction": 2.0, "traffic_bytes_per_launch": 2.0 * fetch + write,
              "waves_per_launch": k["SQ_WAVES"]["avg"],
              "valu_instructions_per_wave": k["SQ_INSTS_VALU"]["avg"] / k["SQ_WAVES"]["avg"],
              "valu2_share": k["SQ_ACTIVE_INST_VALU2"]["avg"] / k["SQ_ACTIVE_INST_VALU"]["avg"],
              "hw_counted_flops_per_wave_lane": k["SQ_INSTS_VALU_FLOPS_FP32"]["avg"] / k["SQ_WAVES"]["avg"],
              "salu_instructions_per_wave": k["SQ_INSTS_SALU"]["avg"] / k["SQ_WAVES"]["avg"],
              "wait_inst_any_share": k["SQ_WAIT_INST_ANY"]["avg"] / k["SQ_WAVE_CYCLES"]["avg"],
              # GRBM_GUI_ACTIVE as the profiler reports it: summed over the 8 XCDs, for the launch of the counter pass
              "gui_active_cycles_per_launch": k.get("GRBM_GUI_ACTIVE", {}).get("avg"),
              "kernel_avg_us_rocprof": res["avg_us"], "vgpr": res["vgpr"], "lds": res["lds"],
              "source": f"profiles/{rnd}_{c}.json (tools/profile_bench.sh: rocprofv3 --kernel-trace --stats, then one --pmc run per counter group)"}
json.dump(out, open(f"profiles/{rnd}_traffic.json", "w"), indent=1)
print(json.dumps(out, indent=1))
