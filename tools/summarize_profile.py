#!/usr/bin/env python3
"""Summarises the rocprofv3 (rocpd sqlite) output of tools/profile_bench.sh into one JSON + text file.

    python tools/summarize_profile.py gpurun_out/prof_<tag> profiles/r01_<tag>

Writes <out>.json (kernel-trace stats per kernel, launch resources, PMC counters averaged per
dispatch of each kernel) and <out>.txt (the same as a table)."""
import glob, json, os, sqlite3, sys


def main():
    src, out = sys.argv[1], sys.argv[2]
    summary = {"kernel_trace_stats": [], "kernel_resources": [], "pmc_per_dispatch": {}, "bench_lines": {}}
    for db in glob.glob(os.path.join(src, "trace", "*.db")):
        cur = sqlite3.connect(db).cursor()
        for name, calls, total, avg, pct in cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels"):
            summary["kernel_trace_stats"].append({"name": name, "calls": calls, "total_us": round(total, 3), "avg_us": round(avg, 3), "pct": round(pct, 3)})
        q = ("select name, grid_x, grid_y, grid_z, workgroup_x, vgpr_count, accum_vgpr_count, sgpr_count, lds_size, scratch_size, "
             "count(*), avg(duration), min(duration), max(duration) from kernels group by name, grid_x, grid_y, grid_z")
        for r in cur.execute(q):
            summary["kernel_resources"].append({"name": r[0], "grid": list(r[1:4]), "workgroup": r[4], "vgpr": r[5], "agpr": r[6], "sgpr": r[7],
                                                "lds": r[8], "scratch": r[9], "dispatches": r[10], "avg_us": round(r[11] / 1e3, 3),
                                                "min_us": round(r[12] / 1e3, 3), "max_us": round(r[13] / 1e3, 3)})
    for d in sorted(glob.glob(os.path.join(src, "pmc_*"))):
        for db in glob.glob(os.path.join(d, "*.db")):
            cur = sqlite3.connect(db).cursor()
            q = "select kernel_name, counter_name, grid_size, count(*), avg(value) from counters_collection group by kernel_name, counter_name, grid_size"
            for kname, cname, grid, n, avg in cur.execute(q):
                summary["pmc_per_dispatch"].setdefault(f"{kname} grid={grid}", {})[cname] = {"dispatches": n, "avg": avg}
    for f in glob.glob(os.path.join(src, "*_stdout.txt")):
        for line in open(f, errors="replace"):
            if line.startswith('{"metric"'):
                summary["bench_lines"][os.path.basename(f)] = json.loads(line)
    os.makedirs(os.path.dirname(out) or ".", exist_ok=True)
    json.dump(summary, open(out + ".json", "w"), indent=1)
    with open(out + ".txt", "w") as t:
        t.write("kernel-trace stats (rocprofv3 --kernel-trace --stats)\n")
        t.write(f"{'kernel':44s} {'calls':>6s} {'total us':>12s} {'avg us':>10s} {'%':>7s}\n")
        for k in summary["kernel_trace_stats"]:
            t.write(f"{k['name'][:44]:44s} {k['calls']:6d} {k['total_us']:12.1f} {k['avg_us']:10.2f} {k['pct']:7.2f}\n")
        t.write("\nlaunch shapes\n")
        for k in summary["kernel_resources"]:
            t.write(f"{k['name'][:32]:32s} grid {k['grid']} wg {k['workgroup']} vgpr {k['vgpr']} sgpr {k['sgpr']} lds {k['lds']} scratch {k['scratch']} "
                    f"x{k['dispatches']} avg {k['avg_us']} us (min {k['min_us']}, max {k['max_us']})\n")
        t.write("\nPMC counters, average per dispatch (one rocprofv3 --pmc run per counter group)\n")
        for k, cs in summary["pmc_per_dispatch"].items():
            for c, v in sorted(cs.items()):
                t.write(f"{k[:60]:60s} {c:28s} {v['avg']:.6g} (x{v['dispatches']})\n")
    print(open(out + ".txt").read())


if __name__ == "__main__":
    main()
