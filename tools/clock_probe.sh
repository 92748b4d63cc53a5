#!/bin/bash
# On the GPU box: GRBM_GUI_ACTIVE per dispatch of a long bench run of one config, against the dispatch's own duration in
# the same run: the clock the chip actually holds under that kernel once it has ramped.
# usage: tools/clock_probe.sh <config> [steps]
set -u
root=${GRAFT_REPO_ROOT:-$(pwd)}
cfg=$1; steps=${2:-200}
out=$root/gpurun_out/clock_$cfg
rm -rf $out; mkdir -p $out
export TMPDIR=/tmp
cd /tmp
timeout -s KILL 240 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace -d $out -o clk -- python3 $root/bench.py --config $cfg --steps $steps --warmup 20 --no-cpu --no-others > $out/stdout.txt 2>&1 || { tail -5 $out/stdout.txt; exit 1; }
python3 - <<PY
import sqlite3, glob
db = sorted(glob.glob("$out/*.db"))[0]
cur = sqlite3.connect(db).cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
rows = list(cur.execute("select dispatch_id, kernel_name, value from counters_collection where counter_name='GRBM_GUI_ACTIVE' and (kernel_name like 'game_%') order by dispatch_id"))
dur = {r[0]: (r[2] - r[1]) / 1000.0 for r in cur.execute(f"select dispatch_id, start, end from {kd}")}
n = len(rows)
for label, part in (("first 5", rows[:5]), ("last third", rows[2 * n // 3:])):
    cyc = sum(r[2] for r in part) / len(part) / 8.0
    us = sum(dur[r[0]] for r in part) / len(part)
    print(f"$cfg {label}: {len(part)} dispatches, {us:.1f} us, {cyc:.0f} cycles per XCD, {cyc / us / 1e3:.3f} GHz")
PY
