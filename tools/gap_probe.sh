#!/bin/bash
# On the GPU box: kernel durations and the idle time between consecutive dispatches of a long bench run of one config
# (kernel trace only, no counters: counter collection serialises dispatches). usage: tools/gap_probe.sh <config> [steps]
set -u
root=${GRAFT_REPO_ROOT:-$(pwd)}
cfg=$1; steps=${2:-100}
out=$root/gpurun_out/gap_$cfg
rm -rf $out; mkdir -p $out
export TMPDIR=/tmp
cd /tmp
timeout -s KILL 240 rocprofv3 --kernel-trace -d $out -o gap -- python3 $root/bench.py --config $cfg --steps $steps --warmup 2 --no-cpu --no-others > $out/stdout.txt 2>&1 || { tail -5 $out/stdout.txt; exit 1; }
python3 - <<PY
import sqlite3, glob, statistics
cur = sqlite3.connect(sorted(glob.glob("$out/*.db"))[0]).cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]; ks = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
rows = list(cur.execute(f"select d.start, d.end, s.kernel_name from {kd} d join {ks} s on d.kernel_id = s.id order by d.start"))
late = rows[len(rows) // 2:]
dur = [(e - s) / 1000.0 for s, e, n in late if n.startswith('game_')]
gaps = [(late[i + 1][0] - late[i][1]) / 1000.0 for i in range(len(late) - 1)]
kinds = {}
for s, e, n in late: kinds[n[:28]] = kinds.get(n[:28], 0) + 1
print("$cfg: second half of the run,", len(late), "dispatches", kinds)
print("  game kernel us: median %.1f mean %.1f; gap to the next dispatch us: median %.1f mean %.1f max %.1f" % (statistics.median(dur), sum(dur) / len(dur), statistics.median(gaps), sum(gaps) / len(gaps), max(gaps)))
PY
grep -o '"avg_launch_ms": [0-9.]*' $out/stdout.txt | head -1
