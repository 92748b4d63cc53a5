#!/bin/bash
# On the GPU box: PMC counters of the frame filter kernels at 8K, one rocprofv3 run per counter group.
# usage: tools/blur_pmc.sh [radii, default 4]
set -u
root=${GRAFT_REPO_ROOT:-$(pwd)}
radii=${1:-4}
out=$root/gpurun_out/pmc_blur
rm -rf $out; mkdir -p $out
export TMPDIR=/tmp
cd /tmp
i=0
for pass in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_WAVES" \
            "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY" \
            "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $pass -d $out/p$i -o pmc -- python3 $root/tools/blur_rate.py $radii 12 > $out/p${i}_stdout.txt 2>&1 || { echo "pass $i failed"; tail -5 $out/p${i}_stdout.txt; exit 1; }
done
python3 - <<PY | tee $out/summary.txt
import sqlite3, glob
for db in sorted(glob.glob("$out/p*/pmc_results.db")):
    cur = sqlite3.connect(db).cursor()
    q = "select kernel_name, counter_name, count(*), avg(value) from counters_collection where kernel_name like '%blur%' group by kernel_name, counter_name"
    for k, c, n, v in cur.execute(q):
        print(f"{k[20:60]:40s} {c:24s} {v:14.6g} (x{n})")
PY
