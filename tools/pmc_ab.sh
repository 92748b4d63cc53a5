#!/bin/bash
# PMC comparison of kernel build variants: tools/pmc_ab.sh <config> "<flagsA>" "<flagsB>" ...
set -u
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/pmc_ab
rm -rf $out; mkdir -p $out
export TMPDIR=/tmp
cd /tmp
i=0
for pass in "${PMC_PASS1:-SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_WAVES}" "GRBM_GUI_ACTIVE" "${PMC_PASS3:-SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_IFETCH}"; do
  i=$((i+1))
  timeout -s KILL 240 rocprofv3 --pmc $pass -d $out/p$i -o pmc -- python3 $root/tools/ab_flags.py "$@" > $out/p${i}_stdout.txt 2>&1 || { echo "pass $i failed"; tail -5 $out/p${i}_stdout.txt; }
done
python3 - <<PY
import sqlite3, glob, collections
for db in sorted(glob.glob("$out/p*/pmc_results.db")):
    cur=sqlite3.connect(db).cursor()
    rows=list(cur.execute("select dispatch_id, counter_name, value from counters_collection where kernel_name='game_frame' order by dispatch_id"))
    ids=sorted({r[0] for r in rows})
    nvar=$(( $# - 1 ))
    # dispatch order: per variant 1 band render at creation, then rounds interleaved
    agg=collections.defaultdict(list)
    body=ids[nvar:]
    pos={d:i for i,d in enumerate(body)}
    for d,c,v in rows:
        if d in pos: agg[(pos[d]%nvar,c)].append(v)
    for (var,c),vals in sorted(agg.items()):
        print(db.split('/')[-2], 'variant',var,c, sum(vals)/len(vals))
PY
