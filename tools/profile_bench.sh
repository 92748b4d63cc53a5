#!/bin/bash
# Profiles bench.py on the GPU box: kernel trace + stats, then PMC passes (each in a run of its own).
# usage: tools/profile_bench.sh <tag> [bench args...]      (PMC_ARGS: extra bench args for the counter passes only)
set -u
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/trace -o trace -- python3 $root/bench.py --no-cpu --no-others "$@" > $out/trace_stdout.txt 2>&1 || { echo "trace run failed"; tail -5 $out/trace_stdout.txt; exit 1; }
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VALU2 SQ_WAIT_INST_ANY" "SQ_INSTS_VALU_FLOPS_FP32 SQ_INSTS_VALU_FLOPS_FP32_TRANS SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  name=$(echo $pass | cut -d' ' -f1)
  GAME_BENCH_CLOCK_WARMUP_S=0 timeout -s KILL 240 rocprofv3 --pmc $pass -d $out/pmc_$name -o pmc -- python3 $root/bench.py --no-cpu --no-others --steps 2 --warmup 1 "$@" ${PMC_ARGS:-} > $out/pmc_${name}_stdout.txt 2>&1 || { echo "pmc pass $name failed"; tail -5 $out/pmc_${name}_stdout.txt; exit 1; }
done
# summarised on the box: the databases stay there (a counter pass of the feedback cinematic alone is tens of megabytes)
mkdir -p $root/gpurun_out/summary
python3 $root/tools/summarize_profile.py $out $root/gpurun_out/summary/$tag > /dev/null && rm -rf $out && ls $root/gpurun_out/summary
