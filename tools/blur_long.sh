#!/bin/bash
# On the GPU box: the frame filter's tests, then its kernels' durations at 8K in long runs (steady clock).
# usage: tools/blur_long.sh [radii, default 1,2,3,4] [frames per radius, default 1500]
set -u
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
timeout -k 10 300 python -u -m pytest tests/test_blur_filter.py -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/blur_tests.log 2>&1 || { tail -20 gpurun_out/blur_tests.log; exit 1; }
tail -1 gpurun_out/blur_tests.log
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/pb
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d /tmp/pb -o blur -- python3 $root/tools/blur_rate.py 0,${1:-1,2,3,4} ${2:-1500} > $root/gpurun_out/blur_rate.log 2>&1 || { tail -5 $root/gpurun_out/blur_rate.log; exit 1; }
python3 $root/tools/kernel_stats.py /tmp/pb/blur_results.db | tee $root/gpurun_out/blur_kernels.txt
