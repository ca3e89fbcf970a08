#!/bin/bash
# Run ON THE GPU BOX (gpurun) from the repo root: the joint (configs[4]) bench line and the rocprofv3 kernel summaries of the step --
# as the bench runs it (captured hipGraph, RPN backward on the second stream) and issued eagerly (DCAP_STEP_GRAPH=0: same launches).
# Outputs under gpurun_out/<tag>_joint_*.
set -e
tag=${1:-r05}
root=$PWD
out=$root/gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
python3 $root/bench.py --config joint --steps 20 > $out/${tag}_joint_bench.log 2>&1
tail -1 $out/${tag}_joint_bench.log > $out/${tag}_joint_bench.json
rocprofv3 --kernel-trace --stats -d $out/jt -o joint -- python3 $root/bench.py --config joint --steps 10 --no-roofline > $out/${tag}_joint_trace.log 2>&1
python3 $root/tools/prof_summary.py $out/jt/joint_results.db $out/${tag}_joint_kernels.csv 13
python3 $root/tools/prof_timeline.py $out/jt/joint_results.db $out/${tag}_joint_timeline.tsv || true
rm -rf $out/jt
if [ "$2" = "eager" ]; then
export DCAP_STEP_GRAPH=0
rocprofv3 --kernel-trace --stats -d $out/jt -o joint -- python3 $root/bench.py --config joint --steps 10 --no-roofline > $out/${tag}_joint_trace_eager.log 2>&1
unset DCAP_STEP_GRAPH
python3 $root/tools/prof_summary.py $out/jt/joint_results.db $out/${tag}_joint_kernels_eager.csv 13
rm -rf $out/jt
fi
