#!/bin/bash
# Run ON THE GPU BOX (through gpurun) from the repo root: the bench line, the rocprofv3 kernel-trace summary of the same
# command, the two PMC passes behind roofline.traffic, the per-layer conv table, the joint-model (configs[4]) bench line +
# kernel summary, the bf16 GEMM micro-benchmark and the 2-rank rehearsal.  Outputs under gpurun_out/profiles_<tag>/ ; copy what
# should be judged into profiles/.
set -e
tag=${1:-r02}
out=$PWD/gpurun_out/profiles_$tag
mkdir -p $out
root=$PWD
cd /tmp && export TMPDIR=/tmp
python3 $root/bench.py --steps 20 --warmup 3 --layer-table $out/conv_layers.tsv > $out/bench.log 2>&1
tail -1 $out/bench.log > $out/bench.json
rocprofv3 --kernel-trace --stats -d $out/trace -o bench -- python3 $root/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline --no-alt-math --no-other-configs > $out/trace.log 2>&1
python3 $root/tools/prof_summary.py $out/trace/bench_results.db $out/bench_kernel_stats.csv 23
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- python3 $root/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-roofline --no-alt-math --no-other-configs --no-pipeline > $out/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- python3 $root/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-roofline --no-alt-math --no-other-configs --no-pipeline > $out/pmc_write.log 2>&1
python3 $root/tools/pmc_traffic.py $out/pmc_fetch $out/pmc_write $out/pmc_traffic.json
python3 $root/bench.py --config joint --steps 10 > $out/joint_bench.log 2>&1
tail -1 $out/joint_bench.log > $out/joint_bench.json
python3 $root/bench.py --config joint --steps 10 --joint-dtype f32 > $out/joint_bench_f32.log 2>&1
tail -1 $out/joint_bench_f32.log > $out/joint_bench_f32.json
rocprofv3 --kernel-trace --stats -d $out/joint -o joint -- python3 $root/bench.py --config joint --steps 10 > $out/joint.log 2>&1
python3 $root/tools/prof_summary.py $out/joint/joint_results.db $out/joint_kernels.csv 13
python3 $root/tools/bgemm_bench.py > $out/bgemm_bench.txt 2>&1
cd $root
DCAP_DIST_BACKEND=gloo timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 5 --warmup 2 > $out/rehearsal_2rank_gloo.log 2>&1 || true
rm -rf $out/trace $out/joint $out/pmc_fetch $out/pmc_write
ls $out
