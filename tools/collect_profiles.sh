#!/bin/bash
# Run ON THE GPU BOX (through gpurun) from the repo root: the bench line, the rocprofv3 kernel-trace summary of the same
# command, the two PMC passes behind roofline.traffic, the per-layer conv table, the joint-model (configs[4]) bench line +
# kernel summary, the bf16 GEMM micro-benchmark and the 2-rank rehearsal.  Outputs under gpurun_out/profiles_<tag>/ ; copy what
# should be judged into profiles/.
set -e
tag=${1:-r04}
part=${2:-all}          # a = bench line, kernel traces, PMC passes, joint legs; b = micro-benchmarks, rehearsals; all = both
out=$PWD/gpurun_out/profiles_$tag
mkdir -p $out
root=$PWD
cd /tmp && export TMPDIR=/tmp
if [ "$part" = "a" ] || [ "$part" = "all" ]; then
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
python3 $root/bench.py --config joint --steps 10 --joint-dropout 0.2 --no-roofline > $out/joint_bench_dropout.log 2>&1
tail -1 $out/joint_bench_dropout.log > $out/joint_bench_dropout.json
rm -rf $out/trace $out/joint $out/pmc_fetch $out/pmc_write
fi
if [ "$part" = "b" ] || [ "$part" = "all" ]; then
# the joint step issued EAGERLY (same launches as the captured graph): per-kernel durations comparable with the earlier rounds' traces
# (export, not `env`: under rocprofv3 the program itself must follow `--`)
export DCAP_JOINT_GRAPH=0
rocprofv3 --kernel-trace --stats -d $out/joint_eager -o joint -- python3 $root/bench.py --config joint --steps 10 --no-roofline > $out/joint_eager.log 2>&1
unset DCAP_JOINT_GRAPH
python3 $root/tools/prof_summary.py $out/joint_eager/joint_results.db $out/joint_kernels_eager.csv 13
rm -rf $out/joint_eager
python3 $root/tools/bgemm_bench.py 2>&1 | grep -v amdgpu.ids > $out/bgemm_bench.txt
(echo "== DCAP_BGEMM_TILE=128 (the round-2 128 x 128 loop on the same shapes)"; DCAP_BGEMM_TILE=128 python3 $root/tools/bgemm_bench.py 2>&1 | grep -v amdgpu.ids) >> $out/bgemm_bench.txt
python3 $root/tools/vocab_ce_bench.py 2>&1 | grep -v amdgpu.ids > $out/vocab_ce_bench.txt
(for t in 0 64 128; do echo "== tile $t (0 = the library's cost model)"; python3 $root/tools/bconv_bench.py --tile $t 2>&1 | grep -v amdgpu.ids; done) > $out/bconv_bench.txt
(echo "== one launch per timestep (default)"; python3 $root/tools/lstm_bench.py 2>&1 | grep B=; echo "== DCAP_LSTM_BWD=steps (gate kernel + split-K GEMM + slab reduce per backward timestep)"; DCAP_LSTM_BWD=steps python3 $root/tools/lstm_bench.py 2>&1 | grep B=
 echo "== recurrent_dropout masks, one fused launch per timestep (round 4 default)"; python3 $root/tools/lstm_bench.py --dropout 2>&1 | grep B=
 echo "== recurrent_dropout masks, DCAP_LSTM_MASKED_FUSED=0 (round 3: mask kernel + 4 GEMMs + gate kernel per step)"; DCAP_LSTM_MASKED_FUSED=0 python3 $root/tools/lstm_bench.py --dropout 2>&1 | grep B=) > $out/lstm_bench.txt
(echo "== default (128x64 producer/consumer rule + streaming short-K kernel)"; python3 $root/tools/conv_bench.py --reps 50 2>&1 | grep -v amdgpu.ids; echo "== DCAP_PW_RULE=0 DCAP_PW_STREAM=0 (round-1 tile rule, no streaming kernel)"; DCAP_PW_RULE=0 DCAP_PW_STREAM=0 python3 $root/tools/conv_bench.py --reps 50 2>&1 | grep -v amdgpu.ids) > $out/conv_bench.txt
(echo "== 3x3 layers, Winograd F(2x2,3x3) (default for frozen weights)"; python3 $root/tools/conv_bench.py --filter 3x3 --winograd --reps 50 2>&1 | grep -v amdgpu.ids
 echo "== the same layers, direct implicit GEMM"; python3 $root/tools/conv_bench.py --filter 3x3 --reps 50 2>&1 | grep -v amdgpu.ids
 echo "== DCAP_WINO_TILES=1 (the FIRST kernel everywhere: 32-tile blocks, register-staged transform through a V image)"; DCAP_WINO_TILES=1 python3 $root/tools/conv_bench.py --filter 3x3 --winograd --reps 50 2>&1 | grep -v amdgpu.ids
 echo "== DCAP_WINO_TILES=32 (the patch-staging kernel with 32-tile items everywhere)"; DCAP_WINO_TILES=32 python3 $root/tools/conv_bench.py --filter 3x3 --winograd --reps 50 2>&1 | grep -v amdgpu.ids
 for v in NOU NODMA NOBAR NOREAD ALL; do
   if [ -f $root/tools/variants/libdcap_y$v.so ]; then echo "== ablation build $v of wino64_kernel (tools/build_variant.sh y$v conv_wino.hip -DWINO_EXP_...: results are wrong, only the time matters)"; DCAP_LIB=$root/tools/variants/libdcap_y$v.so python3 $root/tools/conv_bench.py --filter 3x3 --winograd --reps 50 2>&1 | grep -E "res4|fpn_p2|fpn_p3|res2"; fi
 done) > $out/winograd_bench.txt
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $out/pmc_wino -- python3 $root/tools/conv_bench.py --filter fpn_p3 --winograd --reps 3 > $out/pmc_wino.log 2>&1
python3 - $out <<'PYEOF'
import csv, glob, collections, sys, json
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(out + "/pmc_wino/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]] += float(r["Counter_Value"])
json.dump({k: dict(v) for k, v in agg.items() if "wino" in k}, open(out + "/winograd_sq_counters.json", "w"), indent=1)
PYEOF
rocprofv3 --kernel-trace --stats -d $out/dec -o dec -- python3 $root/tools/decoder_bench.py --captions 64 --steps 50 > $out/dec.log 2>&1
python3 $root/tools/prof_summary.py $out/dec/dec_results.db $out/decoder_kernels.csv 53
cd $root
timeout -k 10 300 python3 bench.py --gpus 2 --steps 5 --warmup 2 > $out/rehearsal_2rank_selflaunch.log 2>&1 || true
timeout -k 10 300 python3 bench.py --config joint --gpus 2 --steps 5 --warmup 2 --no-roofline > $out/rehearsal_2rank_joint.log 2>&1 || true
bash tools/roialign_profile.sh > /dev/null 2>&1 && cp gpurun_out/roialign_profile.txt $out/roialign_hbm.txt || true
python3 $root/tools/wino_fit.py 2>&1 | grep -v amdgpu.ids > $out/winograd_fit.txt
rm -rf $out/dec $out/pmc_wino
fi
ls $out
