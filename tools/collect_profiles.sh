#!/bin/bash
# Run ON THE GPU BOX (through gpurun) from the repo root: the bench line, the rocprofv3 kernel-trace summary of the same
# command, the two PMC passes behind roofline.traffic, the per-layer conv table, the joint-model (configs[4]) bench line +
# kernel summaries + step timeline, the micro-benchmarks and the 2-rank rehearsals.  Outputs under gpurun_out/profiles_<tag>/ ; copy what
# should be judged into profiles/.
set -e
tag=${1:-r06}
part=${2:-all}          # a1 = bench line, kernel trace, PMC passes; a2 = joint legs; a = a1 + a2; b = micro-benchmarks, rehearsals; all = everything
out=$PWD/gpurun_out/profiles_$tag
mkdir -p $out
root=$PWD
cd /tmp && export TMPDIR=/tmp
if [ "$part" = "a1" ] || [ "$part" = "a" ] || [ "$part" = "all" ]; then
python3 $root/bench.py --steps 20 --warmup 3 --layer-table $out/conv_layers.tsv > $out/bench.log 2>&1
tail -1 $out/bench.log > $out/bench.json
rocprofv3 --kernel-trace --stats -d $out/trace -o bench -- python3 $root/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline --no-alt-math --no-other-configs > $out/trace.log 2>&1
python3 $root/tools/prof_summary.py $out/trace/bench_results.db $out/bench_kernel_stats.csv 23
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- python3 $root/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-roofline --no-alt-math --no-other-configs --no-pipeline > $out/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- python3 $root/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-roofline --no-alt-math --no-other-configs --no-pipeline > $out/pmc_write.log 2>&1
python3 $root/tools/pmc_traffic.py $out/pmc_fetch $out/pmc_write $out/pmc_traffic.json
rm -rf $out/trace $out/pmc_fetch $out/pmc_write
fi
if [ "$part" = "a2" ] || [ "$part" = "a" ] || [ "$part" = "all" ]; then
# joint model: bench line (bf16 as specified, fp32, the reference's dropout), kernel summary + step timeline (captured graph and eager)
python3 $root/bench.py --config joint --steps 20 > $out/joint_bench.log 2>&1
tail -1 $out/joint_bench.log > $out/joint_bench.json
python3 $root/bench.py --config joint --steps 10 --joint-dtype f32 --no-roofline > $out/joint_bench_f32.log 2>&1
tail -1 $out/joint_bench_f32.log > $out/joint_bench_f32.json
python3 $root/bench.py --config joint --steps 10 --joint-dropout 0.2 --no-roofline > $out/joint_bench_dropout.log 2>&1
tail -1 $out/joint_bench_dropout.log > $out/joint_bench_dropout.json
python3 $root/bench.py --config joint --steps 10 --joint-images-per-gpu 2 --no-roofline > $out/joint_bench_2img.log 2>&1
tail -1 $out/joint_bench_2img.log > $out/joint_bench_2img.json
DCAP_VOCAB_MATERIALIZE=0 python3 $root/bench.py --config joint --steps 10 --no-roofline > $out/joint_bench_recompute_logits.log 2>&1
tail -1 $out/joint_bench_recompute_logits.log > $out/joint_bench_recompute_logits.json
rocprofv3 --kernel-trace --stats -d $out/joint -o joint -- python3 $root/bench.py --config joint --steps 10 --no-roofline > $out/joint.log 2>&1
python3 $root/tools/prof_summary.py $out/joint/joint_results.db $out/joint_kernels.csv 37   # 8 warm-up + 6 pipeline warm-up + 10 timed (pipelined) + 3 + 10 serial steps beside them
python3 $root/tools/prof_timeline.py $out/joint/joint_results.db $out/joint_timeline.tsv amsgrad 15 || true          # a step of the pipelined leg (the 13 serial steps run behind it)
python3 $root/tools/prof_timeline.py $out/joint/joint_results.db $out/joint_timeline_serial.tsv || true
rm -rf $out/joint
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/jf -- python3 $root/bench.py --config joint --steps 3 --warmup 2 --no-roofline > $out/joint_pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/jw -- python3 $root/bench.py --config joint --steps 3 --warmup 2 --no-roofline > $out/joint_pmc_write.log 2>&1
python3 $root/tools/pmc_traffic.py $out/jf $out/jw $out/joint_pmc_traffic.json || true
rm -rf $out/jf $out/jw
fi
if [ "$part" = "b" ] || [ "$part" = "all" ]; then
python3 $root/tools/bgemm_bench.py 2>&1 | grep -v amdgpu.ids > $out/bgemm_bench.txt
python3 $root/tools/vocab_ce_bench.py 2>&1 | grep -v amdgpu.ids > $out/vocab_ce_bench.txt
(for t in 0 64 128; do echo "== tile $t (0 = the library's cost model)"; python3 $root/tools/bconv_bench.py --tile $t 2>&1 | grep -v amdgpu.ids; done) > $out/bconv_bench.txt
(echo "== one launch per timestep"; python3 $root/tools/lstm_bench.py 2>&1 | grep B=
 echo "== recurrent_dropout masks, one fused launch per timestep"; python3 $root/tools/lstm_bench.py --dropout 2>&1 | grep B=) > $out/lstm_bench.txt
python3 $root/tools/conv_bench.py --reps 50 2>&1 | grep -v amdgpu.ids > $out/conv_bench.txt
(echo "== 3x3 layers, Winograd F(2x2,3x3), products on the bf16 pipe in split arithmetic (default for frozen weights since round 5)"; python3 $root/tools/conv_bench.py --filter 3x3 --winograd --b3 --reps 50 2>&1 | grep -v amdgpu.ids
 echo "== the same layers, Winograd F(2x2,3x3) with fp32 MFMA products (rounds 3-4)"; python3 $root/tools/conv_bench.py --filter 3x3 --winograd --reps 50 2>&1 | grep -v amdgpu.ids
 echo "== the same layers, direct implicit GEMM"; python3 $root/tools/conv_bench.py --filter 3x3 --reps 50 2>&1 | grep -v amdgpu.ids) > $out/winograd_bench.txt
(echo "== fp32 products (wino64_kernel)"; python3 $root/tools/wino_fit.py 2>&1 | grep -v amdgpu.ids; echo "== split-bf16 products (wino64b_kernel)"; python3 $root/tools/wino_fit.py --b3 2>&1 | grep -v amdgpu.ids) > $out/winograd_fit.txt
python3 $root/tools/chain_bench.py 2>&1 | grep -v amdgpu.ids > $out/chain_bench.txt
for m in l2_shared_stream vmem_mfma_overlap coissue_bf16; do
  if [ ! -x $root/tools/micro/$m ]; then /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w -o $root/tools/micro/$m $root/tools/micro/$m.hip; fi
  timeout -k 10 120 $root/tools/micro/$m > $out/micro_$m.txt 2>&1 || true
done
rocprofv3 --kernel-trace --stats -d $out/dec -o dec -- python3 $root/tools/decoder_bench.py --captions 64 --steps 50 > $out/dec.log 2>&1
python3 $root/tools/prof_summary.py $out/dec/dec_results.db $out/decoder_kernels.csv 53
rm -rf $out/dec
cd $root
timeout -k 10 300 python3 bench.py --gpus 2 --steps 5 --warmup 2 > $out/rehearsal_2rank_selflaunch.log 2>&1 || true
timeout -k 10 300 python3 bench.py --config joint --gpus 2 --steps 5 --warmup 2 --no-roofline > $out/rehearsal_2rank_joint.log 2>&1 || true
DCAP_GRAD_DTYPE=bf16 timeout -k 10 300 python3 bench.py --config joint --gpus 2 --steps 5 --warmup 2 --no-roofline > $out/rehearsal_2rank_joint_bf16wire.log 2>&1 || true
bash tools/roialign_profile.sh > /dev/null 2>&1 && cp gpurun_out/roialign_profile.txt $out/roialign_hbm.txt && cp gpurun_out/roialign_hbm.json $out/roialign_hbm.json || true
fi
ls $out
