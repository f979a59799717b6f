#!/bin/bash
# Round evidence run (on the GPU box, through gpurun): bench lines, kernel traces at 8 images / 1 image per step, PMC
# passes (each in its own run, never with a trace).  Writes only small folded files under gpurun_out/final/.
#   tools/collect_profiles.sh <tag>      e.g. r04
tag=${1:-r06}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/final; mkdir -p $out
if [ -z "$SKIP_BENCH" ]; then
timeout 900 python bench.py > $out/${tag}_bench.json 2> $out/bench.err
timeout 600 python bench.py --dtype fp8 --no-cpu-baseline > $out/${tag}_bench_fp8.json 2>> $out/bench.err
fi
# (--no-fp8-line: the default line's fp8 sub-record would put its calibration forward and e4m3 kernels into the fp16 traces)
common="--no-cpu-baseline --no-graph --no-roofline --no-host-feed --no-fp8-line --pad 1.0"
b8="--streams 1 --steps 3 --warmup 1 $common"
rm -rf /tmp/tr8 /tmp/tr1 /tmp/pf /tmp/pw /tmp/pm
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/tr8 -- python bench.py $b8 > /tmp/tr8.log 2>&1
python tools/fold_trace.py "$(find /tmp/tr8 -name '*.db' | head -1)" $out/${tag}_batch8 8 "rocprofv3 --kernel-trace --stats -- python bench.py $b8"
b1="--batch 1 --steps 6 --warmup 2 $common"
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/tr1 -- python bench.py $b1 > /tmp/tr1.log 2>&1
python tools/fold_trace.py "$(find /tmp/tr1 -name '*.db' | head -1)" $out/${tag}_batch1 1 "rocprofv3 --kernel-trace --stats -- python bench.py $b1"
# one replayed graph's launch shapes (4 images): the kernel trace bench.py's live rooflines must agree with, then the HBM traffic
p4="--batch 4 --streams 1 --steps 2 --warmup 1 $common"
rm -rf /tmp/tr4
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/tr4 -- python bench.py $p4 > /tmp/tr4.log 2>&1
python tools/fold_trace.py "$(find /tmp/tr4 -name '*.db' | head -1)" $out/${tag}_batch4 4 "rocprofv3 --kernel-trace --stats -- python bench.py $p4"
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pf -- python bench.py $p4 > /tmp/pf.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/pw -- python bench.py $p4 > /tmp/pw.log 2>&1
python tools/pmc_fold.py --fetch "$(find /tmp/pf -name '*counter_collection.csv' | head -1)" --write "$(find /tmp/pw -name '*counter_collection.csv' | head -1)" \
   --label "bench.py $p4 (4 images = one replayed graph of the default run)" > $out/${tag}_pmc_traffic.json
# ... and the matrix-pipe busy counters at the same launch shapes (own pass: counters never share a run with a trace)
rm -rf /tmp/pm4
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/pm4 -- python bench.py $p4 > /tmp/pm4.log 2>&1
python tools/pmc_fold.py --fetch "$(find /tmp/pf -name '*counter_collection.csv' | head -1)" --write "$(find /tmp/pw -name '*counter_collection.csv' | head -1)" \
   --mfma "$(find /tmp/pm4 -name '*counter_collection.csv' | head -1)" --stats $out/${tag}_batch4_kernel_stats.csv \
   --label "bench.py $p4 (4 images = one replayed graph of the default run)" > $out/${tag}_pmc_b4_fp16.json
# BASELINE config 3 (1152x768): HBM GB/s + MFMA busy per kernel group
c3="--res 1152x768 --batch 1 --steps 4 --warmup 1 $common"
rm -rf /tmp/pf /tmp/pw /tmp/pm /tmp/tr3
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/tr3 -- python bench.py $c3 > /tmp/tr3.log 2>&1
python tools/fold_trace.py "$(find /tmp/tr3 -name '*.db' | head -1)" $out/${tag}_1152x768_batch1 1 "rocprofv3 --kernel-trace --stats -- python bench.py $c3"
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pf -- python bench.py $c3 > /tmp/pf.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/pw -- python bench.py $c3 > /tmp/pw.log 2>&1
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/pm -- python bench.py $c3 > /tmp/pm.log 2>&1
python tools/pmc_fold.py --fetch "$(find /tmp/pf -name '*counter_collection.csv' | head -1)" --write "$(find /tmp/pw -name '*counter_collection.csv' | head -1)" \
   --mfma "$(find /tmp/pm -name '*counter_collection.csv' | head -1)" --stats $out/${tag}_1152x768_batch1_kernel_stats.csv \
   --label "bench.py $c3 (BASELINE config 3)" > $out/${tag}_pmc_1152x768.json
if [ -z "$SKIP_BENCH" ]; then
timeout 600 python tools/export_and_run_plan.py --res 1920x1280 --batch 1 > $out/${tag}_runner_1920x1280.json 2> $out/runner.err
fi
# round 6: the public op (timings over the offset spread, counters), the Swin MLP fusion, the GEMM yardstick against hipBLASLt
timeout 300 python tools/bench_msda_op.py 2>&1 | grep -v amdgpu.ids > $out/${tag}_msda_op.txt
rm -f $out/${tag}_msda_op_pmc.txt
bash tools/pmc_msda_op.sh $out/${tag}_msda_op_pmc.txt 3
python tools/fold_msda_op_pmc.py $out/${tag}_msda_op_pmc.txt > $out/${tag}_msda_op_pmc.json
timeout 300 python tools/bench_swin_mlp.py 2>&1 | grep -v amdgpu.ids > $out/${tag}_swin_mlp.txt
(timeout 400 python tools/bench_linear_vs_lib.py --images 4; timeout 400 python tools/bench_linear_vs_lib.py --images 8) 2>&1 | grep -v amdgpu.ids > $out/${tag}_linear_vs_hipblaslt.txt
ls -la $out; tail -1 $out/${tag}_bench.json | cut -c1-300; cat $out/${tag}_runner_1920x1280.json | cut -c1-600
