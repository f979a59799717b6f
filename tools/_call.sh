cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/r02b; mkdir -p $out
timeout 900 python -m pytest tests/test_msda_backward_gpu.py tests/test_timed_route_gpu.py tests/test_linear_gpu.py tests/test_model_gpu.py -m gpu -q -p no:cacheprovider > $out/pytest.log 2>&1; echo "pytest rc $?" >> $out/pytest.log
grep -E "^(FAILED|ERROR)|passed|failed|^E  " $out/pytest.log | cut -c1-400 | head -40
timeout 300 python tools/diag_fp32_flake.py --same 12 --seeds 100 --out $out/diag_fp32_flake.json > $out/diag.log 2>&1; tail -8 $out/diag.log | cut -c1-1200
timeout 600 python tools/probe_cpu_oracle_threads.py 384x384 > $out/cpu_threads.log 2>&1; cat $out/cpu_threads.log
b8="--streams 1 --steps 3 --warmup 1 --no-cpu-baseline --no-graph --no-roofline --no-host-feed"
rm -rf /tmp/tr8
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/tr8 -- python bench.py $b8 > /tmp/tr8.log 2>&1
python tools/fold_trace.py "$(find /tmp/tr8 -name '*.db' | head -1)" $out/r02b_batch8 8 "rocprofv3 --kernel-trace --stats -- python bench.py $b8"
head -60 $out/r02b_batch8_summary.txt
