cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/r02z; mkdir -p $out
common="--no-cpu-baseline --no-graph --no-roofline --no-host-feed"
p4="--batch 4 --streams 1 --steps 2 --warmup 1 $common"
for dt in fp16 fp8; do
  rm -rf /tmp/pf /tmp/pw /tmp/pm /tmp/tr4
  timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/tr4 -- python bench.py $p4 --dtype $dt > /tmp/tr4.log 2>&1
  python tools/fold_trace.py "$(find /tmp/tr4 -name '*.db' | head -1)" $out/b4_$dt 4 "rocprofv3 --kernel-trace --stats -- python bench.py $p4 --dtype $dt"
  timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pf -- python bench.py $p4 --dtype $dt > /tmp/pf.log 2>&1
  timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/pw -- python bench.py $p4 --dtype $dt > /tmp/pw.log 2>&1
  timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/pm -- python bench.py $p4 --dtype $dt > /tmp/pm.log 2>&1
  python tools/pmc_fold.py --fetch "$(find /tmp/pf -name '*counter_collection.csv' | head -1)" --write "$(find /tmp/pw -name '*counter_collection.csv' | head -1)" \
     --mfma "$(find /tmp/pm -name '*counter_collection.csv' | head -1)" --stats $out/b4_${dt}_kernel_stats.csv \
     --label "bench.py $p4 --dtype $dt (4 images = one replayed graph of the default run)" > $out/pmc_b4_$dt.json
done
ls -la $out
