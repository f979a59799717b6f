#!/bin/bash
# Round-end evidence run (on the GPU box, through gpurun): bench line, kernel traces at 8 images / 1 image per step,
# FETCH_SIZE / WRITE_SIZE passes.  Writes only small folded files under gpurun_out/final/.
#   tools/collect_profiles.sh <tag>      e.g. r01_p
tag=${1:-r01_p}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/final; mkdir -p $out
timeout 900 python bench.py > $out/bench.json 2> $out/bench.err
b8="--streams 1 --steps 3 --warmup 1 --no-cpu-baseline --no-graph --no-roofline"
rm -rf /tmp/tr8 /tmp/tr1 /tmp/pf /tmp/pw
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/tr8 -- python bench.py $b8 > /tmp/tr8.log 2>&1
python tools/fold_trace.py "$(find /tmp/tr8 -name '*.db' | head -1)" $out/${tag}_batch8 8 "rocprofv3 --kernel-trace --stats -- python bench.py $b8"
b1="--batch 1 --steps 6 --warmup 2 --no-cpu-baseline --no-graph --no-roofline"
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/tr1 -- python bench.py $b1 > /tmp/tr1.log 2>&1
python tools/fold_trace.py "$(find /tmp/tr1 -name '*.db' | head -1)" $out/${tag}_batch1 1 "rocprofv3 --kernel-trace --stats -- python bench.py $b1"
p1="--batch 4 --streams 1 --steps 2 --warmup 1 --no-cpu-baseline --no-graph --no-roofline"   # = one replayed graph of the default run
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pf -- python bench.py $p1 > /tmp/pf.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/pw -- python bench.py $p1 > /tmp/pw.log 2>&1
python tools/pmc_traffic.py "$(find /tmp/pf -name '*counter_collection.csv' | head -1)" "$(find /tmp/pw -name '*counter_collection.csv' | head -1)" 0 > $out/pmc_traffic.json
ls -la $out; tail -1 $out/bench.json | cut -c1-600
