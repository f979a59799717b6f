cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/r02x; mkdir -p $out
for cfg in "--batch 8 --streams 1" "--batch 8 --streams 2" "--batch 8 --streams 4" "--batch 16 --streams 2" "--batch 12 --streams 3"; do
  echo "== $cfg"; timeout 300 python bench.py $cfg --no-cpu-baseline --no-roofline --no-host-feed 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['p50_ms_per_image'], d['ms_per_step'])"
done 2>&1 | tee $out/streams.txt
