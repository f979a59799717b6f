cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/r02z; mkdir -p $out
for dt in fp16 fp8 bf16; do timeout 900 python bench.py --dtype $dt --steps 150 --warmup 5 --no-cpu-baseline --no-roofline --no-host-feed 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$dt', d['value'], d['p50_ms_per_image'], d['p90_ms_per_image'])"; done | tee $out/soak150.txt
