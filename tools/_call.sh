cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/r02d; mkdir -p $out
timeout 900 python -m pytest tests/test_fp8_gpu.py tests/test_timed_route_gpu.py -k "midsize" -m gpu -q -p no:cacheprovider -x -s > $out/pytest_fp8.log 2>&1; echo "pytest rc $?" >> $out/pytest_fp8.log
grep -E "^(FAILED|ERROR)|passed|failed|^E  |fp8 model errors|rc " $out/pytest_fp8.log | cut -c1-600 | head -40
timeout 600 python bench.py --dtype fp8 --no-cpu-baseline > $out/bench_fp8.json 2> $out/bench_fp8.err; tail -3 $out/bench_fp8.err

python - <<'PY'
import json
for n in ("fp8","f16"):
    try:
        d=json.loads(open(f"gpurun_out/r02d/bench_{n}.json").read().strip().splitlines()[-1])
    except Exception as e:
        print(n,"no json",e); continue
    print(n, d["value"], "img/s", d["p50_ms_per_image"], "ms/img; lat1", d.get("latency_batch1",{}).get("p50_ms"))
    for k in ("roofline","roofline_fp8","roofline_ffn","roofline_msda","roofline_msda_zero_noise"):
        r=d.get(k)
        if r: print("   ",k,r.get("achieved"),r.get("unit"),"frac",r.get("frac"),"sum_ms",r.get("sum_launch_ms"),"avg_us",r.get("avg_launch_us"), r.get("composite",{}).get("frac"))
PY
