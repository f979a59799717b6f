cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/r02h; mkdir -p $out
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > $out/pytest.log 2>&1; echo "pytest rc $?" >> $out/pytest.log
grep -E "^(FAILED|ERROR)|passed|failed|rc " $out/pytest.log | cut -c1-300 | head -20
timeout 900 python bench.py > $out/bench.json 2> $out/bench.err; tail -2 $out/bench.err
timeout 600 python bench.py --dtype fp8 --no-cpu-baseline > $out/bench_fp8.json 2> $out/bench_fp8.err; tail -2 $out/bench_fp8.err
python - <<'PY'
import json
for n in ("bench","bench_fp8"):
    try:
        d=json.loads(open(f"gpurun_out/r02h/{n}.json").read().strip().splitlines()[-1])
    except Exception as e:
        print(n,"no json",e); continue
    print(n, d["value"], "img/s", d["p50_ms_per_image"], "ms/img; lat1", d.get("latency_batch1",{}).get("p50_ms"), "host_feed", d.get("host_feed",{}).get("images_per_s"))
    for k in ("roofline","roofline_fp8","roofline_ffn","roofline_msda","roofline_msda_zero_noise"):
        r=d.get(k)
        if r: print("   ",k,r.get("achieved"),r.get("unit"),"frac",r.get("frac"),"sum_ms",r.get("sum_launch_ms"),"avg_us",r.get("avg_launch_us"), r.get("composite",{}).get("frac"))
    print("   cpu", json.dumps(d.get("cpu_baseline"))[:400])
PY
