cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/r02w; mkdir -p $out
timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -12 | tee $out/tests.txt
timeout 600 python bench.py --no-cpu-baseline 2>&1 | tail -1 > $out/bench_f16.json
timeout 600 python bench.py --dtype fp8 --no-cpu-baseline 2>&1 | tail -1 > $out/bench_fp8.json
python - <<'PY'
import json
for f in ("bench_f16","bench_fp8"):
    d=json.loads(open(f"gpurun_out/r02w/{f}.json").read())
    print(f, d["value"], d["p50_ms_per_image"], d["latency_batch1"]["p50_ms"], d["roofline"]["frac"], d["roofline"]["composite"]["frac"], d.get("roofline_fp8",{}).get("achieved"), {k:(v["achieved"],v["avg_launch_us"]) for k,v in d.items() if k.startswith("roofline_ffn")}, d["roofline_msda"]["frac"])
PY
