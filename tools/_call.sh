cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/r02j; mkdir -p $out
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > $out/pytest.log 2>&1; echo "pytest rc $?" >> $out/pytest.log
grep -E "^(FAILED|ERROR)|passed|failed|rc " $out/pytest.log | cut -c1-300 | head -20
timeout 600 python bench.py --offset-noise-px 0 --no-cpu-baseline > $out/bench_0px.json 2> $out/bench_0px.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r02j/bench_0px.json").read().strip().splitlines()[-1])
print("0px", d["value"], "img/s", d["p50_ms_per_image"], "ms/img; lat1", d.get("latency_batch1",{}).get("p50_ms"))
PY
