cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/r02p; mkdir -p $out
timeout 1200 python -m pytest tests/test_fp8_gpu.py tests/test_runner_gpu.py tests/test_window_attention_gpu.py tests/test_mha_attention_gpu.py tests/test_linear_gpu.py tests/test_ffn_gpu.py -m gpu -q -p no:cacheprovider 2>&1 | tail -15 | tee $out/tests.txt
timeout 300 python tools/bench_ffn.py --fp8 2>&1 | grep -v amdgpu | grep fp8 | tee $out/ffn_fp8.txt
timeout 300 python tools/bench_window_attention.py 2>&1 | grep -v amdgpu | tee $out/winattn.txt
timeout 300 python tools/bench_linear_b8.py --fp8 2>&1 | grep -v amdgpu | tail -12 | tee $out/lin_fp8.txt
timeout 600 python bench.py --dtype fp8 2>&1 | tail -1 > $out/bench_fp8.json
timeout 600 python bench.py 2>&1 | tail -1 > $out/bench_f16.json
python - <<'PY'
import json
for f in ("bench_fp8","bench_f16"):
    d=json.loads(open(f"gpurun_out/r02p/{f}.json").read())
    print(f, d["value"], d["p50_ms_per_image"], d["latency_batch1"]["p50_ms"], d["roofline"]["frac"], d.get("roofline_fp8",{}).get("achieved"), {k:v for k,v in d.items() if k.startswith("roofline_ffn")})
PY
