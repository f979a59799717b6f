cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/r03a; mkdir -p $out
CODETR_FFN_MFMA32=1 timeout 900 python -m pytest tests/test_ffn_gpu.py -m gpu -q -p no:cacheprovider 2>&1 | tail -12 | tee $out/tests32.txt
for v in 0 1; do echo "== CODETR_FFN_MFMA32=$v"; CODETR_FFN_MFMA32=$v timeout 300 python tools/bench_ffn.py 2>&1 | grep -v amdgpu; done | tee $out/ffn32.txt
