cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/r02y; mkdir -p $out
timeout 900 python -m pytest tests/test_fp8_gpu.py tests/test_ffn_gpu.py -m gpu -q -p no:cacheprovider 2>&1 | tail -4 | tee $out/tests.txt
