cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/r02x; mkdir -p $out
timeout 1200 python -m pytest tests/test_linear_gpu.py tests/test_patch_embed_gpu.py -m gpu -q -p no:cacheprovider 2>&1 | tail -8 | tee $out/tests.txt
timeout 300 python tools/bench_linear_xs.py 2>&1 | grep -v amdgpu | tee $out/xs.txt
