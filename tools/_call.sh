cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/r02n; mkdir -p $out
timeout 600 python -m pytest tests/test_linear_gpu.py -m gpu -q -p no:cacheprovider 2>&1 | tail -2
for s in 0 1; do echo "== CODETR_GEMM_XS_DEEP=$s"; CODETR_GEMM_XS_DEEP=$s timeout 300 python tools/bench_linear_xs.py 2>&1 | grep -v amdgpu; done | tee $out/xs_deep.txt
