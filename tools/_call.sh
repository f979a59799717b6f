cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/r02r; mkdir -p $out
timeout 900 python -m pytest tests/test_fp8_gpu.py -m gpu -q -p no:cacheprovider -x -k "ffn" 2>&1 | tail -5 | tee $out/tests.txt
for M in 32768 204600; do for o in 0 3 7; do timeout 60 tools/micro/_bin/ffn8_0 $M $o; done; done 2>&1 | tee $out/ablate5.txt
for M in 32768 204600; do for o in 3 7; do timeout 60 tools/micro/_bin/ffn8_stamps $M $o; done; done 2>&1 | tee $out/stamps2.txt
timeout 300 python tools/bench_ffn.py --fp8 2>&1 | grep -v amdgpu | grep fp8 | tee $out/ffn_fp8.txt
