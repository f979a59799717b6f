cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/r02l; mkdir -p $out
timeout 600 python -m pytest tests/test_msda_encoder_gpu.py -m gpu -q -p no:cacheprovider 2>&1 | tail -2
for noise in 0 0.5 1 2 3; do
  timeout 120 python tools/bench_msda_encoder.py --noise $noise --batch 4 2>&1 | tail -1
done | tee $out/msda_flagged.txt
