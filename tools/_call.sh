cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/r02u; mkdir -p $out
timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -8 | tee $out/tests.txt
