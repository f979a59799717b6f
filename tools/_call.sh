cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/r02i; mkdir -p $out
timeout 900 python -m pytest tests/test_small_ops_gpu.py tests/test_runner_gpu.py -m gpu -q -p no:cacheprovider -x > $out/pytest_new.log 2>&1; echo "rc $?" >> $out/pytest_new.log
grep -E "^(FAILED|ERROR)|passed|failed|^E  |rc " $out/pytest_new.log | cut -c1-900 | head -30
timeout 900 python -m pytest tests/test_full_size_gpu.py tests/test_mask_pyramid_gpu.py tests/test_cabi_from_c.py -m gpu -q -p no:cacheprovider > $out/pytest_model.log 2>&1; echo "rc $?" >> $out/pytest_model.log
grep -E "^(FAILED|ERROR)|passed|failed|^E  |rc " $out/pytest_model.log | cut -c1-600 | head -30
