cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/r02k; mkdir -p $out
timeout 600 python __graft_entry__.py smoke 2>&1 | tail -12
timeout 600 python bench.py --dtype bf16 --no-cpu-baseline --no-roofline --no-host-feed --steps 5 --warmup 2 2>&1 | tail -1 | cut -c1-400
