cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/r02m; mkdir -p $out
# the multi-rank control flow of bench.py on real hardware: 2 ranks sharing the one GPU, gloo collective (test mode)
CODETR_BENCH_SHARE_GPU=1 timeout 900 python bench.py --gpus 2 --batch 2 --steps 4 --warmup 2 --no-cpu-baseline > $out/bench_2ranks_shared.json 2> $out/bench_2ranks.err
echo "rc $?"; tail -3 $out/bench_2ranks.err | cut -c1-300; cut -c1-700 $out/bench_2ranks_shared.json
# and the driver's form: under torch.distributed.run with one rank
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline --no-roofline 2>&1 | tail -1 | cut -c1-300
