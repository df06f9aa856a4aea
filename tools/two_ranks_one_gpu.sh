#!/bin/bash
# Functional check of the N > 1 path of bench.py on a 1-GPU box: two gloo ranks share the GPU (weak, strong, strong at N = 1).
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/two_ranks; mkdir -p $O; cd $R
export SED_BENCH_BACKEND=gloo
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --clouds 20 --steps 2 --warmup 1 --no-cpu-baseline --no-k64 --no-extra-legs > $O/weak.out 2> $O/weak.err
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --total-clouds 40 --clouds 20 --steps 2 --warmup 1 --no-cpu-baseline --no-k64 --no-extra-legs > $O/strong.out 2> $O/strong.err
unset SED_BENCH_BACKEND
python bench.py --total-clouds 128 --steps 2 --warmup 1 --no-cpu-baseline --no-k64 --no-extra-legs > $O/strong1.out 2> $O/strong1.err
for f in weak strong strong1; do echo "== $f"; tail -1 $O/$f.out | cut -c1-200; tail -2 $O/$f.err; done
