# usage (GPU box): bash tools/probe/bench_world1_nccl.sh  -- bench.py's N > 1 code path with the real nccl backend and ONE rank
# (WORLD_SIZE=1 normally skips it): MTR_BENCH_FORCE_DIST=1 makes bench.py treat world 1 as a sharded run.
MTR_BENCH_FORCE_DIST=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29548 bench.py --gpus 1 --steps 400 --no-cpu-baseline --verify
