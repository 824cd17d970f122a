# Rehearsal of bench.py's multi-rank code paths on a one-GPU box (timings meaningless):
#   1. two ranks that share the card, collectives over gloo with host staging (RX_BENCH_SHARE_GPU)
#   2. one rank through an RCCL group with the exchanges forced on (RX_BENCH_FORCE_DIST)
set -e
cd $GRAFT_REPO_ROOT
export HSA_ENABLE_IPC_MODE_LEGACY=0
RX_BENCH_BACKEND=gloo RX_BENCH_SHARE_GPU=1 RX_BENCH_SHARDED_ROWS=2000000 RX_BENCH_WATCHDOG=150 timeout -k 10 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 5 --warmup 2 --repeats 2 > gpurun_out/reh_gloo2.json 2> gpurun_out/reh_gloo2.err
echo rehearsal done
RX_BENCH_FORCE_DIST=1 timeout -k 10 200 python -u bench.py --extra none --no-cpu-baseline --steps 20 > gpurun_out/force_dist.json 2> gpurun_out/force_dist.err
echo force-dist done
