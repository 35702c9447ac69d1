# launch-contract rehearsal: two ranks sharing the one GPU (gloo), the three bench modes
set -o pipefail
export LDM_BENCH_SHARE_GPU=1 LDM_BENCH_BACKEND=gloo
for mode in "" "--strong" "--train"; do
timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 3 --warmup 1 $mode 2>&1 | grep '"metric"' | cut -c1-330 || exit 1
done
