# functional check of `bench.py --gpus 2` on a 1-GPU box (both ranks on cuda:0, gloo-staged collectives)
cd ${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
GPF_BENCH_ONE_DEVICE=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29544 bench.py --gpus 2 --steps 30 --warmup 3 --particles-per-gpu 200000 2>&1 | tail -3 | cut -c1-900
