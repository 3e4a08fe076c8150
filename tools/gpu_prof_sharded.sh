R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; mkdir -p $R/gpurun_out; cd /tmp; export TMPDIR=/tmp
export GPF_BENCH_FORCE_SHARDED=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_sharded -- python3 $R/bench.py --steps 100 --warmup 5 --no-cpu-baseline > $R/gpurun_out/prof_sharded.log 2>&1
tail -2 $R/gpurun_out/prof_sharded.log | cut -c1-300
f=$(find $R/gpurun_out/prof_sharded -name "*kernel_stats.csv" | head -1); head -30 "$f" | cut -c1-200
