R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; mkdir -p $R/gpurun_out; cd /tmp; export TMPDIR=/tmp
export GPF_SHARD_FORCE_COLLECTIVES=1 GPF_BENCH_FORCE_SHARDED=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0
rm -rf $R/gpurun_out/prof_sharded_coll
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_sharded_coll -- python3 $R/bench.py --steps 100 --warmup 5 --no-cpu-baseline > $R/gpurun_out/prof_sharded_coll.log 2>&1
f=$(find $R/gpurun_out/prof_sharded_coll -name "*kernel_stats.csv" | head -1); head -16 "$f" | cut -c1-180
