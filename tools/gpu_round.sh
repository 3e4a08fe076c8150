# usage: bash tools/gpu_round.sh TAG   -- the round's evidence in one GPU call: bench line (long run + the driver's command),
# rocprofv3 kernel stats of the same command, FETCH_SIZE / WRITE_SIZE passes, per-config times
set -x
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; mkdir -p $R/gpurun_out; cd $R
python bench.py --steps 1000 --warmup 20 > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; tail -2 gpurun_out/${TAG}_bench.err
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_driver_cmd.json 2>> gpurun_out/${TAG}_bench.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -- python3 $R/bench.py --steps 200 --warmup 10 --no-cpu-baseline > $R/gpurun_out/prof_$TAG.log 2>&1
f=$(find $R/gpurun_out/prof_$TAG -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $R/gpurun_out/${TAG}_kernel_stats.csv && head -12 "$f"
grep "^{" $R/gpurun_out/prof_$TAG.log | tail -1 > $R/gpurun_out/${TAG}_bench_under_rocprof.json
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/gpurun_out/pmc_${TAG}_$C -- python3 $R/bench.py --steps 40 --warmup 5 --no-cpu-baseline > $R/gpurun_out/pmc_${TAG}_$C.log 2>&1
  f=$(find $R/gpurun_out/pmc_${TAG}_$C -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp "$f" $R/gpurun_out/${TAG}_pmc_$C.csv
done
cd $R; python tools/bench_configs.py > gpurun_out/${TAG}_configs.log 2>&1; cut -c1-300 gpurun_out/${TAG}_configs.log
# the driver's scaling command as the driver types it (no torchrun, no WORLD_SIZE), two ranks on this box's one GPU
GPF_BENCH_ONE_DEVICE=1 python bench.py --gpus 2 --steps 20 --warmup 5 --particles-per-gpu 500000 > gpurun_out/${TAG}_bench_two_ranks_one_device.json 2> gpurun_out/${TAG}_bench2.err; echo "rc $?"; cut -c1-300 gpurun_out/${TAG}_bench_two_ranks_one_device.json
# the sharded code path on one rank, three transports
OUT=$R/gpurun_out/${TAG}_sharded_one_rank.txt; : > $OUT
for M in multinomial stratified residual; do
  echo "== $M, N = 1e6, 300 steps, library engine (tools/sharded_loop.py)" >> $OUT
  echo -n "no communicator (gathered arrays alias the local ones):      " >> $OUT; python3 tools/sharded_loop.py $M 300 2>/dev/null | grep "us/step" >> $OUT
  echo -n "1-rank RCCL communicator, summaries through the mailbox:     " >> $OUT; GPF_SHARD_FORCE_COLLECTIVES=1 python3 tools/sharded_loop.py $M 300 2>/dev/null | grep "us/step" >> $OUT
  echo -n "1-rank RCCL communicator, summaries as RCCL all-gathers:     " >> $OUT; GPF_SHARD_FORCE_COLLECTIVES=1 GPF_SHARD_SUMMARY=rccl python3 tools/sharded_loop.py $M 300 2>/dev/null | grep "us/step" >> $OUT
done
cat $OUT
rm -rf gpurun_out/prof_$TAG gpurun_out/pmc_${TAG}_FETCH_SIZE gpurun_out/pmc_${TAG}_WRITE_SIZE
