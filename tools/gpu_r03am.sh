R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; mkdir -p $R/gpurun_out; cd $R
SMALL_B=1000,10000 python tools/small_filters.py 300 2>/dev/null | cut -c1-190
TAG=r03
python bench.py --steps 1000 --warmup 20 > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_driver_cmd.json 2>> gpurun_out/${TAG}_bench.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -- python3 $R/bench.py --steps 200 --warmup 10 --no-cpu-baseline > $R/gpurun_out/prof_$TAG.log 2>&1
f=$(find $R/gpurun_out/prof_$TAG -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $R/gpurun_out/${TAG}_kernel_stats.csv
grep "^{" $R/gpurun_out/prof_$TAG.log | tail -1 > $R/gpurun_out/${TAG}_bench_under_rocprof.json
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/gpurun_out/pmc_${TAG}_$C -- python3 $R/bench.py --steps 40 --warmup 5 --no-cpu-baseline > $R/gpurun_out/pmc_${TAG}_$C.log 2>&1
  f=$(find $R/gpurun_out/pmc_${TAG}_$C -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp "$f" $R/gpurun_out/${TAG}_pmc_$C.csv
done
rm -rf $R/gpurun_out/prof_$TAG $R/gpurun_out/pmc_${TAG}_FETCH_SIZE $R/gpurun_out/pmc_${TAG}_WRITE_SIZE
cd $R; tail -c 400 gpurun_out/${TAG}_bench.json
