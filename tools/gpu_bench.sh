# usage: bash tools/gpu_bench.sh [tag]   -- bench + rocprofv3 kernel trace on the GPU box
set -x
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p $R/gpurun_out
cd $R
python bench.py --steps 1000 --warmup 20 > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err
cat gpurun_out/bench_$TAG.json; tail -3 gpurun_out/bench_$TAG.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -- python3 $R/bench.py --steps 200 --warmup 10 --no-cpu-baseline > $R/gpurun_out/prof_$TAG.log 2>&1
tail -3 $R/gpurun_out/prof_$TAG.log
find $R/gpurun_out/prof_$TAG -name "*kernel_stats*" | head; 
f=$(find $R/gpurun_out/prof_$TAG -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -20 "$f"
