R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; mkdir -p $R/gpurun_out; cd /tmp; export TMPDIR=/tmp
D=$R/gpurun_out/prof_b1; rm -rf $D
SMALL_B=1 rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python3 $R/tools/small_filters.py 100 > $D.log 2>&1
f=$(find $D -name "*kernel_stats.csv" | head -1); head -8 $f | cut -c1-90,200-330
grep case $D.log | cut -c1-200
