# kernel stats of the sorted-stratified loop (N = 1e6)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd /tmp; export TMPDIR=/tmp; rm -rf $R/gpurun_out/prof_sort
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_sort -- python3 $R/tools/resample_loop.py stratified_sorted 30 > /dev/null 2>&1
python3 - <<PY
import csv,glob
f=glob.glob('$R/gpurun_out/prof_sort/*/*kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    print(r['Name'][:50].ljust(52), r['Calls'].rjust(5), f"{float(r['AverageNs'])/1e3:9.2f} us", r['Percentage'])
PY
rm -rf $R/gpurun_out/prof_sort
