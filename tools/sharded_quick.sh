# kernel stats of the one-rank sharded loop (library engine, no RCCL traffic): bash tools/sharded_quick.sh [method]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; M=${1:-multinomial}; cd /tmp; export TMPDIR=/tmp; rm -rf $R/gpurun_out/prof_sh
python3 $R/tools/sharded_loop.py $M 200 2>/dev/null | tail -1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_sh -- python3 $R/tools/sharded_loop.py $M 30 > /dev/null 2>&1
python3 - <<PY
import csv,glob
f=glob.glob('$R/gpurun_out/prof_sh/*/*kernel_stats.csv')[0]
for r in list(csv.DictReader(open(f)))[:12]:
    print(r['Name'][:60].ljust(62), r['Calls'].rjust(5), f"{float(r['AverageNs'])/1e3:9.2f} us", r['Percentage'])
PY
rm -rf $R/gpurun_out/prof_sh
