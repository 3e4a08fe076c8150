# PMC passes (separate runs, kernel-trace only): HBM-side bytes per kernel
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/gpurun_out/pmc_${TAG}_$C -- python3 $R/bench.py --steps 40 --warmup 5 --no-cpu-baseline > $R/gpurun_out/pmc_${TAG}_$C.log 2>&1
  tail -2 $R/gpurun_out/pmc_${TAG}_$C.log
  f=$(find $R/gpurun_out/pmc_${TAG}_$C -name "*counter_collection.csv" | head -1)
  echo "== $C $f"
  python3 - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(list)
for r in rows:
    agg[(r['Kernel_Name'].split('(')[0][:60], r['Counter_Name'])].append(float(r['Counter_Value']))
for k, v in sorted(agg.items()):
    print(k, 'launches', len(v), 'mean', sum(v)/len(v))
PY
done
