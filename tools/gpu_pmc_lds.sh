# LDS / VALU utilisation counters of the hot kernels (one --pmc pass per group)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; mkdir -p $R/gpurun_out; cd /tmp; export TMPDIR=/tmp
i=0
for C in "SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS" "SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVE_CYCLES" "SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "GRBM_GUI_ACTIVE SQ_WAVES"; do
  i=$((i+1))
  rm -rf $R/gpurun_out/pmc_lds_$i
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/gpurun_out/pmc_lds_$i -- python3 $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline > $R/gpurun_out/pmc_lds_$i.log 2>&1
  f=$(find $R/gpurun_out/pmc_lds_$i -name "*counter_collection.csv" | head -1)
  echo "== $C"
  [ -n "$f" ] && python3 - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(list)
for r in rows:
    k = r['Kernel_Name']
    if 'k_search<0>' in k or 'k_step<1, 2, false, true' in k or 'k_scan<gpf::InFixQ, 1>' in k:
        agg[(k.split('(')[0][:40], r['Counter_Name'])].append(float(r['Counter_Value']))
for k, v in sorted(agg.items()):
    print(k, 'launches', len(v), 'mean', round(sum(v)/len(v), 1))
PY
  [ -z "$f" ] && tail -3 $R/gpurun_out/pmc_lds_$i.log
done
