# usage: bash tools/gpu_pmc_gather.sh TAG   -- L1->L2 and L2->fabric request counters of the stand-alone resample gather
# (k_gather<2>: 16-byte rows, N = 1e6) for i.i.d. (multinomial) and monotone (stratified) ancestors; one --pmc pass per group
TAG=${1:-x}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; mkdir -p $R/gpurun_out; cd /tmp; export TMPDIR=/tmp
OUT=$R/gpurun_out/gather_$TAG.txt; : > $OUT
for M in multinomial stratified; do
 i=0
 for C in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" "TCC_READ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1)); D=$R/gpurun_out/gatherdir_${TAG}_${M}_$i; rm -rf $D
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $D -- python3 $R/tools/gather_loop.py $M > $D.log 2>&1
  f=$(find $D -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then python3 - "$f" $M >> $OUT <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(list)
for r in rows:
    if 'k_gather' in r['Kernel_Name']:
        agg[r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in sorted(agg.items()):
    v = v[5:] if len(v) > 10 else v
    print(f"{sys.argv[2]:12s} k_gather<2> {k:28s} launches {len(v):3d} mean {sum(v)/len(v):14.1f}")
PY
  else echo "$M pass $i FAILED: $(tail -2 $D.log)" >> $OUT; fi
  rm -rf $D $D.log
 done
done
cat $OUT
