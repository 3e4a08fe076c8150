# usage: bash tools/gpu_pmc_requests.sh TAG [METHOD]   -- L2 -> fabric request-size counters of EVERY kernel of the headline loop
# (tools/resample_loop.py METHOD: k_step<GATHER>, k_scan, k_search_multi at N = 1e6): calibrates the 2 x FETCH_SIZE correction of
# profiles/pmc_traffic.json on the fused kernel itself (round 2 calibrated it on the stand-alone k_gather only).  One --pmc pass per group.
TAG=${1:-x}; METHOD=${2:-multinomial}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; mkdir -p $R/gpurun_out; cd /tmp; export TMPDIR=/tmp
OUT=$R/gpurun_out/requests_$TAG.txt; : > $OUT
i=0
for C in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" "TCC_READ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1)); D=$R/gpurun_out/reqdir_${TAG}_$i; rm -rf $D
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $D -- python3 $R/tools/resample_loop.py $METHOD 40 > $D.log 2>&1
  f=$(find $D -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then python3 - "$f" >> $OUT <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(list)
for r in rows:
    agg[(r['Kernel_Name'].split('(')[0].replace('void gpf::', '').replace('gpf::', '')[:44], r['Counter_Name'])].append(float(r['Counter_Value']))
for k, v in sorted(agg.items()):
    if len(v) >= 20:
        v = v[5:]
        print(f"{k[0]:46s} {k[1]:28s} launches {len(v):3d} mean {sum(v)/len(v):14.1f}")
PY
  else echo "pass $i FAILED: $(tail -2 $D.log)" >> $OUT; fi
  rm -rf $D $D.log
done
cat $OUT
