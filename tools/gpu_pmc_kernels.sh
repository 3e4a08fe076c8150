# usage: [LOOP=sharded_loop.py] bash tools/gpu_pmc_kernels.sh TAG METHOD [STEPS] [N]   -- SQ counters of every kernel of a resample+update loop
# (one rocprofv3 --pmc pass per counter group, kernel trace only; summaries -> gpurun_out/pmc_TAG.txt; every pass under its own timeout: a pass with
#  TA_* counters hung for 25 minutes on this pool -- PMC_EXTRA / PMC_EXTRA2 / PMC_EXTRA3 add counter groups)
TAG=${1:-x}; METHOD=${2:-multinomial}; STEPS=${3:-30}; NP=${4:-}     # (no N: the loop's own size for that config)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; mkdir -p $R/gpurun_out; cd /tmp; export TMPDIR=/tmp
OUT=$R/gpurun_out/pmc_$TAG.txt; : > $OUT
i=0
for C in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE" ${PMC_EXTRA:+"$PMC_EXTRA"} ${PMC_EXTRA2:+"$PMC_EXTRA2"} ${PMC_EXTRA3:+"$PMC_EXTRA3"}; do
  i=$((i+1)); D=$R/gpurun_out/pmcdir_${TAG}_$i; rm -rf $D
  timeout ${PMC_PASS_TIMEOUT:-240} rocprofv3 --kernel-trace --pmc $C --output-format csv -d $D -- python3 $R/tools/${LOOP:-resample_loop.py} $METHOD $STEPS $NP > $D.log 2>&1
  f=$(find $D -name "*counter_collection.csv" | head -1)
  echo "== pass $i: $C" >> $OUT
  if [ -n "$f" ]; then python3 - "$f" >> $OUT <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(list)
for r in rows:
    agg[(r['Kernel_Name'].split('(')[0][:70], r['Counter_Name'])].append(float(r['Counter_Value']))
for k, v in sorted(agg.items()):
    if len(v) >= 5:
        print(f"{k[0]:72s} {k[1]:24s} launches {len(v):4d} mean {sum(v)/len(v):14.1f}")
PY
  else echo "FAILED: $(tail -2 $D.log)" >> $OUT; fi
  rm -rf $D
done
cat $OUT
