# usage: bash tools/gpu_gather_policy.sh TAG   -- the bounded experiment on the i.i.d. row gather of k_step<GATHER>:
# kernel averages of the headline loop for the product library and the -D variants (cache policy of the row load, two rows in flight per
# lane, more workgroups per CU), then the L2 -> fabric request-size counters of the policy variants.  Variants are built beforehand with
# tools/build_variant.sh (pol1 = nt, pol2 = sc1, pol3 = sc0 sc1, pipe, pipe6, sb6, sb8).
TAG=${1:-r05}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; mkdir -p $R/gpurun_out; cd /tmp; export TMPDIR=/tmp
OUT=$R/gpurun_out/${TAG}_gather_policy.txt; : > $OUT
echo "# kernel averages (rocprofv3 --kernel-trace --stats, tools/resample_loop.py multinomial 60; two rounds, alternating)" >> $OUT
for ROUND in 1 2; do
  bash $R/tools/variant_stats.sh $OUT multinomial hip pol1 pol2 pol3 pipe pipe6 sb6 sb8
done
echo "# L2 -> fabric read requests per launch of k_step<GATHER> (TCC_EA0_RDREQ*: 32 / 64 / 128-byte requests), mean over the launches" >> $OUT
for V in hip pol1 pol2 pol3 pipe; do
  if [ "$V" = hip ]; then unset GPF_LIB_OVERRIDE; else export GPF_LIB_OVERRIDE=$R/genparticlefilters.jl_amd/libgpf_$V.so; fi
  D=$R/gpurun_out/gp_$V; rm -rf $D
  rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --output-format csv -d $D -- python3 $R/tools/resample_loop.py multinomial 40 > $D.log 2>&1
  f=$(find $D -name "*counter_collection.csv" | head -1)
  echo "== $V" >> $OUT
  if [ -n "$f" ]; then python3 - "$f" >> $OUT <<'PY'
import csv, sys, collections
agg = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if 'k_step' in r['Kernel_Name']:
        agg[r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in sorted(agg.items()):
    v = v[5:]
    print(f"   {k:28s} launches {len(v):3d} mean {sum(v)/max(len(v),1):12.1f}")
PY
  else echo "   FAILED: $(tail -2 $D.log)" >> $OUT; fi
  rm -rf $D $D.log
done
unset GPF_LIB_OVERRIDE
cat $OUT
