# usage: bash tools/variant_stats.sh OUTFILE METHOD VARIANT...   -- rocprofv3 kernel stats (avg us per launch) of tools/resample_loop.py METHOD 60
# for the product library ("hip") and for -D variants built with tools/build_variant.sh (genparticlefilters.jl_amd/libgpf_VARIANT.so)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; OUT=$1; M=$2; shift 2
mkdir -p $R/gpurun_out; cd /tmp; export TMPDIR=/tmp
for V in "$@"; do
  D=$R/gpurun_out/vs_$V; rm -rf $D
  if [ "$V" = hip ]; then unset GPF_LIB_OVERRIDE; else export GPF_LIB_OVERRIDE=$R/genparticlefilters.jl_amd/libgpf_$V.so; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python3 $R/tools/resample_loop.py $M 60 > $D.log 2>&1
  f=$(find $D -name "*kernel_stats.csv" | head -1)
  echo "== $V ($M)" >> $OUT
  python3 - "$f" >> $OUT <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if int(r['Calls']) >= 30:
        print(f"   {r['Name'].split('(')[0].replace('void gpf::','').replace('gpf::','')[:48]:50s} {int(r['Calls']):5d} {float(r['AverageNs'])/1e3:9.2f} us")
PY
  rm -rf $D $D.log
done
unset GPF_LIB_OVERRIDE
