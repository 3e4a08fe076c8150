R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; mkdir -p $R/gpurun_out; cd $R
python tools/dbg_residual.py 2>&1 | grep -v "RCCL\|Librccl\|amdgpu"
cd /tmp; export TMPDIR=/tmp
for C in config4; do
  rm -rf $R/gpurun_out/prof_$C
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$C -- python3 $R/tools/config_loop.py $C 60 > $R/gpurun_out/prof_$C.log 2>&1
  f=$(find $R/gpurun_out/prof_$C -name "*kernel_stats.csv" | head -1); head -7 $f | cut -c1-100,300-420
done
