# round 3, first GPU call: the GPU suite on the tree as it stands + kernel stats and SQ / FETCH / WRITE passes of the kernels that
# carry configs 4, 5 and the residual resampler
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; mkdir -p $R/gpurun_out; cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r03a_pytest.log 2>&1; tail -3 gpurun_out/r03a_pytest.log
cd /tmp; export TMPDIR=/tmp
for C in config4 config5 residual; do
  rm -rf $R/gpurun_out/prof_$C
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$C -- python3 $R/tools/config_loop.py $C 60 > $R/gpurun_out/prof_$C.log 2>&1
  f=$(find $R/gpurun_out/prof_$C -name "*kernel_stats.csv" | head -1)
  cp $f $R/gpurun_out/r03a_${C}_kernel_stats.csv
  head -8 $f
  rm -rf $R/gpurun_out/prof_$C
  LOOP=config_loop.py bash $R/tools/gpu_pmc_kernels.sh r03a_$C $C 40 > /dev/null 2>&1
done
