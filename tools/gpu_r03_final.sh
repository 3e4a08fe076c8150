# round 3, evidence on the final tree: the GPU suite, the bench line (long run, driver's command, under rocprofv3), FETCH / WRITE passes,
# per-config times, two ranks on one device, the sharded one-rank table (tools/gpu_round.sh), kernel stats + SQ / FETCH / WRITE passes of the
# kernels behind configs 4, 5, residual and sorted stratified, push vs pull on one rank
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; mkdir -p $R/gpurun_out; cd $R
TAG=${1:-r03}
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/${TAG}_pytest.log 2>&1; tail -3 gpurun_out/${TAG}_pytest.log
bash tools/gpu_round.sh $TAG > gpurun_out/${TAG}_round.log 2>&1; tail -30 gpurun_out/${TAG}_round.log | cut -c1-250
cd /tmp; export TMPDIR=/tmp
for C in config4 config5 residual; do
  rm -rf $R/gpurun_out/prof_$C
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$C -- python3 $R/tools/config_loop.py $C 60 > $R/gpurun_out/prof_$C.log 2>&1
  f=$(find $R/gpurun_out/prof_$C -name "*kernel_stats.csv" | head -1)
  cp $f $R/gpurun_out/${TAG}b_${C}_kernel_stats.csv; head -6 $f | cut -c1-160
  rm -rf $R/gpurun_out/prof_$C
  LOOP=config_loop.py bash $R/tools/gpu_pmc_kernels.sh ${TAG}b_$C $C 40 > /dev/null 2>&1
done
rm -rf $R/gpurun_out/prof_sorted
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_sorted -- python3 $R/tools/resample_loop.py stratified_sorted 60 > $R/gpurun_out/prof_sorted.log 2>&1
f=$(find $R/gpurun_out/prof_sorted -name "*kernel_stats.csv" | head -1); cp $f $R/gpurun_out/${TAG}b_sorted_kernel_stats.csv; head -9 $f | cut -c1-160
rm -rf $R/gpurun_out/prof_sorted
bash $R/tools/gpu_pmc_kernels.sh ${TAG}b_sorted stratified_sorted 40 > /dev/null 2>&1
cd $R
OUT=gpurun_out/${TAG}_pull_one_rank.txt; : > $OUT
for M in multinomial residual; do
  for P in push pull; do
    echo -n "$M $P no communicator:        " >> $OUT; GPF_SHARD_PLAN=$P python3 tools/sharded_loop.py $M 300 2>/dev/null | grep "us/step" >> $OUT
    echo -n "$M $P 1-rank RCCL, mailbox:   " >> $OUT; GPF_SHARD_PLAN=$P GPF_SHARD_FORCE_COLLECTIVES=1 python3 tools/sharded_loop.py $M 300 2>/dev/null | grep "us/step" >> $OUT
  done
done
cat $OUT
python tools/small_filters.py 300 2>/dev/null > gpurun_out/${TAG}_small_filters.txt; cut -c1-160 gpurun_out/${TAG}_small_filters.txt
