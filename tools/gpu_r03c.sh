# round 3, third GPU call: the whole GPU suite on the tree with k_search_fine + 512-thread scans, the headline bench, the scan's
# phase cuts (variants built from tools/scan_probes_r03.patch), the one-rank sharded step in its three transport forms
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; mkdir -p $R/gpurun_out; cd $R
timeout 1800 python -m pytest tests -m gpu -x -q > gpurun_out/r03c_pytest.log 2>&1; tail -5 gpurun_out/r03c_pytest.log
python bench.py --steps 1000 --warmup 20 > gpurun_out/r03c_bench.json 2> gpurun_out/r03c_bench.err; tail -2 gpurun_out/r03c_bench.err; cut -c1-400 gpurun_out/r03c_bench.json
OUT=$R/gpurun_out/r03c_scan_phases.txt; : > $OUT
bash tools/variant_stats.sh $OUT multinomial hip scan_b256 scan_nofold scan_nolookback scan_noexp scan_nostore scan_nolb_nostore scan_all4
bash tools/variant_stats.sh $OUT stratified hip scan_b256
cat $OUT | grep -v "k_iota\|k_init"
cd $R
OUT=$R/gpurun_out/r03c_sharded_one_rank.txt; : > $OUT
for M in multinomial stratified residual; do
  echo "== $M, N = 1e6, 300 steps, library engine (tools/sharded_loop.py)" >> $OUT
  echo -n "no communicator (gathered arrays alias the local ones):      " >> $OUT; python3 tools/sharded_loop.py $M 300 2>/dev/null | grep "us/step" >> $OUT
  echo -n "1-rank RCCL communicator, summaries through the mailbox:     " >> $OUT; GPF_SHARD_FORCE_COLLECTIVES=1 python3 tools/sharded_loop.py $M 300 2>/dev/null | grep "us/step" >> $OUT
  echo -n "1-rank RCCL communicator, summaries as RCCL all-gathers:     " >> $OUT; GPF_SHARD_FORCE_COLLECTIVES=1 GPF_SHARD_SUMMARY=rccl python3 tools/sharded_loop.py $M 300 2>/dev/null | grep "us/step" >> $OUT
done
cat $OUT
