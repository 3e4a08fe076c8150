# round 3, second GPU call: the sharded / view / bench-launcher tests on the new tree, then the one-rank sharded step in three
# forms -- no communicator, a real 1-rank RCCL communicator with the summaries through the shard mailbox, the same with the
# summaries as RCCL all-gathers (+70 us/step for the three tiny collectives)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; mkdir -p $R/gpurun_out; cd $R
timeout 1500 python -m pytest tests/test_gpu_sharded.py tests/test_views.py tests/test_line_model.py tests/test_gpu_bench_cli.py tests/test_strata.py -m gpu -x -q > gpurun_out/r03b_pytest.log 2>&1; tail -5 gpurun_out/r03b_pytest.log
OUT=$R/gpurun_out/r03b_sharded_one_rank.txt; : > $OUT
for M in multinomial stratified residual; do
  echo "== $M, N = 1e6, 300 steps, library engine (tools/sharded_loop.py)" >> $OUT
  echo -n "no communicator (gathered arrays alias the local ones):      " >> $OUT; python3 tools/sharded_loop.py $M 300 2>/dev/null | tail -1 >> $OUT
  echo -n "1-rank RCCL communicator, summaries through the mailbox:     " >> $OUT; GPF_SHARD_FORCE_COLLECTIVES=1 python3 tools/sharded_loop.py $M 300 2>/dev/null | tail -1 >> $OUT
  echo -n "1-rank RCCL communicator, summaries as RCCL all-gathers:     " >> $OUT; GPF_SHARD_FORCE_COLLECTIVES=1 GPF_SHARD_SUMMARY=rccl python3 tools/sharded_loop.py $M 300 2>/dev/null | tail -1 >> $OUT
done
cat $OUT
