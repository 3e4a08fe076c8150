cd $GRAFT_REPO_ROOT
export GPF_TAG=r06
timeout 900 python -m pytest tests/test_gpu_sharded.py tests/test_lazy_move.py tests/test_step_ess.py tests/test_gpu_fullsize.py -m gpu -q -x --durations=15 > gpurun_out/r06_pytest_batch3.log 2>&1; tail -25 gpurun_out/r06_pytest_batch3.log | cut -c1-200
bash tools/gpu.sh py:replicas
OUT=gpurun_out/r06_sharded_ess_loop.txt; : > $OUT
for F in 0.5 1.1 0; do
  for MODE in calls one_call; do
    echo -n "ESS < $F N, $MODE, no communicator:        " >> $OUT; python3 tools/sharded_ess_loop.py 300 1000000 $MODE $F 2>/dev/null | tail -1 >> $OUT
    echo -n "ESS < $F N, $MODE, 1-rank RCCL + mailbox:  " >> $OUT; GPF_SHARD_FORCE_COLLECTIVES=1 python3 tools/sharded_ess_loop.py 300 1000000 $MODE $F 2>/dev/null | tail -1 >> $OUT
  done
done
cat $OUT
