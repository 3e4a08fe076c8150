cd $GRAFT_REPO_ROOT
export GPF_TAG=r06
( time timeout 1400 python -m pytest tests -m gpu -q --durations=25 ) > gpurun_out/r06_pytest.log 2>&1; tail -40 gpurun_out/r06_pytest.log | cut -c1-200
OUT=gpurun_out/r06_sharded_ess_loop.txt; : > $OUT
for F in 0.5 1.1 0; do
  for MODE in calls one_call; do
    echo -n "ESS < $F N, $MODE, no communicator:        " >> $OUT; python3 tools/sharded_ess_loop.py 300 1000000 $MODE $F 2>/dev/null | grep "us/step" >> $OUT
    echo -n "ESS < $F N, $MODE, 1-rank RCCL + mailbox:  " >> $OUT; GPF_SHARD_FORCE_COLLECTIVES=1 python3 tools/sharded_ess_loop.py 300 1000000 $MODE $F 2>/dev/null | grep "us/step" >> $OUT
  done
done
echo -n "ESS < 0.5 N, one_call, 1-rank RCCL + mailbox, GPF_SHARD_REUSE_SUMMARY=0:  " >> $OUT; GPF_SHARD_REUSE_SUMMARY=0 GPF_SHARD_FORCE_COLLECTIVES=1 python3 tools/sharded_ess_loop.py 300 1000000 one_call 0.5 2>/dev/null | grep "us/step" >> $OUT
echo -n "ESS < 0.5 N, one_call, 1-rank RCCL + mailbox, GPF_LAZY_MOVE=0:            " >> $OUT; GPF_LAZY_MOVE=0 GPF_SHARD_FORCE_COLLECTIVES=1 python3 tools/sharded_ess_loop.py 300 1000000 one_call 0.5 2>/dev/null | grep "us/step" >> $OUT
echo -n "ESS < 1.1 N, one_call, 1-rank RCCL + mailbox, GPF_SHARD_REUSE_SUMMARY=0:  " >> $OUT; GPF_SHARD_REUSE_SUMMARY=0 GPF_SHARD_FORCE_COLLECTIVES=1 python3 tools/sharded_ess_loop.py 300 1000000 one_call 1.1 2>/dev/null | grep "us/step" >> $OUT
echo -n "ESS < 1.1 N, one_call, 1-rank RCCL + mailbox, GPF_LAZY_MOVE=0:            " >> $OUT; GPF_LAZY_MOVE=0 GPF_SHARD_FORCE_COLLECTIVES=1 python3 tools/sharded_ess_loop.py 300 1000000 one_call 1.1 2>/dev/null | grep "us/step" >> $OUT
cat $OUT
