R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_sharded.py -m gpu -x -q -k "pull or plan" 2>&1 | tail -15
OUT=gpurun_out/r03s_pull_one_rank.txt; : > $OUT
for M in multinomial residual; do
  for P in push pull; do
    echo -n "$M $P no communicator:        " >> $OUT; GPF_SHARD_PLAN=$P python3 tools/sharded_loop.py $M 300 2>/dev/null | grep "us/step" >> $OUT
    echo -n "$M $P 1-rank RCCL, mailbox:   " >> $OUT; GPF_SHARD_PLAN=$P GPF_SHARD_FORCE_COLLECTIVES=1 python3 tools/sharded_loop.py $M 300 2>/dev/null | grep "us/step" >> $OUT
  done
done
cat $OUT
