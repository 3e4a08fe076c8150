# round-6 soak of the random API sequences on the round's new code paths (window exchange, lazy move over a sharded commit, summary reuse, chain gate, sort_particles=true
# across shards in either engine, checkpoint -> restore among the unsharded operations):
#   bash tools/gpu_soak_r06.sh   -> gpurun_out/r06_fuzz_soak.txt
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd $R
OUT=gpurun_out/r06_fuzz_soak.txt; : > $OUT
echo "== sharded sequences, 2-3 ranks on one GPU, 3 engines x 60 seeds (slabs through the receive windows)" >> $OUT
GPF_FUZZ_SHARD_SEEDS=60 timeout 1500 python -m pytest tests/test_gpu_sharded.py -m "gpu or gpu_soak" -q -x -k random_api 2>&1 | tail -3 >> $OUT
echo "== the same, GPF_SHARD_EXCHANGE=p2p_all (i.i.d. rows through the windows too), 30 seeds" >> $OUT
GPF_SHARD_EXCHANGE=p2p_all GPF_FUZZ_SHARD_SEEDS=30 timeout 1500 python -m pytest tests/test_gpu_sharded.py -m "gpu or gpu_soak" -q -x -k "random_api and library" 2>&1 | tail -3 >> $OUT
echo "== the same, GPF_SHARD_EXCHANGE=rccl GPF_SHARD_REUSE_SUMMARY=0, 20 seeds" >> $OUT
GPF_SHARD_EXCHANGE=rccl GPF_SHARD_REUSE_SUMMARY=0 GPF_FUZZ_SHARD_SEEDS=20 timeout 1500 python -m pytest tests/test_gpu_sharded.py -m "gpu or gpu_soak" -q -x -k "random_api and library" 2>&1 | tail -3 >> $OUT
echo "== unsharded sequences, 800 seeds from offset 600000" >> $OUT
GPF_FUZZ_SEEDS=800 GPF_FUZZ_OFFSET=600000 timeout 2400 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -x 2>&1 | tail -3 >> $OUT
cat $OUT
