# counter spacing: k_push_scan (count_us) at simulated shard counts, counters 128 B apart (default) vs contiguous (cs1)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
OUT=gpurun_out/r03r_push_counters.txt; : > $OUT
for V in hip cs1; do
  echo "== libgpf_$V.so" >> $OUT
  GPF_LIB_OVERRIDE=$R/genparticlefilters.jl_amd/libgpf_$V.so PUSH_G=1,2,4,8,16 python3 tools/push_bench.py 2>/dev/null | grep "^{" >> $OUT
done
cat $OUT
python -m pytest tests/test_gpu_sharded.py -m gpu -x -q 2>&1 | tail -3
