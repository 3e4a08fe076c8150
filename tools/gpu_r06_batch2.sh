cd $GRAFT_REPO_ROOT
export GPF_TAG=r06
bash tools/gpu.sh sharded py:replicas file:tests/test_c_example.py
echo "=== config4 A/B: k_step min waves per SIMD for rows of 8 doubles (4 = product, 3 = libgpf_wide3)"
for V in hip wide3 hip wide3; do
  if [ "$V" = hip ]; then unset GPF_LIB_OVERRIDE; else export GPF_LIB_OVERRIDE=$PWD/genparticlefilters.jl_amd/libgpf_$V.so; fi
  echo "-- $V"; python3 tools/bench_configs.py --no-cpu config4g 2>&1 | cut -c1-420
done > gpurun_out/r06_wide_rows_ab.txt 2>&1
unset GPF_LIB_OVERRIDE
cat gpurun_out/r06_wide_rows_ab.txt
