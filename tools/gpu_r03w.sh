R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; mkdir -p $R/gpurun_out; cd $R
timeout 1500 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_fuzz.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -4
export GPF_LIB_OVERRIDE=$R/genparticlefilters.jl_amd/libgpf_dbg.so
(python3 tools/strat_debug.py sorted; python3 tools/strat_debug.py unsorted) 2>&1 | grep -v "^RCCL\|Librccl\|amdgpu.ids" | tee gpurun_out/r03w_strat_debug.txt
unset GPF_LIB_OVERRIDE
OUT=$R/gpurun_out/r03w_strat.txt; : > $OUT
bash tools/variant_stats.sh $OUT stratified_sorted hip hip
bash tools/variant_stats.sh $OUT stratified hip
grep "==\|k_search_strat" $OUT
