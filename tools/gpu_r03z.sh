R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; mkdir -p $R/gpurun_out; cd $R
export GPF_LIB_OVERRIDE=$R/genparticlefilters.jl_amd/libgpf_dbgs.so
python3 tools/sort_debug.py 2>&1 | grep -v "^RCCL\|Librccl\|amdgpu.ids" | tee gpurun_out/r03z_sort_debug.txt
