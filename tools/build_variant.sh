# usage: bash tools/build_variant.sh NAME -DFLAG=VALUE ...   -> genparticlefilters.jl_amd/libgpf_NAME.so (kernel experiments only;
# run with GPF_LIB_OVERRIDE=$PWD/genparticlefilters.jl_amd/libgpf_NAME.so)
NAME=$1; shift
R=$(cd "$(dirname "$0")/.." && pwd)
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fPIC -shared -Wno-unused-value -I$R/include "$@" \
  $R/genparticlefilters.jl_amd/csrc/libgpf.hip -o $R/genparticlefilters.jl_amd/libgpf_$NAME.so
