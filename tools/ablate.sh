# build ablation variants of the HIP library here (CPU box): bash tools/ablate.sh FLAG[=VALUE] ...  -> genparticlefilters.jl_amd/libgpf_FLAG.so each
set -e
cd "$(dirname "$0")/.."
for v in "$@"; do bash tools/build_variant.sh "${v%%=*}" -D$v; done
ls genparticlefilters.jl_amd/libgpf_*.so
