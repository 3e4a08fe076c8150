# build ablation variants of the HIP library here (CPU box), run them on the GPU box with tools/ablate_run.sh
set -e
cd /root/repo
mkdir -p genparticlefilters.jl_amd/abl
for v in "$@"; do
  hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fPIC -shared -Wno-unused-value -Iinclude -D$v \
     genparticlefilters.jl_amd/csrc/libgpf.hip -o genparticlefilters.jl_amd/abl/libgpf_$v.so &
done
wait
ls genparticlefilters.jl_amd/abl
