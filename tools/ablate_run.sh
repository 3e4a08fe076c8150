cd ${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
python bench.py --steps 300 --warmup 20 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('BASE', d['ms_per_step'], d['roofline']['all_kernels_us'])"
for f in genparticlefilters.jl_amd/abl/*.so; do
  GPF_LIB_OVERRIDE=$PWD/$f python bench.py --steps 300 --warmup 20 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$f'.split('libgpf_')[-1], d['ms_per_step'], d['roofline']['all_kernels_us'])"
done
