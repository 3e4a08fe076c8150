cd ${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for f in BASE genparticlefilters.jl_amd/abl/*.so; do
  if [ "$f" = BASE ]; then unset GPF_LIB_OVERRIDE; else export GPF_LIB_OVERRIDE=$PWD/$f; fi
  python - "$f" <<'PY' 2>/dev/null
import sys, json
sys.path.insert(0, '.'); sys.path.insert(0, 'tools')
import bench_configs as b
import io, contextlib
for i in (0, 1):
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        b.run(*b.CONFIGS[i], steps=100, warm=5)
    d = json.loads(buf.getvalue())
    print(sys.argv[1].split('libgpf_')[-1][:28].ljust(28), d['config'][:18], d['us_per_step'], d['kernels_us'])
PY
done
