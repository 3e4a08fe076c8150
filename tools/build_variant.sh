# usage: bash tools/build_variant.sh NAME -DFLAG=VALUE ...   -> genparticlefilters.jl_amd/libgpf_NAME.so (kernel experiments only;
# run with GPF_LIB_OVERRIDE=$PWD/genparticlefilters.jl_amd/libgpf_NAME.so).  Same four translation units as the product build
# (__graft_entry__.build_hip), objects under build/obj_libgpf_NAME/.
NAME=$1; shift
R=$(cd "$(dirname "$0")/.." && pwd)
cd $R && python3 - "$NAME" "$@" <<'PY'
import sys, os
import __graft_entry__ as ge
name, defs = sys.argv[1], [a[2:] for a in sys.argv[2:] if a.startswith("-D")]
extra = [a for a in sys.argv[2:] if not a.startswith("-D")]
print(ge.build_hip(defines=defs, out=os.path.join(ge.PKG, "libgpf_%s.so" % name), extra_flags=extra))
PY
