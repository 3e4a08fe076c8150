# usage: bash tools/gpu_fuzz_hunt.sh SEEDS OFFSET   -- a longer run of the random-sequence tests
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd $R
GPF_FUZZ_SEEDS=${1:-150} GPF_FUZZ_OFFSET=${2:-0} timeout 2400 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -x 2>&1 | tail -15
