set -x
cd ${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -30
