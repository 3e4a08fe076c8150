R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; mkdir -p $R/gpurun_out; cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_line_model.py tests/test_views.py -m gpu -x -q -k "rejuv or move or reweight or line_model or views" > gpurun_out/r03k_pytest.log 2>&1; tail -3 gpurun_out/r03k_pytest.log | cut -c1-200
for V in hip mv3 mv4 mv6 mv8; do
  if [ "$V" = hip ]; then unset GPF_LIB_OVERRIDE; else export GPF_LIB_OVERRIDE=$R/genparticlefilters.jl_amd/libgpf_$V.so; fi
  echo "== $V"; python3 tools/bench_configs.py config4 config5 2>/dev/null | cut -c1-60,90-330
done
