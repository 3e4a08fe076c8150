R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; mkdir -p $R/gpurun_out; cd $R
OUT=$R/gpurun_out/r03n_ns_variants.txt; : > $OUT
bash tools/variant_stats.sh $OUT multinomial ns2 hip ns2 hip
grep "==\|k_search" $OUT
for V in hip ns2; do
  if [ "$V" = hip ]; then unset GPF_LIB_OVERRIDE; else export GPF_LIB_OVERRIDE=$R/genparticlefilters.jl_amd/libgpf_$V.so; fi
  echo "== $V"; python3 tools/bench_configs.py config2 config5 2>/dev/null | cut -c1-40,80-300
  python3 tools/sharded_loop.py multinomial 300 2>/dev/null | grep us/step
  python3 bench.py --steps 500 --warmup 20 --no-cpu-baseline 2>/dev/null | cut -c1-170
done
