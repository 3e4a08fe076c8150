# tools/gpu.sh -- the ONE script behind this repository's gpurun calls (replaces the per-call gpu_r03*.sh files of round 3).
#
#   gpurun --timeout S -- 'bash tools/gpu.sh STEP [STEP ...]'      steps run in order; every output lands under gpurun_out/
#
# steps (TAG = $GPF_TAG, default r06):
#   smoke                 __graft_entry__.smoke()
#   tests[:EXPR]          pytest -m gpu (optionally -k EXPR) -> gpurun_out/TAG_pytest.log
#   soak                  pytest -m "gpu or gpu_soak": the driver's set PLUS the wide multi-process matrices (tests/conftest.py soak_grid) -> TAG_pytest_soak.log
#   file:PATH[:EXPR]      pytest -m gpu on one test file
#   bench[:STEPS]         python bench.py --steps STEPS (default 1000) --warmup 20 -> TAG_bench.json
#   driver                the driver's own command: python bench.py --gpus 1 --steps 20 --warmup 5 -> TAG_bench_driver_cmd.json
#   prof                  rocprofv3 --kernel-trace --stats of bench.py --headline-only (200 steps) -> TAG_kernel_stats.csv, and of
#                         tools/resample_loop.py for every named variant -> TAG_<variant>_kernel_stats.csv
#   pmc                   FETCH_SIZE / WRITE_SIZE passes of bench.py (one counter per pass) -> TAG_pmc_*.csv (+ tools/pmc_to_json.py TAG)
#   configs               tools/bench_configs.py (the other BASELINE configs at their per-GPU sizes) -> TAG_configs.jsonl
#   loop:NAME:ARGS        rocprofv3 kernel stats of `python3 tools/NAME.py ARGS` (ARGS comma-separated) -> TAG_NAME_ARGS_kernel_stats.csv
#   gaps:NAME:ARGS        kernel trace of the same loop; a window of it with the idle gap in front of every kernel (tools/trace_gaps.py) -> TAG_NAME_ARGS_gaps.txt
#   sq:NAME:ARGS          SQ wait / VALU / LDS counters of the same loop (tools/gpu_pmc_kernels.sh) -> TAG_sq_NAME_ARGS
#   sharded               one-rank sharded table (tools/sharded_loop.py; no communicator / 1-rank RCCL) -> TAG_sharded_one_rank.txt
#   essloop               the ESS-triggered loop of BASELINE configs[3] through the sharded path on one rank (tools/sharded_ess_loop.py): separate calls / one call,
#                         no communicator / 1-rank RCCL + mailboxes, ESS < N/2 / always / never; and without the summary reuse / the lazy move -> TAG_sharded_ess_loop.txt
#   tworanks              bench.py --gpus 2 --steps 20 with both ranks on this GPU over tests/loopback_rccl: the SHAPE of a multi-GPU line -> TAG_bench_2ranks_one_gpu_loopback.json
#   variant:OUT:METHOD:DEFS   tools/variant_stats.sh OUT METHOD DEFS (DEFS: comma-separated -D sets, alternating A/B rocprofv3 runs)
#   py:SCRIPT:ARGS        python3 tools/SCRIPT.py ARGS -> TAG_SCRIPT.txt
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${GPF_TAG:-r06}
mkdir -p $R/gpurun_out
export TMPDIR=/tmp
stats_of() {   # stats_of DIR OUT: copy the kernel-stats csv of a rocprofv3 output directory
  f=$(find "$1" -name "*kernel_stats.csv" | head -1)
  if [ -n "$f" ]; then cp "$f" "$2"; head -12 "$f" | cut -c1-170; else echo "no kernel stats under $1"; fi
}
for STEP in "$@"; do
  IFS=: read -r S A1 A2 A3 <<< "$STEP"
  echo "=== $STEP"
  cd $R
  case $S in
    smoke)   python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 ;;
    tests)   ( time timeout 3000 python -m pytest tests -m gpu -q --durations=25 ${A1:+-k "$A1"} ) > gpurun_out/${TAG}_pytest.log 2>&1; tail -15 gpurun_out/${TAG}_pytest.log | cut -c1-220 ;;
    soak)    timeout 3000 python -m pytest tests -m "gpu or gpu_soak" -q > gpurun_out/${TAG}_pytest_soak.log 2>&1; tail -8 gpurun_out/${TAG}_pytest_soak.log | cut -c1-220 ;;
    file)    timeout 3000 python -m pytest "$A1" -m gpu -x -q ${A2:+-k "$A2"} > gpurun_out/${TAG}_pytest_$(basename $A1 .py).log 2>&1; tail -25 gpurun_out/${TAG}_pytest_$(basename $A1 .py).log | cut -c1-220 ;;
    bench)   python bench.py --steps ${A1:-1000} --warmup 20 > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; cut -c1-3000 gpurun_out/${TAG}_bench.json; tail -3 gpurun_out/${TAG}_bench.err ;;
    driver)  python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_driver_cmd.json 2> gpurun_out/${TAG}_bench_driver_cmd.err; cut -c1-600 gpurun_out/${TAG}_bench_driver_cmd.json ;;
    prof)    # the headline loop ALONE (k_step<...GATHER> is also launched by the sorted variants: a mixed file misstates its average), then one
             # stats file per named variant
             cd /tmp; rm -rf $R/gpurun_out/prof_$TAG
             rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -- python3 $R/bench.py --steps 200 --warmup 10 --headline-only > $R/gpurun_out/${TAG}_bench_under_rocprof.json 2> $R/gpurun_out/prof_$TAG.err
             stats_of $R/gpurun_out/prof_$TAG $R/gpurun_out/${TAG}_kernel_stats.csv; rm -rf $R/gpurun_out/prof_$TAG
             for V in multinomial_sorted stratified stratified_sorted residual; do
               D=$R/gpurun_out/prof_${TAG}_$V; rm -rf $D
               rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python3 $R/tools/resample_loop.py $V 200 > $D.log 2>&1
               stats_of $D $R/gpurun_out/${TAG}_${V}_kernel_stats.csv | head -6; rm -rf $D
             done ;;
    pmc)     cd /tmp
             for C in FETCH_SIZE WRITE_SIZE; do
               rm -rf $R/gpurun_out/pmc_${TAG}_$C
               rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/gpurun_out/pmc_${TAG}_$C -- python3 $R/bench.py --steps 40 --warmup 5 --headline-only > $R/gpurun_out/pmc_${TAG}_$C.log 2>&1
               f=$(find $R/gpurun_out/pmc_${TAG}_$C -name "*counter_collection.csv" | head -1)
               [ -n "$f" ] && cp "$f" $R/gpurun_out/${TAG}_pmc_$C.csv
               rm -rf $R/gpurun_out/pmc_${TAG}_$C
             done
             ls -la $R/gpurun_out/${TAG}_pmc_*.csv ;;   # (then, locally: cp gpurun_out/TAG_pmc_*.csv profiles/ && python tools/pmc_to_json.py TAG)
    configs) python tools/bench_configs.py > gpurun_out/${TAG}_configs.jsonl 2> gpurun_out/${TAG}_configs.err; cut -c1-400 gpurun_out/${TAG}_configs.jsonl ;;
    loop)    cd /tmp; D=$R/gpurun_out/prof_${A1}_${A2//,/_}; rm -rf $D
             rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python3 $R/tools/$A1.py ${A2//,/ } > $D.log 2>&1; tail -2 $D.log | cut -c1-200
             stats_of $D $R/gpurun_out/${TAG}_${A1}_${A2//,/_}_kernel_stats.csv; rm -rf $D ;;
    gaps)    cd /tmp; D=$R/gpurun_out/gaps_${A1}_${A2//,/_}; rm -rf $D
             rocprofv3 --kernel-trace --output-format csv -d $D -- python3 $R/tools/$A1.py ${A2//,/ } > $D.log 2>&1; tail -1 $D.log | cut -c1-200
             python3 $R/tools/trace_gaps.py $D 0.6 ${A3:-40} > $R/gpurun_out/${TAG}_${A1}_${A2//,/_}_gaps.txt; cat $R/gpurun_out/${TAG}_${A1}_${A2//,/_}_gaps.txt; rm -rf $D ;;
    sq)      LOOP=$A1.py bash $R/tools/gpu_pmc_kernels.sh ${TAG}_sq_${A1}_${A2//,/_} ${A2//,/ } 2>&1 | tail -40 | cut -c1-200 ;;
    sharded) OUT=gpurun_out/${TAG}_sharded_one_rank.txt; : > $OUT
             for M in multinomial stratified residual multinomial_sorted; do
               echo -n "$M, no communicator:      " >> $OUT; python3 tools/sharded_loop.py $M 300 2>/dev/null | grep "us/step" >> $OUT
               echo -n "$M, 1-rank RCCL, mailbox: " >> $OUT; GPF_SHARD_FORCE_COLLECTIVES=1 python3 tools/sharded_loop.py $M 300 2>/dev/null | grep "us/step" >> $OUT
               case $M in stratified|multinomial_sorted)   # (the line above ran the window exchange and -- a rank with a device of its own -- the fused (max, flags) round; these: the grouped send / receive, the separate k_pack_mflags launch)
                 echo -n "$M, 1-rank RCCL, mailbox, GPF_SHARD_EXCHANGE=rccl: " >> $OUT; GPF_SHARD_EXCHANGE=rccl GPF_SHARD_FORCE_COLLECTIVES=1 python3 tools/sharded_loop.py $M 300 2>/dev/null | grep "us/step" >> $OUT
                 echo -n "$M, 1-rank RCCL, mailbox, GPF_SHARD_FUSE_MF=0:     " >> $OUT; GPF_SHARD_FUSE_MF=0 GPF_SHARD_FORCE_COLLECTIVES=1 python3 tools/sharded_loop.py $M 300 2>/dev/null | grep "us/step" >> $OUT
                 [ $M = stratified ] && { echo -n "$M, 1-rank RCCL, mailbox, GPF_SHARD_PLAN_IN_SCAN=0: " >> $OUT; GPF_SHARD_PLAN_IN_SCAN=0 GPF_SHARD_FORCE_COLLECTIVES=1 python3 tools/sharded_loop.py $M 300 2>/dev/null | grep "us/step" >> $OUT; } ;;
               esac
             done
             # the reference's default sort_particles = true across shards: the replicated plan (every rank sorts all n_global weights); next to it the unsharded call
             for N in 1000000 4000000 8000000; do
               echo -n "stratified_sorted, N = $N, no communicator:      " >> $OUT; python3 tools/sharded_loop.py stratified_sorted 100 $N 2>/dev/null | grep "us/step" >> $OUT
               echo -n "stratified_sorted, N = $N, 1-rank RCCL, mailbox: " >> $OUT; GPF_SHARD_FORCE_COLLECTIVES=1 python3 tools/sharded_loop.py stratified_sorted 100 $N 2>/dev/null | grep "us/step" >> $OUT
               echo -n "stratified_sorted, N = $N, unsharded gpf_resample: " >> $OUT; python3 tools/resample_loop.py stratified_sorted 100 $N 2>/dev/null | grep "us/step" >> $OUT
             done
             cat $OUT ;;
    essloop) OUT=gpurun_out/${TAG}_sharded_ess_loop.txt; : > $OUT
             for F in 0.5 1.1 0; do
               for MODE in calls one_call; do
                 echo -n "ESS < $F N, $MODE, no communicator:        " >> $OUT; python3 tools/sharded_ess_loop.py 300 1000000 $MODE $F 2>/dev/null | grep "us/step" >> $OUT
                 echo -n "ESS < $F N, $MODE, 1-rank RCCL + mailbox:  " >> $OUT; GPF_SHARD_FORCE_COLLECTIVES=1 python3 tools/sharded_ess_loop.py 300 1000000 $MODE $F 2>/dev/null | grep "us/step" >> $OUT
               done
             done
             for F in 0.5 1.1; do
               echo -n "ESS < $F N, one_call, 1-rank RCCL + mailbox, GPF_SHARD_REUSE_SUMMARY=0:  " >> $OUT; GPF_SHARD_REUSE_SUMMARY=0 GPF_SHARD_FORCE_COLLECTIVES=1 python3 tools/sharded_ess_loop.py 300 1000000 one_call $F 2>/dev/null | grep "us/step" >> $OUT
               echo -n "ESS < $F N, one_call, 1-rank RCCL + mailbox, GPF_LAZY_MOVE=0:            " >> $OUT; GPF_LAZY_MOVE=0 GPF_SHARD_FORCE_COLLECTIVES=1 python3 tools/sharded_ess_loop.py 300 1000000 one_call $F 2>/dev/null | grep "us/step" >> $OUT
             done
             cat $OUT ;;
    tworanks) hipcc -O2 -std=c++17 -fPIC -shared tests/loopback_rccl/loopback_rccl.cpp -o /tmp/libloopback_rccl.so
             GPF_BENCH_ONE_DEVICE=1 GPF_SHARD_ENGINE=library GPF_RCCL_LIBRARY=/tmp/libloopback_rccl.so python bench.py --gpus 2 --steps 20 --warmup 5 --particles-per-gpu 500000 \
               > gpurun_out/${TAG}_bench_2ranks_one_gpu_loopback.json 2> gpurun_out/${TAG}_bench_2ranks.err; cut -c1-700 gpurun_out/${TAG}_bench_2ranks_one_gpu_loopback.json; tail -2 gpurun_out/${TAG}_bench_2ranks.err ;;
    variant) bash tools/variant_stats.sh $R/gpurun_out/${TAG}_$A1 $A2 ${A3//,/ } 2>&1 | tail -30 | cut -c1-200 ;;
    py)      python3 tools/$A1.py ${A2//,/ } > gpurun_out/${TAG}_$A1.txt 2> gpurun_out/${TAG}_$A1.err; cut -c1-220 gpurun_out/${TAG}_$A1.txt | tail -40; tail -3 gpurun_out/${TAG}_$A1.err ;;
    *)       echo "unknown step $S" ;;
  esac
done
