# one-rank sharded code path: bench line + kernel stats
R=$GRAFT_REPO_ROOT; cd $R
export GPF_BENCH_FORCE_SHARDED=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29741 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0
python bench.py --steps 300 --warmup 20 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print({k: d[k] for k in ('value','ms_per_step','shard_engine')}, d['roofline']['all_kernels_us'], d['stratified_variant']['ms_per_step'], d['local_resample_variant']['ms_per_step'])"
cd /tmp; export TMPDIR=/tmp; rm -rf $R/gpurun_out/prof_sh
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_sh -- python3 $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline > /dev/null 2>&1
python3 - <<PY
import csv,glob
f=glob.glob('$R/gpurun_out/prof_sh/*/*kernel_stats.csv')[0]
for r in list(csv.DictReader(open(f)))[:14]:
    print(r['Name'][:60].ljust(62), r['Calls'].rjust(5), f"{float(r['AverageNs'])/1e3:9.2f} us", r['Percentage'])
PY
rm -rf $R/gpurun_out/prof_sh
