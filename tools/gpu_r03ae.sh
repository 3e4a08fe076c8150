R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; mkdir -p $R/gpurun_out; cd /tmp; export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_c4; rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_c4 -- python3 $R/tools/config_loop.py config4 250 > $R/gpurun_out/prof_c4.log 2>&1
f=$(find $R/gpurun_out/prof_c4 -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows=[r for r in csv.DictReader(open(sys.argv[1])) if 'k_scan_residual2' in r['Kernel_Name']]
d=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3 for r in rows]
print(len(d), 'launches; us:', [round(x,1) for x in d])
PY
