"""Register / scratch / LDS / occupancy table of every kernel of libgpf (compile-time, no GPU):
    python3 tools/kernel_resources.py > profiles/rNN_kernel_resources.txt
Compiles the four translation units with -Rpass-analysis=kernel-resource-usage and prints one line per kernel."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
csrc = os.path.join(ROOT, "genparticlefilters.jl_amd", "csrc")
err = ""
with tempfile.TemporaryDirectory() as td:
    procs = [subprocess.Popen(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-fPIC", "-fvisibility=hidden",
                               "-Wno-unused-value", "-I" + os.path.join(ROOT, "include"), "-Rpass-analysis=kernel-resource-usage", *sys.argv[1:],
                               "-c", os.path.join(csrc, u), "-o", os.path.join(td, u + ".o")], stderr=subprocess.PIPE, text=True)
             for u in ("libgpf_core.hip", "libgpf_resample.hip", "libgpf_aux.hip", "libgpf_shard.hip")]
    for q in procs:
        err += q.communicate()[1]
class _P: stderr = err
p = _P()
rows, cur = [], None
for ln in p.stderr.splitlines():
    m = re.search(r"remark: (?:\s*)([A-Za-z ]+?)(?: \[bytes/lane\]| \[waves/SIMD\]| \[bytes/block\])?: (\S+) \[-Rpass", ln)
    if not m:
        continue
    k, v = m.group(1).strip(), m.group(2)
    if k == "Function Name":
        cur = {"name": v}; rows.append(cur)
    elif cur is not None:
        cur[k] = v
names = subprocess.run(["c++filt"] + [r["name"] for r in rows], capture_output=True, text=True).stdout.splitlines()
print(f"{'kernel':70s} {'VGPR':>5s} {'AGPR':>5s} {'SGPR':>5s} {'scratch':>8s} {'LDS':>7s} {'waves/SIMD':>10s}")
for r, n in zip(rows, names):
    n = re.sub(r"^void ", "", n); n = n.split("(")[0].replace("gpf::", "")
    print(f"{n[:70]:70s} {r.get('VGPRs', '?'):>5s} {r.get('AGPRs', '?'):>5s} {r.get('TotalSGPRs', '?'):>5s} {r.get('ScratchSize', '?'):>8s} "
          f"{r.get('LDS Size', '?'):>7s} {r.get('Occupancy', '?'):>10s}")
