"""Static resource usage of every kernel (VGPRs, AGPRs, SGPRs, scratch, LDS, occupancy) from
`hipcc -Rpass-analysis=kernel-resource-usage` with the flags of the product build — no GPU needed.
  python tools/kernel_resources.py > profiles/r01_kernel_resources.txt"""
import os
import re
import subprocess
import sys
import importlib.util

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("mot_build", os.path.join(ROOT, "3d-lidar-multi-object-tracking_amd", "build.py"))
B = importlib.util.module_from_spec(spec); spec.loader.exec_module(B)

rows = []
for src in B.SOURCES:
    if src == "mot_api.hip":
        continue
    flags = [f for f in B.HIPCC_FLAGS if f not in ("-shared", "-fPIC")]
    r = subprocess.run([B.hipcc()] + flags + ["-c", os.path.join(B.CSRC, src), "-o", os.devnull, "-Rpass-analysis=kernel-resource-usage"],
                       capture_output=True, text=True)
    cur = None
    for line in r.stderr.splitlines():
        m = re.search(r"remark: +([\w \[\]/]+?): +(\S+) \[-Rpass", line)
        if not m:
            continue
        k, v = m.group(1).strip(), m.group(2)
        if k == "Function Name" or k == "Name":
            cur = dict(file=src, name=subprocess.run(["c++filt", v], capture_output=True, text=True).stdout.split("(")[0].strip()); rows.append(cur)
        elif cur is not None:
            cur[k] = v
cols = ["VGPRs", "AGPRs", "TotalSGPRs", "ScratchSize [bytes/lane]", "Occupancy [waves/SIMD]", "LDS Size [bytes/block]"]
print("%-12s %-28s %6s %6s %6s %8s %10s %10s" % ("file", "kernel", "VGPR", "AGPR", "SGPR", "scratch", "waves/SIMD", "LDS B/blk"))
for r in rows:
    print("%-12s %-28s %6s %6s %6s %8s %10s %10s" % tuple([r["file"], r["name"]] + [r.get(c, "?") for c in cols]))
