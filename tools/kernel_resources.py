#!/usr/bin/env python3
"""Static resources of every kernel of libingvio_hip.so (VGPRs, AGPRs, SGPRs, scratch, LDS, occupancy) as the compiler reports them
(-Rpass-analysis=kernel-resource-usage), compiled with the flags of ingvio_amd/build.py.  Needs no GPU.
usage: python tools/kernel_resources.py > profiles/rNN_kernel_resources.txt"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from ingvio_amd import build as B

def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    return [re.sub(r"\(.*$", "", o.replace("(anonymous namespace)::", "")) for o in out]

rows = []
for src in B.HIP_SOURCES:
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value"] + B.EXTRA_FLAGS.get(src, []) + \
          ["-Rpass-analysis=kernel-resource-usage", "-c", os.path.join(B.CSRC, src), "-o", "/dev/null"]
    err = subprocess.run(cmd, capture_output=True, text=True).stderr
    cur = None
    for line in err.splitlines():
        m = re.search(r"remark: (?:[^:]+:\d+:\d+: )?\s*(.*?) \[-Rpass-analysis", line)
        if not m:
            continue
        t = m.group(1).strip()
        if t.startswith("Function Name:"):
            cur = {"src": src, "name": t.split(":", 1)[1].strip()}
            rows.append(cur)
        elif cur is not None and ":" in t:
            k, v = t.split(":", 1)
            cur[k.strip()] = v.strip()
names = demangle([r["name"] for r in rows])
print("# static LDS only (dynamic shared memory is set at launch); scratch in bytes per lane; occ = waves per SIMD the register budget allows")
print("%-22s %-58s %5s %5s %5s %8s %8s %4s" % ("file", "kernel", "VGPR", "AGPR", "SGPR", "scratch", "LDS [B]", "occ"))
for r, n in zip(rows, names):
    print("%-22s %-58s %5s %5s %5s %8s %8s %4s" % (r["src"], n[:58], r.get("VGPRs", "?"), r.get("AGPRs", "?"), r.get("TotalSGPRs", "?"),
                                                  r.get("ScratchSize [bytes/lane]", "?"), r.get("LDS Size [bytes/block]", "?"), r.get("Occupancy [waves/SIMD]", "?")))
