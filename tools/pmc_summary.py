"""Folds rocprofv3 --pmc counter_collection CSVs into profiles/counters.json (per kernel, per launch), the file bench.py reads
its executed-operation counts and HBM traffic from.

usage: python tools/pmc_summary.py --key c2_B512_F150_C11_N249 --last 3 --source "..." --out profiles/counters.json dir1 [dir2 ...]
       (each dir: one `rocprofv3 --pmc ... -d dir --output-format csv` pass; counters are collected in their own passes, never
       together with trace domains).  --last N keeps only the last N dispatches of every kernel (the frame steps; earlier
       dispatches of the same kernel belong to the set-up of the priors).  --csv also writes the table as CSV.
A counter that appears in several passes is averaged over them."""
import argparse
import collections
import csv
import glob
import json
import os
import re

ap = argparse.ArgumentParser()
ap.add_argument("--key", required=True)
ap.add_argument("--last", type=int, default=0)
ap.add_argument("--source", default="")
ap.add_argument("--out", required=True)
ap.add_argument("--csv", default=None)
ap.add_argument("--per-step", default="", help="kernel=launches per bench step, comma separated: for these the average runs over the "
                "last (--last x launches) dispatches and the count is stored as _launches_per_step (stages made of several launches)")
ap.add_argument("--no-build-id", action="store_true", help="do not record the loaded library's build id (bench.py then treats the "
                "workload's counters as of unknown provenance = stale)")
ap.add_argument("dirs", nargs="+")
a = ap.parse_args()
per_step = {kv.split("=")[0]: int(kv.split("=")[1]) for kv in a.per_step.split(",") if "=" in kv}


def norm(k):
    k = re.sub(r"^void ", "", k)
    k = re.sub(r"\(anonymous namespace\)::", "", k)
    return re.split(r"[<(]", k)[0].strip()


table = collections.defaultdict(lambda: collections.defaultdict(list))       # kernel -> counter -> per-pass averages
launches = {}
for d in a.dirs:
    rows = []
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        rows += list(csv.DictReader(open(f)))
    # (kernel, counter) -> {dispatch id: value summed over the dimension rows of that dispatch}
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in rows:
        per[(norm(r["Kernel_Name"]), r["Counter_Name"])][int(r["Dispatch_Id"])] += float(r["Counter_Value"])
    for (k, c), byd in per.items():
        ids = sorted(byd)
        if a.last > 0:
            ids = ids[-a.last * per_step.get(k, 1):]
        table[k][c].append(sum(byd[i] for i in ids) / len(ids))
        launches[k] = len(ids)
kernels = {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in table.items()}
for k in kernels:
    kernels[k]["_launches_averaged"] = launches[k]
    kernels[k]["_launches_per_step"] = per_step.get(k, 1)
J = {"workloads": {}}
if os.path.exists(a.out):
    try:
        J = json.load(open(a.out))
    except ValueError:
        pass
entry = {"source": a.source, "kernels": kernels}
if not a.no_build_id:
    # the counters belong to the library that produced them: record the per-translation-unit hashes of the library that is
    # loaded HERE (this script runs on the GPU box right after the PMC passes, against the same snapshot)
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from ingvio_amd import capi
    entry["build"] = capi.build_id()["tu"]
J.setdefault("workloads", {})[a.key] = entry
J["note"] = ("per-launch averages of rocprofv3 --pmc counters (tools/pmc_summary.py); FETCH_SIZE / WRITE_SIZE in KiB as rocprofv3 "
             "reports them (bench.py applies the gfx950 x2 on FETCH_SIZE); SQ_INSTS_* are wave-level instruction counts")
with open(a.out, "w") as f:
    json.dump(J, f, indent=1, sort_keys=True)
if a.csv:
    names = sorted({c for k in kernels for c in kernels[k]})
    with open(a.csv, "w") as fo:
        fo.write("kernel," + ",".join(names) + "\n")
        for k in sorted(kernels):
            fo.write(k + "," + ",".join(("%.6g" % kernels[k][c]) if c in kernels[k] else "" for c in names) + "\n")
print("wrote", a.out, "workload", a.key, "kernels:", len(kernels))
