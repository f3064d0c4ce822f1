"""Averages rocprofv3 --pmc counter_collection CSVs per kernel and launch.
usage: python tests/pmc_summary.py out.csv dir1 [dir2 ...]   (each dir: one `rocprofv3 --pmc ... -d dir --output-format csv` pass;
counters are collected in their own passes, never together with trace domains)."""
import collections, csv, glob, re, sys

out, dirs = sys.argv[1], sys.argv[2:]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for d in dirs:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = re.sub(r"^void ", "", r["Kernel_Name"])
            k = re.sub(r"\(anonymous namespace\)::", "", k)
            k = re.split(r"[<(]", k)[0].strip()
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
names = sorted({c for k in acc for c in acc[k]})
with open(out, "w") as fo:
    fo.write("kernel," + ",".join(names) + "\n")
    for k in acc:
        fo.write(k + "," + ",".join(("%.4g" % (acc[k][c] / cnt[k][c])) if cnt[k][c] else "" for c in names) + "\n")
print("wrote", out, "kernels:", len(acc))
