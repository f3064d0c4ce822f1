import sys, json
sys.path.insert(0, "/root/repo")
import bench
for n, f in ((8, 64), (4, 128), (2, 256), (1, 512)):
    r = bench.stagers_rate_processes(n, f, reps=20)
    print("processes", n, "x", f, "aggregate %.0f" % r["aggregate_updates_per_s"], [round(x) for x in r["per_stager_updates_per_s"]], flush=True)
r = bench.stagers_rate(8, 64, reps=10)
print("threads 8 x 64", json.dumps(r))
