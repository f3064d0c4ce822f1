timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "qr_compress" 2>&1 | tail -5
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/qrchol -- python /root/repo/tests/gpu_qr_bench.py 2>&1 | tail -2; cd /root/repo; f=$(ls -t $(find gpurun_out/qrchol -name "*kernel_stats.csv") | head -1); python - "$f" <<'PY'
import csv,sys
tot=0
for r in list(csv.DictReader(open(sys.argv[1])))[:12]:
    print(r["Name"][:60].ljust(60), r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e3)
PY
