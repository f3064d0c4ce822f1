timeout 500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "large_window" 2>&1 | tail -3
python bench.py --config 5 --no-cpu --steps 10 --warmup 3 2>&1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], {k:v['avg_ms'] for k,v in d['kernels'].items()})"
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/c5chol -- python /root/repo/bench.py --config 5 --no-cpu --no-profile --steps 10 --warmup 3 > /dev/null 2>&1; cd /root/repo; f=$(ls -t $(find gpurun_out/c5chol -name "*kernel_stats.csv") | head -1); python - "$f" <<'PY'
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:16]:
    print(r["Name"][:60].ljust(60), r["Calls"], float(r["AverageNs"])/1e3)
PY
for B in 1 8 128; do python bench.py --config 5 --batch $B --no-cpu --steps 10 --warmup 3 2>&1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('B=$B', d['value'], d['ms_per_step'], {k:round(v['avg_ms'],3) for k,v in d['kernels'].items()})"; done
