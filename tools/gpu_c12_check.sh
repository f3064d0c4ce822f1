python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "twelve or window_size" 2>&1 | tail -3
python bench.py --clones 12 --no-aux --no-latency --no-cpu 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('c12',d['value'],d['ms_per_step'])"
python -c "
import json;d=json.load(open('bench_detail.json'));print({k:(round(v.get('avg_ms'),4),v.get('kernel')) for k,v in d['kernels'].items()})"
ingvio_amd/lib/ingvio_replay --synth "feats=150,clones=11,life=13,cohort=0,frames=90,key=0" --time --set "frame_select_interval: 5" | grep LATENCY | cut -c1-60
