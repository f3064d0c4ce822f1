for v in 1 0; do INGVIO_GATE_LDL=$v timeout 120 python bench.py --no-cpu --steps 10 --warmup 2 2>/dev/null > /tmp/o.json; python -c "
import json;d=json.load(open('/tmp/o.json'));print('LDL=$v', round(d['ms_per_step'],4), {k:round(x['avg_ms'],4) for k,x in d['kernels'].items() if k in ('k_feat_gate3','k_feat_gram')})"; done
