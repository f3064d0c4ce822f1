#!/bin/bash
# Dumps `ingvio_replay --trace` of every golden stream into gpurun_out/stream_trace/ (so a mismatch can be studied off the box), then
# runs the stream tests.  usage (GPU box): bash tools/gpu_stream_trace.sh
mkdir -p gpurun_out/stream_trace
python - <<'PY'
import subprocess, sys, os
sys.path.insert(0, '.')
from oracle import gen_stream_golden as g
tool = 'ingvio_amd/lib/ingvio_replay'
for name, (spec, ov) in g.STREAMS.items():
    sets = []
    for line in ov.splitlines():
        if line.strip():
            sets += ['--set', line]
    r = subprocess.run([tool, '--synth', spec, '--trace'] + sets, capture_output=True, text=True, timeout=900)
    open('gpurun_out/stream_trace/%s.txt' % name, 'w').write(r.stdout)
    open('gpurun_out/stream_trace/%s.err' % name, 'w').write(r.stderr[-20000:])
    print(name, 'rc', r.returncode, 'bytes', len(r.stdout))
PY
python -m pytest tests/test_stream_golden.py -q -m gpu 2>&1 | tail -40 | tee gpurun_out/stream_trace/pytest.log
