#!/bin/bash
# full GPU verification of the committed state: parity tests, smoke, default bench (developer tool; run through gpurun)
timeout 900 python -m pytest tests -x -q -m gpu > gpurun_out/verify_pytest.log 2>&1; grep -E "passed|failed" gpurun_out/verify_pytest.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 400 python bench.py > gpurun_out/verify_bench.json 2> gpurun_out/verify_bench.err; echo bench rc=$?
python - <<'PY'
import json
d=[json.loads(l) for l in open('gpurun_out/verify_bench.json').read().splitlines() if l.startswith('{')][-1]
r=d['roofline']
print('value', round(d['value']), 'us/step', round(d['ms_per_step']*1e3,1), 'frame_frac', round(r['frame_frac'],3), 'dominant', r['kernel'], round(r['frac'],3), 'serial us', round(r['serial_us_per_step'],1))
print({k: (round(v['launch_us'],1), round(v['frac'],3)) for k,v in r['kernels'].items()})
c, s = d['cpu_baseline'], d['cpu_baseline_strong']
print('cpu reference shape', c['value'], 'strong', s and s['value'], 'x', d['speedup_vs_cpu_baseline_strong'])
for k,v in d['extra'].items(): print(' ', k, (str(round(v['us_per_step'],1)) + ' us ' + str(round(v['gtexels_per_s'],1)) + ' Gtexel/s') if 'us_per_step' in v else (str(round(v.get('median_us_per_call', v.get('us_per_call', 0)),1)) + ' us/call'))
PY
