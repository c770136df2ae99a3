#!/bin/bash
# full GPU verification of the committed state: parity tests, smoke, default bench (developer tool)
python -m pytest tests -x -q -m gpu 2>&1 | tail -3
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
python bench.py > gpurun_out/verify_bench.json 2> gpurun_out/verify_bench.err; echo bench rc=$?
python - <<'PY'
import json
d=json.loads(open('gpurun_out/verify_bench.json').read().strip().splitlines()[-1])
r=d['roofline']
print('value', round(d['value']), 'ms/step', round(d['ms_per_step']*1e3,1), 'us; frame_frac', round(r['frame_frac'],3), 'roofline', r['kernel'], round(r['frac'],3), 'serial', round(r['serial_us_per_step'],1))
print('kernel_us', {k: round(v,1) for k,v in r['kernel_us'].items()}, 'cpu', round(d['cpu_baseline']['value'],2), 'x', round(d['speedup_vs_cpu_baseline']))
for k,v in d['extra'].items(): print(' ', k, (str(round(v['us_per_step'],1)) + ' us ' + str(round(v['gtexels_per_s'],1)) + ' Gtexel/s') if 'us_per_step' in v else (str(round(v.get('median_us_per_call', v.get('us_per_call', 0)),1)) + ' us/call'))
PY
