#!/bin/bash
# round 3, call 15: the committed tree -- whole GPU suite, smoke, the default bench line
mkdir -p gpurun_out
(time timeout -k 5 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -6) > gpurun_out/r03_15_tests.log 2>&1
(timeout -k 5 300 python -c "import __graft_entry__ as g; g.smoke()") > gpurun_out/r03_15_smoke.log 2>&1
(time timeout -k 5 600 python bench.py --gpus 1 --steps 20 --warmup 5) > gpurun_out/r03_15_bench.json 2> gpurun_out/r03_15_bench.err
tail -8 gpurun_out/r03_15_tests.log; tail -2 gpurun_out/r03_15_smoke.log; python - <<'PY'
import json
d=json.load(open('gpurun_out/r03_15_bench.json'))
r=d['roofline']
print('value', d['value'], 'ms/step', d['ms_per_step'], 'frac', r['frac'], 'kernel', r['kernel'], r['kernel_ms_avg'], 'traffic', r.get('traffic'), 'exact', d['bit_exact_roundtrip'], 'headline', d['headline'], 'knobs', d['knobs'])
for c in d['configs']:
    print(' ', c['name'][:40], c['decode']['kernel'], c['decode']['ms_mean'], c['decode']['frac'], '|', c['encode']['kernels'][:50], c['encode']['ms_mean'], c['encode']['frac'], c['bit_exact_roundtrip'], c.get('oracle_chunks_checked'), c.get('oracle_chunks_total'))
cb=d['cpu_baseline']; print('cpu', cb['value'], cb['cores'], cb['kind'], cb.get('cpu_quota_cores'), cb.get('reference_value'), [ (x['decoder'][:12], x['thread_sweep_GBps']) for x in cb['decoders']])
print('oracle', d['oracle_chunks_checked'], d['oracle_chunks_total'], d['decodes_oracle_container'], d['oracle_check_s'])
PY
tail -3 gpurun_out/r03_15_bench.err
