# Does what ran before the bench process change the headline?  (r5final: first bench after the suite was 0.417 ms.)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5ak; mkdir -p $O
b() { timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-configs --no-cpu-baseline --details $O/d_$1.json > $O/b_$1.out 2> $O/b_$1.err
      tail -n 1 $O/b_$1.out | python -c "
import json,sys
d=json.loads(sys.stdin.read()); p=d.get('placement',{})
print('$1', d['ms_per_step'], d['roofline']['frac'], 'first', p.get('first_pair_ms'), 'min', p.get('min_ms'), 'max', p.get('max_ms'), 'pairs', p.get('pairs'), 'copy', d['roofline'].get('frac_of_measured_copy'), 'sclk', d.get('clocks',{}).get('sclk_hz_measured'))"
      rocm-smi --showpower --showtemp --showclocks 2>/dev/null | grep -E "Power|Temperature \(Sensor (junction|memory)|sclk|mclk|fclk" | tr -s ' ' | cut -c1-100 | tr '\n' ';'; echo; }
b A1; b A2
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
b B
timeout 900 python -m pytest tests/test_gpu_scale.py -m gpu -x -q 2>&1 | tail -1
b C1; b C2
timeout 3000 python -m pytest tests -m gpu -x -q 2>&1 | tail -1
b D1; b D2; b D3
