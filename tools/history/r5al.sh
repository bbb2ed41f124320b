# The slow state of r5ak (every pair of a process >= 0.403 ms after test_gpu_scale.py): does holding a spacer of device memory
# (so that the process's buffers come from another part of physical memory) leave it?
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5al; mkdir -p $O
b() { tag=$1; shift
      timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-configs --no-cpu-baseline --details $O/d_$tag.json "$@" > $O/b_$tag.out 2> $O/b_$tag.err
      tail -n 1 $O/b_$tag.out | python -c "
import json,sys
d=json.loads(sys.stdin.read()); p=d.get('placement',{})
print('$tag', d['ms_per_step'], d['roofline']['frac'], 'first', p.get('first_pair_ms'), 'min', p.get('min_ms'), 'max', p.get('max_ms'), 'pairs', p.get('pairs'), 'sclk', d.get('clocks',{}).get('sclk_hz_measured'))
open('$O/last_min','w').write(str(p.get('min_ms')))"; }
ls /sys/kernel/debug/dri/ 2>&1 | head -3
b fresh
for it in 1 2 3 4; do
  timeout 900 python -m pytest tests/test_gpu_scale.py -m gpu -x -q 2>&1 | tail -1
  b after$it
  if python -c "import sys; sys.exit(0 if float(open('$O/last_min').read()) > 0.395 else 1)"; then
    echo "slow state reached at iteration $it"
    b s64 --debug-spacer-gib 64
    b plain1
    b s160 --debug-spacer-gib 160
    b plain2
    b s16 --debug-spacer-gib 16
    b plain3
    break
  fi
done
