# spaced placement candidates (24 GiB apart) against candidates in a row, fresh processes; then one full default run
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5an; mkdir -p $O
b() { tag=$1; shift
      timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-configs --no-cpu-baseline --details $O/d_$tag.json "$@" > $O/b_$tag.out 2> $O/b_$tag.err
      tail -n 1 $O/b_$tag.out | python -c "
import json,sys
d=json.loads(sys.stdin.read()); p=d.get('placement',{})
print('$tag', d['ms_per_step'], d['roofline']['frac'], 'first', p.get('first_pair_ms'), 'min', p.get('min_ms'), 'max', p.get('max_ms'), 'pairs', p.get('pairs'), 'stride', p.get('stride_gib'))"; }
for i in 1 2 3 4; do b spaced$i; b row$i --placement-stride-gib 0; done
/usr/bin/time -v timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --details $O/full.json > $O/full.out 2> $O/full.err; echo "full rc $? bytes $(tail -n 1 $O/full.out | wc -c)"
grep -E "Elapsed|Maximum resident" $O/full.err
