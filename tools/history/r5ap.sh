# does freeing ~170 GiB of spacers right before the timed region cost anything?  per-launch times, spaced against in-a-row
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5ap; mkdir -p $O
b() { tag=$1; shift
      timeout 300 python bench.py --gpus 1 --steps 40 --warmup 5 --no-configs --no-cpu-baseline --dump-launch-ms --details $O/d_$tag.json "$@" > $O/b_$tag.out 2> $O/b_$tag.err
      python - <<PY
import json
d=json.load(open("$O/d_$tag.json"))
def find(o):
    if isinstance(o,dict):
        for k,v in o.items():
            if 'launch_ms' in k and isinstance(v,list): return v
            r=find(v)
            if r: return r
ms=find(d)
print("$tag", d['roofline']['kernel_ms_avg'], 'chosen', d['placement'].get('probe_ms_chosen'), 'first10', round(sum(ms[:10])/10,4), 'last10', round(sum(ms[-10:])/10,4), 'min', min(ms), 'max', max(ms))
PY
}
for i in 1 2 3; do b spaced$i; b row$i --placement-stride-gib 0; done
timeout 600 python -m pytest tests/test_dist_gloo.py -m gpu -x -q 2>&1 | tail -1
