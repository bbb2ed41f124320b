cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5s
for i in 1 2 3; do
  timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --details gpurun_out/r5s/details$i.json > gpurun_out/r5s/bench$i.out 2> gpurun_out/r5s/bench$i.err; echo "bench $i rc $?"
  tail -c 3500 gpurun_out/r5s/bench$i.out | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms_avg'], d['roofline']['frac'], d['roofline']['traffic'], d.get('value_first_pair'), [ (r['name'], r['decode_ms'], r['enc_tight_ms'], r['oracle_ok']) for r in d['configs']])"
done
