cd $GRAFT_REPO_ROOT
O=gpurun_out/r5ao; mkdir -p $O
t0=$(date +%s)
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --details $O/full.json > $O/full.out 2> $O/full.err; echo "full rc $? bytes $(tail -n 1 $O/full.out | wc -c) seconds $(( $(date +%s) - t0 ))"
tail -3 $O/full.err | cut -c1-300
