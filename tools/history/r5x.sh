cd $GRAFT_REPO_ROOT
for seed in 101 102 103; do timeout 900 python tools/stress.py --cases 700 --seed $seed 2>&1 | tail -2; done
timeout 900 python tools/stress.py --cases 150 --seed 104 --big 2>&1 | tail -2
