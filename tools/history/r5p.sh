cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5p
for add in 0 64 128 256 384 512 1024; do
  echo "== tight slot + $add"
  timeout 300 python tools/time_slots.py --configs c4,word --rounds 1 --tight-slot-add $add 2>&1 | grep "enc tight\|enc slots\|sized-slot"
done | tee gpurun_out/r5p/stride.log
