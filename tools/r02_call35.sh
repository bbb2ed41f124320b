#!/bin/bash
# full GPU tests (per-test timeout) + lane encoder timings, every command under its own short timeout
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
OUT=gpurun_out/c35; mkdir -p $OUT
( timeout 240 python -m pytest tests -m gpu -q -x --timeout 90 2>&1 | tail -25 ) > $OUT/pytest.log
tail -8 $OUT/pytest.log
for v in "" "RANS_AMD_LANES_FUSED=1"; do
  echo "== [$v]"
  for a in "--fmt r64 --ways 2" "--fmt word --ways 2 --sb 12" "--fmt byte --ways 2"; do
    env $v timeout 40 python tools/time_lanes.py $a --encode 2>&1 | grep -v amdgpu.ids
  done
done > $OUT/lanes.log 2>&1
cat $OUT/lanes.log
timeout 60 python tools/time_lanes.py --fmt word --ways 64 --chunk 32768 --log2n 30 --sb 12 --encode 2>&1 | grep -v amdgpu.ids
