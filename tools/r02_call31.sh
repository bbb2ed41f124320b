#!/bin/bash
# fused placement in the staged lane encoder: parity tests + config-2 style timings, fused vs three-kernel path
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
OUT=gpurun_out/c31; mkdir -p $OUT
( timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -15 ) > $OUT/pytest.log
tail -3 $OUT/pytest.log
( timeout 120 python __graft_entry__.py smoke 2>&1 | tail -2 ) > $OUT/smoke.log; cat $OUT/smoke.log
for v in "" "RANS_AMD_ENCODE_UNFUSED=1"; do
  echo "== [$v]"
  for a in "--fmt r64 --ways 2" "--fmt word --ways 2 --sb 12" "--fmt byte --ways 2" "--fmt r64 --ways 8 --chunk 1024" "--fmt r64 --ways 2 --chunk 4096"; do
    env $v timeout 200 python tools/time_lanes.py $a --encode 2>&1 | grep -v amdgpu.ids
  done
done > $OUT/lanes.log 2>&1
cat $OUT/lanes.log
