#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
( timeout 200 python -m pytest tests -m gpu -q -x --timeout 100 -k "lane or two_way or stress" 2>&1 | tail -6 ) > gpurun_out/c49_pytest.log
tail -3 gpurun_out/c49_pytest.log
for v in "RANS_AMD_LANES_FUSED=1" "X=1"; do
  echo "== [$v]"
  for a in "--fmt r64 --ways 2" "--fmt word --ways 2 --sb 12" "--fmt byte --ways 2" "--fmt r64 --ways 2 --chunk 1024"; do
    env $v timeout 40 python tools/time_lanes.py $a --encode 2>&1 | grep -v amdgpu.ids | sed 's/decode.*| //'
  done
done
