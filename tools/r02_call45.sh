#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
( timeout 280 python -m pytest tests -m gpu -q -x --timeout 120 2>&1 | tail -6 ) > gpurun_out/c45_pytest.log
tail -3 gpurun_out/c45_pytest.log
for v in "" "RANS_AMD_ENCODE_UNFUSED=1"; do
echo "== [$v]"
for a in "--fmt byte --ways 64 --sb 14" "--fmt byte --ways 256 --sb 14" "--fmt word --ways 128 --sb 12" "--fmt word --ways 256 --sb 12" "--fmt word --ways 512 --sb 12" "--fmt r64 --ways 256 --sb 14"; do
  env $v timeout 60 python tools/time_lanes.py $a --chunk 32768 --log2n 30 --encode 2>&1 | grep -v amdgpu.ids | sed 's/decode.*| //'
done; done
