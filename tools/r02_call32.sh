#!/bin/bash
# k_encode_lanes_r64x2 + copier-wave placement: parity tests, then timings against the staged kernel / the unfused path
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
OUT=gpurun_out/c32; mkdir -p $OUT
( timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -25 ) > $OUT/pytest.log
tail -12 $OUT/pytest.log
for v in "" "RANS_AMD_ENCODE_UNFUSED=1" "RANS_AMD_NO_R64X2_ENC=1" "RANS_AMD_NO_R64X2_ENC=1 RANS_AMD_ENCODE_UNFUSED=1"; do
  echo "== [$v]"
  for a in "--fmt r64 --ways 2" "--fmt r64 --ways 2 --chunk 4096" "--fmt word --ways 2 --sb 12" "--fmt byte --ways 2"; do
    env $v timeout 200 python tools/time_lanes.py $a --encode 2>&1 | grep -v amdgpu.ids
  done
done > $OUT/lanes.log 2>&1
cat $OUT/lanes.log
timeout 300 python tools/time_lanes.py --fmt word --ways 64 --chunk 32768 --log2n 30 --sb 12 --encode 2>&1 | grep -v amdgpu.ids
timeout 300 python tools/time_lanes.py --fmt byte --ways 64 --chunk 32768 --log2n 30 --sb 14 --encode 2>&1 | grep -v amdgpu.ids
