#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
for v in "" "RANS_AMD_ENC_DEBUG=1" "RANS_AMD_ENC_DEBUG=3" "RANS_AMD_ENCODE_UNFUSED=1"; do
  echo "== [$v]"
  env $v timeout 40 python tools/time_lanes.py --fmt r64 --ways 2 --encode --no-check 2>&1 | grep -v amdgpu.ids
done
