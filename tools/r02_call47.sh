#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
( timeout 280 python -m pytest tests -m gpu -q -x --timeout 120 2>&1 | tail -8 ) > gpurun_out/c47_pytest.log
tail -4 gpurun_out/c47_pytest.log
for r in 1 2; do for v in base nostage; do
  if [ "$v" = base ]; then unset RANS_AMD_LIB; else export RANS_AMD_LIB=$PWD/build/libexp_$v.so; fi
  echo "$v $(timeout 60 python tools/time_lanes.py --fmt word --ways 64 --chunk 32768 --log2n 30 --sb 12 --encode 2>&1 | grep -v amdgpu.ids | sed 's/decode.*| //')"
done; done
