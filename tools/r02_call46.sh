#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
( timeout 280 python -m pytest tests -m gpu -q -x --timeout 120 2>&1 | tail -6 ) > gpurun_out/c46_pytest.log
tail -3 gpurun_out/c46_pytest.log
timeout 120 python - <<'PY' 2>&1 | grep -v amdgpu.ids
import sys; sys.path.insert(0,'.')
import torch, bench, ryg_rans_amd as R
ctx=R.Context(0)
for K,sb,log2n in ((4096,16,29),(256,16,30),(4096,12,29)):
    e,a=bench.measure_config(torch,R,ctx,"alias %d sb%d"%(K,sb),R.FMT_ALIAS,sb,K,64,32768,log2n,1,10,"cuda")
    print(e["name"], "enc", e["encode"]["ms_mean"], e["encode"]["kernels"], "dec", e["decode"]["ms_mean"], e["bit_exact_roundtrip"], flush=True)
PY
