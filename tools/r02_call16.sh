#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
OUT=gpurun_out/c16; mkdir -p $OUT
( timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -25 ) > $OUT/pytest.log
bash tools/ab_configs.sh $OUT/configs.log 1 base RANS_AMD_ALIAS_L2=1:base
python - > $OUT/alias256.log 2>&1 <<'PY'
import sys; sys.path.insert(0,'.')
import torch, bench, ryg_rans_amd as R
ctx=R.Context(0)
for K,sb,log2n in ((256,16,30),(256,12,30),(4096,16,29),(4096,12,29)):
    e,a=bench.measure_config(torch,R,ctx,"alias %d sb%d"%(K,sb),R.FMT_ALIAS,sb,K,64,32768,log2n,1,10,"cuda")
    print(e["name"], "enc", e["encode"]["ms_mean"], "dec", e["decode"]["ms_mean"], e["bit_exact_roundtrip"])
PY
ls -la $OUT
