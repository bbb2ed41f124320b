# histogram: copies per wave (shipped) against copies shared by the block's waves, 16-64 copies, 16 or 32 waves per CU
mkdir -p gpurun_out/r4w
for v in base hsa hsb hsc hsd base; do
  if [ $v = base ]; then unset RANS_AMD_LIB; else export RANS_AMD_LIB=$GRAFT_REPO_ROOT/build/libexp_$v.so; fi
  python tools/hist_rate.py 2>&1 | grep -v amdgpu.ids | sed "s/^/$v /" | tee -a gpurun_out/r4w/hist.log
done
