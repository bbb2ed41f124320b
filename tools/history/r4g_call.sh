# k_decode_word64: 16-wave blocks at 64 VGPRs (9 VGPR + 30 SGPR spills) against 14-wave blocks at 72 VGPRs (2 + 8), interleaved
mkdir -p gpurun_out/r4g
for i in 1 2 3; do
  python bench.py --measure --no-configs --no-cpu-baseline > gpurun_out/r4g/base_$i.json 2>/dev/null
  RANS_AMD_LIB=$PWD/build/libexp_w64lb7.so python bench.py --measure --no-configs --no-cpu-baseline > gpurun_out/r4g/lb7_$i.json 2>/dev/null
done
python - <<PY
import json,glob
for f in sorted(glob.glob("gpurun_out/r4g/*.json")):
    d=json.load(open(f)); print(f.split("/")[-1], d["roofline"]["kernel_ms_avg"], d["roofline"]["frac"], d["placement"].get("probe_ms_min"), d["bit_exact_roundtrip"], d["clocks"].get("per_simd_clocks_per_round_of_64"))
PY
