# headline decode: chunk size sweep (bench.py --chunk), fresh process each, two passes
mkdir -p gpurun_out/r4t
for i in 1 2 3; do
  for c in 16384 24576 32768 49152 65536; do
    python bench.py --chunk $c --no-configs --no-cpu-baseline > gpurun_out/r4t/c$c.$i.json 2> gpurun_out/r4t/c$c.$i.err
    python - <<PY
import json
d=json.load(open("gpurun_out/r4t/c$c.$i.json"))
print("chunk $c pass $i: value %.1f GB/s kernel %.4f ms frac %.4f c=%.5f B/sym exact %s" % (d["value"], d["roofline"]["kernel_ms_avg"], d["roofline"]["frac"], d["config"]["compressed_bytes_per_symbol"], d["bit_exact_roundtrip"]))
PY
  done
done
