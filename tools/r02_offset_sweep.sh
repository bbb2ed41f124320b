for oo in 0 4096 65536 1048576 33554432 50331648; do
 for co in 0 65536 16777216; do
  echo "out+$oo cont+$co: $(python bench.py --no-cpu-baseline --no-configs --steps 20 --warmup 5 --prewarm-ms 60 --debug-out-offset $oo --debug-cont-offset $co 2>/dev/null | python -c 'import json,sys; d=json.load(sys.stdin); r=d["roofline"]; print(r["kernel_ms_avg"], r["frac"])')"
 done
done
