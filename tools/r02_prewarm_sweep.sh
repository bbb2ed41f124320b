for pw in 0 30 60 120 250 600; do
  echo "prewarm $pw: $(python bench.py --no-cpu-baseline --no-configs --steps 20 --warmup 5 --prewarm-ms $pw 2>/dev/null | python -c 'import json,sys; d=json.load(sys.stdin); r=d["roofline"]; print(r["kernel_ms_avg"], r["frac"], d.get("clocks",{}).get("sclk_hz_measured"))')"
done
