#!/bin/bash
# Single-thread "reference as shipped" rates (CPU baseline item 1): the
# reference's four sample programs, unmodified (oracle/_ref/main*, built by `make -C oracle
# mains`), and the same programs compiled against include/ryg_rans_amd/compat/ (built by
# tests/test_compat_headers.py into build/compat/), run on book1 on THIS machine's CPU.
# The mains read ./book1, so they run with the reference directory as cwd (read-only use).
#   bash tools/cpu_reference_rates.sh > profiles/r01_cpu_reference_mains.md
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
REF=${REF:-/root/reference}
make -s -C "$ROOT/oracle" mains >/dev/null
echo "# Reference sample programs on book1 (768771 bytes), single thread"
echo
echo "CPU: $(grep -m1 'model name' /proc/cpuinfo | cut -d: -f2 | sed 's/^ //'), $(nproc) hardware threads visible."
echo "Each program times 5 runs per phase; the table keeps the best run (clocks / symbol, MiB/s as printed)."
echo
echo "| program | headers | phase | bytes | best clocks/sym | best MiB/s |"
echo "|---|---|---|---|---|---|"
for prog in main main64 main_simd main_alias; do
  for flavour in reference compat; do
    if [ $flavour = reference ]; then exe="$ROOT/oracle/_ref/$prog"; else exe="$ROOT/build/compat/$prog"; fi
    [ -x "$exe" ] || continue
    (cd "$REF" && "$exe") | python3 -c '
import re, sys
prog, flavour = sys.argv[1], sys.argv[2]
phase, kind, best, size, order = None, None, {}, {}, []
for line in sys.stdin:
    line = line.strip()
    m = re.match(r"(\d+) clocks, ([0-9.]+) clocks/symbol \(\s*([0-9.]+)M", line)
    if m and phase:
        clocks, rate = float(m.group(1)) / 768771.0, float(m.group(3))
        key = (phase, kind)
        if key not in best:
            order.append(key)
        if key not in best or clocks < best[key][0]:
            best[key] = (clocks, rate)
        continue
    m = re.match(r"(.*rANS) encode:", line)
    if m:
        phase, kind = m.group(1), "encode"
        continue
    m = re.match(r"(.*rANS): (\d+) bytes", line)
    if m:
        phase, kind = m.group(1), "decode"
        size[phase] = m.group(2)
for (phase, kind) in order:
    c, r = best[(phase, kind)]
    print("| %s | %s | %s %s | %s | %.2f | %.1f |" % (prog, flavour, phase, kind, size.get(phase, ""), c, r))
' $prog $flavour
  done
done
