"""Copies the numbers of the committed round-6 record (profiles/r06_bench.json, profiles/r06_rocprof_summary.md) into the places that quote
them: the results paragraph and table of README.md, the headline row and the evidence paragraph of DESIGN.md, the first two rows of
profiles/README.md.  Run after tools/summarize_r06.py.

    python tools/sync_docs_r06.py
"""
import json, re, sys
import os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))) + '/'
d=json.load(open(R+'profiles/r06_bench.json'))
c={r['name']:r for r in d['configs']}
r=d['roofline']
summ=open(R+'profiles/r06_rocprof_summary.md').read()
m=re.search(r"timed ones: \*\*avg ([0-9.]+) us\*\* \(min ([0-9.]+), max ([0-9.]+)\) against the line's kernel_ms_avg ([0-9.]+) us .*?whole row: avg ([0-9.]+) us", summ)
tavg,tmin,tmax,levt,rowavg=[float(x) for x in m.groups()]
tfrac=float(re.search(r"8 TB/s = \*\*([0-9.]+)\*\*", summ).group(1))
lstep=float(re.search(r"line under the profiler: ms_per_step ([0-9.]+)", summ).group(1))
ndisp=int(re.search(r"`k_decode_word64`: (\d+) dispatches", summ).group(1))
def f(x,n=3): return ("%."+str(n)+"f")%x
fp_ms=d['placement']['first_pair_ms_per_step']; fp_fr=d['frac_first_pair']
# README
p=R+'README.md'; s=open(p).read()
a=s.index("symbols resident in HBM: **"); b=s.index("The reference's own fastest decoder (SSE4.1,")
s=s[:a]+'''symbols resident in HBM: **%.2f TB/s decoded, %.4f ms per step, %.1f %% of the 8 TB/s HBM roofline** (0.3878-0.3946 ms on the round's eight boxes) on the wall clock of
the timed steps (algorithmic bytes: stream read + symbols written; measured traffic 1.048 × that), bit-exact; %.1f %% by the
HIP events around the same launches, %.1f %% in the rocprofv3 trace of another process cut to its timed launches.  With the first pair of buffers two plain
allocations return (no placement probe), the same loop: %.4f ms = %.1f %% (0.3827-0.4114 ms over the round's profile runs: the un-probed pair
is a lottery, which is what the probe is for).  ''' % (d['value']/1000, d['ms_per_step'], 100*r['frac'], 100*r['frac_kernel_events'], 100*tfrac, fp_ms, 100*fp_fr)+s[b:]
def row(prefix, new):
    global s
    a=s.index(prefix); b=s.index("\n",a); s=s[:a]+new+s[b:]
row("| word 64-way, 1 GiB, 16 Ki chunks |", "| word 64-way, 1 GiB, 16 Ki chunks | %s (%s) | %s | **%s** (0.82) | %.1f GB/s |" % (f(c['C3-word64']['decode_ms']), f(c['C3-word64']['decode_frac'],2), f(c['C3-word64']['encode_ms']), f(c['C3-word64']['enc_tight_ms']), c['C3-word64']['cpu_GBps']))
row("| config 2: rans64 2-way, 256 MiB, 512-symbol chunks |", "| config 2: rans64 2-way, 256 MiB, 512-symbol chunks | %s (%s) | %s | **%s** (1.00) | %.1f |" % (f(c['C2-r64x2']['decode_ms']), f(c['C2-r64x2']['decode_frac'],2), f(c['C2-r64x2']['encode_ms']), f(c['C2-r64x2']['enc_tight_ms']), c['C2-r64x2']['cpu_GBps']))
row("| config 4: alias, 4096 symbols, 512 Mi u16 |", "| config 4: alias, 4096 symbols, 512 Mi u16 | %s (%s) | %s | **%s** (0.58) | %.1f |" % (f(c['C4-alias4096']['decode_ms']), f(c['C4-alias4096']['decode_frac'],2), f(c['C4-alias4096']['encode_ms']), f(c['C4-alias4096']['enc_tight_ms']), c['C4-alias4096']['cpu_GBps']))
row("| byte 64-way, 14 bits (the reference's) / 12 bits |", "| byte 64-way, 14 bits (the reference's) / 12 bits | %s (%s) / %s (%s) | %s / %s | **%s / %s** (0.82) | %.1f / %.1f |" % (f(c['byte14']['decode_ms']), f(c['byte14']['decode_frac'],2), f(c['byte12']['decode_ms']), f(c['byte12']['decode_frac'],2), f(c['byte14']['encode_ms']), f(c['byte12']['encode_ms']), f(c['byte14']['enc_tight_ms']), f(c['byte12']['enc_tight_ms']), c['byte14']['cpu_GBps'], c['byte12']['cpu_GBps']))
row("| **word 8-way** (the reference's SIMD layout), 1 Ki chunks", "| **word 8-way** (the reference's SIMD layout), 1 Ki chunks: 8 chunks per wave, decoder and encoder | **%s (%s)** (0.902 one chunk per lane) | **%s** (1.795) | **%s** (1.211; 0.94) | %.1f |" % (f(c['word8']['decode_ms']), f(c['word8']['decode_frac'],2), f(c['word8']['encode_ms']), f(c['word8']['enc_tight_ms']), c['word8']['cpu_GBps']))
row("| **byte 2-way** (`main.cpp`'s layout), 1 Ki chunks", "| **byte 2-way** (`main.cpp`'s layout), 1 Ki chunks: 32 chunks per wave in the decoder | **%s (%s)** (1.006) | %s | %s (0.88) | %.1f |" % (f(c['byte2']['decode_ms']), f(c['byte2']['decode_frac'],2), f(c['byte2']['encode_ms']), f(c['byte2']['enc_tight_ms']), c['byte2']['cpu_GBps']))
row("| **per-chunk models, word / byte**", "| **per-chunk models, word / byte** (one kernel: count + normalise + records + code; traffic 1.06 × algorithmic) | %s (%s) / %s (%s) | — | **%s / %s** (0.81 / 0.80 + 0.03 of rows) | — |" % (f(c['word-adaptive']['decode_ms']), f(c['word-adaptive']['decode_frac'],2), f(c['byte-adaptive']['decode_ms']), f(c['byte-adaptive']['decode_frac'],2), f(c['word-adaptive']['enc_tight_ms']), f(c['byte-adaptive']['enc_tight_ms'])))
row("| word 128-way / 256-way |", "| word 128-way / 256-way | %s (%s) / %s (%s) | %s / %s | %s / %s | |" % (f(c['word128']['decode_ms']), f(c['word128']['decode_frac'],2), f(c['word256']['decode_ms']), f(c['word256']['decode_frac'],2), f(c['word128']['encode_ms']), f(c['word256']['encode_ms']), f(c['word128']['enc_tight_ms']), f(c['word256']['enc_tight_ms'])))
open(p,'w').write(s)
# DESIGN
p=R+'DESIGN.md'; s=open(p).read()
a=s.index("| **headline: word 64-way decode, 32 Ki chunks** |"); b=s.index("| word 64-way decode, 16 Ki chunks |")
s=s[:a]+"| **headline: word 64-way decode, 32 Ki chunks** | `k_decode_word64` | **%.4f** per step (wall, the judged clock; 0.3878-0.3946 on the round's other seven boxes); %.4f by HIP events; %.4f in the kernel trace of another process | **%.3f** (0.607-0.618) / %.3f / %.3f | 1.048 | VALU issue (11 VALU/round = 45 of the 46 clocks per round and SIMD) |\n" % (d['ms_per_step'], r['kernel_ms_avg'], tavg/1000, r['frac'], r['frac_kernel_events'], tfrac)+s[b:]
s=re.sub(r"`frac_kernel_events` \(0\.\d+ against 0\.\d+: the gaps between launches\), the kernel trace of another process cut to its timed dispatches gives 0\.\d+\.", "`frac_kernel_events` (%.3f against %.3f: the gaps between launches), the kernel trace of another process cut to its timed dispatches gives %.3f." % (r['frac'], r['frac_kernel_events'], tfrac), s)
a=s.index("**Evidence** (`profiles/r06_rocprof_summary.md`, `tools/r06_profile.sh`): one workload per row.  The driver's command:"); b=s.index("Sections G-I: the per-chunk-model kernels")
s=s[:a]+'''**Evidence** (`profiles/r06_rocprof_summary.md`, `tools/r06_profile.sh`): one workload per row.  The driver's command: %.1f
GB/s, %.4f ms/step, frac **%.3f**; the first (un-probed) pair %.4f ms/step = %.3f (0.3827-0.4114 over the round's profile runs: a lottery,
hence the probe; the round's seven other boxes: 0.3878-0.3946 ms/step, 0.607-0.618).  The headline alone under `rocprofv3 --kernel-trace
--stats`: the record names which dispatches of `k_decode_word64` were the timed ones, the per-dispatch trace cut to them gives
%.1f µs (min %.1f, max %.1f) = %.3f where that run's own line says %.1f µs by HIP events and %.4f ms per step; the
whole row (%s dispatches, the probe's slow pairs among them) averages %.1f µs.  Traffic 2.008e9 B = 1.048 × algorithmic.
''' % (d['value'], d['ms_per_step'], r['frac'], fp_ms, fp_fr, tavg, tmin, tmax, tfrac, levt, lstep, "{:,}".format(ndisp).replace(","," "), rowavg)+s[b:]
open(p,'w').write(s)
# profiles/README
p=R+'profiles/README.md'; s=open(p).read()
a=s.index("| round 6: the judged `bench.py` line and its full record (value"); b=s.index("| round 6: the group kernels step by step")
s=s[:a]+'''| round 6: the judged `bench.py` line and its full record (value %.1f GB/s, %.4f ms/step, roofline.frac %.3f on the step clock / %.3f by HIP events; the first un-probed pair %.4f ms/step = %.3f (0.3827-0.4114 over the round's profile runs); 0.3878-0.3946 ms/step on the round's seven other boxes, every chunk of every container of 11 configs == oracle, reference CPU 13 GB/s on 16 threads) | `r06_bench.json`, `r06_bench_details.json` |
| round 6: the same from rocprofv3 -- the headline ALONE, the per-dispatch trace cut to the timed launches: %.1f µs (%.1f-%.1f) = %.3f; traffic 1.048 ×; per-chunk-model kernels (one-kernel encoder with the chunk resident in registers 0.87 / 0.98 ms under the profiler, traffic 1.056 ×; decoders 0.67 / 0.75 ms, 1.02 ×); the group kernels of the reference's own layouts (H) and what is left on the lane kernels; derived SQ / LDS counters of the kernels whose bound DESIGN states (I) | `r06_rocprof_summary.md`, `r06_kernel_stats.csv`, `r06_headline_dispatches.json`, `r06_traffic.json` |
''' % (d['value'], d['ms_per_step'], r['frac'], r['frac_kernel_events'], fp_ms, fp_fr, tavg, tmin, tmax, tfrac)+s[b:]
open(p,'w').write(s)
print("synced", d['value'], d['ms_per_step'], r['frac'], tavg, tfrac)
