"""Map the placement classes of a device's memory in allocation order (profiles/r04_allocation.md: the headline decode is
4-6 % faster when its container and its output lie in DIFFERENT classes; a class is a window of allocation order).

    python tools/class_map.py [--gib 240] [--cont-every 40]

Allocates `gib` output buffers of 1 GiB in a row with a copy of the container after every `cont_every` of them, times the
headline decode for every (container, output) pair through rans_amd_probe_placement, and prints one row per container.
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ryg_rans_amd as R  # noqa: E402
from bench import gen_zipf  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gib", type=int, default=240)
    ap.add_argument("--cont-every", type=int, default=40)
    ap.add_argument("--chunk", type=int, default=32768)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    ctx = R.Context(0)
    n = 1 << 30
    d = gen_zipf(torch, n, 256, 1.0, 1, dev)
    f, _ = R.normalize_freqs(ctx.count_freqs_device(d, 256), 4096)
    m = ctx.model(R.FMT_WORD, f, 12)
    cont, offs, lens, total = ctx.encode(m, d, 64, a.chunk)
    free, tot = torch.cuda.mem_get_info()
    gib = min(a.gib, int(free / (1 << 30)) - 12)
    print(f"free {free / 2**30:.1f} GiB of {tot / 2**30:.1f}; mapping {gib} GiB", flush=True)
    conts, outs, order = [cont], [], ["c0"]
    for i in range(gib):
        outs.append(torch.empty(n, dtype=torch.uint8, device=dev))
        order.append(f"o{i}")
        if (i + 1) % a.cont_every == 0:
            conts.append(cont.clone())
            order.append(f"c{len(conts) - 1}")
    print("allocation order:", " ".join(order))
    print("addresses: containers", [hex(t.data_ptr()) for t in conts])
    print("addresses: outputs (every 8th)", [hex(t.data_ptr()) for t in outs[::8]])
    for _ in range(300):  # settle the clocks
        ctx.decode(m, cont, total, offs, lens, n, 64, a.chunk, d_out=outs[0], sync=False)
    torch.cuda.synchronize()
    ci, oi, mat = ctx.probe_placement(m, conts, total, offs, lens, n, 64, a.chunk, outs, launches=6, sweeps=2)
    lo = min(min(r) for r in mat)
    hi = max(max(r) for r in mat)
    print(f"best pair c{ci} o{oi}; min {lo:.4f} max {hi:.4f} ms")
    for i, row in enumerate(mat):
        # one character per output: '.' within 2 % of the global minimum, 'o' within 4.5 %, '#' beyond
        s = "".join("." if v < lo * 1.02 else ("o" if v < lo * 1.045 else ("#" if v < lo * 1.09 else "X")) for v in row)
        print(f"c{i} {s}")
    for i, row in enumerate(mat):
        print(f"c{i} ms " + " ".join(f"{v:.3f}" for v in row))


if __name__ == "__main__":
    main()
