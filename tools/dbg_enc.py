import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import ryg_rans_amd as R
from _oracle import Oracle, FMT_R64
oracle = Oracle()
ctx = R.Context(0)
zipf = oracle.gen_zipf(600000 + 333, K=256, s=1.0, seed=23)
for sb, chunk, ways in ((16, 4096, 2), (16, 512, 2), (15, 4096, 2), (14, 4096, 2), (16, 4096, 1), (16, 4096, 64), (16, 2048, 2)):
    f, _ = oracle.normalize(oracle.count_freqs(zipf, 256), 1 << sb)
    om, gm = oracle.model(f, sb), ctx.model(FMT_R64, f, sb)
    want, offs, lens = oracle.encode_chunked(FMT_R64, om, zipf, ways, chunk, align=16)
    cont, o2, l2, total = ctx.encode(gm, torch.from_numpy(zipf).cuda(), ways, chunk)
    got = cont[:total].cpu().numpy()
    l2 = l2.cpu().numpy().astype(np.uint32)
    bad = []
    for c in range(len(lens)):
        o, ln = int(offs[c]), int(lens[c])
        if total != want.size or l2[c] != lens[c] or not np.array_equal(got[o:o + ln], want[o:o + ln]):
            d = np.nonzero(got[o:o + ln] != want[o:o + ln])[0] if l2[c] == lens[c] else []
            bad.append((c, ln, int(l2[c]), (int(d[0]), int(d[-1]), len(d)) if len(d) else None))
    print("sb", sb, "chunk", chunk, "ways", ways, "total", total, want.size, "bad chunks", len(bad), bad[:6], flush=True)
