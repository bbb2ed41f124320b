"""Generates tests/golden/ref_streams.{json,bin}: streams written by the UNMODIFIED
reference (oracle/_ref/libryg_ref.so, built from /root/reference by oracle/Makefile)
on a small seeded input, for every format and N in {1, 2, 8, 64}.  Run here (the
container with /root/reference); the outputs are committed so that GPU-box tests
can check the HIP path against reference-made bytes without the reference.

    python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from _oracle import FMT_ALIAS, FMT_BYTE, FMT_R64, FMT_WORD, FMT_NAMES, Oracle, Ref  # noqa: E402


def main():
    ref, orc = Ref(), Oracle()
    rng = np.random.default_rng(20260925)
    # text-like skew + every byte value once (freq-1 symbols) + odd length
    data = np.concatenate([
        np.minimum(rng.geometric(0.07, 9000) - 1, 255).astype(np.uint8),
        np.arange(256, dtype=np.uint8),
        orc.gen_zipf(2747, K=256, s=1.0, seed=1),
    ])
    assert data.size == 12003
    blob = [data]
    pos = data.size
    streams = []
    for fmt, sb in ((FMT_BYTE, 14), (FMT_WORD, 12), (FMT_R64, 14), (FMT_ALIAS, 16)):
        freqs, _ = ref.build_model(data, 1 << sb)
        for n_ways in (1, 2, 8, 64):
            s = ref.encode(fmt, freqs, sb, data, n_ways)
            back, rc = ref.decode(fmt, freqs, sb, s, data.size, n_ways)
            assert rc == 0 and np.array_equal(back, data)
            streams.append({"name": "%s-sb%d-N%d" % (FMT_NAMES[fmt], sb, n_ways), "fmt": fmt, "scale_bits": sb,
                            "n_ways": n_ways, "offset": pos, "size": int(s.size),
                            "freqs": [int(v) for v in freqs]})
            blob.append(s)
            pos += s.size
    np.concatenate(blob).tofile(os.path.join(HERE, "ref_streams.bin"))
    json.dump({"blob": "ref_streams.bin", "input": [0, int(data.size)], "streams": streams,
               "made_by": "tests/golden/make_golden.py via oracle/_ref (unmodified reference headers)"},
              open(os.path.join(HERE, "ref_streams.json"), "w"))
    print("wrote %d streams, %d bytes" % (len(streams), pos))
    make_book1(ref)


def make_book1(ref):
    """tests/golden/book1_word64.bin: the reference corpus as the 64-way word-format stream the unmodified
    reference headers produce for it (435 798 bytes, sha256 84cc03c8...), plus the
    normalised frequencies at 12 / 14 / 16 bits.  GPU-box tests (no /root/reference there) decode it on the GPU --
    the result must hash to book1's sha256 -- and re-encode it into every appendix B stream."""
    import hashlib
    path = "/root/reference/book1"
    if not os.path.exists(path):
        print("book1 not found: fixture left as it is")
        return
    data = np.fromfile(path, dtype=np.uint8)
    known = json.load(open(os.path.join(HERE, "book1_golden.json")))
    assert hashlib.sha256(data.tobytes()).hexdigest() == known["input_sha256"]
    freqs = {}
    for sb in (12, 14, 16):
        f, _ = ref.build_model(data, 1 << sb)
        freqs[str(sb)] = [int(v) for v in f]
    s = ref.encode(FMT_WORD, np.array(freqs["12"], dtype=np.uint32), 12, data, 64)
    want = [e for e in known["streams"] if e["fmt"] == "word" and e["n_ways"] == 64][0]
    assert s.size == want["size"] and hashlib.sha256(s.tobytes()).hexdigest() == want["sha256"]
    s.tofile(os.path.join(HERE, "book1_word64.bin"))
    json.dump({"stream": "book1_word64.bin", "fmt": FMT_WORD, "scale_bits": 12, "n_ways": 64, "n": int(data.size),
               "stream_sha256": want["sha256"], "freqs": freqs,
               "made_by": "tests/golden/make_golden.py via oracle/_ref (unmodified reference headers)"},
              open(os.path.join(HERE, "book1_word64.json"), "w"))
    print("wrote book1_word64.bin, %d bytes" % s.size)


if __name__ == "__main__":
    main()
