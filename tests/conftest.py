import os
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (HERE, ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # A fresh checkout has no binaries (they are git-ignored): build the product library and the
    # checkers once, exactly as __graft_entry__.build() does.  Nothing is rebuilt when they exist.
    need = [os.path.join(ROOT, "ryg_rans_amd", "lib", "libryg_rans_amd.so"),
            os.path.join(ROOT, "oracle", "librans_oracle.so")]
    if not all(os.path.exists(f) for f in need):
        import __graft_entry__
        __graft_entry__.build()


@pytest.fixture(scope="session")
def oracle():
    from _oracle import Oracle
    return Oracle()


@pytest.fixture(scope="session")
def ref():
    from _oracle import Ref
    if not Ref.available():
        pytest.skip("oracle/_ref/libryg_ref.so not built (needs /root/reference)")
    return Ref()


@pytest.fixture(scope="session")
def book1(oracle):
    """The reference corpus.  Where /root/reference exists it is read from there; elsewhere (the GPU box) it is
    recovered from tests/golden/book1_word64.bin -- the 64-way word stream the unmodified reference made of it
     -- by the CPU oracle, and must hash to book1's published sha256."""
    import hashlib
    import json

    import numpy as np
    path = "/root/reference/book1"
    if os.path.exists(path):
        return np.fromfile(path, dtype=np.uint8)
    meta = json.load(open(os.path.join(HERE, "golden", "book1_word64.json")))
    stream = np.fromfile(os.path.join(HERE, "golden", meta["stream"]), dtype=np.uint8)
    assert hashlib.sha256(stream.tobytes()).hexdigest() == meta["stream_sha256"]
    om = oracle.model(np.array(meta["freqs"]["12"], dtype=np.uint32), 12)
    data = oracle.decode(meta["fmt"], om, stream, meta["n"], meta["n_ways"])
    known = json.load(open(os.path.join(HERE, "golden", "book1_golden.json")))
    assert hashlib.sha256(data.tobytes()).hexdigest() == known["input_sha256"]
    return data
