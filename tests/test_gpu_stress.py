"""Randomised GPU-vs-oracle parity sweep (tools/stress.py) as part of the GPU suite: random format,
scale_bits, source skew, n, N (any value in 1..512), chunk size, buffer misalignment and lane-kernel
generation; GPU encode must equal the oracle's bytes and both containers must decode to the input; since round 4 every
case also goes through the slot layout (every chunk == the oracle's stream in its slot), its compaction (== the compact
container) and the decode of a random chunk range of the slot container; since round 5 also through SIZED slots (the model's
slot size or a random one that overflows some or all chunks: every chunk == the oracle's stream wherever it lies).
(The sweep found the N = 192/320/384/448 encoder bug that the hand-picked cases had missed.)"""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))


@pytest.mark.parametrize("seed", [11, 12])
def test_random_parity_sweep(seed, oracle):
    import torch
    assert torch.cuda.is_available()
    import stress
    assert stress.run(400, seed, oracle=oracle) == 0


def test_random_parity_sweep_big(oracle):
    """The same with inputs of millions of symbols mixed in: several rounds of every persistent grid, and every third
    case in BASELINE config 2's shape (2-way rans64, whole batches of 64 full chunks + a tail) so that the dedicated
    lane decoder / encoder and both placements of the lane encoders (RANS_AMD_OPT_LANE_FUSED_PLACEMENT) are drawn."""
    import torch
    assert torch.cuda.is_available()
    import stress
    assert stress.run(90, 13, oracle=oracle, big=True) == 0
