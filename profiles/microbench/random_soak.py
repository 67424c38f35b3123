"""One-off soak: the randomised parity sweeps of tests/test_gpu_random.py over many more seeds (not part of the suite).
Usage: python profiles/microbench/random_soak.py LO HI [head]   (head: the streams with a forced dense-head block)"""
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "all-pairs-similarity_amd"))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402,F401
import pytest  # noqa: E402
import test_gpu_random as T  # noqa: E402
from oracle import oracle  # noqa: E402

lo, hi = int(sys.argv[1]), int(sys.argv[2])
fn = T.test_random_streams_with_a_forced_head_block if len(sys.argv) > 3 and sys.argv[3] == "head" else T.test_random_streams
bad = 0
for seed in range(lo, hi):
    mp = pytest.MonkeyPatch()
    try:
        fn(oracle, seed, mp)
    except AssertionError as e:
        bad += 1
        print("seed", seed, "FAILED", str(e)[:200], flush=True)
    finally:
        mp.undo()
print("soak", lo, hi, "failures", bad)
