"""bench_latency.py (the LoadGenerator counterpart): JSON contract, both backends, agreement."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("shape", ["template", "production"])
def test_latency_harness_contract(shape):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench_latency.py"), "--shape", shape, "--preload", "20000",
                          "--messages", "12", "--interval-ms", "2"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert d["messages"] == 12 and d["interval_ms"] == 2 and d["index_size"] == 20000
    assert d["config"]["dim"] == {"template": 1024, "production": 1 << 20}[shape] and d["config"]["theta"] == 0.7
    for b in ("gpu", "refcpu"):
        r = d[b]
        for k in ("avg_ms", "max_ms", "min_ms", "p50_ms", "p99_ms", "pairs_found"):
            assert k in r
        assert 0 < r["min_ms"] <= r["p50_ms"] <= r["p99_ms"] <= r["max_ms"]
    # the test phase replays the data set: every message finds at least its own earlier copy, in both directions
    assert d["gpu"]["pairs_found"] == d["refcpu"]["pairs_found"] >= 12
    assert d["gpu"]["avg_ms"] < 50.0  # well inside the reference's 50 ms cadence
