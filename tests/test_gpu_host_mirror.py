"""The C++ mirror of the reference's host interface (ClientConnection -> GpuIndexingWorker -> SimilarityOutput)
replays the section-3.3 KAT on the GPU."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
HOST = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "all-pairs-similarity_amd", "host")


def test_host_selftest():
    subprocess.check_call(["make", "-C", HOST], stdout=subprocess.DEVNULL)
    out = subprocess.run([os.path.join(HOST, "host_selftest")], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "host_selftest: PASS" in out.stdout
