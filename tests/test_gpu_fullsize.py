"""BASELINE.json configs[4] AT FULL SIZE inside the GPU suite: "Synthetic N=10M, dim=1M, nnz=200 power-law, theta=0.9 ... with
dense-tail bf16 MFMA batched rescoring", and the same shape with uniform terms (C5).  The oracle needs hours at this size, so
parity is checked through size-independent properties (profiles/fullsize_powerlaw.py, profiles/fullsize_stratified.py):
every reported score equals the exact float64 dot of its two rows (<= 1e-5), nothing is reported twice or below theta, and
every planted near-duplicate pair whose exact dot reaches theta is reported, in both directions.  What the properties cannot
see (a true pair that is neither reported nor planted) is what the reduced-size tests check against the oracle."""
import importlib.util
import os

import pytest

pytestmark = pytest.mark.gpu
PROFILES = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")


def _load(name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(PROFILES, name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_configs4_power_law_at_full_size():
    """N = 10M x 200 Zipf(1) terms over dim = 1M, theta = 0.9, one GPU: the library's policy takes a dense-head block (the
    frequent terms as a bf16 MFMA contraction), the tail goes through the sparse filter, survivors are re-scored exactly"""
    import torch
    out = _load("fullsize_powerlaw").run(10_000_000, 1_000_000, 200, 0.9)
    torch.cuda.empty_cache()
    assert out["missing"] == 0 and out["planted_pairs_required"] > 400_000
    assert out["max_abs_score_error"] <= 1e-5
    assert out["head_terms"] > 0 and out["head_ms"] > 0 and out["symmetric"] == 1
    assert out["result_pairs"] >= out["planted_pairs_required"]
    assert out["wall_s"] < 60.0  # (20 s on an idle MI355X; a regression to the all-sparse path would need minutes)


def test_c5_uniform_at_full_size():
    """N = 10M x 200 uniform (stratified) terms over dim = 1M, theta = 0.9: 4e12 posting visits through the sparse filter alone;
    every planted pair, nothing else, posting visits == sum df^2"""
    import torch
    out = _load("fullsize_stratified").run(10_000_000, 1_000_000, 200, 0.9)
    torch.cuda.empty_cache()
    assert out["missing"] == 0 and out["unexpected"] == 0 and out["planted_pairs_required"] > 400_000
    assert out["max_abs_score_error"] <= 1e-5 and out["symmetric"] == 1
    assert out["posting_visits"] == out["posting_visits_analytic"]
    assert out["wall_s"] < 30.0
