"""bench.py prints ONE JSON line with the fields the driver reads (reduced workloads so the tests take seconds)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DRIVER_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
               "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline")


def _one_line(out):
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]  # (gloo rehearsals print connection chatter to stdout)
    assert len(lines) == 1, out.stdout[-2000:] + out.stderr[-2000:]
    return json.loads(lines[0])


def test_bench_json_contract():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--n", "60000", "--steps", "2", "--warmup", "1",
                          "--cpu-seconds", "1"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    d = _one_line(out)
    for k in DRIVER_KEYS:
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["vs_baseline"] is None and d["data"] == "synthetic" and "workload" in d["config"]
    r = d["roofline"]
    # the sparse filter: the busiest unit is the LDS pipeline; `frac` is the algorithmic (8 B per visit) figure,
    # `measured_frac` the counter traffic (only quoted when a committed profile matches the workload byte for byte)
    # (k_probe_even when the launch took the F-staging-waves kernel -- uniform C3 does --, else k_probe_coarse)
    assert r["bound"] == "lds" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert "k_probe_even" in r["kernel"] or "k_probe_coarse" in r["kernel"]
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and r["achieved"] > 0
    assert "measured_frac" in r and "traffic" in r and "frac_note" in r
    if r["measured_frac"] is not None:
        assert 0 < r["measured_frac_of_copy_peak"] <= 1.0 and r["measured_frac"] <= 1.0
    ex = r["exact_accum"]  # the fp32-accumulate sibling, same run
    assert "k_probe_wave" in ex["kernel"] and ex["probe_kernel_ms"] > 0 and 0 < ex["frac"] < 1.0
    assert ex["result_pairs"] == d["result_pairs_per_step"]
    assert "32768" in str(d["config"]["tile_rows"])
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and "sample" in c
    assert d["value"] > 1e9  # the north-star floor, even on this reduced workload
    # `value` = candidate pairs the DEVICE decided per second (a whole-batch join is symmetric: every unordered tile pair is
    # met once); the reference-equivalent count (both directions, as IndexingWorkerActor scores them) and the two-directional
    # sibling ride along at the top level
    ref_eq = d["candidate_pairs_per_step"] / (d["ms_per_step"] * 1e-3)
    assert abs(d["value_reference_equivalent"] - ref_eq) / ref_eq < 1e-6
    share = d["device_posting_visits_per_step"] / d["posting_visits_per_step"]
    assert 0.5 <= share <= 1.0 and abs(d["device_share_of_candidate_pairs"] - share) < 1e-12
    assert abs(d["value"] - ref_eq * share) / d["value"] < 1e-6 and "value_counts" in d
    assert abs(d["posting_visits_per_s"] - d["device_posting_visits_per_step"] / (d["ms_per_step"] * 1e-3)) / d["posting_visits_per_s"] < 1e-6
    if share < 1.0:
        assert d["value_two_directional"] > 0 and d["ms_per_step_two_directional"] > d["ms_per_step"] * 0.9


def test_bench_skewed_workload_reports_an_mfma_roofline():
    """C3 with Zipf(1) terms at reduced N: the dense-head block takes the frequent terms, the line names both kernels"""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "c3z1", "--n", "50000", "--steps", "2",
                          "--warmup", "1", "--cpu-seconds", "1"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    d = _one_line(out)
    for k in DRIVER_KEYS:
        assert k in d, k
    assert d["head_terms"] in (64, 128, 256, 512, 1024, 2048, 4096, 8192, 16384, 32768) and d["head_pairs_per_step"] > 0
    # the block's rows are the INT8 rendering (rounded up; v_mfma_i32_32x32x32_i8): priced against the int8 peak, 2 x bf16
    assert d["head_int8"] == 1 and d["dtype"] == "u16+i8+f32"
    roofs = [d["roofline"]] + [d[k] for k in ("roofline_sparse_filter", "roofline_dense_head") if k in d]
    assert len(roofs) == 2 and {r["bound"] for r in roofs} == {"lds", "mfma"}
    m = [r for r in roofs if r["bound"] == "mfma"][0]
    assert m["unit"] == "TOP/s" and m["peak"] == 5000.0 and 0 < m["frac"] < 1.0 and "k_head_gemm" in m["kernel"] and "i8" in m["kernel"]
    assert abs(m["frac"] - m["achieved"] / m["peak"]) < 1e-9
    assert d["value"] > 1e9 and d["result_pairs_per_step"] > 100


def test_bench_stall_exits_nonzero():
    """a run that does not finish inside its deadline must not report success: watchdog, exit code 3"""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--n", "60000", "--steps", "1", "--warmup", "0",
                          "--deadline", "0.05"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 3, (out.returncode, out.stderr[-500:])
    assert "stalled in phase" in out.stderr


def test_bench_two_ranks_report_both_layouts():
    """N = 2 rehearsal on one GPU (gloo: both ranks drive GPU 0, collectives on CPU tensors): the headline is the
    term-range-sharded layout with its all-gather + all-reduce, the candidate-range layout rides along, same result"""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                          "127.0.0.1", "--master-port", "29613", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rows", "40000",
                          "--steps", "2", "--warmup", "1", "--backend", "gloo", "--cpu-seconds", "1"],
                         capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    d = _one_line(out)
    for k in DRIVER_KEYS:
        assert k in d, k
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["cpu_baseline"]["value"] > 0
    assert "2 term-range shards x 1 candidate ranges" in d["config"]["parallelism"] and "all-reduce" in d["collectives"]
    ex = d["exchange"]
    assert ex["term_shards"] == 2 and ex["all_reduce_bytes"] > 0 and ex["union"] >= d["result_pairs_per_step"] > 0
    comp = d["candidate_range_layout"]
    assert comp["grid"] == "1 term-range shards x 2 candidate ranges" and comp["value"] > 0
    assert comp["result_pairs_per_step"] == d["result_pairs_per_step"]


def test_bench_two_ranks_skewed_terms_take_the_dense_head_block():
    """N = 2 rehearsal (gloo) of C3 with Zipf(1) terms at reduced N -- BASELINE.json configs[4]'s layout: term-range shards
    AND the dense-head block, rank 0's policy choice broadcast, the block's contraction cut over the two ranks by candidate
    tile.  The line carries an MFMA roofline per rank next to the sparse filter's, and the candidate-range layout (plain
    handles that decide for themselves) reports the same result set"""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                          "127.0.0.1", "--master-port", "29617", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "c3z1",
                          "--rows", "50000", "--steps", "2", "--warmup", "1", "--backend", "gloo", "--cpu-seconds", "1"],
                         capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    d = _one_line(out)
    for k in DRIVER_KEYS:
        assert k in d, k
    assert d["n_gpus"] == 2 and d["head_terms"] in (64, 128, 256, 512, 1024, 2048, 4096, 8192, 16384, 32768) and d["head_kernel_ms_slowest_rank"] > 0
    roofs = [d["roofline"]] + [d[k] for k in ("roofline_sparse_filter", "roofline_dense_head") if k in d]
    assert len(roofs) == 2 and {r["bound"] for r in roofs} == {"lds", "mfma"}
    m = [r for r in roofs if r["bound"] == "mfma"][0]
    assert 0 < m["frac"] < 1.0 and "k_head_gemm" in m["kernel"] and "% 2 == rank" in m["kernel"]
    assert d["candidate_range_layout"]["result_pairs_per_step"] == d["result_pairs_per_step"] > 100


def test_bench_four_ranks_report_three_layouts():
    """N = 4 rehearsal (gloo, four ranks on GPU 0): the contract's four term-range shards as `value`, the candidate-range layout
    and the 2 x 2 grid (two term ranges x two candidate ranges, partial scores all-reduced inside each pair) beside it; the
    three layouts report the same result set"""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "4", "--master-addr",
                          "127.0.0.1", "--master-port", "29619", os.path.join(ROOT, "bench.py"), "--gpus", "4", "--rows", "40000",
                          "--steps", "2", "--warmup", "1", "--backend", "gloo", "--cpu-seconds", "1"],
                         capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    d = _one_line(out)
    assert d["n_gpus"] == 4 and "4 term-range shards x 1 candidate ranges" in d["config"]["parallelism"]
    mid, comp = d["grid_2_term_ranges_layout"], d["candidate_range_layout"]
    assert mid["grid"] == "2 term-range shards x 2 candidate ranges" and comp["grid"] == "1 term-range shards x 4 candidate ranges"
    assert mid["result_pairs_per_step"] == comp["result_pairs_per_step"] == d["result_pairs_per_step"] > 0
    assert mid["exchange"]["term_shards"] == 2 and mid["exchange"]["all_reduce_bytes"] > 0


def test_bench_gpus_2_typed_without_a_launcher():
    """`python bench.py --gpus 2` as typed (no torch.distributed.run around it): the ranks are started as a child process and
    the line comes back with n_gpus = 2 and rc 0 (gloo rehearsal: both ranks drive GPU 0)"""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--rows", "40000",
                          "--steps", "2", "--warmup", "1", "--cpu-seconds", "1"], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    d = _one_line(out)
    for k in DRIVER_KEYS:
        assert k in d, k
    assert d["n_gpus"] == 2 and d["exchange"]["term_shards"] == 2 and d["result_pairs_per_step"] > 0
    assert d["value"] <= d["value_reference_equivalent"]


def test_bench_group_engine_in_one_process():
    """--engine group: N members of ONE apss_group (the term-sharded index behind the C ABI, exchange below the boundary) in
    one process; on the one-GPU box the members share GPU 0 (rehearsal: exchange by copies).  Same result set as one GPU."""
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--rows", "60000", "--steps", "1", "--warmup", "1",
                          "--no-cpu-baseline", "--no-exact-row", "--no-two-directional-row"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert one.returncode == 0, one.stderr[-2000:]
    d1 = _one_line(one)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--engine", "group", "--share-device", "--rows", "60000",
                          "--steps", "2", "--warmup", "1", "--cpu-seconds", "1"], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    d = _one_line(out)
    for k in DRIVER_KEYS:
        assert k in d, k
    assert d["n_gpus"] == 4 and d["scaling"] == "strong" and "apss_group" in d["config"]["parallelism"]
    g = d["group"]
    assert g["devices"] == [0, 0, 0, 0] and "copies" in g["exchange"] and "rehearsal" in g
    assert len(g["term_cuts"]) == 5 and g["union"] >= d["result_pairs_per_step"] > 0 and g["all_reduce_bytes"] == 4 * g["union"]
    assert d["result_pairs_per_step"] == d1["result_pairs_per_step"] and d["candidate_pairs_per_step"] == d1["candidate_pairs_per_step"]
    assert d["posting_visits_per_step"] == d1["posting_visits_per_step"]
