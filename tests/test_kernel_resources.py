"""Guards the occupancy the probe kernels are designed for: two 512-thread workgroups per CU = 4 waves per SIMD
needs <= 128 VGPRs and no scratch.  (A harmless-looking edit once tipped k_probe_coarse to 130 VGPRs: one workgroup
per CU, 1.6x slower.)  hipcc cross-compiles for gfx950 without a GPU."""
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "all-pairs-similarity_amd", "csrc")


def test_speed_kernels_keep_two_workgroups_per_cu(tmp_path):
    out = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-S", "--cuda-device-only",
                          "-Rpass-analysis=kernel-resource-usage", "-o", str(tmp_path / "k.s"),
                          os.path.join(CSRC, "apss_hip.hip")], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    cur, res = None, {}
    for line in out.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = m.group(1)
            res[cur] = {}
        for key in ("VGPRs", "ScratchSize \\[bytes/lane\\]", "Occupancy \\[waves/SIMD\\]", "LDS Size \\[bytes/block\\]"):
            m = re.search(r"\s%s: (\d+)" % key, line)
            if m and cur:
                res[cur][key.split(" ")[0]] = int(m.group(1))
    checked = heads = 0
    for name, r in res.items():
        two_per_cu = ("k_probe_coarseILi512" in name) or ("k_probe_waveILi512ELi5" in name) or ("k_probe_evenILi512" in name)
        # the long-query instantiation (VROWS, the boolean after the chunk size) is allowed its 8 B/lane of scratch: it is
        # kept separate exactly so that the common kernel stays clear of the register edge
        vrows = re.search(r"k_probe_coarseILi\d+ELi\d+ELi\d+ELi\d+ELb[01]ELi\d+ELb1(ELb[01])+EEE", name) is not None
        if vrows:
            assert r["VGPRs"] <= 128 and r["ScratchSize"] <= 16 and r["Occupancy"] >= 4, (name, r)
            continue
        if two_per_cu:
            assert r["VGPRs"] <= 128 and r["ScratchSize"] == 0 and r["Occupancy"] >= 4, (name, r)
            assert 2 * r["LDS"] <= 160 * 1024, (name, r)  # two workgroups' static LDS in the CU's 160 KB
            checked += 1
        if "k_probe" in name:
            assert r["ScratchSize"] == 0, (name, r)
        if "k_head_gemm" in name:
            # the dense-head contraction: one 8-wave workgroup per CU = two waves per SIMD needs <= 256 VGPRs, no scratch (the
            # A fragments of 64 query slots alone are 128 of them at 256 columns), and its tile buffers + reporting scratch
            # inside one CU's LDS
            assert r["VGPRs"] <= 256 and r["ScratchSize"] == 0 and r["Occupancy"] >= 2 and r["LDS"] <= 160 * 1024, (name, r)
            heads += 1
    assert checked >= 4 and heads >= 4
