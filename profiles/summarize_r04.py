"""Turns the raw rocprofv3 output that profiles/collect_r04.sh left under gpurun_out/r04/<workload> into the committed
summaries profiles/r04_*.  Run from the repo root after the gpurun call(s):  python profiles/summarize_r04.py c3 c3z1"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

DST = "profiles"
CLOCK_HZ, CUS = 2.4e9, 256  # for "fraction of issue time" figures: kernel time x nominal clock x CUs


def latest(pattern):
    fs = sorted(glob.glob(pattern, recursive=True), key=os.path.getmtime)
    return fs[-1] if fs else None


def pmc(src, dirname):
    f = latest(os.path.join(src, dirname, "**", "*counter_collection.csv"))
    if not f:
        return {}
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    n = collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"]][r["Counter_Name"]] += float(r["Counter_Value"])
        n[r["Kernel_Name"]].add(r["Dispatch_Id"])
    return {k: {c: v / len(n[k]) for c, v in d.items()} for k, d in agg.items()}  # per launch


def pick(d, pat):
    for k, v in d.items():
        if pat in k:
            return v
    return {}


def main(workloads):
    path = os.path.join(DST, "r04_probe_traffic.json")
    allout = json.load(open(path)) if os.path.exists(path) else {}
    for w in workloads:
        src = os.path.join("gpurun_out", "r04", w)
        stats = latest(os.path.join(src, "trace", "**", "*kernel_stats.csv"))
        shutil.copy(stats, os.path.join(DST, "r04_kernel_stats_%s.csv" % w))
        bench = json.loads([l for l in open(os.path.join(src, "bench.json")) if l.startswith("{")][-1])
        rows = list(csv.DictReader(open(stats)))
        counters = {}
        for d in ("pmc_fetch", "pmc_write", "pmc_l2", "sq1", "sq2", "sq3", "sq4"):
            for k, v in pmc(src, d).items():
                counters.setdefault(k, {}).update(v)
        out = {"workload": bench["config"]["workload"], "kernels": {}}
        for pat, label in (("k_probe_even", "filter"), ("k_probe_coarse", "filter"), ("k_probe_wave", "exact"), ("k_head_gemm", "head")):
            if label in out["kernels"]:
                continue  # (the thin-round kernel when the launch took it, else the all-wave staging kernel)
            kr = [r for r in rows if pat in r["Name"]]
            c = pick(counters, pat)
            if not kr or not c:
                continue
            kr = kr[0]
            avg_s = float(kr["AverageNs"]) * 1e-9
            e = {"kernel": kr["Name"], "calls": int(kr["Calls"]), "avg_ms_rocprof": avg_s * 1e3}
            if "FETCH_SIZE" in c:
                # MI355X_MICROARCH.md "HBM": FETCH_SIZE is in KB and tallies every 128-B read request at 64 B on gfx950 -> x2;
                # WRITE_SIZE is exact (profiles/microbench/fetch_calib.hip confirmed it for this access shape in round 1)
                e["FETCH_SIZE_KB_raw"], e["WRITE_SIZE_KB_raw"] = c["FETCH_SIZE"], c.get("WRITE_SIZE", 0.0)
                e["traffic_bytes_per_launch_corrected"] = c["FETCH_SIZE"] * 1024 * 2 + c.get("WRITE_SIZE", 0.0) * 1024
                e["memory_side_GBps"] = e["traffic_bytes_per_launch_corrected"] / avg_s / 1e9
            if "TCC_HIT_sum" in c:
                e["l2_hit_rate"] = c["TCC_HIT_sum"] / max(1.0, c["TCC_HIT_sum"] + c["TCC_MISS_sum"])
            cu_cycles = avg_s * CLOCK_HZ * CUS
            for name in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES",
                         "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_ANY",
                         "SQ_LDS_BANK_CONFLICT", "SQ_INSTS_VALU_MFMA_MOPS_BF16", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_INSTS_MFMA"):
                if name in c:
                    e[name] = c[name]
            if "SQ_ACTIVE_INST_LDS" in c:
                # SQ_ACTIVE_INST_* count quad-cycles (guide, cycle constants): x4 = cycles the LDS pipeline was issuing
                e["lds_issue_frac"] = 4.0 * c["SQ_ACTIVE_INST_LDS"] / cu_cycles
                e["valu_busy_frac"] = 4.0 * c["SQ_ACTIVE_INST_VALU"] / (cu_cycles * 4)  # four SIMDs per CU
            if "SQ_LDS_BANK_CONFLICT" in c and "SQ_ACTIVE_INST_LDS" in c:
                e["lds_bank_conflict_share_of_lds_time"] = c["SQ_LDS_BANK_CONFLICT"] / (4.0 * c["SQ_ACTIVE_INST_LDS"])
            if "SQ_VALU_MFMA_BUSY_CYCLES" in c:
                e["mfma_busy_frac"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (cu_cycles * 4)
            out["kernels"][label] = e
        flt = out["kernels"].get("filter", {})
        ent = {"algorithmic_bytes_per_launch": 8 * bench.get("device_posting_visits_per_step", bench["posting_visits_per_step"]) // max(1, bench["roofline"].get("launches", 1)
                                                                                          if bench["roofline"]["bound"] == "lds" else 1)}
        if "traffic_bytes_per_launch_corrected" in flt:
            ent["traffic_bytes_per_launch_corrected"] = flt["traffic_bytes_per_launch_corrected"]
            ent["traffic_over_algorithmic"] = flt["traffic_bytes_per_launch_corrected"] / ent["algorithmic_bytes_per_launch"]
        for k in ("lds_issue_frac", "valu_busy_frac", "lds_bank_conflict_share_of_lds_time", "l2_hit_rate"):
            if k in flt:
                ent[k] = flt[k]
        if "traffic_bytes_per_launch_corrected" in out["kernels"].get("exact", {}):
            ent["exact_traffic_bytes_per_launch_corrected"] = out["kernels"]["exact"]["traffic_bytes_per_launch_corrected"]
        ent["detail"] = out
        sha = os.path.join(src, "csrc.sha256")
        if os.path.exists(sha):
            ent["csrc_sha256"] = open(sha).read().strip()  # the sources the counters were collected on (bench.py checks it)
        allout[w] = ent
        json.dump(bench, open(os.path.join(DST, "r04_bench_%s.json" % w), "w"), indent=1)
        print(w, json.dumps({k: v for k, v in ent.items() if k != "detail"}, indent=1))
        for label, e in out["kernels"].items():
            print("  ", label, {k: (round(v, 4) if isinstance(v, float) else v) for k, v in e.items() if k in (
                "avg_ms_rocprof", "calls", "memory_side_GBps", "l2_hit_rate", "lds_issue_frac", "valu_busy_frac", "mfma_busy_frac",
                "lds_bank_conflict_share_of_lds_time")})
    json.dump(allout, open(path, "w"), indent=1)


if __name__ == "__main__":
    main(sys.argv[1:] or ["c3"])
