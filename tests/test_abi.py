"""CPU-side checks of the drop-in boundary: the C-ABI library builds for gfx950, loads, exports every symbol
include/apss.h declares, and refuses to compute without a GPU (no CPU fallback)."""
import ctypes
import os
import re

import pytest

from apss import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def so():
    return _lib.build()


def test_header_symbols_are_exported(so):
    hdr = open(os.path.join(ROOT, "include", "apss.h")).read()
    declared = set(re.findall(r"\b(apss_[a-z_]+)\s*\(", hdr))
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    L = ctypes.CDLL(so)
    for sym in declared:
        assert getattr(L, sym) is not None


def test_config_struct_matches_header():
    assert ctypes.sizeof(_lib.Config) == 64
    assert ctypes.sizeof(_lib.Stats) == 288
    assert ctypes.sizeof(_lib.GroupStats) == 424


def test_python_constants_and_stats_fields_match_the_header():
    """every APSS_FLAG_* / APSS_DOWNGRADE_* / APSS_E_* value of include/apss.h equals its mirror in apss/_lib.py, and the
    ctypes Stats lists the header's apss_stats fields in the header's order"""
    hdr = open(os.path.join(ROOT, "include", "apss.h")).read()
    seen = 0
    for name, val in re.findall(r"#define\s+APSS_((?:FLAG|DOWNGRADE|E|SYM|GROUP|EXCHANGE)_[A-Z_0-9]+)\s+\(?(-?\d+)u?\)?", hdr):
        assert getattr(_lib, name) == int(val), (name, val)
        seen += 1
    assert seen >= 27, seen
    body = re.search(r"typedef struct apss_stats \{(.*?)\} apss_stats;", hdr, re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = [m.group(1) for m in re.finditer(r"\b(?:int64_t|uint32_t|int32_t|double|float|char)\s+([a-z_0-9]+)\s*(?:\[\d+\])?\s*;", body)]
    assert fields == [f for f, _ in _lib.Stats._fields_], (fields, [f for f, _ in _lib.Stats._fields_])
    body = re.search(r"typedef struct apss_group_stats \{(.*?)\} apss_group_stats;", hdr, re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = [m.group(1) for m in re.finditer(r"\b(?:int64_t|uint32_t|int32_t|double|float|char)\s+([a-z_0-9]+)\s*(?:\[[^\]]+\])?\s*;", body)]
    assert fields == [f for f, _ in _lib.GroupStats._fields_], (fields, [f for f, _ in _lib.GroupStats._fields_])


def test_stats_struct_size_protects_an_older_caller(so):
    """apss_stats / apss_group_stats grow at their end; the caller names ITS size and the library writes no further (checked
    without a GPU on the argument validation only: a handle cannot exist here)"""
    hdr = open(os.path.join(ROOT, "include", "apss.h")).read()
    for name in ("apss_stats", "apss_group_stats"):
        body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (name, name), hdr, re.S).group(1)
        assert re.match(r"\s*int32_t struct_size;", body), name
    src = open(os.path.join(_lib.CSRC, "apss_hip.hip")).read()
    assert "std::min<int32_t>(caller, (int32_t)sizeof(apss_stats))" in src


def test_no_cpu_fallback(so):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from apss.engine import ApssError, ApssIndex
    with pytest.raises(ApssError) as e:
        ApssIndex(16, 0.5)
    assert e.value.code == _lib.E_DEVICE


def test_group_needs_a_gpu_too(so):
    """apss_group_create without a device: APSS_E_DEVICE and a message, never a CPU path"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from apss.engine import ApssError, ApssGroup
    with pytest.raises(ApssError) as e:
        ApssGroup(16, 0.5, [0, 1])
    assert e.value.code == _lib.E_DEVICE and "no usable HIP device" in str(e.value)


def test_product_never_imports_oracle():
    """the product path must not import, link or load anything under oracle/ (mentions in prose are fine)"""
    pkg = os.path.join(ROOT, "all-pairs-similarity_amd")
    pat = re.compile(r"(^\s*(from|import)\s+oracle\b|libapss_oracle|apss_oracle\.h|oracle/)", re.M)
    for d, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h", ".scala", ".c")):
                txt = open(os.path.join(d, f)).read()
                assert not pat.search(txt), (d, f)


def test_one_hip_runtime_whatever_the_import_order(so):
    """libapss_hip.so loaded BEFORE torch and torch imported afterwards: still one libamdhip64 image in the process
    (it used to depend on importing torch first)"""
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r); from apss import _lib; _lib.lib(); a = _lib.hip_runtimes_loaded(); "
            "import torch; b = _lib.hip_runtimes_loaded(); print(len(a), len(b), a == b)") % os.path.join(ROOT, "all-pairs-similarity_amd")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-1000:]
    assert out.stdout.split() == ["1", "1", "True"], out.stdout


def test_every_included_header_is_in_the_staleness_list():
    """an edit to any file libapss_hip.so is compiled from must rebuild it: the #include "..." closure of apss_hip.hip is
    inside _lib.build_sources(), and the Makefile names the same files"""
    csrc = _lib.CSRC
    seen, todo = set(), [os.path.join(csrc, "apss_hip.hip"), os.path.join(csrc, "apss_group.hip")]
    while todo:
        f = os.path.normpath(todo.pop())
        if f in seen:
            continue
        seen.add(f)
        for inc in re.findall(r'^\s*#include\s+"([^"]+)"', open(f).read(), re.M):
            todo.append(os.path.join(os.path.dirname(f), inc))
    srcs = {os.path.normpath(s) for s in _lib.build_sources()}
    assert seen <= srcs, seen - srcs
    mk = open(os.path.join(csrc, "Makefile")).read()
    rule = []
    for obj in ("apss_hip.o", "apss_group.o"):
        rule += re.search(r"^%s:(.*)$" % re.escape(obj), mk, re.M).group(1).split()
    assert {os.path.normpath(os.path.join(csrc, d)) for d in rule} == seen, (rule, seen)
    link = re.search(r"^libapss_hip\.so:(.*)$", mk, re.M).group(1).split()
    assert link == ["apss_hip.o", "apss_group.o"], link
