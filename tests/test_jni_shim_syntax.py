"""The JNI shim cannot be built here (no JDK), but it can be type-checked: gcc -fsyntax-only against a minimal declaration
of the JNI functions it uses (tests/stubs/jni.h) and the real include/apss.h; and its native method names must be the ones
NativeApss.scala declares."""
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
JVM = os.path.join(ROOT, "all-pairs-similarity_amd", "jvm")


def test_jni_shim_type_checks():
    out = subprocess.run(["gcc", "-std=c11", "-fsyntax-only", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "tests", "stubs"),
                          "-I", os.path.join(ROOT, "include"), os.path.join(JVM, "apss_jni.c")], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr


def test_jni_names_match_the_scala_declarations():
    c = open(os.path.join(JVM, "apss_jni.c")).read()
    scala = open(os.path.join(JVM, "NativeApss.scala")).read()
    natives = set(re.findall(r"@native def (\w+)", scala))
    exported = set(re.findall(r"Java_cpslab_gpu_NativeApss_(\w+)\(", c))
    assert natives == exported and natives == {
        "create", "destroy", "lastError", "submit", "fetch", "setHeadTerms", "setHeadFold", "headTerms",
        "createGroup", "destroyGroup", "groupLastError", "groupSubmit", "groupFetch", "groupStats"}
    # no critical sections: the library calls block on the GPU
    assert "GetPrimitiveArrayCritical(" not in c.split("*/", 1)[1]


def test_scala_docs_name_the_values_the_library_accepts():
    """the fold widths in NativeApss.scala are the ones apss_set_head_fold takes (the doc once named 128 | 256)"""
    scala = open(os.path.join(JVM, "NativeApss.scala")).read()
    src = open(os.path.join(ROOT, "all-pairs-similarity_amd", "csrc", "apss_hip.hip")).read()
    assert "columns != 0 && columns != 64 && columns != 128 && columns != 192" in src
    assert "64 | 128 | 192" in scala and "128 | 256, 0 = default" not in scala
