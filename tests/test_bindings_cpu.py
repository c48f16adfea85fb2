"""bindings/: the JVM side of the drop-in as FILES.  This image has no JDK,
scalac or swig, so nothing here is compiled into anything that runs and NOTHING IS PINNED by these tests: they keep the
files honest against the C ABI they bind —
  * the JNI thunks compile (gcc -fsyntax-only, C99) against include/skeres_amd.h and tests/jni_stub/jni.h, a minimal
    declaration of the JNI members they use (NOT the JDK's header);
  * every C entry point a thunk calls is declared in the header and exported by the library;
  * every native method of SkeresNative.java has exactly one thunk, and vice versa;
  * every SkeresNative method the Scala sources call exists."""
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
JNI_C = os.path.join(ROOT, "bindings", "jni", "skeres_amd_jni.c")
JAVA = os.path.join(ROOT, "bindings", "java", "com", "google", "ceres", "SkeresNative.java")


def test_jni_thunks_compile_for_syntax_against_the_header():
    cmd = ["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-fsyntax-only", "-I", os.path.join(ROOT, "tests", "jni_stub"),
           "-I", os.path.join(ROOT, "include"), JNI_C]
    out = subprocess.run(cmd, capture_output=True, text=True)
    assert out.returncode == 0, out.stderr


def test_thunks_call_only_declared_and_exported_entry_points():
    import skeres_amd as sk
    src = open(JNI_C).read()
    called = set(re.findall(r"\b(sk_[a-z0-9_]+)\s*\(", src))
    header = open(os.path.join(ROOT, "include", "skeres_amd.h")).read()
    declared = set(re.findall(r"\b(sk_[a-z0-9_]+)\s*\(", header))
    assert called <= declared, sorted(called - declared)
    lib = sk.lib()
    for name in sorted(called):
        assert hasattr(lib, name), name


def test_java_native_methods_and_thunks_correspond_one_to_one():
    thunks = re.findall(r"SK_JNI\([a-zA-Z]+, (sk[A-Za-z0-9]+)\)", open(JNI_C).read())
    thunks += re.findall(r"SK_OPT_(?:INT|DBL)\((sk[A-Za-z0-9]+),", open(JNI_C).read())
    natives = re.findall(r"public static native [A-Za-z\[\]]+ (sk[A-Za-z0-9]+)\(", open(JAVA).read())
    assert len(thunks) == len(set(thunks)) and len(natives) == len(set(natives))
    assert set(thunks) == set(natives), (sorted(set(thunks) - set(natives)), sorted(set(natives) - set(thunks)))


def test_scala_sources_call_existing_native_methods():
    natives = set(re.findall(r"public static native [A-Za-z\[\]]+ (sk[A-Za-z0-9]+)\(", open(JAVA).read()))
    used = set()
    for dirpath, _, files in os.walk(os.path.join(ROOT, "bindings", "scala")):
        for f in files:
            if f.endswith(".scala"):
                used |= set(re.findall(r"SkeresNative\.(sk[A-Za-z0-9]+)", open(os.path.join(dirpath, f)).read()))
    assert used and used <= natives, sorted(used - natives)
