"""bindings/: the JVM side of the drop-in as FILES.  This image has no JDK,
scalac or swig, so the Scala / Java sources are not compiled and NOTHING IS PINNED by these tests (the JNI thunks ARE executed,
against a mock JNIEnv: tests/test_jni_mock.py): they keep the files honest against the C ABI they bind —
  * the JNI thunks compile (gcc -fsyntax-only, C99) against include/skeres_amd.h and tests/jni_stub/jni.h, a minimal
    declaration of the JNI members they use (NOT the JDK's header);
  * every C entry point a thunk calls is declared in the header and exported by the library;
  * every native method of SkeresNative.java has exactly one thunk, and vice versa;
  * every SkeresNative method the Scala sources call exists;
  * the Scala surface defines every name the reference's own specs, examples and unchanged core sources use of it
    (tests/golden/reference_jvm_api_names.txt)."""
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


def _scala_definitions():
    """{(kind, owner): set of member names} of every class / object / package object under bindings/scala (a brace-counting reader:
    good enough for these files, which keep one definition per line)."""
    defs = {}
    for dirpath, _, files in os.walk(os.path.join(ROOT, "bindings", "scala")):
        for f in files:
            if not f.endswith(".scala"):
                continue
            stack = []  # (kind, owner, depth at which its body opened)
            depth = 0
            pending = None
            for raw in open(os.path.join(dirpath, f)):
                line = raw.split("//")[0]
                m = re.search(r"\b(?:(?:sealed|abstract|final|case)\s+)*(class|object|trait)\s+([A-Za-z_][A-Za-z0-9_]*)", line)
                pk = re.search(r"\bpackage object\s+([A-Za-z_][A-Za-z0-9_]*)", line)
                opened = None
                if pk:
                    opened = ("package", pk.group(1))
                elif m:
                    opened = ("object" if m.group(1) == "object" else "class", m.group(2))
                if opened:
                    defs.setdefault(opened, set())
                    # constructor parameters declared `val` are members
                    for v in re.findall(r"\bval\s+([A-Za-z_][A-Za-z0-9_]*)\s*:", line.split("{")[0]):
                        defs[opened].add(v)
                    if stack and "case object" in line:  # a member of the enclosing object (NumericDiffMethodType.CENTRAL)
                        defs[(stack[-1][0], stack[-1][1])].add(opened[1])
                body = line
                if stack or opened:
                    owner = (opened if opened and "{" in line else (stack[-1][:2] if stack else None))
                    target = opened if opened else (stack[-1][:2] if stack else None)
                    if target:
                        for name in re.findall(r"\b(?:def|val|var|type)\s+([A-Za-z_][A-Za-z0-9_]*)", body.split("{", 1)[1] if opened and "{" in body else ("" if opened else body)):
                            defs[target].add(name)
                    if opened is None and stack:
                        mm = re.search(r"\b(?:val)\s+([A-Za-z_, ]+?)\s*=\s*Value", line)  # Enumeration members
                        if mm:
                            for name in mm.group(1).split(","):
                                defs[stack[-1][:2]].add(name.strip())
                    if opened and "Enumeration" in line:
                        for mm in re.finditer(r"\bval\s+([A-Za-z_, ]+?)\s*=\s*Value", line):
                            for name in mm.group(1).split(","):
                                defs[opened].add(name.strip())
                opens, closes = line.count("{"), line.count("}")
                if opened and opens == 0:
                    pending = opened  # a header that continues on the next line (`case class X(...)` / `extends Y {`)
                elif opened is None and pending and opens > closes:
                    opened, pending = pending, None
                if opened and opens > closes:
                    pending = None
                    stack.append((opened[0], opened[1], depth))
                depth += opens - closes
                while stack and depth <= stack[-1][2]:
                    stack.pop()
    return defs


def test_scala_surface_defines_every_name_the_references_own_callers_use():
    """tests/golden/reference_jvm_api_names.txt: the names the reference's specs, examples and unchanged core sources use of the
    surface that bindings/scala replaces.  NOT a compile (no scalac in this image): a name check that catches a missing
    factory or member — the round-3 verdict's finding (RichDoubleMatrix had no companion object)."""
    defs = _scala_definitions()
    missing = []
    for raw in open(os.path.join(ROOT, "tests", "golden", "reference_jvm_api_names.txt")):
        line = raw.split("#")[0].strip()
        if not line:
            continue
        head, members = line.split(":")
        kind, owner = head.split()
        key = (kind, owner)
        if key not in defs:
            missing.append("%s %s" % key)
            continue
        for name in members.split():
            if name not in defs[key]:
                missing.append("%s %s . %s" % (kind, owner, name))
    assert not missing, missing
