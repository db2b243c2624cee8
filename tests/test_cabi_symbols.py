"""The C-ABI library loads on a machine without a GPU and exports every
function that include/orb_hip.h declares (no compute calls here); the pure
host function orb_hamming agrees with the oracle; the drop-in C++ classes
compile against the header."""
import ctypes as C
import os
import re
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    txt = open(os.path.join(ROOT, "include", "orb_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(orb_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    from orb_slam2_amd import binding

    lib = binding.lib()
    names = declared_functions()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), "liborbhip.so does not export %s" % n
    assert set(binding.EXPORTS) == set(names)
    assert b"gfx950" in lib.orb_version()


def test_host_hamming_matches_oracle(oracle):
    from orb_slam2_amd import binding
    from orb_slam2_amd.synth import synth_descriptors

    d = synth_descriptors(64, 3)
    for i in range(63):
        assert binding.hamming(d[i], d[i + 1]) == oracle.hamming(d[i], d[i + 1], bithack=True)


def test_invalid_arguments_fail_without_a_gpu_call():
    from orb_slam2_amd import binding

    lib = binding.lib()
    assert lib.orb_create(None, None) == -1  # ORB_ERR_INVALID
    assert b"null" in lib.orb_last_error()
    n = C.c_int(7)
    assert lib.orb_extract(None, None, 0, 0, 0, None, None, 0, C.byref(n)) == 0 and n.value == 0  # empty image


def test_dropin_cpp_compiles():
    src = [os.path.join(ROOT, "orb_slam2_amd", "cpp", f) for f in ("ORBextractor.cc", "Frame.cc", "ORBmatcher.cc")]
    src.append(os.path.join(ROOT, "examples", "orb_demo.cc"))
    subprocess.check_call(["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-Werror"] + src)


def test_header_is_plain_c(tmp_path):
    """The boundary is a C ABI: the header must be usable from a C translation unit."""
    src = tmp_path / "use.c"
    src.write_text('#include "orb_hip.h"\nint use(orb_ctx* c) { int n = 0; return orb_quadtree_second_pass_levels(c, &n) + n; }\n')
    subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-fsyntax-only", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(src)])
