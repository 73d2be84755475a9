"""CPU: the C-ABI library loads and exports every symbol include/x265_amd_me.h
declares; struct layouts on the Python side match the header. No compute calls."""
import ctypes as C
import os
import re

import numpy as np

from x265_amd import engine as E

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    text = open(os.path.join(ROOT, "include", "x265_amd_me.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(x265amd_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = E.load_library()
    names = declared_functions()
    assert len(names) >= 15
    for n in names:
        assert hasattr(lib, n), "missing export %s" % n


def test_struct_layouts():
    assert E.JOB_DTYPE.itemsize == 40
    assert E.JOB_DTYPE.fields["merange"][1] == 36
    assert E.JOB_DTYPE.fields["mvc"][1] == 20
    assert E.RESULT_DTYPE.itemsize == 16
    assert C.sizeof(E.Plane) == 32
    assert C.sizeof(E.Field) == 5 * 32
    assert C.sizeof(E.Primitives) == (10 * 25 + 1) * 8


def test_partition_table():
    lib = E.load_library()
    for i, (w, h) in enumerate(E.PARTITIONS):
        assert lib.x265amd_partition_from_sizes(w, h) == i
    assert lib.x265amd_partition_from_sizes(20, 20) == -1
    assert lib.x265amd_partition_from_sizes(64, 8) == -1


def test_create_rejects_bad_arguments():
    lib = E.load_library()
    ctx = C.c_void_p()
    assert lib.x265amd_create(0, 9, C.byref(ctx)) == -1      # unsupported bit depth
    assert lib.x265amd_create(0, 8, None) == -1


def test_header_is_plain_c(tmp_path):
    """include/x265_amd_me.h compiles as C99 and as C++ on its own, and its struct sizes match the Python mirrors."""
    import subprocess
    src = tmp_path / "t.c"
    src.write_text('#include "x265_amd_me.h"\n#include <stdio.h>\nint main(void) { printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\\n", '
                   'sizeof(x265amd_plane), sizeof(x265amd_job), sizeof(x265amd_result), sizeof(x265amd_field), '
                   'sizeof(x265amd_frame_cost), sizeof(x265amd_weight), sizeof(x265amd_bipred_job), sizeof(x265amd_rk_result), sizeof(x265amd_mc_job), sizeof(x265amd_picture)); return 0; }\n')
    exe = tmp_path / "t"
    inc = os.path.join(ROOT, "include")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", "-I", inc, "-o", str(exe), str(src)])
    sizes = [int(v) for v in subprocess.check_output([str(exe)]).split()]
    assert sizes == [C.sizeof(E.Plane), E.JOB_DTYPE.itemsize, E.RESULT_DTYPE.itemsize, C.sizeof(E.Field), C.sizeof(E.FrameCost),
                     C.sizeof(E.Weight), E.BIPRED_JOB_DTYPE.itemsize, E.RK_RESULT_DTYPE.itemsize, E.MC_JOB_DTYPE.itemsize,
                     C.sizeof(E.Picture)]
    subprocess.check_call(["g++", "-std=c++11", "-Wall", "-Werror", "-fsyntax-only", "-x", "c++", "-I", inc, str(src)])
