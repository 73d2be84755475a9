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
