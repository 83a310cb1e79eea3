"""CPU suite: the C-ABI library loads and exports every symbol include/al3d.h declares
(no compute calls -- there is no GPU here), and the ctypes table matches the header."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "al3d.h")


def declared_symbols():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(al3d_[a-z0-9_]+)\s*\(", src)))


def test_header_declares_entry_points():
    syms = declared_symbols()
    assert "al3d_greedy_kcenter_f64" in syms and "al3d_l1_distance_f32" in syms
    assert len(syms) >= 12


def test_library_exports_every_declared_symbol():
    from al3d import lib
    assert os.path.exists(lib.LIB_PATH), "run `python __graft_entry__.py` (build()) first"
    so = ctypes.CDLL(lib.LIB_PATH)
    missing = [s for s in declared_symbols() if not hasattr(so, s)]
    assert not missing, f"libal3d_hip.so lacks {missing}"
    so.al3d_abi_version.restype = ctypes.c_int
    assert so.al3d_abi_version() >= 1


def test_ctypes_table_matches_header():
    from al3d import lib
    assert sorted(lib.SIGNATURES) == declared_symbols()
    lib.load()


def test_bad_arguments_fail_loudly_without_gpu():
    """Argument validation happens before any launch, so it is checkable on CPU."""
    from al3d import lib
    so = lib.load()
    rc = so.al3d_l1_distance_f32(None, 4, 4, 2, None, None)
    assert rc == -1
    assert b"null pointer" in so.al3d_last_error()
    rc = so.al3d_knn_2d_f64(ctypes.c_void_p(8), 4, 99, ctypes.c_void_p(8), ctypes.c_void_p(8), None)
    assert rc == -1 and b"kq" in so.al3d_last_error()
    with pytest.raises(lib.Al3dError):
        lib.call("al3d_combine_maps_f64", None, None, None, 4, 1, 0, 1.0, 1.0, 1.0, 1.0, None, None)


def test_product_never_imports_oracle():
    """The product package must not reference oracle/ (no CPU fallback)."""
    pkg = os.path.join(ROOT, "exploring-diversity-based-active-learning-for-3d-object-detection-"
                             "in-autonomous-driving_amd")
    bad = []
    for dp, _, fns in os.walk(pkg):
        for fn in fns:
            if fn.endswith((".py", ".hip", ".h", ".cpp")) and fn != "al3d_exp_table.h":
                txt = open(os.path.join(dp, fn), errors="replace").read()
                if re.search(r"^\s*(from|import)\s+oracle\b|libal3d_oracle|oracle/", txt, flags=re.M):
                    bad.append(os.path.join(dp, fn))
    assert not bad, bad
