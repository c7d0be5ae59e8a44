"""CPU: libssc_hip.so loads without a GPU and exports every symbol include/ssc.h declares (no compute calls);
the ctypes table mirrors the header; argument validation fails loudly instead of touching memory."""
import ctypes as C
import os
import re

import pytest

from ssc_runtime import lib as L

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols(name="ssc.h"):
    text = open(os.path.join(ROOT, "include", name)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ssc_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = L.load()
    names = header_symbols()
    assert len(names) >= 35
    cdll = C.CDLL(L.LIB_PATH)
    for n in names:
        assert hasattr(cdll, n), f"{n} declared in include/ssc.h but not exported"
        assert n in L.SYMBOLS, f"{n} missing from the ctypes table"
    assert sorted(L.SYMBOLS) == names
    assert lib.ssc_version() == 4 and lib.ssc_arch() == b"gfx950"
    # diagnostics / profiling / tuning switches live in their own header, outside the product ABI
    dbg = header_symbols("ssc_debug.h")
    assert sorted(L.DEBUG_SYMBOLS) == dbg and not set(dbg) & set(names)
    for n in dbg:
        assert hasattr(cdll, n), f"{n} declared in include/ssc_debug.h but not exported"
    assert not [n for n in names if n.startswith("ssc_debug") or n.startswith("ssc_prof")]


def test_debug_switches_by_name():
    lib = L.load()
    v = C.c_int(-1)
    lib.ssc_debug_get(b"large_form", C.byref(v))
    assert v.value in (0, 1, 2, 3)
    prev = v.value
    lib.ssc_debug_set(b"large_form", 2)
    lib.ssc_debug_get(b"large_form", C.byref(v))
    assert v.value == 2
    lib.ssc_debug_set(b"large_form", prev)
    with pytest.raises(L.SscError, match="SSC_EINVAL"):
        lib.ssc_debug_set(b"no_such_switch", 1)


def test_struct_layouts_match_header_field_order():
    text = open(os.path.join(ROOT, "include", "ssc.h")).read()
    body = text[text.index("typedef struct {\n  float* emb;"):text.index("} ssc_params;")]
    fields = re.findall(r"(?:float\*|int)\s+(\w+);", body)
    assert fields == [f for f, _ in L.Params._fields_]
    body = text[text.index("typedef struct {\n  int V, E, H, A, F, Z;"):text.index("} ssc_model_cfg;")]
    names = [n.strip() for line in re.findall(r"(?:int|float)\s+([\w, ]+);", body) for n in line.split(",")]
    assert names == [f for f, _ in L.ModelCfg._fields_]


def test_bad_arguments_are_rejected_without_a_gpu():
    lib = L.load()
    with pytest.raises(L.SscError, match="SSC_EINVAL"):
        lib.ssc_gemm(None, None)
    d = L.GemmDesc()
    d.nseg = 9
    with pytest.raises(L.SscError, match="SSC_EINVAL"):
        lib.ssc_gemm(C.byref(d), None)
    with pytest.raises(L.SscError, match="SSC_EINVAL"):
        lib.ssc_feat_prep(None, 1, 1, 1, None, None, None)
    cfg = L.ModelCfg(10, 4, 4, 4, 4, 4, 0, 0, 0, 0.0, 1.0, 0, 1, 0)
    assert lib.ssc_train_workspace_bytes(C.byref(cfg), 2, 3, 4) > 0
    assert lib.ssc_train_workspace_bytes(C.byref(cfg), 0, 3, 4) == 0
    assert lib.ssc_gemm_auto_splits(64, 4800, 180) >= 1


def test_missing_extension_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(L, "_lib", None)
    monkeypatch.setattr(L, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(ImportError, match="no CPU fallback"):
        L.load()


def test_staged_load_hazard_checker_flags_copies_of_inflight_registers(tmp_path):
    """tools/check_staged_loads.py (the build gate of the GEMM kernels' inline-asm register prefetch): a copy of a
    load's destination before the wait is reported, the same code with the wait first is clean."""
    import subprocess
    import sys
    tool = os.path.join(ROOT, "tools", "check_staged_loads.py")
    bad = tmp_path / "bad.s"
    good = tmp_path / "good.s"
    body = "_ZN12_GLOBAL__N_111gemm_kernelILb1EEEvNS_5KArgsE:\n\tglobal_load_dwordx4 v[2:5], v[10:11], off\n{A}\tglobal_load_dwordx4 v[6:9], v[12:13], off\n{B}\ts_endpgm\n.end_amdhsa_kernel\n"
    bad.write_text(body.format(A="\tv_mov_b32_e32 v20, v3\n", B="\ts_waitcnt vmcnt(0)\n"))
    good.write_text(body.format(A="", B="\ts_waitcnt vmcnt(1)\n\tv_mov_b32_e32 v20, v3\n\ts_waitcnt vmcnt(0)\n\tv_mov_b32_e32 v21, v7\n"))
    assert subprocess.call([sys.executable, tool, str(bad)], stdout=subprocess.DEVNULL) == 1
    assert subprocess.call([sys.executable, tool, str(good)], stdout=subprocess.DEVNULL) == 0


def test_graft_entry_build_passes():
    """__graft_entry__.build() - what the driver runs on a CPU box every round: the library is current (or rebuilt), loads,
    reports the ABI version this header declares, and the host package imports."""
    import __graft_entry__ as ge
    ge.build()
