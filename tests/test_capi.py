"""CPU: the C-ABI library loads and exports every symbol include/bslv_hip.h declares (no compute)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    txt = open(os.path.join(ROOT, "include", "bslv_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(bslv_[a-z0-9_]+)\s*\(", txt)))


def compat_symbols():
    txt = open(os.path.join(ROOT, "include", "bslv_lp_compat.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(lp_[a-z_]+)\s*\(", txt)))


def test_reference_lp_symbols_exported():
    """the 17 lp_* symbols bslv_algs.o / bslv_main.o import from bslv_lp.c + lp_get_time"""
    from bensolve_amd import load_library
    lib = load_library()
    syms = compat_symbols()
    need = ["lp_init", "lp_update_extra_coeffs", "lp_set_mat_row", "lp_clear_obj_coeffs", "lp_set_obj_coeffs", "lp_set_rows",
            "lp_set_rows_hom", "lp_set_cols", "lp_set_cols_hom", "lp_set_options", "lp_solve", "lp_obj_val",
            "lp_primal_solution_cols", "lp_dual_solution_rows", "lp_dual_solution_cols", "lp_get_num", "lp_free"]
    assert set(need) <= set(syms)
    assert not [s for s in syms if not hasattr(lib, s)]


def test_header_symbols_exported():
    from bensolve_amd import load_library
    lib = load_library()
    syms = declared_symbols()
    assert len(syms) > 40
    missing = [s for s in syms if not hasattr(lib, s)]
    assert not missing, missing


def test_no_device_fails_loudly():
    import torch
    from bensolve_amd import load_library
    from bensolve_amd._lib import BslvError
    import numpy as np
    from bensolve_amd.lp import LpEngine
    lib = load_library()
    if torch.cuda.is_available():
        return
    assert lib.bslv_device_count() == 0
    try:
        LpEngine(1, 1, np.ones((1, 1)), np.zeros(2), np.ones(2), np.zeros(2), 0, 0, 2)
    except BslvError as e:
        assert "device" in str(e).lower()
    else:
        raise AssertionError("engine construction must fail without a GPU: there is no CPU fallback")


def test_product_does_not_import_oracle():
    # the product path must not route through oracle/
    for dirpath, _, files in os.walk(os.path.join(ROOT, "bensolve_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".c", ".cpp")) or f == "Makefile":
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                for line in txt.splitlines():
                    code = line.split("//")[0].split("#")[0] if not f.endswith(".py") else line.split("#")[0]
                    assert "oracle_api" not in code and "liboracle" not in code and "oracle/" not in code.replace("oracle/lp_dense.c", "").replace("oracle/poly_dd.c", "").replace("oracle/benson_cpu.c", ""), (f, line)
