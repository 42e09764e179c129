"""CPU test: libcozk.so loads and exports every symbol include/cozk.h declares (no compute calls --
there is no GPU here), and the product refuses to run without a device (no CPU fallback)."""
import ctypes
import importlib
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "cozk.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(cozk_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_exported(cozk):
    lib = cozk._lib.lib()
    names = _declared_symbols()
    assert len(names) > 60
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing
    # every symbol the python layer binds is declared in the header
    from importlib import import_module
    hp = import_module("co-zkvms_amd.harness")
    wk = import_module("co-zkvms_amd.workers")
    for n in list(cozk._lib.SIGNATURES) + hp.HARNESS_SYMBOLS + wk.WORKER_SYMBOLS + importlib.import_module("co-zkvms_amd.party_dist").PARTY_SYMBOLS + importlib.import_module("co-zkvms_amd.spartan").SPARTAN_SYMBOLS + importlib.import_module("co-zkvms_amd.lookups").LOOKUPS_SYMBOLS + importlib.import_module("co-zkvms_amd.outer").OUTER_SYMBOLS + importlib.import_module("co-zkvms_amd.flow").FLOW_SYMBOLS + importlib.import_module("co-zkvms_amd.logup").LOGUP_SYMBOLS:
        assert n in names, n


def test_no_cpu_fallback(cozk):
    n = ctypes.c_int(-1)
    rc = cozk._lib.lib().cozk_device_count(ctypes.byref(n))
    if rc == 0 and n.value > 0:
        pytest.skip("a GPU is visible here")
    with pytest.raises(cozk.CozkError):
        cozk.Context(0)
    with pytest.raises(cozk.CozkError):
        cozk.Harness(mode="plain", log_n=4, n_fr=1, n_u16=0, n_u32=0, n_flags=0, gp_batch=1)


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "co-zkvms_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".inc")):
                txt = open(os.path.join(dp, f), errors="ignore").read()
                assert "pyref" not in txt.replace("oracle/pyref.py", "") or f.endswith((".hip", ".hpp")), f
                assert "import pyharness" not in txt and "coracle" not in txt and "liboracle" not in txt, f
