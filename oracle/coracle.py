"""ORACLE (test infrastructure): ctypes loader of the plain-C CPU restatement (oracle/c/).
Imported only by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg."""
import ctypes
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "_build", "liboracle.so")


def build(force=False):
    srcs = [os.path.join(HERE, "c", "oracle.c"), os.path.join(HERE, "c", "bn254.h")]
    if force or not os.path.exists(LIB) or any(os.path.getmtime(s) > os.path.getmtime(LIB) for s in srcs):
        subprocess.check_call(["make", "-C", HERE, "-s"] + (["-B"] if force else []))
    return LIB


class OrcConfig(ctypes.Structure):
    _fields_ = [("mode", ctypes.c_int), ("log_n", ctypes.c_int), ("n_fr", ctypes.c_int), ("n_u16", ctypes.c_int),
                ("n_u32", ctypes.c_int), ("n_flags", ctypes.c_int), ("n_small", ctypes.c_int), ("gp_batch", ctypes.c_int),
                ("gp_log_leaves", ctypes.c_int), ("seed", ctypes.c_uint64)]


class OrcResult(ctypes.Structure):
    _fields_ = [("t_setup_s", ctypes.c_double), ("t_commit_s", ctypes.c_double), ("t_gp_construct_s", ctypes.c_double),
                ("t_gp_prove_s", ctypes.c_double), ("t_eval_s", ctypes.c_double), ("t_open_s", ctypes.c_double),
                ("t_total_s", ctypes.c_double), ("digest", ctypes.c_uint8 * 32), ("proof_len", ctypes.c_uint64),
                ("threads", ctypes.c_int)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            build()
        _lib = ctypes.CDLL(LIB)
        _lib.orc_pipeline.restype = ctypes.c_int
        _lib.orc_pipeline.argtypes = [ctypes.POINTER(OrcConfig), ctypes.POINTER(OrcResult), ctypes.c_void_p, ctypes.c_size_t]
        _lib.orc_num_threads.restype = ctypes.c_int
    return _lib


def pipeline(cfg, want_proof=True):
    """cfg: dict like oracle/pyharness.run -> (OrcResult, proof bytes or None)"""
    c = OrcConfig()
    c.mode = 1 if cfg["mode"] == "plain" else 2
    for k in ("log_n", "n_fr", "n_u16", "n_u32", "n_flags", "n_small", "gp_batch", "gp_log_leaves", "seed"):
        setattr(c, k, cfg[k])
    res = OrcResult()
    cap = 1 << 22
    buf = (ctypes.c_uint8 * cap)() if want_proof else None
    rc = lib().orc_pipeline(ctypes.byref(c), ctypes.byref(res), buf, cap if want_proof else 0)
    if rc != 0:
        raise RuntimeError("orc_pipeline failed")
    return res, (bytes(buf[:res.proof_len]) if want_proof else None)


def fp_binop(base_field, op, a, b):
    import numpy as np
    a = np.ascontiguousarray(a, dtype=np.uint64)
    b = np.ascontiguousarray(b, dtype=np.uint64)
    out = np.empty_like(a)
    lib().orc_fp_binop(ctypes.c_int(base_field), ctypes.c_int(op), a.ctypes.data_as(ctypes.c_void_p),
                       b.ctypes.data_as(ctypes.c_void_p), out.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(a.shape[0]))
    return out


def fq_mul_fast(a, b):
    """the CPU baseline's unrolled Fq product (oracle/c/bn254.h fq_mul_fast) on (n, 4) Montgomery limb arrays"""
    import numpy as np
    a = np.ascontiguousarray(a, dtype=np.uint64)
    b = np.ascontiguousarray(b, dtype=np.uint64)
    out = np.empty_like(a)
    lib().orc_fq_mul_fast(a.ctypes.data_as(ctypes.c_void_p), b.ctypes.data_as(ctypes.c_void_p), out.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(a.shape[0]))
    return out


def msm(xy, inf, scalars_mont):
    import numpy as np
    xy = np.ascontiguousarray(xy, dtype=np.uint64)
    inf = np.ascontiguousarray(inf, dtype=np.uint8)
    sc = np.ascontiguousarray(scalars_mont, dtype=np.uint64)
    out = np.zeros(8, dtype=np.uint64)
    oi = ctypes.c_int()
    lib().orc_msm(xy.ctypes.data_as(ctypes.c_void_p), inf.ctypes.data_as(ctypes.c_void_p), sc.ctypes.data_as(ctypes.c_void_p),
                  ctypes.c_size_t(xy.shape[0]), out.ctypes.data_as(ctypes.c_void_p), ctypes.byref(oi))
    return out, oi.value


def prf_fr(key, counter, n):
    """PRF(key, counter + i) for i < n as Montgomery limbs (n, 4) -- the C restatement of csrc/prf.hip.hpp"""
    import numpy as np
    out = np.zeros((n, 4), dtype=np.uint64)
    lib().orc_prf_fr(ctypes.c_char_p(bytes(key)), ctypes.c_uint64(counter), ctypes.c_size_t(n), out.ctypes.data_as(ctypes.c_void_p))
    return out
