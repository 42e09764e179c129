"""profiling hooks on a harness party's context (bench.py roofline)"""
import ctypes

from . import _lib as L


def prof_enable(harness, party=0, on=True):
    l = L.lib()
    rc = l.cozk_prof_enable(harness.party_ctx_handle(party), 1 if on else 0)
    if rc != L.OK:
        raise L.CozkError(rc, "prof_enable")


def prof_read(harness, party=0):
    l = L.lib()
    n = ctypes.c_uint64()
    ms = ctypes.c_double()
    adds = ctypes.c_uint64()
    nb = ctypes.c_uint64()
    rc = l.cozk_prof_read(harness.party_ctx_handle(party), ctypes.byref(n), ctypes.byref(ms), ctypes.byref(adds), ctypes.byref(nb))
    if rc != L.OK:
        raise L.CozkError(rc, "prof_read")
    return dict(launches=n.value, total_ms=ms.value, point_adds=adds.value, alg_bytes=nb.value)


def prof_read_kernels(harness, party=0):
    """per-kernel HIP-event totals of the HBM-bound kernels (cozk_prof_read_kernel): name -> launches, total_ms, alg_bytes"""
    l = L.lib()
    out = {}
    slot = 0
    while True:
        name = l.cozk_prof_kernel_name(slot)
        if not name:
            return out
        n, ms, nb = ctypes.c_uint64(), ctypes.c_double(), ctypes.c_uint64()
        rc = l.cozk_prof_read_kernel(harness.party_ctx_handle(party), slot, ctypes.byref(n), ctypes.byref(ms), ctypes.byref(nb))
        if rc != L.OK:
            raise L.CozkError(rc, "prof_read_kernel")
        out[name.decode()] = dict(launches=n.value, total_ms=ms.value, alg_bytes=nb.value)
        slot += 1
