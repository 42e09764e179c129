"""The ONE chained co-jolt worker flow over the C ABI (`cozk_flow_*`): commit-all -> bytecode -> instruction lookups -> read-write
memory -> Spartan -> one reduce_and_prove (co-jolt/src/jolt/vm/jolt/worker.rs:175-266)."""
import ctypes

from . import _lib as L


class FlowConfig(ctypes.Structure):
    _fields_ = [("mode", ctypes.c_int), ("log_n", ctypes.c_int), ("log_m", ctypes.c_int), ("log_b", ctypes.c_int), ("log_mem", ctypes.c_int),
                ("n_mem", ctypes.c_int), ("n_subtables", ctypes.c_int), ("devices", ctypes.c_int * 3), ("seed", ctypes.c_uint64),
                ("precompute", ctypes.c_int), ("small_witness", ctypes.c_int)]


class FlowResult(ctypes.Structure):
    _fields_ = [("verified", ctypes.c_int)] + [(k, ctypes.c_double) for k in (
        "wall_ms", "t_commit_ms", "t_bytecode_ms", "t_primary_ms", "t_lookups_gp_ms", "t_rw_ms", "t_spartan_ms", "t_open_ms", "t_worker_ms",
        "t_spartan_build_ms")] + [(k, ctypes.c_uint64) for k in (
            "bytes_star_up", "bytes_star_down", "bytes_ring", "star_messages", "n_polys", "n_openings", "proof_len")] + [("proof_digest", ctypes.c_uint8 * 32)]


FLOW_SYMBOLS = ["cozk_flow_create", "cozk_flow_error", "cozk_flow_destroy", "cozk_flow_num_polys", "cozk_flow_ctx", "cozk_flow_prove", "cozk_flow_proof_bytes"]
_vp, _i, _sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t


def _decl():
    l = L.lib()
    l.cozk_flow_create.restype = _i
    l.cozk_flow_create.argtypes = [ctypes.POINTER(FlowConfig), ctypes.POINTER(_vp)]
    l.cozk_flow_error.restype = ctypes.c_char_p
    l.cozk_flow_error.argtypes = [_vp]
    l.cozk_flow_destroy.restype = _i
    l.cozk_flow_destroy.argtypes = [_vp]
    l.cozk_flow_num_polys.restype = _sz
    l.cozk_flow_num_polys.argtypes = [_vp]
    l.cozk_flow_ctx.restype = _vp
    l.cozk_flow_ctx.argtypes = [_vp, _i]
    l.cozk_flow_prove.restype = _i
    l.cozk_flow_prove.argtypes = [_vp, _i, ctypes.POINTER(FlowResult)]
    l.cozk_flow_proof_bytes.restype = _i
    l.cozk_flow_proof_bytes.argtypes = [_vp, _vp, _sz]
    return l


class FlowHarness:
    def __init__(self, mode="plain", log_n=4, log_m=3, log_b=3, log_mem=3, n_mem=6, n_subtables=3, devices=(0, 0, 0), seed=1, precompute=1, small_witness=0):
        self._l = _decl()
        cfg = FlowConfig()
        cfg.mode = L.MODE_PLAIN if mode == "plain" else L.MODE_REP3
        cfg.log_n, cfg.log_m, cfg.log_b, cfg.log_mem = log_n, log_m, log_b, log_mem
        cfg.n_mem, cfg.n_subtables = n_mem, n_subtables
        cfg.devices = (ctypes.c_int * 3)(*devices)
        cfg.seed = seed
        cfg.precompute = precompute
        cfg.small_witness = small_witness
        h = _vp()
        rc = self._l.cozk_flow_create(ctypes.byref(cfg), ctypes.byref(h))
        self.h = h
        if rc != L.OK:
            msg = (self._l.cozk_flow_error(h) or b"?").decode() if h else "?"
            if h:
                self._l.cozk_flow_destroy(h)
                self.h = None
            raise L.CozkError(rc, msg)

    def prove(self, verify=True):
        res = FlowResult()
        rc = self._l.cozk_flow_prove(self.h, 1 if verify else 0, ctypes.byref(res))
        if rc != L.OK:
            raise L.CozkError(rc, (self._l.cozk_flow_error(self.h) or b"?").decode())
        return res

    def proof_bytes(self, res):
        buf = (ctypes.c_uint8 * int(res.proof_len))()
        rc = self._l.cozk_flow_proof_bytes(self.h, buf, int(res.proof_len))
        if rc != L.OK:
            raise L.CozkError(rc, "proof_bytes")
        return bytes(buf)

    def num_polys(self):
        return int(self._l.cozk_flow_num_polys(self.h))

    def ctx_handle(self, party=0):
        return self._l.cozk_flow_ctx(self.h, party)

    def last_error(self):
        return (self._l.cozk_flow_error(self.h) or b"").decode()

    def close(self):
        if getattr(self, "h", None):
            self._l.cozk_flow_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
