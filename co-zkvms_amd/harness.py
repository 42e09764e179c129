"""In-process prove harness over the C ABI (`cozk_harness_*`): the counterpart of the reference's
runner (co-jolt/examples/rep3_jolt.rs) restricted to the hot path.  One worker thread per party
inside libcozk, the coordinator on the calling thread; setup (SRS + witness resident in HBM) is
separate from the timed `prove()` step."""
import ctypes

from . import _lib as L


class HarnessConfig(ctypes.Structure):
    _fields_ = [("mode", ctypes.c_int), ("log_n", ctypes.c_int), ("n_fr", ctypes.c_int), ("n_u16", ctypes.c_int),
                ("n_u32", ctypes.c_int), ("n_flags", ctypes.c_int), ("n_small", ctypes.c_int), ("gp_batch", ctypes.c_int),
                ("gp_log_leaves", ctypes.c_int), ("precompute", ctypes.c_int), ("devices", ctypes.c_int * 3),
                ("seed", ctypes.c_uint64), ("log_workers", ctypes.c_int), ("worker_devices", ctypes.c_int * 8),
                ("leaf_fingerprints", ctypes.c_int)]


class HarnessResult(ctypes.Structure):
    _fields_ = [("verified", ctypes.c_int), ("wall_ms", ctypes.c_double), ("t_commit_ms", ctypes.c_double),
                ("t_gp_construct_ms", ctypes.c_double), ("t_gp_prove_ms", ctypes.c_double), ("t_eval_ms", ctypes.c_double),
                ("t_open_ms", ctypes.c_double), ("t_worker_ms", ctypes.c_double), ("bytes_star_up", ctypes.c_uint64),
                ("bytes_star_down", ctypes.c_uint64), ("bytes_ring", ctypes.c_uint64), ("star_messages", ctypes.c_uint64),
                ("proof_len", ctypes.c_uint64), ("proof_digest", ctypes.c_uint8 * 32), ("t_hub_wait_ms", ctypes.c_double),
                ("hub_exchanges", ctypes.c_uint64)]


_declared = False


def _decl():
    global _declared
    l = L.lib()
    if not _declared:
        l.cozk_harness_create.restype = ctypes.c_int
        l.cozk_harness_create.argtypes = [ctypes.POINTER(HarnessConfig), ctypes.POINTER(ctypes.c_void_p)]
        l.cozk_harness_error.restype = ctypes.c_char_p
        l.cozk_harness_error.argtypes = [ctypes.c_void_p]
        l.cozk_harness_destroy.restype = ctypes.c_int
        l.cozk_harness_destroy.argtypes = [ctypes.c_void_p]
        l.cozk_harness_prove.restype = ctypes.c_int
        l.cozk_harness_prove.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(HarnessResult)]
        l.cozk_harness_proof_bytes.restype = ctypes.c_int
        l.cozk_harness_proof_bytes.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
        l.cozk_harness_ctx.restype = ctypes.c_void_p
        l.cozk_harness_ctx.argtypes = [ctypes.c_void_p, ctypes.c_int]
        _declared = True
    return l


# symbols of include/cozk.h declared here rather than in _lib.SIGNATURES
HARNESS_SYMBOLS = ["cozk_harness_create", "cozk_harness_error", "cozk_harness_destroy", "cozk_harness_prove",
                   "cozk_harness_proof_bytes", "cozk_harness_ctx"]


class Harness:
    def __init__(self, mode="plain", log_n=10, n_fr=4, n_u16=1, n_u32=1, n_flags=1, n_small=0, gp_batch=2,
                 gp_log_leaves=None, precompute=True, devices=(0, 0, 0), seed=1, log_workers=0, worker_devices=None,
                 leaf_fingerprints=False):
        self._l = _decl()
        cfg = HarnessConfig()
        cfg.mode = L.MODE_PLAIN if mode == "plain" else L.MODE_REP3
        cfg.log_n = log_n
        cfg.n_fr, cfg.n_u16, cfg.n_u32, cfg.n_flags, cfg.n_small = n_fr, n_u16, n_u32, n_flags, n_small
        cfg.gp_batch = gp_batch
        cfg.gp_log_leaves = gp_log_leaves if gp_log_leaves is not None else log_n + 1
        cfg.precompute = 1 if precompute else 0
        cfg.devices = (ctypes.c_int * 3)(*devices)
        cfg.seed = seed
        cfg.log_workers = log_workers
        wd = list(worker_devices) if worker_devices is not None else [devices[0]] * 8
        cfg.worker_devices = (ctypes.c_int * 8)(*(wd + [wd[-1]] * (8 - len(wd))))
        cfg.leaf_fingerprints = 1 if leaf_fingerprints else 0
        self.cfg = cfg
        h = ctypes.c_void_p()
        rc = self._l.cozk_harness_create(ctypes.byref(cfg), ctypes.byref(h))
        self.h = h
        if rc != L.OK:
            msg = self._l.cozk_harness_error(h) if h else b"?"
            self.close()
            raise L.CozkError(rc, (msg or b"?").decode())

    def prove(self, verify=True):
        res = HarnessResult()
        rc = self._l.cozk_harness_prove(self.h, 1 if verify else 0, ctypes.byref(res))
        if rc != L.OK:
            raise L.CozkError(rc, (self._l.cozk_harness_error(self.h) or b"?").decode())
        return res

    def last_error(self):
        return (self._l.cozk_harness_error(self.h) or b"").decode()

    def proof_bytes(self, res):
        buf = (ctypes.c_uint8 * res.proof_len)()
        rc = self._l.cozk_harness_proof_bytes(self.h, buf, res.proof_len)
        if rc != L.OK:
            raise L.CozkError(rc, "proof_bytes")
        return bytes(buf)

    def party_ctx_handle(self, party=0):
        return self._l.cozk_harness_ctx(self.h, party)

    def close(self):
        if getattr(self, "h", None):
            self._l.cozk_harness_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
