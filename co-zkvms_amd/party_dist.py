"""One Rep3 party per process / GPU (BASELINE config 3): torch.distributed transports for the engine's
network seam.
  * TorchHub  -> cozk_hub_net.all_gather: variable-length byte all-gather over a gloo (CPU) group -- star
                 messages are a few hundred bytes, latency-bound; the replicated coordinator inside libcozk
                 turns the gathered messages into the next challenge on every rank.
  * TorchRing -> cozk_ring_net.reshare: send to the next party / receive from the previous one.  With the
                 nccl backend (= RCCL) the payload stays on the GPUs (xGMI); with gloo it is staged through
                 host memory (rehearsal with several ranks on one GPU).
No field arithmetic here: this is plumbing, covered by gloo tests on CPU."""
import ctypes

import torch
import torch.distributed as dist

from . import _lib as L
from .harness import HarnessConfig, HarnessResult, _decl

_AG = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t,
                       ctypes.POINTER(ctypes.c_size_t))
_RS = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t)


class HubNet(ctypes.Structure):
    _fields_ = [("user", ctypes.c_void_p), ("n_participants", ctypes.c_int), ("my_index", ctypes.c_int), ("all_gather", _AG)]


class RingNetC(ctypes.Structure):
    _fields_ = [("user", ctypes.c_void_p), ("reshare", _RS), ("stream_ordered", ctypes.c_int)]


PARTY_SYMBOLS = ["cozk_ring_unique_id", "cozk_ring_init", "cozk_ring_destroy", "cozk_ring_net_native", "cozk_shm_hub_open", "cozk_shm_hub_net", "cozk_shm_hub_unlink", "cozk_shm_hub_set_timeout_ms", "cozk_shm_hub_abort", "cozk_shm_hub_close",
                 "cozk_harness_create_party", "cozk_harness_create_participant", "cozk_harness_prove_distributed", "cozk_copy"]


_FAST = 2040  # payload bytes that travel with the length header in the single-collective fast path


def all_gather_bytes(group, world, payload):
    """variable-length all-gather of byte strings over `group` (CPU tensors).  Round messages are a few
    hundred bytes: they ride one all_gather of fixed 2 KiB slots ([u64 length | payload]); only if some
    participant's message is longer does a second, max-length all_gather follow (every rank sees all the
    lengths after the first one, so they agree on it without extra traffic)."""
    n = len(payload)
    slot = bytearray(8 + _FAST)
    slot[:8] = n.to_bytes(8, "little")
    if n <= _FAST:
        slot[8:8 + n] = payload
    buf = torch.frombuffer(slot, dtype=torch.uint8)
    outs = [torch.empty(8 + _FAST, dtype=torch.uint8) for _ in range(world)]
    dist.all_gather(outs, buf, group=group)
    raw = [o.numpy().tobytes() for o in outs]
    lens = [int.from_bytes(r[:8], "little") for r in raw]
    if max(lens) <= _FAST:
        return [r[8:8 + l] for r, l in zip(raw, lens)]
    m = max(lens)
    big = bytearray(m)
    big[:n] = payload
    outs2 = [torch.empty(m, dtype=torch.uint8) for _ in range(world)]
    dist.all_gather(outs2, torch.frombuffer(big, dtype=torch.uint8), group=group)
    return [o.numpy().tobytes()[:l] for o, l in zip(outs2, lens)]


class TorchHub:
    def __init__(self, rank, world, group=None):
        self.rank, self.world, self.group = rank, world, group
        self.error = None

        def _ag(_u, send, n, recv, cap, lens):
            try:
                mine = ctypes.string_at(send, n) if n else b""
                parts = all_gather_bytes(self.group, self.world, mine)
                for i, p in enumerate(parts):
                    if len(p) > cap:
                        return 2
                    ctypes.memmove(recv + i * cap, p, len(p))
                    lens[i] = len(p)
                return 0
            except Exception as e:  # never unwind into C
                self.error = e
                return 1

        self._cb = _AG(_ag)
        self.net = HubNet(None, world, rank, self._cb)


class ShmHub:
    """cozk_shm_hub: the native shared-memory all-gather of libcozk (csrc/shm_hub.hip) for one-process-per-GPU
    runs on ONE node.  `group` (a torch.distributed group, any backend) is used only to agree on the segment
    name and to order create -> attach -> unlink; the per-round exchanges never touch it."""
    SHM_SYMBOLS = ["cozk_shm_hub_open", "cozk_shm_hub_net", "cozk_shm_hub_unlink", "cozk_shm_hub_set_timeout_ms", "cozk_shm_hub_abort",
                   "cozk_shm_hub_close"]

    def __init__(self, rank, world, group=None, name=None, slot_bytes=1 << 18, timeout_ms=120000, create=None):
        import os
        import secrets
        self._l = L.lib()
        l = self._l
        l.cozk_shm_hub_open.restype = ctypes.c_int
        l.cozk_shm_hub_open.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_size_t, ctypes.POINTER(ctypes.c_void_p)]
        l.cozk_shm_hub_net.restype = ctypes.c_int
        l.cozk_shm_hub_net.argtypes = [ctypes.c_void_p, ctypes.POINTER(HubNet)]
        l.cozk_shm_hub_unlink.restype = ctypes.c_int
        l.cozk_shm_hub_unlink.argtypes = [ctypes.c_void_p]
        l.cozk_shm_hub_set_timeout_ms.restype = ctypes.c_int
        l.cozk_shm_hub_set_timeout_ms.argtypes = [ctypes.c_void_p, ctypes.c_uint64]
        l.cozk_shm_hub_abort.restype = None
        l.cozk_shm_hub_abort.argtypes = [ctypes.c_void_p]
        l.cozk_shm_hub_close.restype = None
        l.cozk_shm_hub_close.argtypes = [ctypes.c_void_p]
        self.rank, self.world, self.error = rank, world, None
        self.h = ctypes.c_void_p()
        use_dist = name is None
        if use_dist:  # rank 0 picks a fresh name, everyone learns it
            box = ["/cozk_hub_%d_%s" % (os.getpid(), secrets.token_hex(6))] if rank == 0 else [None]
            dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
            name = box[0]
        self.name = name
        rc = L.OK
        if create is None:
            create = rank == 0
        if create:
            rc = l.cozk_shm_hub_open(name.encode(), 1, world, rank, slot_bytes, ctypes.byref(self.h))
        if use_dist:
            dist.barrier(group=group)
        if not create:
            rc = l.cozk_shm_hub_open(name.encode(), 0, world, rank, slot_bytes, ctypes.byref(self.h))
        if use_dist:
            dist.barrier(group=group)
            if rank == 0 and self.h:
                l.cozk_shm_hub_unlink(self.h)
        if rc != L.OK:
            raise L.CozkError(rc, "cozk_shm_hub_open(%s) failed" % name)
        l.cozk_shm_hub_set_timeout_ms(self.h, timeout_ms)
        self.net = HubNet()
        l.cozk_shm_hub_net(self.h, ctypes.byref(self.net))

    def all_gather(self, payload, cap=1 << 18):
        """byte all-gather through the segment (tests / host code that wants the same transport)"""
        recv = (ctypes.c_uint8 * (cap * self.world))()
        lens = (ctypes.c_size_t * self.world)()
        buf = (ctypes.c_uint8 * max(1, len(payload))).from_buffer_copy(payload or b"\0")
        rc = self.net.all_gather(self.net.user, ctypes.cast(buf, ctypes.c_void_p), len(payload), ctypes.cast(recv, ctypes.c_void_p), cap, lens)
        if rc != 0:
            raise RuntimeError("shm hub all_gather failed (%d)" % rc)
        raw = bytes(recv)
        return [raw[i * cap:i * cap + lens[i]] for i in range(self.world)]

    def unlink(self):
        self._l.cozk_shm_hub_unlink(self.h)

    def abort(self):
        if self.h:
            self._l.cozk_shm_hub_abort(self.h)

    def close(self):
        if getattr(self, "h", None):
            self._l.cozk_shm_hub_close(self.h)
            self.h = ctypes.c_void_p()


class TorchRing:
    """reshare over torch.distributed P2P.  device=None: host staging (gloo); else CUDA staging tensors (nccl/RCCL)"""

    def __init__(self, ctx_handle, rank, world, group=None, device=None, next_rank=None, prev_rank=None):
        self.ctx, self.rank, self.world, self.group, self.device = ctx_handle, rank, world, group, device
        # ranks (in `group`) of the next / previous party on this ring; default: the whole group is one ring
        self.next_rank = (rank + 1) % world if next_rank is None else next_rank
        self.prev_rank = (rank + world - 1) % world if prev_rank is None else prev_rank
        self.error = None
        self._l = L.lib()
        self._l.cozk_copy.restype = ctypes.c_int
        self._l.cozk_copy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]

        def _rs(_u, dev_send, dev_recv, nbytes):
            try:
                if self.device is not None:
                    torch.cuda.set_device(self.device)  # callbacks run on the engine's worker thread
                dev = torch.device("cuda", self.device) if self.device is not None else torch.device("cpu")
                sbuf = torch.empty(nbytes, dtype=torch.uint8, device=dev)
                rbuf = torch.empty(nbytes, dtype=torch.uint8, device=dev)
                if self._l.cozk_copy(self.ctx, sbuf.data_ptr(), dev_send, nbytes) != 0:
                    return 3
                nxt, prv = self.next_rank, self.prev_rank
                ops = [dist.P2POp(dist.isend, sbuf, nxt, group=self.group), dist.P2POp(dist.irecv, rbuf, prv, group=self.group)]
                for req in dist.batch_isend_irecv(ops):
                    req.wait()
                if self.device is not None:
                    torch.cuda.synchronize(self.device)
                if self._l.cozk_copy(self.ctx, dev_recv, rbuf.data_ptr(), nbytes) != 0:
                    return 3
                return 0
            except Exception as e:
                self.error = e
                return 1

        self._cb = _RS(_rs)
        self.net = RingNetC(None, self._cb, 0)


class NativeRing:
    """the Rep3 ring inside libcozk (csrc/ring.hip): ncclSend / ncclRecv on the party context's stream, GPU to GPU over
    xGMI, nothing staged and nothing waited for on the host.  `ring_group`: the torch.distributed group (any backend; only
    used to hand the 128-byte id from ring rank 0 to the others) whose ranks form THIS ring, in ring order."""

    def __init__(self, ctx_handle, ring_rank, ring_size=3, ring_group=None):
        self._l = L.lib()
        self.ctx = ctx_handle
        self.error = None
        box = [None]
        if ring_rank == 0:
            buf = (ctypes.c_uint8 * 128)()
            rc = self._l.cozk_ring_unique_id(buf)
            if rc != L.OK:
                raise L.CozkError(rc, "cozk_ring_unique_id failed (librccl unavailable?)")
            box = [bytes(buf)]
        if ring_size > 1:
            src = dist.get_global_rank(ring_group, 0) if ring_group is not None else 0
            dist.broadcast_object_list(box, src=src, group=ring_group)
        rc = self._l.cozk_ring_init(ctx_handle, box[0], ring_rank, ring_size)
        if rc != L.OK:
            raise L.CozkError(rc, (self._l.cozk_last_error(ctx_handle) or b"cozk_ring_init failed").decode())
        self.net = RingNetC()
        rc = self._l.cozk_ring_net_native(ctx_handle, ctypes.byref(self.net))
        if rc != L.OK:
            raise L.CozkError(rc, "cozk_ring_net_native")

    def close(self):
        if self.ctx:
            self._l.cozk_ring_destroy(self.ctx)
            self.ctx = None


class DistributedParty:
    """this process's participant -- (party, worker) -- of one distributed proof
    (`cozk_harness_create_participant` / `cozk_harness_prove_distributed`): a party of a 3-party Rep3 run,
    and/or one worker sub-net (high-variable chunk) of the plain / Rep3 prover"""

    def __init__(self, party, device=0, worker=0, mode="rep3", log_workers=0, **cfgkw):
        self._l = _decl()
        self._l.cozk_harness_create_participant.restype = ctypes.c_int
        self._l.cozk_harness_create_participant.argtypes = [ctypes.POINTER(HarnessConfig), ctypes.c_int, ctypes.c_int,
                                                            ctypes.POINTER(ctypes.c_void_p)]
        self._l.cozk_harness_prove_distributed.restype = ctypes.c_int
        self._l.cozk_harness_prove_distributed.argtypes = [ctypes.c_void_p, ctypes.POINTER(HubNet), ctypes.POINTER(RingNetC), ctypes.c_int,
                                                           ctypes.POINTER(HarnessResult)]
        cfg = HarnessConfig()
        cfg.mode = L.MODE_REP3 if mode == "rep3" else L.MODE_PLAIN
        cfg.log_n = cfgkw.get("log_n", 10)
        cfg.n_fr, cfg.n_u16, cfg.n_u32 = cfgkw.get("n_fr", 4), cfgkw.get("n_u16", 1), cfgkw.get("n_u32", 1)
        cfg.n_flags, cfg.n_small = cfgkw.get("n_flags", 1), cfgkw.get("n_small", 0)
        cfg.gp_batch = cfgkw.get("gp_batch", 2)
        cfg.gp_log_leaves = cfgkw.get("gp_log_leaves", cfg.log_n + 1)
        cfg.precompute = 1 if cfgkw.get("precompute", True) else 0
        cfg.devices = (ctypes.c_int * 3)(device, device, device)
        cfg.seed = cfgkw.get("seed", 1)
        cfg.log_workers = log_workers
        cfg.worker_devices = (ctypes.c_int * 8)(*([device] * 8))
        cfg.leaf_fingerprints = 1 if cfgkw.get("leaf_fingerprints", False) else 0
        self.party, self.worker = party, worker
        self.nparties = 3 if mode == "rep3" else 1
        self.index = worker * self.nparties + party
        h = ctypes.c_void_p()
        rc = self._l.cozk_harness_create_participant(ctypes.byref(cfg), party, worker, ctypes.byref(h))
        self.h = h
        if rc != L.OK:
            msg = self._l.cozk_harness_error(h) if h else b"?"
            raise L.CozkError(rc, (msg or b"?").decode())

    def ctx_handle(self):
        return self._l.cozk_harness_ctx(self.h, self.index)

    def prove(self, hub, ring, verify=True):
        res = HarnessResult()
        rc = self._l.cozk_harness_prove_distributed(self.h, ctypes.byref(hub.net), ctypes.byref(ring.net) if ring is not None else None,
                                                    1 if verify else 0, ctypes.byref(res))
        if rc != L.OK:
            if hasattr(hub, "abort"):
                hub.abort()  # peers blocked in the shared-memory hub fail now instead of timing out
            for t in (hub, ring):
                if t is not None and t.error is not None:
                    raise t.error
            raise L.CozkError(rc, (self._l.cozk_harness_error(self.h) or b"?").decode())
        return res

    def last_error(self):
        return (self._l.cozk_harness_error(self.h) or b"").decode()

    def close(self):
        if getattr(self, "h", None):
            self._l.cozk_harness_destroy(self.h)
            self.h = None
