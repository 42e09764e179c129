"""Host-side handles over the C ABI (include/cozk.h): context, device vectors, SRS bases, MSM.

Mirrors the reference's MSM seam: `VariableBaseMSM::{msm_field_elements, batch_msm}` behind
`PST13::{commit, batch_commit, open}` (co-jolt/src/poly/commitment/pst13.rs:282-331,428-474).
Field elements cross this layer as Python ints (canonical) or numpy uint64[n,4] Montgomery limbs.
"""
import ctypes

import numpy as np

from . import _lib as L

FR_MOD = 21888242871839275222246405745257275088548364400416034343698204186575808495617
FQ_MOD = 21888242871839275222246405745257275088696311157297823662689037894645226208583
_MONT = 1 << 256
_MASK64 = (1 << 64) - 1

_KIND_DTYPE = {L.SCALAR_U8: np.uint8, L.SCALAR_U16: np.uint16, L.SCALAR_U32: np.uint32,
               L.SCALAR_U64: np.uint64, L.SCALAR_I64: np.int64}


def fr_to_mont_limbs(values, mod=FR_MOD):
    """canonical ints -> uint64[n,4] Montgomery limbs (arkworks in-memory layout)"""
    out = np.empty((len(values), 4), dtype=np.uint64)
    for i, v in enumerate(values):
        m = (int(v) % mod) * _MONT % mod
        out[i, 0] = m & _MASK64
        out[i, 1] = (m >> 64) & _MASK64
        out[i, 2] = (m >> 128) & _MASK64
        out[i, 3] = (m >> 192) & _MASK64
    return out


_RINV = {FR_MOD: pow(_MONT, -1, FR_MOD), FQ_MOD: pow(_MONT, -1, FQ_MOD)}


def mont_limbs_to_int(limbs, mod=FR_MOD):
    """uint64[...,4] Montgomery limbs -> list of canonical ints"""
    arr = np.asarray(limbs, dtype=np.uint64).reshape(-1, 4)
    rinv = _RINV[mod]
    res = []
    for row in arr:
        m = int(row[0]) | (int(row[1]) << 64) | (int(row[2]) << 128) | (int(row[3]) << 192)
        res.append(m * rinv % mod)
    return res


def point_to_abi(pt):
    """affine (x, y) canonical ints or None -> (uint64[8], infinity flag)"""
    if pt is None:
        return np.zeros(8, dtype=np.uint64), 1
    xy = np.concatenate([fr_to_mont_limbs([pt[0]], FQ_MOD)[0], fr_to_mont_limbs([pt[1]], FQ_MOD)[0]])
    return xy, 0


def point_from_abi(xy, inf):
    if inf:
        return None
    x = mont_limbs_to_int(xy[:4], FQ_MOD)[0]
    y = mont_limbs_to_int(xy[4:], FQ_MOD)[0]
    return (x, y)


def wire_g1_encode(pt):
    """ark-serialize uncompressed G1Affine bytes of an affine point (cozk_wire_g1_encode; host only)"""
    xy, inf = point_to_abi(pt)
    out = (ctypes.c_uint8 * 64)()
    rc = L.lib().cozk_wire_g1_encode(xy.ctypes.data, inf, out)
    if rc != L.OK:
        raise L.CozkError(rc, "wire_g1_encode")
    return bytes(out)


def wire_g1_decode(b):
    """inverse, with arkworks' Validate::Yes checks (raises CozkError for bytes arkworks would reject)"""
    buf = (ctypes.c_uint8 * 64)(*bytes(b))
    xy = np.zeros(8, dtype=np.uint64)
    inf = ctypes.c_int()
    rc = L.lib().cozk_wire_g1_decode(buf, xy.ctypes.data, ctypes.byref(inf))
    if rc != L.OK:
        raise L.CozkError(rc, "wire_g1_decode: invalid point encoding")
    return point_from_abi(xy, inf.value)


class Context:
    """One per (party, GPU).  Not thread-safe (single-owner, like an IoContext fork)."""

    def __init__(self, device=0):
        self._l = L.lib()
        h = ctypes.c_void_p()
        rc = self._l.cozk_ctx_create(device, ctypes.byref(h))
        if rc != L.OK:
            if rc == L.ERR_NO_DEVICE:
                raise L.CozkError(rc, "no HIP device visible: the cozk engine has no CPU fallback")
            raise L.CozkError(rc, "cozk_ctx_create failed")
        self.h = h
        self.device = device

    def check(self, rc):
        if rc != L.OK:
            msg = self._l.cozk_last_error(self.h)
            raise L.CozkError(rc, msg.decode() if msg else "?")

    def synchronize(self):
        self.check(self._l.cozk_ctx_synchronize(self.h))

    def set_resident_rounds(self, enable):
        """True / False: force the resident round kernel of cozk_layer_prove_rounds on / off; None: the automatic
        default (on only while this is the one live context on its device in the process)"""
        self.check(self._l.cozk_ctx_set_resident_rounds(self.h, -1 if enable is None else (1 if enable else 0)))

    # ---- native Rep3 ring over RCCL (cozk_ring_*): one party per GPU / process
    @staticmethod
    def ring_unique_id():
        """128 opaque bytes for cozk_ring_init; ONE participant draws them, the host hands them to the others"""
        buf = (ctypes.c_uint8 * 128)()
        rc = L.lib().cozk_ring_unique_id(buf)
        if rc != L.OK:
            raise L.CozkError(rc, "cozk_ring_unique_id failed (librccl unavailable?)")
        return bytes(buf)

    def ring_init(self, ring_id, rank, nranks=3):
        self.check(self._l.cozk_ring_init(self.h, bytes(ring_id), rank, nranks))

    def ring_destroy(self):
        self.check(self._l.cozk_ring_destroy(self.h))

    def ring_info(self):
        r, n, b = ctypes.c_int(), ctypes.c_int(), ctypes.c_uint64()
        self.check(self._l.cozk_ring_info(self.h, ctypes.byref(r), ctypes.byref(n), ctypes.byref(b)))
        return r.value, n.value, b.value

    def reshare(self, send):
        """send `send` to the next party, return what the previous party sent (cozk_reshare)"""
        recv = Vec.alloc(self, len(send), L.SCALAR_FR)
        self.check(self._l.cozk_reshare(self.h, send.h, recv.h))
        return recv

    def rep3_mul_vec(self, xa, xb, ya, yb, key_self, key_prev, counter=0):
        """rep3::arithmetic::mul_vec, whole (local product + mask + ring exchange) -> (c.a, c.b) vectors"""
        a, b = ctypes.c_void_p(), ctypes.c_void_p()
        self.check(self._l.cozk_rep3_mul_vec(self.h, xa.h, xb.h, ya.h, yb.h, L.prf_key(key_self), L.prf_key(key_prev), counter,
                                             ctypes.byref(a), ctypes.byref(b)))
        return Vec(self, a, L.SCALAR_FR), Vec(self, b, L.SCALAR_FR)

    def close(self):
        if self.h:
            self._l.cozk_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- profiling of the dominant kernel
    def prof_enable(self, on=True):
        self.check(self._l.cozk_prof_enable(self.h, 1 if on else 0))

    def prof_read(self):
        n = ctypes.c_uint64()
        ms = ctypes.c_double()
        adds = ctypes.c_uint64()
        nbytes = ctypes.c_uint64()
        self.check(self._l.cozk_prof_read(self.h, ctypes.byref(n), ctypes.byref(ms), ctypes.byref(adds), ctypes.byref(nbytes)))
        return n.value, ms.value, adds.value, nbytes.value

    def bench_montmul(self, lanes, iters, variant=0):
        ms = ctypes.c_double()
        self.check(self._l.cozk_bench_montmul(self.h, lanes, iters, variant, ctypes.byref(ms)))
        return ms.value

    # ---- host G1 helpers
    def g1_sum(self, points):
        k = len(points)
        xy = np.zeros((k, 8), dtype=np.uint64)
        inf = np.zeros(k, dtype=np.int32)
        for i, p in enumerate(points):
            xy[i], inf[i] = point_to_abi(p)
        out = np.zeros(8, dtype=np.uint64)
        oi = ctypes.c_int()
        self.check(self._l.cozk_g1_sum(self.h, xy.ctypes.data, inf.ctypes.data, k, out.ctypes.data, ctypes.byref(oi)))
        return point_from_abi(out, oi.value)

    def g1_mul(self, point, scalar):
        xy, inf = point_to_abi(point)
        s = fr_to_mont_limbs([scalar])[0]
        out = np.zeros(8, dtype=np.uint64)
        oi = ctypes.c_int()
        self.check(self._l.cozk_g1_mul(self.h, xy.ctypes.data, inf, s.ctypes.data, out.ctypes.data, ctypes.byref(oi)))
        return point_from_abi(out, oi.value)


class Vec:
    """Device-resident scalar vector (`cozk_vec`)."""

    def __init__(self, ctx, handle, kind):
        self.ctx = ctx
        self.h = handle
        self.kind = kind

    @classmethod
    def alloc(cls, ctx, n, kind=L.SCALAR_FR):
        h = ctypes.c_void_p()
        ctx.check(ctx._l.cozk_vec_alloc(ctx.h, n, kind, ctypes.byref(h)))
        return cls(ctx, h, kind)

    @classmethod
    def from_ints(cls, ctx, values, kind=L.SCALAR_FR):
        """values: canonical ints (FR) or small integers (other kinds)"""
        if kind == L.SCALAR_FR:
            arr = fr_to_mont_limbs(values)
        else:
            arr = np.asarray(values, dtype=_KIND_DTYPE[kind])
        return cls.from_numpy(ctx, arr, kind)

    @classmethod
    def from_numpy(cls, ctx, arr, kind=L.SCALAR_FR):
        arr = np.ascontiguousarray(arr)
        n = arr.shape[0]
        h = ctypes.c_void_p()
        ctx.check(ctx._l.cozk_vec_upload(ctx.h, arr.ctypes.data if n else None, n, kind, ctypes.byref(h)))
        return cls(ctx, h, kind)

    @classmethod
    def random(cls, ctx, n, seed, kind=L.SCALAR_FR, max_bits=0):
        v = cls.alloc(ctx, n, kind)
        ctx.check(ctx._l.cozk_vec_fill_random(ctx.h, v.h, seed, max_bits))
        return v

    def __len__(self):
        return self.ctx._l.cozk_vec_len(self.h)

    def device_ptr(self):
        return self.ctx._l.cozk_vec_device_ptr(self.h)

    def to_numpy(self):
        n = len(self)
        if self.kind == L.SCALAR_FR:
            out = np.empty((n, 4), dtype=np.uint64)
        else:
            out = np.empty(n, dtype=_KIND_DTYPE[self.kind])
        self.ctx.check(self.ctx._l.cozk_vec_download(self.ctx.h, self.h, out.ctypes.data if n else None))
        return out

    def to_ints(self):
        a = self.to_numpy()
        if self.kind == L.SCALAR_FR:
            return mont_limbs_to_int(a)
        return [int(x) for x in a]

    def narrow(self, kind):
        """this FR vector as a U32 / U64 vector of the same values (cozk_vec_narrow: what msm_field_elements' dispatch on the scalars'
        bit length amounts to); raises CozkError if a value does not fit"""
        h = ctypes.c_void_p()
        self.ctx.check(self.ctx._l.cozk_vec_narrow(self.ctx.h, self.h, kind, ctypes.byref(h)))
        return Vec(self.ctx, h, kind)

    def rep3_share(self, key0, key1, party, counter=0):
        """Rep3 shares (a, b) of this secret vector for `party` (cozk_rep3_share_vec); key0 / key1 = 32-byte PRF keys"""
        a, b = ctypes.c_void_p(), ctypes.c_void_p()
        self.ctx.check(self.ctx._l.cozk_rep3_share_vec(self.ctx.h, self.h, L.prf_key(key0), L.prf_key(key1), counter, party,
                                                       ctypes.byref(a), ctypes.byref(b)))
        return Vec(self.ctx, a, L.SCALAR_FR), Vec(self.ctx, b, L.SCALAR_FR)

    def rep3_scatter(self, key0, key1, party, party_ctx, counter=0):
        """the witness scatter device to device (cozk_rep3_scatter): this (dealer-side) secret vector's shares for `party`,
        as vectors owned by `party_ctx` (same or another GPU)"""
        a, b = ctypes.c_void_p(), ctypes.c_void_p()
        self.ctx.check(self.ctx._l.cozk_rep3_scatter(self.ctx.h, self.h, L.prf_key(key0), L.prf_key(key1), counter, party_ctx.h, party,
                                                     ctypes.byref(a), ctypes.byref(b)))
        return Vec(party_ctx, a, L.SCALAR_FR), Vec(party_ctx, b, L.SCALAR_FR)

    @classmethod
    def prf(cls, ctx, n, key, counter=0):
        """out[i] = PRF(key, counter + i): the keyed ChaCha12 stream every share / mask is drawn from"""
        v = cls.alloc(ctx, n, L.SCALAR_FR)
        ctx.check(ctx._l.cozk_vec_fill_prf(ctx.h, v.h, L.prf_key(key), counter))
        return v

    def binop(self, op, other, base_field=False):
        out = Vec.alloc(self.ctx, len(self), L.SCALAR_FR)
        self.ctx.check(self.ctx._l.cozk_vec_binop(self.ctx.h, op, 1 if base_field else 0, self.h, other.h, out.h))
        return out

    def free(self):
        if self.h:
            self.ctx._l.cozk_vec_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Bases:
    """Device-resident SRS slice (`ck.powers_of_g[i]`), optionally with the 16-window table."""

    def __init__(self, ctx, handle):
        self.ctx = ctx
        self.h = handle

    @classmethod
    def upload(cls, ctx, points, precompute=True):
        n = len(points)
        xy = np.zeros((n, 8), dtype=np.uint64)
        inf = np.zeros(n, dtype=np.uint8)
        for i, p in enumerate(points):
            xy[i], inf[i] = point_to_abi(p)
        h = ctypes.c_void_p()
        ctx.check(ctx._l.cozk_bases_upload(ctx.h, xy.ctypes.data, inf.ctypes.data, n, 1 if precompute else 0,
                                           ctypes.byref(h)))
        return cls(ctx, h)

    @classmethod
    def from_scalars(cls, ctx, scalars_vec, g=(1, 2), precompute=True):
        """bases[i] = scalars[i] * g, computed on the GPU (`MultilinearPC::setup`)"""
        gxy, _ = point_to_abi(g)
        h = ctypes.c_void_p()
        ctx.check(ctx._l.cozk_bases_from_scalars(ctx.h, scalars_vec.h, gxy.ctypes.data, 1 if precompute else 0,
                                                 ctypes.byref(h)))
        return cls(ctx, h)

    def pair_sums(self, precompute=True):
        h = ctypes.c_void_p()
        self.ctx.check(self.ctx._l.cozk_bases_pair_sums(self.ctx.h, self.h, 1 if precompute else 0, ctypes.byref(h)))
        return Bases(self.ctx, h)

    def __len__(self):
        return self.ctx._l.cozk_bases_len(self.h)

    def download(self, offset=0, n=None):
        if n is None:
            n = len(self) - offset
        xy = np.zeros((n, 8), dtype=np.uint64)
        inf = np.zeros(n, dtype=np.uint8)
        self.ctx.check(self.ctx._l.cozk_bases_download(self.ctx.h, self.h, offset, n, xy.ctypes.data, inf.ctypes.data))
        return [point_from_abi(xy[i], inf[i]) for i in range(n)]

    # ---- VariableBaseMSM
    def msm(self, scalars, offset=0):
        """scalars: Vec (device resident) -> affine point or None"""
        out = np.zeros(8, dtype=np.uint64)
        oi = ctypes.c_int()
        self.ctx.check(self.ctx._l.cozk_msm_vec(self.ctx.h, self.h, offset, scalars.h, out.ctypes.data, ctypes.byref(oi)))
        return point_from_abi(out, oi.value)

    def msm_host(self, arr, kind=L.SCALAR_FR, offset=0):
        """host scalars (numpy) -> affine point; PCIe-inclusive form of the seam"""
        arr = np.ascontiguousarray(arr)
        n = arr.shape[0]
        out = np.zeros(8, dtype=np.uint64)
        oi = ctypes.c_int()
        self.ctx.check(self.ctx._l.cozk_msm(self.ctx.h, self.h, offset, arr.ctypes.data if n else None, kind, n,
                                            out.ctypes.data, ctypes.byref(oi)))
        return point_from_abi(out, oi.value)

    def batch_msm_raw(self, vecs, offset=0):
        """k device vectors -> (uint64[k,8], int32[k]) without int conversion (bench path)"""
        k = len(vecs)
        arr = (ctypes.c_void_p * k)(*[v.h for v in vecs])
        out = np.zeros((k, 8), dtype=np.uint64)
        inf = np.zeros(k, dtype=np.int32)
        self.ctx.check(self.ctx._l.cozk_batch_msm_vec(self.ctx.h, self.h, offset, arr, k, out.ctypes.data,
                                                      inf.ctypes.data))
        return out, inf

    def batch_msm(self, vecs, offset=0):
        out, inf = self.batch_msm_raw(vecs, offset)
        return [point_from_abi(out[i], inf[i]) for i in range(len(vecs))]

    def free(self):
        if self.h:
            self.ctx._l.cozk_bases_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass
