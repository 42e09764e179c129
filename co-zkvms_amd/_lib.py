"""ctypes loader for libcozk.so (the HIP engine behind include/cozk.h).

There is deliberately NO CPU fallback: if the shared library is missing, or no MI355X is visible
when a context is created, the product path raises.  (The CPU restatement under oracle/ is test
infrastructure and is never imported from here.)
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libcozk.so")

OK = 0
ERR_NO_DEVICE = -5

SCALAR_FR, SCALAR_U8, SCALAR_U16, SCALAR_U32, SCALAR_U64, SCALAR_I64 = range(6)
LOW_TO_HIGH, HIGH_TO_LOW = 0, 1
MODE_PLAIN, MODE_REP3 = 1, 2
OP_ADD, OP_SUB, OP_MUL = 0, 1, 2


PRF_KEY_BYTES = 32


def prf_key(k):
    """a 32-byte ChaCha12 PRF key for the ABI: bytes of length 32, or None (all-zero key, unmasked calls only)"""
    if k is None:
        return None
    k = bytes(k)
    if len(k) != PRF_KEY_BYTES:
        raise ValueError("PRF keys are %d bytes" % PRF_KEY_BYTES)
    return k


class CozkError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"cozk error {code}: {msg}")
        self.code = code


_lib = None


def lib():
    """Load libcozk.so once; raise loudly when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
        _lib = ctypes.CDLL(LIB_PATH)
        _declare(_lib)
    return _lib


_vp = ctypes.c_void_p
_sz = ctypes.c_size_t
_i = ctypes.c_int
_u64 = ctypes.c_uint64
_pp = ctypes.POINTER(ctypes.c_void_p)

# name -> (restype, argtypes); also the list the symbol-export test checks against include/cozk.h
SIGNATURES = {
    "cozk_device_count": (_i, [ctypes.POINTER(_i)]),
    "cozk_ctx_create": (_i, [_i, _pp]),
    "cozk_ctx_destroy": (_i, [_vp]),
    "cozk_last_error": (ctypes.c_char_p, [_vp]),
    "cozk_ctx_synchronize": (_i, [_vp]),
    "cozk_ctx_stream": (_i, [_vp, _pp]),
    "cozk_vec_upload": (_i, [_vp, _vp, _sz, _i, _pp]),
    "cozk_vec_narrow": (_i, [_vp, _vp, _i, _pp]),
    "cozk_vec_alloc": (_i, [_vp, _sz, _i, _pp]),
    "cozk_vec_download": (_i, [_vp, _vp, _vp]),
    "cozk_vec_free": (_i, [_vp]),
    "cozk_vec_len": (_sz, [_vp]),
    "cozk_vec_device_ptr": (_vp, [_vp]),
    "cozk_vec_fill_random": (_i, [_vp, _vp, _u64, _i]),
    "cozk_vec_scale": (_i, [_vp, _vp, _vp]),
    "cozk_layer_compute_cubic_evals": (_i, [_vp, _vp, _vp, _vp]),
    "cozk_vec_binop": (_i, [_vp, _i, _i, _vp, _vp, _vp]),
    "cozk_bases_upload": (_i, [_vp, _vp, _vp, _sz, _i, _pp]),
    "cozk_bases_from_scalars": (_i, [_vp, _vp, _vp, _i, _pp]),
    "cozk_bases_download": (_i, [_vp, _vp, _sz, _sz, _vp, _vp]),
    "cozk_bases_free": (_i, [_vp]),
    "cozk_bases_len": (_sz, [_vp]),
    "cozk_bases_pair_sums": (_i, [_vp, _vp, _i, _pp]),
    "cozk_msm": (_i, [_vp, _vp, _sz, _vp, _i, _sz, _vp, ctypes.POINTER(_i)]),
    "cozk_msm_vec": (_i, [_vp, _vp, _sz, _vp, _vp, ctypes.POINTER(_i)]),
    "cozk_batch_msm_vec": (_i, [_vp, _vp, _sz, _vp, _sz, _vp, _vp]),
    "cozk_batch_msm_slices": (_i, [_vp, _vp, _vp, _vp, _vp, _sz, _vp, _vp]),
    "cozk_g1_sum": (_i, [_vp, _vp, _vp, _sz, _vp, ctypes.POINTER(_i)]),
    "cozk_g1_mul": (_i, [_vp, _vp, _i, _vp, _vp, ctypes.POINTER(_i)]),
    "cozk_poly_create": (_i, [_vp, _i, _vp, _vp, _pp]),
    "cozk_poly_chunk": (_i, [_vp, _vp, _sz, _sz, _pp]),
    "cozk_poly_free": (_i, [_vp]),
    "cozk_poly_len": (_sz, [_vp]),
    "cozk_poly_mode": (_i, [_vp]),
    "cozk_poly_download": (_i, [_vp, _vp, _vp, _vp]),
    "cozk_poly_share_view": (_i, [_vp, _vp, _i, _pp]),
    "cozk_poly_bind": (_i, [_vp, _vp, _vp, _i]),
    "cozk_poly_get_coeff": (_i, [_vp, _vp, _sz, _vp, _vp]),
    "cozk_eq_evals": (_i, [_vp, _vp, _i, _pp]),
    "cozk_poly_batch_evaluate_at_chi": (_i, [_vp, _vp, _sz, _vp, _vp]),
    "cozk_eq_plus_one_evals": (_i, [_vp, _vp, _i, _pp]),
    "cozk_poly_batch_dot_public": (_i, [_vp, _vp, _sz, _vp, _sz, _vp]),
    "cozk_poly_dot_product_with_public": (_i, [_vp, _vp, _vp, _vp, _vp]),
    "cozk_poly_linear_combination": (_i, [_vp, _vp, _vp, _sz, _i, _i, _pp]),
    "cozk_open_quadratic_evals": (_i, [_vp, _vp, _vp, _sz, _vp]),
    "cozk_prod_sumcheck_evals": (_i, [_vp, _vp, _sz, _i, _vp]),
    "cozk_spartan_first_round": (_i, [_vp, _vp, _vp, _vp, _vp, _vp]),
    "cozk_spartan_second_round": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "cozk_sparse_matvec3": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _pp, _pp, _pp]),
    "cozk_pst_fold": (_i, [_vp, _vp, _vp, _vp, _vp]),
    "cozk_layer_create": (_i, [_vp, _i, _vp, _vp, _i, _pp]),
    "cozk_layer_free": (_i, [_vp]),
    "cozk_layer_len": (_sz, [_vp]),
    "cozk_layer_download": (_i, [_vp, _vp, _vp, _vp]),
    "cozk_layer_clone": (_i, [_vp, _vp, _pp]),
    "cozk_layer_bind": (_i, [_vp, _vp, _vp]),
    "cozk_layer_compute_cubic": (_i, [_vp, _vp, _vp, _vp, _vp]),
    "cozk_ctx_set_resident_rounds": (_i, [_vp, _i]),
    "cozk_rep3_share_vec": (_i, [_vp, _vp, ctypes.c_char_p, ctypes.c_char_p, _u64, _i, _pp, _pp]),
    "cozk_rep3_scatter": (_i, [_vp, _vp, ctypes.c_char_p, ctypes.c_char_p, _u64, _vp, _i, _pp, _pp]),
    "cozk_vec_fill_prf": (_i, [_vp, _vp, ctypes.c_char_p, _u64]),
    "cozk_layer_round": (_i, [_vp, _vp, _vp, _vp, _vp, _vp]),
    "cozk_fingerprint_leaves": (_i, [_vp, _vp, _vp, _sz, _vp, _vp, _sz, _vp, _i, _i, _vp, _vp, _sz, _sz]),
    "cozk_layer_prove_rounds": (_i, [_vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp]),
    "cozk_layer_final_claims": (_i, [_vp, _vp, _vp]),
    "cozk_layer_output_local": (_i, [_vp, _vp, _i, ctypes.c_char_p, ctypes.c_char_p, _u64, _pp]),
    "cozk_rep3_mul_vec_local": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _i, ctypes.c_char_p, ctypes.c_char_p, _u64, _pp]),
    "cozk_layer_claimed_outputs": (_i, [_vp, _vp, _vp]),
    "cozk_spliteq_new": (_i, [_vp, _vp, _i, _pp]),
    "cozk_spliteq_free": (_i, [_vp]),
    "cozk_spliteq_lens": (_i, [_vp, ctypes.POINTER(_sz), ctypes.POINTER(_sz)]),
    "cozk_spliteq_bind": (_i, [_vp, _vp, _vp]),
    "cozk_prof_enable": (_i, [_vp, _i]),
    "cozk_prof_read": (_i, [_vp, ctypes.POINTER(_u64), ctypes.POINTER(ctypes.c_double), ctypes.POINTER(_u64), ctypes.POINTER(_u64)]),
    "cozk_prof_kernel_name": (ctypes.c_char_p, [_i]),
    "cozk_prof_read_kernel": (_i, [_vp, _i, ctypes.POINTER(_u64), ctypes.POINTER(ctypes.c_double), ctypes.POINTER(_u64)]),
    "cozk_layer_as_poly": (_i, [_vp, _vp, _pp]),
    "cozk_ring_unique_id": (_i, [_vp]),
    "cozk_ring_init": (_i, [_vp, ctypes.c_char_p, _i, _i]),
    "cozk_ring_destroy": (_i, [_vp]),
    "cozk_ring_info": (_i, [_vp, ctypes.POINTER(_i), ctypes.POINTER(_i), ctypes.POINTER(_u64)]),
    "cozk_reshare": (_i, [_vp, _vp, _vp]),
    "cozk_rep3_mul_vec": (_i, [_vp, _vp, _vp, _vp, _vp, ctypes.c_char_p, ctypes.c_char_p, _u64, _pp, _pp]),
    "cozk_ring_net_native": (_i, [_vp, _vp]),
    "cozk_wire_g1_encode": (_i, [_vp, _i, _vp]),
    "cozk_wire_g1_decode": (_i, [_vp, _vp, ctypes.POINTER(_i)]),
    "cozk_bench_montmul": (_i, [_vp, _sz, _i, _i, ctypes.POINTER(ctypes.c_double)]),
}


def _declare(l):
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(l, name)
        fn.restype = res
        fn.argtypes = args
