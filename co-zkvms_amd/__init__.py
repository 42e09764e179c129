"""co-zkvms_amd: MI355X-native engine for the sumcheck + polynomial-commitment hot path of
ChainSafe/co-zkvms (co-jolt / co-noir-spartan workers).  HIP kernels + C ABI live in csrc/ and are
loaded from libcozk.so (no CPU fallback); this package is the host-side mirror of the reference's
interfaces for that path.  Import with importlib.import_module("co-zkvms_amd")."""
from . import _lib
from ._lib import (CozkError, SCALAR_FR, SCALAR_U8, SCALAR_U16, SCALAR_U32, SCALAR_U64, SCALAR_I64,
                   LOW_TO_HIGH, HIGH_TO_LOW, MODE_PLAIN, MODE_REP3, OP_ADD, OP_SUB, OP_MUL)
from .engine import (Context, Vec, Bases, FR_MOD, FQ_MOD, fr_to_mont_limbs, mont_limbs_to_int,
                     point_to_abi, point_from_abi, wire_g1_encode, wire_g1_decode)
from .poly import (Rep3DensePolynomial, Rep3DenseInterleavedPolynomial, SplitEqPolynomial, eq_evals,
                   open_quadratic_evals, pst_fold, prod_sumcheck_evals, spartan_first_round, spartan_second_round,
                   sparse_matvec3, fingerprint_leaves, rep3_mul_vec_local)
from .harness import Harness, HarnessConfig, HarnessResult
from .spartan import SpartanHarness, SpartanConfig, SpartanResult
