"""Build libcozk.so (all HIP kernels + the C ABI) for gfx950 with hipcc.  Cross-compiles without a GPU."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libcozk.so")
SOURCES = ["capi.hip", "msm.hip", "poly.hip", "harness.hip", "shm_hub.hip", "ring.hip"]
HEADERS = [os.path.join("host", "spartan_pub_workers.hpp"), "fr9.hip.hpp", "fr9_consts.inc", os.path.join("host", "flow_harness.hpp"), os.path.join("host", "jolt_r1cs.hpp"), os.path.join("host", "spartan_jolt.hpp"), "spartan_inner.inc", "ff.hip.hpp", "prf.hip.hpp", "ff_macc.inc", "ff_mul2.inc", "ec.hip.hpp", "fq9.hip.hpp", "fq9_consts.inc", "fq9_mac.inc", "fq9_mul.inc", "common.hpp", "poly.hip.hpp", "toggle_layer.inc", "primary_sumcheck.inc", "spartan_outer.inc", "logup.inc", os.path.join("host", "wire.hpp"), os.path.join("host", "net.hpp"), os.path.join("host", "prover.hpp"), os.path.join("host", "split.hpp"), os.path.join("host", "split_harness.hpp"), os.path.join("host", "spartan_harness.hpp"), os.path.join("host", "lookups_harness.hpp"), os.path.join("host", "outer_harness.hpp"), os.path.join("..", "..", "include", "cozk.h")]


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    srcs = [os.path.join(CSRC, s) for s in SOURCES]
    deps = srcs + [os.path.join(CSRC, h) for h in HEADERS]
    objs = []
    os.makedirs(os.path.join(HERE, "build"), exist_ok=True)
    procs = []
    for s in srcs:
        o = os.path.join(HERE, "build", os.path.basename(s) + ".o")
        objs.append(o)
        if force or _newer(o, deps):
            cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((cmd, subprocess.Popen(cmd)))
    for cmd, p in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed: " + " ".join(cmd))
    if force or procs or _newer(OUT, objs):
        cmd = ["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs + ["-ldl"]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
