"""gloo world-size-3 CPU test of the variable-length byte all-gather that carries the star messages of the
distributed (replicated-coordinator) form: fast path (<= 2040 B in one collective) and long-message path."""
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = textwrap.dedent("""
    import importlib, os, sys
    sys.path.insert(0, %r)
    import torch.distributed as dist
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    P = importlib.import_module("co-zkvms_amd.party_dist")
    g = dist.new_group(backend="gloo")
    for sizes in ([0, 5, 300], [2040, 2040, 2040], [10, 5000, 0], [70000, 1, 2]):
        mine = bytes([(rank * 7 + i) %% 251 for i in range(sizes[rank])])
        got = P.all_gather_bytes(g, world, mine)
        assert [len(x) for x in got] == sizes, (sizes, [len(x) for x in got])
        for r in range(world):
            assert got[r] == bytes([(r * 7 + i) %% 251 for i in range(sizes[r])])
    dist.destroy_process_group()
    print("rank", rank, "ok")
""") % ROOT


def test_all_gather_bytes_gloo_world3(tmp_path):
    script = tmp_path / "w.py"
    script.write_text(WORKER)
    procs = []
    for r in range(3):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="3", MASTER_ADDR="127.0.0.1", MASTER_PORT="29588")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    for p in procs:
        out, _ = p.communicate(timeout=300)
        assert p.returncode == 0, out.decode()
        assert b"ok" in out


SHM_WORKER = textwrap.dedent("""
    import importlib, os, sys
    sys.path.insert(0, %r)
    rank, world, name = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
    P = importlib.import_module("co-zkvms_amd.party_dist")
    hub = P.ShmHub(rank, world, name=name, slot_bytes=1 << 17, timeout_ms=60000, create=False)
    def msg(r, rnd):
        n = [0, 5, 300, 2040, 70000, 1][(r + rnd) %% 6]
        return bytes([(r * 7 + rnd * 13 + i) %% 251 for i in range(n)])
    for rnd in range(400):
        got = hub.all_gather(msg(rank, rnd), cap=1 << 17)
        for r in range(world):
            assert got[r] == msg(r, rnd), (rnd, r, len(got[r]))
    hub.close()
    print("rank", rank, "ok")
""") % ROOT


def test_shm_hub_world4(tmp_path):
    """libcozk's shared-memory hub (csrc/shm_hub.hip): 4 processes, 400 back-to-back variable-length
    all-gathers (0 B .. 70 kB) -- exercises the double-buffer reuse and the publish/acquire ordering"""
    import importlib
    sys.path.insert(0, ROOT)
    P = importlib.import_module("co-zkvms_amd.party_dist")
    name = "/cozk_test_%d" % os.getpid()
    script = tmp_path / "s.py"
    script.write_text(SHM_WORKER)
    creator = P.ShmHub(0, 4, name=name, slot_bytes=1 << 17, create=True)  # makes the segment; the 4 children attach to it
    creator.close()
    try:
        procs = [subprocess.Popen([sys.executable, str(script), str(r), "4", name], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
                 for r in range(4)]
        for p in procs:
            out, _ = p.communicate(timeout=300)
            assert p.returncode == 0, out.decode()
            assert b"ok" in out
    finally:
        try:
            os.unlink("/dev/shm" + name)
        except OSError:
            pass
