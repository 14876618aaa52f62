"""CPU: the self-launcher (simpleinfer_amd/launch.py, what `python bench.py --gpus N` uses when no launcher set RANK /
WORLD_SIZE) and the node-local rank group of include/si_shard.h (POSIX shm rendezvous, barrier, byte all-gather) that the
direct output all-gather is built on.  No GPU."""
import json
import os
import subprocess
import sys
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = os.path.join(ROOT, "tests", "_rank_child.py")


def _spawn(mode, n, *extra, timeout=120):
    from simpleinfer_amd import launch
    code, out = launch.spawn_ranks([sys.executable, CHILD, mode] + [str(a) for a in extra], n, timeout=timeout)
    line = [ln for ln in out.splitlines() if ln.startswith("{")]
    return code, (json.loads(line[-1]) if line else None)


@pytest.mark.parametrize("world", [1, 2, 4, 8])
def test_node_group_barrier_and_allgather(native_libs, world):
    code, res = _spawn("group", world)
    assert code == 0 and res is not None
    assert res["ok"] and res["n"] == 50
    assert res["max"] == pytest.approx(1.5 * (world - 1))
    assert res["big"] == list(range(world))


def test_launcher_env_is_what_torch_distributed_expects():
    code, res = _spawn("gloo", 2)
    assert code == 0 and res == {"sum": 3.0, "world": 2}


def test_launcher_stops_the_other_ranks_when_one_fails():
    t0 = time.time()
    code, res = _spawn("fail", 2)
    assert code == 3 and res is None
    assert time.time() - t0 < 30, "rank 0 (sleeping 60 s) was not stopped"


def test_a_timed_out_barrier_kills_the_group_on_every_rank(native_libs, tmp_path):
    """si_group_barrier: rank 0 times out (-103) while rank 1 is late; rank 1's barrier then fails with SI_SHARD_E_PEER (-105)
    instead of completing with rank 0's stale arrival, and every later barrier on both ranks fails too."""
    code, _ = _spawn("dead_group", 2, tmp_path)
    assert code == 0
    c0 = json.load(open(tmp_path / "codes0.json"))
    c1 = json.load(open(tmp_path / "codes1.json"))
    assert c0[0] == -103 and all(c in (-103, -105) for c in c0[1:]), c0
    assert c1 == [-105, -105, -105], c1


def test_group_rejects_bad_arguments(native_libs):
    import ctypes as C
    _, host = native_libs
    g = C.c_void_p()
    assert host.si_group_create(b"no_slash", 0, 1, 1.0, C.byref(g)) == -101
    assert host.si_group_create(b"/si_test_x", 2, 2, 1.0, C.byref(g)) == -101
    # a lone rank of a 2-rank group times out instead of hanging
    t0 = time.time()
    assert host.si_group_create(("/si_test_lonely_%d" % os.getpid()).encode(), 0, 2, 0.5, C.byref(g)) == -103
    assert time.time() - t0 < 10
    assert host.si_group_create(("/si_test_solo_%d" % os.getpid()).encode(), 0, 1, 1.0, C.byref(g)) == 0
    assert host.si_group_world(g) == 1 and host.si_group_barrier(g) == 0
    buf = C.create_string_buffer(8)
    assert host.si_group_allgather(g, b"abcdefgh", 8, buf) == 0 and buf.raw == b"abcdefgh"
    assert host.si_group_allgather(g, b"x", 5000, buf) == -104
    host.si_group_destroy(g)


def test_bench_self_launches_and_fails_cleanly_without_devices(native_libs):
    """`python bench.py --gpus 2` with no launcher in the environment spawns two ranks itself (the parent never touches the
    GPU); here there is no device, so both ranks must report that and the parent must exit non-zero -- not hang, not
    ask for torchrun."""
    from simpleinfer_amd import device_count
    if device_count() >= 2:
        pytest.skip("two HIP devices are present: the run would succeed")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--size", "64", "--batch", "1"], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode != 0
    assert "torch.distributed.run" not in p.stderr
    assert "HIP device" in p.stderr or "devices" in p.stderr, p.stderr[-2000:]
    assert not [ln for ln in p.stdout.splitlines() if ln.startswith("{")]


def test_rccl_entry_points_fail_cleanly_without_the_library(native_libs, tmp_path):
    """include/si_shard.h si_rccl_*: librccl.so is dlopen()ed on first use, nothing links against it.  With SI_RCCL_LIB naming a
    file that does not exist every entry point reports SI_SHARD_E_SYS (-102) / "not available" instead of failing to load the
    host library or crashing -- and the RCCL-backed gather refuses the same way on every rank."""
    code = subprocess.run([sys.executable, "-c", """
import ctypes as C, os, sys
sys.path.insert(0, %r)
from simpleinfer_amd import _native, shard
H = _native.host()
assert H.si_rccl_available() == 0
g = shard.NodeGroup("/si_test_norccl_%%d" %% os.getpid(), 0, 1, 5.0)
c = C.c_void_p()
rc = H.si_rccl_init(g._g, 0, C.byref(c))
assert rc == -102 and not c.value, rc
d = C.c_void_p()
rc = H.si_gather_create_mode(g._g, 0, 4096, 2, 1, C.byref(d))
assert rc != 0 and not d.value, rc
assert H.si_rccl_allgather(None, None, None, 0, None) == -101
assert H.si_rccl_destroy(None) == 0
print("ok")
""" % ROOT], env=dict(os.environ, SI_RCCL_LIB=str(tmp_path / "no_such_librccl.so")), capture_output=True, text=True)
    assert code.returncode == 0 and "ok" in code.stdout, code.stdout + code.stderr


def test_gather_mode_argument_is_checked(native_libs):
    import ctypes as C
    _, host = native_libs
    d = C.c_void_p()
    assert host.si_gather_create_mode(None, 0, 4096, 2, 0, C.byref(d)) == -101
    assert host.si_gather_mode(None) == -1
