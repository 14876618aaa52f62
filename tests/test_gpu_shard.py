"""GPU: the multi-rank path on ONE device.  hipIpc handles work between processes that share a device, so the direct
output all-gather of include/si_shard.h (IPC-shared gathered buffers, one device-to-device copy per peer, node barrier)
and bench.py's self-launched `--gpus 2` run are exercised end to end on the 1-GPU box; only the xGMI links are missing."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = os.path.join(ROOT, "tests", "_rank_child.py")


@pytest.mark.parametrize("world,slots,graph", [(2, 3, 0), (3, 3, 1), (2, 1, 0), (8, 3, 0), (4, 4, 1)])
def test_direct_gather_between_processes_is_bit_exact(gpu, tmp_path, world, slots, graph):
    """Every rank ends up with every rank's slab, bit for bit, for five consecutive steps with different inputs: overlapped
    (3 slots: step s computes while step s-1's slabs travel), with hipGraph replay (one captured graph per output slot),
    unoverlapped (1 slot), and at the node's real rank count (8 ranks, 7 peer copies per rank and step, on the one device: the
    handle exchange, the shm barrier and the slot protocol at world = 8)."""
    from simpleinfer_amd import launch
    code, out = launch.spawn_ranks([sys.executable, CHILD, "gather", str(tmp_path), str(slots), str(graph)], world, timeout=600)
    lines = [ln for ln in out.splitlines() if ln.startswith("{")]
    assert code == 0 and lines, out
    res = json.loads(lines[-1])
    assert res["ok"] == [1] * world
    assert res["shape"][0] == 2 * world


def test_rccl_behind_the_c_abi_world_1_round_trip(gpu, tmp_path):
    """include/si_shard.h si_rccl_*: librccl.so through dlopen, the unique id through the node group, ncclAllGather on a caller
    stream -- no torch.  One rank is what a 1-GPU box can run (RCCL refuses two ranks on one device): the communicator comes up,
    an out-of-place and an in-place all-gather move the bytes, and the RCCL-backed gather of ShardedForward (mode "rccl") serves
    five steps bit-exactly; mode "auto" stays on the direct path where that works."""
    import numpy as np
    from simpleinfer_amd import hipops, launch, shard
    assert shard.rccl_available()
    g = shard.NodeGroup("/si_test_rccl_%d" % os.getpid(), 0, 1, 30.0)
    comm = shard.RcclComm(g, 0)
    src = np.arange(1 << 16, dtype=np.uint32)
    a = hipops.DeviceBuffer.from_numpy(src)
    b = hipops.DeviceBuffer(src.nbytes)
    comm.allgather(a.ptr, b.ptr, src.nbytes)
    from simpleinfer_amd import _native
    _native.hip().si_hip_device_sync()
    assert np.array_equal(b.to_numpy(src.shape, np.uint32), src)
    comm.allgather(a.ptr, a.ptr, src.nbytes)            # in place (rank 0 of 1: send == recv)
    _native.hip().si_hip_device_sync()
    assert np.array_equal(a.to_numpy(src.shape, np.uint32), src)
    comm.close(); g.close(); a.free(); b.free()
    for gmode, want in (("rccl", "rccl"), ("auto", "direct")):
        code, out = launch.spawn_ranks([sys.executable, CHILD, "gather", str(tmp_path), "3", "0", gmode], 1, timeout=600)
        lines = [ln for ln in out.splitlines() if ln.startswith("{")]
        assert code == 0 and lines, out
        res = json.loads(lines[-1])
        assert res["ok"] == [1] and res["mode"] == want, res


def test_peers_are_named_by_pci_bus_id_when_every_rank_has_its_own_visible_devices(gpu, tmp_path):
    """Launchers and container runtimes hand every rank its own HIP_VISIBLE_DEVICES, so every rank calls its GPU "device 0" and
    a device INDEX means nothing to a peer: si_gather_create advertises the PCI bus id instead and resolves a peer's GPU to the
    local index of that bus id (or relies on the IPC mapping when the GPU is hidden).  On the 1-GPU box: two ranks that reach the
    same GPU through DIFFERENT visibility settings (HIP_VISIBLE_DEVICES=0 for one, ROCR_VISIBLE_DEVICES=0 for the other) -- the
    gather comes up by bus id and is bit-exact."""
    from simpleinfer_amd import launch
    env = {"SI_LAUNCH_PIN_VISIBLE": "1", "SI_LAUNCH_VISIBLE_LIST": "HIP_VISIBLE_DEVICES=0;ROCR_VISIBLE_DEVICES=0"}
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        code, out = launch.spawn_ranks([sys.executable, CHILD, "gather", str(tmp_path), "3", "0"], 2, timeout=600)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    lines = [ln for ln in out.splitlines() if ln.startswith("{")]
    assert code == 0 and lines, out
    res = json.loads(lines[-1])
    assert res["ok"] == [1, 1] and res["mode"] == "direct"


def test_gathered_tensor_survives_the_next_step_with_default_slots(gpu, tmp_path):
    """include/si_shard.h slot lifetime: with the default 4 slots the gathered tensor handed out after Forward(s) is still
    step s-1's after Forward(s+1) -- a slow consumer on one rank while the others run ahead (3 ranks sharing the device)."""
    from simpleinfer_amd import launch
    code, out = launch.spawn_ranks([sys.executable, CHILD, "slow_consumer", str(tmp_path)], 3, timeout=300)
    lines = [ln for ln in out.splitlines() if ln.startswith("{")]
    assert code == 0 and lines, out
    assert json.loads(lines[-1])["ok"] == [1, 1, 1]


def _bench(*args, env_extra=None, timeout=600):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + [str(a) for a in args], env=env, capture_output=True,
                       text=True, timeout=timeout)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    return p, (json.loads(lines[-1]) if lines else None)


def test_bench_line_single_gpu(gpu):
    p, r = _bench("--steps", 3, "--warmup", 1, "--size", 160, "--batch", 4, "--min-time", 0.2, "--cpu-images", 1, "--cpu-batch", 2)
    assert p.returncode == 0 and r is not None, p.stderr[-3000:]
    assert len([ln for ln in p.stdout.splitlines() if ln.strip()]) == 1          # ONE line on stdout
    assert r["n_gpus"] == 1 and r["steps"] == 3 and r["scaling"] == "weak" and r["dtype"] == "f32" and r["unit"] == "images/sec"
    # (ms_per_step is printed to the microsecond: at this size a step is a few hundred of them)
    assert r["value"] > 0 and abs(r["value"] - 4 * 3 / (r["ms_per_step"] * 3e-3)) / r["value"] < 1e-3 + 0.0006 / r["ms_per_step"]
    w = r["windows"]
    assert w["count"] >= 1 and w["ms_per_step_min"] <= w["ms_per_step_median"] <= w["ms_per_step_max"] and w["timed_s_total"] >= 0.2
    roof = r["roofline"]
    assert roof["bound"] == "mfma" and 0 < roof["frac"] < 1.2 and roof["peak"] == 157.3 and "traffic_source" in roof
    for k in ("cpu_baseline", "cpu_baseline_batched"):
        assert r[k]["kind"] == "port" and r[k]["value"] > 0 and r[k]["cores"] >= 1
    assert r["config"]["gather"] is None


def test_bench_self_launches_two_ranks_sharing_the_device(gpu):
    """`python bench.py --gpus 2` (no launcher): the parent spawns two ranks; with SI_BENCH_SHARE_DEVICE=1 both use device 0
    and the IPC all-gather runs for real (RCCL refuses two ranks on one device).  Strong scaling flag: 8 images over 2."""
    p, r = _bench("--gpus", 2, "--steps", 3, "--warmup", 1, "--size", 160, "--global-batch", 8, "--min-time", 0.2, "--gather", "p2p",
                  env_extra={"SI_BENCH_SHARE_DEVICE": "1"})
    assert p.returncode == 0 and r is not None, p.stderr[-3000:]
    assert r["n_gpus"] == 2 and r["scaling"] == "strong" and r["config"]["per_gpu_batch"] == 4 and r["config"]["global_batch"] == 8
    assert r["config"]["gather"] == "p2p" and r["value"] > 0
    assert r["cpu_baseline"] is None      # rank 0 at N = 1 only


def test_bench_two_ranks_strong_scaling_with_both_transports_in_one_run(gpu):
    """the code path the driver's scaling run takes first -- `bench.py --gpus 2 --global-batch 32 --gather both`
    (strong scaling of the headline batch, the direct fan-out and RCCL's all-gather timed in ONE run) -- end to end on the shared
    device.  Both `gather_ab` entries must be there (RCCL refuses two ranks on one device: its entry is then the structured error,
    on two real devices a timing), `step_bound` says where a step's time went, and the run only returns 0 if every rank's gathered
    buffer held every rank's slab after the warm-up and after each timed region (slab_checksums raises otherwise)."""
    p, r = _bench("--gpus", 2, "--steps", 3, "--warmup", 1, "--size", 160, "--global-batch", 32, "--min-time", 0.3, "--gather", "both",
                  env_extra={"SI_BENCH_SHARE_DEVICE": "1"}, timeout=900)
    assert p.returncode == 0 and r is not None, p.stderr[-3000:]
    assert r["n_gpus"] == 2 and r["scaling"] == "strong" and r["config"]["per_gpu_batch"] == 16 and r["config"]["global_batch"] == 32
    ab = r["gather_ab"]
    assert set(ab) == {"p2p", "rccl"}
    assert ab["p2p"]["value"] > 0 and ab["p2p"]["ms_per_step"] > 0 and ab["p2p"]["gather"]["transport"] == "direct"
    assert abs(ab["p2p"]["value"] - r["value"]) / r["value"] < 1e-6          # `value` is the direct transport's
    for k in ("gather_wait_ms", "gather_wait_copies_ms", "gather_wait_barrier_ms", "peer_copy_ms", "gather_gbps_per_peer"):
        assert k in ab["p2p"]["gather"], k
    assert ("value" in ab["rccl"] and ab["rccl"]["value"] > 0 and ab["rccl"]["gather"]["transport"] == "rccl") or "error" in ab["rccl"]
    assert r["step_bound"] in ("compute", "gather (waiting for the peer copies)", "slowest rank (waiting in the node barrier)")
    assert r["gather"] is not None and r["config"]["gather"] in ("both", "p2p")


def test_bench_needs_as_many_devices_as_ranks(gpu):
    from simpleinfer_amd import device_count
    if device_count() >= 2:
        pytest.skip("two devices present")
    p, r = _bench("--gpus", 2, "--steps", 1, "--warmup", 0, "--size", 64, "--batch", 1)
    assert p.returncode != 0 and r is None
    assert "need 2 HIP devices" in p.stderr
