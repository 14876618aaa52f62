"""CPU: the N > 1 path (batch sharding + one all-gather of output slabs) on 2 gloo ranks.

The compute inside each rank is the CPU oracle (the product has no CPU path); what is under test is
simpleinfer_amd/distributed.py -- the code bench.py runs on RCCL -- and the sharding contract: rank r's slab is
rows [r*B/G, (r+1)*B/G) of the full-batch result, bit for bit."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, tmpdir, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import torch
    from oracle import orc
    from simpleinfer_amd import distributed as sd, modelgen as mg
    orc.lib().orc_set_num_threads(2)
    r, w, _ = sd.env_rank_world()
    dist = sd.init_process_group("gloo")
    B = 4
    lo, hi = sd.shard_range(B, r, w)
    x_full = mg.synth_input((B, 64, 64, 3))
    pp, bp = os.path.join(tmpdir, "r%d.param" % r), os.path.join(tmpdir, "r%d.bin" % r)
    mg.build_toy_yolo(hi - lo, 64).save(pp, bp)
    out = orc.run_graph(pp, bp, {"0": x_full[lo:hi]})
    (name, local), = out.items()
    gathered = sd.all_gather_slabs(torch.from_numpy(local))
    # the pipelined exchange bench.py uses for N > 1: three "steps" whose outputs differ, every one of them gathered
    og = sd.OverlappedGather(torch.from_numpy(local))
    buf = torch.from_numpy(local.copy())
    seen = []
    for step in range(3):
        buf.copy_(torch.from_numpy(local) + float(step))     # the engine overwrites its output buffer every Forward
        og.submit(buf)
        buf.fill_(-1.0)                                       # ... and may do so as soon as submit() returns
        og.drain()
        seen.append(og.latest().clone())
    # zero-copy variant: the producer writes straight into the buffer the collective reads (Engine::Output binding)
    for step in range(3, 6):
        og.target().copy_(torch.from_numpy(local) + float(step))
        og.submit_inplace()
        og.drain()
        seen.append(og.latest().clone())
    dist.barrier()
    if r == 0:
        q.put((name, gathered.numpy(), [t.numpy() for t in seen]))
    dist.destroy_process_group()


def test_two_rank_gloo_sharding_matches_full_batch(tmp_path):
    import torch.multiprocessing as mp
    from oracle import orc
    from simpleinfer_amd import modelgen as mg
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, str(tmp_path), q)) for r in range(2)]
    for p in procs:
        p.start()
    name, gathered, pipelined = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    pp, bp = str(tmp_path / "full.param"), str(tmp_path / "full.bin")
    mg.build_toy_yolo(4, 64).save(pp, bp)
    full = orc.run_graph(pp, bp, {"0": mg.synth_input((4, 64, 64, 3))})[name]
    assert gathered.shape == full.shape
    assert np.array_equal(gathered, full)
    for step, g in enumerate(pipelined):
        assert np.array_equal(g, full + np.float32(step)), "overlapped gather, step %d" % step


def test_shard_range():
    from simpleinfer_amd import distributed as sd
    assert [sd.shard_range(256, r, 8) for r in (0, 7)] == [(0, 32), (224, 256)]
    assert sd.shard_range(32, 0, 1) == (0, 32)
    with pytest.raises(ValueError):
        sd.shard_range(30, 0, 8)
