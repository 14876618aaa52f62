"""Child process of the multi-rank tests (started by simpleinfer_amd.launch.spawn_ranks with RANK / WORLD_SIZE / MASTER_* set).
Modes:
  group   CPU: node group over POSIX shm -- barriers, byte all-gather, max reduction
  gloo    CPU: the launcher's env is what torch.distributed expects (gloo all_reduce)
  fail    CPU: rank 1 exits 3 while rank 0 sleeps -- the launcher must stop rank 0 and report failure
  dead_group CPU: a timed-out barrier poisons the group for every rank
  slow_consumer GPU: the gathered tensor survives the next step with the default 4 slots
  gather  GPU: ShardedForward over the direct IPC all-gather; ranks may share one device (hipIpc works within a device)
Rank 0 prints one JSON line."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    mode = sys.argv[1]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    if mode == "fail":
        if rank == 1:
            sys.exit(3)
        time.sleep(60)
        return
    if mode == "gloo":
        import torch
        import torch.distributed as dist
        dist.init_process_group("gloo")
        t = torch.tensor([float(rank + 1)])
        dist.all_reduce(t)
        dist.barrier()
        if rank == 0:
            print(json.dumps({"sum": float(t.item()), "world": dist.get_world_size()}))
        dist.destroy_process_group()
        return
    from simpleinfer_amd import shard
    if mode == "dead_group":
        # a barrier that times out kills the group: the late rank's barrier and every later one fail (no release with a
        # stale arrival count), on every rank
        g = shard.NodeGroup(shard.default_group_name() + "_dead", rank, world, timeout_s=1.0)
        codes = []
        if rank == 1:
            time.sleep(2.5)
        for _ in range(3):
            try:
                g.barrier()
                codes.append(0)
            except shard.ShardError as ex:
                codes.append(ex.code)
        sys.stderr.write("rank %d codes %s\n" % (rank, codes))
        with open(os.path.join(sys.argv[2], "codes%d.json" % rank), "w") as f:
            json.dump(codes, f)
        g.close()
        return
    g = shard.NodeGroup(shard.default_group_name() + "_" + mode, rank, world, timeout_s=30.0)
    if mode == "group":
        seen = []
        for it in range(50):
            parts = g.allgather_bytes(bytes([rank, it % 251]) * 8)
            seen.append(all(p == bytes([r, it % 251]) * 8 for r, p in enumerate(parts)))
            g.barrier()
        mx = g.max_f64(float(rank) * 1.5)
        big = g.allgather_bytes(bytes([rank]) * 4096)
        if rank == 0:
            print(json.dumps({"ok": all(seen), "max": mx, "big": [b[0] for b in big], "n": len(seen)}))
        g.close()
        return
    if mode == "gather":
        import numpy as np
        import simpleinfer_amd as si
        from simpleinfer_amd import hipops, _native
        H = _native.hip()
        dev = rank % si.device_count()
        H.si_hip_set_device(dev)
        td = sys.argv[2]
        per, size, steps, slots = 2, 64, 5, int(sys.argv[3])
        mg = si.modelgen
        pp, bp = os.path.join(td, "m%d.param" % rank), os.path.join(td, "m%d.bin" % rank)
        mg.build_toy_yolo(per, size).save(pp, bp)
        e = si.Engine(device=dev, outputs_to_host=0, graph=int(sys.argv[4]))
        e.load_model(pp, bp)
        oname = e.output_names()[0]
        gmode = sys.argv[5] if len(sys.argv) > 5 else "direct"
        sf = shard.ShardedForward(e, oname, g, dev, slots=slots, mode=gmode)
        ref = si.Engine(device=dev)                      # an independent engine computes every rank's slab for the check
        ref.load_model(pp, bp)
        ok = True
        for step in range(steps):
            xs = [mg.synth_input((per, size, size, 3), seed=100 * step + r) for r in range(world)]
            e.input("0", xs[rank])
            sf.forward()
            if step > 0 and slots > 1:                   # the previous step's gather has completed
                want = []
                for r in range(world):
                    ref.input("0", mg.synth_input((per, size, size, 3), seed=100 * (step - 1) + r))
                    ref.forward()
                    want.append(ref.extract(oname).copy())
                got = hipops.DeviceBuffer.view(sf.gathered_ptr(), int(np.prod(sf.gathered_shape())) * 4).to_numpy(sf.gathered_shape())
                ok = ok and np.array_equal(got, np.concatenate(want, 0))
        sf.flush()
        want = []
        for r in range(world):
            ref.input("0", mg.synth_input((per, size, size, 3), seed=100 * (steps - 1) + r))
            ref.forward()
            want.append(ref.extract(oname).copy())
        got = hipops.DeviceBuffer.view(sf.gathered_ptr(), int(np.prod(sf.gathered_shape())) * 4).to_numpy(sf.gathered_shape())
        ok = ok and np.array_equal(got, np.concatenate(want, 0))
        oks = g.allgather_bytes(bytes([1 if ok else 0]))
        sf.close()
        if rank == 0:
            print(json.dumps({"ok": [b[0] for b in oks], "shape": list(sf.gathered_shape()), "mode": sf.gather.mode}))
        g.close()
        return
    if mode == "slow_consumer":
        # The slot-lifetime contract of include/si_shard.h with the default 4 slots: the gathered tensor of step s-1 that is
        # available after Forward(s) stays intact through the whole NEXT Forward(s+1) -- here rank 0 is a slow consumer that
        # only looks at it after that next step (while the other ranks have raced ahead as far as the barriers let them).
        import numpy as np
        import simpleinfer_amd as si
        from simpleinfer_amd import hipops, _native
        H = _native.hip()
        dev = rank % si.device_count()
        H.si_hip_set_device(dev)
        td = sys.argv[2]
        per, size, steps = 2, 64, 7
        mg = si.modelgen
        pp, bp = os.path.join(td, "m%d.param" % rank), os.path.join(td, "m%d.bin" % rank)
        mg.build_toy_yolo(per, size).save(pp, bp)
        e = si.Engine(device=dev, outputs_to_host=0)
        e.load_model(pp, bp)
        oname = e.output_names()[0]
        sf = shard.ShardedForward(e, oname, g, dev)          # default slots
        assert sf.gather.slots == 4
        ref = si.Engine(device=dev)
        ref.load_model(pp, bp)
        n = int(np.prod(sf.gathered_shape()))
        ok = True
        held = None                                          # (pointer, step) of the gathered tensor handed out one step ago
        for step in range(steps):
            e.input("0", mg.synth_input((per, size, size, 3), seed=100 * step + rank))
            sf.forward()
            if rank == 0:
                time.sleep(0.05)                             # the others are already inside the next step
            if held is not None:
                ptr, hs = held
                want = []
                for r in range(world):
                    ref.input("0", mg.synth_input((per, size, size, 3), seed=100 * hs + r))
                    ref.forward()
                    want.append(ref.extract(oname).copy())
                got = hipops.DeviceBuffer.view(ptr, n * 4).to_numpy(sf.gathered_shape())
                ok = ok and np.array_equal(got, np.concatenate(want, 0))
            held = (sf.gathered_ptr(), step - 1) if step > 0 else None
        sf.flush()
        oks = g.allgather_bytes(bytes([1 if ok else 0]))
        sf.close()
        if rank == 0:
            print(json.dumps({"ok": [b[0] for b in oks], "slots": 4}))
        g.close()
        return
    raise SystemExit("unknown mode " + mode)


if __name__ == "__main__":
    main()
