"""One process per GPU without an external launcher.

``spawn_ranks(cmd, n)`` starts `n` CHILD processes of `cmd`, each with RANK / LOCAL_RANK / WORLD_SIZE /
MASTER_ADDR / MASTER_PORT in its environment (what ``torch.distributed.run`` would set), relays rank 0's stdout and
returns the worst exit code.  The parent never touches the GPU: on this pool a process that has initialised HIP must
not exec or re-launch itself, so the decision to fan out is taken before anything imports the native libraries.
Children that outlive a failed sibling are terminated by PID (never by pattern).
"""
from __future__ import annotations

import os
import socket
import subprocess
import sys
import time
from typing import List, Optional, Sequence, Tuple


def free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launched_by_a_launcher() -> bool:
    """True when RANK / WORLD_SIZE are already in the environment (torch.distributed.run or spawn_ranks)."""
    return "RANK" in os.environ and "WORLD_SIZE" in os.environ


def rank_env(rank: int, world: int, port: int, base: Optional[dict] = None) -> dict:
    env = dict(os.environ if base is None else base)
    env.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world),
               MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), SI_SPAWNED="1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC only on this pool (RCCL, hipIpc*)
    # SI_LAUNCH_PIN_VISIBLE=1: every rank sees ONLY its own GPU (as container runtimes and some launchers arrange it), so every
    # rank calls its GPU "device 0" and peers can only be named by PCI bus id (include/si_shard.h).  SI_LAUNCH_VISIBLE_LIST, a
    # ';'-separated list of VAR=value with one entry per rank, gives the settings instead (tests: the same device hidden behind
    # two different mechanisms, HIP_VISIBLE_DEVICES for one rank and ROCR_VISIBLE_DEVICES for the other).
    if env.get("SI_LAUNCH_PIN_VISIBLE") == "1":
        names = [v for v in env.get("SI_LAUNCH_VISIBLE_LIST", "").split(";") if "=" in v]
        var, val = names[rank].split("=", 1) if rank < len(names) else ("HIP_VISIBLE_DEVICES", str(rank))
        env[var] = val
        env["LOCAL_RANK"] = "0"
        env["LOCAL_WORLD_SIZE"] = "1"
    return env


def spawn_ranks(cmd: Sequence[str], n: int, timeout: Optional[float] = None, poll: float = 0.05) -> Tuple[int, str]:
    """Run `cmd` as ranks 0..n-1.  Returns (exit code, rank 0's stdout).  Exit code is 0 only when every rank exited 0;
    the first failure terminates the remaining ranks."""
    port = free_port()
    procs: List[subprocess.Popen] = []
    for r in range(n):
        procs.append(subprocess.Popen(list(cmd), env=rank_env(r, n, port),
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, stderr=None))
    t0 = time.time()
    code = 0
    out0 = b""
    live = set(range(n))
    try:
        import selectors
        sel = selectors.DefaultSelector()
        sel.register(procs[0].stdout, selectors.EVENT_READ)
        eof = False
        while live:
            if not eof:
                for key, _ in sel.select(timeout=poll):
                    chunk = os.read(key.fileobj.fileno(), 65536)
                    if chunk:
                        out0 += chunk
                    else:
                        eof = True
                        sel.unregister(key.fileobj)
            else:
                time.sleep(poll)
            for r in sorted(live):
                rc = procs[r].poll()
                if rc is None:
                    continue
                live.discard(r)
                if rc != 0 and code == 0:
                    code = rc if rc > 0 else 128 - rc
                    print("launch: rank %d exited with %d; stopping the other ranks" % (r, rc), file=sys.stderr)
            if code != 0 or (timeout is not None and time.time() - t0 > timeout):
                if code == 0:
                    code = 124
                    print("launch: timeout after %.0f s" % (time.time() - t0), file=sys.stderr)
                break
        if not eof and procs[0].stdout is not None:
            try:
                rest = procs[0].stdout.read() if not live or 0 not in live else b""
                out0 += rest or b""
            except Exception:
                pass
    finally:
        for r in sorted(live):
            p = procs[r]
            if p.poll() is None:
                p.terminate()
        for r in sorted(live):
            try:
                procs[r].wait(timeout=10)
            except subprocess.TimeoutExpired:
                procs[r].kill()
                procs[r].wait()
    return code, out0.decode("utf-8", "replace")
