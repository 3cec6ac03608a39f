"""Spawn helper for the multi-process tests (gloo rendezvous on 127.0.0.1)."""
import socket

import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def run_ranks(worker, world, *args, attempts=3, timeout=180):
    """Spawn `world` processes running worker(rank, world, port, *args, queue) and collect one queue item per rank.  A rendezvous
    can fail for reasons outside the code under test (the probed port taken in between, a slow fork): one retry on a new port."""
    last = None
    for _ in range(attempts):
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=worker, args=(r, world, port, *args, q)) for r in range(world)]
        for p in procs:
            p.start()
        try:
            res = [q.get(timeout=timeout) for _ in procs]
            for p in procs:
                p.join(timeout=60)
            if all(p.exitcode == 0 for p in procs):
                return res
            last = RuntimeError(f"exit codes {[p.exitcode for p in procs]}")
        except Exception as ex:          # queue.Empty: a rank died or hung
            last = ex
        for p in procs:
            if p.is_alive():
                p.terminate()
            p.join(timeout=30)
    raise last


