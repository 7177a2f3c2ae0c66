"""Replica sharding across GPUs and the one exchange step of the path.

The reference's intended MPI fan-out is independent replica chains that share
R0 and W and differ only in their seed (SMC.c:40, 43, 66-95): no communication
while sampling.  Here one process drives one GPU (torch.distributed, backend
"nccl" = RCCL over xGMI on ROCm, "gloo" in the CPU tests); replicas are dealt
out in contiguous blocks, seeds follow the GLOBAL replica index so results do
not depend on the GPU count, and the only collective is the final all-gather
of the per-replica observable records plus a sum of the wall-normal profile.
"""
import numpy as np

OBS_RECORD_DOUBLES = 8
OBS_FIELDS = ("accepted", "nsamp", "sumE", "sumE2", "E_last", "therm_accepted", "gathers", "oob")


def shard(nrep_total, rank, world):
    """contiguous block of replicas for `rank`: (first_replica, count)"""
    if world < 1 or not (0 <= rank < world) or nrep_total < 0:
        raise ValueError("bad shard request")
    base, extra = divmod(nrep_total, world)
    count = base + (1 if rank < extra else 0)
    first = rank * base + min(rank, extra)
    return first, count


def unpack(packed, nrep, Ncz):
    """packed = smcx_export_observables_device layout: [nrep][8] records then [nrep][Ncz]"""
    packed = np.asarray(packed, dtype=np.float64)
    rec = packed[:nrep * OBS_RECORD_DOUBLES].reshape(nrep, OBS_RECORD_DOUBLES)
    zh = packed[nrep * OBS_RECORD_DOUBLES:].reshape(nrep, Ncz)
    out = {k: rec[:, i].copy() for i, k in enumerate(OBS_FIELDS)}
    out["zhist"] = zh.copy()
    return out


def summarise(obs, N, maxsteps):
    """ensemble observables from gathered per-replica records (SMC.c:244-248)"""
    n = np.maximum(obs["nsamp"], 1.0)
    meanE = obs["sumE"] / n
    acc = (obs["accepted"] / max(maxsteps, 1)) / N
    prof = obs["zhist"].sum(axis=0)
    g = obs["gathers"].sum()
    return dict(meanE=meanE, acceptance_ratio=acc, zprofile=prof / max(g, 1.0),
                mean_of_meanE=float(meanE.mean()), mean_acceptance=float(acc.mean()))


def gather_observables(local_packed, nrep_local, Ncz, group=None):
    """All-gather the packed observable block of every rank.

    local_packed: 1-D float64 torch tensor (on the GPU for nccl/RCCL, CPU for
    gloo) of length nrep_local*(8+Ncz).  Ranks may hold different replica
    counts.  Returns per-replica arrays in global replica order (numpy).
    """
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return unpack(local_packed.detach().cpu().numpy(), nrep_local, Ncz)
    counts = [torch.zeros(1, dtype=torch.int64, device=local_packed.device) for _ in range(world)]
    dist.all_gather(counts, torch.tensor([nrep_local], dtype=torch.int64,
                                         device=local_packed.device), group=group)
    counts = [int(c.item()) for c in counts]
    width = OBS_RECORD_DOUBLES + Ncz
    mx = max(counts)
    pad = torch.zeros(mx * width, dtype=torch.float64, device=local_packed.device)
    pad[:nrep_local * width] = local_packed
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad, group=group)
    parts = [unpack(b[:c * width].cpu().numpy(), c, Ncz) for b, c in zip(bufs, counts)]
    out = {k: np.concatenate([p[k] for p in parts], axis=0) for k in parts[0]}
    return out


def spawn_ranks(script, argv, n, extra_env=None, timeout=None):
    """Start `n` ranks of `script` (one process per GPU: RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* set,
    rendezvous on 127.0.0.1) as FRESH child processes and wait for them.  The caller must not have
    touched the GPU: a rank initialises its device itself.  Rank 0's stdout is passed through;
    if any rank fails the others are stopped (by PID) and the first non-zero code is returned."""
    import os
    import socket
    import subprocess
    import sys
    import time
    if n < 1:
        raise ValueError("need at least one rank")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if extra_env:
            env.update(extra_env)
        procs.append(subprocess.Popen([sys.executable, script] + list(argv), env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    t0 = time.time()
    rc = 0
    live = list(procs)
    while live:
        for p in list(live):
            code = p.poll()
            if code is None:
                continue
            live.remove(p)
            if code != 0 and rc == 0:
                rc = code
        if rc != 0 or (timeout and time.time() - t0 > timeout):
            for p in live:   # a rank died or the run overran: stop the rest, they would wait for it forever
                p.terminate()
            for p in live:
                try:
                    p.wait(20)
                except subprocess.TimeoutExpired:
                    p.kill()
            if rc == 0:
                rc = 124
            break
        time.sleep(0.05)
    return rc
