#!/usr/bin/env python3
"""drift_flip_scan.py FILE.npz [moves...] -- which single Metropolis decision, flipped, turns the oracle's sweep 0 of the chunk
into the GPU's (accepted count and energy after the sweep)?  A Python sweep from the oracle's primitives (O.eval_move, the
rand() stream), checked against the C oracle first; then every candidate move (default: all) is re-run with its decision forced
the other way."""
import os
import sys
from multiprocessing import Pool

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools", "probes"))
import numpy as np
import oracle_lib as O

A = T = 1.1
RAND_MAX = 2147483647.0


def setup(path):
    d = np.load(path)
    N = int(d["N"])
    s = O.make_sys(N)
    h = [int(v) for v in d["rng_before"][:31]]
    for _ in range(int(d["rng_before"][31])):
        h = [(h[30] - h[27]) & 0xFFFFFFFF] + h[:30]
    rng = O.Rng.from_state(np.array(h + [0], dtype=np.uint32))
    displ = rng.box_muller(np.sqrt(2 * A), 3 * N)
    offset = rng.rand()
    u = np.array([rng.rand() for _ in range(N)]) / RAND_MAX
    return d, s, N, displ, offset, u


def sweep(s, N, R, displ, offset, u, E, start=0, force=None, upto=None):
    """moves nn = start..N-1 of the sweep (SMC.c:292-347) on R in place; force = (nn, decision)"""
    acc = 0
    for nn in range(start, N if upto is None else upto):
        n = (nn + offset) % N
        Um, Fm, _, _ = O.eval_move(s, R, O.W_FIXTURE, n, R[3 * n:3 * n + 3])
        Fm = np.array(Fm)
        dl = Fm * A / T + displ[3 * n:3 * n + 3]
        q = R[3 * n:3 * n + 3] + dl
        q[0] -= s.L * np.rint(q[0] / s.L); q[1] -= s.L * np.rint(q[1] / s.L)
        _, _, Un, Fn = O.eval_move(s, R, O.W_FIXTURE, n, q)
        Fn = np.array(Fn)
        g = Fn - Fm
        dW = (g @ g + 2 * (g @ Fm)) * A / (4 * T)
        with np.errstate(over="ignore", invalid="ignore"):
            ap = np.exp(-(Un - Um + dl @ (Fn + Fm) / 2 + dW) / T)
        ok = bool(u[nn] < ap)
        if force is not None and force[0] == nn:
            ok = force[1]
        if ok:
            R[3 * n:3 * n + 3] = q
            acc += 1
            E += Un - Um
    return acc, E


def one(args):
    path, nn = args
    d, s, N, displ, offset, u = setup(path)
    tr = np.load(os.path.splitext(path)[0] + "_sweep0_trace.npz")["trace"]
    R = d["R_before"].copy()
    E = float(d["E"][0])
    acc = 0
    for m in range(nn):                       # the moves before nn are the oracle's own
        if tr["accepted"][m]:
            n = tr["n"][m]
            R[3 * n:3 * n + 3] = tr["prop"][m]
            E += tr["Un"][m] - tr["Um"][m]
            acc += 1
    a2, E2 = sweep(s, N, R, displ, offset, u, E, start=nn, force=(nn, not bool(tr["accepted"][nn])))
    return nn, acc + a2, E2


if __name__ == "__main__":
    path = sys.argv[1]
    d, s, N, displ, offset, u = setup(path)
    tr = np.load(os.path.splitext(path)[0] + "_sweep0_trace.npz")["trace"]
    assert [(nn + offset) % N for nn in range(8)] == [int(x) for x in tr["n"][:8]], "visiting order"
    R = d["R_before"].copy()
    acc, E = sweep(s, N, R, displ, offset, u, float(d["E"][0]), upto=200)
    ref = sum(int(tr["accepted"][m]) for m in range(200))
    print("python sweep against the C oracle over the first 200 moves: accepted %d / %d" % (acc, ref))
    gpu_acc, gpu_E = int(d["jj"][0]), float(d["E"][1])
    print("GPU after sweep 0: accepted %d, E %.10f" % (gpu_acc, gpu_E))
    cand = [int(v) for v in sys.argv[2:]] or list(range(N))
    with Pool(8) as pool:
        for nn, a2, E2 in pool.imap_unordered(one, [(path, nn) for nn in cand], chunksize=8):
            hit = a2 == gpu_acc and abs(E2 - gpu_E) < 1e-6
            if hit or len(cand) < 50:
                print("flip move %d (particle %d, oracle decision %d): accepted %d, E %.10f %s" % (nn, tr["n"][nn], tr["accepted"][nn], a2, E2, "<<< the GPU's sweep" if hit else ""), flush=True)
