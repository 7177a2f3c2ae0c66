#!/usr/bin/env python3
"""drift_replay.py FILE.npz -- replay on the CPU oracle the chunk of sweeps in which a replica's incremental energy parted from
its recomputed one (tools/probes/drift_hunt.py wrote the state before the chunk, the GPU's per-sweep series and its positions
after).  Prints, per sweep, accepted moves and energy of GPU and oracle and the oracle's own incremental-minus-recomputed
difference; for the first sweep that differs, the moves of the oracle's trace that the GPU's final positions contradict."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as O


def rng_from_export(st):
    """smcx_rng_export layout: 31 words oldest first + the count of generated-but-unconsumed outputs (the last `left` words).
    The consumer's glibc state is the window `left` steps earlier: r[i-31] = r[i] - r[i-3] (mod 2^32) walks it back."""
    h = [int(v) for v in st[:31]]
    left = int(st[31])
    for _ in range(left):
        h = [(h[30] - h[27]) & 0xFFFFFFFF] + h[:30]
    w = np.array(h + [0], dtype=np.uint32)
    return O.Rng.from_state(w)


d = np.load(sys.argv[1])
N, chunk = int(d["N"]), int(d["chunk"])
s = O.make_sys(N)
R = d["R_before"].copy()
rng = rng_from_export(d["rng_before"])
E = float(d["E"][0])
print("GPU: E at chunk start %.12f (incremental - recomputed before the chunk: %.3e), after %.12f, recomputed after %.12f"
      % (E, float(d["prev_diff"]), float(d["E_last"]), float(d["E_rec"])))
print("oracle total energy of the state before the chunk: %.12f" % O.total_energy(s, R, O.W_FIXTURE))
first = None
for k in range(chunk):
    Rk = R.copy()
    rk = O.Rng(1); rk.g = type(rng.g).from_buffer_copy(rng.g)
    acc, E, tr = O.sweep(s, rng, R, O.W_FIXTURE, 1.1, 1.1, E=E, trace=True)
    rec = O.total_energy(s, R, O.W_FIXTURE)
    same = acc == int(d["jj"][k])
    print("sweep %2d: accepted GPU %5d oracle %5d %s  E GPU %.10f oracle %.10f (diff %.3e)  oracle incr - recomputed %.3e"
          % (k, int(d["jj"][k]), acc, "" if same else "<<<", float(d["E"][k + 1]), E, float(d["E"][k + 1]) - E, E - rec))
    if first is None and (not same or abs(float(d["E"][k + 1]) - E) > 1e-6):
        first = (k, Rk, rk, tr)
print("max |R_gpu_after - R_oracle_after| = %.3e" % np.abs(R - d["R_after"]).max())
if first is not None:
    k, Rk, rk, tr = first
    print("first differing sweep: %d; trace fields: %s" % (k, tr.dtype.names))
    np.savez(os.path.splitext(sys.argv[1])[0] + "_sweep%d_trace.npz" % k, trace=tr, R_start=Rk)
    big = np.argsort(-np.abs(tr["Un"] - tr["Um"]))[:5] if "Un" in tr.dtype.names else []
    for m in big:
        print("   move", {n: tr[n][m] for n in tr.dtype.names})
