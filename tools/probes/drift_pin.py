#!/usr/bin/env python3
"""drift_pin.py FILE.npz... -- (on the GPU box) one sweep of the sweep kernel from the saved state and rand() state of a replica
whose incremental energy parted from the recomputed one, against the oracle's sweep with its per-move trace: the first particle
in visiting order whose position after the sweep differs, what the oracle did with it, and -- if the GPU moved it elsewhere --
the force difference that explains the other proposal (delta = Fm A/T + displacement, SMC.c:307-309)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools", "probes"))
import numpy as np
import smcx_loader
import oracle_lib as O

S = smcx_loader.load()
A = T = 1.1
for path in sys.argv[1:]:
    d = np.load(path)
    N = int(d["N"])
    s = O.make_sys(N)
    st = d["rng_before"].copy()
    p = S.default_params(N, 1, flags=S.FLAGS_REFERENCE | S.FLAG_SERIES, tune_kernel=int(os.environ.get("SMCX_PIN_KERNEL", "0")))
    with S.Engine(p) as eng:
        eng.upload(d["R_before"], S.W_REFERENCE)
        eng.rng_import(st[None, :])
        E0 = eng.total_energy()[0]
        eng.run(0, 1, 1)
        E, jj = eng.series(1)
        Rg = eng.positions()[0].reshape(-1, 3)
        Erec = eng.total_energy()[0]
        kname = eng.kernel_form[1]
    h = [int(v) for v in st[:31]]
    for _ in range(int(st[31])):
        h = [(h[30] - h[27]) & 0xFFFFFFFF] + h[:30]
    rng = O.Rng.from_state(np.array(h + [0], dtype=np.uint32))
    R = d["R_before"].copy()
    acc, Eo, tr = O.sweep(s, rng, R, O.W_FIXTURE, A, T, E=float(E0), trace=True)
    Ro = R.reshape(-1, 3)
    R0 = d["R_before"].reshape(-1, 3)
    print(json.dumps({"file": os.path.basename(path), "kernel": kname, "gpu_accepted": int(jj[0][0]), "oracle_accepted": acc,
                      "gpu_E_after": float(E[0][1]), "oracle_E_after": Eo, "gpu_E_recomputed": float(Erec),
                      "gpu_incr_minus_recomputed": float(E[0][1] - Erec)}))
    shown = 0
    for m in range(N):
        n = int(tr["n"][m])
        dev = np.abs(Rg[n] - Ro[n]).max()
        if dev > 1e-9:
            t = tr[m]
            moved_gpu = np.abs(Rg[n] - R0[n]).max() > 0
            out = {"move": m, "particle": n, "z_before": float(R0[n, 2]), "oracle": {"accepted": int(t["accepted"]), "Um": float(t["Um"]),
                   "Un": float(t["Un"]), "ap": float(t["ap"]), "u": float(t["u"]), "prop": [float(x) for x in t["prop"]],
                   "Fm": [float(x) for x in t["Fm"]]},
                   "gpu_position_after": [float(x) for x in Rg[n]], "gpu_moved_it": bool(moved_gpu)}
            if moved_gpu and t["accepted"]:
                dq = Rg[n] - t["prop"]
                dq[0] -= 33.0 * np.rint(dq[0] / 33.0); dq[1] -= 33.0 * np.rint(dq[1] / 33.0)
                out["force_difference_gpu_minus_oracle"] = [float(x) for x in dq * T / A]
            print(json.dumps(out))
            shown += 1
            if shown >= 4:
                break
