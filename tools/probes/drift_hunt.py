#!/usr/bin/env python3
"""drift_hunt.py [SWEEPS [CHUNK [NREP [N NA NZ]]]] -- where does the incremental energy (SMC.c:340-341) part from the recomputed one?

Round 5: after 500 sweeps of the headline launch one replica in thousands carries |E_incremental - E_recomputed| of 0.2 where
rounding explains 1e-11.  This probe runs the launch in chunks of CHUNK sweeps; before every chunk it keeps all positions and
rand() states on the host, after it compares E_last with total_energy() per replica, and for every replica whose difference
CHANGED in the chunk it saves the state before the chunk, the per-sweep series of the chunk and the positions after it to
gpurun_out/r05_drift_<replica>_<chunk>.npz -- material for a replay against the oracle on the CPU (tools/probes/drift_replay.py).
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import smcx_loader

S = smcx_loader.load()
a = [int(v) for v in sys.argv[1:]]
sweeps, chunk, nrep = (a + [500, 10, 4096])[:3] if len(a) < 3 else a[:3]
N, Na, Nz = a[3:6] if len(a) >= 6 else (4096, 8, 16)
p = S.default_params(N, nrep, flags=S.FLAGS_REFERENCE | S.FLAG_SERIES)
saved = 0
with S.Engine(p) as eng:
    print(json.dumps({"kernel": eng.kernel_form[1], "N": N, "replicas": nrep, "sweeps": sweeps, "chunk": chunk}), flush=True)
    eng.upload(S.fcc_init(Na, Nz), S.W_REFERENCE)
    prev = np.zeros(nrep)
    for c in range(sweeps // chunk):
        st0, R0 = eng.rng_export(), eng.positions()
        eng.run(0, chunk, chunk)
        ob = eng.observables()
        Erec = eng.total_energy()
        d = ob["E_last"] - Erec
        new = np.nonzero(np.abs(d - prev) > 1e-7)[0]
        print(json.dumps({"chunk": c, "sweeps_done": (c + 1) * chunk, "max_abs_diff": float(np.abs(d).max()),
                          "replicas_over_1e-7": int((np.abs(d) > 1e-7).sum()), "changed_in_this_chunk": [int(r) for r in new[:20]],
                          "their_change": [float(x) for x in (d - prev)[new[:20]]]}), flush=True)
        if len(new) and saved < 6:
            E, jj = eng.series(chunk)
            R1 = eng.positions()
            for r in new[:3]:
                if saved >= 6:
                    break
                np.savez(os.path.join(ROOT, "gpurun_out", "r05_drift_%d_%d.npz" % (r, c)), R_before=R0[r], rng_before=st0[r], R_after=R1[r],
                         E=E[r], jj=jj[r], E_last=ob["E_last"][r], E_rec=Erec[r], prev_diff=prev[r], N=N, chunk=chunk)
                saved += 1
        prev = d
