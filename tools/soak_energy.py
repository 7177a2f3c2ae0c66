#!/usr/bin/env python3
"""soak_energy.py [--quick] -- the invariant that found round 5's z-sort bug, over every kernel family and many sweeps:
the energy carried incrementally along the chain (SMC.c:340-341: E += Un - Um per accepted move) against the energy recomputed
from the positions (SMC.c:626-646, 822-859).  No chaos enters -- both describe the same state -- so any difference beyond
rounding (1e-9 relative is the tests' bound; 1e-12 is typical) is a wrong increment: a dropped or phantom pair, a stale
position, a wrong wall term.  Each case runs `sweeps` sweeps in chunks and prints the largest relative difference seen after any
chunk and the replicas beyond 1e-9.  For a GPU box (tools/sessions/r05_session10.sh); a few minutes."""
import json
import sys
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import smcx_loader

S = smcx_loader.load()
quick = "--quick" in sys.argv
# (label, N, replicas, lattice, box L, sweeps, chunk, extra params, wanted kernel)
CASES = [
    ("config 3: N=4096 x 4096, into the walls", 4096, 4096, (8, 16), 33.0, 3000, 100, {}, "mc64"),
    ("config 2: N=1024 x 1024", 1024, 1024, (8, 4), 33.0, 4000, 200, {}, "ml16"),
    ("N=1000 x 4096 (ragged)", 1000, 4096, (8, 4), 33.0, 3000, 200, {}, "mc16"),
    ("N=2048 x 2048", 2048, 2048, (8, 8), 33.0, 3000, 200, {}, "mc32"),
    ("N=4096 x 1024 through mb64", 4096, 1024, (8, 16), 33.0, 1500, 100, {"tune_kernel": S.KERNEL_MB}, "mb64"),
    ("N=6144 x 512, four wavefronts", 6144, 512, (16, 6), 33.0, 1000, 100, {}, "mc32x4"),
    ("config 5 share: N=16384 x 256, two teams", 16384, 256, (16, 16), 33.0, 600, 50, {}, "mt64x8"),
    ("N=16384 x 256, 4 wavefronts", 16384, 256, (16, 16), 33.0, 300, 50, {"tune_slots": 64, "tune_waves": 4}, "mc64x4"),
    ("N=16384 x 256, 8 wavefronts", 16384, 256, (16, 16), 33.0, 300, 50, {"tune_slots": 32, "tune_waves": 8}, "mc32x8"),
    ("N=4096 x 1024 in a small box (L=20: dense gas, many pairs)", 4096, 1024, (8, 16), 20.0, 1000, 100, {}, "mc64"),
    ("N=4096 x 1024 without walls", 4096, 1024, (8, 16), 33.0, 1500, 100, {"flags": S.FLAG_E0_RESTART}, "mc64"),
    ("N=4100 replicas x 4096: windows of units", 4096, 4100, (8, 16), 33.0, 600, 100, {}, "mc64"),
]
worst_all = 0.0
for label, N, nrep, lat, L, sweeps, chunk, extra, want in CASES:
    if quick:
        sweeps = max(chunk, sweeps // 5)
    p = S.default_params(N, nrep, L=L, **extra)
    R0 = S.fcc_init(lat[0], lat[1], L=L)[:3 * N]
    with S.Engine(p) as eng:
        name = eng.kernel_form[1].replace("smcx::sweep_kernel_", "")
        eng.upload(R0, S.W_REFERENCE)
        worst, bad, acc = 0.0, set(), 0.0
        for c in range(sweeps // chunk):
            eng.run(0, chunk, chunk)
            ob = eng.observables()
            Erec = eng.total_energy()
            d = np.abs(ob["E_last"] - Erec) / (1.0 + np.abs(Erec))
            worst = max(worst, float(d.max()))
            bad.update(int(r) for r in np.nonzero(d > 1e-9)[0])
            acc = float(ob["acceptance_ratio"].mean())
        zh = ob["zhist"].sum(axis=0).astype(float)
        occ = np.nonzero(zh > 0)[0]
    worst_all = max(worst_all, worst)
    print(json.dumps({"case": label, "kernel": name, "expected_kernel": want, "sweeps": (sweeps // chunk) * chunk,
                      "replica_sweeps": nrep * (sweeps // chunk) * chunk, "acceptance_last_chunk": round(acc, 4),
                      "z_bins_occupied": [int(occ[0]), int(occ[-1])] if len(occ) else None,
                      "max_relative_incremental_minus_recomputed": worst, "replicas_beyond_1e-9": sorted(bad)[:10],
                      "count_beyond_1e-9": len(bad)}), flush=True)
print(json.dumps({"worst_of_all_cases": worst_all, "ok": worst_all < 1e-9}))
