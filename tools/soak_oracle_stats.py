#!/usr/bin/env python3
"""soak_oracle_stats.py [REPLICAS [SWEEPS]] -- the headline kernel against the ORACLE far beyond the chaos horizon, through the
regime a production run sits in (the slab evaporating, particles at the walls): REPLICAS chains of N = 4096 (default 64) over SWEEPS
sweeps (default 400) on the GPU and, with the same seeds, through the CPU oracle (pinned bit for bit on the reference's
oneParticleMoves; one chain per host core, ~0.2 s per sweep).  Trajectories separate after ~10 sweeps, statistics must not:
ensemble mean of the energy at every 50th sweep, of the accepted moves, and the wall-normal profile, each with the difference in
units of its standard error.  For a GPU box (the oracle leg takes REPLICAS x SWEEPS x 0.2 s / cores)."""
import json
import os
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import smcx_loader
import oracle_lib as O          # the checker

S = smcx_loader.load()
nrep = int(sys.argv[1]) if len(sys.argv) > 1 else 64
nsw = int(sys.argv[2]) if len(sys.argv) > 2 else 400
N, gl = 4096, 10
R0 = O.fcc(8, 16)
p = S.default_params(N, nrep, flags=S.FLAGS_REFERENCE | S.FLAG_SERIES)
with S.Engine(p) as eng:
    kname = eng.kernel_form[1]
    eng.upload(R0, O.W_FIXTURE)
    eng.run(0, nsw, gl)
    E, jj = eng.series(nsw)
    zg = eng.observables()["zhist"].sum(axis=0).astype(float)
s = O.make_sys(N)


def one(r):
    ref = O.chain(s, 12345 + r, R0, O.W_FIXTURE, 1.1, 1.1, 0, nsw, gl)
    return ref["E"].copy(), ref["jj"].astype(float), ref["zhist"].astype(float)


with ThreadPoolExecutor(len(os.sched_getaffinity(0))) as ex:
    refs = list(ex.map(one, range(nrep)))
Eo = np.stack([r[0] for r in refs]); jo = np.stack([r[1] for r in refs]); zo = np.sum([r[2] for r in refs], axis=0)
print(json.dumps({"kernel": kname, "replicas": nrep, "sweeps": nsw, "same_chains_over_the_first_sweeps": bool(np.array_equal(jj[:, :5], jo[:, :5].astype(jj.dtype)))}))
worst = 0.0
for k in list(range(50, nsw + 1, 50)):
    a, b = E[:, k], Eo[:, k]
    se = np.sqrt(a.var(ddof=1) / nrep + b.var(ddof=1) / nrep)
    ja, jb = jj[:, max(0, k - 50):k].sum(axis=1).astype(float), jo[:, max(0, k - 50):k].sum(axis=1)
    sj = np.sqrt(ja.var(ddof=1) / nrep + jb.var(ddof=1) / nrep)
    z1, z2 = (a.mean() - b.mean()) / se, (ja.mean() - jb.mean()) / sj
    worst = max(worst, abs(z1), abs(z2))
    print(json.dumps({"sweep": k, "mean_E_gpu": round(float(a.mean()), 4), "mean_E_oracle": round(float(b.mean()), 4), "difference_in_standard_errors": round(float(z1), 2),
                      "accepted_last_50_gpu": round(float(ja.mean()), 1), "accepted_last_50_oracle": round(float(jb.mean()), 1), "difference_in_standard_errors_acc": round(float(z2), 2)}), flush=True)
dev = np.abs(zg - zo) / np.sqrt(np.maximum(zg + zo, 1.0))
print(json.dumps({"z_profile_bins_largest_poisson_deviation": round(float(dev.max()), 2), "particles_counted_equal": bool(zg.sum() == zo.sum()),
                  "worst_difference_in_standard_errors": round(float(worst), 2), "ok": bool(worst < 4.0),
                  "note": "the bins of successive gathers of one chain are correlated, so the Poisson deviation is informative only"}))
