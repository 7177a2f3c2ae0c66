#!/usr/bin/env python3
"""total_energy_time.py -- duration of smcx_total_energy (HIP events around the call are not exported: host clock around the
blocking call, best of 5) at the bench's configurations, lattice start and after 200 sweeps; every value checked against the
energy the chain carried (E_last) and, for replica 0, against the oracle."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import importlib
S = importlib.import_module("montecarlo-surfacer_amd")
import oracle_lib as O

for name, Na, Nz, nrep in (("config 2", 8, 4, 1024), ("config 3", 8, 16, 2048), ("dense film", 16, 4, 2048), ("config 5", 16, 16, 256)):
    R0 = O.fcc(Na, Nz)
    N = R0.size // 3
    p = S.default_params(N, nrep, flags=S.FLAGS_REFERENCE)
    eng = S.Engine(p)
    t0 = time.perf_counter(); eng.upload(R0, O.W_FIXTURE); t_up = time.perf_counter() - t0
    s = O.make_sys(N)
    for label in ("lattice", "after 200 sweeps"):
        if label != "lattice":
            eng.run(0, 200, 200)
        best = 1e9
        for _ in range(5):
            t0 = time.perf_counter(); E = eng.total_energy(); best = min(best, time.perf_counter() - t0)
        ref = O.total_energy(s, np.ascontiguousarray(eng.positions()[0]), O.W_FIXTURE)
        ob = eng.observables()
        inc = np.max(np.abs(ob["E_last"] - E) / np.abs(E)) if label != "lattice" else 0.0
        print("%-10s %-17s N=%5d x %4d  total_energy %7.3f ms  rel. to oracle (replica 0) %.1e  carried vs recomputed (max over replicas) %.1e%s"
              % (name, label, N, nrep, best * 1e3, abs(E[0] - ref) / abs(ref), inc,
                 "  (upload incl. its energy %.1f ms)" % (t_up * 1e3) if label == "lattice" else ""), flush=True)
    eng.close()
