#!/usr/bin/env python3
"""equil_probe.py N NREP NA NZ CHUNKS [--therm] [--save PREFIX:K] -- where does a production run spend its time?

VERDICT r4 #10: the headline is measured on the fcc start relaxing for 25 sweeps; the reference's runs thermalise for ~4e6
sweeps (main.c:15-18) before sampling (SMC.c:110-126).  This probe runs the benchmark system for thousands of sweeps in
chunks (CHUNKS = comma-separated sweep counts) and prints after every chunk: sweeps so far, sweep-kernel ms per sweep in the
chunk, acceptance in the chunk, ensemble mean energy, the z profile's standard deviation and occupied range -- so the state a
number belongs to is identifiable -- and, with --save, the positions of the first K replicas at the end (for the diagnostic
build's `executed` counters and for the oracle).  --therm: the chunks are thermalisation sweeps at 2A (SMC.c:110) each followed by
one production sweep; default: production sweeps at A.
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import smcx_loader

S = smcx_loader.load()
N, nrep, Na, Nz = (int(v) for v in sys.argv[1:5])
chunks = [int(v) for v in sys.argv[5].split(",")]
therm = "--therm" in sys.argv
save = None
kw = {}
for i, a in enumerate(sys.argv):
    if a == "--save":
        save = sys.argv[i + 1].split(":")
    if a == "--kernel":
        kw["tune_kernel"] = int(sys.argv[i + 1])
    if a == "--resort":
        kw["tune_resort"] = int(sys.argv[i + 1])
p = S.default_params(N, nrep, **kw)
done = 0
with S.Engine(p) as eng:
    print(json.dumps({"kernel": eng.kernel_form[1], "N": N, "replicas": nrep, "start": "fcc(%d,%d)" % (Na, Nz), "therm": therm}), flush=True)
    eng.upload(S.fcc_init(Na, Nz), S.W_REFERENCE)
    for k in chunks:
        t0 = time.time()
        if therm:
            eng.run(k, 1, 1)
            k += 1
        else:
            eng.run(0, k, max(1, min(k, 10)))
        ms, launches = eng.last_kernel_ms()
        run_ms = eng.last_run_ms()
        ob = eng.observables()
        done += k
        zh = ob["zhist"].sum(axis=0).astype(float)
        zc = (np.arange(p.Ncz) + 0.5) / p.Ncz * p.Lz - p.Lz / 2
        zm = (zh * zc).sum() / zh.sum()
        zs = np.sqrt((zh * (zc - zm) ** 2).sum() / zh.sum())
        occ = np.nonzero(zh > 1e-4 * zh.sum())[0]
        print(json.dumps({"sweeps_done": done, "chunk": k, "kernel_ms_per_sweep": ms / k, "device_ms_per_sweep": run_ms / k,
                          "acceptance_in_chunk": float(ob["accepted"].sum()) / (nrep * (k if not therm else 1) * N),
                          "therm_acceptance": float(eng.therm_acceptance().mean()) if therm else None,
                          "E_last_mean": float(ob["E_last"].mean()), "E_last_std": float(ob["E_last"].std()),
                          "z_std": zs, "z_bins_occupied": [int(occ[0]), int(occ[-1])],
                          "z_profile_last_gathers": [round(float(x) / zh.sum(), 4) for x in zh],
                          "wall_s": time.time() - t0}), flush=True)
    if save:
        K = int(save[1])
        R = eng.positions()[:K]
        np.save(save[0] + "_R.npy", R)
        E = eng.total_energy()[:K]
        ob = eng.observables()
        print(json.dumps({"saved": save[0] + "_R.npy", "replicas": K, "E_recomputed": [float(x) for x in E[:8]],
                          "E_last": [float(x) for x in ob["E_last"][:8]]}), flush=True)
