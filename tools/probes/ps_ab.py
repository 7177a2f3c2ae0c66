#!/usr/bin/env python3
"""ps_ab.py -- sweep-kernel time per sweep of the loaded library (SMCX_LIB) at the configurations the pre-screen of the next move's
probe B touches (sweep_kernel_mc64 / mc32 / mc16) and one it does not (ml16): lattice start and, for config 3, the equilibrating
state after 1000 sweeps.  For an A/B against libsmcx_nops.so (make VARIANT=nops GENENV="SMCX_GEN_PS=0").  Through gpurun."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: F401
import smcx_loader
S = smcx_loader.load()
CASES = [("config 3 (4096 x 4096)", 4096, 4096, (8, 16), 0), ("config 3 after 1000 sweeps", 4096, 4096, (8, 16), 1000),
         ("dense film (4096 x 4096)", 4096, 4096, (16, 4), 0), ("N=2048 x 4096", 2048, 4096, (8, 8), 0),
         ("N=1000 x 4096", 1000, 4096, (8, 4), 0), ("config 2 (1024 x 1024)", 1024, 1024, (8, 4), 0)]
for label, N, nrep, lat, pre in CASES:
    p = S.default_params(N, nrep)
    with S.Engine(p) as e:
        e.upload(S.fcc_init(*lat)[:3 * N], S.W_REFERENCE)
        e.run(0, 2 + pre, 10 if not pre else pre + 2)
        best = 1e30
        for _ in range(3):
            e.run(0, 20, 10)
            ms, _n = e.last_kernel_ms()
            best = min(best, ms / 20)
        acc = e.observables()["acceptance_ratio"].mean()
        print("%-14s %-28s %-28s %8.4f ms per sweep  (acceptance %.3f)" % (os.path.basename(S.LIB_PATH), label, e.kernel_form[1], best, acc), flush=True)
