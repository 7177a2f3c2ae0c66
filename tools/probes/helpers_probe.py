#!/usr/bin/env python3
"""time of the helper kernels around the sweep (rng_prepass_kernel per chunk, zsort_kernel per sweep) at config 3, from the
whole-run time minus the sweep kernels' own time:  [SMCX_LIB=...] python tools/probes/helpers_probe.py   (through gpurun)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: F401
import smcx_loader
S = smcx_loader.load()
N, nrep, sweeps = 4096, 4096, 20
p = S.default_params(N, nrep)
with S.Engine(p) as e:
    e.upload(S.fcc_init(8, 16), S.W_REFERENCE)
    e.run(0, 2, 10)
    best = (1e30, 0)
    for _ in range(3):
        e.run(0, sweeps, 10)
        ms, _ = e.last_kernel_ms()
        run = e.last_run_ms()
        best = min(best, (run / sweeps, ms / sweeps))
    print("%-18s whole run %.3f ms per sweep, sweep kernels %.3f, everything else %.3f" %
          (os.path.basename(S.LIB_PATH), best[0], best[1], best[0] - best[1]), flush=True)
