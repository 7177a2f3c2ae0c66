#!/usr/bin/env python3
"""A/B of two builds of the library in one session: kernel ms per sweep of configs 2, 3, 5 (per-GPU share) and the dense film.
   SMCX_LIB=.../libsmcx_base.so python tools/probes/ab_probe.py ; python tools/probes/ab_probe.py      (through gpurun)
   prints one line per workload with a checksum of the accepted counts (the two builds must agree)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: F401  (one HIP runtime per process)
import smcx_loader
S = smcx_loader.load()
which = sys.argv[1].split(",") if len(sys.argv) > 1 else ["c2", "c3", "c5", "dense"]
W = {"c2": ("config 2: N=1024 x 1024", 1024, 1024, (8, 4), 40), "c3": ("config 3: N=4096 x 4096", 4096, 4096, (8, 16), 8),
     "c5": ("config 5: N=16384 x 256", 16384, 256, (16, 16), 4), "dense": ("dense film fcc(16,4) x 4096", 4096, 4096, (16, 4), 4),
     "n8192": ("N=8192 x 1024", 8192, 1024, (16, 8), 4), "c5x512": ("N=16384 x 512", 16384, 512, (16, 16), 4)}
for k in which:
    label, N, nrep, lat, sweeps = W[k]
    p = S.default_params(N, nrep)
    with S.Engine(p) as e:
        e.upload(S.fcc_init(*lat), S.W_REFERENCE)
        e.run(0, 1, 10)
        best = 1e30
        for _ in range(3):
            e.run(0, sweeps, 10)
            ms, _ = e.last_kernel_ms()
            best = min(best, ms / sweeps)
        o = e.observables()
        print("%-12s %-30s %-26s %8.3f ms per sweep (kernel, best of 3)  acc %.6f  meanE %.9f" %
              (os.path.basename(S.LIB_PATH), label, e.kernel_form[1], best, o["acceptance_ratio"].mean(), o["meanE"].mean()), flush=True)
