#!/usr/bin/env python3
"""two-team kernels against the kernels they stand in for: ms per sweep at BASELINE configs 2 and 5 (per-GPU share)
   python tools/probes/tt_probe.py            (through gpurun)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: F401  (one HIP runtime per process)
import smcx_loader
S = smcx_loader.load()
for label, N, nrep, lat, sweeps, geoms in (("config 2: N=1024 x 1024 replicas", 1024, 1024, (8, 4), 40, ((16, 1), (16, 2))),
                                           ("N=1024 x 2048 replicas", 1024, 2048, (8, 4), 40, ((16, 1), (16, 2))),
                                           ("N=1024 x 4096 replicas", 1024, 4096, (8, 4), 20, ((16, 1), (16, 2))),
                                           ("config 5: N=16384 x 256 replicas", 16384, 256, (16, 16), 4, ((32, 8), (64, 8), (32, 16))),
                                           ("N=16384 x 512 replicas", 16384, 512, (16, 16), 4, ((32, 8), (64, 8), (32, 16)))):
    for s_, w_ in geoms:
        p = S.default_params(N, nrep, tune_slots=s_, tune_waves=w_)
        with S.Engine(p) as e:
            e.upload(S.fcc_init(*lat), S.W_REFERENCE)
            e.run(0, 1, 10)
            e.run(0, sweeps, 10)
            ms, _ = e.last_kernel_ms()
            run_ms = e.last_run_ms()
            acc = e.observables()["acceptance_ratio"].mean()
            print("%-36s %-28s %8.3f ms per sweep (kernel), %8.3f whole run; %.3e pair-evals/s; acceptance %.4f" %
                  (label, e.kernel_form[1], ms / sweeps, run_ms / sweeps, nrep * sweeps * 2.0 * N * (N - 1) / (run_ms * 1e-3), acc), flush=True)
