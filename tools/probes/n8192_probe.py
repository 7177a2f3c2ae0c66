#!/usr/bin/env python3
"""4096 < N <= 8192: the z-ordered four-wavefront kernel (round 3) against round 1's sweep_kernel_mx<64,2> it replaces"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: F401
import smcx_loader
S = smcx_loader.load()
for N, lat, nrep in ((8192, (16, 8), 512), (8192, (16, 8), 1024), (6144, (16, 6), 512)):
    for kernel in (S.KERNEL_AUTO, S.KERNEL_MX):
        p = S.default_params(N, nrep, tune_kernel=kernel)
        with S.Engine(p) as e:
            e.upload(S.fcc_init(*lat), S.W_REFERENCE)
            e.run(0, 1, 10)
            e.run(0, 5, 10)
            ms, _ = e.last_kernel_ms()
            print("N=%d x %d replicas  %-34s S=%d x %d  %8.3f ms per sweep  %.3e pair-evals/s" %
                  (N, nrep, e.kernel_form[1], e.geometry[0], e.geometry[1], ms / 5, nrep * 5 * 2.0 * N * (N - 1) / (e.last_run_ms() * 1e-3)), flush=True)
