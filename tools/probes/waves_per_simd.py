#!/usr/bin/env python3
"""waves_per_simd.py -- does a fifth wavefront per SIMD still pay?  sweep_kernel_mc32 (N = 2048: 96 VGPRs, five wavefronts per SIMD fit)
at 1024 .. 5120 replicas (1 .. 5 wavefronts per SIMD), and sweep_kernel_mc64 (N = 4096: 128 VGPRs, four fit) at 1024 .. 4096:
sweep kernel ms per sweep and replicas per ms.  If the time grows by much less than the replica count from 4 to 5 wavefronts, the
SIMDs are not issue-saturated at 4 and a 96-VGPR form of the N = 4096 kernel would be worth building."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import smcx_loader
S = smcx_loader.load()
for N, lat, reps in ((2048, (8, 8), (1024, 2048, 3072, 4096, 5120)), (4096, (8, 16), (1024, 2048, 3072, 4096))):
    for nrep in reps:
        p = S.default_params(N, nrep)
        with S.Engine(p) as e:
            e.upload(S.fcc_init(*lat), S.W_REFERENCE)
            e.run(0, 2, 10)
            e.run(0, 20, 10)
            ms, _ = e.last_kernel_ms()
            print(json.dumps({"N": N, "kernel": e.kernel_form[1], "replicas": nrep, "wavefronts_per_simd": nrep / 1024.0,
                              "sweep_kernel_ms_per_sweep": round(ms / 20, 4), "replicas_per_ms": round(nrep / (ms / 20), 1)}), flush=True)
