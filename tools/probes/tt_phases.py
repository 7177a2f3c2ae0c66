#!/usr/bin/env python3
"""where a move's time goes in the two-team kernels: cycles from the start of a move to the barrier and cycles AT the
barrier, summed per team, from the stamps variant of the library:
   cd montecarlo-surfacer_amd/csrc && rm -f smcx_sweep_m*_body*.inc && SMCX_GEN_TT_STAMPS=1 make VARIANT=stamps EXTRA=-DSMCX_TT_STAMPS
   rm -f smcx_sweep_m*_body*.inc && make            (regenerate the product's bodies)
   SMCX_LIB=.../libsmcx_stamps.so python tools/probes/tt_phases.py      (through gpurun)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch  # noqa: F401
import smcx_loader
S = smcx_loader.load()
for label, N, nrep, lat, g in (("config 2", 1024, 1024, (8, 4), (16, 2)), ("config 5", 16384, 256, (16, 16), (64, 8))):
    p = S.default_params(N, nrep, tune_slots=g[0], tune_waves=g[1])
    with S.Engine(p) as e:
        e.upload(S.fcc_init(*lat), S.W_REFERENCE)
        e.run(0, 1, 10)                      # ONE sweep = one launch: the rows hold that launch's sums
        ms, _ = e.last_kernel_ms()
        rows = e.clk_rows().astype(np.float64)
        name = e.kernel_form[1]
    K = g[1] // 2
    moves = N + 2.0                           # loop iterations per sweep (two runs, one extra each)
    preA, waitA, preB, waitB = (rows[:, j].mean() / K / moves for j in range(4))
    total = ms * 1e-3 * 2.39e9 / moves
    print("%s %s: %.3f ms per sweep = %.0f cycles per move at 2.39 GHz; team A: %.0f to the barrier + %.0f at it; team B: %.0f + %.0f; "
          "after the barrier (Metropolis, accept, proposal, loop): %.0f" % (label, name, ms, total, preA, waitA, preB, waitB,
                                                                          total - max(preA + waitA, preB + waitB)), flush=True)
