#!/usr/bin/env python3
"""where a move's time goes in the two-team kernels: cycles between the marks of the stamps variant (gen_sweep_ma.py mark(k)),
summed per team and divided by waves and moves:
   cd montecarlo-surfacer_amd/csrc && rm -f smcx_sweep_m*_body*.inc && SMCX_GEN_TT_STAMPS=1 make VARIANT=stamps EXTRA=-DSMCX_TT_STAMPS
   rm -f smcx_sweep_m*_body*.inc && make            (regenerate the product's bodies)
   SMCX_LIB=.../libsmcx_stamps.so python tools/probes/tt_phases.py      (through gpurun)
Every mark costs about 100 cycles itself (charged to the interval after it): the sum exceeds the product's time per move."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch  # noqa: F401
import smcx_loader
S = smcx_loader.load()
NAMES = {1: "screen", 2: "excl/assign/fetch", 3: "probe round 0 (walls dz, wait, body)", 4: "further rounds", 11: "reduce4",
         5: "displ load + write partials", 6: "AT the barrier", 7: "exchange reads + Metropolis to the compare",
         8: "accept (or not)", 9: "side result", 10: "proposal + row change"}
ORDER = [1, 2, 3, 4, 11, 5, 6, 7, 8, 9, 10]
for label, N, nrep, lat, g in (("config 2", 1024, 1024, (8, 4), (16, 2)), ("config 5", 16384, 256, (16, 16), (64, 8))):
    p = S.default_params(N, nrep, tune_slots=g[0], tune_waves=g[1])
    with S.Engine(p) as e:
        e.upload(S.fcc_init(*lat), S.W_REFERENCE)
        e.run(0, 1, 10)                      # ONE sweep = one launch: the rows hold that launch's sums
        ms, _ = e.last_kernel_ms()
        rows = e.clk_rows(32).astype(np.float64)
        name = e.kernel_form[1]
        acc = e.observables()["acceptance_ratio"].mean()
    K = g[1] // 2
    moves = N + 2.0                           # loop iterations per sweep (two runs, one extra each)
    per = rows.mean(axis=0) / K / moves
    total = ms * 1e-3 * 2.39e9 / moves
    print("%s %s: %.3f ms per sweep = %.0f cycles per move at 2.39 GHz (with the marks), acceptance %.3f" % (label, name, ms, total, acc))
    for k in ORDER:
        print("   %-46s team A %7.0f   team B %7.0f" % (NAMES[k], per[k], per[16 + k]))
    print("   %-46s team A %7.0f   team B %7.0f" % ("sum", per[1:16].sum(), per[17:32].sum()), flush=True)
