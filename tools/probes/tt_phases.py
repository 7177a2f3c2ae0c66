#!/usr/bin/env python3
"""where a move's time goes in the two-team kernels: cycles between the marks of the stamps variant (gen_sweep_ma.py mark(k)),
summed per WAVEFRONT of a replica and divided by the moves (team A = the first half of the wavefronts):
   cd montecarlo-surfacer_amd/csrc && rm -f smcx_sweep_m*_body*.inc && SMCX_GEN_TT_STAMPS=1 make VARIANT=stamps EXTRA=-DSMCX_TT_STAMPS
   rm -f smcx_sweep_m*_body*.inc && make            (regenerate the product's bodies)
   SMCX_LIB=.../libsmcx_stamps.so python tools/probes/tt_phases.py      (through gpurun)
Every mark costs about 190 cycles itself (charged to the interval after it): the sum exceeds the product's time per move."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch  # noqa: F401
import smcx_loader
S = smcx_loader.load()
NAMES = {1: "screen", 2: "excl/assign/fetch", 3: "probe round 0", 4: "further rounds", 11: "reduce4",
         5: "displ load + write partials", 6: "AT the barrier", 7: "exchange + Metropolis + proposal chains",
         8: "accept (or not)", 9: "select", 10: "readlanes + row change"}
ORDER = [1, 2, 3, 4, 11, 5, 6, 7, 8, 9, 10]
# (round 4: the 16 x 2 two-team kernel of config 2 is retired -- sweep_kernel_ml16 serves it; mt64x8 is the one two-team kernel left)
for label, N, nrep, lat, g in (("config 5", 16384, 256, (16, 16), (64, 8)),):
    p = S.default_params(N, nrep, tune_slots=g[0], tune_waves=g[1])
    with S.Engine(p) as e:
        e.upload(S.fcc_init(*lat), S.W_REFERENCE)
        e.run(0, 1, 10)                      # ONE sweep = one launch: the rows hold that launch's sums
        ms, _ = e.last_kernel_ms()
        rows = e.clk_rows(256).astype(np.float64)
        name = e.kernel_form[1]
        acc = e.observables()["acceptance_ratio"].mean()
    W = g[1]
    moves = N + 2.0                           # loop iterations per sweep (two runs, one extra each)
    per = rows.mean(axis=0).reshape(16, 16)[:W] / moves
    total = ms * 1e-3 * 2.39e9 / moves
    print("%s %s: %.3f ms per sweep = %.0f cycles per move at 2.39 GHz (with the marks), acceptance %.3f; columns = wavefronts "
          "0..%d (team A first)" % (label, name, ms, total, acc, W - 1))
    for k in ORDER:
        print("   %-42s %s" % (NAMES[k], " ".join("%6.0f" % per[w, k] for w in range(W))))
    print("   %-42s %s" % ("sum", " ".join("%6.0f" % per[w, 1:16].sum() for w in range(W))), flush=True)
