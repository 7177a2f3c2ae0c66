#!/usr/bin/env python3
"""how far do two correct kernels drift apart by rounding alone?  N = 4000 on fcc(10,10) (the ragged case of
test_hand_scheduled_kernel_matches_compiled_kernel), 64 replicas: max |dR| and max |dE| between sweep_kernel_mi and the
z-ordered default kernel after 1, 2, 3 sweeps, and of each against the ORACLE chain for four replicas (through gpurun):
   python tools/probes/ragged_divergence.py            (SMCX_LIB picks the library)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np
import torch  # noqa: F401
import smcx_loader
import oracle_lib as O
S = smcx_loader.load()
N, lat, nrep = 4000, (10, 10), 64
R0 = S.fcc_init(*lat)
for nsw in (1, 2, 3):
    out = {}
    for tag, kernel in (("mi", S.KERNEL_MI), ("mc", S.KERNEL_AUTO)):
        p = S.default_params(N, nrep, flags=S.FLAGS_REFERENCE | S.FLAG_SERIES, tune_slots=64, tune_waves=1, tune_kernel=kernel)
        with S.Engine(p) as e:
            e.upload(R0, S.W_REFERENCE)
            e.run(0, nsw, nsw)
            E, jj = e.series(nsw)
            out[tag] = (e.positions().copy(), E.copy(), jj.copy(), e.kernel_form[1])
    dR = np.abs(out["mi"][0] - out["mc"][0]).max(axis=1)
    dE = np.abs(out["mi"][1] - out["mc"][1]).max()
    line = "%d sweeps: %s vs %s  max|dR| %.3e (median over replicas %.3e)  max|dE| %.3e  jj equal %s" % (
        nsw, out["mi"][3], out["mc"][3], dR.max(), np.median(dR), dE, np.array_equal(out["mi"][2], out["mc"][2]))
    s = O.make_sys(N, M=p.M, L=p.L, Lz=p.Lz, cutoff=p.cutoff, a0=p.a0, b0=p.b0, Ncx=p.Ncx, Ncz=p.Ncz)
    worst = int(np.argmax(dR))
    for r in sorted({0, 1, worst}):
        ref = O.chain(s, 12345 + r, np.asarray(R0, dtype=np.float64), O.W_FIXTURE, 1.1, 1.1, 0, nsw, nsw)
        line += "\n      replica %2d against the oracle: mi %.3e  mc %.3e   (accepted %s / %s / %s)" % (
            r, np.abs(out["mi"][0][r] - ref["R"]).max(), np.abs(out["mc"][0][r] - ref["R"]).max(),
            list(out["mi"][2][r]), list(out["mc"][2][r]), list(ref["jj"]))
    print(line, flush=True)
