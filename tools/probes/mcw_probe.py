"""first checks of the four-wave kernel against the oracle: python tools/probes/mcw_probe.py [nsweeps]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
import smcx_loader, oracle_lib as O
S = smcx_loader.load()
T = A = 1.1
nsw = int(sys.argv[1]) if len(sys.argv) > 1 else 1
slots, waves = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (64, 4)
N, lat = 16384, (16, 16)
R0 = O.fcc(*lat)
p = S.default_params(N, 2, flags=S.FLAGS_REFERENCE | S.FLAG_SERIES, tune_slots=slots, tune_waves=waves)
eng = S.Engine(p); print(eng.kernel_form, eng.geometry, flush=True)
eng.upload(R0, O.W_FIXTURE)
eng.run(0, nsw, 1)
Es, jj = eng.series(nsw); Rg = eng.positions()
s = O.make_sys(N)
for r in range(2):
    ref = O.chain(s, 12345 + r, R0, O.W_FIXTURE, T, A, 0, nsw, 1)
    print(r, "acc", jj[r], ref["jj"], "E", Es[r], ref["E"], "maxdR", np.abs(Rg[r]-ref["R"]).max(), flush=True)
