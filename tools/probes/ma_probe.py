"""one-off check of the hand-scheduled kernels against the oracle: python tools/probes/ma_probe.py [nsweeps] [nrep]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
import smcx_loader, oracle_lib as O
S = smcx_loader.load()
T = A = 1.1
nsw = int(sys.argv[1]) if len(sys.argv) > 1 else 1
nrep = int(sys.argv[2]) if len(sys.argv) > 2 else 2
R0 = O.fcc(8, 16)
p = S.default_params(4096, nrep, flags=S.FLAGS_REFERENCE | S.FLAG_SERIES, tune_slots=64, tune_waves=1)
eng = S.Engine(p); print(eng.kernel_form, flush=True)
eng.upload(R0, O.W_FIXTURE)
eng.run(0, nsw, 1)
ob = eng.observables(); Es, jj = eng.series(nsw); Rg = eng.positions()
s = O.make_sys(4096)
for r in range(min(nrep, 2)):
    ref = O.chain(s, 12345 + r, R0, O.W_FIXTURE, T, A, 0, nsw, 1)
    print(r, "acc", jj[r], ref["jj"], "E", Es[r], ref["E"], "maxdR", np.abs(Rg[r]-ref["R"]).max(), flush=True)
