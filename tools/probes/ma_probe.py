import sys, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np, torch
import smcx_loader, oracle_lib as O
S = smcx_loader.load()
T = A = 1.1
R0 = O.fcc(8, 16)
nrep = 2
p = S.default_params(4096, nrep, flags=S.FLAGS_REFERENCE | S.FLAG_SERIES, tune_slots=64, tune_waves=1)
eng = S.Engine(p); print(eng.kernel_form, flush=True)
eng.upload(R0, O.W_FIXTURE)
eng.run(0, 1, 1)
ob = eng.observables(); Es, jj = eng.series(1); Rg = eng.positions()
s = O.make_sys(4096)
for r in range(nrep):
    ref = O.chain(s, 12345 + r, R0, O.W_FIXTURE, T, A, 0, 1, 1)
    print(r, "acc", jj[r], ref["jj"], "E", Es[r], ref["E"], "maxdR", np.abs(Rg[r]-ref["R"]).max(), flush=True)
