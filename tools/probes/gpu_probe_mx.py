"""ad hoc: where do the screened and the fp64 kernel differ? (not a test)"""
import sys
import numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, ".")  # run from the repo root
import torch  # noqa: F401
import oracle_lib as O
import smcx_loader
S = smcx_loader.load()
N, lat, L, slots, waves, nsw = 4096, (8, 16), 33.0, 64, 1, int(sys.argv[1]) if len(sys.argv) > 1 else 1
rs = np.random.RandomState(N + slots)
R0 = O.fcc(lat[0], lat[1], L=L).reshape(-1, 3)[:N].copy()
R0 += 0.05 * rs.standard_normal(R0.shape)
R0[:, 0] -= L * np.rint(R0[:, 0] / L); R0[:, 1] -= L * np.rint(R0[:, 1] / L)
out = []
for kernel, (s, w) in ((1, (64, 1)), (2, (64, 1)), (1, (16, 4)), (1, (32, 2))):
    p = S.default_params(N, 1, L=L, tune_slots=s, tune_waves=w, tune_kernel=kernel, flags=S.FLAGS_REFERENCE | S.FLAG_SERIES)
    eng = S.Engine(p); eng.upload(R0.ravel(), O.W_FIXTURE); eng.run(0, nsw, 1)
    E, jj = eng.series(nsw)
    out.append((eng.positions()[0].copy(), E[0].copy(), jj[0].copy())); eng.close()
ref = out[0]
for name, o in zip(("mx(64,1)", "fp64(16,4)", "fp64(32,2)"), out[1:]):
    d = np.nonzero(ref[0].view(np.uint64) != o[0].view(np.uint64))[0]
    print(name, "E equal", np.array_equal(ref[1].view(np.uint64), o[1].view(np.uint64)), "jj equal", np.array_equal(ref[2], o[2]),
          "pos differing entries", len(d), "particles", np.unique(d // 3)[:10], "coords", np.bincount(d % 3, minlength=3),
          "max rel", np.max(np.abs(ref[0] - o[0]) / (np.abs(ref[0]) + 1e-300)) if len(d) else 0)
