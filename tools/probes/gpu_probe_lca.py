"""ad hoc: time the cluster analysis at benchmark sizes (not a test)"""
import sys, time
import numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, ".")  # run from the repo root
import torch  # noqa: F401  (one HIP runtime per process: torch first)
import oracle_lib as O
import smcx_loader
S = smcx_loader.load()
for (N, Na, Nz, nrep, label) in [(4096, 8, 16, 4096, "benchmark lattice"), (4096, 16, 4, 1024, "dense film"),
                                 (16384, 16, 16, 256, "N=16384 dense")]:
    R0 = O.fcc(Na, Nz)
    p = S.default_params(N, nrep)
    eng = S.Engine(p)
    eng.upload(R0, O.W_FIXTURE)
    eng.cluster_update()
    t = time.time(); eng.cluster_update(); dt = time.time() - t
    n1, h2, h3, ov, k = eng.cluster_counts()
    print(label, "N", N, "nrep", nrep, "lca ms", 1e3 * dt, "n1/rep", n1[0] // 2, "h2", h2[0][:8] // 2, "ov", ov[0], flush=True)
    if N == 4096 and Na == 16:
        ref, ovr = O.cluster_analysis(N, R0, 33.0, 1.7)
        c1, c2, c3 = O.cluster_counts(N, ref)
        print("  oracle n1", c1, "h2", c2[:8], "match", c1 == n1[0] // 2 and np.array_equal(c2, h2[0] // 2) and np.array_equal(c3, h3[0] // 2))
    eng.close()
