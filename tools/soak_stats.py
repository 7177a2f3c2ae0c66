#!/usr/bin/env python3
"""Beyond the chaos horizon two correct kernels give different trajectories and the same statistics.  Full-size ensembles of
the z-ordered kernels against the all-fp64 kernel (tune_kernel = SMCX_KERNEL_FP64: no screen, no compact copies) from the same
start and seeds: ensemble mean of the energy after every tenth sweep, of the run's mean energy and of the acceptance ratio,
with the difference in units of its standard error.  A pair the screen dropped systematically would show as a bias of many
standard errors (4096 replicas: the standard error of the mean energy is ~1e-4 of its value).
   python tools/soak_stats.py            (through gpurun; ~1 minute)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401
import smcx_loader
S = smcx_loader.load()
CASES = [  # N, lattice, replicas, thermalisation, production sweeps, kernel arm (tune kwargs)
    (4096, (8, 16), 4096, 20, 100, {}),
    (1024, (8, 4), 1024, 20, 200, {}),
    (4096, (16, 4), 1024, 5, 30, {}),
    (16384, (16, 16), 128, 2, 10, {}),
]
worst = 0.0
for N, lat, nrep, eq, nsw, kw in CASES:
    res = []
    for arm in (kw, dict(tune_kernel=S.KERNEL_FP64)):
        p = S.default_params(N, nrep, flags=S.FLAGS_REFERENCE | S.FLAG_SERIES, **arm)
        with S.Engine(p) as e:
            name = e.kernel_form[1]
            e.upload(S.fcc_init(*lat), S.W_REFERENCE)
            e.run(eq, nsw, 10)
            E, jj = e.series(nsw)
            ob = e.observables()
            res.append((name, E, jj, ob["meanE"].copy(), ob["acceptance_ratio"].copy(), e.last_kernel_ms()[0] / (eq + nsw)))
    (na, Ea, ja, ma, aa, ta), (nb, Eb, jb, mb, ab, tb) = res
    same = int((ja == jb).all(axis=0).cumprod().sum())        # sweeps over which every replica's accepted count agrees

    def z(x, y):
        se = np.sqrt(x.var(ddof=1) / x.size + y.var(ddof=1) / y.size)
        return (x.mean() - y.mean()) / se if se > 0 else 0.0
    zs = [z(Ea[:, k], Eb[:, k]) for k in range(10, nsw + 1, 10)]
    zm, za = z(ma, mb), z(aa, ab)
    worst = max([worst, abs(zm), abs(za)] + [abs(v) for v in zs])
    print("N=%5d fcc%-8s %4d replicas x (%d+%d) sweeps: %s (%.2f ms/sweep) vs %s (%.2f): accepted counts of all replicas equal over "
          "the first %d sweeps; <mean E> %.6f vs %.6f (z = %+.2f), acceptance %.5f vs %.5f (z = %+.2f); z of <E> after every "
          "tenth sweep: %s" % (N, lat, nrep, eq, nsw, na.split("::")[1], ta, nb.split("::")[1], tb, same, ma.mean(), mb.mean(), zm,
                               aa.mean(), ab.mean(), za, " ".join("%+.1f" % v for v in zs)), flush=True)
print("largest |z|: %.2f" % worst)
sys.exit(1 if worst > 5.0 else 0)
