#!/usr/bin/env python3
"""oracle/time_ref.py N Na Nz seed sweeps -- BENCH INFRASTRUCTURE (bench.py's cpu_baseline leg only): times `sweeps` calls
of the REAL reference's oneParticleMoves (SMC.c:278-351, compiled where it lies by oracle/build_ref.sh into
oracle/_ref/libref_smc_N<n>_O3.so, -O3 -march=x86-64-v3) on one core, the way one MPI rank of the reference runs one chain
(SMC.c:40, 66-95).  One process per chain: the reference draws from libc's single hidden rand() state.  Prints the
seconds spent inside the sweep loop."""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O  # noqa: E402  (the fcc start and the wall fixture of the synthetic workload)

N, Na, Nz, seed, sweeps = (int(x) for x in sys.argv[1:6])
L = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libref_smc_N%d_O3.so" % N))
dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int)
L.refw_sweeps.argtypes = [C.c_long, dp, dp, dp] + [C.c_double] * 4 + [C.c_int, ip, dp]
L.refw_energy.argtypes = [dp, C.c_double]
L.refw_energy.restype = C.c_double
L.refw_walls_energy.argtypes = [dp, dp, C.c_double, C.c_double]
L.refw_walls_energy.restype = C.c_double
s = O.make_sys(N)
R = O.fcc(Na, Nz).copy()
Rn = R.copy()
W = np.ascontiguousarray(O.W_FIXTURE, dtype=np.float64)
E = np.zeros(sweeps + 1)
jj = np.zeros(sweeps, dtype=np.int32)
p = lambda a, t=C.c_double: a.ctypes.data_as(C.POINTER(t))
E[0] = L.refw_energy(p(R), s.L) + L.refw_walls_energy(p(R), p(W), s.L, s.Lz)
t0 = time.perf_counter()
L.refw_sweeps(seed, p(R), p(Rn), p(W), s.L, s.Lz, 1.1, 1.1, sweeps, p(jj, C.c_int), p(E))
print("%.6f %d %.9f" % (time.perf_counter() - t0, int(jj.sum()), E[-1]))
