#!/usr/bin/env python3
"""Soak of the diagnostic build (libsmcx_check.so, SMCX_CHECK_MB=2): long, thermalising trajectories through every family of
z-ordered sweep kernels; beside EVERY screen pass the fp64 cutoff test of EVERY cell runs on the device and counts the pairs
inside the cutoff whose candidate bit the pass did not set.  Prints one line per case; exits 1 if any count is not zero.
   python tools/soak_check.py            (through gpurun, a few minutes)
The parity suite runs the same check for 2-4 sweeps from the lattice starts (tests/test_gpu_configs.py); this runs it for
tens of sweeps with thermalisation at 2A first, where the film has melted and the z order is rebuilt from moved particles."""
import ctypes as C
import importlib.util
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["SMCX_LIB"] = os.path.join(ROOT, "montecarlo-surfacer_amd", "libsmcx_check.so")
os.environ["SMCX_CHECK_MB"] = "2"
spec = importlib.util.spec_from_file_location("smcx_chk", os.path.join(ROOT, "montecarlo-surfacer_amd", "__init__.py"))
K = importlib.util.module_from_spec(spec); spec.loader.exec_module(K)
LONG = "long" in sys.argv                                 # round 5: thousands of sweeps, into the walls (the states production runs sit in)
SCALE = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 1     # replicas x SCALE
CASES = [  # N, lattice, replicas, thermalisation sweeps (at 2A), production sweeps, gather lapse, slots, waves
    (4096, (8, 16), 128, 10, 30, 10, 64, 1),     # the benchmark's kernel, config 3's start
    (4096, (16, 4), 32, 5, 10, 5, 64, 1),        # dense film
    (4000, (10, 10), 64, 10, 20, 7, 64, 1),      # ragged N
    (2100, (5, 21), 64, 10, 20, 3, 64, 1),       # thin tall start, a gather every third sweep
    (1024, (8, 4), 256, 20, 60, 10, 16, 1),      # config 2's kernel: ml16 (few replicas of N <= 1024: positions in LDS)
    (1000, (5, 10), 128, 20, 60, 10, 16, 1),     # ragged, ml16
    (1024, (8, 4), 1100, 4, 12, 6, 16, 1),       # mc16 (more than 1024 replicas)
    (2048, (8, 8), 64, 10, 30, 10, 32, 1),       # mc32
    (16384, (16, 16), 8, 2, 6, 3, 64, 8),        # config 5, two teams
    (16384, (16, 16), 8, 2, 6, 3, 32, 8),        # mc32x8
    (16384, (16, 16), 8, 2, 6, 3, 64, 4),        # mc64x4
    (9000, (15, 10), 8, 4, 8, 4, 64, 8),         # ragged, two teams
    (10000, (10, 25), 8, 4, 8, 4, 32, 8),        # tall
    (8192, (16, 8), 8, 4, 8, 4, 32, 4),          # 4096 < N <= 8192
    (4800, (10, 12), 8, 6, 12, 4, 32, 4),
]
if LONG:
    CASES = [
        (4096, (8, 16), 64, 0, 2500, 100, 64, 1),    # the benchmark's kernel until the gas fills the box and sits at both walls
        (1024, (8, 4), 128, 0, 3000, 100, 16, 1),    # ml16
        (2048, (8, 8), 64, 0, 2000, 100, 32, 1),     # mc32
        (16384, (16, 16), 8, 0, 300, 50, 64, 8),     # two teams while the slab evaporates
        (6144, (16, 6), 8, 0, 600, 50, 32, 4),       # mc32x4
    ]
bad = 0
for N, lat, nrep, eq, nsw, gl, slots, waves in CASES:
    nrep *= SCALE
    p = K.default_params(N, nrep, tune_slots=slots, tune_waves=waves)
    t0 = time.time()
    with K.Engine(p) as eng:
        name = eng.kernel_form[1]
        eng.upload(K.fcc_init(*lat), K.W_REFERENCE)
        eng.run(eq, nsw, gl)
        cnt = (C.c_uint64 * 8)()
        f = K._lib().smcx_debug_work_counts
        f.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
        assert f(eng._h, cnt) == 0
        ob = eng.observables()
    moves = nrep * (eq + nsw) * N
    print("%-28s N=%5d fcc%-8s %4d replicas x (%d+%d) sweeps = %.2e moves: %d pairs inside the cutoff, %d candidate bits, "
          "%d MISSED; acceptance %.3f; %.0f s" % (name, N, lat, nrep, eq, nsw, moves, cnt[0], cnt[1], cnt[2],
                                                  ob["acceptance_ratio"].mean(), time.time() - t0), flush=True)
    bad += int(cnt[2])
sys.exit(1 if bad else 0)
