#!/usr/bin/env python3
"""Generates tests/golden/*.json.  Runs only where /root/reference exists
(this container): the values come from
  - the REAL reference matematicose.c, compiled where it lies into
    oracle/_ref/libmatematicose_ref.so (vecBoxMuller, mean, intmean, variance),
  - this image's real glibc srand()/rand(),
  - reference outputs recorded in SURVEY.md (the reference's SMC.c cannot be
    built here: it needs <fftw3.h> and misccose.c, both absent).
Floats are stored as C99 hex strings so they round-trip exactly.
Usage: make -C oracle ref && python tests/golden/make_golden.py
"""
import ctypes as C
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.path.join(ROOT, "oracle", "_ref", "libmatematicose_ref.so")


def hexes(a):
    return [float(v).hex() for v in np.asarray(a, dtype=np.float64).ravel()]


def main():
    libc = C.CDLL("libc.so.6")
    libc.rand.restype = C.c_int
    ref = C.CDLL(REF)
    dp = C.POINTER(C.c_double)
    ref.vecBoxMuller.argtypes = [C.c_double, C.c_size_t, dp]
    ref.vecBoxMuller.restype = None
    for f in (ref.mean, ref.variance, ref.sum):
        f.argtypes = [dp, C.c_size_t]
        f.restype = C.c_double
    ref.intmean.argtypes = [C.POINTER(C.c_int), C.c_size_t]
    ref.intmean.restype = C.c_double

    # 1. glibc rand()
    rand = {}
    for seed in (0, 1, 42, 12345, 2147483647, 4294967295):
        libc.srand(C.c_uint(seed))
        rand[str(seed)] = [libc.rand() for _ in range(100)]
    json.dump({"source": "glibc %s srand/rand of this image" % os.confstr("CS_GNU_LIBC_VERSION"),
               "first100": rand}, open(os.path.join(HERE, "glibc_rand.json"), "w"), indent=0)

    # 2. the reference's vecBoxMuller (matematicose.c:183-193) on libc's rand()
    cases = []
    for seed, sigma, length in ((42, 0.5, 9), (42, 0.0, 9), (12345, 1.4832396974191326, 6),
                                (12345, 1.4832396974191326, 324), (1, 2.0977130404327, 48),
                                (7, 1.0, 7)):
        A = np.full(length, -777.0)
        libc.srand(C.c_uint(seed))
        ref.vecBoxMuller(sigma, length, A.ctypes.data_as(dp))
        nxt = libc.rand()  # the stream position after the call
        cases.append({"seed": seed, "sigma": float(sigma).hex(), "length": length,
                      "prefill": float(-777.0).hex(), "out": hexes(A), "next_rand": nxt})
    # 3. mean / intmean / variance (matematicose.c:50-63, 96-103)
    rs = np.random.RandomState(3)
    E = -300.0 + rs.standard_normal(257)
    jj = rs.randint(0, 1024, size=100).astype(np.int32)
    stats = {"E": hexes(E), "jj": [int(v) for v in jj],
             "mean": float(ref.mean(E.ctypes.data_as(dp), E.size)).hex(),
             "variance": float(ref.variance(E.ctypes.data_as(dp), E.size)).hex(),
             "intmean": float(ref.intmean(jj.ctypes.data_as(C.POINTER(C.c_int)), jj.size)).hex()}
    json.dump({"source": "/root/reference/matematicose.c compiled into oracle/_ref (real reference code)",
               "vecBoxMuller": cases, "stats": stats},
              open(os.path.join(HERE, "matematicose_ref.json"), "w"), indent=0)

    # 4. outputs of the real reference recorded by the survey
    pins = {
        "source": "SURVEY.md [probe] values: the reference built and run with shims during the survey",
        "W": {"cite": "SURVEY.md 8a row W; initializeWalls(1.6,0.0,3.0,0.5) SMC.c:475-501, glibc 2.35",
              "values": [962.2264072645321, 57.35316319850277, 874.39446992695275, 52.11797177356199,
                         857.36680597299653, 51.103043912231705, 1024.1964124687327, 61.046863345428257,
                         925.40789594507817, 55.158608910148025, 913.63518965684239, 54.456900933792724,
                         848.90539177252572, 50.598704324515197, 992.35137245273086, 59.148751047416368,
                         844.42493013196849, 50.331648000000015]},
        "E0": {"cite": "SURVEY.md 8d: energy+wallsEnergy of fcc(Na,Nz), L=33, Lz=240, W fixture",
               "cases": [{"Na": 8, "Nz": 4, "N": 1024, "E0": -36.5227192},
                         {"Na": 8, "Nz": 16, "N": 4096, "E0": -156.0516184},
                         {"Na": 16, "Nz": 4, "N": 4096, "E0": -9730.483583},
                         {"Na": 16, "Nz": 16, "N": 16384, "E0": -41824.76491}],
               "digits": 10},
        "chain_N108": {"cite": "SURVEY.md 8c sanity values: seed 12345, N=108, L=33, Lz=200, T=A=1.1, "
                               "reference initializeBox/initializeWalls, 20 sweeps, no thermalisation",
                       "E_wall0": -4.0581627260515186e-14, "E_pp0": 0.0,
                       "E_incremental_20": -3.8631457699032183,
                       "E_recomputed_20": -3.863145769903213,
                       "accepted": 2048, "moves": 2160},
        "acceptance": {"cite": "SURVEY.md section 6 / BASELINE.md section 2 (seed 12345, T=A=1.1)",
                       "cases": [{"N": 256, "Na": 4, "Nz": 4, "sweeps": 2000, "ratio": 0.958},
                                 {"N": 1024, "Na": 8, "Nz": 4, "sweeps": 50, "ratio": 0.537},
                                 {"N": 4096, "Na": 8, "Nz": 16, "sweeps": 5, "ratio": 0.487}],
                       "digits": 3},
        "nowall_N256": {"cite": "SURVEY.md 8c (vi): SMC_noMPI_noWall.c, N=256, rho=0.1, L=cbrt(2560)",
                        "E0": -36.1886050855328, "P0": -0.0281510305248713, "energySingle0": -0.282694180641838,
                        "energySingle1": -0.262872304878899, "acceptance_500": 0.980},
        "rand_12345": {"cite": "SURVEY.md 8a row R lists these three outputs of srand(12345) "
                               "(as a set; libc's call order is 383100999, 858300821, 357768173)",
                       "values": [357768173, 858300821, 383100999]},
    }
    json.dump(pins, open(os.path.join(HERE, "reference_pins.json"), "w"), indent=1)
    print("golden files written")


if __name__ == "__main__":
    main()
