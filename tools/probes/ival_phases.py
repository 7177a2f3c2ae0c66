#!/usr/bin/env python3
"""ival_phases.py -- where a move of sweep_kernel_mc64 spends its cycles, per WAVEFRONT: the interval variants built by
tools/probes/ival_phases.sh (one library per interval: the cycles between two points of every move summed per replica, two
s_memtime reads per move as the only change) at four wavefronts per SIMD (4096 replicas) and at one (1024).  A launch is two sweeps
(the z sort's period); the clock rows hold the last launch: column 0 start, 2 end (s_memtime), 3 the interval's sum.  Through gpurun."""
import glob, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import subprocess
import numpy as np

NAMES = {"0_31": "move head + screen A", "31_32": "screen B", "32_21": "pass set-up: hand-overs, until the candidates' fetch is issued",
         "21_22": "fetch issued -> positions arrived (wall dz and masks in between)", "22_23": "fp64 body + side capture",
         "23_24": "reduction of the eight sums", "24_8": "Metropolis step + accept path", "8_10": "Fm of n+1, proposal, row test",
         "0_10": "(check) first to last point"}
if len(sys.argv) > 1 and sys.argv[1] == "--one":
    import torch  # noqa: F401
    import smcx_loader
    S = smcx_loader.load()
    N = 4096
    for nrep in (4096, 1024):
        p = S.default_params(N, nrep)
        with S.Engine(p) as e:
            e.upload(S.fcc_init(8, 16), S.W_REFERENCE)
            e.run(0, 2, 10)
            e.run(0, 4, 10)
            ms, nl = e.last_kernel_ms()
            rows = e.clk_rows(4).astype(np.float64)
            assert e.kernel_form[1].endswith("mc64"), e.kernel_form
        moves = 2.0 * N                          # the last launch: two sweeps
        total = (rows[:, 2] - rows[:, 0]) / moves
        iv = (rows[:, 3] % 2 ** 32) / moves
        print("%s %d %.1f %.1f %.1f %.4f" % (os.environ.get("IV_TAG", "?"), nrep, total.mean(), iv.mean(), iv.std(), ms / 4), flush=True)
    sys.exit(0)
res = {}
for lib in sorted(glob.glob(os.path.join(ROOT, "montecarlo-surfacer_amd", "libsmcx_iv_*.so"))):
    tag = os.path.basename(lib)[len("libsmcx_iv_"):-3]
    out = subprocess.run([sys.executable, os.path.abspath(__file__), "--one"], env=dict(os.environ, SMCX_LIB=lib, IV_TAG=tag),
                         capture_output=True, text=True)
    for ln in out.stdout.splitlines():
        f = ln.split()
        if len(f) == 6 and f[0] == tag:
            res[(tag, int(f[1]))] = [float(x) for x in f[2:]]
    if out.returncode:
        print(tag, "failed:", out.stderr[-500:])
for nrep, label in ((4096, "four wavefronts per SIMD (4096 replicas)"), (1024, "one wavefront per SIMD (1024 replicas)")):
    print("== N = 4096, %s: cycles per move of ONE wavefront (mean over replicas; +- = std over replicas)" % label)
    s = 0.0
    for tag in ("0_31", "31_32", "32_21", "21_22", "22_23", "23_24", "24_8", "8_10", "0_10"):
        if (tag, nrep) not in res:
            continue
        total, iv, sd, ms = res[(tag, nrep)]
        if tag != "0_10":
            s += iv
        print("   %-68s %7.0f +- %4.0f   (this variant: %6.0f per move in all, %.3f ms per sweep)" % (NAMES[tag], iv, sd, total, ms))
    print("   %-68s %7.0f   (the rest of a move: loop end, row fills, the points' own cost)" % ("sum of the eight intervals", s), flush=True)
