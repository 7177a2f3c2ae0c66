#!/usr/bin/env python3
"""tt_list_overflow.py -- evidence for the round-5 fix of the two-team kernel's hand-over list.

Runs the OVERFULL state of tests/test_gpu_rare_paths.py (N = 16384 in L = 18, ~100 candidate bits per wavefront and probe)
(a) through the diagnostic builds -- round 4's generator (list of 64: SMCX_GEN_TTCAP=64, libsmcx_cap64chk.so) and the product's
(63, libsmcx_check.so) -- printing the count of items handed over without a working lane, and (b) through the corresponding
product builds against the ORACLE (one replica, one sweep): accepted count and energy.  Build the variants first:
  make -C montecarlo-surfacer_amd/csrc VARIANT=cap64chk CHECK=1 GENENV="SMCX_GEN_TTCAP=64"
  make -C montecarlo-surfacer_amd/csrc VARIANT=cap64 GENENV="SMCX_GEN_TTCAP=64"
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PKG = os.path.join(ROOT, "montecarlo-surfacer_amd")

WORKER = r"""
import sys, os, json, ctypes as C
root, lib, mode = sys.argv[1:4]
os.environ["SMCX_LIB"] = lib
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np
import smcx_loader
S = smcx_loader.load()
import oracle_lib as O
import test_gpu_rare_paths as T
R0, L, Lz, wmode, kernel, extra = T._tt_state(O, "mt64x8_overfull_no_walls")
N = R0.size // 3
flags = S.FLAG_SERIES | S.FLAG_E0_RESTART
p = S.default_params(N, 1, L=L, Lz=Lz, flags=flags, tune_slots=64, tune_waves=8, **extra)
with S.Engine(p) as eng:
    name = eng.kernel_form[1]
    eng.upload(R0, O.W_FIXTURE)
    eng.run(0, 1, 1)
    E, jj = eng.series(1)
    out = {"lib": os.path.basename(lib), "kernel": name, "accepted": int(jj[0][0]), "E_after": float(E[0][1])}
    if mode == "check":
        cnt = (C.c_uint64 * 8)()
        f = S._lib().smcx_debug_work_counts
        f.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
        assert f(eng._h, cnt) == 0
        out.update(pairs_inside=int(cnt[0]), candidate_bits=int(cnt[1]), missed_by_screen=int(cnt[2]), further_rounds=int(cnt[5]),
                   items_without_working_lane=int(cnt[7]))
if mode == "oracle":
    s = O.make_sys(N, M=p.M, L=L, Lz=Lz, cutoff=p.cutoff, a0=0.0, b0=0.0, Ncx=p.Ncx, Ncz=p.Ncz)
    ref = O.chain(s, 12345, R0, np.zeros_like(O.W_FIXTURE), 1.1, extra["A"], 0, 1, 1)
    out.update(oracle_accepted=int(ref["jj"][0]), oracle_E_after=float(ref["E"][1]),
               equal=bool(int(ref["jj"][0]) == int(jj[0][0]) and abs(ref["E"][1] - E[0][1]) <= 1e-9 * abs(ref["E"][1])))
print(json.dumps(out))
"""

for lib, mode in (("libsmcx_cap64chk.so", "check"), ("libsmcx_check.so", "check"), ("libsmcx_cap64.so", "oracle"), ("libsmcx.so", "oracle")):
    path = os.path.join(PKG, lib)
    if not os.path.exists(path):
        print(json.dumps({"lib": lib, "note": "not built"}))
        continue
    env = dict(os.environ)
    if mode == "check":
        env["SMCX_CHECK_MB"] = "2"
    r = subprocess.run([sys.executable, "-c", WORKER, ROOT, path, mode], env=env, capture_output=True, text=True, timeout=600)
    print(r.stdout.strip().splitlines()[-1] if r.returncode == 0 and r.stdout.strip() else json.dumps({"lib": lib, "failed": r.stderr[-600:]}), flush=True)
