#!/usr/bin/env python3
"""replica_cliff.py [--gl G] [--steps K] -- time per sweep against the replica count (config 3's system, N = 4096), round 5.

A sweep is sequential inside a replica and the device holds 4096 replicas of sweep_kernel_mc64 at once, so until round 4 a launch
of 4097 replicas cost two rounds (1.55 x).  Round 5 runs the sweeps between two gathers as WINDOWS of 4096 (replica, block)
units (csrc/smcx_sweep_ma.hip: MaArgs2).  This probe times both: the product library and the variant library
libsmcx_nowin.so (make -C montecarlo-surfacer_amd/csrc VARIANT=nowin) with SMCX_NO_WINDOWS=1 (plain launches, round 4's
behaviour), each replica count in a child process; and, for few replicas, the one-wavefront plan of N = 4096 against the
four-wavefront kernel on an all but identical system (N = 4160: sweep_kernel_mc32x4), per move."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
WORKER = r"""
import sys, os, json
root, N, nrep, gl, steps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
sys.path.insert(0, root)
import numpy as np
import smcx_loader
S = smcx_loader.load()
R0 = S.fcc_init(8, 16) if N == 4096 else S.fcc_init(9, 13)[:3 * N]
p = S.default_params(N, nrep)
with S.Engine(p) as e:
    e.upload(R0, S.W_REFERENCE)
    e.run(0, 2, gl)
    e.run(0, steps, gl)
    ms, launches = e.last_kernel_ms()
    print(json.dumps({"N": N, "replicas": nrep, "kernel": e.kernel_form[1], "granule": e.replica_granule()[0], "gather_lapse": gl,
                      "steps": steps, "sweep_kernel_ms_per_sweep": ms / steps, "device_ms_per_sweep": e.last_run_ms() / steps,
                      "sweep_launches": launches, "us_per_move": ms / steps / N * 1e3,
                      "pair_evals_per_s": nrep * steps * 2.0 * N * (N - 1) / (e.last_run_ms() * 1e-3)}))
"""
gl = int(sys.argv[sys.argv.index("--gl") + 1]) if "--gl" in sys.argv else 10
steps = int(sys.argv[sys.argv.index("--steps") + 1]) if "--steps" in sys.argv else 20


def run(lib, N, nrep, env_extra):
    env = dict(os.environ, SMCX_LIB=os.path.join(ROOT, "montecarlo-surfacer_amd", lib), **env_extra)
    r = subprocess.run([sys.executable, "-c", WORKER, ROOT, str(N), str(nrep), str(gl), str(steps)], env=env, capture_output=True, text=True, timeout=900)
    if r.returncode != 0:
        return {"replicas": nrep, "failed": r.stderr[-300:]}
    return json.loads(r.stdout.strip().splitlines()[-1])


base = None
for nrep in (64, 512, 2048, 4096, 4097, 4608, 6144, 8192, 9000):
    a = run("libsmcx.so", 4096, nrep, {})
    b = run("libsmcx_nowin.so", 4096, nrep, {"SMCX_ALLOW_ENV_TUNING": "1", "SMCX_NO_WINDOWS": "1"}) if nrep > 4096 and nrep % 4096 else None
    if nrep == 4096:
        base = a["device_ms_per_sweep"]
    a["plain_launches_device_ms_per_sweep"] = b and b.get("device_ms_per_sweep")
    a["vs_4096_replicas"] = base and a["device_ms_per_sweep"] / base
    print(json.dumps(a), flush=True)
for nrep in (64, 512):
    print(json.dumps(run("libsmcx.so", 4160, nrep, {})), flush=True)      # four wavefronts per replica on (almost) the same system
