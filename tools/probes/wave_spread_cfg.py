"""wave lifetime spread for another configuration: python tools/probes/wave_spread_cfg.py N nrep Na Nz [sweeps]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
import smcx_loader
S = smcx_loader.load()
N, nrep, Na, Nz = (int(v) for v in sys.argv[1:5])
nsw = int(sys.argv[5]) if len(sys.argv) > 5 else 2
slots, waves = (int(sys.argv[6]), int(sys.argv[7])) if len(sys.argv) > 7 else (0, 0)
p = S.default_params(N, nrep, tune_slots=slots, tune_waves=waves)
eng = S.Engine(p); print(eng.kernel_form, eng.geometry, flush=True)
eng.upload(S.fcc_init(Na, Nz), S.W_REFERENCE)
for k in range(2):
    eng.run(0, nsw, 10)
    out = eng.wave_spread()
    ms, n = eng.last_kernel_ms()
    print("replica lifetime us (first wave of each): min %.0f median %.0f max %.0f ; span %.0f ; HIP events %.0f per launch" %
          (out[0], out[1], out[2], out[3], ms * 1e3 / n), flush=True)
