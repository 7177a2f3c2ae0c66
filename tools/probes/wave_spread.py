"""spread of the wavefronts' lifetimes inside one launch of the sweep kernel (bench workload)"""
import sys, os, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
import smcx_loader
S = smcx_loader.load()
p = S.default_params(4096, 4096)
eng = S.Engine(p); print(eng.kernel_form, flush=True)
eng.upload(S.fcc_init(8, 16), S.W_REFERENCE)
for k in range(3):
    eng.run(0, 4, 10)
    out = eng.wave_spread()
    ms, n = eng.last_kernel_ms()
    print("wave lifetime us: min %.0f median %.0f max %.0f ; first start to last end %.0f ; HIP events %.0f per launch" %
          (out[0], out[1], out[2], out[3], ms * 1e3 / n), flush=True)
