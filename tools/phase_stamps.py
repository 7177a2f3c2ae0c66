"""Per-phase cycle shares of a move in the screened sweep kernel, from the diagnostic build
(make -C montecarlo-surfacer_amd/csrc STAMPS=1; run with SMCX_LIB=.../libsmcx_stamps.so)."""
import sys, os, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch  # noqa: F401
import smcx_loader
S = smcx_loader.load()
N, Na, Nz, nrep, sweeps, s, w = (int(v) for v in sys.argv[1:8])
p = S.default_params(N, nrep, tune_slots=s, tune_waves=w, tune_kernel=2)
print("kernel:", S.Engine(p).kernel_form)
eng = S.Engine(p)
eng.upload(S.fcc_init(Na, Nz), S.W_REFERENCE)
eng.run(0, sweeps, 1000)
out = np.zeros((nrep, 8))
lib = C.CDLL(os.environ["SMCX_LIB"])
lib.smcx_debug_stamps.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
assert lib.smcx_debug_stamps(eng._h, out.ctypes.data_as(C.POINTER(C.c_double))) == 0
ms, _ = eng.last_kernel_ms()
per_move = out.mean(axis=0) / (sweeps * N)
form, kname = eng.kernel_form
if "sweep_kernel_mi" in kname:
    names = ["loop control", "probe B compact", "screen", "unsafe/exclusion bits", "probe A (walls, cand., reduce)",
             "Metropolis + update", "probe B (walls, side, cand., reduce)", "next proposal, rotation"]
else:
    names = ["proposal+probes", "screening", "candidates", "walls/side block", "reduce+Metropolis", "next particle", "-", "-"]
print("N=%d nrep=%d S=%d W=%d: %.1f ms/sweep, %.0f cycles per move (s_memtime), shares:" % (N, nrep, s, w, ms / sweeps, per_move.sum()))
for n_, c in zip(names, per_move):
    print("  %-38s %7.0f cycles  %5.1f %%" % (n_, c, 100 * c / per_move.sum()))
eng.close()
