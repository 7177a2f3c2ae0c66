"""the geometry rule for N <= 2048 now that one-wavefront kernels (sweep_kernel_mc16/32) exist: rule's choice vs S = N/64 x 1"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
import smcx_loader
S = smcx_loader.load()
for N, lat in ((1024, (8, 4)), (2048, (8, 8))):
    for nrep in (128, 256, 512, 1024, 2048, 4096):
        row = []
        for slots, waves in ((0, 0), (N // 64, 1)):
            p = S.default_params(N, nrep, tune_slots=slots, tune_waves=waves)
            with S.Engine(p) as e:
                e.upload(S.fcc_init(*lat), S.W_REFERENCE)
                e.run(0, 2, 10)
                e.run(0, 10, 10)
                row.append("%s %dx%d %.3f ms/sweep" % (e.kernel_form[1].replace("smcx::sweep_kernel_", ""), e.geometry[0], e.geometry[1], e.last_run_ms() / 10))
        print("N=%d nrep=%d | rule: %s | one wave: %s" % (N, nrep, row[0], row[1]), flush=True)
