#!/usr/bin/env python3
"""raw m0 words of the interval / count variants (tools/probes/ival_phases.sh): SMCX_LIB=... python tools/probes/ival_debug.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch  # noqa: F401
import smcx_loader
S = smcx_loader.load()
p = S.default_params(4096, 1024)
with S.Engine(p) as e:
    e.upload(S.fcc_init(8, 16), S.W_REFERENCE)
    e.run(0, 4, 10)
    rows = e.clk_rows(4)
    print(os.path.basename(S.LIB_PATH), e.kernel_form, "m0 of replicas 0..5:", [int(x) & 0xffffffff for x in rows[:6, 3]],
          "cycles of the launch:", [int(rows[i, 2] - rows[i, 0]) for i in range(3)], flush=True)
