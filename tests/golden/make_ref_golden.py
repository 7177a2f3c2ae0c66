#!/usr/bin/env python3
"""Generates tests/golden/ref_smc.json: outputs of the REAL reference functions of SMC.c and
SMC_noMPI_noWall.c (compiled from /root/reference where they lie by oracle/build_ref.sh into
oracle/_ref/libref_smc_N<n>.so / libref_nw_N<n>.so) on the cases of tests/ref_cases.py.
Runs only in the build container (the reference tree is absent on the GPU box).  The file holds
inputs (the case descriptions) and outputs (hex floats, integers, sha256 digests of arrays) --
no reference source text.
Usage: make -C oracle ref && python tests/golden/make_ref_golden.py
"""
import json
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

import ref_cases as RC  # noqa: E402


def main():
    be = RC.RefBackend()
    t0 = time.time()
    out = RC.compute(be, RC.GOLDEN_CASES,
                     progress=lambda c: print("%6.1fs %s" % (time.time() - t0, c), flush=True))
    doc = {"source": "real reference functions (SMC.c:269-1049, 1094-1169; SMC_noMPI_noWall.c:258-723, 787-896; "
                     "matematicose.c) compiled -O2 -ffp-contract=off by oracle/build_ref.sh, glibc %s rand()"
                     % os.confstr("CS_GNU_LIBC_VERSION"),
           "cases": [{"case": c, "expect": e} for c, e in zip(RC.GOLDEN_CASES, out)]}
    with open(os.path.join(HERE, "ref_smc.json"), "w") as f:
        json.dump(doc, f, indent=0)
    print("wrote ref_smc.json: %d cases, %.0f KB" % (len(out), os.path.getsize(os.path.join(HERE, "ref_smc.json")) / 1e3))


if __name__ == "__main__":
    main()
