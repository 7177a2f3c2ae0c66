#!/usr/bin/env python3
"""would two half-size batches on two streams overlap one batch's helper kernels (zsort, rng prepass) with the other's
sweeps?  Two handles of 2048 replicas each, driven by two host threads, against one handle of 4096 replicas (config 3):
   python tools/probes/two_halves_probe.py [sweeps]      (through gpurun)"""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: F401
import smcx_loader
S = smcx_loader.load()
N, sweeps = 4096, int(sys.argv[1]) if len(sys.argv) > 1 else 40
R0 = S.fcc_init(8, 16)


def make(nrep, seed):
    p = S.default_params(N, nrep, base_seed=seed)
    e = S.Engine(p)
    e.upload(R0, S.W_REFERENCE)
    e.run(0, 2, 10)
    return e


one = make(4096, 12345)
t0 = time.perf_counter(); one.run(0, sweeps, 10); t1 = time.perf_counter() - t0
print("one handle, 4096 replicas: %.3f ms per sweep of 4096 replicas (host wall), kernels %.3f" %
      (t1 * 1e3 / sweeps, one.last_kernel_ms()[0] / sweeps), flush=True)
one.close()
for delay in (0.0, 0.004):
    a, b = make(2048, 12345), make(2048, 12345 + 2048)
    def go(e, d):
        time.sleep(d)
        e.run(0, sweeps, 10)
    th = [threading.Thread(target=go, args=(a, 0.0)), threading.Thread(target=go, args=(b, delay))]
    t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    t2 = time.perf_counter() - t0
    print("two handles x 2048 replicas, second started %.0f ms later: %.3f ms per sweep of 4096 replicas (host wall); "
          "each handle's own run %.3f / %.3f ms per sweep, sweep kernels %.3f / %.3f" %
          (delay * 1e3, t2 * 1e3 / sweeps, a.last_run_ms() / sweeps, b.last_run_ms() / sweeps,
           a.last_kernel_ms()[0] / sweeps, b.last_kernel_ms()[0] / sweeps), flush=True)
    a.close(); b.close()
