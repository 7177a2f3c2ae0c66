"""Throughput of the sweep kernel for several launch geometries (GPU box, ad hoc)."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401  (one HIP runtime per process)
import smcx_loader
S = smcx_loader.load()

def run(N, lat, nrep, sweeps, slots, waves):
    p = S.default_params(N, nrep, tune_slots=slots, tune_waves=waves)
    try:
        eng = S.Engine(p)
    except S.SmcxError as e:
        print("N=%d S=%d W=%d: %s" % (N, slots, waves, e)); return
    eng.upload(S.fcc_init(*lat), S.W_REFERENCE)
    eng.run(0, 1, 10)
    eng.run(0, sweeps, 10)
    ms, nl = eng.last_kernel_ms()
    ob = eng.observables()
    pe = nrep * sweeps * 2.0 * N * (N - 1) / (ms * 1e-3)
    print("N=%5d nrep=%5d S=%2d W=%2d  %8.1f ms/sweep  %.3e pair-evals/s  acc=%.4f" %
          (N, nrep, slots, waves, ms / sweeps, pe, ob["acceptance_ratio"].mean()), flush=True)
    eng.close()

if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "4096"
    if which == "4096":
        for s, w in ((64, 1), (32, 2), (16, 4), (8, 8), (32, 4), (16, 8)):
            run(4096, (8, 16), 4096, 4, s, w)
    elif which == "1024":
        for s, w in ((16, 1), (8, 2), (4, 4), (16, 2), (32, 1)):
            run(1024, (8, 4), 1024, 20, s, w)
        for s, w in ((16, 1), (8, 2)):
            run(1024, (8, 4), 8192, 5, s, w)
    elif which == "one":   # geom_sweep.py one N Na Nz nrep sweeps S W
        N, Na, Nz, nrep, sweeps, s, w = (int(v) for v in sys.argv[2:9])
        run(N, (Na, Nz), nrep, sweeps, s, w)
    elif which == "16384":
        for s, w in ((32, 8), (16, 16), (32, 16)):
            run(16384, (16, 16), 512, 1, s, w)
