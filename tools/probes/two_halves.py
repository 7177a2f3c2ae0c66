"""does running the replicas as two independent half-batches (two handles, two streams, chunk phases offset) hide the
helper kernels (rand() pre-pass, z sort) of one half behind the sweep kernel of the other?"""
import sys, os, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
import smcx_loader
S = smcx_loader.load()
NSW = 40

def make(nrep, seed0):
    p = S.default_params(4096, nrep)
    p.base_seed = seed0
    e = S.Engine(p)
    e.upload(S.fcc_init(8, 16), S.W_REFERENCE)
    return e

one = make(4096, 12345)
one.run(0, 2, 10)
torch.cuda.synchronize(); t0 = time.perf_counter()
one.run(0, NSW, 10)
torch.cuda.synchronize(); t1 = time.perf_counter() - t0
print("one handle, 4096 replicas: %.2f ms per sweep" % (t1 * 1e3 / NSW), flush=True)
one.close()

for offset in (0, 5):
    a, b = make(2048, 12345), make(2048, 12345 + 2048)
    a.run(0, 2, 10); b.run(0, 2, 10)
    def ra(): a.run(0, NSW, 10)
    def rb():
        if offset: b.run(0, offset, 10)
        b.run(0, NSW - offset, 10)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ta, tb = threading.Thread(target=ra), threading.Thread(target=rb)
    ta.start(); tb.start(); ta.join(); tb.join()
    torch.cuda.synchronize(); t2 = time.perf_counter() - t0
    print("two handles x 2048 replicas, chunk offset %d: %.2f ms per sweep of all 4096" % (offset, t2 * 1e3 / NSW), flush=True)
    a.close(); b.close()
