# round-4 session 6 (through gpurun, repo root): sweep_kernel_ml16 (the one-wavefront merged kernel with the cells' positions
# in LDS) against the oracle and against the two-team kernel at config 2
set -o pipefail
timeout -k 10 300 python - <<'PY' > gpurun_out/r04_ml16_first.log 2>&1 || { tail -20 gpurun_out/r04_ml16_first.log; exit 1; }
import sys, numpy as np
sys.path.insert(0, "tests")
import smcx_loader, oracle_lib as O
S = smcx_loader.load()
rs = np.random.RandomState(3)
cases = [("lattice", 1024, O.fcc(8, 4), 4, 3, True), ("ragged", 1000, O.fcc(8, 4)[:3000], 3, 3, True), ("no walls", 1024, O.fcc(8, 4), 2, 3, False)]
Rd = O.fcc(8, 4).reshape(-1, 3).copy(); Rd[:, 2] *= 0.3; Rd[:, 2] += -118.6 - Rd[:, 2].min()          # dense film at the lower wall
cases.append(("film at the wall", 1024, Rd.ravel(), 2, 2, True))
for name, N, R0, nrep, nsw, walls in cases:
    fl = (S.FLAG_WALLS if walls else 0) | S.FLAG_SERIES
    p = S.default_params(N, nrep, flags=fl, tune_slots=16, tune_waves=1)
    with S.Engine(p) as eng:
        kn = eng.kernel_form[1]
        eng.upload(R0, O.W_FIXTURE)
        eng.run(1, nsw, 1)
        Es, jj = eng.series(nsw)
        Rg = eng.positions()
    s = O.make_sys(N) if walls else O.make_sys(N, a0=0.0, b0=0.0)
    W = O.W_FIXTURE if walls else np.zeros_like(O.W_FIXTURE)
    for r in range(nrep):
        ref = O.chain(s, 12345 + r, R0, W, 1.1, 1.1, 1, nsw, 1, e0_restart=False)
        ok = np.array_equal(jj[r], ref["jj"]) and np.abs(Es[r] - ref["E"]).max() <= 1e-9 * (1 + np.abs(ref["E"]).max()) and np.abs(Rg[r] - ref["R"]).max() < 1e-8
        print(name, kn, r, "jj", jj[r], ref["jj"], "dE %.2e" % np.abs(Es[r] - ref["E"]).max(), "OK" if ok else "MISMATCH", flush=True)
PY
cat gpurun_out/r04_ml16_first.log
grep -q MISMATCH gpurun_out/r04_ml16_first.log && exit 1
for args in "--N 1024 --replicas 1024" "--N 1024 --replicas 1024 --slots 16 --waves 1" "--N 1024 --replicas 1024" "--N 1024 --replicas 1024 --slots 16 --waves 1" "--N 1024 --replicas 512 --slots 16 --waves 1" "--N 1024 --replicas 768 --slots 16 --waves 1"; do
python bench.py --no-cpu --steps 40 --warmup 4 $args 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); r = j['roofline']
        print('%-50s %8.4f ms/step  %.4e  sweep %.4f ms  %s' % ('$args', j['ms_per_step'], j['value'], r['ms_per_sweep'], r['kernel']))
"
done | tee gpurun_out/r04_config2_ml16.txt
