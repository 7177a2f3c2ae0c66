# round-4 session 3 (through gpurun, repo root): the merged-pass variant (libsmcx_mg.so = SMCX_GEN_MERGE=1) against the oracle:
# one-wavefront kernels only; every step under its own timeout (a first run of new hand-written code)
set -o pipefail
export SMCX_LIB=$PWD/montecarlo-surfacer_amd/libsmcx_mg.so
export SMCX_CHECK_LIB=$PWD/montecarlo-surfacer_amd/libsmcx_mgc.so
timeout -k 10 120 python - <<'PY' > gpurun_out/r04_mg_first.log 2>&1 || { tail -20 gpurun_out/r04_mg_first.log; exit 1; }
import sys, numpy as np
sys.path.insert(0, "tests")
import smcx_loader, oracle_lib as O
S = smcx_loader.load()
print(S.LIB_PATH)
for N, lat, nrep, nsw in ((4096, (8, 16), 4, 2), (1024, (8, 4), 4, 3), (2048, (8, 8), 4, 2)):
    R0 = O.fcc(*lat)
    p = S.default_params(N, nrep, flags=S.FLAG_WALLS | S.FLAG_SERIES, tune_slots={4096: 64, 2048: 32, 1024: 16}[N], tune_waves=1)
    with S.Engine(p) as eng:
        print(N, eng.kernel_form, flush=True)
        eng.upload(R0, O.W_FIXTURE)
        eng.run(0, nsw, 1)
        Es, jj = eng.series(nsw)
        ob = eng.observables()
    s = O.make_sys(N)
    for r in range(nrep):
        ref = O.chain(s, 12345 + r, R0, O.W_FIXTURE, 1.1, 1.1, 0, nsw, 1)
        print(N, r, "jj", jj[r], ref["jj"], "dE", np.abs(Es[r] - ref["E"]).max(), flush=True)
PY
cat gpurun_out/r04_mg_first.log
