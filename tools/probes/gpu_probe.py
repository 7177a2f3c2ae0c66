"""Ad-hoc first GPU probe (not a pytest file): chains vs oracle."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import numpy as np
import oracle_lib as O
import smcx_loader
S = smcx_loader.load()

def run_case(Na, Nz, nrep, sweeps, slots=0, waves=0, eq=0, gl=1):
    R0 = O.fcc(Na, Nz); N = len(R0)//3
    p = S.default_params(N, nrep, tune_slots=slots, tune_waves=waves)
    eng = S.Engine(p)
    eng.upload(R0, O.W_FIXTURE)
    s = O.make_sys(N)
    E0 = O.total_energy(s, R0, O.W_FIXTURE)
    e0 = eng.total_energy()
    t=time.time(); eng.run(eq, sweeps, gl); dt=time.time()-t
    ob = eng.observables(); ms,_ = eng.last_kernel_ms()
    worst = 0
    for r in range(min(nrep, 3)):
        ref = O.chain(s, 12345 + r, R0, O.W_FIXTURE, 1.1, 1.1, eq, sweeps, gl)
        dE = abs(ob['meanE'][r]-ref['meanE'])/abs(ref['meanE'])
        da = abs(ob['acceptance_ratio'][r]-ref['acceptance_ratio'])
        dz = int(np.abs(ob['zhist'][r].astype(np.int64)-ref['zhist'].astype(np.int64)).sum())
        dR = np.abs(eng.positions()[r]-ref['R']).max()
        print(f"  rep{r}: relE={dE:.2e} dacc={da:.2e} dz={dz} dR={dR:.2e} acc={ref['acceptance_ratio']:.3f} Elast={ob['E_last'][r]:.12g}/{ref['Efinal']:.12g}")
        worst = max(worst, dE)
    pe = nrep*sweeps*2.0*N*(N-1)/(ms*1e-3) if ms>0 else 0
    print(f"N={N} nrep={nrep} geom={eng.geometry} E0 rel={abs(e0[0]-E0)/abs(E0):.2e} kernel_ms={ms:.2f} wall={dt:.2f}s pair-evals/s={pe:.3e} worst relE={worst:.2e}", flush=True)
    eng.close()

if __name__ == "__main__":
    print("devices", S.device_count())
    run_case(4, 4, 4, 5)              # N=256, S=4 WPR=1
    run_case(4, 4, 4, 5, slots=16, waves=4)   # padded multi-wave
    run_case(8, 4, 4, 5)              # N=1024 S=16 WPR=1
    run_case(8, 4, 4, 5, slots=16, waves=2)
    run_case(8, 4, 4, 5, slots=16, waves=4)
    run_case(8, 4, 4, 10, eq=3, gl=2)
    run_case(8, 16, 2, 2)             # N=4096 auto
    run_case(8, 4, 1024, 10)
    run_case(8, 16, 512, 4)
    run_case(8, 16, 512, 4, slots=16, waves=4)
    run_case(8, 16, 512, 4, slots=64, waves=1)
