import sys; sys.path.insert(0,"."); sys.path.insert(0,"tests")
import numpy as np, torch, smcx_loader
S=smcx_loader.load()
for N,lat,nrep in ((32768,(16,32),128),(8192,(16,8),512),(20000,(16,32),64)):
    R0=S.fcc_init(*lat)[:3*N]
    p=S.default_params(N,nrep,flags=S.FLAGS_REFERENCE|S.FLAG_SERIES)
    try:
        eng=S.Engine(p)
    except S.SmcxError as e:
        print(N,"create:",e); continue
    eng.upload(R0,S.W_REFERENCE); eng.run(0,2,1)
    E,jj=eng.series(2); Et=eng.total_energy(); ob=eng.observables()
    print(N,nrep,eng.geometry[:2],eng.kernel_form[1],"incremental-vs-recomputed rel",np.max(np.abs(E[:,-1]-Et)/np.abs(Et)),"acc",ob["acceptance_ratio"].mean(),"zhist",ob["zhist"][0].sum()==2*N)
    eng.close()
