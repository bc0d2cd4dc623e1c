import sys, numpy as np
import os; ROOT=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0,ROOT)
from riemannhamiltonianmontecarlo_amd import _capi
from riemannhamiltonianmontecarlo_amd.data import synthetic_logreg
o=_capi.RmhmcLib(os.path.join(ROOT,'oracle','librmhmc_oracle_asan.so'))
for flags in (_capi.COMPAT, _capi.COMPAT|_capi.FLAG_ORACLE_LITERAL, 0):
    for (M,D,n) in ((37,1,2),(50,5,3),(203,33,2),(1,2,1)):
        XX,t=synthetic_logreg(M,D,1)
        rs=np.random.RandomState(0)
        with o.context(M,D,n,flags=flags) as ctx:
            ctx.set_data(XX,t)
            w=0.1*rs.randn(n,D); p=rs.randn(n,D)
            ctx.log_posterior(w); ctx.metric(w); ctx.metric_terms(w,p)
            ctx.leapfrog(w,p,0.5,1,2,4)
            ctx.transition(w,rs.randn(n,D),rs.rand(n),rs.randn(n),rs.rand(n))
            ctx.sample(6,2,seed=1)
            if not (flags & _capi.FLAG_ORACLE_LITERAL):
                ctx.chains_init(seed=2); ctx.chains_run(5); ctx.chains_state()
print("asan run ok")
