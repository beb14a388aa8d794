"""profiling experiment: a recursion-shaped batch (thousands of ~46 bp regions x 200 genomes)"""
import sys,time,os; sys.path.insert(0,"."); sys.path.insert(0,"tests")
import numpy as np
from parsnp_amd import synth
from parsnp_amd.binding import Lib, Session
G=200; n=500_000; NR=8000
ref,gs=synth.population(seed=5,n=n,n_genomes=G,div=0.02,indel_frac=0.0)
rng=np.random.default_rng(1)
starts=np.zeros((NR,G+1),np.int64); lens=np.zeros((NR,G+1),np.int64); mins=np.full(NR,7,np.int32)
for r in range(NR):
    st=int(rng.integers(0,n-200)); ln=int(rng.integers(31,80))
    starts[r,:]=st; lens[r,:]=ln
with Session(Lib(sys.argv[1] if len(sys.argv) > 1 else None),[ref]+gs) as s:
    s.multi_mum_batch(starts,lens,mins)
    for dbg in ("0","1","4","2"):
        os.environ["PM_DEBUG_SEED"]=dbg
        try: out=s.multi_mum_batch(starts,lens,mins)
        except Exception as e: print(e)
        t=dict(s.last_timing()); print("debug",dbg,{k:round(v,2) for k,v in t.items()})
