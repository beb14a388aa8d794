import sys,time,os; ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,"tests"))
from parsnp_amd import synth
from parsnp_amd.binding import Lib, Session
ref,gs=synth.population(seed=5,n=5_000_000,n_genomes=40,div=0.02,indel_frac=0.05)
with Session(Lib(sys.argv[1] if len(sys.argv) > 1 else None),[ref]+gs) as s:
    s.whole(25)
    for dbg in (sys.argv[2].split(",") if len(sys.argv) > 2 else ("0","1","4","2")):
        os.environ["PM_DEBUG_SEED"]=dbg
        try: s.whole(25)
        except Exception as e: pass
        t=dict(s.last_timing()); print("debug",dbg,"seed_extend ms",t.get("seed_extend"))
