import sys, time; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np
from oracle.bindings import Oracle
from x265_amd.engine import *
from util import *
orc=Oracle()
W,H=320,192
for depth in (8,10):
    e=Engine(0,depth)
    fenc,ref,org,stride=clip_planes(W,H,depth)
    pf=e.upload_plane(fenc,W,H,MARGIN,MARGIN,stride); pr=e.upload_plane(ref,W,H,MARGIN,MARGIN,stride)
    t=e.cost_table(20); print('cost table eq', np.array_equal(t, orc.cost_table(20)))
    for mname,method in (('dia',0),('hex',1),('star',3),('full',4)):
        for subme in (0,1,2,5,7):
            rng=np.random.default_rng(100+method*10+subme)
            jobs=random_jobs(rng, 150 if method!=4 else 40, W,H,method)
            t0=time.time(); r=e.search_batch(pf,pr,jobs,20,method,subme); t1=time.time()
            mv,cost,sc,qc=oracle_search(orc,depth,fenc,[ref],org,stride,jobs,20,method,subme)
            ok=np.array_equal(r['mv'],mv) and np.array_equal(r['cost'],cost)
            okc=np.array_equal(r['sad_calls'],sc) and np.array_equal(r['subpel_calls'],qc)
            print(depth,mname,subme,'OK' if ok else 'MISMATCH','counts',okc,'%.3fs'%(t1-t0), flush=True)
            if not ok:
                bad=np.flatnonzero((r['mv']!=mv).any(1)|(r['cost']!=cost))
                for b in bad[:5]: print('  job',jobs[b],'gpu',r[b],'cpu',mv[b],cost[b],sc[b],qc[b])
    e.close()
