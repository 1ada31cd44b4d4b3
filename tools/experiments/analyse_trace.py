import numpy as np, sys
from collections import defaultdict
cases = [a.split(':') for a in sys.argv[1:]] or [('128_32', 8), ('192_64', 12)]      # name:chunks
for name,ncp in cases:
    ncp = int(ncp)
    t=np.load('gpurun_out/trace_%s.npy'%name).astype(np.int64)
    t=t[t[:,2]!=0]; n=len(t)
    hw=t[:,0]; xcc=t[:,1]&0xf
    cu=(hw>>8)&0xf; se=(hw>>13)&0x7; sh=(hw>>12)&1
    cuid=xcc*1000+se*100+sh*16+cu
    rt0,rt1=t[:,126],t[:,127]
    ts=t[:,2:126]
    nst=5*ncp+2
    T=ts[:,:nst]
    tot=T[:,nst-1]-T[:,0]
    clk=tot/((rt1-rt0)/100e6)/1e9
    print(name,'nwg',n,'CUs',len(set(cuid)),'kernel realtime span %.1f us'%((rt1.max()-rt0.min())/100.), 'shader clock GHz: median %.2f  p10 %.2f p90 %.2f'%(np.median(clk),np.percentile(clk,10),np.percentile(clk,90)))
    ph={}
    for i,nm in enumerate(['issue','wait','bar1','mfma','bar2']):
        ph[nm]=np.array([T[:,5*c+i+1]-T[:,5*c+i] for c in range(ncp)]).T
    epi=T[:,nst-1]-T[:,nst-2]
    for nm,a in ph.items():
        print('  %-6s mean %7.0f  chunk0 %7.0f  median %7.0f  p90 %7.0f'%(nm,a.mean(),a[:,0].mean(),np.median(a),np.percentile(a,90)))
    print('  epi mean %.0f  tot mean %.0f  (sum/chunk %.0f)'%(epi.mean(),tot.mean(),sum(a.mean() for a in ph.values())))
    d=defaultdict(list)
    for i in range(n): d[cuid[i]].append((rt0[i]-rt0.min(),rt1[i]-rt0.min()))
    cnts=[len(v) for v in d.values()]
    print('  WGs per CU min/mean/max',min(cnts),np.mean(cnts),max(cnts))
    v=sorted(d[list(d.keys())[5]])
    print('  one CU timeline us:',[(round(a/100.,1),round(b/100.,1)) for a,b in v])
