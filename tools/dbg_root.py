import numpy as np, sys
sys.path.insert(0,'/root/repo')
exec(open('/root/repo/tools/dbg_port.py').read().split("vdd=float")[0])
import scipy.sparse as sp, scipy.sparse.linalg as spla, scipy.optimize as so
raw=np.fromfile('/tmp/port_dump.bin'); n=265
hdr=raw[:4]; u1=raw[4:4+n]; beta=raw[4+n:4+2*n]
tn,h,a0,_=hdr
print("tn,h,a0",tn,h,a0)
circ=bm.dff_circuit(); st,port=make_port(circ,{"vdd":5.5},125.0,"tran")
names=st.node_names+st.current_names+st.charge_names+st.limit_names
def FJ(u):
    G,C,b,lw=port.rebuild(u,tn)
    Gm=sp.csr_matrix((G,st.colidx,st.rowptr),shape=(n,n)); Cm=sp.csr_matrix((C,st.colidx,st.rowptr),shape=(n,n))
    return Cm@(a0*u+beta)+Gm@u-b, (Gm+a0*Cm)
# row scaling: use weights for a meaningful norm
atol=st.state_abstol(vntol=1e-6,iabstol=1e-9,chgtol=1e-6)
u=u1.copy()   # iterate after first update; fine as a start
F,J=FJ(u); print("start |F|",np.abs(F).max())
# Newton with backtracking line search on ||D F||, D = row scaling by initial Jacobian row norms
D=1.0/np.maximum(np.abs(J).max(axis=1).toarray().ravel(),1e-30)
for it in range(60):
    F,J=FJ(u); f0=np.linalg.norm(D*F)
    d=spla.spsolve(J.tocsc(),F)
    lam=1.0
    while lam>1e-8:
        Fn,_=FJ(u-lam*d); fn=np.linalg.norm(D*Fn)
        if fn < (1-1e-4*lam)*f0: break
        lam*=0.5
    u=u-lam*d
    dn=np.sqrt(np.mean((d/(atol+1e-4*np.abs(u)))**2))
    print(it,"lam",lam,"|DF|",fn,"dnorm",dn, names[np.argmax(np.abs(D*Fn))])
    if dn<1e-3 and lam==1.0: break
