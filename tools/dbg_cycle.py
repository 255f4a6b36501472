import numpy as np, sys
sys.path.insert(0,'/root/repo')
exec(open('/root/repo/tools/dbg_port.py').read().split("vdd=float")[0])
raw=np.fromfile('/tmp/port_dump.bin')
n=265
recs=[]
i=0
while i < len(raw):
    hdr=raw[i:i+4]; u=raw[i+4:i+4+n]; beta=raw[i+4+n:i+4+2*n]; recs.append((hdr,u,beta)); i+=4+2*n
(hA,uA,bA),(hB,uB,bB)=recs[0],recs[1]
print(hA,hB, np.abs(bA-bB).max())
circ=bm.dff_circuit(); st,port=make_port(circ,{"vdd":5.5},125.0,"tran")
tn,h,a0,_=hA
import scipy.sparse as sp
def F(u):
    G,C,b,lw=port.rebuild(u,tn)
    Gm=sp.csr_matrix((G,st.colidx,st.rowptr),shape=(n,n)); Cm=sp.csr_matrix((C,st.colidx,st.rowptr),shape=(n,n))
    return Cm@(a0*u+bA)+Gm@u-b, Gm+a0*Cm
# note uA is the iterate AFTER update at k=8 → next stamping point; uB after k=9
FA,JA=F(uA); FB,JB=F(uB)
names=st.node_names+st.current_names+st.charge_names+st.limit_names
d=uB-uA
idx=np.argsort(-np.abs(d))[:8]
print("largest differences between cycle points:", [(names[i], uA[i], uB[i]) for i in idx])
# scan along the segment
import scipy.sparse.linalg as spla
print("Newton from A:", np.abs(spla.spsolve(JA.tocsc(),FA)).max(), "from B", np.abs(spla.spsolve(JB.tocsc(),FB)).max())
prev=None
for s in np.linspace(-0.2,1.2,57):
    u=uA+s*d; Fs,J=F(u)
    line=""
    if prev is not None:
        jump=np.abs(Fs-prev); k=np.argsort(-jump)[:3]
        line=" jumps: "+", ".join("%s %.3e"%(names[i],jump[i]) for i in k)
    prev=Fs
    print(round(s,3), "max|F| %.3e"%np.abs(Fs).max(), names[np.argmax(np.abs(Fs))], line)
import scipy.linalg as sl
for nm,J in (("A",JA),("B",JB)):
    lu=spla.splu(J.tocsc())
    d=lu.U.diagonal(); 
    sign=np.prod(np.sign(d))*(-1)**( (np.sum(lu.perm_r!=np.arange(n)) ) )
    print(nm, "logabsdet", np.sum(np.log(np.abs(d))), "neg pivots", np.sum(d<0))
# which entries of J differ most
D=(JA-JB).tocoo()
k=np.argsort(-np.abs(D.data))[:12]
for i in k: print(names[D.row[i]], names[D.col[i]], JA[D.row[i],D.col[i]], JB[D.row[i],D.col[i]])
# tp7 voltages
for nm in ("net0","cki","Q_internal","VNW"):
    print(nm, uA[st.index_of(nm)], uB[st.index_of(nm)])
