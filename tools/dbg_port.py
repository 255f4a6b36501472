import numpy as np, sys, time
sys.path.insert(0,'/root/repo')
import cadnip_jl_amd as cj
from cadnip_jl_amd import api, benchmarks as bm, hip
from cadnip_jl_amd.structure import expand_breakpoints, TYPE_ID
from oracle import cpu_port
from oracle import mna_ref as M
from oracle.netlist_ref import make_builder
import scipy.sparse as sp

def make_port(circ, params, temp=27.0, mode="tran"):
    st = cj.discover(circ, params)
    packed = cj.pack_params(st, circ, {k: np.array([v]) for k,v in params.items()}, np.array([temp]), 1)
    port = cpu_port.Port(st, [p[0] for p in packed], TYPE_ID)
    port.set_spec(mode=mode)
    return st, port

def analyze_port(st, port, vscale, gamma=1e9, seed=1234, n_samples=6):
    rng=np.random.default_rng(seed); acc=np.zeros(st.nnz)
    for k in range(n_samples):
        if k==0:
            u=np.zeros(st.n); u[st.n-st.n_limits:]=st.limit_init; port.set_spec(initjct=1)
        elif k==1: u=np.zeros(st.n)
        else:
            u=(rng.random(st.n)*1.2-0.1)*vscale; u[st.n_nodes:st.n_nodes+st.n_currents]=0
        G,C,b,lw=port.rebuild(u,0.0); port.set_spec(initjct=0)
        acc=np.maximum(acc,np.abs(np.nan_to_num(G+gamma*C)))
    prog=hip.host_lu_analyze(st.n, st.rowptr, st.colidx, acc)
    port.set_lu(prog); return prog

vdd=float(sys.argv[1]) if len(sys.argv)>1 else 5.0
temp=float(sys.argv[2]) if len(sys.argv)>2 else 27.0
circ=bm.dff_circuit(meyer=bool(int(sys.argv[4])) if len(sys.argv)>4 else False); params={"vdd":vdd}
st,port=make_port(circ,params,temp,"tranop")
# check vs python oracle stamps
b=make_builder(circ.to_dicts(params)); spec=M.MNASpec(mode="tranop",temp=temp)
ctx=M.build_with_detection(b,{},spec); cs=M.compile_structure(b,{},spec,ctx=ctx); ws=M.create_workspace(cs,ctx=ctx)
rng=np.random.default_rng(0); u=rng.random(st.n)*5
M.fast_rebuild(ws,u,0.0); G,C,bb,lw=port.rebuild(u,0.0)
Gr=np.empty(st.nnz); Gr[st.to_ref_nz]=G; Cr=np.empty(st.nnz); Cr[st.to_ref_nz]=C
print("stamp parity port vs py:", np.abs(Gr-cs.G.data).max()/np.abs(cs.G.data).max(), np.abs(Cr-cs.C.data).max()/max(np.abs(cs.C.data).max(),1e-300), np.abs(bb-ws.dctx.b).max()/np.abs(ws.dctx.b).max())
analyze_port(st,port,vdd)
t=time.time(); u0,ok,it=port.dc(abstol=1e-9); print("port dc",ok,it,time.time()-t, [(nm,round(u0[st.index_of(nm)],3)) for nm in ("Q","Q_neg","net0","net7")])
breaks=expand_breakpoints(st.breakpoints,bm.DFF_TSPAN)
port.set_spec(mode="tran")
atol=st.state_abstol(vntol=1e-6,iabstol=1e-9,chgtol=1e-6)
ts=np.linspace(0,7e-7,71)
out,uf,stats,trace=port.tran(u0,0.0,7e-7,atol,1e-4,breaks=breaks,save_t=ts,obs=[st.index_of("Q")],err_mask=st.differential_mask(),trace_cap=100000,use_pcnr=bool(int(sys.argv[3])) if len(sys.argv)>3 else True)
print(stats, "t_end", trace[-1] if len(trace) else None)
print(np.round(out[:,0],2))
