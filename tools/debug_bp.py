import sys, numpy as np, torch
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
from mfem_amd import capi
from mfem_amd.bp import BakeoffProblem
from mfem_amd.tables import bakeoff_tables
import oracle_lib as orc
capi.call("mfem_amd_init",0)
prob, p, n = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
bp = BakeoffProblem(prob, p, (n,n,n), "cuda:0")
kind,D,Q,B,G,Bn,Gn,W = bakeoff_tables(prob,p)
pr = bp.prob; ne=pr.ne
J=np.zeros(9*Q**3*ne); orc.lib.orc_jacobians_3d(2,Q,ne,Bn,Gn,pr.nodes_E,J)
if kind:
    pa=np.zeros(6*Q**3*ne); orc.lib.orc_diffusion_setup_3d(Q,1,ne,W,J,np.ones(1),1,pa)
else:
    det=np.zeros(Q**3*ne); orc.lib.orc_det_3d(Q,ne,J,det); pa=np.zeros(Q**3*ne); orc.lib.orc_mass_setup_3d(Q,ne,W,det,np.ones(1),1,pa)
op=orc.Operator(kind,D,Q,ne,pr.ndofs,B,G,pa,pr.gather_map,pr.offsets,pr.indices,bp.ess_host)
x=np.random.default_rng(0).random(pr.ndofs)
xd=torch.from_numpy(x).cuda(); yd=torch.empty_like(xd); bp.mult(xd,yd); torch.cuda.synchronize()
yo=op.mult(x); print("mult err", np.abs(yd.cpu().numpy()-yo).max()/np.abs(yo).max())
dg=op.diag(); bp.dinv(); torch.cuda.synchronize(); print("diag err", np.abs(bp.diag.cpu().numpy()-dg).max()/np.abs(dg).max())
b=bp.rhs_unit_source(); torch.cuda.synchronize(); print("sum b", float(b.sum()), "min b", float(b.min()))
dinv=orc.jacobi_setup(dg,bp.ess_host)
Xo,its,hist,conv=op.pcg(b.cpu().numpy(),dinv,1e-12)
X,res,h=bp.pcg(b,1e-12,want_hist=True); torch.cuda.synchronize()
print("oracle its",its,"|X|",np.linalg.norm(Xo)," gpu its",res.iterations,"|X|",float(torch.linalg.vector_norm(X)), "conv", res.converged)
print(hist[:4], h[:4])
