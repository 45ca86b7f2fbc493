"""Evidence for DESIGN.md: on the real cloth fit (cond(inner) ~ 8e13) scipy.linalg.lstsq (gelsd), which the
reference uses at regressors.py:155, leaves a 7e-5 relative residual and differs by 1.6e-1 from Cholesky, LU, gelsy
and plain SVD solves, which agree with each other.  Run: python tools/gelsd_accuracy_note.py (CPU only)."""
import numpy as np, sys, scipy.linalg as sl
sys.path.insert(0,'/root/repo/tests'); sys.path.insert(0,'/root/repo')
from conftest import relf
from oracle import nk_oracle as O
g=dict(np.load('/root/repo/tests/golden/f6_cloth_known_gain.npz'))
tr,u=g['trajs'],g['inputs']
X=np.hstack([np.vstack((tr[i][:,:-1],u[i][:,:-1])) for i in range(30)]).T.copy(); Y=np.hstack([tr[i][:,1:] for i in range(30)]).T.copy()
np.random.seed(0)
a=O.KoopmanNystromOracle(6,kernel=O.ThreeDimensionalKernel(10,10,10,192),gamma=1e-7,m=100,faithful=True); a.fit(X,Y)
st=a.stages; inner=st['inner']; S=st['S']; K=st['K_mm']; cross=st['cross']; inner_rec=st['inner_rec']; left_rec=st['left_rec']
m=100;p=6
w,V=np.linalg.eigh(K); Sinv=(V/np.sqrt(w))@V.T
Kxo=K-1e-6*np.eye(m)
right=sl.block_diag(Kxo@Sinv,np.eye(p))
def chol_solve(Amat,R,refine=0,ld=False):
    c=sl.cho_factor(Amat); x=sl.cho_solve(c,R)
    for _ in range(refine):
        if ld:
            r=(R.astype(np.longdouble)-Amat.astype(np.longdouble)@x.astype(np.longdouble)).astype(np.float64)
        else:
            r=R-Amat@x
        x=x+sl.cho_solve(c,r)
    return x
sol_ref=sl.lstsq(inner,right)[0]
for refine,ld in ((0,False),(1,False),(3,False),(1,True),(3,True),(6,True)):
    sol=chol_solve(inner,right,refine,ld)
    G=Sinv@(cross@sol)
    sol_rec=chol_solve(inner_rec,S,refine,ld)
    Cm=left_rec@sol_rec
    W=Cm@G
    phi=np.vstack([a.lift(X[:200,:192].T),X[:200,192:].T]); pred=(W@phi).T
    print(refine,ld,'sol %.1e'%relf(sol,sol_ref),'A %.1e'%relf(G[:,:m],a.A),'C %.1e'%relf(Cm,a.C),'W %.1e'%relf(W,a.weights),'pred %.1e'%relf(pred,a.predict(X[:200])))
# eigh-based solve
w2,V2=np.linalg.eigh(inner); sol=(V2/w2)@(V2.T@right); G=Sinv@(cross@sol); print('eigh','sol %.1e'%relf(sol,sol_ref),'A %.1e'%relf(G[:,:m],a.A), 'min eig %.2e max %.2e'%(w2.min(),w2.max()))
print('residual lstsq', np.linalg.norm(inner@sol_ref-right)/np.linalg.norm(right), 'chol+ref', np.linalg.norm(inner@chol_solve(inner,right,3,True)-right)/np.linalg.norm(right))
print('eigs small', w2[:8])
U_,s_,Vt_=np.linalg.svd(inner)
print('svals small', s_[-8:])
for tau in (0,1e-16,2.2e-16,1e-15,1e-14,1.5e-14,3e-14,1e-13):
    keep=s_>tau*s_[0]
    solt=(Vt_[keep].T/s_[keep])@(U_[:,keep].T@right)
    print(tau, keep.sum(), 'vs lstsq %.2e'%relf(solt,sol_ref))
x,res,rank,sv=sl.lstsq(inner,right); print('lstsq rank',rank, 'sv min',sv.min())
x2=sl.lstsq(inner,right,lapack_driver='gelsy')[0]; print('gelsy vs gelsd %.2e'%relf(x2,sol_ref))
x3=np.linalg.solve(inner,right); print('LU vs gelsd %.2e'%relf(x3,sol_ref), 'LU vs chol %.2e'%relf(x3,chol_solve(inner,right)))
