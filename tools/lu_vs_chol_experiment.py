"""Why an "LU-equivalent inverse mode" cannot bring the Kronecker conditional to 1e-6 of the literal oracle on ill-conditioned factors:
the FACTORED algebra evaluated with the oracle's OWN inverse (np.linalg.inv, LAPACK LU as tf.matrix_inverse scripts/onoff.py:192) is as far
from the literal dense order (np_kron(inv) @ Kmn, :206-211) as the Cholesky-based one -- the gap is cond * eps through a different but
mathematically equal op order, not LU vs Cholesky.  Run: python tools/lu_vs_chol_experiment.py (CPU, seconds)."""
import sys, numpy as np
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'oracle')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import zigp_oracle as o
from test_gpu_kron import make_kron_problem, ELL_T_HARD, _mp_kron_inf
from scipy.linalg import cholesky, solve_triangular, lu_factor, lu_solve

def gj_inv(A):
    n=A.shape[0]; M=np.hstack([A.copy(), np.eye(n)])
    for k in range(n):
        p=k+np.argmax(np.abs(M[k:,k]))
        if p!=k: M[[k,p]]=M[[p,k]]
        M[k]=M[k]/M[k,k]
        for i in range(n):
            if i!=k: M[i]=M[i]-M[i,k]*M[k]
    return M[:,n:]
def lu_inv_doolittle(A):
    # unblocked right-looking LU with partial pivoting, then solve for identity columns
    n=A.shape[0]; LU=A.copy(); piv=np.arange(n)
    for k in range(n):
        p=k+np.argmax(np.abs(LU[k:,k]))
        if p!=k: LU[[k,p]]=LU[[p,k]]; piv[[k,p]]=piv[[p,k]]
        LU[k+1:,k]/=LU[k,k]
        LU[k+1:,k+1:]-=np.outer(LU[k+1:,k],LU[k,k+1:])
    I=np.eye(n)[piv]
    Y=np.zeros((n,n))
    for i in range(n): Y[i]=I[i]-LU[i,:i]@Y[:i]
    Xs=np.zeros((n,n))
    for i in range(n-1,-1,-1): Xs[i]=(Y[i]-LU[i,i+1:]@Xs[i+1:])/LU[i,i]
    return Xs
def chol_inv(A):
    L=cholesky(A,lower=True); W=solve_triangular(L,np.eye(A.shape[0]),lower=True); return W.T@W

def kron_inf_with(inv, Xnew, Z_list, ell_list, var_list, q_mu, q_sqrt, jitter):
    Kmm=[o.rbf_K(Z_list[p],None,ell_list[p],var_list[p])+np.eye(Z_list[p].shape[0])*jitter for p in range(2)]
    P=[inv(K) for K in Kmm]
    M0,M1=P[0].shape[0],P[1].shape[0]
    U=q_mu.reshape(M0,M1); S2=(q_sqrt**2).reshape(M0,M1)
    k0=o.rbf_K(Z_list[0],Xnew[:,:2],ell_list[0],var_list[0]); k1=o.rbf_K(Z_list[1],Xnew[:,2:],ell_list[1],var_list[1])
    Al=P[0]@U@P[1]; a0=P[0]@k0; a1=P[1]@k1
    mu=np.einsum('in,ij,jn->n',k0,Al,k1)
    var=float(var_list[0])*float(var_list[1])-(k0*a0).sum(0)*(k1*a1).sum(0)+np.einsum('in,ij,jn->n',a0**2,S2,a1**2)
    return mu,var,[np.linalg.cond(K) for K in Kmm]
X,Y,p=make_kron_problem(700,32,32,seed=700,ell_t=ELL_T_HARD)
ref=o.kron_build_predict(X,p,1e-5,0.0)
rel=lambda a,b: np.max(np.abs(a-b))/np.max(np.abs(b))
for name,inv in (('np.inv',np.linalg.inv),('gauss-jordan',gj_inv),('doolittle',lu_inv_doolittle),('cholesky',chol_inv)):
    for tag,(im,iv) in (('f',(3,4)),('g',(5,6))):
        mu,var,cond=kron_inf_with(inv,X,p['Z'+tag],p['ell_'+tag],[float(np.squeeze(v)) for v in p['var_'+tag]],p['u_%sm'%tag],p['u_%ss_sqrt'%tag],1e-5)
        print('%-13s %s mean vs oracle %.2e  var vs oracle %.2e  cond %s'%(name,tag,rel(mu,ref[im].reshape(-1)),rel(var,ref[iv].reshape(-1)),['%.1e'%c for c in cond]))
