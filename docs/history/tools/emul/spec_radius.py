"""TOOL: lambda_max of the normalised Fisher matrix (the contraction test of the MLE re-fit's `unstable` flag, fit_common.h)
against the drift between the float32 loop and the reference's arithmetic on emulated fits, and for the fuzz residuals."""
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
from math import erf
from study import fuzz_spots, traces
from scipy.special import erf as verf

def fisher_lmax(theta, box, NP):
    """lambda_max of D^-1/2 M D^-1/2 for the expected Fisher matrix of the pixel-integrated Gaussian at theta (n,6)."""
    n=len(theta); idx=np.arange(box)[None,:]
    x0,y0,N,bg,sx=theta[:,0:1],theta[:,1:2],theta[:,2:3],theta[:,3:4],theta[:,4:5]
    sy = theta[:,5:6] if NP==6 else sx
    def terms(mu,s):
        a=(idx-mu+0.5)/(np.sqrt(2)*s); b=(idx-mu-0.5)/(np.sqrt(2)*s)
        E=0.5*(verf(a)-verf(b))
        gp=np.exp(-0.5*((idx-mu+0.5)/s)**2); gm=np.exp(-0.5*((idx-mu-0.5)/s)**2)
        A=(gm-gp)/(np.sqrt(2*np.pi)*s)
        S=((idx-mu-0.5)*gm-(idx-mu+0.5)*gp)/(np.sqrt(2*np.pi)*s*s)
        return E,A,S
    Ex,Ax,Sx=terms(x0,sx); Ey,Ay,Sy=terms(y0,sy)
    # du[k][n, j(row), i(col)]
    du=[N[:,:,None]*Ey[:,:,None]*Ax[:,None,:], N[:,:,None]*Ay[:,:,None]*Ex[:,None,:], Ey[:,:,None]*Ex[:,None,:], np.ones((n,box,box))]
    if NP==6:
        du+= [N[:,:,None]*Ey[:,:,None]*Sx[:,None,:], N[:,:,None]*Sy[:,:,None]*Ex[:,None,:]]
    else:
        du+= [N[:,:,None]*(Ey[:,:,None]*Sx[:,None,:]+Sy[:,:,None]*Ex[:,None,:])]
    model=N[:,:,None]*Ey[:,:,None]*Ex[:,None,:]+bg[:,:,None]
    K=len(du)
    M=np.empty((n,K,K))
    for a in range(K):
        for b in range(a,K):
            M[:,a,b]=M[:,b,a]=(du[a]*du[b]/model).sum(axis=(1,2))
    d=np.sqrt(np.einsum('nii->ni',M))
    C=M/(d[:,:,None]*d[:,None,:])
    ok=np.isfinite(C).all(axis=(1,2))
    lm=np.full(n,np.nan)
    lm[ok]=np.linalg.eigvalsh(C[ok])[:,-1]
    return lm

if __name__ == '__main__':
    rng=np.random.default_rng(17)
    for style,box,method in (("real",7,"sigmaxy"),("real",7,"sigma"),("real",13,"sigmaxy"),("fuzz",13,"sigmaxy"),("fuzz",21,"sigma"),("fuzz",7,"sigmaxy")):
        n=20000
        spots=fuzz_spots(box,n,rng,style)
        tr,ir,tf,aux,itf=traces(spots,1e-3,40,method,T=41)
        NP=6 if method=="sigmaxy" else 5
        thf=tf[np.arange(n),np.minimum(itf,40)].astype(np.float64)
        lm=fisher_lmax(thf,box,NP)
        d=np.abs(tf[:,:,:NP].astype(np.float64)-tr[:,:,:NP])/np.maximum(np.abs(tr[:,:,:NP]),1.0)
        K=np.minimum(np.minimum(itf,ir),40)
        drift=d[np.arange(n),K].max(axis=1)
        conv=(itf<40)&np.isfinite(lm)
        print(style,box,method,"n conv",conv.sum(),"lmax quantiles 50/90/99/99.9:",np.round(np.nanquantile(lm[conv],[.5,.9,.99,.999]),3),
              "frac>1.8 %.4f >1.9 %.4f >2.0 %.4f"%(np.mean(lm[conv]>1.8),np.mean(lm[conv]>1.9),np.mean(lm[conv]>2.0)))
        for lo,hi in ((0,1.5),(1.5,1.8),(1.8,1.9),(1.9,2.0),(2.0,2.2),(2.2,9)):
            m=conv&(lm>=lo)&(lm<hi)
            if m.sum()>3: print("     lmax [%.1f,%.1f) n=%d drift p50 %.1e p99 %.1e max %.1e  mean it %.1f"%(lo,hi,m.sum(),np.median(drift[m]),np.quantile(drift[m],.99),drift[m].max(),itf[m].mean()))

    print("---- dumps")
    import glob
    for f in sorted(glob.glob(os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'tests', 'golden', 'mle_fuzz_regressions', '*.npz'))):
        z=np.load(f); box=int(z['box']); NP=6 if str(z['method'])=='sigmaxy' else 5
        print(f.split('/')[-1], box, str(z['method']), 'lmax gpu theta', fisher_lmax(z['theta_gpu'].astype(np.float64),box,NP), 'orc theta', fisher_lmax(z['theta_orc'].astype(np.float64),box,NP))
