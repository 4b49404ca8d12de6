#!/usr/bin/env python3
"""fp32 error of nested Winograd F(m,3) convolutions against fp64 (numpy, CPU): F(2,2,2) is what conv3_wino_pkernel computes; F(2,2,4) / F(2,4,4) / F(4,4,4)
are the larger tiles DESIGN.md section 3d prices.  64 input channels accumulated in fp32, ReLU-like inputs, max error relative to the mean magnitude of the output."""
import numpy as np
rng=np.random.default_rng(0)
def mats(m):
    if m==2:
        BT=np.array([[1,0,-1,0],[0,1,1,0],[0,-1,1,0],[0,1,0,-1]],float)
        G=np.array([[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]],float)
        AT=np.array([[1,1,1,0],[0,1,-1,-1]],float)
    else:
        BT=np.array([[4,0,-5,0,1,0],[0,-4,-4,1,1,0],[0,4,-4,-1,1,0],[0,-2,-1,2,1,0],[0,2,-1,-2,1,0],[0,4,0,-5,0,1]],float)
        G=np.array([[1/4,0,0],[-1/6,-1/6,-1/6],[-1/6,1/6,-1/6],[1/24,1/12,1/6],[1/24,-1/12,1/6],[0,0,1]],float)
        AT=np.array([[1,1,1,1,1,0],[0,1,-1,2,-2,0],[0,1,1,4,4,0],[0,1,-1,8,-8,1]],float)
    return BT,G,AT
def conv_tile(x,w,ms,dt):
    # x: [C, a0,a1,a2] input tile, w: [C,3,3,3]; returns [m0,m1,m2] in dtype dt arithmetic
    M=[mats(m) for m in ms]
    X=x.astype(dt); W=w.astype(dt)
    for ax,(BT,G,AT) in enumerate(M):
        X=np.moveaxis(np.tensordot(BT.astype(dt),X,axes=(1,ax+1)).astype(dt),0,ax+1)
        W=np.moveaxis(np.tensordot(G.astype(dt),W,axes=(1,ax+1)).astype(dt),0,ax+1)
    # elementwise multiply, accumulate over channels sequentially in dt
    P=np.zeros(X.shape[1:],dt)
    for c in range(X.shape[0]): P=(P+X[c]*W[c]).astype(dt)
    Y=P
    for ax,(BT,G,AT) in enumerate(M):
        Y=np.moveaxis(np.tensordot(AT.astype(dt),Y,axes=(1,ax)).astype(dt),0,ax)
    return Y
def direct(x,w,ms,dt):
    X=x.astype(dt); W=w.astype(dt)
    Y=np.zeros(ms,dt)
    for c in range(X.shape[0]):
        for i in range(ms[0]):
            for j in range(ms[1]):
                for k in range(ms[2]):
                    Y[i,j,k]=Y[i,j,k]+dt((X[c,i:i+3,j:j+3,k:k+3]*W[c]).sum(dtype=dt))
    return Y
C=64
for ms in [(2,2,2),(2,2,4),(2,4,4),(4,4,4)]:
    e=[];ed=[]
    for trial in range(40):
        x=rng.standard_normal((C,ms[0]+2,ms[1]+2,ms[2]+2)); x=np.maximum(x,0)   # relu-like
        w=rng.standard_normal((C,3,3,3))*0.05
        ref=conv_tile(x,w,(2,2,2),np.float64) if ms==(2,2,2) else None
        ref=direct(x,w,ms,np.float64)
        y=conv_tile(x,w,ms,np.float32)
        yd=direct(x,w,ms,np.float32)
        s=np.abs(ref).mean()
        e.append(np.abs(y-ref).max()/s); ed.append(np.abs(yd-ref).max()/s)
    print(ms,'wino err %.2e  direct err %.2e  ratio %.1f'%(np.mean(e),np.mean(ed),np.mean(e)/np.mean(ed)))
