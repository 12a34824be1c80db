"""F(4x4, 3x3) Winograd: fp32 rounding error of candidate interpolation point sets against an fp64 convolution (transform in fp32, exact products,
fp32 accumulation over 128 channels, output transform in fp32) -- why pack.WINO4_POINTS = (0, +-3/4, +-3/2, inf): same operation count as
Lavin's (0, +-1, +-2, inf), dyadic coefficients, a third of the error.   python tools/wino4_points.py"""
import numpy as np, itertools
from fractions import Fraction as Fr
f32=np.float32
def mats(pts):
    """Toom-Cook F(4,3) with finite points pts (5) + infinity (wincnn construction)."""
    n=6; m=4; r=3
    p=[Fr(x) for x in pts]
    # polynomials
    def polymul(a,b):
        out=[Fr(0)]*(len(a)+len(b)-1)
        for i,x in enumerate(a):
            for j,y in enumerate(b): out[i+j]+=x*y
        return out
    M=[Fr(1)]
    for x in p: M=polymul(M,[-x,Fr(1)])          # degree 5, coefficients low->high
    AT=[[ (p[j]**i if not (p[j]==0 and i==0) else Fr(1)) for j in range(5)]+[Fr(1 if i==m-1 else 0)] for i in range(m)]
    Gm=[]
    for j in range(5):
        N=Fr(1)
        for k in range(5):
            if k!=j: N*= (p[j]-p[k])
        Gm.append([ (p[j]**k if not (p[j]==0 and k==0) else Fr(1))/N for k in range(r)])
    Gm.append([Fr(0),Fr(0),Fr(1)])
    BT=[]
    for j in range(5):
        # M_j(x) = M(x)/(x-p_j)
        q=[Fr(1)]
        for k in range(5):
            if k!=j: q=polymul(q,[-p[k],Fr(1)])
        BT.append(q+[Fr(0)])
    BT.append(M)
    cv=lambda A: np.array([[float(x) for x in row] for row in A])
    return cv(AT),cv(Gm),cv(BT)
def check(AT,G,BT):
    rng=np.random.default_rng(0); d=rng.standard_normal((6,6)); g=rng.standard_normal((3,3))
    Y=AT@((G@g@G.T)*(BT@d@BT.T))@AT.T
    ref=np.array([[ (d[y:y+3,x:x+3]*g).sum() for x in range(4)] for y in range(4)])
    return abs(Y-ref).max()
def err(AT,G,BT,C=128,T=64,seed=1):
    rng=np.random.default_rng(seed)
    d=rng.standard_normal((T,C,6,6)).astype(f32); g=(rng.standard_normal((C,16,3,3))*(9*C)**-0.5).astype(f32)
    dd=d.astype(np.float64); gg=g.astype(np.float64)
    ref=np.zeros((T,16,4,4))
    for y in range(4):
        for x in range(4): ref[:,:,y,x]=np.einsum('tcab,coab->to',dd[:,:,y:y+3,x:x+3],gg)
    U=np.einsum('xa,coab,yb->coxy',G,gg,G)
    BTf=BT.astype(f32); ATf=AT.astype(f32)
    V=np.einsum('xa,tcab->tcxb',BTf,d).astype(f32); V=np.einsum('tcxb,yb->tcxy',V,BTf).astype(f32)
    M=np.zeros((T,16,6,6),f32)
    for c in range(C): M=(M.astype(np.float64)+V[:,c,None].astype(np.float64)*U[c][None]).astype(f32)
    Y=np.einsum('za,toab->tozb',ATf,M).astype(f32); Y=np.einsum('tozb,yb->tozy',Y,ATf).astype(f32)
    return np.abs(Y-ref).max()/np.abs(ref).max()
for pts in ([0,1,-1,2,-2],[0,1,-1,Fr(1,2),-Fr(1,2)],[0,1,-1,Fr(1,2),-2],[0,1,-1,2,-Fr(1,2)],[0,Fr(1,2),-Fr(1,2),2,-2],[0,1,-1,Fr(3,2),-Fr(3,2)],[0,Fr(3,4),-Fr(3,4),Fr(3,2),-Fr(3,2)],[0,1,-1,Fr(1,2),-Fr(3,2)]):
    AT,G,BT=mats(pts)
    print([str(x) for x in pts],'exact-check %.1e'%check(AT,G,BT),' fp32 rel err %.2e %.2e'%(err(AT,G,BT),err(AT,G,BT,seed=2)))
