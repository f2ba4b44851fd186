"""Minimax fit of the one-transcendental GELU of common.h cn_gelu_e1: gelu(x) = max(x, 0) - a 2^P(a), a = |x| (scipy; development aid)."""
import numpy as np
from scipy.special import erfc
from scipy.optimize import least_squares
def Phi_neg(a): return 0.5*erfc(a/np.sqrt(2))
a=np.concatenate([np.linspace(0,6,40001),np.linspace(6,10,4001)])
target=a*Phi_neg(a)
def model(c,a):
    return a*np.exp2(np.polyval(c[::-1],a))
def fit(deg):
    m=a<=5
    c=np.polyfit(a[m],np.log2(Phi_neg(a[m])),deg)[::-1]
    for p in (2,4,8,16,32,64,128):
        def res(c):
            r=(model(c,a)-target)*1e5
            return np.sign(r)*np.abs(r)**(p/2)
        c=least_squares(res,c,method='lm',max_nfev=8000).x
    return c
def f32eval(c,x):
    x=x.astype(np.float32); c=[np.float32(v) for v in c]
    aa=np.abs(x)
    p=np.float32(c[-1])
    for k in range(len(c)-2,-1,-1):
        p=(p.astype(np.float64)*aa+c[k]).astype(np.float32)  # fma ~ single rounding
    e=np.exp2(p.astype(np.float64)).astype(np.float32)
    return (np.maximum(x,0)-(aa.astype(np.float64)*e)).astype(np.float32)
xs=np.linspace(-12,12,2000001)
ref=xs*0.5*erfc(-xs/np.sqrt(2))
for deg in (3,5,7):
    c=fit(deg)
    g=f32eval(c,xs)
    print(deg,"fit err",np.max(np.abs(model(c,a)-target)),"f32 eval err",np.max(np.abs(g-ref)))
    print("  coeffs",[float(np.float32(v)) for v in c])
    # monotone check
    big=np.linspace(0,1000,100001)
    P=np.polyval(c[::-1],big)
    print("  monotone decreasing:",bool(np.all(np.diff(P)<0)),"P(10)",np.polyval(c[::-1],10.0))
