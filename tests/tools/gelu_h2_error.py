"""Accuracy of the packed-half GELU of fc1 (maavss_amd/csrc/vit_epilogue.h pg_gelu_h2) against the exact-erf GELU, simulated in numpy\nwith one rounding per half operation (fused multiply-adds exact up to the rounding): which re-expansion point of the polynomial keeps\nthe Horner chain from cancelling.  CPU only.  python tests/tools/gelu_h2_error.py"""
import numpy as np, math
from numpy.polynomial import polynomial as P
kC=[-9.018102001e-10, 7.941707090e-08,-3.038026629e-06, 6.689195681e-05, -9.506666631e-04, 9.298265605e-03, -6.552827696e-02, 3.984659427e-01]
a = np.array(kC[::-1])            # a[k] coefficient of u^k
def erf_gelu(v):
    from math import erf
    return np.array([0.5*x*(1+erf(x/math.sqrt(2))) for x in v])
h = np.float16
def fma16(x,y,z):  # fused: exact product+sum in f64, one rounding
    return (x.astype(np.float64)*y.astype(np.float64)+z.astype(np.float64)).astype(h)
def gelu_f16(v, scale, shift):
    # t = (c*c)*scale - shift ; Q as polynomial in t
    # coefficients: u = (t + shift)/scale
    # poly in u -> poly in t
    pu = a.copy()
    # substitute u = (t+shift)/scale
    lin = np.array([shift/scale, 1.0/scale])
    pt = np.zeros(1)
    for k in range(len(pu)-1, -1, -1):
        pt = P.polyadd(P.polymul(pt, lin), [pu[k]])
    ct = pt.astype(h)
    hv = v.astype(h)
    c = np.clip(hv, h(-4.2), h(4.2))
    cs = (c*h(scale)).astype(h)
    t = fma16(cs, c, np.full_like(c, h(-shift)))
    q = fma16(t, np.full_like(t, ct[7]), np.full_like(t, ct[6]))
    for k in range(5,-1,-1):
        q = fma16(q, t, np.full_like(t, ct[k]))
    r = fma16(c, q, np.full_like(c, h(0.5)))
    out = (hv.astype(np.float64)*r.astype(np.float64)).astype(h)
    return out, pt
v = np.linspace(-6,6,240001)
ref = erf_gelu(v)
ref16 = ref.astype(h).astype(np.float64)
base_err = np.abs(ref16-ref)
print("f16 rounding of exact gelu: max abs %.3e rms %.3e"%(base_err.max(), np.sqrt((base_err**2).mean())))
# f32 polynomial then round
c = np.clip(v,-4.2,4.2); u=c*c
q=np.polyval(kC,u); g32=(v*(c*q+0.5))
e=np.abs(g32.astype(h).astype(np.float64)-ref); print("f32 poly + f16 round: max %.3e rms %.3e"%(e.max(), np.sqrt((e**2).mean())))
for scale,shift in [(1/16,0.0),(1/16,0.5),(1/8.82,1.0),(1/16,0.55),(1/12,0.7),(1/10,0.9)]:
    out,pt = gelu_f16(v,scale,shift)
    e = np.abs(out.astype(np.float64)-ref)
    # weighted by N(0,1) density
    w = np.exp(-v*v/2); w/=w.sum()
    print("scale 1/%.2f shift %.2f: coef max %.2f | max abs %.3e rms(uniform) %.3e rms(N(0,1)-weighted) %.3e | at |v|<=1 max %.2e, 1-2 %.2e, 2-3 %.2e, 3-4.2 %.2e, >4.2 %.2e"%(1/scale,shift,np.abs(pt).max(),e.max(),np.sqrt((e**2).mean()),np.sqrt((w*e**2).sum()),
          e[np.abs(v)<=1].max(), e[(np.abs(v)>1)&(np.abs(v)<=2)].max(), e[(np.abs(v)>2)&(np.abs(v)<=3)].max(), e[(np.abs(v)>3)&(np.abs(v)<=4.2)].max(), e[np.abs(v)>4.2].max()))
w = np.exp(-v*v/2); w/=w.sum()
print("baseline N(0,1)-weighted rms: f16 rounding only %.3e ; f32 poly+round %.3e"%(np.sqrt((w*base_err**2).sum()), np.sqrt((w*(np.abs(g32.astype(h).astype(np.float64)-ref))**2).sum())))
