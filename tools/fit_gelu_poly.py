"""Coefficients of the GELU form in fullysparsefusion_amd/csrc/common.h (fsf_gelu / fsf_gelu2).

    GELU(y) = max(y, 0) - t E(t),   t = |y| / sqrt 2,   E(t) = erfc(t) / sqrt 2 = 2^P(t)

P(t) ~ log2(erfc t) - 1/2 is fitted by a polynomial, minimax (Lawson's iteratively re-weighted least squares on Chebyshev nodes) in the
error it CAUSES in GELU: dGELU/dP = t E(t) ln 2, which vanishes at t = 0 and beyond t ~ 4.  `python tools/fit_gelu_poly.py` prints, per
degree, the weighted fit error, the fp32-evaluated GELU error against float64 over [-8, 8] + normal draws, and the coefficients;
degree 8 is what common.h carries (negative leading coefficient and monotone: no range clamp needed).  numpy + scipy, CPU only.
"""
import numpy as np
from scipy.special import erfc, erf
from numpy.polynomial import chebyshev as C, polynomial as Pn
def target(t): return np.log2(erfc(t)) - 0.5
def gelu64(x): return 0.5*x*(1+erf(x/np.sqrt(2)))
def fit(T, deg, iters=40):
    # weighted minimax-ish via iteratively reweighted LS (Lawson)
    t = 0.5*T*(1-np.cos(np.pi*(np.arange(4000)+0.5)/4000))
    w0 = t*erfc(t)/np.sqrt(2)*np.log(2)  # d(gelu)/dP
    lw = np.ones_like(t)
    for it in range(iters):
        W = w0*np.sqrt(lw)
        V = C.chebvander(2*t/T-1, deg)
        coef,*_ = np.linalg.lstsq(V*W[:,None], target(t)*W, rcond=None)
        err = np.abs((V@coef - target(t))*w0)
        lw = lw*(err/err.max()+1e-3); lw/=lw.mean()
    # to monomial in t
    p = C.cheb2poly(coef)  # in u = 2t/T-1
    # substitute
    mono = np.zeros(deg+1)
    base = np.array([ -1.0, 2.0/T])
    acc = np.array([1.0])
    for k in range(deg+1):
        mono[:len(acc)] += p[k]*acc
        acc = np.convolve(acc, base)
    return mono, err.max()
def eval32(mono, x, T):
    x = x.astype(np.float32)
    t = np.minimum(np.abs(x)*np.float32(0.70710678118654752440), np.float32(T))
    p = np.full_like(t, np.float32(mono[-1]))
    for c in mono[-2::-1]:
        p = (p.astype(np.float64)*t.astype(np.float64) + np.float64(np.float32(c))).astype(np.float32)  # fma emulation
    e = np.exp2(p.astype(np.float64)).astype(np.float32)
    w = t*e
    return np.maximum(x, np.float32(0)) - w
for T in (4.2, 4.5, 5.0):
  for deg in (5,6,7,8):
    mono, e = fit(T, deg)
    xs = np.concatenate([np.linspace(-8,8,4_000_001), np.random.default_rng(0).normal(size=2_000_000)*1.5])
    y = eval32(mono, xs, T)
    ref = gelu64(xs.astype(np.float32).astype(np.float64))
    err = np.abs(y-ref)
    print(T, deg, "fit weighted err %.2e"%e, "gelu abs err max %.3e"%err.max(), "at x=%.3f"%xs[err.argmax()], "lead coeff %.3e"%mono[-1])
