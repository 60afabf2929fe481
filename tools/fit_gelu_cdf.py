#!/usr/bin/env python3
"""Coefficients of the rational approximation of the normal CDF used by the GELU-only epilogue (csrc/common.h gelu_cdf_rational):
Phi(x) - 0.5 = x P(x^2) / Q(x^2) on |x| <= c (clamped beyond), least squares with a crude minimax re-weighting, then the error of the
fp32 Horner evaluation the kernel performs.  Prints candidates for several (c, deg P, deg Q) and the final (5, 4, 4) fit.
Reference function: nn.GELU() = x * Phi(x) (mirage/utils.py:142,156).  Needs scipy (build container)."""
import numpy as np
from scipy.special import erf
from scipy.optimize import least_squares
# fit Phi(x) - 0.5 = x * P(u)/Q(u), u = x^2, on |x| <= c; P deg np_, Q deg nq (Q(0)=1)
def fit(c, np_, nq, iters=30):
    x = np.linspace(0, c, 20001)
    u = x*x
    f = 0.5*erf(x/np.sqrt(2))
    w = np.ones_like(x)
    def model(p, u):
        P = np.polyval(p[:np_+1][::-1], u)
        Q = 1 + u*np.polyval(p[np_+1:][::-1], u)
        return P/Q
    def res(p):
        return w*(x*model(p,u) - f)
    p0 = np.zeros(np_+1+nq); p0[0] = 1/np.sqrt(2*np.pi)
    best=None
    for it in range(iters):
        r = least_squares(res, p0, method='lm', xtol=1e-15, ftol=1e-15, gtol=1e-15, max_nfev=20000)
        p0 = r.x
        e = np.abs(x*model(p0,u)-f)
        w = w*(1+ 2*e/e.max())      # crude minimax reweighting
        w /= w.mean()
        if best is None or e.max()<best[0]: best=(e.max(), p0.copy())
    return best
for c in (4.5, 5.0):
    for (a,b) in ((3,3),(4,3),(4,4),(5,4)):
        e,p = fit(c,a,b)
        print(c,a,b,"max abs err",e)

print("---- final fit")
e,p = fit(5.0,4,4,iters=120)
print("max abs err (float64 eval)", e)
np_,nq=4,4
Pc = p[:np_+1]; Qc = p[np_+1:]
print("P =", [float(v) for v in Pc]); print("Q =", [float(v) for v in Qc])
# fp32 evaluation exactly as the kernel will do it
def phi32(x):
    x = np.float32(x)
    xc = np.clip(x, np.float32(-5), np.float32(5)).astype(np.float32)
    u = (xc*xc).astype(np.float32)
    P = np.float32(Pc[4])
    for k in (3,2,1,0): P = (P*u + np.float32(Pc[k])).astype(np.float32)
    Q = np.float32(Qc[3])
    for k in (2,1,0): Q = (Q*u + np.float32(Qc[k])).astype(np.float32)
    Q = (Q*u + np.float32(1)).astype(np.float32)
    r = (np.float32(1)/Q).astype(np.float32)
    t = (xc*P).astype(np.float32)
    return (t*r + np.float32(0.5)).astype(np.float32)
xs = np.linspace(-8, 8, 2_000_001).astype(np.float32)
ref = 0.5*(1+erf(xs.astype(np.float64)/np.sqrt(2)))
got = phi32(xs).astype(np.float64)
err = np.abs(got-ref)
print("fp32 eval: max abs err of Phi", err.max(), "at x =", xs[err.argmax()])
g_err = np.abs(xs.astype(np.float64)*got - xs.astype(np.float64)*ref)
print("max abs err of gelu", g_err.max(), "at", xs[g_err.argmax()])
# relative to bf16 ulp of the result
gel = np.abs(xs.astype(np.float64)*ref); ulp = 2.0**(np.floor(np.log2(np.maximum(gel,1e-30)))-7)
mask = gel > 1e-4
print("max err / bf16 ulp where |gelu| > 1e-4:", (g_err[mask]/ulp[mask]).max())
