"""Derive near-minimax polynomial coefficients (Chebyshev interpolation in 60-digit
arithmetic, rounded to f64) for the deterministic math spec.
Run once; the printed constants are pasted into oracle/gpf_oracle_math.h and
genparticlefilters.jl_amd/csrc/gpf_math.hpp.  Needs mpmath (ships with sympy)."""
import mpmath as mp
mp.mp.dps = 60

def cheb_fit(f, a, b, n):
    # interpolate f on [a,b] at n+1 Chebyshev nodes, return monomial coeffs (in x)
    xs = [ (a+b)/2 + (b-a)/2*mp.cos(mp.pi*(2*k+1)/(2*(n+1))) for k in range(n+1)]
    A = mp.matrix(n+1, n+1); y = mp.matrix(n+1,1)
    for i,x in enumerate(xs):
        for j in range(n+1): A[i,j] = x**j
        y[i] = f(x)
    c = mp.lu_solve(A, y)
    return [c[i] for i in range(n+1)]

def report(name, coeffs):
    print(name)
    for i,c in enumerate(coeffs):
        d = float(c)
        print(f"  c[{i}] = {d!r}  /* {d.hex()} */")

def maxerr(f, coeffs, a, b, m=4000):
    cs = [mp.mpf(float(c)) for c in coeffs]
    e = 0
    for k in range(m+1):
        x = a + (b-a)*mp.mpf(k)/m
        p = mp.polyval(cs[::-1], x)
        e = max(e, abs(p - f(x)))
    return e

# atan(x) = x + x^3 * P(z), z = x^2 in [0, tan(pi/8)^2]
za = mp.mpf(0); zb = (mp.sqrt(2)-1)**2 * mp.mpf('1.0001')
def fat(z):
    if z == 0: return mp.mpf(-1)/3
    s = mp.sqrt(z); return (mp.atan(s) - s)/(s*z)
c = cheb_fit(fat, za, zb, 12)
report("ATAN P(z) deg 12", c); print("  maxerr", mp.nstr(maxerr(fat, c, za, zb), 5))
