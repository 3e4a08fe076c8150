/*
 * gpf_oracle_math.h -- TEST INFRASTRUCTURE ONLY (CPU oracle; see oracle/README.md).
 *
 * Plain-C restatement of the deterministic numerical specification of DESIGN.md §3
 * ("numerics spec"): counter-based RNG (Philox4x32-10, Salmon et al. SC'11, the
 * published Random123 algorithm and its known-answer vectors), uniform/normal
 * conversion, and exp / log / sincos(2*pi*u) / atan2 written ONLY with IEEE-754
 * correctly-rounded primitives (+ - * / sqrt fma) and integer bit manipulation, so the
 * same spec evaluates bit-for-bit identically on the host (gcc) and on gfx950 (hipcc).
 *
 * Why own transcendental functions: the reference (GenParticleFilters.jl) gets its
 * randomness from Julia's global RNG (src/resample.jl:59,113,162; no seed anywhere) and
 * its math from Julia's libm; neither is reproducible outside Julia, so "bit-exact
 * ancestors under a fixed seed" is defined against THIS oracle.
 *
 * The product kernels have their own, separately written copy of this spec
 * (genparticlefilters.jl_amd/csrc/gpf_math.hpp); tests/ compare the two bitwise.
 * Nothing in the product may include this file.
 *
 * Compile with: gcc -O2 -ffp-contract=off -mfma   (contraction OFF: every * and + below
 * is a separately rounded operation unless written as fma()).
 */
#ifndef GPF_ORACLE_MATH_H
#define GPF_ORACLE_MATH_H

#include <stdint.h>
#include <string.h>
#include <math.h>

typedef unsigned __int128 o_u128;

/* ---------------------------------------------------------------- bit casts */
static inline uint64_t o_d2u(double x) { uint64_t u; memcpy(&u, &x, 8); return u; }
static inline double   o_u2d(uint64_t u) { double x; memcpy(&x, &u, 8); return x; }
/* 2^e as a double, e in [-1022, 1023] */
static inline double o_pow2(int e) { return o_u2d((uint64_t)(e + 1023) << 52); }

/* ---------------------------------------------------------------- Philox4x32-10 */
typedef struct { uint32_t v[4]; } o_philox_t;

static inline o_philox_t o_philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                         uint32_t k0, uint32_t k1)
{
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
    const uint32_t W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)M0 * c0;
        uint64_t p1 = (uint64_t)M1 * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += W0; k1 += W1;
    }
    o_philox_t out = {{c0, c1, c2, c3}};
    return out;
}

/* Stream layout: key = seed, counter = (gid, blk, epoch, tag). */
enum { O_TAG_INIT = 1, O_TAG_UPDATE = 2, O_TAG_RESAMPLE = 3, O_TAG_MOVE = 4, O_TAG_REWEIGHT = 5,
       O_TAG_DATA = 7 };

static inline o_philox_t o_rng(uint64_t seed, uint32_t gid, uint32_t blk, uint32_t epoch, uint32_t tag)
{
    return o_philox4x32_10(gid, blk, epoch, tag, (uint32_t)seed, (uint32_t)(seed >> 32));
}

/* 52-bit uniform strictly inside (0,1): (k + 1/2) * 2^-52, k = hi:32 bits | top 20 bits of lo */
static inline double o_u52(uint32_t hi, uint32_t lo)
{
    uint64_t k = ((uint64_t)hi << 20) | (uint64_t)(lo >> 12);
    return ((double)k + 0.5) * 0x1p-52;
}
static inline uint64_t o_u64(uint32_t hi, uint32_t lo) { return ((uint64_t)hi << 32) | lo; }
static inline uint64_t o_mulhi64(uint64_t a, uint64_t b) { return (uint64_t)(((o_u128)a * b) >> 64); }
/* The resample stream (tag RESAMPLE): ONE Philox block serves TWO output slots -- slot id s reads block s >> 1,
 * words (0,1) when s is even and words (2,3) when it is odd.  Each slot still has its own
 * independent 64-bit uniform, indexed by the slot id alone. */
static inline uint64_t o_resample_u64(uint64_t seed, uint32_t slot, uint32_t epoch)
{
    o_philox_t b = o_rng(seed, slot >> 1, 0, epoch, O_TAG_RESAMPLE);
    return (slot & 1u) ? o_u64(b.v[2], b.v[3]) : o_u64(b.v[0], b.v[1]);
}

/* ---------------------------------------------------------------- log */
/* natural log of a positive, finite, NORMAL double (callers never pass subnormals).
 * fdlibm-style: x = 2^k * m, m in [sqrt(1/2), sqrt(2)); f = m-1; s = f/(2+f);
 * log(1+f) = f - hfsq + s*(hfsq + R(s^2)); result = k*ln2_hi - ((hfsq - (s*(hfsq+R) + k*ln2_lo)) - f) */
static inline double o_log(double x)
{
    const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10;
    const double Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01,
                 Lg3 = 2.857142874366239149e-01, Lg4 = 2.222219843214978396e-01,
                 Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
                 Lg7 = 1.479819860511658591e-01;
    uint64_t u = o_d2u(x);
    int k = (int)(u >> 52) - 1023;
    uint64_t man = u & 0x000FFFFFFFFFFFFFull;
    /* m in [1,2); if m >= sqrt(2) (mantissa threshold) halve it */
    if (man >= 0x6A09E667F3BCDull) { k += 1; u = man | 0x3FE0000000000000ull; }
    else                           {          u = man | 0x3FF0000000000000ull; }
    double m = o_u2d(u);
    double f = m - 1.0;
    double s = f / (2.0 + f);
    double z = s * s;
    double w = z * z;
    double t1 = w * fma(w, fma(w, Lg6, Lg4), Lg2);
    double t2 = z * fma(w, fma(w, fma(w, Lg7, Lg5), Lg3), Lg1);
    double R = t2 + t1;
    double hfsq = 0.5 * f * f;
    double dk = (double)k;
    return dk * ln2_hi - ((hfsq - (s * (hfsq + R) + dk * ln2_lo)) - f);
}

/* ---------------------------------------------------------------- exp */
/* core: for finite x with |x| < 745, returns r-part e = exp(x - k ln2) in ~[0.70,1.42] and *kout = k */
static inline double o_exp_core(double x, int *kout)
{
    const double invln2 = 1.44269504088896338700e+00;
    const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10;
    const double P1 = 1.66666666666666019037e-01, P2 = -2.77777777770155933842e-03,
                 P3 = 6.61375632143793436117e-05, P4 = -1.65339022054652515390e-06,
                 P5 = 4.13813679705723846039e-08;
    double t = x * invln2;
    /* round half away from zero, via truncation (conversion is exact for |t| < 2^31) */
    int k = (int)(t + (t < 0.0 ? -0.5 : 0.5));
    double dk = (double)k;
    double hi = x - dk * ln2_hi;       /* dk*ln2_hi exact: ln2_hi has 32 significant bits, |k| < 2^11 */
    double lo = dk * ln2_lo;
    double r = hi - lo;
    double rr = r * r;
    double c = r - rr * fma(rr, fma(rr, fma(rr, fma(rr, P5, P4), P3), P2), P1);
    double e = 1.0 - ((lo - (r * c) / (2.0 - c)) - hi);
    *kout = k;
    return e;
}

/* general exp for the model log-likelihoods (finite range, no subnormal results needed):
 * x > 709 -> +inf, x < -708 -> 0 (flush; documented), NaN -> NaN */
static inline double o_exp(double x)
{
    if (x != x) return x;
    if (x > 709.0) return INFINITY;
    if (x < -708.0) return 0.0;
    int k; double e = o_exp_core(x, &k);
    /* two-step scaling keeps every factor a normal power of two: |k| <= 1023 */
    int k1 = k / 2, k2 = k - k1;
    return (e * o_pow2(k1)) * o_pow2(k2);
}

/* fixed-point weight: q = (uint64) (exp(d) * 2^K + 1/2), d <= 0 (or -inf), 0 <= K <= 62.
 * d = -inf or d < -708 -> 0.   */
static inline uint64_t o_exp_fix(double d, int K)
{
    if (!(d >= -708.0)) return 0;      /* also catches NaN (callers exclude NaN before) */
    int k; double e = o_exp_core(d, &k);
    int sh = k + K;                    /* e * 2^sh, e in ~[0.70, 1.42] */
    if (sh < -2) return 0;             /* e*2^sh < 0.36 -> rounds to 0 */
    double v = e * o_pow2(sh) + 0.5;   /* exact scaling; one rounding in the add */
    return (uint64_t)v;                /* truncation */
}

/* ---------------------------------------------------------------- sin/cos of 2*pi*u */
static inline double o_ksin(double x)   /* |x| <= pi/4 */
{
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
                 S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
                 S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    double z = x * x;
    double p = fma(z, fma(z, fma(z, fma(z, fma(z, S6, S5), S4), S3), S2), S1);
    return fma(x * z, p, x);
}
static inline double o_kcos(double x)   /* |x| <= pi/4 */
{
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
                 C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
                 C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    double z = x * x;
    double p = fma(z, fma(z, fma(z, fma(z, fma(z, C6, C5), C4), C3), C2), C1);
    return fma(z * z, p, fma(z, -0.5, 1.0));
}
/* u in [0,1): returns cos(2 pi u), sin(2 pi u). Octant reduction is exact (8u, floor). */
static inline void o_sincos2pi(double u, double *s_out, double *c_out)
{
    const double PIO4 = 7.85398163397448278999e-01;
    double a = u * 8.0;
    int oct = (int)a;                 /* 0..7 */
    double f = a - (double)oct;       /* exact, in [0,1) */
    if (oct & 1) f = 1.0 - f;
    double th = f * PIO4;
    double s = o_ksin(th), c = o_kcos(th);
    double cs, sn;
    switch (oct & 7) {
        case 0: cs =  c; sn =  s; break;
        case 1: cs =  s; sn =  c; break;
        case 2: cs = -s; sn =  c; break;
        case 3: cs = -c; sn =  s; break;
        case 4: cs = -c; sn = -s; break;
        case 5: cs = -s; sn = -c; break;
        case 6: cs =  s; sn = -c; break;
        default: cs =  c; sn = -s; break;
    }
    *s_out = sn; *c_out = cs;
}

/* Box-Muller pair from one Philox block */
static inline void o_normal2(o_philox_t b, double *z0, double *z1)
{
    double u1 = o_u52(b.v[0], b.v[1]);
    double u2 = o_u52(b.v[2], b.v[3]);
    double r = sqrt(-2.0 * o_log(u1));
    double s, c;
    o_sincos2pi(u2, &s, &c);
    *z0 = r * c; *z1 = r * s;
}

/* ---------------------------------------------------------------- spacing logarithm (multinomial_sorted) */
/* -log of the slot's 52-bit uniform u = (k + 1/2) 2^-52, to ~6e-12 absolute (tools/derive_splog.py): (2k + 1) = m 2^e, m in [1, 2);
 * i = the mantissa's top 6 bits; ln m = O_SP_LN[i] + log1p(r), r = m O_SP_INV[i] - 1 (|r| < 2^-7: four terms of the series);
 * -ln u = (53 - e) ln 2 - ln m, clamped at 0. */
static const double O_SP_INV[64] = {
    0x1.fc07f01fc07f0p-1, 0x1.f44659e4a4271p-1, 0x1.ecc07b301ecc0p-1, 0x1.e573ac901e574p-1,
    0x1.de5d6e3f8868ap-1, 0x1.d77b654b82c34p-1, 0x1.d0cb58f6ec074p-1, 0x1.ca4b3055ee191p-1,
    0x1.c3f8f01c3f8f0p-1, 0x1.bdd2b899406f7p-1, 0x1.b7d6c3dda338bp-1, 0x1.b2036406c80d9p-1,
    0x1.ac5701ac5701bp-1, 0x1.a6d01a6d01a6dp-1, 0x1.a16d3f97a4b02p-1, 0x1.9c2d14ee4a102p-1,
    0x1.970e4f80cb872p-1, 0x1.920fb49d0e229p-1, 0x1.8d3018d3018d3p-1, 0x1.886e5f0abb04ap-1,
    0x1.83c977ab2beddp-1, 0x1.7f405fd017f40p-1, 0x1.7ad2208e0ecc3p-1, 0x1.767dce434a9b1p-1,
    0x1.724287f46debcp-1, 0x1.6e1f76b4337c7p-1, 0x1.6a13cd1537290p-1, 0x1.661ec6a5122f9p-1,
    0x1.623fa77016240p-1, 0x1.5e75bb8d015e7p-1, 0x1.5ac056b015ac0p-1, 0x1.571ed3c506b3ap-1,
    0x1.5390948f40febp-1, 0x1.5015015015015p-1, 0x1.4cab88725af6ep-1, 0x1.49539e3b2d067p-1,
    0x1.460cbc7f5cf9ap-1, 0x1.42d6625d51f87p-1, 0x1.3fb013fb013fbp-1, 0x1.3c995a47babe7p-1,
    0x1.3991c2c187f63p-1, 0x1.3698df3de0748p-1, 0x1.33ae45b57bcb2p-1, 0x1.30d190130d190p-1,
    0x1.2e025c04b8097p-1, 0x1.2b404ad012b40p-1, 0x1.288b01288b013p-1, 0x1.25e22708092f1p-1,
    0x1.23456789abcdfp-1, 0x1.20b470c67c0d9p-1, 0x1.1e2ef3b3fb874p-1, 0x1.1bb4a4046ed29p-1,
    0x1.19453808ca29cp-1, 0x1.16e0689427379p-1, 0x1.1485f0e0acd3bp-1, 0x1.12358e75d3033p-1,
    0x1.0fef010fef011p-1, 0x1.0db20a88f4696p-1, 0x1.0b7e6ec259dc8p-1, 0x1.0953f39010954p-1,
    0x1.073260a47f7c6p-1, 0x1.05197f7d73404p-1, 0x1.03091b51f5e1ap-1, 0x1.0101010101010p-1,
};
static const double O_SP_LN[64] = {
    0x1.fe02a6b106799p-8, 0x1.7b91b07d5b126p-6, 0x1.39e87b9febd68p-5, 0x1.b42dd711971b9p-5,
    0x1.16536eea37ae3p-4, 0x1.51b073f06183cp-4, 0x1.8c345d6319b23p-4, 0x1.c5e548f5bc743p-4,
    0x1.fec9131dbeabcp-4, 0x1.1b72ad52f67a2p-3, 0x1.371fc201e8f75p-3, 0x1.526e5e3a1b438p-3,
    0x1.6d60fe719d21bp-3, 0x1.87fa06520c911p-3, 0x1.a23bc1fe2b561p-3, 0x1.bc286742d8cd4p-3,
    0x1.d5c216b4fbb94p-3, 0x1.ef0adcbdc5935p-3, 0x1.0402594b4d041p-2, 0x1.1058bf9ae4ad4p-2,
    0x1.1c898c16999fbp-2, 0x1.2895a13de86a4p-2, 0x1.347dd9a987d56p-2, 0x1.404308686a7e4p-2,
    0x1.4be5f957778a1p-2, 0x1.5767717455a6cp-2, 0x1.62c82f2b9c796p-2, 0x1.6e08eaa2ba1e4p-2,
    0x1.792a55fdd47a1p-2, 0x1.842d1da1e8b18p-2, 0x1.8f11e873662c8p-2, 0x1.99d958117e08ap-2,
    0x1.a484090e5bb09p-2, 0x1.af1293247786bp-2, 0x1.b9858969310fdp-2, 0x1.c3dd7a7cdad4dp-2,
    0x1.ce1af0b85f3ecp-2, 0x1.d83e7258a2f3ep-2, 0x1.e24881a7c6c26p-2, 0x1.ec399d2468cc1p-2,
    0x1.f6123fa7028adp-2, 0x1.ffd2e0857f497p-2, 0x1.04bdf9da926d2p-1, 0x1.0986f4f573521p-1,
    0x1.0e44985d1cc8cp-1, 0x1.12f719593efbdp-1, 0x1.179eabbd899a0p-1, 0x1.1c3b81f713c25p-1,
    0x1.20cdcd192ab6ep-1, 0x1.2555bce98f7cap-1, 0x1.29d37fec2b08bp-1, 0x1.2e47436e40268p-1,
    0x1.32b1339121d71p-1, 0x1.37117b54747b6p-1, 0x1.3b68449fffc23p-1, 0x1.3fb5b84d16f43p-1,
    0x1.43f9fe2f9ce67p-1, 0x1.48353d1ea88dfp-1, 0x1.4c679afccee39p-1, 0x1.50913cc01686bp-1,
    0x1.54b2467999498p-1, 0x1.58cadb5cd7989p-1, 0x1.5cdb1dc6c1765p-1, 0x1.60e32f44788d9p-1,
};
static inline double o_neglog_u52(uint64_t U)
{
    const uint64_t bits = o_d2u((double)(((U >> 12) << 1) | 1ull));
    const int e = (int)(bits >> 52) - 1023;
    const uint64_t mb = bits & 0x000FFFFFFFFFFFFFull;
    const double m = o_u2d(mb | 0x3FF0000000000000ull);
    const int i = (int)(mb >> 46);
    const double r = m * O_SP_INV[i] - 1.0;
    const double p = r * (1.0 - r * (0.5 - r * (1.0 / 3.0 - r * 0.25)));
    const double v = (double)(53 - e) * 0x1.62e42fefa39efp-1 - (O_SP_LN[i] + p);
    return v > 0.0 ? v : 0.0;
}

/* ---------------------------------------------------------------- atan2 */
static inline double o_atan_small(double x)  /* |x| <= tan(pi/8)(1+1e-4) */
{
    static const double A[13] = {
        -0x1.5555555555555p-2,  0x1.9999999999997p-3, -0x1.24924924920acp-3,
         0x1.c71c71c6ddb13p-4, -0x1.745d173561ae0p-4,  0x1.3b13aea970e58p-4,
        -0x1.1110ceddb22d6p-4,  0x1.e1d8e89291b2fp-5, -0x1.aebd143c55a36p-5,
         0x1.829c9a46152b5p-5, -0x1.503b1df32251dp-5,  0x1.f5ea203927f08p-6,
        -0x1.c807c7fd8901cp-7 };
    double z = x * x;
    double p = A[12];
    for (int i = 11; i >= 0; --i) p = fma(p, z, A[i]);
    return fma(x * z, p, x);
}
/* atan2(y,x) for finite inputs, not both zero-special-cased: (0,0) -> 0. Result in (-pi, pi]. */
static inline double o_atan2(double y, double x)
{
    const double PI = 3.14159265358979311600e+00, PIO2 = 1.57079632679489655800e+00,
                 PIO4 = 7.85398163397448278999e-01, T8 = 0.41421356237309503; /* tan(pi/8) */
    double ax = x < 0.0 ? -x : x, ay = y < 0.0 ? -y : y;
    if (ax == 0.0 && ay == 0.0) return 0.0;
    int swap = ay > ax;
    double num = swap ? ax : ay, den = swap ? ay : ax;   /* t = num/den in [0,1] */
    double t = num / den;
    double a;
    if (t > T8) a = PIO4 + o_atan_small((t - 1.0) / (t + 1.0));
    else        a = o_atan_small(t);
    if (swap) a = PIO2 - a;
    if (x < 0.0) a = PI - a;
    return y < 0.0 ? -a : a;
}

#endif
