// gpf_math.hpp -- deterministic numerics for the gfx950 particle-filter kernels.
//
// Implements DESIGN.md §3 ("numerics spec"): Philox4x32-10 counter RNG, uniform -> normal
// conversion and exp / log / sincos(2*pi*u) / atan2 built only from IEEE-754 correctly rounded
// operations (+ - * / sqrt fma) and integer bit manipulation.  The translation unit is compiled
// with -ffp-contract=off, so a*b+c is two roundings unless written as gpf::fma_().
//
// Why: the reference (GenParticleFilters.jl) draws from Julia's unseeded global RNG
// (src/resample.jl:59,113,162) and Julia's libm; "bit-exact ancestor indices under a fixed seed"
// therefore needs a stream and a math library that are identical on the host and on the GPU
// (SURVEY.md §7 H1/H2).  The CPU oracle carries an independently written C copy of this spec
// (oracle/gpf_oracle_math.h); tests compare the two bit-for-bit.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define GPF_HD __host__ __device__ __forceinline__

namespace gpf {

// ------------------------------------------------------------------ primitives
GPF_HD uint64_t d2u(double x) { return __builtin_bit_cast(uint64_t, x); }
GPF_HD double u2d(uint64_t u) { return __builtin_bit_cast(double, u); }
GPF_HD double pow2i(int e) { return u2d((uint64_t)(e + 1023) << 52); }   // 2^e, -1022 <= e <= 1023
GPF_HD double fma_(double a, double b, double c) { return __builtin_fma(a, b, c); }
GPF_HD double sqrt_(double x) { return __builtin_sqrt(x); }

GPF_HD uint64_t mulhi64(uint64_t a, uint64_t b)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __umul64hi(a, b);
#else
    return (uint64_t)(((unsigned __int128)a * b) >> 64);
#endif
}

// exact floor(x / d) (x < 2^62, quotient < 2^53): Float64 estimate (error < 1) + integer correction; replaces the
// ~100-instruction u64 division of the GPU.  invd = 1.0 / (double)d.
GPF_HD uint64_t div_small(uint64_t x, uint64_t d, double invd)
{
    uint64_t q = (uint64_t)((double)x * invd);
    int64_t r = (int64_t)(x - q * d);
    if (r < 0) { --q; r += (int64_t)d; }
    if (r < 0) { --q; r += (int64_t)d; }
    if (r >= (int64_t)d) { ++q; r -= (int64_t)d; }
    if (r >= (int64_t)d) { ++q; }
    return q;
}

// ------------------------------------------------------------------ Philox4x32-10
struct Philox { uint32_t w0, w1, w2, w3; };

GPF_HD Philox philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1)
{
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        c1 = (uint32_t)p1; c3 = (uint32_t)p0; c0 = n0; c2 = n2;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return Philox{c0, c1, c2, c3};
}

// stream tags: which pf_* operation consumes the block
enum : uint32_t { TAG_INIT = 1, TAG_UPDATE = 2, TAG_RESAMPLE = 3, TAG_MOVE = 4, TAG_REWEIGHT = 5 };

// counter = (global particle / slot id, block index, epoch, tag); key = seed
GPF_HD Philox rng(uint64_t seed, uint32_t gid, uint32_t blk, uint32_t epoch, uint32_t tag)
{
    return philox4x32_10(gid, blk, epoch, tag, (uint32_t)seed, (uint32_t)(seed >> 32));
}

// (k + 1/2) 2^-52 with k the top 52 of the 64 bits hi:lo -- strictly inside (0,1)
GPF_HD double u52(uint32_t hi, uint32_t lo)
{
    const uint64_t k = ((uint64_t)hi << 20) | (uint64_t)(lo >> 12);
    return ((double)k + 0.5) * 0x1p-52;
}
GPF_HD uint64_t u64(uint32_t hi, uint32_t lo) { return ((uint64_t)hi << 32) | lo; }
// The resample stream: ONE Philox block serves TWO output slots -- slot id s reads block s >> 1, words (0,1) when s is
// even, (2,3) when odd.  Every slot keeps its own 64-bit uniform, indexed by the slot id alone.
GPF_HD uint64_t resample_pick(const Philox& b, uint32_t slot) { return (slot & 1u) ? u64(b.w2, b.w3) : u64(b.w0, b.w1); }
GPF_HD uint64_t resample_u64(uint64_t seed, uint32_t slot, uint32_t epoch)
{
    return resample_pick(rng(seed, slot >> 1, 0, epoch, TAG_RESAMPLE), slot);
}
// ... of NS consecutive slots from s0 on (NS even): one block per ALIGNED slot pair, one more when the run starts odd.  (resample_u64 once per
// slot computes every block twice: the parity of a run's first slot is uniform over a kernel -- one branch -- but not known to the compiler.)
template <int NS>
GPF_HD void resample_u64_run(uint64_t seed, uint32_t s0, uint32_t epoch, uint64_t (&U)[NS])
{
    static_assert(NS % 2 == 0, "whole slot pairs");
    const uint32_t sb = s0 >> 1;
    if (!(s0 & 1u)) {
#pragma unroll
        for (int q = 0; q < NS / 2; ++q) {
            const Philox b = rng(seed, sb + (uint32_t)q, 0, epoch, TAG_RESAMPLE);
            U[2 * q] = u64(b.w0, b.w1); U[2 * q + 1] = u64(b.w2, b.w3);
        }
    } else {
#pragma unroll
        for (int q = 0; q <= NS / 2; ++q) {
            const Philox b = rng(seed, sb + (uint32_t)q, 0, epoch, TAG_RESAMPLE);
            if (q > 0) U[2 * q - 1] = u64(b.w0, b.w1);
            if (q < NS / 2) U[2 * q] = u64(b.w2, b.w3);
        }
    }
}

// ------------------------------------------------------------------ log (positive normal x)
GPF_HD double log_(double x)
{
    constexpr double LN2_HI = 6.93147180369123816490e-01, LN2_LO = 1.90821492927058770002e-10;
    constexpr double L1 = 6.666666666666735130e-01, L2 = 3.999999999940941908e-01,
                     L3 = 2.857142874366239149e-01, L4 = 2.222219843214978396e-01,
                     L5 = 1.818357216161805012e-01, L6 = 1.531383769920937332e-01,
                     L7 = 1.479819860511658591e-01;
    const uint64_t bits = d2u(x);
    const uint64_t man = bits & 0x000FFFFFFFFFFFFFull;
    const bool big = man >= 0x6A09E667F3BCDull;             // mantissa of sqrt(2)
    const int k = (int)(bits >> 52) - 1023 + (big ? 1 : 0);
    const double m = u2d(man | (big ? 0x3FE0000000000000ull : 0x3FF0000000000000ull));
    const double f = m - 1.0;
    const double s = f / (2.0 + f);
    const double z = s * s;
    const double w = z * z;
    const double t1 = w * fma_(w, fma_(w, L6, L4), L2);
    const double t2 = z * fma_(w, fma_(w, fma_(w, L7, L5), L3), L1);
    const double R = t2 + t1;
    const double hfsq = 0.5 * f * f;
    const double dk = (double)k;
    return dk * LN2_HI - ((hfsq - (s * (hfsq + R) + dk * LN2_LO)) - f);
}

// ------------------------------------------------------------------ exp
// e = exp(x - k ln2) in ~[0.70, 1.42], |x| < 745
GPF_HD double exp_core(double x, int& k)
{
    constexpr double INVLN2 = 1.44269504088896338700e+00;
    constexpr double LN2_HI = 6.93147180369123816490e-01, LN2_LO = 1.90821492927058770002e-10;
    constexpr double P1 = 1.66666666666666019037e-01, P2 = -2.77777777770155933842e-03,
                     P3 = 6.61375632143793436117e-05, P4 = -1.65339022054652515390e-06,
                     P5 = 4.13813679705723846039e-08;
    const double t = x * INVLN2;
    k = (int)(t + (t < 0.0 ? -0.5 : 0.5));
    const double dk = (double)k;
    const double hi = x - dk * LN2_HI;
    const double lo = dk * LN2_LO;
    const double r = hi - lo;
    const double rr = r * r;
    const double c = r - rr * fma_(rr, fma_(rr, fma_(rr, fma_(rr, P5, P4), P3), P2), P1);
    return 1.0 - ((lo - (r * c) / (2.0 - c)) - hi);
}

GPF_HD double exp_(double x)
{
    if (x != x) return x;
    if (x > 709.0) return __builtin_huge_val();
    if (x < -708.0) return 0.0;
    int k;
    const double e = exp_core(x, k);
    const int k1 = k / 2, k2 = k - k1;
    return (e * pow2i(k1)) * pow2i(k2);
}

// fixed-point weight q = trunc(exp(d) 2^K + 1/2), d <= 0
GPF_HD uint64_t exp_fix(double d, int K)
{
    if (!(d >= -708.0)) return 0;
    int k;
    const double e = exp_core(d, k);
    const int sh = k + K;
    if (sh < -2) return 0;
    const double v = e * pow2i(sh) + 0.5;
    return (uint64_t)v;
}

// ------------------------------------------------------------------ sin / cos of 2 pi u
GPF_HD double ksin(double x)
{
    constexpr double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
                     S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
                     S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    const double z = x * x;
    const double p = fma_(z, fma_(z, fma_(z, fma_(z, fma_(z, S6, S5), S4), S3), S2), S1);
    return fma_(x * z, p, x);
}
GPF_HD double kcos(double x)
{
    constexpr double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
                     C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
                     C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    const double z = x * x;
    const double p = fma_(z, fma_(z, fma_(z, fma_(z, fma_(z, C6, C5), C4), C3), C2), C1);
    return fma_(z * z, p, fma_(z, -0.5, 1.0));
}
GPF_HD void sincos2pi(double u, double& sn, double& cs)
{
    constexpr double PIO4 = 7.85398163397448278999e-01;
    const double a = u * 8.0;
    const int oct = (int)a;
    double f = a - (double)oct;
    if (oct & 1) f = 1.0 - f;
    const double th = f * PIO4;
    const double s = ksin(th), c = kcos(th);
    // octant o: angle = o*pi/4 + th (even o) or (o+1)*pi/4 - th (odd o)
    const bool swap = ((oct + 1) & 2) != 0;        // octants 1,2,5,6: sin/cos exchange roles
    const double cc = swap ? s : c;
    const double ss = swap ? c : s;
    const bool cneg = ((oct + 2) & 4) != 0;        // octants 2,3,4,5: cos negative
    const bool sneg = (oct & 4) != 0;              // octants 4..7:    sin negative
    cs = cneg ? -cc : cc;
    sn = sneg ? -ss : ss;
}

// Box-Muller pair from one Philox block
GPF_HD void normal2(const Philox& b, double& z0, double& z1)
{
    const double u1 = u52(b.w0, b.w1);
    const double u2 = u52(b.w2, b.w3);
    const double r = sqrt_(-2.0 * log_(u1));
    double s, c;
    sincos2pi(u2, s, c);
    z0 = r * c;
    z1 = r * s;
}

// ------------------------------------------------------------------ atan2
GPF_HD double atan_small(double x)
{
    const double z = x * x;
    double p = -0x1.c807c7fd8901cp-7;
    p = fma_(p, z,  0x1.f5ea203927f08p-6);
    p = fma_(p, z, -0x1.503b1df32251dp-5);
    p = fma_(p, z,  0x1.829c9a46152b5p-5);
    p = fma_(p, z, -0x1.aebd143c55a36p-5);
    p = fma_(p, z,  0x1.e1d8e89291b2fp-5);
    p = fma_(p, z, -0x1.1110ceddb22d6p-4);
    p = fma_(p, z,  0x1.3b13aea970e58p-4);
    p = fma_(p, z, -0x1.745d173561ae0p-4);
    p = fma_(p, z,  0x1.c71c71c6ddb13p-4);
    p = fma_(p, z, -0x1.24924924920acp-3);
    p = fma_(p, z,  0x1.9999999999997p-3);
    p = fma_(p, z, -0x1.5555555555555p-2);
    return fma_(x * z, p, x);
}
GPF_HD double atan2_(double y, double x)
{
    constexpr double PI = 3.14159265358979311600e+00, PIO2 = 1.57079632679489655800e+00,
                     PIO4 = 7.85398163397448278999e-01, T8 = 0.41421356237309503;
    const double ax = x < 0.0 ? -x : x, ay = y < 0.0 ? -y : y;
    if (ax == 0.0 && ay == 0.0) return 0.0;
    const bool swap = ay > ax;
    const double t = (swap ? ax : ay) / (swap ? ay : ax);
    double a = (t > T8) ? PIO4 + atan_small((t - 1.0) / (t + 1.0)) : atan_small(t);
    if (swap) a = PIO2 - a;
    if (x < 0.0) a = PI - a;
    return y < 0.0 ? -a : a;
}

// ------------------------------------------------------------------ scalar weight summaries
enum : int { FLAG_NAN = 1, FLAG_POSINF = 2, FLAG_ALL_NEGINF = 4 };

// K = min(52, 62 - ceil(log2 N)): N 2^K <= 2^62
GPF_HD int ceil_log2(int64_t n) { return n > 1 ? 64 - (int)__builtin_clzll((unsigned long long)(n - 1)) : 0; }
GPF_HD int fix_K(int64_t n_global)
{
    const int K = 62 - ceil_log2(n_global);
    return K > 52 ? 52 : K;
}
// ------------------------------------------------------------------ uniform spacings ("multinomial_sorted")
// The opt-in sorted form of the multinomial resampler (resample.jl:59) draws its N uniforms ALREADY SORTED.  With e_0..e_N i.i.d. Exp(1)
// and P_j = e_0 + ... + e_j, the ratios P_j / P_N are the order statistics of N uniforms.  The sum over a tile of SP_TILE consecutive
// slots is Gamma(tile size), independent of the tile's normalised partial sums: the tile totals are drawn directly (gamma_tile), the
// spacings place the slots inside their tile.  All in exact integers:
//     G_t = trunc(gamma(c_t [+ 1: last tile]) 2^Eg), Gtot = sum G_t + 1, Vlo_t = floor((G_0 + .. + G_{t-1}) 2^64 / Gtot)
//     e_i = trunc(neglog_u52(u_i) 2^SP_E) (-log to ~6e-12), p_j = the tile's e_i up to slot j, s_t = the tile's sum + 1 (last tile: + e_N)
//     Tlo_t = mulhi64(Vlo_t, S), Tw_t = Tlo_{t+1} - Tlo_t,  T_j = Tlo_t + min(Tw_t, trunc(((double)p_j (1.0 / (double)s_t)) (double)Tw_t))
// (the slot's place inside its tile in Float64: monotone in p_j, four roundings, the same on host and device; an exact integer
// division here cost ~36 quarter-rate 32-bit multiplications per slot)
constexpr int SP_TILE = 2048;                          // slots per tile (== the search kernel's slots per workgroup)
constexpr int SP_E = 44;                               // 2049 spacings below 2^6 each: a tile's sum stays below 2^62
GPF_HD int gamma_E(int64_t ntl) { const int E = 50 - ceil_log2(ntl); return E > 48 ? 48 : E; }   // a tile total is < 2^12
// -log of the slot's 52-bit uniform u = (k + 1/2) 2^-52 for the spacings, to ~6e-12 absolute (tools/derive_splog.py) at a fifth of
// log_'s cost -- the resampler draws one per slot: (2k + 1) = m 2^e, m in [1, 2); i = the mantissa's top 6 bits;
// ln m = SP_LN[i] + log1p(r), r = m SP_INV[i] - 1 (|r| < 2^-7: four terms); -ln u = (53 - e) ln 2 - ln m, clamped at 0.
#if defined(__HIP_DEVICE_COMPILE__)
#define GPF_TABLE __device__ static const
#else
#define GPF_TABLE static const
#endif
GPF_TABLE double SP_INV[64] = {
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
GPF_TABLE double SP_LN[64] = {
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
GPF_HD double neglog_u52(uint64_t U)
{
    const uint64_t bits = d2u((double)(((U >> 12) << 1) | 1ull));      // 2k + 1: a 53-bit odd integer, exact
    const int e = (int)(bits >> 52) - 1023;
    const uint64_t mb = bits & 0x000FFFFFFFFFFFFFull;
    const double m = u2d(mb | 0x3FF0000000000000ull);
    const int i = (int)(mb >> 46);
    const double r = m * SP_INV[i] - 1.0;
    const double p = r * (1.0 - r * (0.5 - r * (1.0 / 3.0 - r * 0.25)));
    const double v = (double)(53 - e) * 0x1.62e42fefa39efp-1 - (SP_LN[i] + p);
    return v > 0.0 ? v : 0.0;
}
GPF_HD uint64_t spacing_of(uint64_t U) { return (uint64_t)(neglog_u52(U) * pow2i(SP_E)); }
// Gamma(shape, 1), integer shape >= 1 (Marsaglia & Tsang 2000): attempt k reads blocks 1 + 2k (normal) and 2 + 2k (uniform) of counter
// gid on the resample stream; 8 attempts, then d.  |x| <= 8.6 bounds the variate below 1.2 shape + 60 < 2^12.
GPF_HD uint64_t gamma_tile(uint64_t seed, uint32_t gid, uint32_t epoch, int64_t shape, int Eg)
{
    const double d = (double)shape - 1.0 / 3.0;
    const double c = 1.0 / sqrt_(9.0 * d);
    const double sc = pow2i(Eg);
    for (int k = 0; k < 8; ++k) {
        double x, x1;
        normal2(rng(seed, gid, (uint32_t)(1 + 2 * k), epoch, TAG_RESAMPLE), x, x1);
        const Philox b = rng(seed, gid, (uint32_t)(2 + 2 * k), epoch, TAG_RESAMPLE);
        const double u = u52(b.w0, b.w1);
        const double v1 = 1.0 + c * x;
        if (!(v1 > 0.0)) continue;
        const double v = (v1 * v1) * v1;
        if (log_(u) < ((0.5 * (x * x) + d) - d * v) + d * log_(v)) return (uint64_t)((d * v) * sc);
    }
    return (uint64_t)(d * sc);
}
// floor((u1 2^64 + u0) / den) for u1 2^64 + u0 < den 2^64 by one multiplication with a precomputed reciprocal (Moeller & Granlund,
// "Improved division by invariant integers", algorithm 4: exact): d = den << sh (top bit set), v = floor((2^128 - 1) / d) - 2^64.
struct Div128 { uint64_t d, v; int sh; };
GPF_HD Div128 div128_setup(uint64_t den)
{
    Div128 c;
    c.sh = (int)__builtin_clzll(den);
    c.d = den << c.sh;
    // v = floor(((2^64 - 1 - d) 2^64 + 2^64 - 1) / d).  A Float64 estimate (off by < 2^13: 53 significant bits of a 64-bit quotient), then
    // the exact 128-bit remainder corrects it -- twice by a Float64 quotient of the remainder, then by single steps.  The result is the
    // exact integer whatever the rounding of the estimates (a 64-step long division cost 1.7 us of a lone wave's time).
    const double dd = (double)c.d;
    const uint64_t nh = ~c.d, nl = ~0ull;                // numerator, high and low word (nh < d)
    double est = ((double)nh * 0x1p64 + 0x1p64) / dd;    // ~ v, in [0, 2^64]
    uint64_t q = est >= 0x1p64 ? ~0ull : (uint64_t)est;
    // remainder r = n - q d as a signed 128-bit number (|r| < 2^78), kept as (rh: signed high, rl: low)
    for (int it = 0; it < 3; ++it) {
        const uint64_t pl = q * c.d, ph = mulhi64(q, c.d);
        const uint64_t rl = nl - pl;
        const int64_t rh = (int64_t)(nh - ph - (nl < pl ? 1u : 0u));
        // |r| small: r as a double (exact enough: the correction only has to shrink |r| below d within the iterations)
        const double rd = (double)rh * 0x1p64 + (double)rl;
        if (rh == 0 && rl < c.d) break;                    // 0 <= r < d: q is the quotient
        if (it < 2) {
            const double adj = rd / dd;                      // quotient of the remainder (floor / ceil settled by the next round)
            const int64_t a = (int64_t)adj;
            q += (uint64_t)(a != 0 ? a : (rh < 0 ? -1 : 1));
        } else {
            // last resort: single steps (at most a few)
            uint64_t l = rl; int64_t hgh = rh;
            while (hgh < 0) { const uint64_t nl2 = l + c.d; hgh += (nl2 < l) ? 1 : 0; l = nl2; --q; }
            while (hgh > 0 || l >= c.d) { const uint64_t nl2 = l - c.d; hgh -= (l < c.d) ? 1 : 0; l = nl2; ++q; }
        }
    }
    c.v = q;
    return c;
}
GPF_HD uint64_t div128_2(uint64_t n1, uint64_t n0, const Div128& c)       // floor((n1 2^64 + n0) / den), n1 < den
{
    const uint64_t u1 = c.sh ? (n1 << c.sh) | (n0 >> (64 - c.sh)) : n1, u0 = n0 << c.sh;
    uint64_t q0 = c.v * u1, q1 = mulhi64(c.v, u1);
    q0 += u0;
    q1 += u1 + (q0 < u0 ? 1u : 0u) + 1;                // (q1, q0) += (u1, u0); q1 += 1   (mod 2^64, as in the algorithm)
    uint64_t r = u0 - q1 * c.d;
    if (r > q0) { q1 -= 1; r += c.d; }
    if (r >= c.d) { q1 += 1; }
    return q1;
}
GPF_HD uint64_t div128(uint64_t P, const Div128& c) { return div128_2(P, 0, c); }       // floor(P 2^64 / den), P < den
GPF_HD uint64_t muldiv128(uint64_t p, uint64_t W, const Div128& c) { return div128_2(mulhi64(p, W), p * W, c); }   // floor(p W / den), p < den
// a slot's target inside its tile: Tlo + min(Tw, trunc(((double)p inv_s) dTw))
GPF_HD uint64_t sorted_target(uint64_t p, double inv_s, uint64_t Tlo, uint64_t Tw, double dTw)
{
    const uint64_t tt = (uint64_t)(((double)p * inv_s) * dTw);
    return Tlo + (tt > Tw ? Tw : tt);
}

// logsumexp = m + log(S 2^-K)   (resample.jl:180 / utils.jl:100 on the exact integer sum)
GPF_HD double lse_from(double m, uint64_t S, int K, int flags)
{
    if (flags & (FLAG_NAN | FLAG_POSINF)) return __builtin_nan("");
    if (flags & FLAG_ALL_NEGINF) return -__builtin_huge_val();
    return m + log_((double)S * pow2i(-K));
}
// ESS = S^2 / Q   (utils.jl:163-164)
GPF_HD double ess_from(uint64_t S, uint64_t Qhi, uint64_t Qlo)
{
    const double Sd = (double)S;
    const double Qd = (double)Qhi * 0x1p64 + (double)Qlo;
    return (Sd * Sd) / Qd;
}
GPF_HD int residual_shift(uint64_t S, int64_t N)
{
    const int bl = S ? 64 - (int)__builtin_clzll((unsigned long long)S) : 0;     // bit length of S
    const int sh = bl + ceil_log2(N) - 62;
    return sh > 0 ? sh : 0;
}

} // namespace gpf
