// pf_math.h -- bit-reproducible transcendentals for the particle-filter kernels.
//
// The reference calls std::cos/std::sin (kernel.cu:185-186), thrust's erfcinv-based
// normal_distribution (kernel.cu:381-385) and asin (kernel.cu:1079); their last-ulp
// behaviour belongs to CUDA's math library, which nothing in the reference pins.  Here each
// one is a fixed sequence of IEEE-754 double operations (+ - * / sqrt fma rint), evaluated
// identically by gfx950 and by any IEEE host, then rounded once to float: the correctly
// rounded result except for ~1e-8 of arguments.  Cost on MI355X is small because the fp64
// vector rate is half the fp32 rate.  Build with -ffp-contract=off (no implicit FMA).
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#define PF_HD __host__ __device__ __forceinline__

namespace pf {

// IEEE correctly rounded sqrt / divide.  NOT __fsqrt_rn: on ROCm that intrinsic lowers to the
// native (1 ulp) v_sqrt_f32.  Plain sqrtf / operator/ are correctly rounded under hipcc's default
// -fhip-fp32-correctly-rounded-divide-sqrt (checked bit-for-bit on gfx950 by tests/test_gpu_score.py).
PF_HD float fsqrt(float x) { return __builtin_sqrtf(x); }
PF_HD float fdiv(float a, float b) { return a / b; }

// fma(a, b, c) whose addend c is a loop-invariant constant: the compiler turns such an fma into the two-address v_fmac_f64 and pays a
// 64-bit register copy of the constant in front of every one of them (ten per sincos in the scan-match loop, 5 % of that kernel's
// VALU work); the three-address form needs none.  Same operation, same bits.
PF_HD double fma_const_addend(double a, double b, double c)
{
#if defined(__HIP_DEVICE_COMPILE__)
    double r;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
#else
    return fma(a, b, c);
#endif
}

// sin and cos of a float argument: Cody-Waite reduction by pi/2 (33+53-bit constants,
// exact for |x| < 1e6), fdlibm kernel polynomials, all in double with explicit fma.
// sincos_core: the reduced kernels sr = sin(r), cr = cos(r) and the quadrant n.
PF_HD void sincos_core(float x, double &sr, double &cr, int &n)
{
    const double TWO_OVER_PI = 6.36619772367581382433e-01;
    const double PIO2_1 = 1.57079632673412561417e+00;
    const double PIO2_1T = 6.07710050650619224932e-11;
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
                 S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
                 S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
                 C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
                 C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    double xd = (double)x;
    double fn = rint(xd * TWO_OVER_PI);
    double r = fma(-fn, PIO2_1, xd);
    r = fma(-fn, PIO2_1T, r);
    n = (int)fn;
    double z = r * r;
    double ps = fma_const_addend(S6, z, S5);
    ps = fma_const_addend(ps, z, S4);
    ps = fma_const_addend(ps, z, S3);
    ps = fma_const_addend(ps, z, S2);
    ps = fma_const_addend(ps, z, S1);
    sr = fma(r * z, ps, r);
    double pc = fma_const_addend(C6, z, C5);
    pc = fma_const_addend(pc, z, C4);
    pc = fma_const_addend(pc, z, C3);
    pc = fma_const_addend(pc, z, C2);
    pc = fma_const_addend(pc, z, C1);
    cr = fma(z * z, pc, fma(-0.5, z, 1.0));
}
PF_HD void sincosf_spec(float x, float &s, float &c)
{
    double sr, cr;
    int n;
    sincos_core(x, sr, cr, n);
    // quadrant: n&1 swaps, bit 1 of n negates sin, bit 1 of (n+1) negates cos.  Done on the rounded floats: rounding to nearest
    // commutes with negation and with selection, so the bits are those of selecting in double and rounding then
    const float sf = (float)sr, cf = (float)cr;
    float sv = (n & 1) ? cf : sf;
    float cv = (n & 1) ? sf : cf;
    if (n & 2) sv = -sv;
    if ((n + 1) & 2) cv = -cv;
    s = sv;
    c = cv;
}
// the same as doubles (~1e-16), before any rounding to float
PF_HD void sincos_d(float x, double &s, double &c)
{
    double sr, cr;
    int n;
    sincos_core(x, sr, cr, n);
    double sv = (n & 1) ? cr : sr;
    double cv = (n & 1) ? sr : cr;
    if (n & 2) sv = -sv;
    if ((n + 1) & 2) cv = -cv;
    s = sv;
    c = cv;
}

// cos / sin of rot = fl(angle + theta), the float sum CleanLidarScan forms (kernel.cu:183-186), by angle addition in double:
//   cos(A + T + d) = cos(A + T) (1 - d^2 / 2) - sin(A + T) d,    d = rot - (A + T) = the rounding error of the float sum,
// with cos / sin of the two float arguments from sincos_d.  A fixed sequence of IEEE double operations like the rest of this
// file (same bits on gfx950 and x86-64; restated in oracle/pfslam_oracle.c), accurate to a few 1e-16 before the one rounding to
// float.  The beam's part comes from a per-beam table, the heading's part is computed once per particle: the scan-match loop pays
// 14 double operations per end point instead of an argument reduction and two degree-6 polynomials.
struct AngleParts { double c, s, a; }; // cos, sin of a float angle and the angle itself, as doubles
PF_HD AngleParts angle_parts(float angle)
{
    AngleParts p;
    sincos_d(angle, p.s, p.c);
    p.a = (double)angle;
    return p;
}
// |theta| >= PF_SUM_THETA_MAX takes the direct form: d is at most half an ulp of rot and the series above drops d^3 / 6 -- below
// 1024 rad that is < 6e-15, far inside the distance of any float cos / sin to a rounding boundary that the double kernels can
// resolve (0 of 155 000 results differ from sincosf_spec(rot) for |theta| <= 3000, 6 of 31 000 at 1e4, most beyond 1e5:
// tests/test_oracle_pinning.py).  Headings are never normalised (kernel.cu:394 adds noise every frame), so the bound is part of
// the definition.  The branch is taken by no lane in any realistic run: a compare and a skipped jump per end point.
#define PF_SUM_THETA_MAX 1024.0f
// GUARD false: the caller knows every heading it will see is below the bound (the scan-match kernel of the cell rows, told by the
// host: the direct form's registers cost that kernel two of its eight waves per SIMD and 12 % of its time -- 79 instead of 60 VGPRs,
// 0.488 instead of 0.433 ms per scoring pass, tools/experiments/r04/ab_big_theta.sh -- although no lane ever takes the branch).
// GUARD 2 (PF_TRIG_DEVLIB, pfslam_set_trig): not the specification at all but the device library's cosf / sinf of rot -- what the
// reference's own text (std::cos / std::sin, kernel.cu:185-186) compiles to on this platform.  The mode exists for ONE purpose: with
// it the product's kernels must equal the reference's kernels compiled for gfx950 (oracle/_ref/kernel_ref.hsaco) with ZERO mismatches,
// which isolates everything else the product does differently (cell rows, lane order, reductions) from the last-ulp choice of the
// transcendentals.  Off by default: the specification is what the CPU oracle can follow bit for bit.
#define PF_TRIG_DEVLIB 2
template <int GUARD = 1>
PF_HD void sincos_sum_spec(const AngleParts &A, const AngleParts &T, float rot, float &s, float &c)
{
    if (GUARD == PF_TRIG_DEVLIB) {
        s = ::sinf(rot);
        c = ::cosf(rot);
        return;
    }
    if (GUARD && !(fabs(T.a) < (double)PF_SUM_THETA_MAX)) { // (a NaN heading too)
        sincosf_spec(rot, s, c);
        return;
    }
    const double at = A.a + T.a;
    const double d = (double)rot - at;
    const double c0 = fma(-A.s, T.s, A.c * T.c);
    const double s0 = fma(A.c, T.s, A.s * T.c);
    const double h = -0.5 * (d * d);
    const double c1 = fma(-d, s0, c0);
    const double s1 = fma(d, c0, s0);
    c = (float)fma(h, c0, c1);
    s = (float)fma(h, s0, s1);
}

// natural log of a positive normal double: x = m 2^e, log m = 2 atanh((m-1)/(m+1))
PF_HD double log_spec(double x)
{
    const double LN2_HI = 6.93147180369123816490e-01;
    const double LN2_LO = 1.90821492927058770002e-10;
    const double SQRT2 = 1.41421356237309514547e+00;
    union { double d; uint64_t u; } b;
    b.d = x;
    int e = (int)((b.u >> 52) & 0x7ffu) - 1023;
    b.u = (b.u & 0x000fffffffffffffull) | 0x3ff0000000000000ull;
    double m = b.d;
    if (m > SQRT2) {
        m = m * 0.5;
        e += 1;
    }
    double s = (m - 1.0) / (m + 1.0);
    double z = s * s;
    double p = 1.0 / 23.0;
    p = fma(p, z, 1.0 / 21.0);
    p = fma(p, z, 1.0 / 19.0);
    p = fma(p, z, 1.0 / 17.0);
    p = fma(p, z, 1.0 / 15.0);
    p = fma(p, z, 1.0 / 13.0);
    p = fma(p, z, 1.0 / 11.0);
    p = fma(p, z, 1.0 / 9.0);
    p = fma(p, z, 1.0 / 7.0);
    p = fma(p, z, 1.0 / 5.0);
    p = fma(p, z, 1.0 / 3.0);
    double lm = fma(s * z, p, s);
    lm = lm + lm;
    double ed = (double)e;
    return fma(ed, LN2_HI, fma(ed, LN2_LO, lm));
}

// Inverse normal CDF (Cephes ndtri -- the routine behind thrust's erfcinv), mul/add unfused.
// DEVLOG: the natural logarithm of the device library instead of log_spec -- rocThrust's own ndtri (thrust/random/detail/erfcinv.h, the
// routine its normal_distribution calls on this platform) compiled for the device: same operations, same order, its log is ::log.
template <bool DEVLOG>
PF_HD double ndtri_t(double y0)
{
    const double s2pi = 2.50662827463100050242E0;
    const double EXPM2 = 0.13533528323661269189;
    double x, y, z, y2, x0, x1;
    int code = 1;
    y = y0;
    if (y > (1.0 - EXPM2)) {
        y = 1.0 - y;
        code = 0;
    }
    if (y > EXPM2) {
        y = y - 0.5;
        y2 = y * y;
        double p = -5.99633501014107895267E1;
        p = p * y2 + 9.80010754185999661536E1;
        p = p * y2 + -5.66762857469070293439E1;
        p = p * y2 + 1.39312609387279679503E1;
        p = p * y2 + -1.23916583867381258016E0;
        double q = y2 + 1.95448858338141759834E0;
        q = q * y2 + 4.67627912898881538453E0;
        q = q * y2 + 8.63602421390890590575E1;
        q = q * y2 + -2.25462687854119370527E2;
        q = q * y2 + 2.00260212380060660359E2;
        q = q * y2 + -8.20372256168333339912E1;
        q = q * y2 + 1.59056225126211695515E1;
        q = q * y2 + -1.18331621121330003142E0;
        x = y + y * (y2 * p / q);
        x = x * s2pi;
        return x;
    }
    x = sqrt(-2.0 * (DEVLOG ? ::log(y) : log_spec(y)));
    x0 = x - (DEVLOG ? ::log(x) : log_spec(x)) / x;
    z = 1.0 / x;
    double p, q;
    if (x < 8.0) {
        p = 4.05544892305962419923E0;
        p = p * z + 3.15251094599893866154E1;
        p = p * z + 5.71628192246421288162E1;
        p = p * z + 4.40805073893200834700E1;
        p = p * z + 1.46849561928858024014E1;
        p = p * z + 2.18663306850790267539E0;
        p = p * z + -1.40256079171354495875E-1;
        p = p * z + -3.50424626827848203418E-2;
        p = p * z + -8.57456785154685413611E-4;
        q = z + 1.57799883256466749731E1;
        q = q * z + 4.53907635128879210584E1;
        q = q * z + 4.13172038254672030440E1;
        q = q * z + 1.50425385692907503408E1;
        q = q * z + 2.50464946208309415979E0;
        q = q * z + -1.42182922854787788574E-1;
        q = q * z + -3.80806407691578277194E-2;
        q = q * z + -9.33259480895457427372E-4;
    } else {
        p = 3.23774891776946035970E0;
        p = p * z + 6.91522889068984211695E0;
        p = p * z + 3.93881025292474443415E0;
        p = p * z + 1.33303460815807542389E0;
        p = p * z + 2.01485389549179081538E-1;
        p = p * z + 1.23716634817820021358E-2;
        p = p * z + 3.01581553508235416007E-4;
        p = p * z + 2.65806974686737550832E-6;
        p = p * z + 6.23974539184983293730E-9;
        q = z + 6.02427039364742014255E0;
        q = q * z + 3.67983563856160859403E0;
        q = q * z + 1.37702099489081330271E0;
        q = q * z + 2.16236993594496635890E-1;
        q = q * z + 1.34204006088543189037E-2;
        q = q * z + 3.28014464682127739104E-4;
        q = q * z + 2.89247864745380683936E-6;
        q = q * z + 6.79019408009981274425E-9;
    }
    x1 = z * p / q;
    x = x0 - x1;
    if (code != 0) x = -x;
    return x;
}

PF_HD double ndtri_spec(double y0) { return ndtri_t<false>(y0); }

PF_HD float erfcinvf_spec(float y)
{
    const double ONE_O_SQRT2 = 0x1.6a09e667f3bcdp-1;
    if (y <= 0.0f) return INFINITY;
    if (y >= 2.0f) return -INFINITY;
    return (float)(-ndtri_spec(0.5 * (double)y) * ONE_O_SQRT2);
}

PF_HD float asinf_spec(float x)
{
    const double pS0 = 1.66666666666666657415e-01, pS1 = -3.25565818622400915405e-01,
                 pS2 = 2.01212532134862925881e-01, pS3 = -4.00555345006794114027e-02,
                 pS4 = 7.91534994289814532176e-04, pS5 = 3.47933107596021167570e-05;
    const double qS1 = -2.40339491173441421878e+00, qS2 = 2.02094576023350569471e+00,
                 qS3 = -6.88283971605453293030e-01, qS4 = 7.70381505559019352791e-02;
    const double PIO2 = 1.57079632679489655800e+00;
    double xd = (double)x;
    double ax = fabs(xd);
    if (!(ax <= 1.0)) return NAN;
    double z = (ax <= 0.5) ? xd * xd : (1.0 - ax) * 0.5;
    double p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
    double q = 1.0 + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
    if (ax <= 0.5) return (float)(xd + xd * (p / q));
    double s = sqrt(z);
    double r = PIO2 - 2.0 * (s + s * (p / q));
    return (float)(xd < 0.0 ? -r : r);
}

// rsqrt(float) as the CUDA host headers give it to svd3.h
PF_HD float rsqrtf_spec(float x) { return (float)(1.0 / sqrt((double)x)); }

// ---- RNG: utilhash / makeSeededRandomEngine (kernel.cu:89-102) + thrust::minstd_rand ----
PF_HD uint32_t utilhash(uint32_t a)
{
    a = (a + 0x7ed55d16u) + (a << 12);
    a = (a ^ 0xc761c23cu) ^ (a >> 19);
    a = (a + 0x165667b1u) + (a << 5);
    a = (a + 0xd3a2646cu) ^ (a << 9);
    a = (a + 0xfd7046c5u) + (a << 3);
    a = (a ^ 0xb55a4f09u) ^ (a >> 16);
    return a;
}
PF_HD uint32_t engine_seed(int iter, int index, int depth)
{
    uint32_t key = 0x80000000u | ((uint32_t)depth << 22) | (uint32_t)iter;
    uint32_t h = utilhash(key) ^ utilhash((uint32_t)index);
    uint32_t x = h % 2147483647u;
    return x == 0u ? 1u : x;
}
PF_HD uint32_t minstd_next(uint32_t &state)
{
    state = (uint32_t)((uint64_t)state * 48271ull % 2147483647ull);
    return state;
}
PF_HD float uniform_real(uint32_t &state, float a, float b)
{
    float result = (float)(minstd_next(state) - 1u);
    result = fdiv(result, 1.0f + (float)(2147483646u - 1u));
    return (result * (b - a)) + a;
}
// devlib: rocThrust's own arithmetic instead of the specification (pfslam_set_trig, see sincos_sum_spec).  The specification follows CUDA's
// thrust, where erfcinv(2 * p) of a float is CUDA's float erfcinvf and the expression stays in float.
PF_HD float normal(uint32_t &state, float mean, float stddev, bool devlib = false)
{
    const uint32_t urng_range = 2147483646u - 1u;
    const float S1 = 4.656612873077392578125e-10f; // 1.0f / (float)urng_range == 2^-31
    const float S2 = 2.3283064365386962890625e-10f; // S1 / 2
    float S3 = -1.4142135623730950488016887242097f;
    uint32_t u = minstd_next(state) - 1u;
    if (u > (urng_range / 2)) {
        u = urng_range - u;
        S3 = -S3;
    }
    float p = (float)u * S1 + S2;
    if (devlib) {
        // rocThrust's normal_distribution as the reference's text compiles it here: `mean + stddev * S3 * erfcinv(2 * p)` with thrust's own
        // erfcinv(double) -- the product stddev * S3 in float, everything behind it in double, ONE rounding to float at the return; its
        // constant one_o_sqrt2 is the double quotient 1 / sqrt(2.0), one ulp below the nearest double to 2^-1/2
        const double x = (double)(2 * p);
        const double one_o_sqrt2 = 1.0 / 1.4142135623730951; // (1 / sqrt(2.0)), folded
        double e;
        if (x < 0.0 || x > 2.0) e = (double)NAN;
        else if (x == 0.0) e = (double)INFINITY;
        else if (x == 2.0) e = -(double)INFINITY;
        else e = -ndtri_t<true>(0.5 * x) * one_o_sqrt2;
        return (float)((double)mean + (double)(stddev * S3) * e);
    }
    return mean + stddev * S3 * erfcinvf_spec(2 * p);
}

} // namespace pf
