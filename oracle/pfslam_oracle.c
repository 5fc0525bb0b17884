/*
 * pfslam_oracle.c -- CPU oracle (TEST INFRASTRUCTURE, see pfslam_oracle.h).
 *
 * Plain-C restatement of the particle-filter SLAM inner loop of the reference
 * (michaelwillett/GPU-ICP-SLAM, src/kernel.cu).  Every function cites the
 * reference lines it follows.  Build: `make -C oracle` (gcc -O3 -ffp-contract=off).
 *
 * Definitions chosen for the reference's undefined / racy behaviour:
 *   H1  tree[tree[best].parent] with parent == -1            -> traversal stops
 *   H2  uninitialised ICP target slots of out-of-range beams -> zero-filled
 *   H3  in-place resample race                               -> gather from a snapshot
 *   H4  kernUpdateMapKD duplicate RMW race                   -> every hit applied, list order
 *   H6  freePC upload length bug                             -> full list (flag: zero tail)
 *   H11 half-array D2H of the particle weights (kernel.cu:1341) -> reproduced (flag)
 *   thrust::reduce / inclusive_scan (order unspecified)      -> canonical orders below
 */
#include "pfslam_oracle.h"

#include <pthread.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* utilities.h:12 */
#define ORC_PI 3.1415926535897932384626422832795028841971f
#define ORC_LIDAR_RANGE 20.0f   /* kernel.cu:44 */
#define ORC_FREE_WEIGHT (-1)    /* kernel.cu:32 */
#define ORC_OCCUPIED_WEIGHT 4   /* kernel.cu:33 */
#define ORC_EFFECTIVE_PARTICLES .7 /* kernel.cu:31 (double) */
#define ORC_SVD_EPSILON 0.00001f   /* utilities.h:15 */

/* ------------------------------------------------------------------ */
/* A2  RNG: utilhash + makeSeededRandomEngine (kernel.cu:89-102) and   */
/*     thrust::minstd_rand = LCG(a=48271, c=0, m=2^31-1)               */
/* ------------------------------------------------------------------ */
uint32_t orc_utilhash(uint32_t a)
{
    a = (a + 0x7ed55d16u) + (a << 12);
    a = (a ^ 0xc761c23cu) ^ (a >> 19);
    a = (a + 0x165667b1u) + (a << 5);
    a = (a + 0xd3a2646cu) ^ (a << 9);
    a = (a + 0xfd7046c5u) + (a << 3);
    a = (a ^ 0xb55a4f09u) ^ (a >> 16);
    return a;
}

/* kernel.cu:100-101: h = hash((1<<31)|(depth<<22)|iter) ^ hash(index); engine(h).
 * All shifts evaluated in 32-bit two's complement (what the hardware does for the
 * reference's signed-overflowing `1 << 31` and `depth << 22`).  thrust's
 * linear_congruential_engine::seed: x = s % m, and 0 -> 1 because c == 0. */
uint32_t orc_engine_seed(int iter, int index, int depth)
{
    uint32_t key = 0x80000000u | ((uint32_t)depth << 22) | (uint32_t)iter;
    uint32_t h = orc_utilhash(key) ^ orc_utilhash((uint32_t)index);
    uint32_t x = h % 2147483647u;
    if (x == 0u) x = 1u;
    return x;
}

uint32_t orc_minstd_next(uint32_t *state)
{
    uint64_t x = (uint64_t)(*state) * 48271ull % 2147483647ull;
    *state = (uint32_t)x;
    return *state;
}

/* thrust/random/detail/uniform_real_distribution.inl: (urng()-min) / (1 + (max-min)) * (b-a) + a
 * with min = 1, max = 2147483646 for minstd_rand. */
float orc_uniform_real(uint32_t *state, float a, float b)
{
    float result = (float)(orc_minstd_next(state) - 1u);
    result /= (1.0f + (float)(2147483646u - 1u));
    return (result * (b - a)) + a;
}

/* thrust/random/detail/normal_distribution_base.h (normal_distribution_nvcc::sample):
 * the variant nvcc selects; erfcinv is the CUDA math library's in the reference,
 * here the pf_math specification. */
float orc_normal(uint32_t *state, float mean, float stddev)
{
    const uint32_t urng_range = 2147483646u - 1u;
    const float S1 = 1.0f / (float)urng_range; /* == 2^-31 */
    const float S2 = S1 / 2;
    float S3 = -1.4142135623730950488016887242097f;
    uint32_t u = orc_minstd_next(state) - 1u;
    if (u > (urng_range / 2)) {
        u = urng_range - u;
        S3 = -S3;
    }
    float p = (float)u * S1 + S2;
    return mean + stddev * S3 * orc_erfcinvf(2 * p);
}

/* ------------------------------------------------------------------ */
/* pf_math: bit-reproducible transcendentals.                          */
/* Each is a fixed sequence of IEEE-754 double operations (+,-,*,/,     */
/* sqrt, fma, rint), so gcc on x86-64 and hipcc on gfx950 give the same  */
/* bits.  Accuracy ~1e-16 before the final rounding to float, i.e. the   */
/* correctly rounded float result except in ~1e-8 of arguments.          */
/* ------------------------------------------------------------------ */
/* sin and cos of a float argument as doubles (~1e-16): what orc_sincosf rounds to float */
void orc_sincos_d(float x, double *s, double *c)
{
    /* Cody-Waite reduction by pi/2 (2 constants, 33+53 bits), fdlibm kernel polynomials */
    static const double TWO_OVER_PI = 6.36619772367581382433e-01;
    static const double PIO2_1 = 1.57079632673412561417e+00;
    static const double PIO2_1T = 6.07710050650619224932e-11;
    static const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
                        S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
                        S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    static const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
                        C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
                        C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    double xd = (double)x;
    double fn = rint(xd * TWO_OVER_PI);
    double r = fma(-fn, PIO2_1, xd);
    r = fma(-fn, PIO2_1T, r);
    int n = (int)fn;
    double z = r * r;
    double ps = S6;
    ps = fma(ps, z, S5);
    ps = fma(ps, z, S4);
    ps = fma(ps, z, S3);
    ps = fma(ps, z, S2);
    ps = fma(ps, z, S1);
    double sr = fma(r * z, ps, r);
    double pc = C6;
    pc = fma(pc, z, C5);
    pc = fma(pc, z, C4);
    pc = fma(pc, z, C3);
    pc = fma(pc, z, C2);
    pc = fma(pc, z, C1);
    double cr = fma(z * z, pc, fma(-0.5, z, 1.0));
    double sv, cv;
    switch (n & 3) {
    case 0: sv = sr; cv = cr; break;
    case 1: sv = cr; cv = -sr; break;
    case 2: sv = -sr; cv = -cr; break;
    default: sv = -cr; cv = sr; break;
    }
    *s = sv;
    *c = cv;
}
void orc_sincosf(float x, float *s, float *c)
{
    double sd, cd;
    orc_sincos_d(x, &sd, &cd);
    *s = (float)sd;
    *c = (float)cd;
}

double orc_log(double x)
{
    /* x > 0, normal.  x = m * 2^e, m in (sqrt(1/2), sqrt(2)];
     * log m = 2 atanh(s), s = (m-1)/(m+1), odd series to s^23 */
    static const double LN2_HI = 6.93147180369123816490e-01;
    static const double LN2_LO = 1.90821492927058770002e-10;
    static const double SQRT2 = 1.41421356237309514547e+00;
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

/* Cephes ndtri (inverse normal CDF), the routine rocThrust's erfcinv wraps
 * (thrust/random/detail/erfcinv.h); log replaced by orc_log, mul/add unfused. */
static double orc_polevl(double x, const double *coef, int N)
{
    double ans = coef[0];
    for (int i = 1; i <= N; i++) ans = ans * x + coef[i];
    return ans;
}
static double orc_p1evl(double x, const double *coef, int N)
{
    double ans = x + coef[0];
    for (int i = 1; i < N; i++) ans = ans * x + coef[i];
    return ans;
}

double orc_ndtri(double y0)
{
    static const double s2pi = 2.50662827463100050242E0;
    static const double P0[5] = {-5.99633501014107895267E1, 9.80010754185999661536E1,
                                 -5.66762857469070293439E1, 1.39312609387279679503E1,
                                 -1.23916583867381258016E0};
    static const double Q0[8] = {1.95448858338141759834E0, 4.67627912898881538453E0,
                                 8.63602421390890590575E1, -2.25462687854119370527E2,
                                 2.00260212380060660359E2, -8.20372256168333339912E1,
                                 1.59056225126211695515E1, -1.18331621121330003142E0};
    static const double P1[9] = {4.05544892305962419923E0, 3.15251094599893866154E1,
                                 5.71628192246421288162E1, 4.40805073893200834700E1,
                                 1.46849561928858024014E1, 2.18663306850790267539E0,
                                 -1.40256079171354495875E-1, -3.50424626827848203418E-2,
                                 -8.57456785154685413611E-4};
    static const double Q1[8] = {1.57799883256466749731E1, 4.53907635128879210584E1,
                                 4.13172038254672030440E1, 1.50425385692907503408E1,
                                 2.50464946208309415979E0, -1.42182922854787788574E-1,
                                 -3.80806407691578277194E-2, -9.33259480895457427372E-4};
    static const double P2[9] = {3.23774891776946035970E0, 6.91522889068984211695E0,
                                 3.93881025292474443415E0, 1.33303460815807542389E0,
                                 2.01485389549179081538E-1, 1.23716634817820021358E-2,
                                 3.01581553508235416007E-4, 2.65806974686737550832E-6,
                                 6.23974539184983293730E-9};
    static const double Q2[8] = {6.02427039364742014255E0, 3.67983563856160859403E0,
                                 1.37702099489081330271E0, 2.16236993594496635890E-1,
                                 1.34204006088543189037E-2, 3.28014464682127739104E-4,
                                 2.89247864745380683936E-6, 6.79019408009981274425E-9};
    static const double EXPM2 = 0.13533528323661269189;
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
        x = y + y * (y2 * orc_polevl(y2, P0, 4) / orc_p1evl(y2, Q0, 8));
        x = x * s2pi;
        return x;
    }
    x = sqrt(-2.0 * orc_log(y));
    x0 = x - orc_log(x) / x;
    z = 1.0 / x;
    if (x < 8.0)
        x1 = z * orc_polevl(z, P1, 8) / orc_p1evl(z, Q1, 8);
    else
        x1 = z * orc_polevl(z, P2, 8) / orc_p1evl(z, Q2, 8);
    x = x0 - x1;
    if (code != 0) x = -x;
    return x;
}

float orc_erfcinvf(float y)
{
    static const double ONE_O_SQRT2 = 0x1.6a09e667f3bcdp-1;
    if (y <= 0.0f) return INFINITY;
    if (y >= 2.0f) return -INFINITY;
    return (float)(-orc_ndtri(0.5 * (double)y) * ONE_O_SQRT2);
}

float orc_asinf(float x)
{
    /* fdlibm e_asin.c rational approximation, evaluated unfused in double */
    static const double pS0 = 1.66666666666666657415e-01, pS1 = -3.25565818622400915405e-01,
                        pS2 = 2.01212532134862925881e-01, pS3 = -4.00555345006794114027e-02,
                        pS4 = 7.91534994289814532176e-04, pS5 = 3.47933107596021167570e-05;
    static const double qS1 = -2.40339491173441421878e+00, qS2 = 2.02094576023350569471e+00,
                        qS3 = -6.88283971605453293030e-01, qS4 = 7.70381505559019352791e-02;
    static const double PIO2 = 1.57079632679489655800e+00;
    double xd = (double)x;
    double ax = fabs(xd);
    if (!(ax <= 1.0)) return NAN;
    if (ax <= 0.5) {
        double z = xd * xd;
        double p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
        double q = 1.0 + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
        return (float)(xd + xd * (p / q));
    }
    double z = (1.0 - ax) * 0.5;
    double s = sqrt(z);
    double p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
    double q = 1.0 + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
    double r = PIO2 - 2.0 * (s + s * (p / q));
    return (float)(xd < 0.0 ? -r : r);
}

/* rsqrt(float) as the CUDA host headers supply it to svd3.h:158,284 */
float orc_rsqrtf(float x)
{
    return (float)(1.0 / sqrt((double)x));
}

/* Canonical order for thrust::reduce (kernel.cu:459,463,1025,1026,1050), whose
 * order the reference leaves unspecified: a 64-lane strided accumulation
 * followed by an xor butterfly (32,16,...,1); arrays longer than 4096 are
 * reduced tile-by-tile (4096) and the tile sums reduced recursively. */
static float orc_sum_tile(const float *v, int n, int stride)
{
    float acc[64];
    for (int l = 0; l < 64; l++) {
        float a = 0.0f;
        for (int i = l; i < n; i += 64) a = a + v[(size_t)i * stride];
        acc[l] = a;
    }
    for (int off = 32; off >= 1; off >>= 1) {
        float nxt[64];
        for (int l = 0; l < 64; l++) nxt[l] = acc[l] + acc[l ^ off];
        memcpy(acc, nxt, sizeof(acc));
    }
    return acc[0];
}

float orc_sum_f32(const float *v, int n, int stride)
{
    if (n <= 4096) return orc_sum_tile(v, n, stride);
    int nt = (n + 4095) / 4096;
    float *part = (float *)malloc(sizeof(float) * (size_t)nt);
    for (int t = 0; t < nt; t++) {
        int cnt = n - t * 4096;
        if (cnt > 4096) cnt = 4096;
        part[t] = orc_sum_tile(v + (size_t)t * 4096 * stride, cnt, stride);
    }
    float r = orc_sum_f32(part, nt, 1);
    free(part);
    return r;
}

/* Canonical order for thrust::inclusive_scan (kernel.cu:478): chunks of 16
 * scanned sequentially, chunk offsets scanned sequentially inside tiles of
 * 1024, tile offsets scanned sequentially; cdf = (tile_off + chunk_off) + local. */
void orc_inclusive_scan_f32(const float *w, int n, float *cdf)
{
    int nt = (n + 1023) / 1024;
    float tile_off = 0.0f;
    for (int t = 0; t < nt; t++) {
        int t0 = t * 1024;
        float chunk_off = 0.0f;
        for (int c = 0; c < 64; c++) {
            int c0 = t0 + c * 16;
            if (c0 >= n) break;
            float base = tile_off + chunk_off;
            float run = 0.0f;
            for (int k = 0; k < 16 && c0 + k < n; k++) {
                run = (k == 0) ? w[c0 + k] : run + w[c0 + k];
                cdf[c0 + k] = base + run;
            }
            chunk_off = chunk_off + run;
        }
        tile_off = tile_off + chunk_off;
    }
}

/* ------------------------------------------------------------------ */
/* A3  motion update: ParticleAddNoise (kernel.cu:375-387)             */
/* ------------------------------------------------------------------ */
void orc_add_noise(orc_particle *p, int n, int frame, int global_idx0)
{
    const float cov[3] = {0.015, 0.015, .01}; /* kernel.cu:45, used as std-dev (H10) */
    for (int i = 0; i < n; i++) {
        uint32_t e2 = orc_engine_seed(frame, global_idx0 + i, 0);
        float nx = orc_normal(&e2, 0.0f, cov[0]);
        float ny = orc_normal(&e2, 0.0f, cov[1]);
        float nt = orc_normal(&e2, 0.0f, cov[2]);
        p[i].x += nx;
        p[i].y += ny;
        p[i].theta += nt;
    }
}

/* ------------------------------------------------------------------ */
/* A4  CleanLidarScan (kernel.cu:182-187), LIDAR_ANGLE (kernel.cu:42)  */
/* ------------------------------------------------------------------ */
/* cos / sin of rot = fl(angle + theta), the float sum the reference forms (kernel.cu:183-186), by angle addition in double:
 *   cos(A + T + d) = cos(A + T) (1 - d^2 / 2) - sin(A + T) d,   d = rot - (A + T) = the rounding error of the float sum,
 * with cos / sin of the two float arguments from orc_sincos_d.  A fixed sequence of IEEE double operations like everything
 * else here (same bits on x86-64 and gfx950), accurate to a few 1e-16 before the one rounding to float -- still far inside the
 * 1-2 ulp of any libm cosf.  Why: the beam angle's part is a per-beam table and the heading's part is computed once per
 * particle, so the product's scan-match loop pays 14 double operations per end point instead of 26 + an argument reduction. */
typedef struct { double c, s, a; } orc_angle_parts;
static orc_angle_parts orc_parts(float angle)
{
    orc_angle_parts p;
    orc_sincos_d(angle, &p.s, &p.c);
    p.a = (double)angle;
    return p;
}
/* |theta| >= ORC_SUM_THETA_MAX: the direct form.  d is the rounding error of the float sum, half an ulp of rot at most, and the
 * series above drops d^3 / 6: below 1024 rad that is < 6e-15 (checked: 0 of 155 000 results differ from orc_sincosf(rot) for
 * |theta| <= 3000; 6 of 31 000 at 1e4; headings are never normalised, so the bound is part of the definition, not an assumption) */
#define ORC_SUM_THETA_MAX 1024.0f
static void orc_sincos_sum(const orc_angle_parts *A, const orc_angle_parts *T, float rot, float *s, float *c)
{
    if (!(fabs(T->a) < ORC_SUM_THETA_MAX)) { /* (a NaN heading too) */
        orc_sincosf(rot, s, c);
        return;
    }
    const double at = A->a + T->a;
    const double d = (double)rot - at;
    const double c0 = fma(-A->s, T->s, A->c * T->c);
    const double s0 = fma(A->c, T->s, A->s * T->c);
    const double h = -0.5 * (d * d);
    const double c1 = fma(-d, s0, c0);
    const double s1 = fma(d, c0, s0);
    *c = (float)fma(h, c0, c1);
    *s = (float)fma(h, s0, s1);
}
static float orc_lidar_angle(int n) { return (-135.0f + n * .25f) * ORC_PI / 180; } /* LIDAR_ANGLE(n), kernel.cu:42 */
/* the beams' parts, computed once (they depend on the beam index only) */
#define ORC_BEAM_TABLE 4096
static const orc_angle_parts *orc_beam_parts(void)
{
    static orc_angle_parts table[ORC_BEAM_TABLE];
    static volatile int ready = 0;
    static pthread_mutex_t mu = PTHREAD_MUTEX_INITIALIZER;
    if (!ready) {
        pthread_mutex_lock(&mu);
        if (!ready) {
            for (int n = 0; n < ORC_BEAM_TABLE; n++) table[n] = orc_parts(orc_lidar_angle(n));
            __sync_synchronize();
            ready = 1;
        }
        pthread_mutex_unlock(&mu);
    }
    return table;
}
/* with the heading's parts at hand (loops over the beams of one pose) */
static void orc_clean_lidar_scan_pre(int n, float scan, float theta, const orc_angle_parts *T, float *x, float *y)
{
    const float ang = orc_lidar_angle(n);
    const float rot = ang + theta;
    orc_angle_parts An;
    const orc_angle_parts *A;
    if (n >= 0 && n < ORC_BEAM_TABLE) A = orc_beam_parts() + n;
    else { An = orc_parts(ang); A = &An; }
    float s, c;
    orc_sincos_sum(A, T, rot, &s, &c);
    *x = scan * c;
    *y = scan * s;
}
void orc_clean_lidar_scan(int n, float scan, float theta, float *x, float *y)
{
    const orc_angle_parts T = orc_parts(theta);
    orc_clean_lidar_scan_pre(n, scan, theta, &T, x, y);
}

/* glm::distance(vec3, vec3) = sqrt(dot(d, d)), dot = (x*x + y*y) + z*z
 * (glm/detail/func_geometric.inl:65-72, 102-115) */
static inline float orc_dist3(float ax, float ay, float az, float bx, float by, float bz)
{
    float dx = bx - ax, dy = by - ay, dz = bz - az;
    float tx = dx * dx, ty = dy * dy, tz = dz * dz;
    return sqrtf(tx + ty + tz);
}

/* getHyperplaneDist (kernel.cu:843-860) */
static inline float orc_hyperplane(float px, float py, float pz, const orc_node *nd, int *branch)
{
    float retv = 0.0f;
    if (nd->axis == 0) { *branch = px < nd->x; retv = fabsf(px - nd->x); }
    if (nd->axis == 1) { *branch = py < nd->y; retv = fabsf(py - nd->y); }
    if (nd->axis == 2) { *branch = pz < nd->z; retv = fabsf(pz - nd->z); }
    return retv;
}

/* The reference's "nearest neighbour": greedy descent + sibling re-descents.
 * Identical text at kernel.cu:881-919, 931-969, 1147-1184, 1239-1276. */
int orc_kd_traverse(const orc_node *tree, float px, float py, float pz, int *visits)
{
    float bestDist = orc_dist3(px, py, pz, tree[0].x, tree[0].y, tree[0].z);
    int bestIdx = 0, head = 0, done = 0, branch = 0, nodeFullyExplored = 0;
    int nv = 0;
    while (!done) {
        while (head >= 0) {
            const orc_node test = tree[head];
            nv++;
            float d = orc_dist3(px, py, pz, test.x, test.y, test.z);
            if (d < bestDist) {
                bestDist = d;
                bestIdx = head;
                nodeFullyExplored = 0;
            }
            orc_hyperplane(px, py, pz, &test, &branch);
            head = branch ? test.left : test.right;
        }
        if (nodeFullyExplored) {
            done = 1;
        } else {
            int pi = tree[bestIdx].parent;
            if (pi < 0) { /* H1: reference reads tree[-1]; defined here as "stop" */
                done = 1;
            } else {
                const orc_node parent = tree[pi];
                nv++;
                if (orc_hyperplane(px, py, pz, &parent, &branch) < bestDist) {
                    head = !branch ? parent.left : parent.right;
                    nodeFullyExplored = 1;
                } else {
                    done = 1;
                }
            }
        }
    }
    if (visits) *visits = nv;
    return bestIdx;
}

void orc_traverse_batch(const orc_node *tree, const float *xyz, int n, int32_t *best, int32_t *visits)
{
    for (int i = 0; i < n; i++) {
        int v;
        best[i] = orc_kd_traverse(tree, xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2], &v);
        if (visits) visits[i] = v;
    }
}

/* ------------------------------------------------------------------ */
/* A5  EvaluateParticleKD (kernel.cu:1198-1298)                        */
/* ------------------------------------------------------------------ */
static float orc_evaluate_particle_kd(const orc_node *tree, const orc_particle *pt, const float *scan,
                                      int n_beams, uint64_t *nv, uint64_t *nvalid)
{
    float retv = 0.0f;
    const orc_angle_parts T = orc_parts(pt->theta);
    for (int j = 0; j < n_beams; j++) {
        float wx, wy;
        orc_clean_lidar_scan_pre(j, scan[j], pt->theta, &T, &wx, &wy);
        if (fabsf(wx) < ORC_LIDAR_RANGE && fabsf(wy) < ORC_LIDAR_RANGE) {
            wx += pt->x;
            wy += pt->y;
            int v;
            int b = orc_kd_traverse(tree, wx, wy, 0.0f, &v);
            if (nv) *nv += (uint64_t)v;
            if (nvalid) *nvalid += 1;
            retv += tree[b].w;
        }
    }
    return retv;
}

void orc_score_kd(const orc_node *tree, const orc_particle *p, int n, const float *scan,
                  int n_beams, float *fit, uint64_t *node_visits, uint64_t *valid_beams)
{
    uint64_t nv = 0, nb = 0;
    for (int i = 0; i < n; i++) fit[i] = orc_evaluate_particle_kd(tree, &p[i], scan, n_beams, &nv, &nb);
    if (node_visits) *node_visits = nv;
    if (valid_beams) *valid_beams = nb;
}

#include <pthread.h>
typedef struct {
    const orc_node *tree; const orc_particle *p; const float *scan; float *fit;
    int i0, i1, n_beams;
} orc_mt_job;
static void *orc_mt_worker(void *arg)
{
    orc_mt_job *j = (orc_mt_job *)arg;
    for (int i = j->i0; i < j->i1; i++)
        j->fit[i] = orc_evaluate_particle_kd(j->tree, &j->p[i], j->scan, j->n_beams, 0, 0);
    return 0;
}
/* same function, particles split over host threads (cpu_baseline leg of bench.py) */
void orc_score_kd_mt(const orc_node *tree, const orc_particle *p, int n, const float *scan,
                     int n_beams, float *fit, int n_threads)
{
    if (n_threads < 1) n_threads = 1;
    if (n_threads > 256) n_threads = 256;
    pthread_t th[256];
    orc_mt_job jobs[256];
    for (int t = 0; t < n_threads; t++) {
        jobs[t].tree = tree; jobs[t].p = p; jobs[t].scan = scan; jobs[t].fit = fit;
        jobs[t].n_beams = n_beams;
        jobs[t].i0 = (int)((long long)n * t / n_threads);
        jobs[t].i1 = (int)((long long)n * (t + 1) / n_threads);
        pthread_create(&th[t], 0, orc_mt_worker, &jobs[t]);
    }
    for (int t = 0; t < n_threads; t++) pthread_join(th[t], 0);
}

/* ------------------------------------------------------------------ */
/* A6  thrust::minmax_element (first occurrence of both) + weights     */
/* ------------------------------------------------------------------ */
void orc_minmax_first_f32(const float *v, int n, int *imin, int *imax)
{
    int a = 0, b = 0;
    for (int i = 1; i < n; i++) {
        if (v[i] < v[a]) a = i;
        if (v[b] < v[i]) b = i;
    }
    *imin = a;
    *imax = b;
}
void orc_minmax_first_i32(const int32_t *v, int n, int *imin, int *imax)
{
    int a = 0, b = 0;
    for (int i = 1; i < n; i++) {
        if (v[i] < v[a]) a = i;
        if (v[b] < v[i]) b = i;
    }
    *imin = a;
    *imax = b;
}

/* kernUpdateWeights float overload (kernel.cu:297-304): `int min` truncates (A6/H8) */
void orc_update_weights_f32(orc_particle *p, int n, const float *fit, float c, int min_trunc)
{
    for (int i = 0; i < n; i++) p[i].w = p[i].w * (fit[i] - min_trunc) * c;
}
/* int overload (kernel.cu:287-294) */
void orc_update_weights_i32(orc_particle *p, int n, const int32_t *fit, float c, int min_v)
{
    for (int i = 0; i < n; i++) p[i].w = p[i].w * ((float)fit[i] - min_v) * c;
}

/* ------------------------------------------------------------------ */
/* A9  3x3 SVD, McAdams et al. as implemented in svd3.h:52-401          */
/* ------------------------------------------------------------------ */
#define ORC_GAMMA 5.828427124  /* svd3.h:22 (double) */
#define ORC_CSTAR 0.923879532  /* svd3.h:23 */
#define ORC_SSTAR 0.3826834323 /* svd3.h:24 */

static float orc_rsqrt1(float x) /* svd3.h:52-60 */
{
    float xhalf = 0.5f * x;
    union { float f; int32_t i; } u;
    u.f = x;
    u.i = 0x5f37599e - (u.i >> 1);
    x = u.f;
    x = x * (1.5f - xhalf * x * x);
    x = x * (1.5f - xhalf * x * x);
    return x;
}
static float orc_accurate_sqrt(float x) { return x * orc_rsqrt1(x); } /* svd3.h:62-65 */

static void orc_cond_swap(int c, float *X, float *Y) /* svd3.h:67-73 */
{
    float Z = *X;
    *X = c ? *Y : *X;
    *Y = c ? Z : *Y;
}
static void orc_cond_neg_swap(int c, float *X, float *Y) /* svd3.h:75-81 */
{
    float Z = -*X;
    *X = c ? *Y : *X;
    *Y = c ? Z : *Y;
}

/* svd3.h:145-161 */
static void orc_approx_givens_quat(float a11, float a12, float a22, float *ch, float *sh)
{
    *ch = 2 * (a11 - a22);
    *sh = a12;
    int b = ORC_GAMMA * *sh * *sh < *ch * *ch; /* double * float * float < float*float */
    float w = orc_rsqrtf(*ch * *ch + *sh * *sh);
    *ch = b ? w * *ch : (float)ORC_CSTAR;
    *sh = b ? w * *sh : (float)ORC_SSTAR;
}

/* svd3.h:163-216; s = {s11, s21, s22, s31, s32, s33} */
static void orc_jacobi_conjugation(int x, int y, int z, float *s, float *qV)
{
    float ch, sh;
    orc_approx_givens_quat(s[0], s[1], s[2], &ch, &sh);
    float scale = ch * ch + sh * sh;
    float a = (ch * ch - sh * sh) / scale;
    float b = (2 * sh * ch) / scale;
    float _s11 = s[0], _s21 = s[1], _s22 = s[2], _s31 = s[3], _s32 = s[4], _s33 = s[5];
    s[0] = a * (a * _s11 + b * _s21) + b * (a * _s21 + b * _s22);
    s[1] = a * (-b * _s11 + a * _s21) + b * (-b * _s21 + a * _s22);
    s[2] = -b * (-b * _s11 + a * _s21) + a * (-b * _s21 + a * _s22);
    s[3] = a * _s31 + b * _s32;
    s[4] = -b * _s31 + a * _s32;
    s[5] = _s33;
    float tmp[3];
    tmp[0] = qV[0] * sh;
    tmp[1] = qV[1] * sh;
    tmp[2] = qV[2] * sh;
    sh *= qV[3];
    qV[0] *= ch;
    qV[1] *= ch;
    qV[2] *= ch;
    qV[3] *= ch;
    qV[z] += sh;
    qV[3] -= tmp[z];
    qV[x] += tmp[y];
    qV[y] -= tmp[x];
    _s11 = s[2];
    _s21 = s[4]; _s22 = s[5];
    _s31 = s[1]; _s32 = s[3]; _s33 = s[0];
    s[0] = _s11;
    s[1] = _s21; s[2] = _s22;
    s[3] = _s31; s[4] = _s32; s[5] = _s33;
}

static float orc_dist2(float x, float y, float z) { return x * x + y * y + z * z; } /* svd3.h:218 */

/* svd3.h:277-292 */
static void orc_qr_givens_quat(float a1, float a2, float *ch, float *sh)
{
    float epsilon = (float)ORC_SVD_EPSILON;
    float rho = orc_accurate_sqrt(a1 * a1 + a2 * a2);
    *sh = rho > epsilon ? a2 : 0;
    *ch = fabsf(a1) + fmaxf(rho, epsilon);
    int b = a1 < 0;
    orc_cond_swap(b, sh, ch);
    float w = orc_rsqrtf(*ch * *ch + *sh * *sh);
    *ch *= w;
    *sh *= w;
}

/* row-major a[9] = a11..a33 in, u/s/v row-major out (svd3.h:354-401) */
void orc_svd3(const float a[9], float u[9], float sm[9], float v[9])
{
    float a11 = a[0], a12 = a[1], a13 = a[2], a21 = a[3], a22 = a[4], a23 = a[5], a31 = a[6],
          a32 = a[7], a33 = a[8];
    /* ATA = A^T A (multAtB, svd3.h:102-117) */
    float ATA11 = a11 * a11 + a21 * a21 + a31 * a31;
    float ATA21 = a12 * a11 + a22 * a21 + a32 * a31;
    float ATA22 = a12 * a12 + a22 * a22 + a32 * a32;
    float ATA31 = a13 * a11 + a23 * a21 + a33 * a31;
    float ATA32 = a13 * a12 + a23 * a22 + a33 * a32;
    float ATA33 = a13 * a13 + a23 * a23 + a33 * a33;
    /* jacobiEigenanlysis (svd3.h:224-244): 4 sweeps of (0,1),(1,2),(0,2) */
    float s[6] = {ATA11, ATA21, ATA22, ATA31, ATA32, ATA33};
    float qV[4] = {0, 0, 0, 1};
    for (int i = 0; i < 4; i++) {
        orc_jacobi_conjugation(0, 1, 2, s, qV);
        orc_jacobi_conjugation(1, 2, 0, s, qV);
        orc_jacobi_conjugation(2, 0, 1, s, qV);
    }
    /* quatToMat3 (svd3.h:119-143) */
    float w = qV[3], x = qV[0], y = qV[1], z = qV[2];
    float qxx = x * x, qyy = y * y, qzz = z * z, qxz = x * z, qxy = x * y, qyz = y * z;
    float qwx = w * x, qwy = w * y, qwz = w * z;
    float v11 = 1 - 2 * (qyy + qzz), v12 = 2 * (qxy - qwz), v13 = 2 * (qxz + qwy);
    float v21 = 2 * (qxy + qwz), v22 = 1 - 2 * (qxx + qzz), v23 = 2 * (qyz - qwx);
    float v31 = 2 * (qxz - qwy), v32 = 2 * (qyz + qwx), v33 = 1 - 2 * (qxx + qyy);
    /* B = A V (multAB, svd3.h:84-100) */
    float b11 = a11 * v11 + a12 * v21 + a13 * v31, b12 = a11 * v12 + a12 * v22 + a13 * v32,
          b13 = a11 * v13 + a12 * v23 + a13 * v33;
    float b21 = a21 * v11 + a22 * v21 + a23 * v31, b22 = a21 * v12 + a22 * v22 + a23 * v32,
          b23 = a21 * v13 + a22 * v23 + a23 * v33;
    float b31 = a31 * v11 + a32 * v21 + a33 * v31, b32 = a31 * v12 + a32 * v22 + a33 * v32,
          b33 = a31 * v13 + a32 * v23 + a33 * v33;
    /* sortSingularValues (svd3.h:247-274) */
    float rho1 = orc_dist2(b11, b21, b31), rho2 = orc_dist2(b12, b22, b32),
          rho3 = orc_dist2(b13, b23, b33);
    int c;
    c = rho1 < rho2;
    orc_cond_neg_swap(c, &b11, &b12); orc_cond_neg_swap(c, &v11, &v12);
    orc_cond_neg_swap(c, &b21, &b22); orc_cond_neg_swap(c, &v21, &v22);
    orc_cond_neg_swap(c, &b31, &b32); orc_cond_neg_swap(c, &v31, &v32);
    orc_cond_swap(c, &rho1, &rho2);
    c = rho1 < rho3;
    orc_cond_neg_swap(c, &b11, &b13); orc_cond_neg_swap(c, &v11, &v13);
    orc_cond_neg_swap(c, &b21, &b23); orc_cond_neg_swap(c, &v21, &v23);
    orc_cond_neg_swap(c, &b31, &b33); orc_cond_neg_swap(c, &v31, &v33);
    orc_cond_swap(c, &rho1, &rho3);
    c = rho2 < rho3;
    orc_cond_neg_swap(c, &b12, &b13); orc_cond_neg_swap(c, &v12, &v13);
    orc_cond_neg_swap(c, &b22, &b23); orc_cond_neg_swap(c, &v22, &v23);
    orc_cond_neg_swap(c, &b32, &b33); orc_cond_neg_swap(c, &v32, &v33);
    /* QRDecomposition (svd3.h:295-352) */
    float ch1, sh1, ch2, sh2, ch3, sh3, aa, bb;
    float r11, r12, r13, r21, r22, r23, r31, r32, r33;
    orc_qr_givens_quat(b11, b21, &ch1, &sh1);
    aa = 1 - 2 * sh1 * sh1;
    bb = 2 * ch1 * sh1;
    r11 = aa * b11 + bb * b21; r12 = aa * b12 + bb * b22; r13 = aa * b13 + bb * b23;
    r21 = -bb * b11 + aa * b21; r22 = -bb * b12 + aa * b22; r23 = -bb * b13 + aa * b23;
    r31 = b31; r32 = b32; r33 = b33;
    orc_qr_givens_quat(r11, r31, &ch2, &sh2);
    aa = 1 - 2 * sh2 * sh2;
    bb = 2 * ch2 * sh2;
    b11 = aa * r11 + bb * r31; b12 = aa * r12 + bb * r32; b13 = aa * r13 + bb * r33;
    b21 = r21; b22 = r22; b23 = r23;
    b31 = -bb * r11 + aa * r31; b32 = -bb * r12 + aa * r32; b33 = -bb * r13 + aa * r33;
    orc_qr_givens_quat(b22, b32, &ch3, &sh3);
    aa = 1 - 2 * sh3 * sh3;
    bb = 2 * ch3 * sh3;
    r11 = b11; r12 = b12; r13 = b13;
    r21 = aa * b21 + bb * b31; r22 = aa * b22 + bb * b32; r23 = aa * b23 + bb * b33;
    r31 = -bb * b21 + aa * b31; r32 = -bb * b22 + aa * b32; r33 = -bb * b23 + aa * b33;
    float sh12 = sh1 * sh1, sh22 = sh2 * sh2, sh32 = sh3 * sh3;
    u[0] = (-1 + 2 * sh12) * (-1 + 2 * sh22);
    u[1] = 4 * ch2 * ch3 * (-1 + 2 * sh12) * sh2 * sh3 + 2 * ch1 * sh1 * (-1 + 2 * sh32);
    u[2] = 4 * ch1 * ch3 * sh1 * sh3 - 2 * ch2 * (-1 + 2 * sh12) * sh2 * (-1 + 2 * sh32);
    u[3] = 2 * ch1 * sh1 * (1 - 2 * sh22);
    u[4] = -8 * ch1 * ch2 * ch3 * sh1 * sh2 * sh3 + (-1 + 2 * sh12) * (-1 + 2 * sh32);
    u[5] = -2 * ch3 * sh3 + 4 * sh1 * (ch3 * sh1 * sh3 + ch1 * ch2 * sh2 * (-1 + 2 * sh32));
    u[6] = 2 * ch2 * sh2;
    u[7] = 2 * ch3 * (1 - 2 * sh22) * sh3;
    u[8] = (-1 + 2 * sh22) * (-1 + 2 * sh32);
    sm[0] = r11; sm[1] = r12; sm[2] = r13; sm[3] = r21; sm[4] = r22; sm[5] = r23;
    sm[6] = r31; sm[7] = r32; sm[8] = r33;
    v[0] = v11; v[1] = v12; v[2] = v13; v[3] = v21; v[4] = v22; v[5] = v23;
    v[6] = v31; v[7] = v32; v[8] = v33;
}

/* ------------------------------------------------------------------ */
/* A7-A9  transformPointICP (kernel.cu:993-1093)                        */
/* ------------------------------------------------------------------ */
void orc_icp(const orc_node *tree, const float robot[3], const float start[3], const float *scan,
             int n_beams, float out_pose[3], float *dbg)
{
    int n = n_beams;
    float *tar = (float *)calloc((size_t)n * 4, sizeof(float)); /* H2: zero-filled dev_target */
    float *cor = (float *)calloc((size_t)n * 4, sizeof(float));
    float *W = (float *)calloc((size_t)n * 9, sizeof(float));
    int nvalid = 0;
    /* kernGetWallsKD (kernel.cu:974-991), with the PREVIOUS robotPos */
    for (int i = 0; i < n; i++) {
        float wx, wy;
        orc_clean_lidar_scan(i, scan[i], robot[2], &wx, &wy);
        if (fabsf(wx) < ORC_LIDAR_RANGE && fabsf(wy) < ORC_LIDAR_RANGE) {
            tar[4 * i + 0] = robot[0] + wx;
            tar[4 * i + 1] = robot[1] + wy;
            tar[4 * i + 2] = 0.0f;
            tar[4 * i + 3] = ORC_OCCUPIED_WEIGHT;
            nvalid++;
        }
    }
    /* findCorrespondenceKD (kernel.cu:874-922): all n slots */
    for (int i = 0; i < n; i++) {
        int b = orc_kd_traverse(tree, tar[4 * i], tar[4 * i + 1], tar[4 * i + 2], 0);
        cor[4 * i + 0] = tree[b].x;
        cor[4 * i + 1] = tree[b].y;
        cor[4 * i + 2] = tree[b].z;
        cor[4 * i + 3] = tree[b].w;
    }
    /* means over sizeTarget (kernel.cu:1025-1029) */
    float mu_tar[3], mu_cor[3];
    for (int k = 0; k < 3; k++) {
        mu_tar[k] = orc_sum_f32(tar + k, n, 4) / (float)n;
        mu_cor[k] = orc_sum_f32(cor + k, n, 4) / (float)n;
    }
    /* centre (kernel.cu:1031-1042; translate matrix * (p,1) == p + (-mu)), outer product (862-872) */
    for (int i = 0; i < n; i++) {
        float t[3], c[3];
        for (int k = 0; k < 3; k++) {
            t[k] = tar[4 * i + k] + (-mu_tar[k]);
            c[k] = cor[4 * i + k] + (-mu_cor[k]);
        }
        /* out = mat3(t*c.x, t*c.y, t*c.z): column j = t * c[j] */
        for (int j = 0; j < 3; j++)
            for (int r = 0; r < 3; r++) W[(size_t)i * 9 + j * 3 + r] = t[r] * c[j];
    }
    float Wm[9]; /* glm storage: Wm[col*3+row] */
    for (int e = 0; e < 9; e++) Wm[e] = orc_sum_f32(W + e, n, 9);
    /* svd(W[0][0], W[0][1], W[0][2], W[1][0], ...): a_rc = W[col r][row c] (kernel.cu:1058) */
    float A[9];
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) A[r * 3 + c] = Wm[r * 3 + c];
    float U[9], S[9], V[9];
    orc_svd3(A, U, S, V);
    /* g_U(row r, col c) = u_rc ; g_Vt(row r, col c) = v_cr (kernel.cu:1065-1066).
     * R = g_U * g_Vt with glm's evaluation order (type_mat3x3.inl:505-538):
     * R[col j][row i] = gU[0][i]*gVt[j][0] + gU[1][i]*gVt[j][1] + gU[2][i]*gVt[j][2] */
    float R[9]; /* glm storage R[col*3+row] */
    for (int j = 0; j < 3; j++)
        for (int i = 0; i < 3; i++)
            R[j * 3 + i] = U[i * 3 + 0] * V[j * 3 + 0] + U[i * 3 + 1] * V[j * 3 + 1] +
                           U[i * 3 + 2] * V[j * 3 + 2];
    /* t = mu_cor - R*mu_tar (mat3*vec3: m[0][i]*v.x + m[1][i]*v.y + m[2][i]*v.z) */
    float t[3];
    for (int i = 0; i < 3; i++)
        t[i] = mu_cor[i] - (R[0 * 3 + i] * mu_tar[0] + R[1 * 3 + i] * mu_tar[1] + R[2 * 3 + i] * mu_tar[2]);
    float theta = orc_asinf(R[0 * 3 + 1]); /* asin(R[0][1]) kernel.cu:1079 */
    out_pose[0] = start[0] + t[0];
    out_pose[1] = start[1] + t[1];
    out_pose[2] = start[2] + theta;
    if (dbg) {
        memcpy(dbg, A, sizeof(A));
        memcpy(dbg + 9, mu_tar, sizeof(mu_tar));
        memcpy(dbg + 12, mu_cor, sizeof(mu_cor));
        memcpy(dbg + 15, R, sizeof(R));
        memcpy(dbg + 24, t, sizeof(t));
        dbg[27] = theta;
        dbg[28] = (float)nvalid;
    }
    free(tar);
    free(cor);
    free(W);
}

/* ------------------------------------------------------------------ */
/* A10  traceRay (kernel.cu:190-240) and kernGetWalls (kernel.cu:524-549) */
/* ------------------------------------------------------------------ */
void orc_trace_ray(int sx, int sy, int ex, int ey, int dimx, int dimy, uint8_t *out)
{
    int dx0 = ex - sx, dy0 = ey - sy;
    int steep = abs(dy0) > abs(dx0);
    int tmp;
    if (steep) {
        tmp = sx; sx = sy; sy = tmp;
        tmp = ex; ex = ey; ey = tmp;
    }
    if (sx > ex) {
        tmp = sx; sx = ex; ex = tmp;
        tmp = sy; sy = ey; ey = tmp;
    }
    int deltax = ex - sx;
    int deltay = abs(ey - sy);
    float error = deltax / 2;
    int y = sy;
    int ystep = (ey > sy) ? 1 : -1;
    for (int x = sx; x < ex; x++) {
        int idx;
        if (steep)
            idx = y * dimx + x;
        else
            idx = x * dimx + y;
        if (x < dimx && y < dimy && x >= 0 && y >= 0 && idx < dimx * dimy) out[idx] = 1;
        error -= deltay;
        if (error < 0) {
            y += ystep;
            error += deltax;
        }
    }
}

void orc_get_walls(const float *scan, int n_beams, int cx, int cy, float theta, uint8_t *free_mask,
                   uint8_t *wall_mask, int dimx, int dimy, float res_x, float res_y)
{
    for (int i = 0; i < n_beams; i++) {
        float wx, wy;
        orc_clean_lidar_scan(i, scan[i], theta, &wx, &wy);
        if (fabsf(wx) < ORC_LIDAR_RANGE && fabsf(wy) < ORC_LIDAR_RANGE) {
            wx = roundf(wx / res_x);
            wy = roundf(wy / res_y);
            wx += (float)cx;
            wy += (float)cy;
            orc_trace_ray(cx, cy, (int)wx, (int)wy, dimx, dimy, free_mask);
            if (wx >= 0 && wx < dimx && wy >= 0 && wy < dimy)
                wall_mask[(int)(wx * dimx + wy)] = 1;
        }
    }
}

/* ------------------------------------------------------------------ */
/* A11  masks -> point lists (host loops, kernel.cu:1435-1461)          */
/* ------------------------------------------------------------------ */
static inline float orc_round_frac(float a, float frac) { return roundf((a / frac)) * frac; } /* kernel.cu:52 */

void orc_masks_to_points(const uint8_t *free_mask, const uint8_t *wall_mask, int dimx, int dimy,
                         const orc_patch *patch, const float robot[3], float *wall_xyzw,
                         int *n_wall, float *free_xyzw, int *n_free)
{
    int nw = 0, nf = 0;
    for (int x = 0; x < dimx; x++) {
        for (int y = 0; y < dimy; y++) {
            int idx = (x * dimx) + y;
            if (wall_mask[idx]) {
                float px = x * patch->res_x - patch->scale_x / 2.0f + robot[0];
                float py = y * patch->res_y - patch->scale_y / 2.0f + robot[1];
                px = orc_round_frac(px, patch->res_x);
                py = orc_round_frac(py, patch->res_y);
                wall_xyzw[4 * nw + 0] = px;
                wall_xyzw[4 * nw + 1] = py;
                wall_xyzw[4 * nw + 2] = 0.0f;
                wall_xyzw[4 * nw + 3] = 0.0f; /* glm::vec4() zero-initialises (type_vec4.inl:39-43) */
                nw++;
            }
            if (free_mask[idx]) {
                float px = x * patch->res_x - patch->scale_x / 2.0f + robot[0];
                float py = y * patch->res_y - patch->scale_y / 2.0f + robot[1];
                px = orc_round_frac(px, patch->res_x);
                py = orc_round_frac(py, patch->res_y);
                free_xyzw[4 * nf + 0] = px;
                free_xyzw[4 * nf + 1] = py;
                free_xyzw[4 * nf + 2] = 0.0f;
                free_xyzw[4 * nf + 3] = 0.0f;
                nf++;
            }
        }
    }
    *n_wall = nw;
    *n_free = nf;
}

/* A13 kernUpdateMapKD (kernel.cu:1350-1364); H4: every list entry applied in order */
void orc_update_map_kd(orc_node *tree, const float *pts, const int32_t *idx, int n, int val,
                       const orc_patch *patch)
{
    long clamp_val = (1 << (sizeof(int8_t) * 8 - 1)) - 15; /* 113 */
    float minDist = sqrtf(patch->res_x * patch->res_x + patch->res_y * patch->res_y);
    for (int i = 0; i < n; i++) {
        orc_node *nd = &tree[idx[i]];
        float d = orc_dist3(pts[4 * i], pts[4 * i + 1], pts[4 * i + 2], nd->x, nd->y, nd->z);
        if (d < minDist) {
            float v = nd->w + val;
            nd->w = (v < -clamp_val) ? -clamp_val : (v > clamp_val) ? clamp_val : v; /* CLAMP, kernel.cu:51 */
        }
    }
}

/* A14 kernTestCorrespondance (kernel.cu:1367-1379) */
void orc_test_correspondence(const orc_node *tree, const float *pts, const int32_t *idx, int n,
                             uint8_t *create, const orc_patch *patch)
{
    float minDist = sqrtf(patch->res_x * patch->res_x + patch->res_y * patch->res_y) / 2.0f;
    for (int i = 0; i < n; i++) {
        const orc_node *nd = &tree[idx[i]];
        float d = orc_dist3(pts[4 * i], pts[4 * i + 1], pts[4 * i + 2], nd->x, nd->y, nd->z);
        create[i] = (d > minDist);
    }
}

/* A15 KDTree::InsertNode (kdtree.cpp:69-105) */
void orc_kd_insert_node(const float p[4], orc_node *list, int list_size)
{
    int next = 0, parent = 0, axis = 0;
    do {
        parent = next;
        axis = (list[next].parent == -1) ? 0 : (list[list[next].parent].axis + 1) % 3;
        int lt = 0;
        if (axis == 0) lt = p[0] < list[next].x;
        if (axis == 1) lt = p[1] < list[next].y;
        if (axis == 2) lt = p[2] < list[next].z;
        next = lt ? list[next].left : list[next].right;
    } while (next != -1);
    int lt = 0;
    if (axis == 0) lt = p[0] < list[parent].x;
    if (axis == 1) lt = p[1] < list[parent].y;
    if (axis == 2) lt = p[2] < list[parent].z;
    if (lt)
        list[parent].left = list_size;
    else
        list[parent].right = list_size;
    orc_node nd;
    nd.left = -1;
    nd.right = -1;
    nd.parent = parent;
    nd.axis = (axis + 1) % 3;
    nd.x = p[0]; nd.y = p[1]; nd.z = p[2]; nd.w = p[3];
    list[list_size] = nd;
}

/* ------------------------------------------------------------------ */
/* A16  PFResample (kernel.cu:447-511) + kernWeightedSample (429-444)   */
/* ------------------------------------------------------------------ */
void orc_weighted_sample_indices(const float *cdf, int n, float neff, int frame, int i0, int count,
                                 int32_t *src_idx)
{
    float max = cdf[n - 1]; /* kernel.cu:481 */
    for (int k = 0; k < count; k++) {
        int i = i0 + k;
        uint32_t gen = orc_engine_seed((int)neff, frame, i); /* (iter=Neff, index=frame, depth=i), H5 */
        float rnd = orc_uniform_real(&gen, 0.0f, max);
        int idx = 0;
        while (idx < n && rnd > cdf[idx]) idx++;
        if (idx >= n) idx = n - 1; /* reference would read particles[N]; unreachable for finite weights */
        src_idx[k] = idx;
    }
}

int orc_resample(orc_particle *p, int n, int frame, float *neff_out, int32_t *src_idx)
{
    float *w = (float *)calloc((size_t)(n > 0 ? n : 1), sizeof(float));
    for (int i = 0; i < n; i++) w[i] = p[i].w * p[i].w; /* kernCopyWeights squared */
    float r2 = orc_sum_f32(w, n, 1);
    for (int i = 0; i < n; i++) w[i] = p[i].w;
    float r = orc_sum_f32(w, n, 1);
    float Neff = r * r / r2;
    if (neff_out) *neff_out = Neff;
    int did = 0;
    if (Neff < ORC_EFFECTIVE_PARTICLES * n) {
        float *cdf = (float *)malloc(sizeof(float) * (size_t)n);
        int32_t *src = (int32_t *)malloc(sizeof(int32_t) * (size_t)n);
        orc_particle *snap = (orc_particle *)malloc(sizeof(orc_particle) * (size_t)n);
        orc_inclusive_scan_f32(w, n, cdf);
        orc_weighted_sample_indices(cdf, n, Neff, frame, 0, n, src);
        memcpy(snap, p, sizeof(orc_particle) * (size_t)n); /* H3: gather from a snapshot */
        for (int i = 0; i < n; i++) {
            p[i] = snap[src[i]];
            p[i].w = 1.0f;
        }
        if (src_idx) memcpy(src_idx, src, sizeof(int32_t) * (size_t)n);
        free(cdf);
        free(src);
        free(snap);
        did = 1;
    }
    free(w);
    return did;
}

/* ------------------------------------------------------------------ */
/* A17/A18  2-D occupancy-grid path (GPU-branch semantics, H7)          */
/* ------------------------------------------------------------------ */
/* EvaluateParticle + mapCorrelation (kernel.cu:243-274) */
void orc_score_grid(const int8_t *grid, int dimx, int dimy, const orc_patch *patch,
                    const orc_particle *p, int n, const float *scan, int n_beams, int32_t *fit)
{
    for (int i = 0; i < n; i++) {
        int retv = 0;
        const orc_angle_parts T = orc_parts(p[i].theta);
        for (int j = 0; j < n_beams; j++) {
            float wx, wy;
            orc_clean_lidar_scan_pre(j, scan[j], p[i].theta, &T, &wx, &wy);
            wx += p[i].x;
            wy += p[i].y;
            wx = roundf(0.5f * patch->scale_x / patch->res_x + wx / patch->res_x);
            wy = roundf(0.5f * patch->scale_y / patch->res_y + wy / patch->res_y);
            if (wx >= 0 && wx < dimx && wy >= 0 && wy < dimy) {
                int idx = (int)wx * dimx + (int)wy;
                retv += grid[idx];
            }
        }
        fit[i] = retv;
    }
}

/* PFUpdateMap GPU branch (kernel.cu:551-577) + kernUpdateMap (513-522) */
void orc_update_map_grid(int8_t *grid, int dimx, int dimy, const orc_patch *patch,
                         const float robot[3], const float *scan, int n_beams)
{
    int cx = (int)roundf(0.5f * dimx + robot[0] / patch->res_x + patch->res_x / 2);
    int cy = (int)roundf(0.5f * dimy + robot[1] / patch->res_y + patch->res_y / 2);
    size_t M = (size_t)dimx * dimy;
    uint8_t *fm = (uint8_t *)calloc(M, 1), *wm = (uint8_t *)calloc(M, 1);
    orc_get_walls(scan, n_beams, cx, cy, robot[2], fm, wm, dimx, dimy, patch->res_x, patch->res_y);
    long clamp_val = (1 << (sizeof(int8_t) * 8 - 1)) - 15;
    for (size_t i = 0; i < M; i++)
        if (fm[i]) {
            long v = grid[i] + ORC_FREE_WEIGHT;
            grid[i] = (int8_t)((v < -clamp_val) ? -clamp_val : (v > clamp_val) ? clamp_val : v);
        }
    for (size_t i = 0; i < M; i++)
        if (wm[i]) {
            long v = grid[i] + ORC_OCCUPIED_WEIGHT;
            grid[i] = (int8_t)((v < -clamp_val) ? -clamp_val : (v > clamp_val) ? clamp_val : v);
        }
    free(fm);
    free(wm);
}

/* ------------------------------------------------------------------ */
/* Topology graph + loop-closure proposal (kernel.cu:623-795)           */
/* ------------------------------------------------------------------ */
#define ORC_MAX_NODE_DIST 2.5f        /* kernel.cu:34 */
#define ORC_WALL_CONFIDENCE 30        /* kernel.cu:36 */
#define ORC_MIN_WALL_COUNT 2          /* kernel.cu:37 */
#define ORC_CLOSURE_MAP_DIST 6.0f     /* kernel.cu:38 */
#define ORC_CLOSURE_GRAPH_DIST 20.0f  /* kernel.cu:39 */

static float orc_dist2d(const float a[2], const float b[2])
{ /* glm::distance(vec2, vec2) = sqrt(dx*dx + dy*dy) */
    float dx = b[0] - a[0], dy = b[1] - a[1];
    return sqrtf(dx * dx + dy * dy);
}

void orc_topology_init(orc_topology *t) /* particleFilterInit, kernel.cu:147-157 */
{
    memset(t, 0, sizeof(*t));
    t->n_nodes = 1;
    t->node_idx = 0;
}

static void orc_create_node(orc_topology *t, const float robot[3]) /* CreateNode, kernel.cu:623-648 */
{
    if (t->n_nodes >= ORC_TOPO_MAX_NODES) return;
    float pos[2] = {robot[0], robot[1]};
    float edgeLen = orc_dist2d(pos, t->pos[t->node_idx]);
    for (int j = 0; j < t->n_nodes; j++) t->dist[j] += edgeLen;
    int k = t->n_nodes++;
    t->pos[k][0] = pos[0];
    t->pos[k][1] = pos[1];
    t->dist[k] = 0.0f;
    t->n_edges[k] = 0;
    if (t->n_edges[k] < 8) t->edges[k][t->n_edges[k]++] = t->node_idx;
    if (t->n_edges[t->node_idx] < 8) t->edges[t->node_idx][t->n_edges[t->node_idx]++] = k;
    t->node_idx = k;
}

int orc_topology_update(orc_topology *t, const float robot[3])
{
    int newNode = 1;
    for (int j = 0; j < t->n_nodes; j++) newNode &= (orc_dist2d(robot, t->pos[j]) > ORC_MAX_NODE_DIST);
    if (newNode) orc_create_node(t, robot);
    return newNode;
}

int orc_find_walls(const int8_t *grid, int dimx, int dimy, const orc_patch *patch, const float a[2], const float b[2])
{
    int ax = (int)roundf(0.5f * dimx + a[0] / patch->res_x + patch->res_x / 2);
    int ay = (int)roundf(0.5f * dimy + a[1] / patch->res_y + patch->res_y / 2);
    int bx = (int)roundf(0.5f * dimx + b[0] / patch->res_x + patch->res_x / 2);
    int by = (int)roundf(0.5f * dimy + b[1] / patch->res_y + patch->res_y / 2);
    size_t M = (size_t)dimx * dimy;
    uint8_t *mask = (uint8_t *)calloc(M, 1);
    orc_trace_ray(ax, ay, bx, by, dimx, dimy, mask);
    int n = 0; /* CheckVisibility (kernel.cu:651-659); its += is racy in the reference, counted exactly here */
    for (size_t i = 0; i < M; i++)
        if (mask[i]) n += (grid[i] > ORC_WALL_CONFIDENCE) ? 1 : 0;
    free(mask);
    return n;
}

int orc_check_loop_closure(const orc_topology *t, const int8_t *grid, int dimx, int dimy, const orc_patch *patch,
                           const float robot[3], int32_t *pairs, int cap)
{
    int n = 0;
    for (int j = 0; j < t->n_nodes; j++) {
        if (orc_dist2d(robot, t->pos[j]) < ORC_CLOSURE_MAP_DIST) {
            float edgeLen = orc_dist2d(robot, t->pos[t->node_idx]);
            if (edgeLen + t->dist[j] > ORC_CLOSURE_GRAPH_DIST) {
                for (int k = 0; k < t->n_nodes; k++) {
                    int nWalls = orc_find_walls(grid, dimx, dimy, patch, robot, t->pos[k]);
                    if (nWalls < ORC_MIN_WALL_COUNT) {
                        if (n < cap) {
                            pairs[2 * n] = j;
                            pairs[2 * n + 1] = k;
                        }
                        n++;
                    }
                }
            }
        }
    }
    return n;
}

/* ------------------------------------------------------------------ */
/* Whole step: particleFilter (kernel.cu:1702-1762), KD path            */
/* ------------------------------------------------------------------ */
struct orc_slam {
    orc_slam_config cfg;
    int dimx, dimy;
    orc_particle *dev;  /* dev_particles */
    orc_particle *host; /* host `particles[]` mirror (kernel.cu:61) */
    orc_node *kd;
    int kd_size;
    float robot[3];
    float *fit;
    uint8_t *free_mask, *wall_mask;
    float *wall_pts, *free_pts;
    int32_t *wall_c, *free_c;
    uint8_t *create;
    int32_t trace[8];
    int n_wall, n_free;
    int8_t *grid;   /* occupancyGrid, kernel.cu:122-124 (allocated on first grid use) */
    int32_t *fit_i; /* dev_fit */
    /* UpdateTopology / CheckLoopClosure at the end of the frame (kernel.cu:1750-1751, commented out in the shipped step) */
    int topo_in_step;
    orc_topology *topo;
    int32_t *closures; /* pairs proposed by the last frame */
    int n_closures;
};

orc_slam *orc_slam_create(const orc_slam_config *cfg)
{
    orc_slam *s = (orc_slam *)calloc(1, sizeof(orc_slam));
    s->cfg = *cfg;
    /* map_dim (kernel.cu:120): int(scale / resolution) in float */
    s->dimx = (int)(cfg->patch.scale_x / cfg->patch.res_x);
    s->dimy = (int)(cfg->patch.scale_y / cfg->patch.res_y);
    int n = cfg->n_particles;
    s->dev = (orc_particle *)calloc((size_t)n, sizeof(orc_particle));
    s->host = (orc_particle *)calloc((size_t)n, sizeof(orc_particle));
    for (int i = 0; i < n; i++) { /* kernel.cu:126-130 */
        s->host[i].x = s->host[i].y = s->host[i].theta = 0.0f;
        s->host[i].w = 1.0f;
        s->host[i].cluster = 0;
    }
    memcpy(s->dev, s->host, sizeof(orc_particle) * (size_t)n);
    s->kd = (orc_node *)calloc((size_t)cfg->kd_capacity, sizeof(orc_node));
    s->fit = (float *)calloc((size_t)n, sizeof(float));
    size_t M = (size_t)s->dimx * s->dimy;
    s->free_mask = (uint8_t *)calloc(M, 1);
    s->wall_mask = (uint8_t *)calloc(M, 1);
    s->wall_pts = (float *)calloc(M * 4, sizeof(float));
    s->free_pts = (float *)calloc(M * 4, sizeof(float));
    s->wall_c = (int32_t *)calloc(M, sizeof(int32_t));
    s->free_c = (int32_t *)calloc(M, sizeof(int32_t));
    s->create = (uint8_t *)calloc(M, 1);
    return s;
}

void orc_slam_destroy(orc_slam *s)
{
    if (!s) return;
    free(s->dev); free(s->host); free(s->kd); free(s->fit); free(s->free_mask); free(s->wall_mask);
    free(s->wall_pts); free(s->free_pts); free(s->wall_c); free(s->free_c); free(s->create);
    free(s->grid); free(s->fit_i); free(s->topo); free(s->closures);
    free(s);
}

void orc_slam_set_map(orc_slam *s, const orc_node *tree, int n)
{
    memcpy(s->kd, tree, sizeof(orc_node) * (size_t)n);
    s->kd_size = n;
}

/* PFUpdateMapKD (kernel.cu:1406-1540) */
static void orc_pf_update_map_kd(orc_slam *s, const float *scan)
{
    const orc_patch *pa = &s->cfg.patch;
    int cx = (int)roundf(0.5f * s->dimx + pa->res_x / 2); /* kernel.cu:1408-1411 */
    int cy = (int)roundf(0.5f * s->dimy + pa->res_y / 2);
    size_t M = (size_t)s->dimx * s->dimy;
    memset(s->free_mask, 0, M);
    memset(s->wall_mask, 0, M);
    orc_get_walls(scan, s->cfg.n_beams, cx, cy, s->robot[2], s->free_mask, s->wall_mask, s->dimx,
                  s->dimy, pa->res_x, pa->res_y);
    int nw, nf;
    orc_masks_to_points(s->free_mask, s->wall_mask, s->dimx, s->dimy, pa, s->robot, s->wall_pts, &nw,
                        s->free_pts, &nf);
    s->n_wall = nw;
    s->n_free = nf;
    int n_insert = 0;
    if (s->kd_size > 0) {
        if (s->cfg.free_upload_bug) { /* H6: only wallPC.size() entries uploaded, tail zero */
            for (int i = nw; i < nf; i++)
                s->free_pts[4 * i] = s->free_pts[4 * i + 1] = s->free_pts[4 * i + 2] = s->free_pts[4 * i + 3] = 0.0f;
        }
        for (int i = 0; i < nf; i++)
            s->free_c[i] = orc_kd_traverse(s->kd, s->free_pts[4 * i], s->free_pts[4 * i + 1], s->free_pts[4 * i + 2], 0);
        for (int i = 0; i < nw; i++)
            s->wall_c[i] = orc_kd_traverse(s->kd, s->wall_pts[4 * i], s->wall_pts[4 * i + 1], s->wall_pts[4 * i + 2], 0);
        orc_update_map_kd(s->kd, s->free_pts, s->free_c, nf, ORC_FREE_WEIGHT, pa);
        orc_update_map_kd(s->kd, s->wall_pts, s->wall_c, nw, ORC_OCCUPIED_WEIGHT, pa);
        orc_test_correspondence(s->kd, s->wall_pts, s->wall_c, nw, s->create, pa);
        for (int i = 0; i < nw; i++) {
            if (s->create[i] && s->kd_size < s->cfg.kd_capacity) {
                float p[4] = {s->wall_pts[4 * i], s->wall_pts[4 * i + 1], s->wall_pts[4 * i + 2], -100.0f};
                orc_kd_insert_node(p, s->kd, s->kd_size++);
                n_insert++;
            }
        }
    } else if (nw > 0) {
        orc_kd_create(s->wall_pts, nw, s->kd); /* kernel.cu:1533 */
        s->kd_size += nw;
    }
    s->trace[2] = nw;
    s->trace[3] = nf;
    s->trace[4] = n_insert;
}

/* The two calls the reference leaves commented out at the end of particleFilter (kernel.cu:1750-1751), where they stand.
 * FindWalls reads the 2-D occupancy grid (dev_occupancyGrid, kernel.cu:680), which the KD path never updates: there it stays
 * at its initial -100 and every node is "visible"; in the 2-D frame loop it is the live map. */
#define ORC_MAX_CLOSURES 65536
static void orc_slam_grid_alloc(orc_slam *s);
static void orc_slam_frame_topology(orc_slam *s)
{
    if (!s->topo_in_step) return;
    orc_slam_grid_alloc(s);
    orc_topology_update(s->topo, s->robot);
    int n = orc_check_loop_closure(s->topo, s->grid, s->dimx, s->dimy, &s->cfg.patch, s->robot, s->closures, ORC_MAX_CLOSURES);
    s->n_closures = n < ORC_MAX_CLOSURES ? n : ORC_MAX_CLOSURES;
}
void orc_slam_set_topology(orc_slam *s, int enable)
{
    s->topo_in_step = enable;
    if (enable && !s->topo) {
        s->topo = (orc_topology *)calloc(1, sizeof(orc_topology));
        orc_topology_init(s->topo);
        s->closures = (int32_t *)calloc((size_t)2 * ORC_MAX_CLOSURES, sizeof(int32_t));
    }
    s->n_closures = 0;
}
int orc_slam_last_closures(const orc_slam *s, int32_t *pairs, int cap)
{
    int n = s->n_closures < cap ? s->n_closures : cap;
    if (n > 0) memcpy(pairs, s->closures, (size_t)n * 8);
    return s->n_closures;
}
const orc_topology *orc_slam_topology(const orc_slam *s) { return s->topo; }

void orc_slam_step(orc_slam *s, int frame, const float *scan)
{
    int n = s->cfg.n_particles;
    memset(s->trace, 0, sizeof(s->trace));
    s->trace[0] = -1;
    if (s->cfg.balance_period > 0 && frame % s->cfg.balance_period == 5 && s->kd_size > 0)
        orc_kd_balance(s->kd, s->kd_size); /* kernel.cu:1707-1711 */
    if (s->kd_size == 0) { /* kernel.cu:1714-1717 */
        s->robot[0] = s->robot[1] = s->robot[2] = 0.0f;
        orc_pf_update_map_kd(s, scan);
    } else {
        /* PFMotionUpdate (kernel.cu:400-418): H2D host->dev, noise, D2H */
        memcpy(s->dev, s->host, sizeof(orc_particle) * (size_t)n);
        orc_add_noise(s->dev, n, frame, 0);
        memcpy(s->host, s->dev, sizeof(orc_particle) * (size_t)n);
        /* PFMeasurementUpdateKD (kernel.cu:1311-1348) */
        {   /* ORC_THREADS=k: particles scored on k threads (same per-particle arithmetic; soak tests on the GPU box) */
            const char *e = getenv("ORC_THREADS");
            int nt = e ? atoi(e) : 1;
            if (nt > 1) orc_score_kd_mt(s->kd, s->dev, n, scan, s->cfg.n_beams, s->fit, nt);
            else orc_score_kd(s->kd, s->dev, n, scan, s->cfg.n_beams, s->fit, 0, 0);
        }
        int imin, imax;
        orc_minmax_first_f32(s->fit, n, &imin, &imax);
        float rng = s->fit[imax] - s->fit[imin];
        int best = imax;
        if (rng > 0.0f) {
            float f = 1 / rng;
            orc_update_weights_f32(s->dev, n, s->fit, f, (int)s->fit[imin]);
        }
        if (s->cfg.strict_host_mirror) {
            /* H11: cudaMemcpy(particles, dev_particles, N*sizeof(glm::vec4)) copies only the first
             * 16*N bytes of the 32-byte particles (kernel.cu:1341) */
            memcpy(s->host, s->dev, (size_t)n * 16);
        } else {
            memcpy(s->host, s->dev, sizeof(orc_particle) * (size_t)n);
        }
        float start[3] = {s->host[best].x, s->host[best].y, s->host[best].theta};
        float pose[3];
        orc_icp(s->kd, s->robot, start, scan, s->cfg.n_beams, pose, 0);
        s->robot[0] = pose[0];
        s->robot[1] = pose[1];
        s->robot[2] = pose[2];
        s->trace[0] = best;
        /* PFUpdateMapKD */
        orc_pf_update_map_kd(s, scan);
        /* PFResample (kernel.cu:447-511): on dev particles; D2H only if resampled */
        float neff;
        int did = orc_resample(s->dev, n, frame, &neff, 0);
        if (did) memcpy(s->host, s->dev, sizeof(orc_particle) * (size_t)n);
        s->trace[1] = did;
        memcpy(&s->trace[5], &neff, 4);
        orc_slam_frame_topology(s); /* //UpdateTopology(); //CheckLoopClosure(); kernel.cu:1750-1751 */
    }
    s->trace[6] = s->kd_size;
}

/* 2-D occupancy-grid variant of the step.  The reference defines the four stages (PFMotionUpdate 400-418,
 * PFMeasurementUpdate 307-339 GPU branch, PFUpdateMap 551-577 GPU branch, PFResample 447-511) but its shipped
 * particleFilter() calls the KD versions; this is the same frame loop with the 2-D stages in their places
 * (SURVEY 3.3).  Every frame runs all four stages: there is no first-frame special case because the grid
 * starts at -100 everywhere (kernel.cu:124), which scores every particle equally. */
static void orc_slam_grid_alloc(orc_slam *s)
{
    if (s->grid) return;
    size_t M = (size_t)s->dimx * s->dimy;
    s->grid = (int8_t *)malloc(M);
    memset(s->grid, -100, M);
    s->fit_i = (int32_t *)calloc((size_t)s->cfg.n_particles, sizeof(int32_t));
}
void orc_slam_set_grid(orc_slam *s, const int8_t *grid)
{
    orc_slam_grid_alloc(s);
    memcpy(s->grid, grid, (size_t)s->dimx * s->dimy);
}
const int8_t *orc_slam_grid(orc_slam *s)
{
    orc_slam_grid_alloc(s);
    return s->grid;
}
void orc_slam_step_grid(orc_slam *s, int frame, const float *scan)
{
    int n = s->cfg.n_particles;
    orc_slam_grid_alloc(s);
    memset(s->trace, 0, sizeof(s->trace));
    /* PFMotionUpdate */
    memcpy(s->dev, s->host, sizeof(orc_particle) * (size_t)n);
    orc_add_noise(s->dev, n, frame, 0);
    memcpy(s->host, s->dev, sizeof(orc_particle) * (size_t)n);
    /* PFMeasurementUpdate, GPU branch (kernel.cu:309-339) */
    orc_score_grid(s->grid, s->dimx, s->dimy, &s->cfg.patch, s->dev, n, scan, s->cfg.n_beams, s->fit_i);
    int imin, imax;
    orc_minmax_first_i32(s->fit_i, n, &imin, &imax);
    int rng = s->fit_i[imax] - s->fit_i[imin];
    int best = imax;
    if (rng > 0) {
        float f = 1 / (float)rng;
        orc_update_weights_i32(s->dev, n, s->fit_i, f, s->fit_i[imin]);
    }
    if (s->cfg.strict_host_mirror)
        memcpy(s->host, s->dev, (size_t)n * 16); /* same half-array copy as the KD path (kernel.cu:337) */
    else
        memcpy(s->host, s->dev, sizeof(orc_particle) * (size_t)n);
    s->robot[0] = s->host[best].x;
    s->robot[1] = s->host[best].y;
    s->robot[2] = s->host[best].theta;
    s->trace[0] = best;
    /* PFUpdateMap, GPU branch */
    orc_update_map_grid(s->grid, s->dimx, s->dimy, &s->cfg.patch, s->robot, scan, s->cfg.n_beams);
    /* PFResample */
    float neff;
    int did = orc_resample(s->dev, n, frame, &neff, 0);
    if (did) memcpy(s->host, s->dev, sizeof(orc_particle) * (size_t)n);
    s->trace[1] = did;
    memcpy(&s->trace[5], &neff, 4);
    orc_slam_frame_topology(s);
}

/* The reference's own CPU path of the 2-D frame loop: the `GPU_* == 0` branches (H7 semantics, which DIFFER from the GPU
 * branches above and are therefore a timing baseline for BASELINE configs[0], not the parity target):
 *   PFMotionUpdate      kernel.cu:414-417   ParticleAddNoise on the host array, same per-particle engines
 *   PFMeasurementUpdate kernel.cu:340-369   strict `>` / `<` scan for best / worst, w *= (float)(fit - worst) / (float)(best - worst)
 *   PFUpdateMap         kernel.cu:578-620   no 20 m reject; a ray is traced only if its end cell is inside the map; free cells
 *                                           get -1 once per cell (mask), wall cells +4 once per BEAM; clamp +-113
 *   PFResample          kernel.cu:463-508   sequential sums, sequential prefix sum, ONE engine seeded (Neff, frame, 0) drawing N
 *                                           numbers, in-place sequential copies (later picks see earlier overwrites)
 * There is one particle array only (the host one). */
void orc_slam_step_grid_cpu(orc_slam *s, int frame, const float *scan)
{
    const int n = s->cfg.n_particles, nb = s->cfg.n_beams, dimx = s->dimx, dimy = s->dimy;
    const orc_patch *pa = &s->cfg.patch;
    orc_particle *p = s->host;
    orc_slam_grid_alloc(s);
    memset(s->trace, 0, sizeof(s->trace));
    orc_add_noise(p, n, frame, 0); /* kernel.cu:414-417 */
    /* PFMeasurementUpdate, CPU branch */
    int best = -128 * nb, worst = 128 * nb, iBest = 0;
    orc_score_grid(s->grid, dimx, dimy, pa, p, n, scan, nb, s->fit_i); /* EvaluateParticle is __host__ __device__ */
    for (int i = 0; i < n; i++) {
        if (s->fit_i[i] > best) {
            best = s->fit_i[i];
            iBest = i;
        }
        if (s->fit_i[i] < worst) worst = s->fit_i[i];
    }
    if ((best - worst) > 0)
        for (int i = 0; i < n; i++) {
            float f = (float)(s->fit_i[i] - worst) / (float)(best - worst);
            p[i].w *= f;
        }
    s->robot[0] = p[iBest].x;
    s->robot[1] = p[iBest].y;
    s->robot[2] = p[iBest].theta;
    s->trace[0] = iBest;
    /* PFUpdateMap, CPU branch */
    {
        const int cx = (int)roundf(0.5f * dimx + s->robot[0] / pa->res_x + pa->res_x / 2);
        const int cy = (int)roundf(0.5f * dimy + s->robot[1] / pa->res_y + pa->res_y / 2);
        const size_t M = (size_t)dimx * dimy;
        const long clamp_val = (1 << (sizeof(int8_t) * 8 - 1)) - 15;
        memset(s->free_mask, 0, M);
        float *wx = s->wall_pts, *wy = s->wall_pts + nb; /* glm::vec2 walls[LIDAR_SIZE] */
        for (int i = 0; i < nb; i++) {
            orc_clean_lidar_scan(i, scan[i], s->robot[2], &wx[i], &wy[i]);
            wx[i] = roundf(wx[i] / pa->res_x);
            wy[i] = roundf(wy[i] / pa->res_y);
            wx[i] += (float)cx;
            wy[i] += (float)cy;
            if (wx[i] >= 0 && wx[i] < dimx && wy[i] >= 0 && wy[i] < dimy)
                orc_trace_ray(cx, cy, (int)wx[i], (int)wy[i], dimx, dimy, s->free_mask);
        }
        for (size_t idx = 0; idx < M; idx++)
            if (s->free_mask[idx]) {
                long v = s->grid[idx] + ORC_FREE_WEIGHT;
                s->grid[idx] = (int8_t)((v < -clamp_val) ? -clamp_val : (v > clamp_val) ? clamp_val : v);
            }
        for (int i = 0; i < nb; i++)
            if (wx[i] >= 0 && wx[i] < dimx && wy[i] >= 0 && wy[i] < dimy) {
                int idx = (int)wx[i] * dimx + (int)wy[i];
                long v = s->grid[idx] + ORC_OCCUPIED_WEIGHT;
                s->grid[idx] = (int8_t)((v < -clamp_val) ? -clamp_val : (v > clamp_val) ? clamp_val : v);
            }
    }
    /* PFResample, CPU branch */
    float r = 0, r2 = 0;
    for (int i = 0; i < n; i++) {
        r += p[i].w;
        r2 += (p[i].w) * (p[i].w);
    }
    float Neff = r * r / r2;
    int did = 0;
    if (Neff < ORC_EFFECTIVE_PARTICLES * n) {
        float *weightsum = s->fit; /* float weightsum[PARTICLE_COUNT] */
        weightsum[0] = p[0].w;
        for (int i = 1; i < n; i++) weightsum[i] = weightsum[i - 1] + p[i].w;
        uint32_t gen = orc_engine_seed((int)Neff, frame, 0);
        for (int i = 0; i < n; i++) {
            int idx = 0;
            float rnd = orc_uniform_real(&gen, 0.0f, weightsum[n - 1]);
            while (idx < n && rnd > weightsum[idx]) idx++;
            if (idx >= n) idx = n - 1; /* the reference would read particles[N] */
            p[i] = p[idx];
            p[i].w = 1.0f;
        }
        did = 1;
    }
    memcpy(s->dev, p, sizeof(orc_particle) * (size_t)n); /* "push particles to GPU to draw" */
    s->trace[1] = did;
    memcpy(&s->trace[5], &Neff, 4);
}

void orc_slam_set_particles(orc_slam *s, const orc_particle *p)
{
    memcpy(s->dev, p, sizeof(orc_particle) * (size_t)s->cfg.n_particles);
    memcpy(s->host, p, sizeof(orc_particle) * (size_t)s->cfg.n_particles);
}
/* odometry hook of the harness (no reference counterpart: the reference's filter has no motion model besides the diffusion of
 * kernel.cu:375-397): every pose the filter holds -- all particles, both copies, and robotPos -- moves by the same increment,
 * one float addition per component */
void orc_slam_shift_particles(orc_slam *s, const float d[3])
{
    for (int i = 0; i < s->cfg.n_particles; i++) {
        s->dev[i].x = s->dev[i].x + d[0]; s->dev[i].y = s->dev[i].y + d[1]; s->dev[i].theta = s->dev[i].theta + d[2];
        s->host[i].x = s->dev[i].x; s->host[i].y = s->dev[i].y; s->host[i].theta = s->dev[i].theta;
    }
    for (int k = 0; k < 3; k++) s->robot[k] = s->robot[k] + d[k];
}
void orc_slam_get_pose(const orc_slam *s, float pose[3]) { memcpy(pose, s->robot, 12); }
int orc_slam_kd_size(const orc_slam *s) { return s->kd_size; }
const orc_node *orc_slam_tree(const orc_slam *s) { return s->kd; }
const orc_particle *orc_slam_particles(const orc_slam *s) { return s->dev; }
void orc_slam_last_trace(const orc_slam *s, int32_t out[8]) { memcpy(out, s->trace, sizeof(s->trace)); }
int orc_slam_last_cells(const orc_slam *s, int which, int32_t *out, int cap)
{
    const uint8_t *m = which == 0 ? s->wall_mask : s->free_mask;
    size_t M = (size_t)s->dimx * s->dimy;
    int k = 0;
    for (size_t i = 0; i < M; i++)
        if (m[i]) {
            if (k < cap) out[k] = (int32_t)i;
            k++;
        }
    return k;
}
