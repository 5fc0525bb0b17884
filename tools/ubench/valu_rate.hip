// valu_rate.hip -- issue cost of the VALU instructions the traversal uses, gfx950, 8 waves per SIMD, independent chains.
//   hipcc --offload-arch=gfx950 -O3 -o valu_rate tools/ubench/valu_rate.hip && ./valu_rate
// Prints cycles per wave-instruction per SIMD (kernel cycles / instructions issued on one SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP8(x) x x x x x x x x
#define BODY(ASM) \
    for (int it = 0; it < iters; it++) { REP8(REP8(asm volatile(ASM : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));)) }

typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int OP>
__global__ __launch_bounds__(256) void k_valu(int iters, float *out)
{
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, b = 1.0000001f, c = 1e-30f;
    f32x2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a1, a0}, p3 = {a3, a2}, pb = {b, b};
    unsigned long long m = 0x5555555555555555ull;
    if (OP == 0) BODY("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5")
    if (OP == 1) BODY("v_mul_f32 %0, %0, %4\n v_mul_f32 %1, %1, %4\n v_mul_f32 %2, %2, %4\n v_mul_f32 %3, %3, %4")
    if (OP == 2) BODY("v_add_f32 %0, %0, %5\n v_add_f32 %1, %1, %5\n v_add_f32 %2, %2, %5\n v_add_f32 %3, %3, %5")
    if (OP == 3) BODY("v_lshlrev_b32 %0, 1, %0\n v_lshlrev_b32 %1, 1, %1\n v_lshlrev_b32 %2, 1, %2\n v_lshlrev_b32 %3, 1, %3")
    if (OP == 4) BODY("v_bfe_i32 %0, %0, 0, 30\n v_bfe_i32 %1, %1, 0, 30\n v_bfe_i32 %2, %2, 0, 30\n v_bfe_i32 %3, %3, 0, 30")
    if (OP == 5) BODY("v_cvt_f32_i32 %0, %0\n v_cvt_f32_i32 %1, %1\n v_cvt_f32_i32 %2, %2\n v_cvt_f32_i32 %3, %3")
    if (OP == 6) {
        for (int it = 0; it < iters; it++) {
            REP8(REP8(asm volatile("v_cndmask_b32 %0, %0, %4, %5\n v_cndmask_b32 %1, %1, %4, %5\n v_cndmask_b32 %2, %2, %4, %5\n v_cndmask_b32 %3, %3, %4, %5"
                                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "s"(m));))
        }
    }
    if (OP == 7) {
        for (int it = 0; it < iters; it++) {
            REP8(REP8(asm volatile("v_cmp_lt_f32 vcc, %0, %4\n v_cmp_lt_f32 vcc, %1, %4\n v_cmp_lt_f32 vcc, %2, %4\n v_cmp_lt_f32 vcc, %3, %4"
                                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b) : "vcc");))
        }
    }
    if (OP == 8) {
        for (int it = 0; it < iters; it++) {
            REP8(REP8(asm volatile("v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4"
                                   : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pb));))
        }
        a0 = p0.x + p1.y; a1 = p2.x; a2 = p3.y; a3 = 0;
    }
    if (OP == 9) {
        for (int it = 0; it < iters; it++) {
            REP8(REP8(asm volatile("v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4"
                                   : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pb));))
        }
        a0 = p0.x + p1.y; a1 = p2.x; a2 = p3.y; a3 = 0;
    }
    if (OP == 10) {
        for (int it = 0; it < iters; it++) {
            REP8(REP8(asm volatile("v_cmp_lt_f32 %4, %0, %5\n v_cmp_lt_f32 %4, %1, %5\n v_cmp_lt_f32 %4, %2, %5\n v_cmp_lt_f32 %4, %3, %5"
                                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+s"(m) : "v"(b));))
        }
    }
    if (OP >= 11 && OP <= 13) { // fp64: the end point of the scan-match kernel (sincos_sum_spec) is 14 of these per (lane, beam)
        double d0 = threadIdx.x, d1 = d0 + 1, d2 = d0 + 2, d3 = d0 + 3, db = 1.0000001, dc = 1e-30;
        for (int it = 0; it < iters; it++) {
            if (OP == 11) { REP8(REP8(asm volatile("v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(db), "v"(dc));)) }
            if (OP == 12) { REP8(REP8(asm volatile("v_mul_f64 %0, %0, %4\n v_mul_f64 %1, %1, %4\n v_mul_f64 %2, %2, %4\n v_mul_f64 %3, %3, %4" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(db), "v"(dc));)) }
            if (OP == 13) { REP8(REP8(asm volatile("v_add_f64 %0, %0, %5\n v_add_f64 %1, %1, %5\n v_add_f64 %2, %2, %5\n v_add_f64 %3, %3, %5" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(db), "v"(dc));)) }
        }
        a0 = (float)(d0 + d1 + d2 + d3);
    }
    if (OP == 14) BODY("v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4")
    // round 5, second table: which of the other instructions of the scan-match kernel's loop belong to the 2.4-cycle class
#define ONE2(OPN, I) if (OP == OPN) BODY(I " %0, %0, %4\n " I " %1, %1, %4\n " I " %2, %2, %4\n " I " %3, %3, %4")
#define ONE1(OPN, I) if (OP == OPN) BODY(I " %0, %0\n " I " %1, %1\n " I " %2, %2\n " I " %3, %3")
    ONE1(15, "v_mov_b32") ONE2(16, "v_min_f32") ONE2(17, "v_max_f32") ONE2(18, "v_and_b32") ONE2(19, "v_or_b32") ONE2(20, "v_sub_f32")
    ONE2(21, "v_sub_u32") ONE2(28, "v_xor_b32") ONE2(29, "v_lshrrev_b32") ONE2(30, "v_subrev_u32") ONE2(31, "v_mul_u32_u24") ONE2(32, "v_min_u32")
    ONE1(23, "v_cvt_flr_i32_f32") ONE1(24, "v_fract_f32") ONE1(33, "v_floor_f32") ONE1(34, "v_cvt_i32_f32") ONE1(35, "v_rcp_f32") ONE1(36, "v_sqrt_f32")
    if (OP == 22) BODY("v_mad_u32_u24 %0, %0, %4, %5\n v_mad_u32_u24 %1, %1, %4, %5\n v_mad_u32_u24 %2, %2, %4, %5\n v_mad_u32_u24 %3, %3, %4, %5")
    if (OP == 25) BODY("v_lshl_or_b32 %0, %0, 1, %4\n v_lshl_or_b32 %1, %1, 1, %4\n v_lshl_or_b32 %2, %2, 1, %4\n v_lshl_or_b32 %3, %3, 1, %4")
    if (OP == 37) BODY("v_fmac_f32 %0, %4, %5\n v_fmac_f32 %1, %4, %5\n v_fmac_f32 %2, %4, %5\n v_fmac_f32 %3, %4, %5")
    if (OP == 38) BODY("v_add3_u32 %0, %0, %4, %5\n v_add3_u32 %1, %1, %4, %5\n v_add3_u32 %2, %2, %4, %5\n v_add3_u32 %3, %3, %4, %5")
    if (OP == 39) BODY("v_med3_f32 %0, %0, %4, %5\n v_med3_f32 %1, %1, %4, %5\n v_med3_f32 %2, %2, %4, %5\n v_med3_f32 %3, %3, %4, %5")
    if (OP == 40) BODY("v_bfi_b32 %0, %0, %4, %5\n v_bfi_b32 %1, %1, %4, %5\n v_bfi_b32 %2, %2, %4, %5\n v_bfi_b32 %3, %3, %4, %5")
    if (OP == 41) BODY("v_and_or_b32 %0, %0, %4, %5\n v_and_or_b32 %1, %1, %4, %5\n v_and_or_b32 %2, %2, %4, %5\n v_and_or_b32 %3, %3, %4, %5")
    if (OP == 42) {
        for (int it = 0; it < iters; it++) { // select on vcc (the compiler's usual form)
            REP8(REP8(asm volatile("v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %1, %1, %4, vcc\n v_cndmask_b32 %2, %2, %4, vcc\n v_cndmask_b32 %3, %3, %4, vcc"
                                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b) : "vcc");))
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + (float)(m & 1);
}

template <int OP>
static void run(const char *name, float *out, int cus, double ghz)
{
    const int blocks = cus * 8, iters = 200; // 8 blocks of 4 waves per CU = 8 waves per SIMD
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    k_valu<OP><<<blocks, 256>>>(10, out);
    (void)hipEventRecord(a);
    k_valu<OP><<<blocks, 256>>>(iters, out);
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, a, b);
    const double instr_per_simd = 8.0 * iters * 64 * 4; // 8 waves x iters x 64 asm blocks x 4 instructions
    printf("%-22s %.3f ms -> %.2f cycles per wave-instruction per SIMD\n", name, ms, ms * 1e-3 * ghz * 1e9 / instr_per_simd);
}

int main()
{
    hipDeviceProp_t p;
    (void)hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    const double ghz = p.clockRate * 1e-6;
    float *out;
    (void)hipMalloc(&out, (size_t)cus * 8 * 256 * 4);
    printf("%d CUs, %.2f GHz nominal\n", cus, ghz);
    run<0>("v_fma_f32", out, cus, ghz); run<1>("v_mul_f32", out, cus, ghz); run<2>("v_add_f32", out, cus, ghz);
    run<3>("v_lshlrev_b32", out, cus, ghz); run<4>("v_bfe_i32", out, cus, ghz); run<5>("v_cvt_f32_i32", out, cus, ghz);
    run<6>("v_cndmask_b32 (sgpr)", out, cus, ghz); run<7>("v_cmp_lt_f32 vcc", out, cus, ghz); run<10>("v_cmp_lt_f32 sgpr", out, cus, ghz);
    run<8>("v_pk_mul_f32", out, cus, ghz); run<9>("v_pk_add_f32", out, cus, ghz);
    run<11>("v_fma_f64", out, cus, ghz); run<12>("v_mul_f64", out, cus, ghz); run<13>("v_add_f64", out, cus, ghz);
    run<14>("v_add_u32", out, cus, ghz);
    run<15>("v_mov_b32", out, cus, ghz); run<16>("v_min_f32", out, cus, ghz); run<17>("v_max_f32", out, cus, ghz); run<20>("v_sub_f32", out, cus, ghz);
    run<37>("v_fmac_f32", out, cus, ghz); run<39>("v_med3_f32", out, cus, ghz);
    run<18>("v_and_b32", out, cus, ghz); run<19>("v_or_b32", out, cus, ghz); run<28>("v_xor_b32", out, cus, ghz); run<29>("v_lshrrev_b32", out, cus, ghz);
    run<21>("v_sub_u32", out, cus, ghz); run<30>("v_subrev_u32", out, cus, ghz); run<32>("v_min_u32", out, cus, ghz); run<31>("v_mul_u32_u24", out, cus, ghz);
    run<22>("v_mad_u32_u24", out, cus, ghz); run<25>("v_lshl_or_b32", out, cus, ghz); run<38>("v_add3_u32", out, cus, ghz); run<40>("v_bfi_b32", out, cus, ghz);
    run<41>("v_and_or_b32", out, cus, ghz); run<42>("v_cndmask_b32 (vcc)", out, cus, ghz);
    run<23>("v_cvt_flr_i32_f32", out, cus, ghz); run<24>("v_fract_f32", out, cus, ghz); run<33>("v_floor_f32", out, cus, ghz); run<34>("v_cvt_i32_f32", out, cus, ghz);
    run<35>("v_rcp_f32", out, cus, ghz); run<36>("v_sqrt_f32", out, cus, ghz);
    return 0;
}
