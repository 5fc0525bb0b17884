// gather_rate.hip -- what does one wave-level gather cost on a CU of MI355X?  Dependent-free gathers from an L2-resident table
// (2 MB, like the hot tree), 8 waves per SIMD, every lane a pseudo-random (or wave-uniform, or pairwise-adjacent) index.
//   hipcc --offload-arch=gfx950 -O3 -o gather_rate tools/ubench/gather_rate.hip && ./gather_rate
// Prints cycles per wave-instruction per CU (= kernel cycles * CUs / total wave-instructions) for dword / x2 / x3 / x4 loads.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x3 __attribute__((ext_vector_type(3)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int WIDTH, int MODE> // MODE 0 random per lane, 1 wave-uniform, 2 lanes in groups of 4 share a 64-byte line
__global__ __launch_bounds__(256) void k_gather(const unsigned *table, int n_rec, int iters, unsigned *out)
{
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void *)table, 0, 0x7ffffff0, 0x00020000);
    unsigned lane = threadIdx.x & 63, gid = blockIdx.x * 256 + threadIdx.x;
    unsigned state = (MODE == 1 || MODE >= 3) ? (gid >> 6) * 2654435761u + 12345u : gid * 2654435761u + 12345u;
    unsigned acc = 0;
    for (int it = 0; it < iters; it++) {
        unsigned idx[8];
#pragma unroll
        for (int k = 0; k < 8; k++) { // 8 independent gathers in flight per lane: throughput, not latency
            state = state * 1664525u + 1013904223u;
            unsigned i = (state >> 8) & (unsigned)(n_rec - 1); // n_rec is a power of two: index generation must stay far cheaper than the gather
            if (MODE == 2) i = (i & ~3u) | (lane & 3u);
            if (MODE == 5) i = (i & ~63u) | lane;
            idx[k] = i * 16u;
        }
        if (MODE == 3 && (lane & 3u) != 0) continue;  // one active lane per quad
        if (MODE == 4 && lane >= 16) continue;         // the first 16 lanes (4 whole quads) active
#pragma unroll
        for (int k = 0; k < 8; k++) {
            if (WIDTH == 1) acc += __builtin_amdgcn_raw_buffer_load_b32(r, idx[k], 0, 0);
            if (WIDTH == 2) { u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(r, idx[k], 0, 0); acc += v.x ^ v.y; }
            if (WIDTH == 3) { u32x3 v = __builtin_amdgcn_raw_buffer_load_b96(r, idx[k], 0, 0); acc += v.x ^ v.y ^ v.z; }
            if (WIDTH == 4) { u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, idx[k], 0, 0); acc += v.x ^ v.y ^ v.z ^ v.w; }
        }
    }
    out[gid] = acc;
}

template <int WIDTH, int MODE>
static void run(const unsigned *table, int n_rec, unsigned *out, int cus, double ghz)
{
    const int blocks = cus * 8 * 4, iters = 400; // 8 blocks of 4 waves per CU = 8 waves per SIMD, 4 rounds
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    k_gather<WIDTH, MODE><<<blocks, 256>>>(table, n_rec, 10, out);
    hipEventRecord(a);
    k_gather<WIDTH, MODE><<<blocks, 256>>>(table, n_rec, iters, out);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    const double wave_instr = (double)blocks * 4 * iters * 8;
    const double cyc = ms * 1e-3 * ghz * 1e9 * cus / wave_instr;
    printf("width %d dword  mode %d (%s): %.3f ms  -> %.2f cycles per wave-gather per CU (at %.2f GHz)\n", WIDTH, MODE,
           MODE == 0 ? "random lanes" : MODE == 1 ? "wave-uniform" : MODE == 2 ? "4 lanes per 64-B line" : MODE == 3 ? "wave-uniform, 1 lane of 4 active" : MODE == 5 ? "wave-uniform base + lane (64 consecutive records)" : "wave-uniform, first 16 lanes active", ms, cyc, ghz);
}

int main()
{
    const int n_rec = 131072;
    std::vector<unsigned> h((size_t)n_rec * 4);
    for (size_t i = 0; i < h.size(); i++) h[i] = (unsigned)(i * 2654435761u);
    unsigned *table, *out;
    hipMalloc(&table, h.size() * 4);
    hipMemcpy(table, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    const double ghz = p.clockRate * 1e-6;
    hipMalloc(&out, (size_t)cus * 32 * 256 * 4);
    printf("%s: %d CUs, %.2f GHz nominal\n", p.name, cus, ghz);
    run<1, 0>(table, n_rec, out, cus, ghz); run<2, 0>(table, n_rec, out, cus, ghz); run<3, 0>(table, n_rec, out, cus, ghz); run<4, 0>(table, n_rec, out, cus, ghz);
    run<1, 1>(table, n_rec, out, cus, ghz); run<2, 1>(table, n_rec, out, cus, ghz); run<4, 1>(table, n_rec, out, cus, ghz);
    run<4, 3>(table, n_rec, out, cus, ghz); run<4, 4>(table, n_rec, out, cus, ghz); run<4, 5>(table, n_rec, out, cus, ghz); run<1, 5>(table, n_rec, out, cus, ghz);
    return 0;
}
