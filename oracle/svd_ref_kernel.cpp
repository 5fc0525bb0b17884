/*
 * svd_ref_kernel.cpp -- runs the REFERENCE's own src/svd3.h on the GPU (test infrastructure).
 *
 * svd3.h is host code that calls CUDA's rsqrt(); as host code it cannot be built here (no CUDA headers, and no
 * stand-ins are written).  Compiled for the DEVICE only, with clang's `force_cuda_host_device` pragma around the
 * unmodified header, every function in it becomes device-callable and rsqrt() resolves to the ROCm device library's
 * own rsqrt (double, as the float argument promotes) -- no reference source is copied or altered.  The Makefile builds
 * this into oracle/_ref/svd_ref.hsaco (only when /root/reference is present); hsaco_launcher.cpp loads it.
 */
#include <hip/hip_runtime.h>
#include <math.h>
#include "utilities.h" /* reference header: EPSILON (utilities.h:15) */
#pragma clang force_cuda_host_device begin
#include "svd3.h"      /* reference header, unmodified */
#pragma clang force_cuda_host_device end

extern "C" __global__ void ref_svd3_kernel(const float *a, float *o, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float *A = a + 9 * i;
    float *u = o + 27 * i, *s = u + 9, *v = u + 18;
    svd(A[0], A[1], A[2], A[3], A[4], A[5], A[6], A[7], A[8],
        u[0], u[1], u[2], u[3], u[4], u[5], u[6], u[7], u[8],
        s[0], s[1], s[2], s[3], s[4], s[5], s[6], s[7], s[8],
        v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7], v[8]);
}
