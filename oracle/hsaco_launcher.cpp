/*
 * hsaco_launcher.cpp -- loads oracle/_ref/svd_ref.hsaco (the reference's svd3.h compiled for gfx950) and runs its
 * kernel on n row-major 3x3 matrices.  Test infrastructure; contains no reference source.
 */
#include <hip/hip_runtime.h>

extern "C" int ref_svd3_gpu(const char *hsaco_path, const float *a_host, float *out_host, int n)
{
    hipModule_t mod;
    hipFunction_t fn;
    if (hipModuleLoad(&mod, hsaco_path) != hipSuccess) return 1;
    if (hipModuleGetFunction(&fn, mod, "ref_svd3_kernel") != hipSuccess) return 2;
    float *da = nullptr, *dout = nullptr;
    if (hipMalloc((void **)&da, (size_t)n * 36) != hipSuccess || hipMalloc((void **)&dout, (size_t)n * 108) != hipSuccess) return 3;
    if (hipMemcpy(da, a_host, (size_t)n * 36, hipMemcpyHostToDevice) != hipSuccess) return 4;
    void *args[] = {&da, &dout, &n};
    if (hipModuleLaunchKernel(fn, (n + 63) / 64, 1, 1, 64, 1, 1, 0, nullptr, args, nullptr) != hipSuccess) return 5;
    if (hipMemcpy(out_host, dout, (size_t)n * 108, hipMemcpyDeviceToHost) != hipSuccess) return 6;
    (void)hipFree(da);
    (void)hipFree(dout);
    (void)hipModuleUnload(mod);
    return 0;
}
