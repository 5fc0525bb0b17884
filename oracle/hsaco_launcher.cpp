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

/*
 * Generic part (round 5): load a code object once, launch any of its kernels with a packed argument buffer, move bytes.
 * Used by tests/test_gpu_ref_kernels.py for oracle/_ref/kernel_ref.hsaco (the reference's kernel.cu, device code only).
 * All entry points return 0 or a positive step number / the hipError_t + 100.
 */
#include <map>
#include <string>
static std::map<std::string, hipModule_t> g_mods;

extern "C" int ref_dev_alloc(void **p, size_t bytes) { return hipMalloc(p, bytes ? bytes : 1) == hipSuccess ? 0 : 1; }
extern "C" int ref_dev_free(void *p) { return hipFree(p) == hipSuccess ? 0 : 1; }
extern "C" int ref_dev_memset(void *p, int v, size_t bytes) { return hipMemset(p, v, bytes) == hipSuccess ? 0 : 1; }
extern "C" int ref_h2d(void *dst, const void *src, size_t bytes) { return hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice) == hipSuccess ? 0 : 1; }
extern "C" int ref_d2h(void *dst, const void *src, size_t bytes) { return hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost) == hipSuccess ? 0 : 1; }

extern "C" int ref_launch(const char *hsaco_path, const char *kernel, unsigned grid_x, unsigned block_x, void *argbuf, size_t argsize)
{
    hipModule_t mod;
    auto it = g_mods.find(hsaco_path);
    if (it == g_mods.end()) {
        if (hipModuleLoad(&mod, hsaco_path) != hipSuccess) return 1;
        g_mods[hsaco_path] = mod;
    } else mod = it->second;
    hipFunction_t fn;
    if (hipModuleGetFunction(&fn, mod, kernel) != hipSuccess) return 2;
    void *extra[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, argbuf, HIP_LAUNCH_PARAM_BUFFER_SIZE, &argsize, HIP_LAUNCH_PARAM_END};
    hipError_t e = hipModuleLaunchKernel(fn, grid_x, 1, 1, block_x, 1, 1, 0, nullptr, nullptr, extra);
    if (e != hipSuccess) return 100 + (int)e;
    e = hipDeviceSynchronize();
    return e == hipSuccess ? 0 : 100 + (int)e;
}
