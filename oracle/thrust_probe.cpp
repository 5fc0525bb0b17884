/*
 * thrust_probe.cpp -- pins the oracle's RNG restatement against the image's
 * rocThrust (thrust is the reference's un-vendored dependency: CUDA 7.5 toolkit,
 * include_list_release.txt:11).  Compiled for the HOST only
 * (-DTHRUST_DEVICE_SYSTEM=THRUST_DEVICE_SYSTEM_CPP) with hipcc so that thrust
 * selects the same erfcinv-based normal_distribution variant nvcc does.
 * Test infrastructure only.
 */
#include <thrust/random.h>

extern "C" void tp_minstd(unsigned seed, int n, unsigned *out)
{
    thrust::default_random_engine e(seed);
    for (int i = 0; i < n; i++) out[i] = e();
}
extern "C" void tp_uniform(unsigned seed, float a, float b, int n, float *out)
{
    thrust::default_random_engine e(seed);
    thrust::random::uniform_real_distribution<float> d(a, b);
    for (int i = 0; i < n; i++) out[i] = d(e);
}
/* three successive normals with three distributions from one engine, as ParticleAddNoise does */
extern "C" void tp_normal3(unsigned seed, float sx, float sy, float st, float *out)
{
    thrust::default_random_engine e(seed);
    thrust::random::normal_distribution<float> dx(0.0f, sx), dy(0.0f, sy), dt(0.0f, st);
    out[0] = dx(e);
    out[1] = dy(e);
    out[2] = dt(e);
}
