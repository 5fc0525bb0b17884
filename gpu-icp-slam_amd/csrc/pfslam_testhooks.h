/*
 * pfslam_testhooks.h -- NOT part of the installed interface (include/pfslam.h).  Entry points that exist only so that
 * tests/ can check device-side arithmetic against the oracle; exported from libpfslam_hip.so, declared here.
 */
#ifndef PFSLAM_TESTHOOKS_H
#define PFSLAM_TESTHOOKS_H
#include "../../include/pfslam.h"
#ifdef __cplusplus
extern "C" {
#endif
/* evaluate the bit-reproducible math specification (pf_math.h) on the device.
 * which: 0 sincos (out: n x {sin, cos}), 1 erfcinv, 2 asin, 3 rsqrt, 4 sqrt_rn, 5 x / 0.025f,
 *        6 sub-cell index of x on the 0.025 m lattice (out: n x {sub_index, sub_index_fast or INT_MIN when not safe}, int bits) */
int pfslam_debug_math(pfslam_handle *h, int which, const float *in_host, int n, float *out_host);
#ifdef __cplusplus
}
#endif
#endif
