// kd_device.h -- device view of the point-cloud map and the reference's KD "nearest neighbour".
//
// HBM layout (DESIGN.md "Data layout"): the 32-byte KDTree::Node (kdtree.hpp:16-27) is split
// into a 16-byte hot record {x, y, left|axis, right} that a traversal step reads with one
// dwordx4 load, plus cold side arrays parent[], z[] and the mutable weight w[] (the only field
// the map update writes).  Indices are the reference's array indices, so "best index" results
// are directly comparable with findCorrespondenceIndexKD (kernel.cu:924-972).
#pragma once
#include "pf_math.h"

namespace pf {

struct KdView {
    const uint4 *hot;   // {x bits, y bits, (left+1) | axis<<30, right}
    const float *z;     // node z (all zero for a planar map)
    const int *parent;  // parent index, -1 at the root
    const float *w;     // occupancy weight (Node.value.w)
};

__host__ __device__ __forceinline__ uint4 pack_hot(float x, float y, int axis, int left, int right)
{
    uint4 r;
#if defined(__HIP_DEVICE_COMPILE__)
    r.x = __float_as_uint(x);
    r.y = __float_as_uint(y);
#else
    union { float f; uint32_t u; } a, b;
    a.f = x; b.f = y;
    r.x = a.u; r.y = b.u;
#endif
    r.z = (uint32_t)(left + 1) | ((uint32_t)axis << 30);
    r.w = (uint32_t)right;
    return r;
}

// The traversal of kernel.cu:881-919 (== 931-969, 1147-1184, 1239-1276): greedy descent, then
// while the best node changed, one look at the best node's parent hyperplane and a re-descent of
// the sibling side.  It is NOT an exact nearest-neighbour search and is reproduced as is.
//
// Distances follow glm::distance: sqrt((dx*dx + dy*dy) + dz*dz), one rounding per operation.
// sqrt is monotone, so `d < bestDist` can only hold when the squared sum is below the best
// squared sum; the correctly rounded sqrt is evaluated only then (a handful of times per query
// instead of once per visited node) and the comparison itself is still made on the rooted values.
//
// PLANAR: every node has z == 0 and the query has z == 0 (the SLAM map is 2-D): the z term is an
// exact +0, z-axis levels always branch right and have hyperplane distance 0.
// H1: the reference reads tree[-1] when the best node is the root; here the search stops.
template <bool PLANAR>
__device__ __forceinline__ int kd_nearest_ref(const KdView &t, float px, float py, float pz)
{
    uint4 nd = t.hot[0];
    float nx = __uint_as_float(nd.x), ny = __uint_as_float(nd.y), nz = 0.0f;
    float dx = nx - px, dy = ny - py;
    float s = dx * dx + dy * dy;
    if (!PLANAR) {
        nz = t.z[0];
        float dz = nz - pz;
        s = s + dz * dz;
    }
    float sBest = s;
    float bestDist = fsqrt(s);
    int bestIdx = 0;
    bool explored = false;
    int head = 0;
    for (;;) {
        while (head >= 0) {
            nd = t.hot[head];
            nx = __uint_as_float(nd.x);
            ny = __uint_as_float(nd.y);
            dx = nx - px;
            dy = ny - py;
            s = dx * dx + dy * dy;
            if (!PLANAR) {
                nz = t.z[head];
                float dz = nz - pz;
                s = s + dz * dz;
            }
            if (s < sBest) {
                float d = fsqrt(s);
                if (d < bestDist) {
                    bestDist = d;
                    sBest = s;
                    bestIdx = head;
                    explored = false;
                }
            }
            const uint32_t axis = nd.z >> 30;
            const bool br = axis == 0 ? (px < nx) : axis == 1 ? (py < ny) : (PLANAR ? false : (pz < nz));
            head = br ? (int)(nd.z & 0x3fffffffu) - 1 : (int)nd.w;
        }
        if (explored) break;
        const int pi = t.parent[bestIdx];
        if (pi < 0) break;
        nd = t.hot[pi];
        nx = __uint_as_float(nd.x);
        ny = __uint_as_float(nd.y);
        const uint32_t axis = nd.z >> 30;
        float hd;
        bool br;
        if (axis == 0) {
            br = px < nx;
            hd = fabsf(px - nx);
        } else if (axis == 1) {
            br = py < ny;
            hd = fabsf(py - ny);
        } else if (PLANAR) {
            br = false;
            hd = 0.0f;
        } else {
            nz = t.z[pi];
            br = pz < nz;
            hd = fabsf(pz - nz);
        }
        if (hd < bestDist) {
            head = !br ? (int)(nd.z & 0x3fffffffu) - 1 : (int)nd.w;
            explored = true;
        } else {
            break;
        }
    }
    return bestIdx;
}

// LIDAR_ANGLE(i) (kernel.cu:42) + CleanLidarScan (kernel.cu:182-187)
__device__ __forceinline__ void clean_lidar_scan(int n, float range, float theta, float &x, float &y)
{
    const float PI_F = 3.1415926535897932384626422832795028841971f; // utilities.h:12
    float rot = fdiv((-135.0f + (float)n * .25f) * PI_F, 180.0f) + theta;
    float s, c;
    sincosf_spec(rot, s, c);
    x = range * c;
    y = range * s;
}

} // namespace pf
