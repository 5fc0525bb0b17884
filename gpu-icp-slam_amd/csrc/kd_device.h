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
    const uint4 *hot;   // {x bits, y bits, left (30-bit two's complement) | axis<<30, right}
    const float *z;     // node z; for a PLANAR map (all z == 0): the true left child of z-level nodes, as int bits
    const int *parent;  // parent index, -1 at the root
    const float *w;     // occupancy weight (Node.value.w)
    int planar;         // every node has z == 0
};

// Traversal census (pfslam_score_census): what one launch of the score kernel really issued.  Filled by the CENSUS
// instantiation of the kernel only -- the timed instantiation carries none of this.
struct KdCensus {
    unsigned long long trips;        // wave-level trips of the descent loop = wave-level 16-byte gathers of node records
    unsigned long long lanes;        // active lanes summed over those trips = node visits
    unsigned long long tests;        // wave-level parent-hyperplane tests (each: one 4-byte + one 16-byte wave gather)
    unsigned long long test_lanes;   // lanes in them
    unsigned long long uniform;      // descent-loop trips in which every active lane stood on the same node
    unsigned long long prefix;       // ... and all 64 lanes were still on the common path from the root (first descent only)
    unsigned long long redesc;       // re-descents started (lanes)
    unsigned long long redesc_noop;  // ... that left the best node unchanged
};

// In a planar map a z-level node (axis 2) always sends the query right (0 < 0 is false).  Its hot record therefore
// stores the RIGHT child in both link fields -- the descent needs no axis test -- and the true left child, which only
// the parent-hyperplane re-descent can reach, moves to the (otherwise all-zero) z side array.
__host__ __device__ __forceinline__ int hot_left(uint32_t la) { return ((int)(la << 2)) >> 2; } // sign-extend 30 bits

__host__ __device__ __forceinline__ uint4 pack_hot(float x, float y, int axis, int left, int right, bool planar)
{
    if (planar && axis == 2) left = right;
    uint4 r;
#if defined(__HIP_DEVICE_COMPILE__)
    r.x = __float_as_uint(x);
    r.y = __float_as_uint(y);
#else
    union { float f; uint32_t u; } a, b;
    a.f = x; b.f = y;
    r.x = a.u; r.y = b.u;
#endif
    r.z = ((uint32_t)left & 0x3fffffffu) | ((uint32_t)axis << 30);
    r.w = (uint32_t)right;
    return r;
}

// Tree gathers go through buffer descriptors: a 32-bit byte offset (one shift) instead of a sign-extended 64-bit flat
// address per visit -- measured 2.40 vs 2.46 ms for the score kernel.  The descriptors are wave-uniform (SGPRs).
// idx << 4 must stay below the descriptor's 0x7ffffff0 bytes: pfslam_create bounds kd_capacity by 2^27 - 1 nodes.
#if defined(__HIP_DEVICE_COMPILE__)
typedef __amdgpu_buffer_rsrc_t kd_rsrc_t;
__device__ __forceinline__ kd_rsrc_t kd_rsrc(const void *base)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), /*stride*/ 0, /*bytes*/ 0x7ffffff0, 0x00020000);
}
__device__ __forceinline__ uint4 kd_load_hot(kd_rsrc_t r, int idx)
{
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, idx << 4, 0, 0);
    return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ int kd_load_i32(kd_rsrc_t r, int idx)
{
    return (int)__builtin_amdgcn_raw_buffer_load_b32(r, idx << 2, 0, 0);
}
__device__ __forceinline__ int kd_load_i32_bytes(kd_rsrc_t r, int byte_offset)
{
    return (int)__builtin_amdgcn_raw_buffer_load_b32(r, byte_offset, 0, 0);
}
// 16 bytes at byte offset `base` + `imm` (imm a compile-time constant: it goes into the instruction's offset field, so several
// requests off one address cost no address arithmetic)
__device__ __forceinline__ uint4 kd_load_hot_at(kd_rsrc_t r, int base, int imm)
{
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, base + imm, 0, 0);
    return make_uint4(v.x, v.y, v.z, v.w);
}
#else // hipcc's host pass only parses the device templates below; the descriptor type does not exist there
struct kd_rsrc_t { const void *base; };
__device__ kd_rsrc_t kd_rsrc(const void *base);
__device__ uint4 kd_load_hot(kd_rsrc_t r, int idx);
__device__ int kd_load_i32(kd_rsrc_t r, int idx);
__device__ int kd_load_i32_bytes(kd_rsrc_t r, int byte_offset);
__device__ uint4 kd_load_hot_at(kd_rsrc_t r, int base, int imm);
#endif

// Per-lane census counters (registers): a wave-level event is booked on its first active lane, every active lane books
// itself; census_flush adds a wave's totals to the global record once, at the end of the kernel.
struct KdCensusLocal { unsigned trips, lanes, tests, test_lanes, uniform, prefix, redesc, redesc_noop; };
__device__ __forceinline__ void census_add(unsigned &ev, unsigned &lanes)
{
    const unsigned long long ex = __builtin_amdgcn_read_exec();
    if ((int)(threadIdx.x & 63) == __ffsll((long long)ex) - 1) ev++;
    lanes++;
}
__device__ __forceinline__ void census_flush(const KdCensusLocal &c, KdCensus *out)
{
    unsigned long long v[8] = {c.trips, c.lanes, c.tests, c.test_lanes, c.uniform, c.prefix, c.redesc, c.redesc_noop};
    for (int k = 0; k < 8; k++)
        for (int off = 32; off >= 1; off >>= 1) v[k] += __shfl_xor(v[k], off, 64);
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&out->trips, v[0]);
        atomicAdd(&out->lanes, v[1]);
        atomicAdd(&out->tests, v[2]);
        atomicAdd(&out->test_lanes, v[3]);
        atomicAdd(&out->uniform, v[4]);
        atomicAdd(&out->prefix, v[5]);
        atomicAdd(&out->redesc, v[6]);
        atomicAdd(&out->redesc_noop, v[7]);
    }
}

// The traversal of kernel.cu:881-919 (== 931-969, 1147-1184, 1239-1276): greedy descent, then
// while the best node changed, one look at the best node's parent hyperplane and a re-descent of
// the sibling side.  It is NOT an exact nearest-neighbour search and is reproduced as is.
//
// Distances follow glm::distance: sqrt((dx*dx + dy*dy) + dz*dz), one rounding per operation.
// The per-visit `d < bestDist` is decided on the squared sums with a guard band (see the loop); the
// rooted best distance is only materialised once per descent, for the parent-hyperplane test.
//
// PLANAR: every node has z == 0 and the query has z == 0 (the SLAM map is 2-D): the z term is an
// exact +0, z-axis levels always branch right and have hyperplane distance 0.
// H1: the reference reads tree[-1] when the best node is the root; here the search stops.
#define PF_GUARD_K 0.999999523162841796875f /* 1 - 2^-21 */

// kd_resume continues a traversal from the state (sBest, bestIdx, head) of its FIRST descent: the score kernel enters here
// after it has worked off the part of the root path that the whole wave shares (see KdPlan below); head < 0 = the first
// descent is already complete.  kd_nearest_ref is the whole traversal: resume from the root with nothing seen.
//
// LEAF: also report where the FIRST descent fell off the tree, as slot = node * 2 + (1 = right link, 0 = left link).  The
// descent takes the left child exactly when query < node on the split axis, which is KDTree::InsertNode's rule
// (kdtree.cpp:69-105), so that slot is where the query point would be inserted (the map update uses it, k_test_new).
//
// BOUNDED (round 5): the traversal of the tree AS IT WAS when it had `bound` nodes.  Between two re-balances the tree is append-only
// (KDTree::InsertNode, kdtree.cpp:69-105): an insert writes nodes with indices >= the old size and hangs each on a link that was
// empty, so "ignore every child index >= bound" IS the old tree -- whatever the insert that runs beside this traversal has or has
// not written yet (a 16-byte record is read in one piece; a link is either still empty or names a node >= bound).  The frame's
// free-cell pass of kernUpdateMapKD (kernel.cu:1468-1483 runs before the insert of kernel.cu:1512-1517) uses it on a stream of its own.
template <bool PLANAR, bool CENSUS = false, bool LEAF = false, bool BOUNDED = false>
__device__ __forceinline__ int kd_resume(const KdView &t, float px, float py, float pz, float sBest, int bestIdx, int head,
                                         KdCensusLocal *census = nullptr, int *leaf_slot = nullptr, int bound = 0x7fffffff)
{
    int prevBest = -1;
    int leaf = -1;         // LEAF only
    bool leaf_open = true; // LEAF only: still in the first descent
    const kd_rsrc_t hot_rsrc = kd_rsrc(t.hot), parent_rsrc = kd_rsrc(t.parent);
    bool on_prefix = head == 0; // census only: trips on the common path of all 64 lanes from the root
    bool in_redesc = false;     // census only
    for (;;) {
        while (head >= 0) { // greedy descent
            if (CENSUS) {
                census_add(census->trips, census->lanes);
                const unsigned long long ex = __builtin_amdgcn_read_exec();
                const bool uni = __builtin_amdgcn_ballot_w64(head != __builtin_amdgcn_readfirstlane(head)) == 0ull;
                on_prefix = on_prefix && uni && ex == ~0ull;
                if ((int)(threadIdx.x & 63) == __ffsll((long long)ex) - 1) {
                    census->uniform += uni ? 1u : 0u;
                    census->prefix += on_prefix ? 1u : 0u;
                }
            }
            const uint4 nd = kd_load_hot(hot_rsrc, head);
            // (node - query) as a 2-vector: v_pk_add_f32 / v_pk_mul_f32, one rounding per component as in the scalar form
            typedef float f32x2 __attribute__((ext_vector_type(2)));
            const f32x2 nxy = {__uint_as_float(nd.x), __uint_as_float(nd.y)};
            const f32x2 pxy = {px, py};
            const f32x2 d = nxy - pxy;
            const f32x2 dd = d * d;
            float s = dd.x + dd.y;
            float dz = 0.0f;
            int zleft = 0; // generic kernel on a planar map (query with z != 0): the z array holds z-level left children
            if (!PLANAR) {
                const float zraw = t.z[head];
                const float nz = t.planar ? 0.0f : zraw;
                zleft = __float_as_int(zraw);
                dz = nz - pz;
                s = s + dz * dz;
            }
            // d < bestDist, decided without a sqrt: if s is below sBest by more than a guard band of 2^-21
            // (relative), the correctly rounded roots differ for certain (sqrt_rn has relative error <= 2^-24 and
            // halves relative gaps); only inside the band -- ~1e-6 of visits -- are the two roots compared.
            // (The relative bound needs normal floats: it assumes two DISTINCT map points are never both within
            // 1e-15 m of a query, i.e. squared distances below 1e-30 occur only for exact coincidence, s == 0.)
            const float sGuard = sBest * PF_GUARD_K; // recomputed per visit: one multiply instead of a multiply + a select
            bool take = s < sGuard;
            const bool inBand = (s < sBest) != take; // sGuard < sBest, so this is "sGuard <= s < sBest"
            if (__builtin_amdgcn_ballot_w64(inBand) != 0ull) { // wave-uniform, almost never taken
                float sb = sBest;
                asm volatile("" : "+v"(sb)); // pins the two sqrt inside the branch (the compiler would speculate them)
                take = take | (inBand && fsqrt(s) < fsqrt(sb));
            }
            sBest = take ? s : sBest;
            bestIdx = take ? head : bestIdx;
            // query < node on the split axis, read off the difference already computed: denormals are kept
            // (float_denorm_mode 3), so node - query > 0 exactly when query < node (NaN: both false)
            const uint32_t axis = nd.z >> 30;
            float da = axis == 0 ? d.x : d.y;
            if (!PLANAR) da = axis == 2 ? dz : da;
            const bool lt = da > 0.0f; // PLANAR z levels: both links hold the right child
            int left = hot_left(nd.z);
            if (!PLANAR) left = (t.planar && axis == 2) ? zleft : left;
            if (LEAF && leaf_open) leaf = head * 2 + ((lt && !(PLANAR && axis == 2)) ? 0 : 1);
            head = lt ? left : (int)nd.w;
            if (BOUNDED && head >= bound) head = -1;
        }
        if (LEAF && leaf_open) {
            leaf_open = false;
            *leaf_slot = leaf;
        }
        // `nodeFullyExplored` of the reference == "the last re-descent did not change the best node"
        on_prefix = false;
        if (CENSUS && in_redesc && bestIdx == prevBest) census->redesc_noop++;
        if (bestIdx == prevBest) break;
        if (CENSUS) census_add(census->tests, census->test_lanes);
        prevBest = bestIdx;
        const float bestDist = fsqrt(sBest);
        const int pi = kd_load_i32(parent_rsrc, bestIdx);
        if (pi < 0) break; // H1
        const uint4 nd = kd_load_hot(hot_rsrc, pi);
        const float nx = __uint_as_float(nd.x), ny = __uint_as_float(nd.y);
        const uint32_t axis = nd.z >> 30;
        float pa = axis == 0 ? px : py, na = axis == 0 ? nx : ny;
        float hd;
        bool lt;
        int left = hot_left(nd.z);
        if (PLANAR) {
            hd = axis < 2 ? fabsf(pa - na) : 0.0f;
            lt = (pa < na) & (axis < 2);
            if (axis == 2) left = __float_as_int(t.z[pi]); // true left of a planar z-level node
        } else {
            if (axis == 2) {
                const float zraw = t.z[pi];
                pa = pz;
                na = t.planar ? 0.0f : zraw;
                if (t.planar) left = __float_as_int(zraw);
            }
            hd = fabsf(pa - na);
            lt = pa < na;
        }
        if (!(hd < bestDist)) break;
        head = lt ? (int)nd.w : left; // the side the query is NOT on
        if (BOUNDED && head >= bound) head = -1;
        if (CENSUS) {
            in_redesc = true;
            census->redesc++;
        }
    }
    return bestIdx;
}

template <bool PLANAR, bool CENSUS = false>
__device__ __forceinline__ int kd_nearest_ref(const KdView &t, float px, float py, float pz, KdCensusLocal *census = nullptr)
{
    // bestDist starts as the distance to the root; visiting the root first reproduces that state
    return kd_resume<PLANAR, CENSUS>(t, px, py, pz, INFINITY, 0, 0, census);
}

// ------------------------------------------------------------------------------------------------------------------
// Shared-prefix plan of the score kernel (planar maps).
//
// The 64 lanes of a wave score the SAME beam from 64 neighbouring poses (Hilbert order), so their queries lie in a box W a
// few centimetres wide, and -- measured by pfslam_score_census -- 75-95 % of all descent-loop trips are spent on nodes that
// every lane of the wave visits, in the same order: the root path, until W first straddles a split plane.  The first
// descent of the reference traversal leaves best = the FIRST node of the path with the minimal (rounded) distance, nothing
// else of the path matters.  So, once per (wave, beam), ONE lane of a small planning kernel walks that common root path and
//   * stops where W straddles a split plane (`resume` = that node; -1 = the path was common down to the leaf), and
//   * keeps of the path only the nodes that can be the nearest one for SOME point of W: node i is dropped when its smallest
//     possible squared distance to W exceeds the largest possible squared distance of another path node, with a 1e-5 relative
//     margin on either side (>> the 2^-21 guard band, >> float rounding) -- a dropped node is then strictly farther than
//     that other node for every lane, so it can be neither the minimum nor tie with it.  Order is kept.
// The score kernel then evaluates, per lane, only the few surviving candidates (from scalar registers: no gather, no child
// selection, no loop divergence) with exactly the per-visit arithmetic of kd_resume, and continues per lane from `resume`.
// Results are bit-identical to the plain traversal by construction; tests compare both against the oracle.
// ------------------------------------------------------------------------------------------------------------------
#define PF_PLAN_CAND 7 /* candidates per row: 128-byte rows */
struct KdPlanRow {
    int n_cand;     // candidates below (path order)
    int resume;     // node at which the per-lane traversal continues, -1 = first descent complete
    int path_len;   // nodes of the common root path (statistics)
    float range;    // scan[beam]: rides along so that the score kernel needs one scalar stream only
    float4 cand[PF_PLAN_CAND]; // {x, y, node index bits, lower bound (planning scratch)}
};
static_assert(sizeof(KdPlanRow) == 16 + 16 * PF_PLAN_CAND, "plan row layout");

struct KdGroupBox { float xlo, xhi, ylo, yhi, tlo, thi; int count, pad; }; // pose bounding box of a wave's 64 particles

// LIDAR_ANGLE(i) (kernel.cu:42) + CleanLidarScan (kernel.cu:182-187).  cos / sin of rot = angle + theta by the angle-addition
// specification of pf_math.h (sincos_sum_spec).
__device__ __forceinline__ float lidar_angle(int n)
{
    const float PI_F = 3.1415926535897932384626422832795028841971f; // utilities.h:12
    return fdiv((-135.0f + (float)n * .25f) * PI_F, 180.0f);
}
// the hot loops: the beam's angle and parts from a per-beam table (wave-uniform scalars), the heading's parts once per particle
template <int GUARD = 1>
__device__ __forceinline__ void clean_lidar_scan_parts(float angle, const AngleParts &A, float range, float theta, const AngleParts &T,
                                                       float &x, float &y)
{
    const float rot = angle + theta;
    float s, c;
    sincos_sum_spec<GUARD>(A, T, rot, s, c);
    x = range * c;
    y = range * s;
}
__device__ __forceinline__ void clean_lidar_scan(int n, float range, float theta, float &x, float &y, int trig = 0)
{
    const float angle = lidar_angle(n);
    if (trig) clean_lidar_scan_parts<PF_TRIG_DEVLIB>(angle, AngleParts{0.0, 0.0, 0.0}, range, theta, AngleParts{0.0, 0.0, 0.0}, x, y); // (pfslam_set_trig)
    else clean_lidar_scan_parts(angle, angle_parts(angle), range, theta, angle_parts(theta), x, y);
}
// per-beam table entry: {cos, sin, angle as double, angle as float (low half of the 4th double's slot)}
struct BeamParts { double c, s, a; float angle, pad; };
static_assert(sizeof(BeamParts) == 32, "beam table entry");

} // namespace pf
