/*
 * kdtree_oracle.cpp -- CPU oracle (TEST INFRASTRUCTURE) for KDTree::Create /
 * InsertList / Balance (reference src/kdtree.cpp:25-67).
 *
 * This one piece of the oracle is C++ rather than plain C on purpose: the
 * reference's tree topology is *defined* by what libstdc++'s unstable
 * std::sort does with the massive key ties of grid-snapped points
 * (kdtree.cpp:27, 45-50), so the restatement has to call the same std::sort
 * with the same comparison on the same sequences.  It sorts sub-ranges of one
 * buffer in place instead of copying a vector per recursion level (the copy is
 * the same sequence, so std::sort makes the same decisions).  Pinned against
 * the reference's own kdtree.cpp via oracle/_ref/libkdtree_ref.so
 * (tests/test_oracle_pinning.py).
 */
#include "pfslam_oracle.h"

#include <algorithm>
#include <vector>

namespace {
struct P4 { float x, y, z, w; };
bool lessX(const P4 &a, const P4 &b) { return a.x < b.x; }
bool lessY(const P4 &a, const P4 &b) { return a.y < b.y; }
bool lessZ(const P4 &a, const P4 &b) { return a.z < b.z; }

void insert_list(std::vector<P4> &pts, int lo, int hi, orc_node *list, int idx, int parent)
{
    int axis = (parent == -1) ? 0 : (list[parent].axis + 1) % 3;
    if (axis == 0) std::sort(pts.begin() + lo, pts.begin() + hi, lessX);
    if (axis == 1) std::sort(pts.begin() + lo, pts.begin() + hi, lessY);
    if (axis == 2) std::sort(pts.begin() + lo, pts.begin() + hi, lessZ);
    int n = hi - lo;
    int mid = n / 2;
    const P4 &m = pts[lo + mid];
    orc_node nd;
    nd.axis = axis; nd.left = -1; nd.right = -1; nd.parent = parent;
    nd.x = m.x; nd.y = m.y; nd.z = m.z; nd.w = m.w;
    list[idx] = nd;
    if (mid > 0) {
        list[idx].left = idx + 1;
        insert_list(pts, lo, lo + mid, list, idx + 1, idx);
    }
    if (mid < n - 1) {
        list[idx].right = idx + mid + 1;
        insert_list(pts, lo + mid + 1, hi, list, idx + mid + 1, idx);
    }
}
} // namespace

extern "C" void orc_kd_create(const float *pts_xyzw, int n, orc_node *list)
{
    if (n <= 0) return;
    std::vector<P4> pts(n);
    for (int i = 0; i < n; i++) pts[i] = P4{pts_xyzw[4 * i], pts_xyzw[4 * i + 1], pts_xyzw[4 * i + 2], pts_xyzw[4 * i + 3]};
    std::sort(pts.begin(), pts.end(), lessX); /* KDTree::Create sorts once before InsertList sorts again */
    insert_list(pts, 0, n, list, 0, -1);
}

extern "C" void orc_kd_balance(orc_node *list, int n)
{
    std::vector<float> v(4 * (size_t)n);
    for (int i = 0; i < n; i++) {
        v[4 * i] = list[i].x; v[4 * i + 1] = list[i].y; v[4 * i + 2] = list[i].z; v[4 * i + 3] = list[i].w;
    }
    orc_kd_create(v.data(), n, list);
}
