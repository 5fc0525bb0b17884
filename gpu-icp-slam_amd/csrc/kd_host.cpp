// kd_host.cpp -- host-side map structure: the counterpart of the reference's
// KDTree::Create / InsertList / InsertNode / Balance (src/kdtree.cpp:25-105).
//
// The tree is the pre-order array the kernels traverse: node i's left child is i+1, its right
// child i+mid+1, -1 = none; axis cycles x,y,z from the root.  The topology the reference
// produces is whatever libstdc++'s (unstable) std::sort does with the tied keys of grid-snapped
// points, so the build sorts the same sequences with the same comparison -- but on sub-ranges of
// one scratch buffer rather than on a fresh std::vector copy per recursion level.
#include "../../include/pfslam.h"

#include <algorithm>
#include <atomic>
#include <thread>
#include <vector>

namespace {
struct Pt { float x, y, z, w; };
inline bool byX(const Pt &a, const Pt &b) { return a.x < b.x; }
inline bool byY(const Pt &a, const Pt &b) { return a.y < b.y; }
inline bool byZ(const Pt &a, const Pt &b) { return a.z < b.z; }

// std::sort, on several threads, with std::sort's result.
//
// The permutation libstdc++'s std::sort leaves among tied keys IS the map's topology (kdtree.cpp:27,45-50 sort grid-snapped
// coordinates: thousands of ties), so the sort cannot be replaced -- but it can be run in parallel: std::sort is
// __introsort_loop (quicksort: median-of-three pivot, unguarded partition, the right part by recursion and the left part by
// iteration, heapsort below the depth limit, ranges of <= 16 left alone) followed by __final_insertion_sort over the whole range.
// What the loop does to a sub-range depends on that sub-range and its depth budget only, so the recursive call may as well run
// on another thread; every partition, every swap and the final insertion pass are libstdc++'s own code, called with the
// arguments std::sort would call them with.  The re-balance (KDTree::Balance every 100 frames, a full re-build on the host while
// the GPU waits) is bounded by the first sorts of the build, which see the whole map on one thread otherwise.
std::atomic<int> g_sort_threads{0};
int sort_thread_budget()
{
    static const int n = std::max(1u, std::min(std::thread::hardware_concurrency(), 64u));
    return n;
}
template <typename It, typename Cmp>
void par_introsort_loop(It first, It last, long depth_limit, Cmp comp)
{
    std::vector<std::thread> kids;
    while (last - first > 16) { // _S_threshold
        if (depth_limit == 0) {
            std::__partial_sort(first, last, last, comp); // heapsort of the rest
            break;
        }
        --depth_limit;
        It cut = std::__unguarded_partition_pivot(first, last, comp);
        if (last - cut > 16384 && g_sort_threads.fetch_add(1) < sort_thread_budget()) {
            kids.emplace_back([=] {
                par_introsort_loop(cut, last, depth_limit, comp);
                g_sort_threads.fetch_sub(1);
            });
        } else {
            if (last - cut > 16384) g_sort_threads.fetch_sub(1); // budget exhausted: undo the reservation
            std::__introsort_loop(cut, last, depth_limit, comp);
        }
        last = cut;
    }
    for (auto &t : kids) t.join();
}
template <typename It, typename Cmp>
void exact_sort(It first, It last, Cmp cmp)
{
    if (first == last) return;
    if (last - first <= 32768) { // small: the library call itself
        std::sort(first, last, cmp);
        return;
    }
    auto comp = __gnu_cxx::__ops::__iter_comp_iter(cmp);
    par_introsort_loop(first, last, (long)std::__lg(last - first) * 2, comp);
    std::__final_insertion_sort(first, last, comp);
}

// `fork` > 0: the two sub-ranges are disjoint slices of `buf` and disjoint slices of `out` (pre-order layout), so
// they are built on two threads; every sort still sees exactly the sequence the sequential build would give it,
// hence the same (unstable-sort dependent) topology as the reference.
void build_range(std::vector<Pt> &buf, int lo, int hi, pfslam_node *out, int idx, int parent, int fork)
{
    const int axis = parent < 0 ? 0 : (out[parent].axis + 1) % 3;
    auto first = buf.begin() + lo, last = buf.begin() + hi;
    switch (axis) {
    case 0: exact_sort(first, last, byX); break;
    case 1: exact_sort(first, last, byY); break;
    default: exact_sort(first, last, byZ); break;
    }
    const int count = hi - lo, mid = count / 2;
    const Pt &m = buf[lo + mid];
    out[idx] = pfslam_node{axis, -1, -1, parent, m.x, m.y, m.z, m.w};
    const bool has_left = mid > 0, has_right = mid < count - 1;
    if (has_left) out[idx].left = idx + 1;
    if (has_right) out[idx].right = idx + mid + 1;
    if (fork > 0 && has_left && has_right && count > 8192) {
        std::thread t([&] { build_range(buf, lo, lo + mid, out, idx + 1, idx, fork - 1); });
        build_range(buf, lo + mid + 1, hi, out, idx + mid + 1, idx, fork - 1);
        t.join();
        return;
    }
    if (has_left) build_range(buf, lo, lo + mid, out, idx + 1, idx, 0);
    if (has_right) build_range(buf, lo + mid + 1, hi, out, idx + mid + 1, idx, 0);
}
} // namespace

extern "C" int pfslam_kd_create(const float *pts_xyzw, int n, pfslam_node *out)
{
    if (n < 0 || (n > 0 && (!pts_xyzw || !out))) return 1;
    if (n == 0) return 0;
    std::vector<Pt> buf(n);
    for (int i = 0; i < n; i++) buf[i] = Pt{pts_xyzw[4 * i], pts_xyzw[4 * i + 1], pts_xyzw[4 * i + 2], pts_xyzw[4 * i + 3]};
    exact_sort(buf.begin(), buf.end(), byX); // KDTree::Create pre-sorts on x before the recursive sort
    const unsigned hw = std::thread::hardware_concurrency();
    // up to 16 concurrent sub-builds, 32 for a big map (500 k points on the 16 cores of the GPU box: 29 -> 25 ms; no gain at 100 k)
    build_range(buf, 0, n, out, 0, -1, hw >= 16 ? (n >= 200000 ? 5 : 4) : hw >= 4 ? 2 : hw >= 2 ? 1 : 0);
    return 0;
}

// KDTree::InsertList (kdtree.cpp:46-67): the sub-tree of `n` points in pre-order at list[idx ...], hanging below `parent`
// (split axis = parent's + 1; -1 = root).  Like the reference it does not touch the parent's child links.
extern "C" int pfslam_kd_insert_list(const float *pts_xyzw, int n, pfslam_node *list, int idx, int parent)
{
    if (n <= 0 || !pts_xyzw || !list || idx < 0 || parent < -1) return 1;
    std::vector<Pt> buf(n);
    for (int i = 0; i < n; i++) buf[i] = Pt{pts_xyzw[4 * i], pts_xyzw[4 * i + 1], pts_xyzw[4 * i + 2], pts_xyzw[4 * i + 3]};
    build_range(buf, 0, n, list, idx, parent, 0);
    return 0;
}

extern "C" int pfslam_kd_insert_node(const float p[4], pfslam_node *list, int list_size)
{
    if (!p || !list || list_size <= 0) return 1;
    int cur = 0, at = 0, axis = 0;
    bool goLeft = false;
    while (cur != -1) {
        at = cur;
        axis = list[at].parent == -1 ? 0 : (list[list[at].parent].axis + 1) % 3;
        const float key = axis == 0 ? list[at].x : axis == 1 ? list[at].y : list[at].z;
        goLeft = p[axis] < key;
        cur = goLeft ? list[at].left : list[at].right;
    }
    (goLeft ? list[at].left : list[at].right) = list_size;
    list[list_size] = pfslam_node{(axis + 1) % 3, -1, -1, at, p[0], p[1], p[2], p[3]};
    return 0;
}

extern "C" int pfslam_kd_balance(pfslam_node *list, int n)
{
    if (n < 0 || (n > 0 && !list)) return 1;
    std::vector<float> pts(4 * (size_t)n);
    for (int i = 0; i < n; i++) {
        pts[4 * i] = list[i].x; pts[4 * i + 1] = list[i].y; pts[4 * i + 2] = list[i].z; pts[4 * i + 3] = list[i].w;
    }
    return pfslam_kd_create(pts.data(), n, list);
}
