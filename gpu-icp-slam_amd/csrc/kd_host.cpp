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
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <system_error>
#include <thread>
#include <vector>
#if defined(__linux__)
#include <sched.h>
#endif

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
// Threads this process may keep busy: the cores it can actually run on (scheduler affinity and the cgroup CPU quota -- 16 of the 256
// logical CPUs on the GPU box), shared with the other ranks of a multi-GPU job on the same node (LOCAL_WORLD_SIZE, as torchrun and
// pfslam_mgpu set it; PFSLAM_SORT_THREADS overrides).  Eight ranks re-balancing at once, each with 64 sort threads, is what this avoids.
int usable_cores()
{
    int n = (int)std::thread::hardware_concurrency();
#if defined(__linux__)
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof(set), &set) == 0) n = std::min(n > 0 ? n : 1 << 30, CPU_COUNT(&set));
    if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
        char quota[32];
        long period = 0;
        if (fscanf(f, "%31s %ld", quota, &period) == 2 && strcmp(quota, "max") != 0 && period > 0)
            n = std::min(n, std::max(1, (int)((atol(quota) + period / 2) / period)));
        fclose(f);
    }
#endif
    return std::max(n, 1);
}
int sort_thread_budget()
{
    static const int n = [] {
        if (const char *e = getenv("PFSLAM_SORT_THREADS")) return std::max(1, atoi(e));
        int ranks = 1;
        if (const char *e = getenv("LOCAL_WORLD_SIZE")) ranks = std::max(1, atoi(e));
        return std::max(1, std::min(usable_cores() / ranks, 64));
    }();
    return n;
}
std::atomic<int> g_sort_threads{0};

#if defined(__GLIBCXX__)
// libstdc++'s std::sort, taken apart: the helpers below are its INTERNALS, and the permutation they leave among tied keys is the
// map's topology.  Guarded three ways: compiled only against libstdc++ (any other library: plain std::sort), a static check of
// the insertion-sort threshold the code relies on, and a start-up self-check of the parallel form against std::sort itself on
// a heavily tied array (parallel_sort_ok) -- a library that changes the pivot rule or the helpers' contracts turns the parallel
// path off instead of silently building a different tree.
static_assert((int)std::_S_threshold == 16, "exact_sort restates libstdc++'s introsort: its insertion-sort threshold has changed");
template <typename It, typename Cmp>
void par_introsort_loop(It first, It last, long depth_limit, Cmp comp)
{
    std::vector<std::thread> kids;
    while (last - first > (long)std::_S_threshold) {
        if (depth_limit == 0) {
            std::__partial_sort(first, last, last, comp); // heapsort of the rest
            break;
        }
        --depth_limit;
        It cut = std::__unguarded_partition_pivot(first, last, comp);
        bool forked = false;
        if (last - cut > 16384 && g_sort_threads.fetch_add(1) < sort_thread_budget()) {
            try {
                kids.emplace_back([=] {
                    par_introsort_loop(cut, last, depth_limit, comp);
                    g_sort_threads.fetch_sub(1);
                });
                forked = true;
            } catch (const std::system_error &) { // no thread to be had: this one does the work
                g_sort_threads.fetch_sub(1);
            }
        } else if (last - cut > 16384) g_sort_threads.fetch_sub(1); // budget exhausted: undo the reservation
        if (!forked) std::__introsort_loop(cut, last, depth_limit, comp);
        last = cut;
    }
    for (auto &t : kids) t.join();
}
template <typename It, typename Cmp>
void par_sort(It first, It last, Cmp cmp)
{
    auto comp = __gnu_cxx::__ops::__iter_comp_iter(cmp);
    par_introsort_loop(first, last, (long)std::__lg(last - first) * 2, comp);
    std::__final_insertion_sort(first, last, comp);
}
// once per process: 40 000 points on 7 distinct x values, sorted both ways -- the same bytes, or the parallel form is not used
bool parallel_sort_ok()
{
    static const bool ok = [] {
        std::vector<Pt> a(40000);
        uint32_t st = 12345u;
        for (size_t i = 0; i < a.size(); i++) {
            st = st * 1664525u + 1013904223u;
            a[i] = Pt{(float)((st >> 16) % 7u) * 0.025f, (float)i, 0.0f, (float)(st >> 8)};
        }
        std::vector<Pt> b = a;
        std::sort(a.begin(), a.end(), byX);
        par_sort(b.begin(), b.end(), byX);
        return memcmp(a.data(), b.data(), a.size() * sizeof(Pt)) == 0;
    }();
    return ok;
}
#endif
template <typename It, typename Cmp>
void exact_sort(It first, It last, Cmp cmp)
{
    if (first == last) return;
#if defined(__GLIBCXX__)
    if (last - first > 32768 && sort_thread_budget() > 1 && parallel_sort_ok()) {
        par_sort(first, last, cmp);
        return;
    }
#endif
    std::sort(first, last, cmp); // small, one thread, or a standard library this file does not know: the library call itself
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
        std::thread t;
        try {
            t = std::thread([&] { build_range(buf, lo, lo + mid, out, idx + 1, idx, fork - 1); });
        } catch (const std::system_error &) { // no thread to be had (an extern "C" entry point must not terminate): in line
            build_range(buf, lo, lo + mid, out, idx + 1, idx, 0);
        }
        build_range(buf, lo + mid + 1, hi, out, idx + mid + 1, idx, fork - 1);
        if (t.joinable()) t.join();
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
    const unsigned hw = (unsigned)sort_thread_budget(); // usable cores, shared with the node's other ranks
    // up to 16 concurrent sub-builds, 32 for a big map (500 k points on the 16 cores of the GPU box: 29 -> 25 ms; no gain at 100 k)
    build_range(buf, 0, n, out, 0, -1, hw >= 16 ? (n >= 200000 ? 5 : 4) : hw >= 8 ? 3 : hw >= 4 ? 2 : hw >= 2 ? 1 : 0);
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

// 1 = the parallel form of the host sort is in use (libstdc++, more than one thread, self-check passed); 0 = plain std::sort
extern "C" int pfslam_kd_parallel_sort(void)
{
#if defined(__GLIBCXX__)
    return sort_thread_budget() > 1 && parallel_sort_ok() ? 1 : 0;
#else
    return 0;
#endif
}
// the threads a host-side build may use: usable cores (affinity, cgroup quota) / ranks on this node
extern "C" int pfslam_kd_sort_threads(void) { return sort_thread_budget(); }

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
