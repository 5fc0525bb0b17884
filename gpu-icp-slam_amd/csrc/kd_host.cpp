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
#include <memory>
#include <mutex>
#include <numeric>
#include <system_error>
#include <thread>
#include <vector>
#if defined(__linux__)
#include <sched.h>
#endif

namespace {
struct Pt { float x, y, z, w; };
inline bool byX(const Pt &a, const Pt &b) { return a.x < b.x; }

// std::sort's RESULT, faster than std::sort computes it.
//
// The permutation libstdc++'s std::sort leaves among tied keys IS the map's topology (kdtree.cpp:27,45-50 sort grid-snapped
// coordinates: thousands of ties), so the sort cannot be replaced by another one -- but what it does can be restated.  std::sort is
// __introsort_loop (quicksort: median-of-three pivot, unguarded Hoare partition, the right part by recursion and the left part by
// iteration, heapsort below the depth limit, ranges of <= 16 left alone) followed by __final_insertion_sort over the whole range.
//   * What the loop does to a sub-range depends on that sub-range and its depth budget only: the recursive call may run on another
//     thread (par_introsort_loop).
//   * The partition is a fixed pairing of elements (block_partition): the same swaps without the data-dependent branches that std::sort
//     spends most of its time on with grid-snapped keys.
//   * Insertion sort is a STABLE sort, and after the loop no element has to cross the border of its <= 16-element piece (every piece
//     is <= the next one): the final pass equals a stable sort of every piece, done by ranks without a branch (piece_sort).
//   * A range whose keys are all tied is permuted in a way that depends on its length only (TiedPerms below).
// Pivot selection and the heapsort fallback are libstdc++'s own helpers.  The re-balance (KDTree::Balance every 100 frames, a full
// re-build on the host while the GPU waits) is this code: 310 -> ~130 ms of CPU time for a 500 k-point planar map.
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
// The one-build-per-node protocol (pfslam_shard_balance_build: rank 0 builds while every other rank of the node waits in the broadcast)
// lifts the split by LOCAL_WORLD_SIZE for the length of that build: pfslam_kd_whole_node(1) ... (0).
static std::atomic<int> g_whole_node{0};
extern "C" void pfslam_kd_whole_node(int on) { g_whole_node.store(on ? 1 : 0); }
int sort_thread_budget()
{
    if (g_whole_node.load() && !getenv("PFSLAM_SORT_THREADS")) {
        static const int whole = std::max(1, std::min(usable_cores(), 64));
        return whole;
    }
    static const int n = [] {
        if (const char *e = getenv("PFSLAM_SORT_THREADS")) return std::max(1, atoi(e));
        int ranks = 1;
        if (const char *e = getenv("LOCAL_WORLD_SIZE")) ranks = std::max(1, atoi(e));
        return std::max(1, std::min(usable_cores() / ranks, 64));
    }();
    return n;
}
std::atomic<int> g_sort_threads{0};
// PFSLAM_PLAIN_SORT=1: every sort of the build is the library's std::sort call, on one thread (what the tests compare the rest with)
bool plain_sort_forced()
{
    static const bool v = getenv("PFSLAM_PLAIN_SORT") && atoi(getenv("PFSLAM_PLAIN_SORT")) != 0;
    return v;
}

#if defined(__GLIBCXX__)
// libstdc++'s std::sort, taken apart: the helpers used below are its INTERNALS.  Guarded three ways: compiled only against libstdc++
// (any other library: plain std::sort), a static check of the insertion-sort threshold the code relies on, and a start-up self-check
// of the restated form against std::sort itself on heavily tied arrays (restated_sort_ok) -- a library that changes the pivot rule,
// the partition or the helpers' contracts turns the restated path off instead of silently building a different tree.  A map with a
// NaN coordinate (no strict weak order: nothing above holds) is built by the library call as well (pfslam_kd_create).
static_assert((int)std::_S_threshold == 16, "exact_sort restates libstdc++'s introsort: its insertion-sort threshold has changed");

// __unguarded_partition(first, last, pivot): the pivot sits in front of the range, the left scan stops on every element that is not
// below it, the right scan on every element that is not above it (both stop on ties), the two are swapped, and the scans go on until
// they meet; the meeting point is returned.  I.e. the k-th "left stopper" is swapped with the k-th "right stopper" for as long as the
// former lies left of the latter, and every swapped element lands where both scans have passed.  While the scan fronts are more than
// two blocks apart the stoppers of a block of 64 elements on either side are collected without a branch (BlockQuicksort's offset
// buffers) and swapped pairwise in order -- the scalar loop's swaps.  Where the fronts meet, [f, e) is what neither scan has passed
// (f: the next left stopper, or the front itself; e: one past the next right stopper): everything left of f stops the right scan and
// everything from e on stops the left scan, so with the stoppers A_0 < A_1 < ... and B_0 > B_1 > ... of [f, e) (sentinels e and f - 1)
// the scalar loop swaps A_k, B_k while A_k < B_k and then returns where its left scan stops: A_k, or the place B_(k-1) whose new
// element (the old A_(k-1)) stops it first.
template <typename T, typename Cmp>
T *block_partition(T *first, T *last, T *pivot, Cmp comp)
{
    constexpr int B = 64;
    unsigned char offL[2 * B + 2], offR[2 * B + 2];
    int nL = 0, nR = 0, sL = 0, sR = 0;
    T *l = first, *r = last;
    while (r - l > 2 * B) {
        if (nL == 0) {
            sL = 0;
            for (int i = 0; i < B; i++) {
                offL[nL] = (unsigned char)i;
                nL += !comp(l + i, pivot);
            }
        }
        if (nR == 0) {
            sR = 0;
            for (int i = 0; i < B; i++) {
                offR[nR] = (unsigned char)i;
                nR += !comp(pivot, r - 1 - i);
            }
        }
        const int m = std::min(nL, nR);
        for (int k = 0; k < m; k++) std::iter_swap(l + offL[sL + k], r - 1 - offR[sR + k]);
        nL -= m; nR -= m; sL += m; sR += m;
        if (nL == 0) l += B;
        if (nR == 0) r -= B;
    }
    T *f = nL > 0 ? l + offL[sL] : l, *e = nR > 0 ? r - offR[sR] : r;
    const int m = (int)(e - f); // <= 2 B
    int nA = 0, nB = 0;         // positions + 1: A ascending with the sentinel m + 1, B descending with the sentinel 0
    for (int i = 0; i < m; i++) {
        offL[nA] = (unsigned char)(i + 1);
        nA += !comp(f + i, pivot);
    }
    for (int i = m - 1; i >= 0; i--) {
        offR[nB] = (unsigned char)(i + 1);
        nB += !comp(pivot, f + i);
    }
    offL[nA] = (unsigned char)(m + 1);
    offR[nB] = 0;
    int k = 0;
    while (offL[k] < offR[k]) {
        std::iter_swap(f + (offL[k] - 1), f + (offR[k] - 1));
        k++;
    }
    return f + ((k > 0 ? std::min<int>(offL[k], offR[k - 1]) : (int)offL[0]) - 1);
}
// __unguarded_partition_pivot with the partition above
template <typename T, typename Cmp>
T *partition_pivot(T *first, T *last, Cmp comp)
{
    T *mid = first + (last - first) / 2;
    std::__move_median_to_first(first, first + 1, mid, last - 1, comp);
    return block_partition(first + 1, last, first, comp);
}
template <typename T, typename Key>
bool all_tied(const T *p, long n, Key key)
{
    const float k0 = key(p[0]);
    if (k0 != k0) return false; // a NaN in front says nothing about the rest
    for (long i = 1; i < n; i++)
        if (key(p[i]) < k0 || k0 < key(p[i])) return false;
    return true;
}
// a piece of <= 16 elements, sorted the way insertion sort leaves it: stably.  rank = elements below + equal elements in front.
template <typename T, typename Key>
inline void piece_sort(T *p, int s, Key key)
{
    if (s < 2) return;
    float k[16];
    for (int i = 0; i < s; i++) k[i] = key(p[i]);
    bool sorted = true;
    for (int i = 1; i < s; i++) sorted &= !(k[i] < k[i - 1]);
    if (sorted) return;
    for (int i = s; i < 16; i++) k[i] = __builtin_inff(); // (behind every real key, never in front of an equal one)
    int rank[16];
    for (int i = 0; i < 16; i++) {
        int r = 0;
        for (int j = 0; j < 16; j++) r += (int)(k[j] < k[i]) | ((int)(j < i) & (int)(k[j] == k[i]));
        rank[i] = r;
    }
    T tmp[16];
    for (int i = 0; i < s; i++) tmp[rank[i]] = p[i];
    memcpy(p, tmp, (size_t)s * sizeof(T));
}
// __introsort_loop + the part of __final_insertion_sort that belongs to the range, on one thread
template <typename T, typename Key, typename Cmp>
void introsort_loop(T *first, T *last, long depth_limit, Key key, Cmp comp)
{
    while (last - first > (long)std::_S_threshold) {
        if (depth_limit == 0) {
            std::__partial_sort(first, last, last, comp); // heapsort of the rest: sorted, nothing left for the insertion pass
            return;
        }
        if (all_tied(first, last - first, key)) { // every branch of the library's own loop is predictable here, and no element moves in the insertion pass
            std::__introsort_loop(first, last, depth_limit, comp);
            return;
        }
        --depth_limit;
        T *cut = partition_pivot(first, last, comp);
        introsort_loop(cut, last, depth_limit, key, comp);
        last = cut;
    }
    piece_sort(first, (int)(last - first), key);
}
// ... and with the recursive call on another thread while there are threads to be had
template <typename T, typename Key, typename Cmp>
void par_introsort_loop(T *first, T *last, long depth_limit, Key key, Cmp comp)
{
    std::vector<std::thread> kids;
    bool whole = false;
    while (last - first > (long)std::_S_threshold) {
        if (depth_limit == 0) {
            std::__partial_sort(first, last, last, comp);
            whole = true;
            break;
        }
        --depth_limit;
        T *cut = partition_pivot(first, last, comp);
        bool forked = false;
        if (last - cut > 16384 && g_sort_threads.fetch_add(1) < sort_thread_budget()) {
            try {
                kids.emplace_back([=] {
                    par_introsort_loop(cut, last, depth_limit, key, comp);
                    g_sort_threads.fetch_sub(1);
                });
                forked = true;
            } catch (const std::system_error &) { // no thread to be had: this one does the work
                g_sort_threads.fetch_sub(1);
            }
        } else if (last - cut > 16384) g_sort_threads.fetch_sub(1); // budget exhausted: undo the reservation
        if (!forked) introsort_loop(cut, last, depth_limit, key, comp);
        last = cut;
    }
    if (!whole) piece_sort(first, (int)(last - first), key);
    for (auto &t : kids) t.join();
}
template <typename T, typename Key>
void restated_sort(T *first, T *last, Key key)
{
    if (last - first < 2) return;
    auto comp = __gnu_cxx::__ops::__iter_comp_iter([key](const T &a, const T &b) { return key(a) < key(b); });
    const long depth = (long)std::__lg(last - first) * 2;
    if (last - first > 32768 && sort_thread_budget() > 1) par_introsort_loop(first, last, depth, key, comp);
    else introsort_loop(first, last, depth, key, comp);
}
// once per process: the restated sort against std::sort itself, byte for byte -- 40 000 points on 7 distinct keys (thousands of ties,
// several threads), 3 000 on 300, 9 000 untied, 700 and 13 all tied, and a range of integers under an always-false comparison -- or
// every sort of the build is the library call
bool restated_sort_ok()
{
    static const bool ok = [] {
        uint32_t st = 12345u;
        const int sizes[6] = {40000, 3000, 9000, 700, 13, 300}, distinct[6] = {7, 300, 1 << 20, 1, 1, 2};
        for (int t = 0; t < 6; t++) {
            std::vector<Pt> a(sizes[t]);
            for (size_t i = 0; i < a.size(); i++) {
                st = st * 1664525u + 1013904223u;
                a[i] = Pt{(float)((st >> 10) % (uint32_t)distinct[t]) * 0.025f, (float)i, 0.0f, (float)(st >> 8)};
            }
            std::vector<Pt> b = a;
            std::sort(a.begin(), a.end(), byX);
            restated_sort(b.data(), b.data() + b.size(), [](const Pt &p) { return p.x; });
            if (memcmp(a.data(), b.data(), a.size() * sizeof(Pt)) != 0) return false;
        }
        std::vector<int> u(5000), v;
        std::iota(u.begin(), u.end(), 0);
        v = u;
        std::sort(u.begin(), u.end(), [](int, int) { return false; });
        restated_sort(v.data(), v.data() + v.size(), [](int) { return 0.0f; });
        return u == v;
    }();
    return ok;
}
#endif
// std::sort(first, last, key(a) < key(b)) -- its result
template <typename T, typename Key>
void exact_sort(T *first, T *last, Key key)
{
    if (last - first < 2) return;
#if defined(__GLIBCXX__)
    if (!plain_sort_forced() && restated_sort_ok()) {
        restated_sort(first, last, key);
        return;
    }
#endif
    std::sort(first, last, [key](const T &a, const T &b) { return key(a) < key(b); }); // the library call itself
}

// Ranges whose keys are ALL TIED.  A sort never looks at anything but the outcome of its comparisons, so on a range of `len`
// elements that all compare equal it performs a fixed sequence of moves: a permutation that depends on `len` only.  The map of a 2-D
// LiDAR is planar -- every z level of KDTree::Create (kdtree.cpp:45-50: a third of the levels) sorts ties only --, and deep in the tree
// a range is often one piece of an axis-parallel wall (all x or all y equal).  The sub-ranges of one depth have at most two different
// lengths (mid and count - mid - 1 of lengths that differ by at most one), so the build keeps one permutation per (depth, length),
// obtained by running the very same sort on the indices 0 .. len-1 with a comparison that is always false, and replaces the
// sort of a tied range by a gather: O(len) instead of O(len log len), the same bytes.
struct TiedPerms {
    struct Entry {
        int len = 0;
        std::once_flag once;
        std::vector<int> perm;
    };
    std::vector<std::unique_ptr<Entry>> entries; // two per depth
    explicit TiedPerms(int n)
    {
        int lo = n, hi = n; // the lengths of a depth are lo and hi (hi - lo <= 1)
        while (hi > 0) {
            for (int len : {lo, hi}) {
                entries.emplace_back(new Entry);
                entries.back()->len = len;
            }
            const int nlo = std::min(lo / 2, lo - lo / 2 - 1), nhi = std::max(hi / 2, hi - hi / 2 - 1);
            lo = std::max(nlo, 0);
            hi = nhi;
        }
    }
    const int *get(int depth, int len)
    {
        if (depth < 0 || (size_t)(2 * depth + 1) >= entries.size()) return nullptr;
        Entry *e = entries[2 * depth]->len == len ? entries[2 * depth].get() : entries[2 * depth + 1]->len == len ? entries[2 * depth + 1].get() : nullptr;
        if (!e) return nullptr; // (not a length of this depth: the plain sort)
        std::call_once(e->once, [e] {
            e->perm.resize(e->len);
            std::iota(e->perm.begin(), e->perm.end(), 0);
            exact_sort(e->perm.data(), e->perm.data() + e->len, [](int) { return 0.0f; });
        });
        return e->perm.data();
    }
};
// the sort of one level: std::sort's result, by a gather when the range is tied
template <typename Key>
void level_sort(Pt *first, int n, Key key, TiedPerms *tied, int depth)
{
    if (!tied) { // PFSLAM_PLAIN_SORT, or a map with a NaN coordinate: the library call, nothing else
        std::sort(first, first + n, [key](const Pt &a, const Pt &b) { return key(a) < key(b); });
        return;
    }
    if (n > 16 && all_tied(first, n, key)) {
        if (const int *perm = tied->get(depth, n)) {
            Pt small[64];
            std::vector<Pt> big;
            Pt *tmp = small;
            if (n > 64) {
                big.assign(first, first + n);
                tmp = big.data();
            } else memcpy(small, first, (size_t)n * sizeof(Pt));
            for (int i = 0; i < n; i++) first[i] = tmp[perm[i]];
            return;
        }
    }
    exact_sort(first, first + n, key);
}

// `fork` > 0: the two sub-ranges are disjoint slices of `buf` and disjoint slices of `out` (pre-order layout), so
// they are built on two threads; every sort still sees exactly the sequence the sequential build would give it,
// hence the same (unstable-sort dependent) topology as the reference.
void build_range(std::vector<Pt> &buf, int lo, int hi, pfslam_node *out, int idx, int parent, int fork, TiedPerms *tied = nullptr, int depth = 0)
{
    const int axis = parent < 0 ? 0 : (out[parent].axis + 1) % 3;
    Pt *first = buf.data() + lo;
    switch (axis) {
    case 0: level_sort(first, hi - lo, [](const Pt &p) { return p.x; }, tied, depth); break;
    case 1: level_sort(first, hi - lo, [](const Pt &p) { return p.y; }, tied, depth); break;
    default: level_sort(first, hi - lo, [](const Pt &p) { return p.z; }, tied, depth); break;
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
            t = std::thread([&] { build_range(buf, lo, lo + mid, out, idx + 1, idx, fork - 1, tied, depth + 1); });
        } catch (const std::system_error &) { // no thread to be had (an extern "C" entry point must not terminate): in line
            build_range(buf, lo, lo + mid, out, idx + 1, idx, 0, tied, depth + 1);
        }
        build_range(buf, lo + mid + 1, hi, out, idx + mid + 1, idx, fork - 1, tied, depth + 1);
        if (t.joinable()) t.join();
        return;
    }
    if (has_left) build_range(buf, lo, lo + mid, out, idx + 1, idx, 0, tied, depth + 1);
    if (has_right) build_range(buf, lo + mid + 1, hi, out, idx + mid + 1, idx, 0, tied, depth + 1);
}
} // namespace

extern "C" int pfslam_kd_create(const float *pts_xyzw, int n, pfslam_node *out)
{
    if (n < 0 || (n > 0 && (!pts_xyzw || !out))) return 1;
    if (n == 0) return 0;
    std::vector<Pt> buf(n);
    TiedPerms tied(n);
    // PFSLAM_PLAIN_SORT, or a NaN coordinate (no strict weak order: what std::sort does then is its own business): no shortcut of any kind
    std::atomic<int> nan{plain_sort_forced() ? 1 : 0}, zdiff{0};
    const float z0 = pts_xyzw[2];
    {   // copy + NaN scan, in slices on the threads the sorts will use (16 MB of traffic at 500 k points: 1 ms on one)
        const int T = n >= (1 << 17) ? std::min(sort_thread_budget(), 8) : 1;
        auto slice = [&](int t) {
            const int lo = (int)((long long)n * t / T), hi = (int)((long long)n * (t + 1) / T);
            bool bad = false, zd = false;
            for (int i = lo; i < hi; i++) {
                buf[i] = Pt{pts_xyzw[4 * i], pts_xyzw[4 * i + 1], pts_xyzw[4 * i + 2], pts_xyzw[4 * i + 3]};
                bad |= buf[i].x != buf[i].x || buf[i].y != buf[i].y || buf[i].z != buf[i].z;
                zd |= buf[i].z != z0;
            }
            if (bad) nan = 1;
            if (zd) zdiff = 1;
        };
        std::vector<std::thread> ts;
        for (int t = 1; t < T; t++) {
            try { ts.emplace_back(slice, t); } catch (const std::system_error &) { slice(t); }
        }
        slice(0);
        for (auto &t : ts) t.join();
    }
    const bool plain = nan.load() != 0;
    // The permutations of the upper z levels of a planar map depend on n only: a thread of its own has them ready by the time the build
    // gets there (depth 2 at 500 k points: a 125 k-index sort, ~1 ms that four sub-builds would otherwise wait for)
    std::thread ahead;
    if (!plain && !zdiff.load() && n >= (1 << 16) && sort_thread_budget() > 2) {
        try {
            ahead = std::thread([&] {
                for (int d = 2; d <= 8; d += 3)
                    for (int e = 0; e < 2; e++)
                        if ((size_t)(2 * d + e) < tied.entries.size()) tied.get(d, tied.entries[2 * d + e]->len);
            });
        } catch (const std::system_error &) {}
    }
    struct Joiner { std::thread &t; ~Joiner() { if (t.joinable()) t.join(); } } joiner{ahead};
    level_sort(buf.data(), n, [](const Pt &p) { return p.x; }, plain ? nullptr : &tied, -1); // KDTree::Create pre-sorts on x before the recursive sort
    const unsigned hw = (unsigned)sort_thread_budget(); // usable cores, shared with the node's other ranks
    // up to 16 concurrent sub-builds, 32 for a big map (500 k points on the 16 cores of the GPU box: 29 -> 25 ms; no gain at 100 k)
    build_range(buf, 0, n, out, 0, -1, hw >= 16 ? (n >= 200000 ? 5 : 4) : hw >= 8 ? 3 : hw >= 4 ? 2 : hw >= 2 ? 1 : 0, plain ? nullptr : &tied);
    return 0;
}

// KDTree::InsertList (kdtree.cpp:46-67): the sub-tree of `n` points in pre-order at list[idx ...], hanging below `parent`
// (split axis = parent's + 1; -1 = root).  Like the reference it does not touch the parent's child links.
extern "C" int pfslam_kd_insert_list(const float *pts_xyzw, int n, pfslam_node *list, int idx, int parent)
{
    if (n <= 0 || !pts_xyzw || !list || idx < 0 || parent < -1) return 1;
    std::vector<Pt> buf(n);
    for (int i = 0; i < n; i++) buf[i] = Pt{pts_xyzw[4 * i], pts_xyzw[4 * i + 1], pts_xyzw[4 * i + 2], pts_xyzw[4 * i + 3]};
    TiedPerms tied(n); // (its lengths are those of a range of n: the depth counts from the sub-tree's root)
    bool plain = plain_sort_forced();
    for (int i = 0; i < n && !plain; i++) plain = buf[i].x != buf[i].x || buf[i].y != buf[i].y || buf[i].z != buf[i].z;
    build_range(buf, 0, n, list, idx, parent, 0, plain ? nullptr : &tied);
    return 0;
}

// 1 = the parallel form of the host sort is in use (libstdc++, more than one thread, self-check passed); 0 = plain std::sort
extern "C" int pfslam_kd_parallel_sort(void)
{
#if defined(__GLIBCXX__)
    return sort_thread_budget() > 1 && !plain_sort_forced() && restated_sort_ok() ? 1 : 0;
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
