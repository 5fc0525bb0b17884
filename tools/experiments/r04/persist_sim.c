// persist_sim.c -- CPU prototype for lattice-cell rows that PERSIST across frames (round 4, VERDICT r03 task 1).
//
// Between two re-balances the tree is append-only (KDTree::InsertNode, kdtree.cpp:69-105): a cell's rows (first-descent candidates
// + re-descent candidates, kd_cells.hip.inc) stay valid until a node is hung on a link that the cell's first descent or one of its
// re-descent walks ENDS on.  This program replays consecutive bench frames (make_frames.py) and reports, per frame:
//   cells hit, of them never built (new) / built but invalidated by an insert since / reused; the share of the frame's queries in
//   each class; the share of (wave, beam) rows of the scan-match kernel with at least one lane in a new / a non-reused cell (what an
//   asynchronous build -- rows available one frame late -- would send down the generic traversal); the longest chain of dependent
//   node reads among the cells that have to be (re)built.
//   gcc -O2 -fopenmp -o /tmp/persist_sim tools/experiments/r04/persist_sim.c -lm && /tmp/persist_sim /tmp/pf 6 45
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct { int axis, left, right, parent; float x, y, z, w; } Node;
static void *slurp(const char *p, size_t *n)
{
    FILE *f = fopen(p, "rb");
    if (!f) { perror(p); exit(1); }
    fseek(f, 0, SEEK_END); *n = ftell(f); fseek(f, 0, SEEK_SET);
    void *b = malloc(*n);
    if (fread(b, 1, *n, f) != *n) exit(1);
    fclose(f);
    return b;
}
static const float RES = 0.025f;
static int lattice(float q)
{
    int k = (int)floorf(q / RES);
    if ((float)(k + 1) * RES <= q) k++;
    else if (!((float)k * RES <= q)) k--;
    return k;
}
#define MAXT 40
typedef struct { int64_t key; int built; int nterm; int term[MAXT]; int hit_frame; int count; int depth; } Cell;
static Cell *tab;
static size_t cap = 1u << 22;
static Cell *find(int64_t key, int create)
{
    size_t h = (size_t)((uint64_t)key * 0x9E3779B97F4A7C15ull >> 40) & (cap - 1);
    while (tab[h].key != -1 && tab[h].key != key) h = (h + 1) & (cap - 1);
    if (tab[h].key == -1) {
        if (!create) return NULL;
        tab[h].key = key; tab[h].built = -1; tab[h].hit_frame = -1;
    }
    return &tab[h];
}
// build the rows' dependency set of a cell: terminal links (node * 2 + side) of the first descent and of every re-descent walk
static void build(Cell *c, const Node *t, int frame)
{
    const int kx = (int)(c->key >> 24) - 100000, ky = (int)(c->key & 0xffffff) - 100000;
    const float xlo = (float)kx * RES, xhi = (float)(kx + 1) * RES, ylo = (float)ky * RES, yhi = (float)(ky + 1) * RES;
    int head = 0, nc = 0, depth = 0, last = 0, side = 0;
    float lbs[256]; int idx[256];
    float U = INFINITY;
    while (head >= 0) {
        const Node *nd = &t[head];
        const float dxn = fmaxf(fmaxf(xlo - nd->x, nd->x - xhi), 0.0f), dyn = fmaxf(fmaxf(ylo - nd->y, nd->y - yhi), 0.0f);
        const float dxf = fmaxf(fabsf(nd->x - xlo), fabsf(nd->x - xhi)), dyf = fmaxf(fabsf(nd->y - ylo), fabsf(nd->y - yhi));
        const float lb = (dxn * dxn + dyn * dyn) * 0.99999f, ub = (dxf * dxf + dyf * dyf) * 1.00001f;
        if (ub < U) U = ub;
        if (lb <= U && nc < 256) { lbs[nc] = lb; idx[nc] = head; nc++; }
        depth++;
        const int gl = nd->axis == 0 ? xlo < nd->x : nd->axis == 1 ? ylo < nd->y : 0;
        last = head; side = gl ? 0 : 1;
        head = gl ? nd->left : nd->right;
    }
    c->nterm = 0;
    c->term[c->nterm++] = last * 2 + side;
    int maxwalk = 0;
    for (int k = 0; k < nc; k++) if (lbs[k] <= U) {
        const int ci = idx[k], pi = t[ci].parent;
        if (pi < 0) continue;
        const Node *pn = &t[pi];
        const int lt = pn->axis == 0 ? xlo < pn->x : pn->axis == 1 ? ylo < pn->y : 0;
        int h2 = lt ? pn->right : pn->left, l2 = pi, s2 = lt ? 1 : 0, walk = 1;
        while (h2 >= 0) {
            const Node *nd = &t[h2];
            const int gl = nd->axis == 0 ? xlo < nd->x : nd->axis == 1 ? ylo < nd->y : 0;
            l2 = h2; s2 = gl ? 0 : 1;
            h2 = gl ? nd->left : nd->right;
            walk++;
        }
        if (walk > maxwalk) maxwalk = walk;
        if (c->nterm < MAXT) c->term[c->nterm++] = l2 * 2 + s2;
        else { fprintf(stderr, "MAXT\n"); }
    }
    c->built = frame;
    c->depth = depth + maxwalk;
}
int main(int argc, char **argv)
{
    const int first = atoi(argv[2]), last = atoi(argv[3]);
    tab = malloc(cap * sizeof(Cell));
    for (size_t i = 0; i < cap; i++) tab[i].key = -1;
    int *stamp = calloc(2 * 400000, sizeof(int)); // frame in which link (node * 2 + side) last received a node
    int Kprev = -1;
    printf("frame nodes inserted | cells hit new invalidated reused | %% queries new inval | %% (wave,beam) rows with a new-cell lane / a non-reused lane | rebuild: cells, deepest chain (mean)\n");
    for (int f = first; f <= last; f++) {
        char path[512];
        size_t nb;
        snprintf(path, sizeof path, "%s.%d.nodes", argv[1], f); Node *t = slurp(path, &nb); const int K = nb / sizeof(Node);
        snprintf(path, sizeof path, "%s.%d.particles", argv[1], f); float *P = slurp(path, &nb); const int N = nb / 12;
        snprintf(path, sizeof path, "%s.%d.scan", argv[1], f); float *scan = slurp(path, &nb); const int B = nb / 4;
        if (Kprev >= 0)
            for (int i = Kprev; i < K; i++) {
                const int p = t[i].parent;
                stamp[p * 2 + (t[p].right == i ? 1 : 0)] = f; // touched before frame f's scoring pass
                if (t[p].left == i && t[p].right == i) fprintf(stderr, "both\n");
            }
        // lane order: Morton order over the cloud (stand-in for the product's Hilbert counting sort)
        double mu[3] = {0, 0, 0}, sd[3] = {0, 0, 0};
        for (int i = 0; i < N; i++) for (int d = 0; d < 3; d++) mu[d] += P[3 * i + d];
        for (int d = 0; d < 3; d++) mu[d] /= N;
        for (int i = 0; i < N; i++) for (int d = 0; d < 3; d++) sd[d] += (P[3 * i + d] - mu[d]) * (P[3 * i + d] - mu[d]);
        for (int d = 0; d < 3; d++) sd[d] = sqrt(sd[d] / N);
        const double reach = 8.0;
        double e = fmax(fmax(sd[0], sd[1]), sd[2] * reach) * 6.4 / 64;
        uint64_t *ord = malloc(N * sizeof(uint64_t));
        for (int i = 0; i < N; i++) {
            unsigned c[3];
            for (int d = 0; d < 3; d++) {
                double v = (P[3 * i + d] - mu[d]) * (d == 2 ? reach : 1.0) / e + 32;
                c[d] = (unsigned)fmin(fmax(v, 0), 63);
            }
            uint64_t m = 0;
            for (int b = 0; b < 6; b++) for (int d = 0; d < 3; d++) m |= (uint64_t)((c[d] >> b) & 1) << (3 * b + d);
            ord[i] = (m << 32) | (unsigned)i;
        }
        int cmp(const void *a, const void *b) { return *(const uint64_t *)a < *(const uint64_t *)b ? -1 : *(const uint64_t *)a > *(const uint64_t *)b; }
        qsort(ord, N, sizeof(uint64_t), cmp);
        int64_t *keys = malloc((size_t)N * B * sizeof(int64_t));
#pragma omp parallel for schedule(static)
        for (int s = 0; s < N; s++) {
            const int i = (int)(ord[s] & 0xffffffffu);
            const float x = P[3 * i], y = P[3 * i + 1], th = P[3 * i + 2];
            for (int j = 0; j < B; j++) {
                const float rot = ((-135.0f + j * .25f) * 3.14159265f) / 180.0f + th;
                float wx = scan[j] * cosf(rot), wy = scan[j] * sinf(rot);
                int64_t key = -1;
                if (fabsf(wx) < 20.0f && fabsf(wy) < 20.0f) {
                    wx += x; wy += y;
                    key = ((int64_t)(lattice(wx) + 100000) << 24) | (int64_t)(lattice(wy) + 100000);
                }
                keys[(size_t)s * B + j] = key;
            }
        }
        long queries = 0;
        int nhit = 0;
        Cell **hits = malloc(sizeof(Cell *) * 1000000);
        for (size_t q = 0; q < (size_t)N * B; q++) {
            if (keys[q] < 0) continue;
            static int64_t lk = -2; static Cell *lc;
            Cell *c = keys[q] == lk ? lc : find(keys[q], 1);
            lk = keys[q]; lc = c;
            if (c->hit_frame != f) { c->hit_frame = f; c->count = 0; hits[nhit++] = c; }
            c->count++;
            queries++;
        }
        // classify: 0 reused, 1 new, 2 invalidated
        int n_new = 0, n_inv = 0, n_re = 0, deepest = 0; long q_new = 0, q_inv = 0, depth_sum = 0;
        for (int k = 0; k < nhit; k++) {
            Cell *c = hits[k];
            int cls = 0;
            if (c->built < 0) cls = 1;
            else for (int u = 0; u < c->nterm; u++) if (stamp[c->term[u]] > c->built) { cls = 2; break; }
            // note: stamp == f means "touched before f's scoring"; a cell built in frame f' saw the tree of frame f', which includes
            // the touches stamped f' -> invalid iff stamp > built
            c->depth = -cls; // class for the second pass (negative), overwritten by build
        }
        // wave-level
        long rows = 0, rows_new = 0, rows_any = 0;
        for (int s0 = 0; s0 < N; s0 += 64)
            for (int j = 0; j < B; j++) {
                int any = 0, nw = 0, act = 0;
                for (int s = s0; s < s0 + 64 && s < N; s++) {
                    const int64_t key = keys[(size_t)s * B + j];
                    if (key < 0) continue;
                    act = 1;
                    const Cell *c = find(key, 0);
                    if (c->depth == -1) nw = any = 1;
                    else if (c->depth == -2) any = 1;
                }
                rows += act; rows_new += nw; rows_any += any;
            }
        for (int k = 0; k < nhit; k++) {
            Cell *c = hits[k];
            const int cls = -c->depth;
            if (cls == 0) { n_re++; c->depth = 0; continue; }
            if (cls == 1) { n_new++; q_new += c->count; } else { n_inv++; q_inv += c->count; }
            build(c, t, f);
            if (c->depth > deepest) deepest = c->depth;
            depth_sum += c->depth;
        }
        printf("%3d %6d %4d | %5d %5d %5d %5d | %6.3f %6.3f | %6.3f %6.3f | %5d %3d (%.1f)\n", f, K, Kprev >= 0 ? K - Kprev : 0, nhit, n_new, n_inv, n_re,
               100.0 * q_new / queries, 100.0 * q_inv / queries, 100.0 * rows_new / rows, 100.0 * rows_any / rows, n_new + n_inv, deepest,
               (n_new + n_inv) ? (double)depth_sum / (n_new + n_inv) : 0.0);
        fflush(stdout);
        Kprev = K;
        free(t); free(P); free(scan); free(keys); free(ord); free(hits);
    }
    return 0;
}
