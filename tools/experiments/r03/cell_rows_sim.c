// cell_rows_sim.c -- CPU prototype of the "lattice-cell rows" formulation of the scan-match kernel (round 3).
//
// Every map point the SLAM step produces lies on the lattice k * res (ROUND_FRAC, kernel.cu:52), so every split plane of the
// KD tree is a lattice plane, and the FIRST DESCENT of the reference traversal (kernel.cu:1239-1259) -- a chain of
// `query < node` decisions -- is the same for every query inside one lattice cell.  Per cell one can therefore precompute the
// few nodes of that path that can be the nearest one for some point of the cell (the pruning of the round-2 plan, with W = the
// 2.5 cm cell instead of a wave's ~10 cm box).  This program measures, on an aged map:
//   * distinct cells the 100 k x 1081 queries of a frame fall into, queries per cell
//   * path length and surviving candidates per cell (query-weighted)
//   * how often the parent-hyperplane test after the first descent passes (a per-lane re-descent follows)
//   gcc -O2 -o /tmp/cell_rows_sim tools/experiments/r03/cell_rows_sim.c -lm && /tmp/cell_rows_sim /tmp/aged
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
static int lattice(float q) // max k with fl(k * res) <= q
{
    int k = (int)floorf(q / RES);
    if ((float)(k + 1) * RES <= q) k++;
    else if (!((float)k * RES <= q)) k--;
    return k;
}
typedef struct { int64_t key; int count; int row; } Slot;
#define MAXC 12
typedef struct { int n; int idx[MAXC]; unsigned char rd_nonempty[MAXC]; } Row;
int main(int argc, char **argv)
{
    char path[512];
    size_t nb;
    snprintf(path, sizeof path, "%s.nodes", argv[1]); Node *t = slurp(path, &nb); int K = nb / sizeof(Node);
    snprintf(path, sizeof path, "%s.particles", argv[1]); float *P = slurp(path, &nb); int N = nb / 12;
    snprintf(path, sizeof path, "%s.scan", argv[1]); float *scan = slurp(path, &nb); int B = nb / 4;
    int first_new = argc > 2 ? atoi(argv[2]) : 100000;
    printf("%d nodes, %d particles, %d beams\n", K, N, B);
    size_t cap = 1u << 24;
    Slot *tab = calloc(cap, sizeof(Slot));
    for (size_t i = 0; i < cap; i++) tab[i].key = -1;
    long queries = 0;
    for (int i = 0; i < N; i++) {
        const float x = P[3 * i], y = P[3 * i + 1], th = P[3 * i + 2];
        for (int j = 0; j < B; j++) {
            const float rot = ((-135.0f + j * .25f) * 3.14159265f) / 180.0f + th;
            float wx = scan[j] * cosf(rot), wy = scan[j] * sinf(rot);
            if (!(fabsf(wx) < 20.0f && fabsf(wy) < 20.0f)) continue;
            wx += x; wy += y;
            const int64_t key = ((int64_t)(lattice(wx) + 100000) << 24) | (int64_t)(lattice(wy) + 100000);
            size_t h = (size_t)(key * 0x9E3779B97F4A7C15ull >> 40) & (cap - 1);
            while (tab[h].key != -1 && tab[h].key != key) h = (h + 1) & (cap - 1);
            tab[h].key = key; tab[h].count++;
            queries++;
        }
    }
    Row *rows = calloc(200000, sizeof(Row));
    long cells = 0, hist_c[34] = {0}, hist_q[34] = {0}, path_q = 0, new_on_path_q = 0, cand_new_q = 0;
    double cand_q = 0;
    for (size_t s = 0; s < cap; s++) {
        if (tab[s].key == -1) continue;
        cells++;
        const int kx = (int)(tab[s].key >> 24) - 100000, ky = (int)(tab[s].key & 0xffffff) - 100000;
        const float xlo = (float)kx * RES, xhi = (float)(kx + 1) * RES, ylo = (float)ky * RES, yhi = (float)(ky + 1) * RES;
        int head = 0, nc = 0, plen = 0, nnew = 0;
        float lbs[64]; int idx[64];
        float U = INFINITY;
        while (head >= 0) {
            const Node *nd = &t[head];
            const float dxn = fmaxf(fmaxf(xlo - nd->x, nd->x - xhi), 0.0f), dyn = fmaxf(fmaxf(ylo - nd->y, nd->y - yhi), 0.0f);
            const float dxf = fmaxf(fabsf(nd->x - xlo), fabsf(nd->x - xhi)), dyf = fmaxf(fabsf(nd->y - ylo), fabsf(nd->y - yhi));
            const float lb = (dxn * dxn + dyn * dyn) * 0.99999f, ub = (dxf * dxf + dyf * dyf) * 1.00001f;
            if (ub < U) U = ub;
            if (lb <= U && nc < 64) { lbs[nc] = lb; idx[nc] = head; nc++; }
            plen++;
            if (head >= first_new) nnew++;
            int go_left;
            if (nd->axis == 0) go_left = xlo < nd->x;      // the whole cell lies on one side: planes are lattice planes
            else if (nd->axis == 1) go_left = ylo < nd->y;
            else go_left = 0;
            head = go_left ? nd->left : nd->right;
        }
        int m = 0, mnew = 0;
        Row *row = &rows[cells - 1];
        tab[s].row = (int)cells - 1;
        row->n = 0;
        for (int k = 0; k < nc; k++) if (lbs[k] <= U) {
            m++; if (idx[k] >= first_new) mnew++;
            if (row->n < MAXC) {
                // re-descent from this candidate (if it wins): sibling side of its parent, path decided by the cell again;
                // can any node there beat a best distance <= sqrt(U)?
                const int c = idx[k], pi = t[c].parent;
                int nonempty = 0;
                if (pi >= 0) {
                    const Node *pn = &t[pi];
                    int lt = pn->axis == 0 ? xlo < pn->x : pn->axis == 1 ? ylo < pn->y : 0;
                    int h2 = lt ? pn->right : pn->left; // the side the query is NOT on
                    while (h2 >= 0) {
                        const Node *nd = &t[h2];
                        const float dxn = fmaxf(fmaxf(xlo - nd->x, nd->x - xhi), 0.0f), dyn = fmaxf(fmaxf(ylo - nd->y, nd->y - yhi), 0.0f);
                        if ((dxn * dxn + dyn * dyn) * 0.99999f <= U) nonempty = 1;
                        int gl = nd->axis == 0 ? xlo < nd->x : nd->axis == 1 ? ylo < nd->y : 0;
                        h2 = gl ? nd->left : nd->right;
                    }
                }
                row->idx[row->n] = c; row->rd_nonempty[row->n] = nonempty; row->n++;
            }
        }
        hist_c[m > 33 ? 33 : m]++;
        hist_q[m > 33 ? 33 : m] += tab[s].count;
        cand_q += (double)m * tab[s].count;
        cand_new_q += (long)mnew * tab[s].count;
        path_q += (long)plen * tab[s].count;
        new_on_path_q += (long)nnew * tab[s].count;
    }
    // second pass: what does a query do after its first descent?
    long q_pass = 0, q_fallback = 0, q_over = 0, q_root = 0;
    for (int i = 0; i < N; i++) {
        const float x = P[3 * i], y = P[3 * i + 1], th = P[3 * i + 2];
        for (int j = 0; j < B; j++) {
            const float rot = ((-135.0f + j * .25f) * 3.14159265f) / 180.0f + th;
            float wx = scan[j] * cosf(rot), wy = scan[j] * sinf(rot);
            if (!(fabsf(wx) < 20.0f && fabsf(wy) < 20.0f)) continue;
            wx += x; wy += y;
            const int64_t key = ((int64_t)(lattice(wx) + 100000) << 24) | (int64_t)(lattice(wy) + 100000);
            size_t h = (size_t)(key * 0x9E3779B97F4A7C15ull >> 40) & (cap - 1);
            while (tab[h].key != key) h = (h + 1) & (cap - 1);
            const Row *row = &rows[tab[h].row];
            if (row->n >= MAXC) { q_over++; continue; }
            float best = INFINITY; int bk = -1;
            for (int k = 0; k < row->n; k++) {
                const Node *nd = &t[row->idx[k]];
                const float dx = nd->x - wx, dy = nd->y - wy, d = dx * dx + dy * dy;
                if (d < best) { best = d; bk = k; }
            }
            const int pi = t[row->idx[bk]].parent;
            if (pi < 0) { q_root++; continue; }
            const Node *pn = &t[pi];
            const float hd = pn->axis == 0 ? fabsf(wx - pn->x) : pn->axis == 1 ? fabsf(wy - pn->y) : 0.0f;
            if (hd < sqrtf(best)) { q_pass++; if (row->rd_nonempty[bk]) q_fallback++; }
        }
    }
    printf("after the first descent: parent-plane test passes for %.2f %% of the queries; of all queries %.2f %% have a re-descent that\n"
           "can change the best node (generic per-lane tail needed); %.2f %% in cells with >= %d candidates\n",
           100.0 * q_pass / queries, 100.0 * q_fallback / queries, 100.0 * q_over / queries, MAXC);
    printf("queries %ld, distinct cells %ld (%.1f queries per cell)\n", queries, cells, (double)queries / cells);
    printf("query-weighted: path %.2f nodes (%.2f inserted since the balance), candidates %.2f (%.2f inserted)\n", (double)path_q / queries,
           (double)new_on_path_q / queries, cand_q / queries, (double)cand_new_q / queries);
    printf("candidates per cell: share of cells / share of queries\n");
    for (int m = 0; m < 34; m++)
        if (hist_c[m]) printf("  %2d%s  %6.2f %%  %6.2f %%\n", m, m == 33 ? "+" : " ", 100.0 * hist_c[m] / cells, 100.0 * hist_q[m] / queries);
    return 0;
}
