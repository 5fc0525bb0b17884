// offline prototype of the shared-prefix plan WITH forks; verifies exactness vs the plain traversal and reports work counts
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef struct { int axis, left, right, parent; float x, y, z, w; } node;
static node *T; static int N;
static float *P; static int NP; static float *S; static int NB = 1081;
static int FMAX = 0, CMAX = 11;

static int resume_from(float px, float py, float sBest, int best, int head, long *visits, int *trail_len)
{
    int prev = -1; int tl = 0;
    for (;;) {
        while (head >= 0) {
            node *n = &T[head]; (*visits)++; tl++;
            float dx = n->x - px, dy = n->y - py; float s = dx * dx + dy * dy;
            if (sqrtf(s) < sqrtf(sBest)) { sBest = s; best = head; }
            float q = n->axis == 0 ? px : (n->axis == 1 ? py : 0.f), v = n->axis == 0 ? n->x : (n->axis == 1 ? n->y : 0.f);
            head = (q < v) ? n->left : n->right;
        }
        if (best == prev) break;
        prev = best;
        int pi = T[best].parent; if (pi < 0) break;
        node *p = &T[pi];
        float q = p->axis == 0 ? px : (p->axis == 1 ? py : 0.f), v = p->axis == 0 ? p->x : (p->axis == 1 ? p->y : 0.f);
        if (!(fabsf(q - v) < sqrtf(sBest))) break;
        head = (q < v) ? p->right : p->left;
    }
    *trail_len = tl;
    return best;
}

typedef struct { float xlo, xhi, ylo, yhi; } box;
typedef struct { int idx; float lb; unsigned care, want; } cand;
typedef struct { unsigned care, want; float U; int resume; } leaf;
typedef struct { int axis; float v; } fork_t;
static cand C[64]; static int nC; static leaf L[16]; static int nL; static fork_t F[8]; static int nF; static long plan_visits;
static int overflow;

static void walk(int nd, box b, unsigned care, unsigned want, float U)
{
    while (nd >= 0) {
        node *n = &T[nd];
        float lo = n->axis == 0 ? b.xlo : b.ylo, hi = n->axis == 0 ? b.xhi : b.yhi, v = n->axis == 0 ? n->x : n->y;
        int all_left = hi < v, all_right = lo >= v;
        int straddle = n->axis < 2 && !(all_left || all_right);
        if (straddle && nF >= FMAX) break;   // lanes take over here
        if (nC >= CMAX || straddle) { // drop the entries of the CURRENT box that the tighter U has ruled out (entries of enclosing boxes are shared with other branches)
            int m = 0;
            for (int c = 0; c < nC; c++) { if (C[c].care == care && C[c].want == want && C[c].lb > U) continue; C[m++] = C[c]; }
            nC = m;
            if (nC >= CMAX) break;
        }
        plan_visits++;
        float dxn = fmaxf(fmaxf(b.xlo - n->x, n->x - b.xhi), 0.f), dyn = fmaxf(fmaxf(b.ylo - n->y, n->y - b.yhi), 0.f);
        float dxf = fmaxf(fabsf(n->x - b.xlo), fabsf(n->x - b.xhi)), dyf = fmaxf(fabsf(n->y - b.ylo), fabsf(n->y - b.yhi));
        float lb = (dxn * dxn + dyn * dyn) * 0.99999f, ub = (dxf * dxf + dyf * dyf) * 1.00001f;
        U = fminf(U, ub);
        if (lb <= U) { C[nC].idx = nd; C[nC].lb = lb; C[nC].care = care; C[nC].want = want; nC++; }
        if (straddle) {
            int f = nF++; F[f].axis = n->axis; F[f].v = v;
            box bl = b, br = b;
            if (n->axis == 0) { bl.xhi = v; br.xlo = v; } else { bl.yhi = v; br.ylo = v; }   // left: q < v ; right: q >= v (closed bound is conservative)
            walk(n->left, bl, care | (1u << f), want | (1u << f), U);
            walk(n->right, br, care | (1u << f), want, U);
            return;
        }
        nd = (n->axis < 2 && all_left) ? n->left : n->right;
    }
    L[nL].care = care; L[nL].want = want; L[nL].U = U; L[nL].resume = nd; nL++;
}

static unsigned spread6(unsigned v) { v &= 0x3f; v = (v | (v << 8)) & 0x300f; v = (v | (v << 4)) & 0x30c3; v = (v | (v << 2)) & 0x9249; return v; }
static unsigned hilbert(unsigned a, unsigned b, unsigned c)
{
    unsigned X[3] = {a, b, c}; const unsigned M = 1u << 5;
    for (unsigned Q = M; Q > 1; Q >>= 1) { unsigned Pm = Q - 1; for (int i = 0; i < 3; i++) { if (X[i] & Q) X[0] ^= Pm; else { unsigned t = (X[0] ^ X[i]) & Pm; X[0] ^= t; X[i] ^= t; } } }
    X[1] ^= X[0]; X[2] ^= X[1]; unsigned t = 0; for (unsigned Q = M; Q > 1; Q >>= 1) if (X[2] & Q) t ^= Q - 1; X[0] ^= t; X[1] ^= t; X[2] ^= t;
    return (spread6(X[0]) << 2) | (spread6(X[1]) << 1) | spread6(X[2]);
}
static unsigned *keys; static int cmpk(const void *a, const void *b) { unsigned ka = keys[*(int *)a], kb = keys[*(int *)b]; return ka < kb ? -1 : ka > kb ? 1 : (*(int *)a - *(int *)b); }

int main(int argc, char **argv)
{
    FMAX = argc > 1 ? atoi(argv[1]) : 0; CMAX = argc > 2 ? atoi(argv[2]) : 11; int groups_to_run = argc > 3 ? atoi(argv[3]) : 60;
    FILE *f = fopen("/tmp/exp/tree.bin", "rb"); fseek(f, 0, SEEK_END); N = ftell(f) / 32; rewind(f); T = malloc(N * 32); if (fread(T, 32, N, f) != (size_t)N) return 1; fclose(f);
    f = fopen("/tmp/exp/part.bin", "rb"); fseek(f, 0, SEEK_END); NP = ftell(f) / 12; rewind(f); P = malloc(NP * 12); if (fread(P, 12, NP, f) != (size_t)NP) return 1; fclose(f);
    f = fopen("/tmp/exp/scan.bin", "rb"); S = malloc(NB * 4); if (fread(S, 4, NB, f) != (size_t)NB) return 1; fclose(f);
    // reach
    double sum = 0; int cnt = 0; for (int j = 0; j < NB; j++) { float r = fabsf(S[j]); if (r < 28.3f) { sum += r; cnt++; } } float reach = cnt ? sum / cnt : 8;
    // lane order
    int ns = NP < 1024 ? NP : 1024; double mx = 0, my = 0, mt = 0; for (int k = 0; k < ns; k++) { mx += P[3 * k]; my += P[3 * k + 1]; mt += P[3 * k + 2]; } mx /= ns; my /= ns; mt /= ns;
    double vx = 0, vy = 0, vt = 0; for (int k = 0; k < ns; k++) { vx += pow(P[3 * k] - mx, 2); vy += pow(P[3 * k + 1] - my, 2); vt += pow(P[3 * k + 2] - mt, 2); }
    float e = fmaxf(0.1f * fmaxf(fmaxf(sqrt(vx / ns), sqrt(vy / ns)), sqrt(vt / ns) * reach), 2.5e-4f);
    keys = malloc(NP * 4); int *order = malloc(NP * 4);
    for (int i = 0; i < NP; i++) { unsigned a = fminf(fmaxf((P[3 * i + 2] - mt) * reach / e + 32, 0), 63), b = fminf(fmaxf((P[3 * i] - mx) / e + 32, 0), 63), c = fminf(fmaxf((P[3 * i + 1] - my) / e + 32, 0), 63); keys[i] = hilbert(a, b, c); order[i] = i; }
    qsort(order, NP, 4, cmpk);
    int G = NP / 64; long q = 0, cand_eval = 0, tail_visits = 0, wave_trips = 0, full_visits = 0, mism = 0, rows = 0, ncand = 0, nleaf = 0, nfork = 0, complete = 0;
    srand(1);
    for (int gi = 0; gi < groups_to_run; gi++) {
        int g = rand() % G;
        float xlo = 1e9, xhi = -1e9, ylo = 1e9, yhi = -1e9, tlo = 1e9, thi = -1e9;
        for (int l = 0; l < 64; l++) { float *p = &P[3 * order[g * 64 + l]]; xlo = fminf(xlo, p[0]); xhi = fmaxf(xhi, p[0]); ylo = fminf(ylo, p[1]); yhi = fmaxf(yhi, p[1]); tlo = fminf(tlo, p[2]); thi = fmaxf(thi, p[2]); }
        for (int j = 0; j < NB; j += 7) {
            float r = S[j]; float tc = 0.5f * (tlo + thi), dth = 0.5f * (thi - tlo);
            float rot = ((-135.0f + (float)j * .25f) * 3.14159265f) / 180.0f + tc; float cx = r * cosf(rot), cy = r * sinf(rot);
            float dd = dth + 1e-6f, eps = 2e-4f + 1e-5f * fabsf(r);
            float hx = (fabsf(cy) + fabsf(r) * dd) * dd + eps, hy = (fabsf(cx) + fabsf(r) * dd) * dd + eps;
            box W = {xlo + cx - hx, xhi + cx + hx, ylo + cy - hy, yhi + cy + hy};
            nC = nL = nF = 0; walk(0, W, 0, 0, INFINITY);
            // final pruning: keep candidate if some compatible leaf has lb <= U_leaf
            int keep[64]; int nk = 0;
            for (int c = 0; c < nC; c++) { int k = 0; for (int l = 0; l < nL; l++) if ((L[l].care & C[c].care) == C[c].care && (L[l].want & C[c].care) == C[c].want && C[c].lb <= L[l].U) k = 1; keep[c] = k; nk += k; }
            rows++; ncand += nk; nleaf += nL; nfork += nF; int allc = 1; for (int l = 0; l < nL; l++) allc &= L[l].resume < 0; complete += allc;
            int maxtail = 0;
            for (int l = 0; l < 64; l++) {
                float *p = &P[3 * order[g * 64 + l]];
                float rr = ((-135.0f + (float)j * .25f) * 3.14159265f) / 180.0f + p[2]; float wx = r * cosf(rr), wy = r * sinf(rr);
                if (!(fabsf(wx) < 20 && fabsf(wy) < 20)) continue;
                wx += p[0]; wy += p[1]; q++;
                if (!(wx >= W.xlo && wx <= W.xhi && wy >= W.ylo && wy <= W.yhi)) { printf("W does not contain a lane end point!\n"); return 2; }
                long fv = 0; int tl; int ref = resume_from(wx, wy, INFINITY, 0, 0, &fv, &tl); full_visits += fv;
                unsigned bits = 0; for (int k = 0; k < nF; k++) { float qa = F[k].axis == 0 ? wx : wy; if (qa < F[k].v) bits |= 1u << k; }
                float sBest = INFINITY; int best = 0;
                for (int c = 0; c < nC; c++) { if (!keep[c]) continue; cand_eval++; if ((bits & C[c].care) != C[c].want) continue; node *n = &T[C[c].idx]; float dx = n->x - wx, dy = n->y - wy, s = dx * dx + dy * dy; if (sqrtf(s) < sqrtf(sBest)) { sBest = s; best = C[c].idx; } }
                int res = -2; for (int k = 0; k < nL; k++) if ((bits & L[k].care) == L[k].want) res = L[k].resume;
                if (res == -2) { printf("no leaf for lane\n"); return 3; }
                long tv = 0; int got = resume_from(wx, wy, sBest, best, res, &tv, &tl); tail_visits += tv; if (tl > maxtail) maxtail = tl;
                mism += got != ref;
            }
            wave_trips += maxtail;
        }
    }
    printf("FMAX %d CMAX %d: rows %ld mismatches %ld | full visits/query %.2f | cand evals/lane-query %.2f kept cands/row %.2f forks/row %.2f leaves/row %.2f complete %.3f | per-lane tail+redescent visits/query %.2f | wave trips/row (max lane) %.2f\n",
           FMAX, CMAX, rows, mism, (double)full_visits / q, (double)cand_eval / q, (double)ncand / rows, (double)nfork / rows, (double)nleaf / rows, (double)complete / rows, (double)tail_visits / q, (double)wave_trips / rows);
    return 0;
}
