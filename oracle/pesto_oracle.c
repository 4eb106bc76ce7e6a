/* pesto_oracle.c - CPU restatement of PeSTo's geometric-transformer forward pass.
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library; the shipped path (pesto_amd/) never does.
 *
 * Parity pinning: the reference ships no tests for this path.  This restatement
 * is pinned against golden vectors captured by importing the reference PyTorch CPU path in the
 * build container (tests/golden/make_golden.py): per-stage vectors and whole-forward vectors with
 * the real i_v4_0 / i_v3_0 / i_v3_1 weights (tests/test_oracle.py).
 *
 * Every function cites the reference lines (relative to /root/reference) it follows.  Arithmetic is
 * float32 throughout, like the reference; loops are written per atom (the reference materialises
 * [N,n,193] tensors instead), parallelised over atoms with OpenMP.
 *
 * Two builds of this one source (oracle/Makefile): libpesto_oracle.so sums every dot product / weighted sum in float32, one term
 * after the other (the port that bench.py times as cpu_baseline); libpesto_oracle_wide.so (-DORACLE_ACC_DOUBLE) keeps float32
 * storage and rounds every result to float32 but ACCUMULATES in double.  Why: on ill-conditioned inputs (random clouds with a
 * 2-atom member, |state| ~ 50 after 16 layers) the sequential float32 sums over 64 - 193 terms are 2 - 3x noisier per layer than
 * the reference's blocked MKL kernels (measured against the reference run in float64: tests/golden/make_fuzz_pins.py,
 * profiles/r04_fuzz_pins.txt) - the wide build is the checker for those inputs; both are pinned against the same goldens.
 */
#include <math.h>
#include <omp.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/pesto_hip.h"
#include "pesto_oracle.h"

#define S 32      /* Ns: state width               (model/config.py:28) */
#define NH 2      /* attention heads                (model/config.py:28) */
#define NK 3      /* key size                       (model/config.py:28) */
#define PH 4      /* pool heads                     (model/config.py:61) */
#define XE (6 * S + 1) /* edge feature width = 193  (src/model_operations.py:45) */
#ifdef ORACLE_ACC_DOUBLE
typedef double acc_t;  /* accumulator of dot products and weighted sums (results are rounded to float32 either way) */
#else
typedef float acc_t;
#endif

/* ------------------------------------------------------------------ weight table */
typedef struct { const float *w, *b; int n_in, n_out; } lin_t;
typedef struct { lin_t l[3]; int depth; } mlp_t;
typedef struct { mlp_t nqm, eqkm, epkm, evm, qpm; lin_t ppm; } layer_t;
struct oracle_model {
    pesto_config cfg;
    float* blob;
    mlp_t em, sam, zdm, dm;
    lin_t zdm_vec;
    layer_t layers[PESTO_MAX_LAYERS];
};

/* cursor over the blob: base may be NULL when only the size is wanted */
typedef struct { const float* base; int64_t off; } cursor_t;
static const float* take(cursor_t* cur, int64_t n) {
    const float* p = cur->base ? cur->base + cur->off : NULL;
    cur->off += n;
    return p;
}

static lin_t take_lin(cursor_t* cur, int n_in, int n_out, int bias) {
    lin_t l; l.n_in = n_in; l.n_out = n_out;
    l.w = take(cur, (int64_t)n_in * n_out);
    l.b = bias ? take(cur, n_out) : NULL;
    return l;
}

/* Sequential(Linear, ELU, Linear, ELU, Linear): keys .0 .2 .4 in state_dict order */
static mlp_t take_mlp(cursor_t* cur, int d0, int d1, int d2, int d3) {
    mlp_t m; m.depth = 3;
    m.l[0] = take_lin(cur, d0, d1, 1);
    m.l[1] = take_lin(cur, d1, d2, 1);
    m.l[2] = take_lin(cur, d2, d3, 1);
    return m;
}

static mlp_t take_mlp1(cursor_t* cur, int d0, int d1) {
    mlp_t m; memset(&m, 0, sizeof m); m.depth = 1;
    m.l[0] = take_lin(cur, d0, d1, 1);
    return m;
}

static int config_ok(const pesto_config* c) {
    if (!c || c->n0 < 1 || c->n0 > 512 || c->n_layers < 1 || c->n_layers > PESTO_MAX_LAYERS) return 0;
    if (c->n_out < 1 || c->n_out > 32) return 0;
    if ((c->em_depth != 1 && c->em_depth != 3) || (c->dm_depth != 1 && c->dm_depth != 3)) return 0;
    for (int l = 0; l < c->n_layers; ++l)
        if (c->nn[l] != 8 && c->nn[l] != 16 && c->nn[l] != 32 && c->nn[l] != 64) return 0;
    return 1;
}

/* blob order = state_dict order minus m_nn/sdk  (model/model.py:10-30; model_operations.py:27-85,172-195) */
static int64_t bind(struct oracle_model* m, const float* blob) {
    const pesto_config* c = &m->cfg;
    cursor_t cursor = { blob, 0 };
    cursor_t* cur_p = &cursor;
    m->em = c->em_depth == 3 ? take_mlp(cur_p, c->n0, S, S, S) : take_mlp1(cur_p, c->n0, S);
    for (int l = 0; l < c->n_layers; ++l) {
        layer_t* L = &m->layers[l];
        L->nqm = take_mlp(cur_p, 2 * S, S, S, 2 * NK * NH);
        L->eqkm = take_mlp(cur_p, XE, S, S, NK);
        L->epkm = take_mlp(cur_p, XE, S, S, 3 * NK);
        L->evm = take_mlp(cur_p, XE, 2 * S, 2 * S, 2 * S);
        L->qpm = take_mlp(cur_p, NH * S, S, S, S);
        L->ppm = take_lin(cur_p, NH * S, S, 0);
    }
    m->sam = take_mlp(cur_p, 2 * S, S, S, 2 * PH);
    m->zdm = take_mlp(cur_p, PH * S, S, S, S);
    m->zdm_vec = take_lin(cur_p, PH * S, S, 0);
    m->dm = c->dm_depth == 3 ? take_mlp(cur_p, 2 * S, S, S, c->n_out) : take_mlp1(cur_p, 2 * S, c->n_out);
    return cursor.off;
}

int oracle_blob_size(const pesto_config* cfg, int64_t* n) {
    if (!config_ok(cfg) || !n) return -1;
    struct oracle_model tmp; tmp.cfg = *cfg;
    *n = bind(&tmp, NULL);
    return 0;
}

int oracle_create(const pesto_config* cfg, const float* weights, int64_t n_weights, struct oracle_model** out) {
    int64_t need;
    if (!out || oracle_blob_size(cfg, &need) != 0 || need != n_weights) return -1;
    struct oracle_model* m = (struct oracle_model*)calloc(1, sizeof *m);
    if (!m) return -3;
    m->cfg = *cfg;
    m->blob = (float*)malloc(sizeof(float) * (size_t)need);
    if (!m->blob) { free(m); return -3; }
    memcpy(m->blob, weights, sizeof(float) * (size_t)need);
    bind(m, m->blob);
    *out = m;
    return 0;
}

void oracle_destroy(struct oracle_model* m) { if (m) { free(m->blob); free(m); } }
/* OpenMP team size of the following calls (bench.py times the port at the thread count its rho was measured at); n < 1: all cores */
void oracle_set_threads(int n) { omp_set_num_threads(n > 0 ? n : omp_get_num_procs()); }

/* ------------------------------------------------------------------ small dense pieces */
/* torch.nn.ELU(alpha=1): x > 0 ? x : exp(x) - 1 */
static inline float elu(float x) { return x > 0.0f ? x : expf(x) - 1.0f; }

/* torch.nn.Linear: y = x W^T + b, W [out,in] row-major */
static void linear(const lin_t* l, const float* x, float* y) {
    for (int o = 0; o < l->n_out; ++o) {
        const float* w = l->w + (size_t)o * l->n_in;
        acc_t acc = 0;
        for (int i = 0; i < l->n_in; ++i) acc += (acc_t)w[i] * x[i];
        y[o] = (float)(acc + (l->b ? l->b[o] : 0.0f));
    }
}

static void mlp(const mlp_t* m, const float* x, float* y) {
    float h0[128], h1[128];
    if (m->depth == 1) { linear(&m->l[0], x, y); return; }
    linear(&m->l[0], x, h0);
    for (int i = 0; i < m->l[0].n_out; ++i) h0[i] = elu(h0[i]);
    linear(&m->l[1], h0, h1);
    for (int i = 0; i < m->l[1].n_out; ++i) h1[i] = elu(h1[i]);
    linear(&m->l[2], h1, y);
}

/* rows x Linear, written so gcc vectorises over rows' output columns: Y[r][o] = b[o] + sum_i X[r][i] W[o][i].
 * Wt is the TRANSPOSED weight [in][out] (built per call site, once per layer). */
static void linear_rows(int rows, int n_in, int n_out, const float* X, int ldx, const float* Wt, const float* b,
                        float* Y, int ldy, int apply_elu) {
    for (int r0 = 0; r0 < rows; r0 += 4) {
        int rb = rows - r0 < 4 ? rows - r0 : 4;
        acc_t acc[4][64];
        for (int o0 = 0; o0 < n_out; o0 += 64) {
            int ob = n_out - o0 < 64 ? n_out - o0 : 64;
            for (int r = 0; r < rb; ++r) for (int o = 0; o < ob; ++o) acc[r][o] = 0;
            for (int i = 0; i < n_in; ++i) {
                const float* w = Wt + (size_t)i * n_out + o0;
                for (int r = 0; r < rb; ++r) {
                    acc_t x = X[(size_t)(r0 + r) * ldx + i];
                    for (int o = 0; o < ob; ++o) acc[r][o] += x * w[o];
                }
            }
            for (int r = 0; r < rb; ++r)
                for (int o = 0; o < ob; ++o) {
                    float v = (float)(acc[r][o] + (b ? b[o0 + o] : 0.0f));
                    Y[(size_t)(r0 + r) * ldy + o0 + o] = apply_elu ? elu(v) : v;
                }
        }
    }
}

static float* transpose(const lin_t* l) {
    float* t = (float*)malloc(sizeof(float) * (size_t)l->n_in * l->n_out);
    for (int o = 0; o < l->n_out; ++o)
        for (int i = 0; i < l->n_in; ++i) t[(size_t)i * l->n_out + o] = l->w[(size_t)o * l->n_in + i];
    return t;
}

typedef struct { float* t[3]; } mlp_tr;
static mlp_tr transpose_mlp(const mlp_t* m) { mlp_tr r; for (int i = 0; i < 3; ++i) r.t[i] = transpose(&m->l[i]); return r; }
static void free_mlp_tr(mlp_tr* r) { for (int i = 0; i < 3; ++i) free(r->t[i]); }

/* 3-layer MLP over `rows` rows (ELU between), out ld = last n_out */
static void mlp_rows(const mlp_t* m, const mlp_tr* tr, int rows, const float* X, int ldx, float* H0, float* H1, float* Y) {
    linear_rows(rows, m->l[0].n_in, m->l[0].n_out, X, ldx, tr->t[0], m->l[0].b, H0, m->l[0].n_out, 1);
    linear_rows(rows, m->l[1].n_in, m->l[1].n_out, H0, m->l[0].n_out, tr->t[1], m->l[1].b, H1, m->l[1].n_out, 1);
    linear_rows(rows, m->l[2].n_in, m->l[2].n_out, H1, m->l[1].n_out, tr->t[2], m->l[2].b, Y, m->l[2].n_out, 0);
}

static void softmax_inplace(float* a, int n) {
    float mx = a[0];
    for (int i = 1; i < n; ++i) mx = a[i] > mx ? a[i] : mx;
    acc_t sum = 0;
    for (int i = 0; i < n; ++i) { a[i] = expf(a[i] - mx); sum += a[i]; }
    for (int i = 0; i < n; ++i) a[i] = (float)(a[i] / sum);
}

/* ------------------------------------------------------------------ stage: embedding
 * model/model.py:34  q = self.em.forward(q0) */
int oracle_embed(const struct oracle_model* m, int64_t N, const float* q0, float* q_out) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < N; ++i) mlp(&m->em, q0 + i * m->cfg.n0, q_out + i * S);
    return 0;
}

/* ------------------------------------------------------------------ stage: geometry + sink
 * src/model_operations.py:6-22 unpack_state_features.  ids: 1-based, 0 = padding; X is indexed with
 * ids-1, and -1 wraps to the LAST atom of the batch (python negative indexing, :8).
 * Outputs have the sink row 0 prepended (:17-20): ids_s [N+1,k], D [N+1,k], R [N+1,k,3]. */
int oracle_unpack(int64_t N, int k, const float* X, const int32_t* ids, int32_t* ids_s, float* D, float* R) {
    for (int c = 0; c < k; ++c) { ids_s[c] = 0; D[c] = 0.0f; R[3 * c] = R[3 * c + 1] = R[3 * c + 2] = 0.0f; }
    float dmax = 0.0f;
    int have = 0;
    for (int64_t i = 0; i < N; ++i)
        for (int c = 0; c < k; ++c) {
            int64_t j = (int64_t)ids[i * k + c] - 1;
            if (j < 0) j += N;
            float rx = X[3 * j] - X[3 * i], ry = X[3 * j + 1] - X[3 * i + 1], rz = X[3 * j + 2] - X[3 * i + 2];
            float d = sqrtf(fmaf(rz, rz, fmaf(ry, ry, rx * rx)));      /* :10 (torch.norm over xyz: an FMA chain in the reference's build) */
            size_t e = (size_t)(i + 1) * k + c;
            ids_s[e] = ids[i * k + c];
            D[e] = d; R[3 * e] = rx; R[3 * e + 1] = ry; R[3 * e + 2] = rz;
            if (!have || d > dmax) { dmax = d; have = 1; }
        }
    for (int64_t i = 0; i < N; ++i)
        for (int c = 0; c < k; ++c) {
            size_t e = (size_t)(i + 1) * k + c;
            float d = D[e] + dmax * (D[e] < 1e-2f ? 1.0f : 0.0f);       /* :12 global max over the batch */
            D[e] = d;
            R[3 * e] /= d; R[3 * e + 1] /= d; R[3 * e + 2] /= d;        /* :14 */
        }
    return 0;
}

/* ------------------------------------------------------------------ stage: one state-update layer
 * src/model_operations.py:225-242 (StateUpdateLayer.forward) calling :87-154 (StateUpdate.forward).
 * q [N1,32], p [N1,3,32] (N1 = N+1, row 0 = sink) are updated in place; every row reads the OLD state. */
static _Thread_local float* tl_scratch = NULL;      /* per-thread edge scratch of oracle_layer (kept for the thread's lifetime) */

int oracle_layer(const struct oracle_model* m, int layer, int64_t N1, int k, const int32_t* ids_s, const float* D,
                 const float* R, float* q, float* p) {
    if (layer < 0 || layer >= m->cfg.n_layers) return -1;
    const layer_t* L = &m->layers[layer];
    const int n = m->cfg.nn[layer];                                    /* :230 ids_topk[:, :nn] */
    if (n > k) return -1;
    const float sdk = sqrtf((float)NK);                                /* :85 */
    float* q_old = (float*)malloc(sizeof(float) * (size_t)N1 * S);
    float* p_old = (float*)malloc(sizeof(float) * (size_t)N1 * 3 * S);
    float* pn = (float*)malloc(sizeof(float) * (size_t)N1 * S);        /* ||p|| over xyz, per atom */
    mlp_tr t_eq = transpose_mlp(&L->eqkm), t_ep = transpose_mlp(&L->epkm), t_ev = transpose_mlp(&L->evm);

#pragma omp parallel
    {
        /* per-thread scratch, allocated ONCE per thread (the first layer a thread works on) and kept: 32 layers x 256 threads x six
         * malloc / free pairs inside the parallel region serialised on the allocator */
        if (!tl_scratch) tl_scratch = (float*)malloc(sizeof(float) * (64 * XE + 2 * 64 * 64 + 64 * NK + 64 * 3 * NK + 64 * 2 * S));
        float* Xe = tl_scratch;
        float* H0 = Xe + 64 * XE;
        float* H1 = H0 + 64 * 64;
        float* Kq = H1 + 64 * 64;
        float* Kp = Kq + 64 * NK;
        float* V = Kp + 64 * 3 * NK;
        /* the old state and ||p||: copied / computed by the team (was a serial prologue per layer) */
#pragma omp for schedule(static)
        for (int64_t i = 0; i < N1; ++i) {
            memcpy(q_old + i * S, q + i * S, sizeof(float) * S);
            memcpy(p_old + i * 3 * S, p + i * 3 * S, sizeof(float) * 3 * S);
            const float* pi = p + i * 3 * S;
            for (int s = 0; s < S; ++s)
                pn[i * S + s] = sqrtf(pi[s] * pi[s] + pi[S + s] * pi[S + s] + pi[2 * S + s] * pi[2 * S + s]);   /* :105, :113 */
        }
#pragma omp for schedule(static, 2)   /* equal-cost atoms: no shared work counter (dynamic chunks of 16 left 68 of 256 threads idle) */
        for (int64_t i = 0; i < N1; ++i) {
            const float* qi = q_old + i * S;
            const float* pi = p_old + i * 3 * S;
            float Xn[2 * S];                                           /* :103-106 */
            for (int s = 0; s < S; ++s) { Xn[s] = qi[s]; Xn[S + s] = pn[i * S + s]; }
            for (int c = 0; c < n; ++c) {                              /* :109-116 */
                int64_t j = ids_s[i * k + c];
                const float* r = R + ((size_t)i * k + c) * 3;
                const float* qj = q_old + j * S;
                const float* pj = p_old + j * 3 * S;
                float* x = Xe + (size_t)c * XE;
                x[0] = D[i * k + c];
                for (int s = 0; s < 2 * S; ++s) x[1 + s] = Xn[s];
                for (int s = 0; s < S; ++s) {
                    x[1 + 2 * S + s] = qj[s];
                    x[1 + 3 * S + s] = pn[j * S + s];
                    x[1 + 4 * S + s] = pi[s] * r[0] + pi[S + s] * r[1] + pi[2 * S + s] * r[2];
                    x[1 + 5 * S + s] = pj[s] * r[0] + pj[S + s] * r[1] + pj[2 * S + s] * r[2];
                }
            }
            float Q[2 * NH * NK];                                      /* :119 view [2, Nh, Nk] */
            mlp(&L->nqm, Xn, Q);
            mlp_rows(&L->eqkm, &t_eq, n, Xe, XE, H0, H1, Kq);          /* :122 */
            mlp_rows(&L->epkm, &t_ep, n, Xe, XE, H0, H1, Kp);          /* :125 raw [n, 9] */
            mlp_rows(&L->evm, &t_ev, n, Xe, XE, H0, H1, V);            /* :128 [n, 64]: V0 = [:32], V1 = [32:] */

            float Zq[NH * S], Zp[3][NH * S];
            for (int h = 0; h < NH; ++h) {
                float Mq[64] = {0}, Mp[3 * 64] = {0};
                for (int c = 0; c < n; ++c) {                          /* :139 */
                    acc_t a = 0;
                    for (int kk = 0; kk < NK; ++kk) a += (acc_t)Q[h * NK + kk] * Kq[c * NK + kk];
                    Mq[c] = (float)a / sdk;
                }
                softmax_inplace(Mq, n);
                /* :125 chunk-major: slot t*n + c holds epkm output columns [t*Nk, (t+1)*Nk) of edge c */
                for (int t = 0; t < 3; ++t)
                    for (int c = 0; c < n; ++c) {                      /* :140 */
                        acc_t a = 0;
                        for (int kk = 0; kk < NK; ++kk) a += (acc_t)Q[NH * NK + h * NK + kk] * Kp[c * 3 * NK + t * NK + kk];
                        Mp[t * n + c] = (float)a / sdk;
                    }
                softmax_inplace(Mp, 3 * n);
                for (int s = 0; s < S; ++s) {                          /* :143 Zq index h*S+s */
                    acc_t a = 0;
                    for (int c = 0; c < n; ++c) a += (acc_t)Mq[c] * V[c * 2 * S + s];
                    Zq[h * S + s] = (float)a;
                }
                for (int x = 0; x < 3; ++x)                            /* :131-136, :144 */
                    for (int s = 0; s < S; ++s) {
                        acc_t a = 0;
                        for (int c = 0; c < n; ++c) a += (acc_t)Mp[c] * (V[c * 2 * S + S + s] * R[((size_t)i * k + c) * 3 + x]);
                        for (int c = 0; c < n; ++c) a += (acc_t)Mp[n + c] * pi[x * S + s];
                        for (int c = 0; c < n; ++c) a += (acc_t)Mp[2 * n + c] * p_old[(size_t)ids_s[i * k + c] * 3 * S + x * S + s];
                        Zp[x][h * S + s] = (float)a;
                    }
            }
            float qh[S], ph[S];
            mlp(&L->qpm, Zq, qh);                                      /* :147 */
            for (int s = 0; s < S; ++s) q[i * S + s] = qi[s] + qh[s];  /* :151 */
            for (int x = 0; x < 3; ++x) {
                linear(&L->ppm, Zp[x], ph);                            /* :148 (no bias) */
                for (int s = 0; s < S; ++s) p[i * 3 * S + x * S + s] = pi[x * S + s] + ph[s];   /* :152 */
            }
        }
    }
    for (int s = 0; s < S; ++s) q[s] = q[s] * 0.0f;                    /* :239 sink */
    for (int s = 0; s < 3 * S; ++s) p[s] = p[s] * 0.0f;                /* :240 */
    free_mlp_tr(&t_eq); free_mlp_tr(&t_ep); free_mlp_tr(&t_ev);
    free(q_old); free(p_old); free(pn);
    return 0;
}

/* ------------------------------------------------------------------ stage: residue pool + decoder
 * src/model_operations.py:197-213 (StatePoolLayer.forward) and model/model.py:49-50.
 * The reference softmaxes over ALL atoms with an additive mask F = (1-M+1e-6)/(M-1e-6) (:199): members get
 * +1.000001e-6, non-members -1000001 whose exp underflows to exactly 0 in float32, so the result equals a
 * softmax restricted to the residue's atoms.  q [N,32], p [N,3,32] WITHOUT sink row. */
int oracle_pool(const struct oracle_model* m, int64_t N, int64_t Rr, const float* q, const float* p,
                const int32_t* res_of_atom, float* qr, float* pr, float* z) {
    float* a = (float*)malloc(sizeof(float) * (size_t)N * 2 * PH);
    const float f_member = (1.0f - 1.0f + 1e-6f) / (1.0f - 1e-6f);
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < N; ++i) {
        float zin[2 * S];                                              /* :202 */
        for (int s = 0; s < S; ++s) {
            const float* pi = p + i * 3 * S;
            zin[s] = q[i * S + s];
            zin[S + s] = sqrtf(pi[s] * pi[s] + pi[S + s] * pi[S + s] + pi[2 * S + s] * pi[2 * S + s]);
        }
        mlp(&m->sam, zin, a + i * 2 * PH);
        for (int c = 0; c < 2 * PH; ++c) a[i * 2 * PH + c] += f_member;   /* :205 "+ F" for members */
    }
    int status = 0;
#pragma omp parallel for schedule(dynamic, 8)
    for (int64_t r = 0; r < Rr; ++r) {
        float mx[2 * PH];
        acc_t den[2 * PH], qh_a[PH * S], ph_a[3][PH * S];
        float qh[PH * S], ph[3][PH * S];
        int cnt = 0;
        for (int c = 0; c < 2 * PH; ++c) { mx[c] = -INFINITY; den[c] = 0; }
        for (int64_t i = 0; i < N; ++i)
            if (res_of_atom[i] == r) { ++cnt; for (int c = 0; c < 2 * PH; ++c) mx[c] = fmaxf(mx[c], a[i * 2 * PH + c]); }
        if (cnt == 0) {
#pragma omp atomic write
            status = -1;
            continue;
        }
        for (int64_t i = 0; i < N; ++i)
            if (res_of_atom[i] == r) for (int c = 0; c < 2 * PH; ++c) den[c] += expf(a[i * 2 * PH + c] - mx[c]);
        memset(qh_a, 0, sizeof qh_a); memset(ph_a, 0, sizeof ph_a);
        for (int64_t i = 0; i < N; ++i) {
            if (res_of_atom[i] != r) continue;
            for (int h = 0; h < PH; ++h) {
                /* :205 view(...,-1,2): channel 2h = scalar head h, 2h+1 = vector head h */
                float w0 = (float)(expf(a[i * 2 * PH + 2 * h] - mx[2 * h]) / den[2 * h]);
                float w1 = (float)(expf(a[i * 2 * PH + 2 * h + 1] - mx[2 * h + 1]) / den[2 * h + 1]);
                for (int s = 0; s < S; ++s) {
                    qh_a[s * PH + h] += (acc_t)q[i * S + s] * w0;                                   /* :206, flatten s*Nh+h (:210) */
                    for (int x = 0; x < 3; ++x) ph_a[x][s * PH + h] += (acc_t)p[i * 3 * S + x * S + s] * w1;   /* :207, :211 */
                }
            }
        }
        for (int s = 0; s < PH * S; ++s) { qh[s] = (float)qh_a[s]; for (int x = 0; x < 3; ++x) ph[x][s] = (float)ph_a[x][s]; }
        float zr[2 * S];
        mlp(&m->zdm, qh, qr + r * S);                                  /* :210 */
        for (int x = 0; x < 3; ++x) linear(&m->zdm_vec, ph[x], pr + r * 3 * S + x * S);   /* :211 */
        for (int s = 0; s < S; ++s) {                                  /* model/model.py:49 */
            const float* v = pr + r * 3 * S;
            zr[s] = qr[r * S + s];
            zr[S + s] = sqrtf(v[s] * v[s] + v[S + s] * v[S + s] + v[2 * S + s] * v[2 * S + s]);
        }
        mlp(&m->dm, zr, z + r * m->cfg.n_out);                         /* model/model.py:50 */
    }
    free(a);
    return status;
}

/* ------------------------------------------------------------------ whole forward
 * model/model.py:32-52 */
int oracle_forward(const struct oracle_model* m, int64_t N, int64_t Rr, int k, const float* X, const int32_t* ids,
                   const float* q0, const int32_t* res_of_atom, float* z, float* q_state, float* p_state, int stop_after) {
    const int64_t N1 = N + 1;
    float* q = (float*)calloc((size_t)N1 * S, sizeof(float));
    float* p = (float*)calloc((size_t)N1 * 3 * S, sizeof(float));      /* :37 p0 = zeros */
    int32_t* ids_s = (int32_t*)malloc(sizeof(int32_t) * (size_t)N1 * k);
    float* D = (float*)malloc(sizeof(float) * (size_t)N1 * k);
    float* R = (float*)malloc(sizeof(float) * (size_t)N1 * k * 3);
    float* qr = (float*)malloc(sizeof(float) * (size_t)Rr * S);
    float* pr = (float*)malloc(sizeof(float) * (size_t)Rr * 3 * S);
    int rc = 0;
    oracle_embed(m, N, q0, q + S);                                     /* :34, sink row 0 stays 0 (:17) */
    oracle_unpack(N, k, X, ids, ids_s, D, R);                          /* :40 */
    int nl = m->cfg.n_layers;
    if (stop_after >= 0 && stop_after < nl) nl = stop_after;
    for (int l = 0; l < nl && rc == 0; ++l) rc = oracle_layer(m, l, N1, k, ids_s, D, R, q, p);   /* :43 */
    if (q_state) memcpy(q_state, q, sizeof(float) * (size_t)N1 * S);
    if (p_state) memcpy(p_state, p, sizeof(float) * (size_t)N1 * 3 * S);
    if (rc == 0 && z) rc = oracle_pool(m, N, Rr, q + S, p + 3 * S, res_of_atom, qr, pr, z);   /* :46-50 */
    free(q); free(p); free(ids_s); free(D); free(R); free(qr); free(pr);
    return rc;
}
