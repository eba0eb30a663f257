/*
 * hhe_oracle.c -- CPU ORACLE (test infrastructure, NOT the product; see hhe_oracle.h).
 *
 * Plain C restatement of the reference's PASTA-3 -> BFV transciphering path and
 * of the SEAL 4.0.0 primitives under it.  Exact arithmetic everywhere
 * (unsigned __int128); every function cites the reference file:line it follows.
 * "SURVEY A.x" = /root/repo/SURVEY.md Appendix A (specs verified against
 * libseal-4.0.a by the survey; that binary is never linked here).
 */
#include "hhe_oracle.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef unsigned __int128 u128;
typedef uint64_t u64;

/* ------------------------------------------------------------------ */
/* modular helpers                                                     */
/* ------------------------------------------------------------------ */
typedef struct {
    u64 q;
    u64 r_lo, r_hi;        /* floor(2^128/q), as in seal/modulus.h const_ratio */
    u64 *w, *ws, *iw, *iws; /* NTT tables (NULL when not an NTT modulus) */
    u64 ninv, ninvs;
} modtab;

static inline u64 addmod(u64 a, u64 b, u64 q) { u64 s = a + b; return s >= q ? s - q : s; }
static inline u64 submod(u64 a, u64 b, u64 q) { return a >= b ? a - b : a + q - b; }
static inline u64 negmod(u64 a, u64 q) { return a ? q - a : 0; }

/* barrett_reduce_128 (seal/util/uintarithsmallmod.h:166-210) */
static inline u64 red128(u128 x, const modtab *m)
{
    u64 x0 = (u64)x, x1 = (u64)(x >> 64);
    u64 carry = (u64)(((u128)x0 * m->r_lo) >> 64);
    u128 t2 = (u128)x0 * m->r_hi;
    u128 s1 = (u128)(u64)t2 + carry;
    u64 t3 = (u64)(t2 >> 64) + (u64)(s1 >> 64);
    t2 = (u128)x1 * m->r_lo;
    u128 s2 = (u128)(u64)s1 + (u64)t2;
    carry = (u64)(t2 >> 64) + (u64)(s2 >> 64);
    u64 quo = x1 * m->r_hi + t3 + carry;
    u64 r = x0 - quo * m->q;
    while (r >= m->q) r -= m->q;
    return r;
}
static inline u64 mulmod(u64 a, u64 b, const modtab *m) { return red128((u128)a * b, m); }
static inline u64 red64(u64 a, const modtab *m) { return red128((u128)a, m); }

static u64 mulmod_slow(u64 a, u64 b, u64 q) { return (u64)(((u128)a * b) % q); }
static u64 powmod(u64 a, u64 e, u64 q)
{
    u64 r = 1 % q;
    a %= q;
    while (e) {
        if (e & 1) r = mulmod_slow(r, a, q);
        a = mulmod_slow(a, a, q);
        e >>= 1;
    }
    return r;
}
/* modular inverse for any odd/even modulus with gcd(a,q)=1 (ext. Euclid) */
static u64 invmod(u64 a, u64 q)
{
    __int128 t0 = 0, t1 = 1, r0 = q, r1 = a % q;
    while (r1) {
        __int128 qq = r0 / r1, tmp;
        tmp = t0 - qq * t1; t0 = t1; t1 = tmp;
        tmp = r0 - qq * r1; r0 = r1; r1 = tmp;
    }
    if (t0 < 0) t0 += q;
    return (u64)t0;
}
static void modtab_init(modtab *m, u64 q)
{
    memset(m, 0, sizeof(*m));
    m->q = q;
    /* floor(2^128 / q) */
    u128 hi = (((u128)1) << 64) / q;              /* floor(2^64/q)            */
    u128 rem = (((u128)1) << 64) % q;             /* 2^64 mod q               */
    /* 2^128/q = hi*2^64 + floor(rem*2^64/q) */
    u128 lo = (rem << 64) / q;
    m->r_hi = (u64)hi;
    m->r_lo = (u64)lo;
}
static inline u64 shoup_pre(u64 w, u64 q) { return (u64)((((u128)w) << 64) / q); }
/* x*w mod q in [0,2q) (MultiplyUIntModOperand, seal/util/uintarithsmallmod.h:270-326) */
static inline u64 shoup_lazy(u64 x, u64 w, u64 ws, u64 q)
{
    u64 hi = (u64)(((u128)x * ws) >> 64);
    return x * w - hi * q;
}

static u64 bitrev(u64 v, int bits)
{
    u64 r = 0;
    for (int i = 0; i < bits; i++) { r = (r << 1) | (v & 1); v >>= 1; }
    return r;
}

/* ------------------------------------------------------------------ */
/* number theory  (seal/util/numth.h)                                  */
/* ------------------------------------------------------------------ */
int orc_is_prime(u64 n)
{
    if (n < 2) return 0;
    static const u64 small[] = {2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37};
    for (size_t i = 0; i < 12; i++) {
        if (n == small[i]) return 1;
        if (n % small[i] == 0) return 0;
    }
    u64 d = n - 1; int r = 0;
    while (!(d & 1)) { d >>= 1; r++; }
    for (size_t i = 0; i < 12; i++) { /* deterministic Miller-Rabin for 64-bit */
        u64 x = powmod(small[i], d, n);
        if (x == 1 || x == n - 1) continue;
        int comp = 1;
        for (int k = 1; k < r; k++) {
            x = mulmod_slow(x, x, n);
            if (x == n - 1) { comp = 0; break; }
        }
        if (comp) return 0;
    }
    return 1;
}

/* get_primes (seal/util/numth.h:138-139): start at the largest value of the bit
 * size congruent to 1 mod factor and walk down by factor. */
int orc_get_primes(u64 factor, int bit_size, size_t count, u64 *out)
{
    u64 value = ((((u64)1) << bit_size) - 1) / factor * factor + 1;
    u64 lower = ((u64)1) << (bit_size - 1);
    size_t found = 0;
    while (found < count && value > lower) {
        if (orc_is_prime(value)) out[found++] = value;
        value -= factor;
    }
    return found == count ? 0 : -1;
}

/* CoeffModulus::Create(N, bit_sizes) (seal/modulus.h): per distinct size the
 * primes are handed out from the back of get_primes' list (smallest first). */
int orc_coeff_modulus_create(size_t n, const int *bit_sizes, size_t count, u64 *out)
{
    int cnt[64] = {0};
    u64 *tab[64] = {0};
    int used[64] = {0};
    for (size_t i = 0; i < count; i++) cnt[bit_sizes[i]]++;
    for (int b = 2; b < 64; b++)
        if (cnt[b]) {
            tab[b] = (u64 *)malloc(sizeof(u64) * cnt[b]);
            if (orc_get_primes(2 * (u64)n, b, cnt[b], tab[b])) return -1;
        }
    for (size_t i = 0; i < count; i++) {
        int b = bit_sizes[i];
        out[i] = tab[b][cnt[b] - 1 - used[b]];
        used[b]++;
    }
    for (int b = 2; b < 64; b++) free(tab[b]);
    return 0;
}

/* try_minimal_primitive_root (seal/util/numth.h:155-157): the smallest
 * primitive degree-th root of unity (degree a power of two). SURVEY A.1. */
u64 orc_minimal_primitive_root(u64 degree, u64 q)
{
    u64 e = (q - 1) / degree, root = 0;
    for (u64 g = 2;; g++) {
        root = powmod(g, e, q);
        if (powmod(root, degree >> 1, q) == q - 1) break;
    }
    u64 gsq = mulmod_slow(root, root, q), cur = root, best = root;
    for (u64 i = 0; i < (degree >> 1); i++) {
        if (cur < best) best = cur;
        cur = mulmod_slow(cur, gsq, q);
    }
    return best;
}

/* util::naf (seal/util/numth.h:22-42) */
int orc_naf(int value, int *out)
{
    int n = 0, sign = value < 0;
    value = abs(value);
    for (int i = 0; value; i++) {
        int zi = (value & 1) ? 2 - (value & 3) : 0;
        value = (value - zi) >> 1;
        if (zi) out[n++] = (sign ? -zi : zi) * (1 << i);
    }
    return n;
}

/* ------------------------------------------------------------------ */
/* context                                                             */
/* ------------------------------------------------------------------ */
struct orc_ctx {
    int logn; size_t n; int K, L;
    u64 t;
    modtab m[ORC_MAXK];       /* coeff primes, last = special (seal/context.h) */
    modtab bsk[ORC_MAXK + 1]; /* B primes [0..L-1], m_sk at [L] (seal/util/rns.h:324-399) */
    modtab pt;                /* plain modulus */
    modtab mgamma;
    uint32_t *slot_map;
    u64 root[ORC_MAXK];
    u64 delta[ORC_MAXK], q_mod_t, up_thr, up_inc[ORC_MAXK];
    u64 qsp, qsp_half, qsp_inv[ORC_MAXK], qsp_half_mod[ORC_MAXK];
    u64 gamma, msk;
    u64 inv_punct_q[ORC_MAXK];
    u64 punct_q_bsk[ORC_MAXK][ORC_MAXK + 1];
    u64 punct_q_mt[ORC_MAXK];
    u64 neg_inv_q_mt;
    u64 q_mod_bsk[ORC_MAXK + 1], inv_mt_bsk[ORC_MAXK + 1], inv_q_bsk[ORC_MAXK + 1];
    u64 inv_punct_B[ORC_MAXK];
    u64 punct_B_q[ORC_MAXK][ORC_MAXK];
    u64 punct_B_msk[ORC_MAXK];
    u64 inv_B_msk, B_mod_q[ORC_MAXK];
    /* decrypt_scale_and_round constants (seal/util/rns.h) */
    u64 tg_mod_q[ORC_MAXK], punct_q_t[ORC_MAXK], punct_q_g[ORC_MAXK];
    u64 neg_inv_q_t, neg_inv_q_g, inv_g_t;
};

/* NTTTables (seal/util/ntt.h:69-183): root_powers[bitrev(k)] = psi^k, psi minimal */
static void ntt_tables_init(modtab *m, int logn, u64 *root_out)
{
    size_t n = (size_t)1 << logn;
    u64 q = m->q;
    u64 psi = orc_minimal_primitive_root(2 * n, q);
    if (root_out) *root_out = psi;
    m->w = (u64 *)malloc(8 * n); m->ws = (u64 *)malloc(8 * n);
    m->iw = (u64 *)malloc(8 * n); m->iws = (u64 *)malloc(8 * n);
    u64 ipsi = invmod(psi, q);
    u64 p = 1, ip = 1;
    for (size_t k = 0; k < n; k++) {
        size_t r = bitrev(k, logn);
        m->w[r] = p; m->ws[r] = shoup_pre(p, q);
        m->iw[r] = ip; m->iws[r] = shoup_pre(ip, q);
        p = mulmod_slow(p, psi, q);
        ip = mulmod_slow(ip, ipsi, q);
    }
    m->ninv = invmod(n % q, q);
    m->ninvs = shoup_pre(m->ninv, q);
}
static void modtab_free(modtab *m) { free(m->w); free(m->ws); free(m->iw); free(m->iws); }

static u64 prod_mod_except(const u64 *v, int cnt, int except, u64 p)
{
    u64 r = 1 % p;
    for (int i = 0; i < cnt; i++)
        if (i != except) r = mulmod_slow(r, v[i] % p, p);
    return r;
}

orc_ctx *orc_ctx_create(int logn, int K, const u64 *q, u64 t)
{
    if (K < 2 || K > ORC_MAXK - 2 || logn < 2 || logn > 17) return NULL;
    orc_ctx *c = (orc_ctx *)calloc(1, sizeof(orc_ctx));
    c->logn = logn; c->n = (size_t)1 << logn; c->K = K; c->L = K - 1; c->t = t;
    size_t n = c->n; int L = c->L;
    for (int i = 0; i < K; i++) {
        modtab_init(&c->m[i], q[i]);
        ntt_tables_init(&c->m[i], logn, &c->root[i]);
    }
    modtab_init(&c->pt, t);
    if ((t - 1) % (2 * n) == 0 && orc_is_prime(t)) ntt_tables_init(&c->pt, logn, NULL);
    /* matrix_reps_index_map */
    if (c->pt.w) {
        c->slot_map = (uint32_t *)malloc(4 * n);
        u64 mm = 2 * n, pos = 1;
        for (size_t i = 0; i < n / 2; i++) {
            c->slot_map[i] = (uint32_t)bitrev((pos - 1) >> 1, logn);
            c->slot_map[n / 2 + i] = (uint32_t)bitrev((mm - pos - 1) >> 1, logn);
            pos = pos * 3 % mm;
        }
    }
    u64 dq[ORC_MAXK];
    for (int i = 0; i < L; i++) dq[i] = q[i];
    /* scaling variant (seal/context.h:350-401) */
    c->q_mod_t = prod_mod_except(dq, L, -1, t);
    c->up_thr = (t + 1) >> 1;
    for (int j = 0; j < L; j++) {
        c->up_inc[j] = q[j] - t;
        /* floor(Q/t) = (Q - Q mod t)/t  ==  -(Q mod t) * t^-1  (mod q_j) */
        u64 tinv = invmod(t % q[j], q[j]);
        c->delta[j] = negmod(mulmod_slow(c->q_mod_t % q[j], tinv, q[j]), q[j]);
    }
    /* key-switch (seal/util/rns.h:383-384) */
    c->qsp = q[K - 1]; c->qsp_half = c->qsp >> 1;
    for (int j = 0; j < L; j++) {
        c->qsp_inv[j] = invmod(c->qsp % q[j], q[j]);
        c->qsp_half_mod[j] = c->qsp_half % q[j];
    }
    /* BEHZ auxiliary base (seal/util/rns.h:324-399):
     * get_primes(2N, 61, L+2) -> m_sk, gamma, B_0..B_{L-1} */
    u64 aux[ORC_MAXK + 2];
    if (orc_get_primes(2 * n, 61, (size_t)L + 2, aux)) { free(c); return NULL; }
    c->msk = aux[0]; c->gamma = aux[1];
    u64 Bp[ORC_MAXK];
    for (int i = 0; i < L; i++) Bp[i] = aux[2 + i];
    for (int i = 0; i < L; i++) { modtab_init(&c->bsk[i], Bp[i]); ntt_tables_init(&c->bsk[i], logn, NULL); }
    modtab_init(&c->bsk[L], c->msk); ntt_tables_init(&c->bsk[L], logn, NULL);
    modtab_init(&c->mgamma, c->gamma);
    const u64 MT = ((u64)1) << 32;
    for (int i = 0; i < L; i++) {
        c->inv_punct_q[i] = invmod(prod_mod_except(dq, L, i, q[i]), q[i]);
        for (int p = 0; p <= L; p++) c->punct_q_bsk[i][p] = prod_mod_except(dq, L, i, c->bsk[p].q);
        c->punct_q_mt[i] = prod_mod_except(dq, L, i, MT);
        c->punct_q_t[i] = prod_mod_except(dq, L, i, t);
        c->punct_q_g[i] = prod_mod_except(dq, L, i, c->gamma);
        c->tg_mod_q[i] = mulmod_slow(t % q[i], c->gamma % q[i], q[i]);
    }
    c->neg_inv_q_mt = (MT - invmod(prod_mod_except(dq, L, -1, MT), MT)) & (MT - 1);
    c->neg_inv_q_t = negmod(invmod(c->q_mod_t, t), t);
    c->neg_inv_q_g = negmod(invmod(prod_mod_except(dq, L, -1, c->gamma), c->gamma), c->gamma);
    c->inv_g_t = invmod(c->gamma % t, t);
    for (int p = 0; p <= L; p++) {
        u64 P = c->bsk[p].q;
        c->q_mod_bsk[p] = prod_mod_except(dq, L, -1, P);
        c->inv_mt_bsk[p] = invmod(MT % P, P);
        c->inv_q_bsk[p] = invmod(c->q_mod_bsk[p], P);
    }
    for (int i = 0; i < L; i++) {
        c->inv_punct_B[i] = invmod(prod_mod_except(Bp, L, i, Bp[i]), Bp[i]);
        for (int j = 0; j < L; j++) c->punct_B_q[i][j] = prod_mod_except(Bp, L, i, q[j]);
        c->punct_B_msk[i] = prod_mod_except(Bp, L, i, c->msk);
    }
    c->inv_B_msk = invmod(prod_mod_except(Bp, L, -1, c->msk), c->msk);
    for (int j = 0; j < L; j++) c->B_mod_q[j] = prod_mod_except(Bp, L, -1, q[j]);
    return c;
}

void orc_ctx_destroy(orc_ctx *c)
{
    if (!c) return;
    for (int i = 0; i < c->K; i++) modtab_free(&c->m[i]);
    for (int i = 0; i <= c->L; i++) modtab_free(&c->bsk[i]);
    modtab_free(&c->pt);
    free(c->slot_map);
    free(c);
}
size_t orc_ctx_n(const orc_ctx *c) { return c->n; }
int orc_ctx_L(const orc_ctx *c) { return c->L; }
int orc_ctx_K(const orc_ctx *c) { return c->K; }
u64 orc_ctx_query(const orc_ctx *c, const char *what, int i)
{
    if (!strcmp(what, "root")) return c->root[i];
    if (!strcmp(what, "bsk")) return c->bsk[i].q;
    if (!strcmp(what, "gamma")) return c->gamma;
    if (!strcmp(what, "delta")) return c->delta[i];
    if (!strcmp(what, "q_mod_t")) return c->q_mod_t;
    if (!strcmp(what, "qsp_inv")) return c->qsp_inv[i];
    return 0;
}
static const modtab *get_mod(const orc_ctx *c, int mi)
{
    if (mi < 0) return &c->pt;
    if (mi < c->K) return &c->m[mi];
    return &c->bsk[mi - c->K];
}
void orc_ctx_ntt_table(const orc_ctx *c, int mi, int inverse, int shoup, u64 *out)
{
    const modtab *m = get_mod(c, mi);
    const u64 *src = inverse ? (shoup ? m->iws : m->iw) : (shoup ? m->ws : m->w);
    memcpy(out, src, 8 * c->n);
}

/* ------------------------------------------------------------------ */
/* NTT  (seal/util/dwthandler.h:94-191 forward CT, :202-356 inverse GS) */
/* ------------------------------------------------------------------ */
static void ntt_fwd(const modtab *m, int logn, u64 *a)
{
    size_t n = (size_t)1 << logn, t = n;
    u64 q = m->q, q2 = 2 * q;
    for (size_t mm = 1; mm < n; mm <<= 1) {
        t >>= 1;
        for (size_t i = 0; i < mm; i++) {
            u64 w = m->w[mm + i], ws = m->ws[mm + i];
            u64 *x = a + 2 * i * t, *y = x + t;
            for (size_t j = 0; j < t; j++) {
                u64 u = x[j]; u -= (u >= q2) ? q2 : 0;       /* [0,2q) */
                u64 v = shoup_lazy(y[j], w, ws, q);           /* [0,2q) */
                x[j] = u + v;                                  /* [0,4q) */
                y[j] = u + q2 - v;
            }
        }
    }
    for (size_t j = 0; j < n; j++) {
        u64 u = a[j];
        u -= (u >= q2) ? q2 : 0;
        u -= (u >= q) ? q : 0;
        a[j] = u;
    }
}
static void ntt_inv(const modtab *m, int logn, u64 *a)
{
    size_t n = (size_t)1 << logn, t = 1;
    u64 q = m->q, q2 = 2 * q;
    for (size_t mm = n; mm > 1; mm >>= 1) {
        size_t h = mm >> 1;
        for (size_t i = 0; i < h; i++) {
            u64 w = m->iw[h + i], ws = m->iws[h + i];
            u64 *x = a + 2 * i * t, *y = x + t;
            for (size_t j = 0; j < t; j++) {
                u64 u = x[j], v = y[j];                        /* [0,2q) */
                u64 s = u + v; s -= (s >= q2) ? q2 : 0;
                x[j] = s;
                y[j] = shoup_lazy(u + q2 - v, w, ws, q);
            }
        }
        t <<= 1;
    }
    for (size_t j = 0; j < n; j++) {
        u64 u = shoup_lazy(a[j], m->ninv, m->ninvs, q);
        u -= (u >= q) ? q : 0;
        a[j] = u;
    }
}
void orc_ntt_fwd(const orc_ctx *c, int mi, u64 *a) { ntt_fwd(get_mod(c, mi), c->logn, a); }
void orc_ntt_inv(const orc_ctx *c, int mi, u64 *a) { ntt_inv(get_mod(c, mi), c->logn, a); }

/* ------------------------------------------------------------------ */
/* BatchEncoder  (seal/batchencoder.h:80-217)              */
/* ------------------------------------------------------------------ */
void orc_encode(const orc_ctx *c, const u64 *vals, size_t count, u64 *plain)
{
    size_t n = c->n;
    memset(plain, 0, 8 * n);
    for (size_t i = 0; i < count && i < n; i++) plain[c->slot_map[i]] = vals[i] % c->t;
    ntt_inv(&c->pt, c->logn, plain);
}
void orc_decode(const orc_ctx *c, const u64 *plain, u64 *vals)
{
    size_t n = c->n;
    u64 *tmp = (u64 *)malloc(8 * n);
    memcpy(tmp, plain, 8 * n);
    ntt_fwd(&c->pt, c->logn, tmp);
    for (size_t i = 0; i < n; i++) vals[i] = tmp[c->slot_map[i]];
    free(tmp);
}

/* ------------------------------------------------------------------ */
/* Galois  (seal/util/galois.h)                            */
/* ------------------------------------------------------------------ */
uint32_t orc_galois_elt_from_step(const orc_ctx *c, int step)
{
    u64 n = c->n, m = 2 * n;
    if (step == 0) return (uint32_t)(m - 1);
    u64 pos = (u64)abs(step);
    if (pos >= (n >> 1)) return 0;
    u64 e = step < 0 ? (n >> 1) - pos : pos;
    u64 g = 1;
    for (u64 i = 0; i < e; i++) g = g * 3 % m;
    return (uint32_t)g;
}
/* GaloisTool::get_elts_all (seal/util/galois.h:131): 3^(2^i), 3^-(2^i), then 2N-1 */
int orc_galois_elts_all(const orc_ctx *c, uint32_t *out)
{
    u64 m = 2 * c->n;
    int cnt = 0;
    u64 pos = 3, neg = invmod(3, m);
    for (int i = 0; i < c->logn - 1; i++) {
        out[cnt++] = (uint32_t)pos;
        pos = pos * pos % m;
        out[cnt++] = (uint32_t)neg;
        neg = neg * neg % m;
    }
    out[cnt++] = (uint32_t)(m - 1);
    return cnt;
}
static void galois_poly(int logn, u64 q, uint32_t elt, const u64 *in, u64 *out)
{
    size_t n = (size_t)1 << logn;
    for (size_t i = 0; i < n; i++) {
        u64 raw = (u64)i * elt;
        size_t idx = raw & (n - 1);
        u64 v = in[i];
        if ((raw >> logn) & 1) v = negmod(v, q);
        out[idx] = v;
    }
}
void orc_apply_galois_poly(const orc_ctx *c, int mi, uint32_t elt, const u64 *in, u64 *out)
{
    galois_poly(c->logn, get_mod(c, mi)->q, elt, in, out);
}

/* ------------------------------------------------------------------ */
/* RNG (oracle-side key material only)                                 */
/* ------------------------------------------------------------------ */
typedef struct { u64 s[4]; } rng_t;
static u64 splitmix(u64 *x)
{
    u64 z = (*x += 0x9e3779b97f4a7c15ULL);
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
    return z ^ (z >> 31);
}
static void rng_seed(rng_t *r, u64 seed) { for (int i = 0; i < 4; i++) r->s[i] = splitmix(&seed); }
static inline u64 rotl(u64 x, int k) { return (x << k) | (x >> (64 - k)); }
static u64 rng_next(rng_t *r)
{
    u64 *s = r->s, res = rotl(s[1] * 5, 7) * 9, t = s[1] << 17;
    s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3]; s[2] ^= t; s[3] = rotl(s[3], 45);
    return res;
}
static u64 rng_uniform(rng_t *r, u64 q)
{
    u64 mask = ~(u64)0 >> __builtin_clzll(q);
    for (;;) { u64 v = rng_next(r) & mask; if (v < q) return v; }
}
static void sample_ternary(rng_t *r, size_t n, int8_t *out)
{
    for (size_t i = 0; i < n; i++) out[i] = (int8_t)((int)rng_uniform(r, 3) - 1);
}
static void sample_cbd(rng_t *r, size_t n, int8_t *out)
{ /* centred binomial, 21 bits - 21 bits (sigma ~ 3.24), as SEAL 4.0 sample_poly_cbd */
    for (size_t i = 0; i < n; i++) {
        u64 v = rng_next(r);
        out[i] = (int8_t)(__builtin_popcountll(v & 0x1fffff) - __builtin_popcountll((v >> 21) & 0x1fffff));
    }
}
static void small_to_rns(const int8_t *s, size_t n, u64 q, u64 *out)
{
    for (size_t i = 0; i < n; i++) out[i] = s[i] < 0 ? q - (u64)(-s[i]) : (u64)s[i];
}

/* ------------------------------------------------------------------ */
/* keys and encryption                                                 */
/* ------------------------------------------------------------------ */
void orc_keygen_secret(const orc_ctx *c, u64 seed, u64 *sk)
{
    size_t n = c->n;
    rng_t r; rng_seed(&r, seed);
    int8_t *s = (int8_t *)malloc(n);
    sample_ternary(&r, n, s);
    for (int j = 0; j < c->K; j++) {
        small_to_rns(s, n, c->m[j].q, sk + (size_t)j * n);
        ntt_fwd(&c->m[j], c->logn, sk + (size_t)j * n);
    }
    free(s);
}
/* (c0,c1) = (-(a s + e), a) at key level, NTT form (Encryptor::encrypt_zero_symmetric shape) */
static void enc_zero_sym_ntt(const orc_ctx *c, const u64 *sk, rng_t *r, u64 *c0, u64 *c1)
{
    size_t n = c->n;
    int8_t *e = (int8_t *)malloc(n);
    sample_cbd(r, n, e);
    for (int j = 0; j < c->K; j++) {
        const modtab *m = &c->m[j];
        u64 *p0 = c0 + (size_t)j * n, *p1 = c1 + (size_t)j * n;
        small_to_rns(e, n, m->q, p0);
        ntt_fwd(m, c->logn, p0);
        for (size_t i = 0; i < n; i++) {
            u64 a = rng_uniform(r, m->q);
            p1[i] = a;
            p0[i] = negmod(addmod(mulmod(a, sk[(size_t)j * n + i], m), p0[i], m->q), m->q);
        }
    }
    free(e);
}
void orc_keygen_public(const orc_ctx *c, const u64 *sk, u64 seed, u64 *pk)
{
    rng_t r; rng_seed(&r, seed);
    enc_zero_sym_ntt(c, sk, &r, pk, pk + (size_t)c->K * c->n);
}
/* KeyGenerator::generate_one_kswitch_key (seal/keygenerator.h; kswitchkeys.h:90-130):
 * key[I] = Enc_sym(0) at key level with (q_sp mod q_I) * new_key[I] added to limb I of c0 */
static void gen_kswitch(const orc_ctx *c, const u64 *sk, const u64 *new_key_ntt, rng_t *r, u64 *ksk)
{
    size_t n = c->n; int K = c->K, L = c->L;
    for (int I = 0; I < L; I++) {
        u64 *k0 = ksk + ((size_t)I * 2 + 0) * K * n, *k1 = ksk + ((size_t)I * 2 + 1) * K * n;
        enc_zero_sym_ntt(c, sk, r, k0, k1);
        const modtab *m = &c->m[I];
        u64 factor = c->qsp % m->q;
        for (size_t i = 0; i < n; i++) {
            u64 tmp = mulmod(new_key_ntt[(size_t)I * n + i], factor, m);
            k0[(size_t)I * n + i] = addmod(k0[(size_t)I * n + i], tmp, m->q);
        }
    }
}
void orc_keygen_relin(const orc_ctx *c, const u64 *sk, u64 seed, u64 *ksk)
{
    size_t n = c->n;
    rng_t r; rng_seed(&r, seed);
    u64 *s2 = (u64 *)malloc(8 * n * c->K);
    for (int j = 0; j < c->K; j++)
        for (size_t i = 0; i < n; i++) s2[(size_t)j * n + i] = mulmod(sk[(size_t)j * n + i], sk[(size_t)j * n + i], &c->m[j]);
    gen_kswitch(c, sk, s2, &r, ksk);
    free(s2);
}
void orc_keygen_galois(const orc_ctx *c, const u64 *sk, uint32_t elt, u64 seed, u64 *ksk)
{
    size_t n = c->n;
    rng_t r; rng_seed(&r, seed);
    u64 *sg = (u64 *)malloc(8 * n * c->K), *tmp = (u64 *)malloc(8 * n);
    for (int j = 0; j < c->K; j++) {
        memcpy(tmp, sk + (size_t)j * n, 8 * n);
        ntt_inv(&c->m[j], c->logn, tmp);
        galois_poly(c->logn, c->m[j].q, elt, tmp, sg + (size_t)j * n);
        ntt_fwd(&c->m[j], c->logn, sg + (size_t)j * n);
    }
    gen_kswitch(c, sk, sg, &r, ksk);
    free(sg); free(tmp);
}

/* multiply_add_plain_with_scaling_variant (seal/util/scalingvariant.h:23) */
static void add_scaled_plain(const orc_ctx *c, const u64 *plain, u64 *c0, int subtract)
{
    size_t n = c->n;
    for (size_t i = 0; i < n; i++) {
        u128 num = (u128)plain[i] * c->q_mod_t + c->up_thr;
        u64 fix = (u64)(num / c->t);
        for (int j = 0; j < c->L; j++) {
            const modtab *m = &c->m[j];
            u64 v = red128((u128)plain[i] * c->delta[j] + fix, m);
            u64 *p = c0 + (size_t)j * n + i;
            *p = subtract ? submod(*p, v, m->q) : addmod(*p, v, m->q);
        }
    }
}
static void encrypt_common(const orc_ctx *c, const u64 *k0, const u64 *k1, int sym, const u64 *sk,
                           const u64 *plain, rng_t *r, u64 *ct)
{
    size_t n = c->n; int L = c->L;
    int8_t *u = (int8_t *)malloc(n), *e0 = (int8_t *)malloc(n), *e1 = (int8_t *)malloc(n);
    u64 *un = (u64 *)malloc(8 * n), *tmp = (u64 *)malloc(8 * n);
    sample_ternary(r, n, u); sample_cbd(r, n, e0); sample_cbd(r, n, e1);
    for (int j = 0; j < L; j++) {
        const modtab *m = &c->m[j];
        u64 *c0 = ct + (size_t)j * n, *c1 = ct + ((size_t)L + j) * n;
        if (!sym) {
            small_to_rns(u, n, m->q, un);
            ntt_fwd(m, c->logn, un);
            for (size_t i = 0; i < n; i++) {
                c0[i] = mulmod(un[i], k0[(size_t)j * n + i], m);
                c1[i] = mulmod(un[i], k1[(size_t)j * n + i], m);
            }
            ntt_inv(m, c->logn, c0); ntt_inv(m, c->logn, c1);
            small_to_rns(e0, n, m->q, tmp);
            for (size_t i = 0; i < n; i++) c0[i] = addmod(c0[i], tmp[i], m->q);
            small_to_rns(e1, n, m->q, tmp);
            for (size_t i = 0; i < n; i++) c1[i] = addmod(c1[i], tmp[i], m->q);
        } else {
            for (size_t i = 0; i < n; i++) {
                u64 a = rng_uniform(r, m->q);
                c1[i] = a;
                c0[i] = mulmod(a, sk[(size_t)j * n + i], m);
            }
            small_to_rns(e0, n, m->q, tmp);
            ntt_fwd(m, c->logn, tmp);
            for (size_t i = 0; i < n; i++) c0[i] = negmod(addmod(c0[i], tmp[i], m->q), m->q);
            ntt_inv(m, c->logn, c0); ntt_inv(m, c->logn, c1);
        }
    }
    add_scaled_plain(c, plain, ct, 0);
    free(u); free(e0); free(e1); free(un); free(tmp);
}
void orc_encrypt(const orc_ctx *c, const u64 *pk, const u64 *plain, u64 seed, u64 *ct)
{
    rng_t r; rng_seed(&r, seed);
    encrypt_common(c, pk, pk + (size_t)c->K * c->n, 0, NULL, plain, &r, ct);
}
void orc_encrypt_symmetric(const orc_ctx *c, const u64 *sk, const u64 *plain, u64 seed, u64 *ct)
{
    rng_t r; rng_seed(&r, seed);
    encrypt_common(c, NULL, NULL, 1, sk, plain, &r, ct);
}
/* phase = c0 + c1 s (+ c2 s^2) per data limb, coefficient form [L][N] */
static void phase(const orc_ctx *c, const u64 *sk, const u64 *ct, int size, u64 *ph)
{
    size_t n = c->n; int L = c->L;
    u64 *tmp = (u64 *)malloc(8 * n), *acc = (u64 *)malloc(8 * n);
    for (int j = 0; j < L; j++) {
        const modtab *m = &c->m[j];
        const u64 *s = sk + (size_t)j * n;
        memset(acc, 0, 8 * n);
        /* Horner in NTT domain: ((c2 s + c1) s) then + c0 in coeff domain */
        for (int k = size - 1; k >= 1; k--) {
            memcpy(tmp, ct + ((size_t)k * L + j) * n, 8 * n);
            ntt_fwd(m, c->logn, tmp);
            for (size_t i = 0; i < n; i++) acc[i] = mulmod(addmod(acc[i], tmp[i], m->q), s[i], m);
        }
        ntt_inv(m, c->logn, acc);
        for (size_t i = 0; i < n; i++) ph[(size_t)j * n + i] = addmod(acc[i], ct[(size_t)j * n + i], m->q);
    }
    free(tmp); free(acc);
}
void orc_phase(const orc_ctx *c, const u64 *sk, const u64 *ct, int size, u64 *ph) { phase(c, sk, ct, size, ph); }

/* Decryptor::decrypt -> RNSTool::decrypt_scale_and_round (seal/util/rns.h:230-243) */
void orc_decrypt(const orc_ctx *c, const u64 *sk, const u64 *ct, int size, u64 *plain)
{
    size_t n = c->n; int L = c->L;
    u64 *ph = (u64 *)malloc(8 * n * L);
    phase(c, sk, ct, size, ph);
    u64 gdiv2 = c->gamma >> 1;
    for (size_t i = 0; i < n; i++) {
        u128 st = 0, sg = 0;
        for (int j = 0; j < L; j++) {
            const modtab *m = &c->m[j];
            u64 v = mulmod(ph[(size_t)j * n + i], c->tg_mod_q[j], m);
            v = mulmod(v, c->inv_punct_q[j], m);
            st += (u128)(v % c->t) * c->punct_q_t[j];
            sg = (sg + (u128)red64(v, &c->mgamma) * c->punct_q_g[j]) % c->gamma;
        }
        u64 vt = (u64)(st % c->t), vg = (u64)sg;
        vt = mulmod_slow(vt, c->neg_inv_q_t, c->t);
        vg = mulmod_slow(vg, c->neg_inv_q_g, c->gamma);
        u64 d;
        if (vg > gdiv2) d = addmod(vt, (c->gamma - vg) % c->t, c->t);
        else d = submod(vt, vg % c->t, c->t);
        plain[i] = d ? mulmod_slow(d, c->inv_g_t, c->t) : 0;
    }
    free(ph);
}

/* ------------------------------------------------------------------ */
/* Evaluator element-wise ops  (seal/evaluator.h:92-132, 665-680)       */
/* ------------------------------------------------------------------ */
void orc_add(const orc_ctx *c, const u64 *a, const u64 *b, int size, u64 *out)
{
    size_t n = c->n;
    for (int k = 0; k < size; k++)
        for (int j = 0; j < c->L; j++) {
            u64 q = c->m[j].q;
            size_t o = ((size_t)k * c->L + j) * n;
            for (size_t i = 0; i < n; i++) out[o + i] = addmod(a[o + i], b[o + i], q);
        }
}
void orc_negate(const orc_ctx *c, const u64 *a, int size, u64 *out)
{
    size_t n = c->n;
    for (int k = 0; k < size; k++)
        for (int j = 0; j < c->L; j++) {
            u64 q = c->m[j].q;
            size_t o = ((size_t)k * c->L + j) * n;
            for (size_t i = 0; i < n; i++) out[o + i] = negmod(a[o + i], q);
        }
}
void orc_add_plain(const orc_ctx *c, const u64 *a, const u64 *plain, u64 *out)
{
    if (out != a) memcpy(out, a, 8 * 2 * c->L * c->n);
    add_scaled_plain(c, plain, out, 0);
}
void orc_sub_plain(const orc_ctx *c, const u64 *a, const u64 *plain, u64 *out)
{
    if (out != a) memcpy(out, a, 8 * 2 * c->L * c->n);
    add_scaled_plain(c, plain, out, 1);
}
/* Evaluator::multiply_plain -> multiply_plain_normal (seal/evaluator.h:729-747,1264) */
void orc_multiply_plain(const orc_ctx *c, const u64 *a, const u64 *plain, u64 *out)
{
    size_t n = c->n; int L = c->L;
    u64 *pl = (u64 *)malloc(8 * n), *tmp = (u64 *)malloc(8 * n);
    for (int j = 0; j < L; j++) {
        const modtab *m = &c->m[j];
        for (size_t i = 0; i < n; i++) pl[i] = plain[i] >= c->up_thr ? plain[i] + c->up_inc[j] : plain[i];
        ntt_fwd(m, c->logn, pl);
        for (int k = 0; k < 2; k++) {
            size_t o = ((size_t)k * L + j) * n;
            memcpy(tmp, a + o, 8 * n);
            ntt_fwd(m, c->logn, tmp);
            for (size_t i = 0; i < n; i++) tmp[i] = mulmod(tmp[i], pl[i], m);
            ntt_inv(m, c->logn, tmp);
            memcpy(out + o, tmp, 8 * n);
        }
    }
    free(pl); free(tmp);
}

/* ------------------------------------------------------------------ */
/* key switching  (Evaluator::switch_key_inplace, seal/evaluator.h:1260) */
/* ------------------------------------------------------------------ */
void orc_switch_key(const orc_ctx *c, u64 *ct, const u64 *d, const u64 *ksk)
{
    size_t n = c->n; int K = c->K, L = c->L;
    u64 *S = (u64 *)calloc((size_t)2 * K * n, 8); /* [k][J][N] accumulators (NTT form) */
    u64 *tmp = (u64 *)malloc(8 * n);
    u128 *acc = (u128 *)malloc(sizeof(u128) * n * 2);
    for (int J = 0; J < K; J++) {
        const modtab *mJ = &c->m[J];
        memset(acc, 0, sizeof(u128) * n * 2);
        for (int I = 0; I < L; I++) {
            const u64 *src = d + (size_t)I * n;
            /* d[I] is canonical in [0,q_I); re-reduce only when q_I > q_J */
            if (c->m[I].q <= mJ->q) memcpy(tmp, src, 8 * n);
            else for (size_t i = 0; i < n; i++) tmp[i] = red64(src[i], mJ);
            ntt_fwd(mJ, c->logn, tmp);
            const u64 *k0 = ksk + (((size_t)I * 2 + 0) * K + J) * n;
            const u64 *k1 = ksk + (((size_t)I * 2 + 1) * K + J) * n;
            for (size_t i = 0; i < n; i++) {
                acc[i] += (u128)tmp[i] * k0[i];
                acc[n + i] += (u128)tmp[i] * k1[i];
            }
            if ((I & 3) == 3) /* keep the lazy 128-bit sum bounded (q<2^61: 4 terms < 2^124) */
                for (size_t i = 0; i < 2 * n; i++) acc[i] = red128(acc[i], mJ);
        }
        for (size_t i = 0; i < n; i++) {
            S[((size_t)0 * K + J) * n + i] = red128(acc[i], mJ);
            S[((size_t)1 * K + J) * n + i] = red128(acc[n + i], mJ);
        }
    }
    for (int k = 0; k < 2; k++) {
        u64 *sp = S + ((size_t)k * K + (K - 1)) * n;
        ntt_inv(&c->m[K - 1], c->logn, sp);
        for (size_t i = 0; i < n; i++) sp[i] = addmod(sp[i], c->qsp_half, c->qsp); /* r_k */
        for (int j = 0; j < L; j++) {
            const modtab *m = &c->m[j];
            u64 *sj = S + ((size_t)k * K + j) * n;
            ntt_inv(m, c->logn, sj);
            u64 *dst = ct + ((size_t)k * L + j) * n;
            for (size_t i = 0; i < n; i++) {
                u64 r = red64(sp[i], m);
                u64 v = addmod(submod(sj[i], r, m->q), c->qsp_half_mod[j], m->q);
                v = mulmod(v, c->qsp_inv[j], m);
                dst[i] = addmod(dst[i], v, m->q);
            }
        }
    }
    free(S); free(tmp); free(acc);
}

/* Evaluator::apply_galois_inplace (seal/evaluator.h:889) */
void orc_apply_galois(const orc_ctx *c, const u64 *a, uint32_t elt, const u64 *ksk, u64 *out)
{
    size_t n = c->n; int L = c->L;
    u64 *d = (u64 *)malloc(8 * n * L), *res = (u64 *)calloc((size_t)2 * L * n, 8);
    for (int j = 0; j < L; j++) {
        galois_poly(c->logn, c->m[j].q, elt, a + (size_t)j * n, res + (size_t)j * n);
        galois_poly(c->logn, c->m[j].q, elt, a + ((size_t)L + j) * n, d + (size_t)j * n);
    }
    orc_switch_key(c, res, d, ksk);
    memcpy(out, res, 8 * 2 * L * n);
    free(d); free(res);
}
static const u64 *find_gk(const orc_ctx *c, const orc_gkeys *gk, uint32_t elt)
{
    size_t keysz = (size_t)c->L * 2 * c->K * c->n;
    for (int i = 0; i < gk->nk; i++)
        if (gk->elts[i] == elt) return gk->keys + (size_t)i * keysz;
    return NULL;
}
/* Evaluator::rotate_internal (seal/evaluator.h:1234): direct key if present, else NAF */
int orc_rotate_rows(const orc_ctx *c, const u64 *a, int step, const orc_gkeys *gk, u64 *out)
{
    size_t sz = (size_t)2 * c->L * c->n;
    if (step == 0) { if (out != a) memcpy(out, a, 8 * sz); return 0; }
    uint32_t elt = orc_galois_elt_from_step(c, step);
    if (!elt) return -1;
    const u64 *k = find_gk(c, gk, elt);
    if (k) { orc_apply_galois(c, a, elt, k, out); return 1; }
    int nf[40], cnt = orc_naf(step, nf), ks = 0;
    if (cnt == 1) return -1; /* "Galois key not present" */
    u64 *cur = (u64 *)malloc(8 * sz);
    memcpy(cur, a, 8 * sz);
    for (int i = 0; i < cnt; i++) {
        if ((size_t)abs(nf[i]) == (c->n >> 1)) continue;
        int r = orc_rotate_rows(c, cur, nf[i], gk, cur);
        if (r < 0) { free(cur); return -1; }
        ks += r;
    }
    memcpy(out, cur, 8 * sz);
    free(cur);
    return ks;
}
int orc_rotate_columns(const orc_ctx *c, const u64 *a, const orc_gkeys *gk, u64 *out)
{
    uint32_t elt = (uint32_t)(2 * c->n - 1);
    const u64 *k = find_gk(c, gk, elt);
    if (!k) return -1;
    orc_apply_galois(c, a, elt, k, out);
    return 1;
}

/* ------------------------------------------------------------------ */
/* BEHZ multiply  (Evaluator::bfv_multiply, seal/evaluator.h:1211-1217; */
/* RNSTool, seal/util/rns.h:205-243)                        */
/* ------------------------------------------------------------------ */
/* x [L][N] base q (coeff form) -> xq [L][N] NTT form, xb [L+1][N] base Bsk NTT form */
static void behz_extend_ntt(const orc_ctx *c, const u64 *x, u64 *xq, u64 *xb)
{
    size_t n = c->n; int L = c->L;
    const u64 MT = ((u64)1) << 32, MTm = MT - 1;
    u64 *tmp = (u64 *)malloc(8 * n * L);
    for (int i = 0; i < L; i++) {
        const modtab *m = &c->m[i];
        for (size_t l = 0; l < n; l++) {
            u64 v = mulmod(x[(size_t)i * n + l], MT % m->q, m);  /* fastbconv_m_tilde: x*m~ */
            tmp[(size_t)i * n + l] = mulmod(v, c->inv_punct_q[i], m);
        }
    }
    for (size_t l = 0; l < n; l++) {
        u64 ymt = 0;
        for (int i = 0; i < L; i++) ymt += tmp[(size_t)i * n + l] * c->punct_q_mt[i];
        ymt &= MTm;
        u64 r = (ymt * c->neg_inv_q_mt) & MTm;               /* sm_mrq */
        for (int p = 0; p <= L; p++) {
            const modtab *mp = &c->bsk[p];
            u128 acc = 0;
            for (int i = 0; i < L; i++) {
                acc += (u128)tmp[(size_t)i * n + l] * c->punct_q_bsk[i][p];
                if ((i & 3) == 3) acc = red128(acc, mp);
            }
            u64 y = red128(acc, mp);
            u64 rr = r >= (MT >> 1) ? r + (mp->q - MT) : r;
            u64 v = red128((u128)c->q_mod_bsk[p] * rr + y, mp);
            xb[(size_t)p * n + l] = mulmod(v, c->inv_mt_bsk[p], mp);
        }
    }
    for (int i = 0; i < L; i++) {
        memcpy(xq + (size_t)i * n, x + (size_t)i * n, 8 * n);
        ntt_fwd(&c->m[i], c->logn, xq + (size_t)i * n);
    }
    for (int p = 0; p <= L; p++) ntt_fwd(&c->bsk[p], c->logn, xb + (size_t)p * n);
    free(tmp);
}
void orc_multiply(const orc_ctx *c, const u64 *a, const u64 *b, u64 *out3)
{
    size_t n = c->n; int L = c->L, LB = L + 1;
    size_t pq = (size_t)L * n, pb = (size_t)LB * n;
    u64 *aq = (u64 *)malloc(8 * 2 * pq), *ab = (u64 *)malloc(8 * 2 * pb);
    u64 *bq = (u64 *)malloc(8 * 2 * pq), *bb = (u64 *)malloc(8 * 2 * pb);
    u64 *dq = (u64 *)malloc(8 * 3 * pq), *db = (u64 *)malloc(8 * 3 * pb);
    for (int k = 0; k < 2; k++) {
        behz_extend_ntt(c, a + k * pq, aq + k * pq, ab + k * pb);
        behz_extend_ntt(c, b + k * pq, bq + k * pq, bb + k * pb);
    }
    /* tensor in both bases, INTT, times t */
    for (int base = 0; base < 2; base++) {
        int cnt = base ? LB : L;
        size_t ps = base ? pb : pq;
        const u64 *A = base ? ab : aq, *B = base ? bb : bq;
        u64 *D = base ? db : dq;
        for (int j = 0; j < cnt; j++) {
            const modtab *m = base ? &c->bsk[j] : &c->m[j];
            const u64 *a0 = A + (size_t)j * n, *a1 = A + ps + (size_t)j * n;
            const u64 *b0 = B + (size_t)j * n, *b1 = B + ps + (size_t)j * n;
            u64 *d0 = D + (size_t)j * n, *d1 = D + ps + (size_t)j * n, *d2 = D + 2 * ps + (size_t)j * n;
            for (size_t l = 0; l < n; l++) {
                d0[l] = mulmod(a0[l], b0[l], m);
                d1[l] = red128((u128)a0[l] * b1[l] + (u128)a1[l] * b0[l], m);
                d2[l] = mulmod(a1[l], b1[l], m);
            }
            u64 tt = c->t % m->q;
            u64 *dd[3] = {d0, d1, d2};
            for (int k = 0; k < 3; k++) {
                ntt_inv(m, c->logn, dd[k]);
                for (size_t l = 0; l < n; l++) dd[k][l] = mulmod(dd[k][l], tt, m);
            }
        }
    }
    /* fast_floor then fastbconv_sk, per output poly */
    u64 *tq = (u64 *)malloc(8 * pq), *f = (u64 *)malloc(8 * pb), *tb = (u64 *)malloc(8 * pq);
    for (int k = 0; k < 3; k++) {
        const u64 *xq = dq + k * pq, *xb = db + k * pb;
        for (int i = 0; i < L; i++)
            for (size_t l = 0; l < n; l++) tq[(size_t)i * n + l] = mulmod(xq[(size_t)i * n + l], c->inv_punct_q[i], &c->m[i]);
        for (int p = 0; p <= L; p++) {
            const modtab *mp = &c->bsk[p];
            for (size_t l = 0; l < n; l++) {
                u128 acc = 0;
                for (int i = 0; i < L; i++) {
                    acc += (u128)tq[(size_t)i * n + l] * c->punct_q_bsk[i][p];
                    if ((i & 3) == 3) acc = red128(acc, mp);
                }
                u64 conv = red128(acc, mp);
                f[(size_t)p * n + l] = mulmod(addmod(xb[(size_t)p * n + l], mp->q - conv, mp->q), c->inv_q_bsk[p], mp);
            }
        }
        /* fastbconv_sk: B -> q with Shenoy-Kumaresan correction through m_sk */
        for (int i = 0; i < L; i++)
            for (size_t l = 0; l < n; l++) tb[(size_t)i * n + l] = mulmod(f[(size_t)i * n + l], c->inv_punct_B[i], &c->bsk[i]);
        const modtab *ms = &c->bsk[L];
        u64 msk_half = c->msk >> 1;
        for (size_t l = 0; l < n; l++) {
            u128 acc = 0;
            for (int i = 0; i < L; i++) {
                acc += (u128)tb[(size_t)i * n + l] * c->punct_B_msk[i];
                if ((i & 3) == 3) acc = red128(acc, ms);
            }
            u64 conv = red128(acc, ms);
            u64 alpha = mulmod(addmod(conv, ms->q - f[(size_t)L * n + l], ms->q), c->inv_B_msk, ms);
            for (int j = 0; j < L; j++) {
                const modtab *m = &c->m[j];
                u128 a2 = 0;
                for (int i = 0; i < L; i++) {
                    a2 += (u128)tb[(size_t)i * n + l] * c->punct_B_q[i][j];
                    if ((i & 3) == 3) a2 = red128(a2, m);
                }
                u64 v = red128(a2, m);
                if (alpha > msk_half) v = red128((u128)(c->msk - alpha) * c->B_mod_q[j] + v, m);
                else v = red128((u128)alpha * (m->q - c->B_mod_q[j]) + v, m);
                out3[k * pq + (size_t)j * n + l] = v;
            }
        }
    }
    free(aq); free(ab); free(bq); free(bb); free(dq); free(db); free(tq); free(f); free(tb);
}
/* Evaluator::relinearize_inplace (seal/evaluator.h:301-304): key-switch c2 with relin key 0 */
void orc_relinearize(const orc_ctx *c, const u64 *a3, const u64 *rk, u64 *out2)
{
    size_t pq = (size_t)c->L * c->n;
    u64 *res = (u64 *)malloc(8 * 2 * pq);
    memcpy(res, a3, 8 * 2 * pq);
    orc_switch_key(c, res, a3 + 2 * pq, rk);
    memcpy(out2, res, 8 * 2 * pq);
    free(res);
}

/* ------------------------------------------------------------------ */
/* SHAKE128 (FIPS 202) -- the XOF behind libs/keccak's Keccak_Hash*      */
/* ------------------------------------------------------------------ */
static void keccak_f(u64 *s)
{
    static const u64 RC[24] = {
        0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL,
        0x000000000000808bULL, 0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL,
        0x000000000000008aULL, 0x0000000000000088ULL, 0x0000000080008009ULL, 0x000000008000000aULL,
        0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL, 0x8000000000008003ULL,
        0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800aULL, 0x800000008000000aULL,
        0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
    static const int rotc[24] = {1, 3, 6, 10, 15, 21, 28, 36, 45, 55, 2, 14, 27, 41, 56, 8, 25, 43, 62, 18, 39, 61, 20, 44};
    static const int piln[24] = {10, 7, 11, 17, 18, 3, 5, 16, 8, 21, 24, 4, 15, 23, 19, 13, 12, 2, 20, 14, 22, 9, 6, 1};
    for (int round = 0; round < 24; round++) {
        u64 bc[5], t;
        for (int i = 0; i < 5; i++) bc[i] = s[i] ^ s[i + 5] ^ s[i + 10] ^ s[i + 15] ^ s[i + 20];
        for (int i = 0; i < 5; i++) {
            t = bc[(i + 4) % 5] ^ rotl(bc[(i + 1) % 5], 1);
            for (int j = 0; j < 25; j += 5) s[j + i] ^= t;
        }
        t = s[1];
        for (int i = 0; i < 24; i++) {
            int j = piln[i];
            u64 b = s[j];
            s[j] = rotl(t, rotc[i]);
            t = b;
        }
        for (int j = 0; j < 25; j += 5) {
            for (int i = 0; i < 5; i++) bc[i] = s[j + i];
            for (int i = 0; i < 5; i++) s[j + i] ^= (~bc[(i + 1) % 5]) & bc[(i + 2) % 5];
        }
        s[0] ^= RC[round];
    }
}
void orc_shake128_init(orc_shake *s, const uint8_t *in, size_t len)
{
    memset(s, 0, sizeof(*s));
    uint8_t blk[168];
    const size_t rate = 168;
    while (len >= rate) {
        for (size_t i = 0; i < rate / 8; i++) { u64 v; memcpy(&v, in + 8 * i, 8); s->st[i] ^= v; }
        keccak_f(s->st);
        in += rate; len -= rate;
    }
    memset(blk, 0, rate);
    memcpy(blk, in, len);
    blk[len] ^= 0x1f;
    blk[rate - 1] ^= 0x80;
    for (size_t i = 0; i < rate / 8; i++) { u64 v; memcpy(&v, blk + 8 * i, 8); s->st[i] ^= v; }
    s->pos = 168; /* nothing squeezed yet: permute on first squeeze */
}
void orc_shake128_squeeze(orc_shake *s, uint8_t *out, size_t len)
{
    while (len) {
        if (s->pos == 168) {
            keccak_f(s->st);
            memcpy(s->buf, s->st, 168);
            s->pos = 0;
        }
        size_t take = 168 - (size_t)s->pos;
        if (take > len) take = len;
        memcpy(out, s->buf + s->pos, take);
        s->pos += (int)take; out += take; len -= take;
    }
}

/* ------------------------------------------------------------------ */
/* PASTA-3 plain  (src/pasta/pasta_3_plain.cpp)                         */
/* ------------------------------------------------------------------ */
typedef struct { orc_shake sh; u64 p, mask; } pasta_xof;
/* Pasta::init_shake (pasta_3_plain.cpp:56-68) */
static void pasta_init(pasta_xof *x, u64 p, u64 nonce, u64 block)
{
    uint8_t seed[16];
    for (int i = 0; i < 8; i++) { seed[i] = (uint8_t)(nonce >> (56 - 8 * i)); seed[8 + i] = (uint8_t)(block >> (56 - 8 * i)); }
    orc_shake128_init(&x->sh, seed, 16);
    x->p = p;
    int bits = 64 - __builtin_clzll(p);
    x->mask = bits == 64 ? ~(u64)0 : ((((u64)1) << bits) - 1);
}
/* Pasta::generate_random_field_element (pasta_3_plain.cpp:72-82) */
static u64 pasta_elem(pasta_xof *x, int allow_zero)
{
    for (;;) {
        uint8_t b[8];
        orc_shake128_squeeze(&x->sh, b, 8);
        u64 e = 0;
        for (int i = 0; i < 8; i++) e = (e << 8) | b[i];
        e &= x->mask;
        if (!allow_zero && e == 0) continue;
        if (e < x->p) return e;
    }
}
/* Pasta::get_random_matrix / calculate_row (pasta_3_plain.cpp:86-119) */
static void pasta_matrix(pasta_xof *x, u64 *mat)
{
    u64 p = x->p;
    for (int j = 0; j < PASTA_T; j++) mat[j] = pasta_elem(x, 0);
    for (int i = 1; i < PASTA_T; i++) {
        const u64 *prev = mat + (size_t)(i - 1) * PASTA_T;
        u64 *row = mat + (size_t)i * PASTA_T;
        for (int j = 0; j < PASTA_T; j++) {
            u64 tmp = mulmod_slow(mat[j], prev[PASTA_T - 1], p);
            if (j) tmp = (tmp + prev[j - 1]) % p;
            row[j] = tmp;
        }
    }
}
/* draw order per affine layer: M1, M2, rc1, rc2 (pasta_3_seal.cpp:131-133; pasta_3_plain.cpp:286-295) */
void orc_pasta_block_randomness(u64 t, u64 nonce, u64 block, u64 *mats, u64 *rcs)
{
    pasta_xof x;
    pasta_init(&x, t, nonce, block);
    for (int r = 0; r <= PASTA_R; r++) {
        pasta_matrix(&x, mats + ((size_t)r * 2 + 0) * PASTA_T * PASTA_T);
        pasta_matrix(&x, mats + ((size_t)r * 2 + 1) * PASTA_T * PASTA_T);
        for (int i = 0; i < 2 * PASTA_T; i++) rcs[(size_t)r * 2 * PASTA_T + i] = pasta_elem(&x, 1);
    }
}
/* Pasta::gen_keystream (pasta_3_plain.cpp:156-282) */
void orc_pasta_keystream(u64 t, const u64 *key, u64 nonce, u64 block, u64 *ks)
{
    u64 *mats = (u64 *)malloc(8 * 4 * 2 * PASTA_T * PASTA_T), rcs[4 * 2 * PASTA_T];
    orc_pasta_block_randomness(t, nonce, block, mats, rcs);
    u64 s[2][PASTA_T], ns[PASTA_T];
    for (int i = 0; i < PASTA_T; i++) { s[0][i] = key[i] % t; s[1][i] = key[PASTA_T + i] % t; }
    for (int r = 0; r <= PASTA_R; r++) {
        for (int h = 0; h < 2; h++) { /* matmul + add_rc */
            const u64 *M = mats + ((size_t)r * 2 + h) * PASTA_T * PASTA_T;
            for (int i = 0; i < PASTA_T; i++) {
                u64 acc = 0;
                for (int j = 0; j < PASTA_T; j++) acc = (acc + mulmod_slow(M[(size_t)i * PASTA_T + j], s[h][j], t)) % t;
                ns[i] = (acc + rcs[(size_t)r * 2 * PASTA_T + (size_t)h * PASTA_T + i]) % t;
            }
            memcpy(s[h], ns, sizeof(ns));
        }
        for (int i = 0; i < PASTA_T; i++) { /* mix */
            u64 sum = (s[0][i] + s[1][i]) % t;
            s[0][i] = (s[0][i] + sum) % t;
            s[1][i] = (s[1][i] + sum) % t;
        }
        if (r == PASTA_R) break;
        for (int h = 0; h < 2; h++) {
            if (r == PASTA_R - 1) { /* sbox_cube */
                for (int i = 0; i < PASTA_T; i++) {
                    u64 sq = mulmod_slow(s[h][i], s[h][i], t);
                    s[h][i] = mulmod_slow(sq, s[h][i], t);
                }
            } else { /* sbox_feistel */
                ns[0] = s[h][0];
                for (int i = 1; i < PASTA_T; i++) ns[i] = (mulmod_slow(s[h][i - 1], s[h][i - 1], t) + s[h][i]) % t;
                memcpy(s[h], ns, sizeof(ns));
            }
        }
    }
    memcpy(ks, s[0], 8 * PASTA_T);
    free(mats);
}
#define PASTA_NONCE 123456789ULL
/* PASTA::encrypt / decrypt (pasta_3_plain.cpp:9-47) */
void orc_pasta_encrypt(u64 t, const u64 *key, const u64 *pt, size_t n, u64 *ct)
{
    u64 ks[PASTA_T];
    for (size_t b = 0; b * PASTA_T < n; b++) {
        orc_pasta_keystream(t, key, PASTA_NONCE, b, ks);
        for (size_t i = b * PASTA_T; i < (b + 1) * PASTA_T && i < n; i++) ct[i] = (pt[i] + ks[i - b * PASTA_T]) % t;
    }
}
void orc_pasta_decrypt(u64 t, const u64 *key, const u64 *ct, size_t n, u64 *pt)
{
    u64 ks[PASTA_T];
    for (size_t b = 0; b * PASTA_T < n; b++) {
        orc_pasta_keystream(t, key, PASTA_NONCE, b, ks);
        for (size_t i = b * PASTA_T; i < (b + 1) * PASTA_T && i < n; i++) {
            u64 v = ct[i], k = ks[i - b * PASTA_T];
            pt[i] = v >= k ? v - k : v + t - k;
        }
    }
}

/* ------------------------------------------------------------------ */
/* the hot path: PASTA_SEAL::decomposition (pasta_3_seal.cpp:106-172)   */
/* ------------------------------------------------------------------ */
void orc_pasta_pack_key(const orc_ctx *c, const u64 *key, u64 *plain)
{ /* pasta_3_seal.cpp:29-35 */
    size_t half = c->n >> 1;
    u64 *v = (u64 *)calloc(half + PASTA_T, 8);
    for (int i = 0; i < PASTA_T; i++) { v[i] = key[i]; v[half + i] = key[PASTA_T + i]; }
    orc_encode(c, v, half + PASTA_T, plain);
    free(v);
}
/* PASTA_SEAL::diagonal (pasta_3_seal.cpp:370-413) */
static int he_diagonal(const orc_ctx *c, u64 *state, const u64 *m1, const u64 *m2, const orc_gkeys *gk)
{
    size_t n = c->n, half = n >> 1, sz = (size_t)2 * c->L * n;
    if ((size_t)PASTA_T * 2 != n && (size_t)PASTA_T * 4 > n) return -2; /* "too little slots" :376-377 */
    u64 *rot = (u64 *)malloc(8 * sz), *sum = (u64 *)malloc(8 * sz), *tmp = (u64 *)malloc(8 * sz);
    u64 *diag = (u64 *)malloc(8 * (half + PASTA_T)), *pl = (u64 *)malloc(8 * n);
    int rc = 0;
    if (half != PASTA_T) {
        if (orc_rotate_rows(c, state, PASTA_T, gk, rot) < 0) { rc = -1; goto done; }
        orc_add(c, state, rot, 2, state);
    }
    for (int i = 0; i < PASTA_T; i++) {
        memset(diag, 0, 8 * (half + PASTA_T));
        for (int j = 0; j < PASTA_T; j++) {
            diag[j] = m1[(size_t)j * PASTA_T + (j + PASTA_T - i) % PASTA_T];
            diag[j + half] = m2[(size_t)j * PASTA_T + (j + PASTA_T - i) % PASTA_T];
        }
        orc_encode(c, diag, half + PASTA_T, pl);
        if (i == 0) orc_multiply_plain(c, state, pl, sum);
        else {
            if (orc_rotate_rows(c, state, -1, gk, state) < 0) { rc = -1; goto done; }
            orc_multiply_plain(c, state, pl, tmp);
            orc_add(c, sum, tmp, 2, sum);
        }
    }
    memcpy(state, sum, 8 * sz);
done:
    free(rot); free(sum); free(tmp); free(diag); free(pl);
    return rc;
}
/* PASTA_SEAL::babystep_giantstep (pasta_3_seal.cpp:267-366), N1=16, N2=8 (pasta_3_seal.h:35-36) */
static int he_bsgs(const orc_ctx *c, u64 *state, const u64 *m1, const u64 *m2, const orc_gkeys *gk)
{
    enum { N1 = 16, N2 = 8 };
    size_t n = c->n, half = n >> 1, sz = (size_t)2 * c->L * n;
    if ((size_t)PASTA_T * 2 != n && (size_t)PASTA_T * 4 > n) return -2;
    u64 *pls = (u64 *)malloc(8 * n * PASTA_T);
    u64 *diag = (u64 *)malloc(8 * n), *tmpd = (u64 *)malloc(8 * half);
    for (int i = 0; i < PASTA_T; i++) {
        int k = i / N1;
        memset(diag, 0, 8 * n); memset(tmpd, 0, 8 * half);
        u64 d1[PASTA_T], d2[PASTA_T];
        for (int j = 0; j < PASTA_T; j++) {
            d1[j] = m1[(size_t)j * PASTA_T + (j + PASTA_T - i) % PASTA_T];
            d2[j] = m2[(size_t)j * PASTA_T + (j + PASTA_T - i) % PASTA_T];
        }
        /* std::rotate(begin, begin + k*N1, end): left rotation by k*N1 (:297-301) */
        for (int j = 0; j < PASTA_T; j++) {
            diag[j] = d1[(j + k * N1) % PASTA_T];
            tmpd[j] = d2[(j + k * N1) % PASTA_T];
        }
        if (half != PASTA_T) { /* :304-317 */
            for (int mI = 0; mI < k * N1; mI++) {
                size_t src = PASTA_T - 1 - mI, dst = half - 1 - mI;
                diag[dst] = diag[src]; diag[src] = 0;
                tmpd[dst] = tmpd[src]; tmpd[src] = 0;
            }
        }
        for (size_t j = half; j < n; j++) diag[j] = tmpd[j - half];
        orc_encode(c, diag, n, pls + (size_t)i * n);
    }
    u64 *rot = (u64 *)malloc(8 * sz * N1), *inner = (u64 *)malloc(8 * sz), *outer = (u64 *)malloc(8 * sz), *tmp = (u64 *)malloc(8 * sz);
    int rc = 0;
    if (half != PASTA_T) {
        if (orc_rotate_rows(c, state, PASTA_T, gk, tmp) < 0) { rc = -1; goto done; }
        orc_add(c, state, tmp, 2, state);
    }
    memcpy(rot, state, 8 * sz);
    for (int j = 1; j < N1; j++)
        if (orc_rotate_rows(c, rot + (size_t)(j - 1) * sz, -1, gk, rot + (size_t)j * sz) < 0) { rc = -1; goto done; }
    for (int k = 0; k < N2; k++) {
        orc_multiply_plain(c, rot, pls + (size_t)(k * N1) * n, inner);
        for (int j = 1; j < N1; j++) {
            orc_multiply_plain(c, rot + (size_t)j * sz, pls + (size_t)(k * N1 + j) * n, tmp);
            orc_add(c, inner, tmp, 2, inner);
        }
        if (!k) memcpy(outer, inner, 8 * sz);
        else {
            if (orc_rotate_rows(c, inner, -k * N1, gk, inner) < 0) { rc = -1; goto done; }
            orc_add(c, outer, inner, 2, outer);
        }
    }
    memcpy(state, outer, 8 * sz);
done:
    free(pls); free(diag); free(tmpd); free(rot); free(inner); free(outer); free(tmp);
    return rc;
}
int orc_pasta_transcipher_block(const orc_ctx *c, const u64 *enc_key, const u64 *rk, const orc_gkeys *gk,
                                const u64 *cw, size_t ncw, u64 block_index, int use_bsgs, u64 *out)
{
    size_t n = c->n, half = n >> 1, sz = (size_t)2 * c->L * n, pq = (size_t)c->L * n;
    u64 *mats = (u64 *)malloc(8 * 4 * 2 * PASTA_T * PASTA_T), rcs[4 * 2 * PASTA_T];
    orc_pasta_block_randomness(c->t, PASTA_NONCE, block_index, mats, rcs);
    u64 *state = (u64 *)malloc(8 * sz), *tmp = (u64 *)malloc(8 * sz), *t3 = (u64 *)malloc(8 * 3 * pq);
    u64 *vec = (u64 *)malloc(8 * n), *pl = (u64 *)malloc(8 * n);
    int rc = 0;
    memcpy(state, enc_key, 8 * sz);
    for (int r = 0; r <= PASTA_R; r++) {
        const u64 *m1 = mats + ((size_t)r * 2 + 0) * PASTA_T * PASTA_T, *m2 = mats + ((size_t)r * 2 + 1) * PASTA_T * PASTA_T;
        rc = use_bsgs ? he_bsgs(c, state, m1, m2, gk) : he_diagonal(c, state, m1, m2, gk); /* matmul :251-263 */
        if (rc) goto done;
        /* add_rc :205-211 */
        memset(vec, 0, 8 * n);
        for (int i = 0; i < PASTA_T; i++) { vec[i] = rcs[(size_t)r * 256 + i]; vec[half + i] = rcs[(size_t)r * 256 + PASTA_T + i]; }
        orc_encode(c, vec, half + PASTA_T, pl);
        orc_add_plain(c, state, pl, state);
        /* mix :417-423 */
        if (orc_rotate_columns(c, state, gk, tmp) < 0) { rc = -1; goto done; }
        orc_add(c, tmp, state, 2, tmp);
        orc_add(c, state, tmp, 2, state);
        if (r == PASTA_R) break;
        if (r == PASTA_R - 1) {
            /* sbox_cube :215-218 = exponentiate_inplace(state,3,rk) = relin(mul(relin(mul(x,x)),x)) */
            orc_multiply(c, state, state, t3);
            orc_relinearize(c, t3, rk, tmp);
            orc_multiply(c, tmp, state, t3);
            orc_relinearize(c, t3, rk, state);
        } else {
            /* sbox_feistel :222-247 */
            if (orc_rotate_rows(c, state, -1, gk, tmp) < 0) { rc = -1; goto done; }
            for (size_t i = 0; i < half + PASTA_T; i++) vec[i] = 1;
            vec[0] = 0; vec[half] = 0;
            for (size_t i = PASTA_T; i < half; i++) vec[i] = 0;
            orc_encode(c, vec, half + PASTA_T, pl);
            orc_multiply_plain(c, tmp, pl, tmp);
            orc_multiply(c, tmp, tmp, t3);
            orc_relinearize(c, t3, rk, tmp);
            orc_add(c, state, tmp, 2, state);
        }
    }
    /* add cipher :161-169 */
    orc_encode(c, cw, ncw, pl);
    orc_negate(c, state, 2, state);
    orc_add_plain(c, state, pl, out);
done:
    free(mats); free(state); free(tmp); free(t3); free(vec); free(pl);
    return rc;
}
int orc_pasta_transcipher_batch(const orc_ctx *c, const u64 *enc_key, const u64 *rk, const orc_gkeys *gk,
                                const u64 *cw, const uint32_t *ncw, const u64 *block_index, size_t nb,
                                int threads, u64 *out)
{
    size_t sz = (size_t)2 * c->L * c->n;
    int bad = 0;
    (void)threads;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads > 0 ? threads : 1)
#endif
    for (long b = 0; b < (long)nb; b++) {
        int r = orc_pasta_transcipher_block(c, enc_key, rk, gk, cw + (size_t)b * PASTA_T, ncw[b], block_index[b], 0, out + (size_t)b * sz);
        if (r) bad = r;
    }
    return bad;
}
/* SEALZpCipher::mask (SEAL_Cipher.cpp:161-166) */
void orc_mask(const orc_ctx *c, const u64 *a, const u64 *mask_vals, size_t count, u64 *out)
{
    u64 *pl = (u64 *)malloc(8 * c->n);
    orc_encode(c, mask_vals, count, pl);
    orc_multiply_plain(c, a, pl, out);
    free(pl);
}
/* SEALZpCipher::flatten (SEAL_Cipher.cpp:170-181) */
int orc_flatten(const orc_ctx *c, const u64 *blocks, size_t nblocks, const orc_gkeys *gk, u64 *out)
{
    size_t sz = (size_t)2 * c->L * c->n;
    u64 *tmp = (u64 *)malloc(8 * sz);
    memcpy(out, blocks, 8 * sz);
    for (size_t i = 1; i < nblocks; i++) {
        if (orc_rotate_rows(c, blocks + i * sz, -(int)(i * PASTA_T), gk, tmp) < 0) { free(tmp); return -1; }
        orc_add(c, out, tmp, 2, out);
    }
    free(tmp);
    return 0;
}
/* packed_enc_multiply + relinearize_inplace + encrypted_vec_sum
 * (sealhelper.cpp:268-274; CSP.cpp:306; sealhelper.cpp:379-392) */
int orc_fc_row(const orc_ctx *c, const u64 *vi, const u64 *w, const u64 *rk, const orc_gkeys *gk,
               size_t n_inputs, u64 *out)
{
    size_t sz = (size_t)2 * c->L * c->n, pq = (size_t)c->L * c->n;
    u64 *t3 = (u64 *)malloc(8 * 3 * pq), *prod = (u64 *)malloc(8 * sz), *rot = (u64 *)malloc(8 * sz);
    int ks = 0;
    orc_multiply(c, vi, w, t3);
    orc_relinearize(c, t3, rk, prod);
    memcpy(out, prod, 8 * sz);
    for (long i = 1; i < (long)n_inputs; i++) {
        int r = orc_rotate_rows(c, prod, -(int)i, gk, rot);
        if (r < 0) { ks = -1; break; }
        ks += r;
        orc_add(c, out, rot, 2, out);
    }
    free(t3); free(prod); free(rot);
    return ks;
}
