// hhe_context.cpp -- parameter derivation and device-resident tables for one BFV context.
// Follows what SEAL 4.0.0's SEALContext computes for the same (N, coeff_modulus, t):
// NTTTables (seal/util/ntt.h:69-183), BatchEncoder index map (seal/batchencoder.h),
// RNSTool's BEHZ base (seal/util/rns.h:324-399), the BFV scaling-variant constants
// (seal/context.h:350-401).  See SURVEY.md Appendix A for the restated arithmetic.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include "hhe_internal.h"
#include "../../include/hhe_gfx950.h"

typedef unsigned __int128 u128;

static thread_local std::string g_err;
void hhe_set_error(const std::string &msg) { g_err = msg; }
extern "C" const char *hhe_last_error(void) { return g_err.c_str(); }
extern "C" const char *hhe_backend(void) { return rt_backend_name(); }

// ------------------------------------------------------------------ number theory
u64 nt_mulmod(u64 a, u64 b, u64 m) { return (u64)((u128)a * b % m); }
u64 nt_powmod(u64 a, u64 e, u64 m)
{
    u64 r = 1 % m, x = a % m;
    for (; e; e >>= 1, x = nt_mulmod(x, x, m))
        if (e & 1) r = nt_mulmod(r, x, m);
    return r;
}
u64 nt_invmod(u64 a, u64 m)
{
    __int128 r0 = m, r1 = a % m, s0 = 0, s1 = 1;
    while (r1 != 0) {
        __int128 k = r0 / r1;
        __int128 r2 = r0 - k * r1; r0 = r1; r1 = r2;
        __int128 s2 = s0 - k * s1; s0 = s1; s1 = s2;
    }
    return (u64)(s0 < 0 ? s0 + m : s0);
}
bool nt_is_prime(u64 v)
{
    if (v < 2) return false;
    for (u64 p : {2ull, 3ull, 5ull, 7ull, 11ull, 13ull, 17ull, 19ull, 23ull, 29ull, 31ull, 37ull}) {
        if (v == p) return true;
        if (v % p == 0) return false;
    }
    u64 d = v - 1;
    int s = 0;
    while ((d & 1) == 0) { d >>= 1; ++s; }
    // bases 2..37 are a deterministic Miller-Rabin certificate below 2^64
    for (u64 base : {2ull, 3ull, 5ull, 7ull, 11ull, 13ull, 17ull, 19ull, 23ull, 29ull, 31ull, 37ull}) {
        u64 x = nt_powmod(base, d, v);
        if (x == 1 || x == v - 1) continue;
        bool witness = true;
        for (int i = 1; i < s && witness; ++i) {
            x = nt_mulmod(x, x, v);
            if (x == v - 1) witness = false;
        }
        if (witness) return false;
    }
    return true;
}
// util::get_primes (seal/util/numth.h:138-139): descending from the top of the bit size
bool nt_get_primes(u64 factor, int bits, size_t count, std::vector<u64> &out)
{
    out.clear();
    u64 cand = (((u64)1 << bits) - 1) / factor * factor + 1;
    const u64 floor_v = (u64)1 << (bits - 1);
    for (; out.size() < count && cand > floor_v; cand -= factor)
        if (nt_is_prime(cand)) out.push_back(cand);
    return out.size() == count;
}
// util::try_minimal_primitive_root (seal/util/numth.h:155-157)
u64 nt_minimal_primitive_root(u64 degree, u64 q)
{
    const u64 cofactor = (q - 1) / degree;
    u64 root = 0;
    for (u64 g = 2; g < q; ++g) {
        root = nt_powmod(g, cofactor, q);
        if (nt_powmod(root, degree / 2, q) == q - 1) break;
    }
    // all primitive roots are the odd powers of one of them
    const u64 step = nt_mulmod(root, root, q);
    u64 best = root, cur = root;
    for (u64 i = 1; i < degree / 2; ++i) {
        cur = nt_mulmod(cur, step, q);
        if (cur < best) best = cur;
    }
    return best;
}
// util::naf (seal/util/numth.h:22-42)
std::vector<int> nt_naf(int value)
{
    std::vector<int> terms;
    const bool neg = value < 0;
    unsigned v = (unsigned)std::abs(value);
    for (int bit = 0; v; ++bit) {
        int z = (v & 1u) ? 2 - (int)(v & 3u) : 0;
        v = (unsigned)((int)v - z) >> 1;
        if (z) terms.push_back((neg ? -z : z) * (1 << bit));
    }
    return terms;
}
// GaloisTool::get_elt_from_step (seal/util/galois.h:124)
u32 galois_elt_from_step(const hhe_ctx *c, int step)
{
    const u64 m = 2 * (u64)c->n;
    if (step == 0) return (u32)(m - 1);
    const u64 mag = (u64)std::abs(step);
    if (mag >= c->n / 2) return 0;
    const u64 e = step > 0 ? mag : c->n / 2 - mag;
    return (u32)nt_powmod(3, e, m);
}

static u64 bit_reverse(u64 v, int bits)
{
    u64 r = 0;
    for (int i = 0; i < bits; ++i, v >>= 1) r = (r << 1) | (v & 1);
    return r;
}
static u64 shoup_quot(u64 w, u64 q) { return (u64)(((u128)w << 64) / q); }
static u64 product_mod(const std::vector<u64> &v, int skip, u64 m)
{
    u64 r = 1 % m;
    for (int i = 0; i < (int)v.size(); ++i)
        if (i != skip) r = nt_mulmod(r, v[i] % m, m);
    return r;
}

// ------------------------------------------------------------------ context
static void fill_mod(ModDev &md, u64 q, int logn, u64 t, bool with_tables, u64 *host_tab, const u64 *dev_tab, u64 *root_out)
{
    memset(&md, 0, sizeof(md));
    md.q = q;
    md.nq = 0 - q;
    const u128 two64 = (u128)1 << 64;
    md.r_hi = (u64)(two64 / q);
    md.r_lo = (u64)(((two64 % q) << 64) / q);
    {   // pseudo-Mersenne form q = 2^b - c (hhe_common.h ModDev::pm_*)
        int b = 0;
        while (b < 64 && (q >> b)) ++b;
        const u64 c = b < 64 ? ((u64)1 << b) - q : 0;
        if (b >= 33 && b <= 60 && c < ((u64)1 << 32) && (u128)c * (((u128)1 << (64 - b)) + 2) <= ((u128)1 << b)) {
            md.pm_sh = (u32)(b - 32); md.pm_mask = ((u32)1 << (b - 32)) - 1; md.pm_c = (u32)c; md.pm_ok = 1;
        }
    }
    if (!with_tables) return;
    const size_t n = (size_t)1 << logn;
    const u64 psi = nt_minimal_primitive_root(2 * n, q), ipsi = nt_invmod(psi, q);
    if (root_out) *root_out = psi;
    u64 *w = host_tab, *ws = host_tab + n, *iw = host_tab + 2 * n, *fw = host_tab + 4 * n;
    u64 pw = 1, ipw = 1;
    for (size_t k = 0; k < n; ++k) {
        const size_t r = bit_reverse(k, logn);
        w[r] = pw; ws[r] = shoup_quot(pw, q);
        fw[2 * r] = w[r]; fw[2 * r + 1] = ws[r];
        iw[2 * r] = ipw; iw[2 * r + 1] = shoup_quot(ipw, q);
        pw = nt_mulmod(pw, psi, q);
        ipw = nt_mulmod(ipw, ipsi, q);
    }
    md.ninv = nt_invmod(n % q, q);
    md.ninv_s = shoup_quot(md.ninv, q);
    md.ninv_t = nt_mulmod(md.ninv, t % q, q);
    md.ninv_t_s = shoup_quot(md.ninv_t, q);
    md.w = dev_tab; md.ws = dev_tab + n; md.iw = dev_tab + 2 * n; md.fw = dev_tab + 4 * n;
}

// CoeffModulus::BFVDefault(N) (seal/coeffmodulus via util/globals.cpp, SEAL 4.0.0) as used by
// SEALZpCipher::create_context (src/pasta/SEAL_Cipher.cpp:62-65) and the hard-coded N = 65536 chain (:50-60).
// N = 16384 is pinned by SURVEY A.10; every list is prime, = 1 mod 2N, and sums to the bit count of
// seal/util/hestdparms.h:seal_he_std_parms_128_tc (tests/test_abi.py).
extern "C" int hhe_bfv_default_coeff_modulus(size_t n, uint64_t *out, size_t *count)
{
    static const std::vector<u64> t1024 = {0x7e00001ULL}, t2048 = {0x3fffffff000001ULL},
        t4096 = {0xffffee001ULL, 0xffffc4001ULL, 0x1ffffe0001ULL},
        t8192 = {0x7fffffd8001ULL, 0x7fffffc8001ULL, 0xfffffffc001ULL, 0xffffff6c001ULL, 0xfffffebc001ULL},
        t16384 = {0xfffffffd8001ULL, 0xfffffffa0001ULL, 0xfffffff00001ULL, 0x1fffffff68001ULL, 0x1fffffff50001ULL,
                  0x1ffffffee8001ULL, 0x1ffffffea0001ULL, 0x1ffffffe88001ULL, 0x1ffffffe48001ULL},
        t32768 = {0x7fffffffe90001ULL, 0x7fffffffbf0001ULL, 0x7fffffffbd0001ULL, 0x7fffffffba0001ULL, 0x7fffffffaa0001ULL,
                  0x7fffffffa50001ULL, 0x7fffffff9f0001ULL, 0x7fffffff7e0001ULL, 0x7fffffff770001ULL, 0x7fffffff380001ULL,
                  0x7fffffff330001ULL, 0x7fffffff2d0001ULL, 0x7fffffff170001ULL, 0x7fffffff150001ULL, 0x7ffffffef00001ULL,
                  0xfffffffff70001ULL},
        t65536 = {0xffffffffffc0001ULL, 0xfffffffff840001ULL, 0xfffffffff6a0001ULL, 0xfffffffff5a0001ULL, 0xfffffffff2a0001ULL,
                  0xfffffffff240001ULL, 0xffffffffefe0001ULL, 0xffffffffeca0001ULL, 0xffffffffe9e0001ULL, 0xffffffffe7c0001ULL,
                  0xffffffffe740001ULL, 0xffffffffe520001ULL, 0xffffffffe4c0001ULL, 0xffffffffe440001ULL, 0xffffffffe400001ULL,
                  0xffffffffdda0001ULL, 0xffffffffdd20001ULL, 0xffffffffdbc0001ULL, 0xffffffffdb60001ULL, 0xffffffffd8a0001ULL,
                  0xffffffffd840001ULL, 0xffffffffd6e0001ULL, 0xffffffffd680001ULL, 0xffffffffd2a0001ULL, 0xffffffffd000001ULL,
                  0xffffffffcf00001ULL, 0xffffffffcea0001ULL, 0xffffffffcdc0001ULL, 0xffffffffcc40001ULL};
    const std::vector<u64> *t = nullptr;
    switch (n) {
    case 1024: t = &t1024; break;
    case 2048: t = &t2048; break;
    case 4096: t = &t4096; break;
    case 8192: t = &t8192; break;
    case 16384: t = &t16384; break;
    case 32768: t = &t32768; break;
    case 65536: t = &t65536; break;
    default: hhe_set_error("non-standard poly_modulus_degree"); return HHE_ERR_INVALID;
    }
    if (!count) return HHE_ERR_INVALID;
    if (out && *count >= t->size()) memcpy(out, t->data(), t->size() * 8);
    *count = t->size();
    return HHE_OK;
}

extern "C" int hhe_ctx_create(int logn, int K, const uint64_t *q, uint64_t t, int device, hhe_ctx **out)
{
    if (!out || !q || logn < 10 || logn > 16 || K < 2 || K > HHE_MAXK) {
        hhe_set_error("hhe_ctx_create: invalid arguments (need 10 <= logn <= 16, 2 <= K <= 33)");
        return HHE_ERR_INVALID;
    }
    const size_t n = (size_t)1 << logn;
    for (int i = 0; i < K; ++i)
        if (q[i] >> 61 || (q[i] - 1) % (2 * n) != 0 || !nt_is_prime(q[i])) {
            hhe_set_error("hhe_ctx_create: coeff modulus must be primes < 2^61 congruent to 1 mod 2N");
            return HHE_ERR_INVALID;
        }
    if (t >> 61 || (t - 1) % (2 * n) != 0 || !nt_is_prime(t)) {
        hhe_set_error("hhe_ctx_create: plain modulus must be a prime congruent to 1 mod 2N (batching)");
        return HHE_ERR_INVALID;
    }
    if (rt_set_device(device)) { hhe_set_error(rt_last_error()); return HHE_ERR_DEVICE; }
    hhe_ctx *c = new hhe_ctx();
    c->keys0.ctx = c;
    for (auto &ks : c->rk_slots) ks.ctx = c;
    c->logn = logn; c->n = n; c->K = K; c->L = K - 1; c->t = t; c->device = device;
    c->q.assign(q, q + K);
    const int L = c->L;
    // BEHZ base (seal/util/rns.h): get_primes(2N, 61, L+2) = m_sk, gamma, B_0..B_{L-1}
    std::vector<u64> aux;
    if (!nt_get_primes(2 * n, 61, (size_t)L + 2, aux)) { delete c; hhe_set_error("no BEHZ primes"); return HHE_ERR_INVALID; }
    c->gamma = aux[1];
    c->bsk.assign(aux.begin() + 2, aux.end());
    c->bsk.push_back(aux[0]);
    c->nmod = K + (L + 1) + 1;
    c->mod_t = K + L + 1;

    std::vector<u64> allq(c->q);
    allq.insert(allq.end(), c->bsk.begin(), c->bsk.end());
    allq.push_back(t);
    std::vector<ModDev> mods(c->nmod);
    c->pm_ok.assign(c->nmod, 0);
    std::vector<u64> host_tab((size_t)c->nmod * 6 * n);
    c->d_tables = (u64 *)rt_malloc(host_tab.size() * 8);
    c->d_mods = (ModDev *)rt_malloc(sizeof(ModDev) * c->nmod);
    c->roots.resize(K);
    for (int i = 0; i < c->nmod; ++i)
        fill_mod(mods[i], allq[i], logn, t, true, host_tab.data() + (size_t)i * 6 * n, c->d_tables + (size_t)i * 6 * n,
                 i < K ? &c->roots[i] : nullptr);
    for (int i = 0; i < c->nmod; ++i) c->pm_ok[i] = (int)mods[i].pm_ok;
    // BatchEncoder matrix_reps_index_map
    c->slot_map.resize(n);
    {
        const u64 m = 2 * n;
        u64 pos = 1;
        for (size_t i = 0; i < n / 2; ++i) {
            c->slot_map[i] = (u32)bit_reverse((pos - 1) >> 1, logn);
            c->slot_map[n / 2 + i] = (u32)bit_reverse((m - pos - 1) >> 1, logn);
            pos = pos * 3 % m;
        }
    }
    c->d_slot_map = (u32 *)rt_malloc(4 * n);

    std::vector<u64> dq(c->q.begin(), c->q.begin() + L), Bq(c->bsk.begin(), c->bsk.begin() + L);
    const u64 msk = c->bsk[L], MT = (u64)1 << 32;
    // add_plain scaling variant
    AddPlainArgs &ap = c->apl;
    memset(&ap, 0, sizeof(ap));
    ap.t = t; ap.q_mod_t = product_mod(dq, -1, t); ap.thr = (t + 1) >> 1;
    {
        const u128 two64 = (u128)1 << 64;
        ap.t_r_hi = (u64)(two64 / t);
        ap.t_r_lo = (u64)(((two64 % t) << 64) / t);
    }
    for (int j = 0; j < L; ++j) {
        // floor(Q/t) mod q_j = -(Q mod t) * t^-1 mod q_j  (Q = 0 mod q_j)
        const u64 v = nt_mulmod(ap.q_mod_t % dq[j], nt_invmod(t % dq[j], dq[j]), dq[j]);
        ap.delta[j] = v ? dq[j] - v : 0;
    }
    // key-switch mod-down
    KsFinishArgs &kf = c->ksf;
    memset(&kf, 0, sizeof(kf));
    const u64 qsp = c->q[K - 1];
    kf.half = qsp >> 1;
    for (int j = 0; j < L; ++j) {
        kf.half_mod[j] = kf.half % dq[j];
        kf.qsp_inv[j] = nt_invmod(qsp % dq[j], dq[j]);
        kf.qsp_inv_s[j] = shoup_quot(kf.qsp_inv[j], dq[j]);
    }
    // BEHZ constants
    BehzDev bz;
    memset(&bz, 0, sizeof(bz));
    bz.L = L; bz.msk = msk;
    for (int i = 0; i < L; ++i) {
        bz.inv_punct_q[i] = nt_invmod(product_mod(dq, i, dq[i]), dq[i]);
        bz.inv_punct_q_s[i] = shoup_quot(bz.inv_punct_q[i], dq[i]);
        bz.mt_mod_q[i] = MT % dq[i];
        for (int p = 0; p <= L; ++p) bz.punct_q_bsk[i][p] = product_mod(dq, i, c->bsk[p]);
        bz.punct_q_mt[i] = product_mod(dq, i, MT);
        bz.inv_punct_B[i] = nt_invmod(product_mod(Bq, i, Bq[i]), Bq[i]);
        for (int j = 0; j < L; ++j) bz.punct_B_q[i][j] = product_mod(Bq, i, dq[j]);
        bz.punct_B_msk[i] = product_mod(Bq, i, msk);
        bz.B_mod_q[i] = product_mod(Bq, -1, dq[i]);
        bz.neg_B_mod_q[i] = dq[i] - bz.B_mod_q[i];
    }
    bz.neg_inv_q_mt = (MT - nt_invmod(product_mod(dq, -1, MT), MT)) & (MT - 1);
    for (int p = 0; p <= L; ++p) {
        const u64 P = c->bsk[p];
        bz.q_mod_bsk[p] = product_mod(dq, -1, P);
        bz.inv_mt_bsk[p] = nt_invmod(MT % P, P);
        bz.inv_q_bsk[p] = nt_invmod(bz.q_mod_bsk[p], P);
    }
    bz.inv_B_msk = nt_invmod(product_mod(Bq, -1, msk), msk);
    c->d_behz = (BehzDev *)rt_malloc(sizeof(BehzDev));

    if (!c->d_tables || !c->d_mods || !c->d_slot_map || !c->d_behz ||
        rt_h2d(c->d_tables, host_tab.data(), host_tab.size() * 8, nullptr) ||
        rt_h2d(c->d_mods, mods.data(), sizeof(ModDev) * c->nmod, nullptr) ||
        rt_h2d(c->d_slot_map, c->slot_map.data(), 4 * n, nullptr) ||
        rt_h2d(c->d_behz, &bz, sizeof(bz), nullptr) || rt_sync(nullptr)) {
        hhe_set_error(std::string("hhe_ctx_create: device setup failed: ") + rt_last_error());
        hhe_ctx_destroy(c);
        return HHE_ERR_DEVICE;
    }
    c->ksc.half = kf.half;
    for (int j = 0; j < L; ++j) {
        c->ksc.half_mod[j] = kf.half_mod[j]; c->ksc.qsp_inv[j] = kf.qsp_inv[j]; c->ksc.qsp_inv_s[j] = kf.qsp_inv_s[j];
        c->ksc.qsp_mod[j] = qsp % dq[j]; c->ksc.qsp_mod_s[j] = shoup_quot(c->ksc.qsp_mod[j], dq[j]);
        c->ksc.ninv_qinv[j] = nt_mulmod(mods[j].ninv, kf.qsp_inv[j], dq[j]);
        c->ksc.ninv_qinv_s[j] = shoup_quot(c->ksc.ninv_qinv[j], dq[j]);
        c->ksc.hq2[j] = 2 * dq[j] + nt_mulmod(kf.half_mod[j], kf.qsp_inv[j], dq[j]);
    }
    if (const char *mm = getenv("HHE_MATMUL")) c->matmul_mode = atoi(mm);
    if (const char *e = getenv("HHE_BLOCK_CACHE_MB")) c->block_cache_limit = (size_t)std::max(0, atoi(e)) << 20;
    if (const char *e = getenv("HHE_FC_ROWFUSED")) c->fc_row_fused = atoi(e);
    if (const char *e = getenv("HHE_FC_CSUM")) c->fc_csum = atoi(e);
    if (const char *e = getenv("HHE_FC_C0HAT")) c->fc_c0hat = atoi(e);
    if (const char *e = getenv("HHE_FC_CSUMGROUP")) c->fc_csum_group = std::max(1, std::min(HHE_CSUM_GROUP, atoi(e)));
    if (const char *e = getenv("HHE_FC_LEAFSUM")) c->fc_leaf_sums = atoi(e);
    if (const char *e = getenv("HHE_FC_LEAFGROUP")) c->fc_leaf_group = std::max(1, std::min(HHE_LEAF_GROUP, atoi(e)));
    if (const char *e = getenv("HHE_FC_SHARED")) c->fc_shared = atoi(e);
    if (const char *e = getenv("HHE_FC_CHUNK")) c->fc_chunk = (size_t)std::max(0, atoi(e));
    if (const char *e = getenv("HHE_STREAMS")) c->nstreams = std::max(0, std::min(HHE_MAX_STREAMS, atoi(e)));
    if (const char *e = getenv("HHE_CHUNK")) c->chunk = (size_t)std::max(1, atoi(e));
    for (int s = 1; s <= c->nstreams; ++s) {
        c->lanes[s].stream = rt_stream_create();
        c->lanes[s].own_stream = true;
        c->lanes[s].ev_done = rt_event_create();
    }
    c->ev_fork = rt_event_create();
    {   // a digit d_I < q_I may enter NTT_J unreduced when q_I < 4 q_J (butterfly inputs live in [0,4q))
        u64 qmax = 0, qmin = ~(u64)0;
        for (int i = 0; i < L; ++i) qmax = std::max(qmax, c->q[i]);
        for (int i = 0; i < K; ++i) qmin = std::min(qmin, c->q[i]);
        c->digit_reduce = (qmax / 4 >= qmin) ? 1 : 0;
    }
    kf.mods = c->d_mods; kf.logn = logn; kf.L = L; kf.K = K;
    ap.mods = c->d_mods; ap.logn = logn; ap.L = L;
    *out = c;
    return HHE_OK;
}

static void free_lane(Lane &ln)
{
    rt_free(ln.ws_T); rt_free(ln.ws_S); rt_free(ln.ws_d); rt_free(ln.ws_ct3); rt_free(ln.ws_plain); rt_free(ln.ws_vals);
    for (auto &p : ln.ws_ct) { rt_free(p); p = nullptr; }
    rt_free(ln.ws_rot); ln.ws_rot = nullptr; ln.rot_cap = 0;
    for (auto &sl : ln.fc_slots) { rt_free(sl.tp); rt_free(sl.ct); rt_free(sl.c0hat); }
    ln.fc_slots.clear(); ln.fc_slot_cap = 0;
    for (u64 *p : ln.csum_bufs) rt_free(p);
    ln.csum_bufs.clear();
    rt_free(ln.ws_leaf); ln.ws_leaf = nullptr; ln.leaf_cap = 0;
    rt_free((void *)ln.d_ptrs); ln.d_ptrs = nullptr; ln.ptr_cap = 0;
    for (auto &pr : ln.prof_ev) { rt_event_destroy(pr.first); rt_event_destroy(pr.second); }
    ln.prof_ev.clear(); ln.prof_used = 0;
    rt_free(ln.bz_aq); rt_free(ln.bz_bq); rt_free(ln.bz_ab); rt_free(ln.bz_bb); rt_free(ln.bz_dq); rt_free(ln.bz_db);
    ln.ws_T = ln.ws_S = ln.ws_d = ln.ws_ct3 = ln.ws_plain = ln.ws_vals = nullptr;
    ln.bz_aq = ln.bz_bq = ln.bz_ab = ln.bz_bb = ln.bz_dq = ln.bz_db = nullptr;
    ln.cap = 0;
}
void sync_ctx(hhe_ctx *c)
{
    for (auto &ln : c->lanes) {
        if (&ln == &c->lanes[0] || ln.own_stream) rt_sync(ln.stream);
    }
}

int lane_reserve(hhe_ctx *c, Lane &ln, size_t B)
{
    if (B == 0) { hhe_set_error("empty batch"); return HHE_ERR_INVALID; }
    if (B <= ln.cap) return HHE_OK;
    sync_ctx(c);
    free_lane(ln);
    const size_t n = c->n, L = c->L, K = c->K;
    auto alloc = [&](size_t words) { return (u64 *)rt_malloc(words * 8); };
    bool ok = true;
    ok &= !!(ln.ws_T = alloc(B * L * K * n));
    ok &= !!(ln.ws_S = alloc(B * 2 * K * n));
    ok &= !!(ln.ws_d = alloc(B * L * n));
    for (auto &p : ln.ws_ct) ok &= !!(p = alloc(B * 2 * L * n));
    ok &= !!(ln.ws_ct3 = alloc(B * 3 * L * n));
    ok &= !!(ln.ws_plain = alloc(B * n));
    ok &= !!(ln.ws_vals = alloc(B * PASTA_T));
    ok &= !!(ln.bz_aq = alloc(B * 2 * L * n));
    ok &= !!(ln.bz_bq = alloc(B * 2 * L * n));
    ok &= !!(ln.bz_ab = alloc(B * 2 * (L + 1) * n));
    ok &= !!(ln.bz_bb = alloc(B * 2 * (L + 1) * n));
    ok &= !!(ln.bz_dq = alloc(B * 3 * L * n));
    ok &= !!(ln.bz_db = alloc(B * 3 * (L + 1) * n));
    ok &= !!(ln.d_ptrs = (const u64 **)rt_malloc(2 * B * sizeof(u64 *)));
    ln.ptr_cap = B;
    if (!ok) { free_lane(ln); hhe_set_error(std::string("workspace allocation failed: ") + rt_last_error()); return HHE_ERR_DEVICE; }
    ln.cap = B;
    return HHE_OK;
}

extern "C" void hhe_pasta3_clear_block_cache(hhe_ctx *c)
{
    HHE_LOCK(c);
    if (!c) return;
    sync_ctx(c);
    for (auto &kv : c->blocks) { rt_free(kv.second.diag); rt_free(kv.second.pdiag); rt_free(kv.second.rc); rt_free(kv.second.bsgs); }
    c->blocks.clear();
    c->block_bytes = 0;
}

extern "C" void hhe_ctx_destroy(hhe_ctx *c)
{
    if (!c) return;
    sync_ctx(c);
    hhe_pasta3_clear_block_cache(c);
    for (auto &ln : c->lanes) {
        free_lane(ln);
        rt_event_destroy(ln.ev_done);
        rt_event_destroy(ln.ev_stage);
        if (ln.own_stream) rt_stream_destroy(ln.stream);
    }
    rt_event_destroy(c->ev_fork);
    keyset_clear(&c->keys0);
    for (auto &ks : c->rk_slots) keyset_clear(&ks);
    for (hhe_keyset *ks : c->sets) { keyset_clear(ks); delete ks; }  // sets the caller did not destroy
    for (auto &kv : c->d_key_shoup) rt_free(kv.second);
    rt_free(c->d_feistel_mask);
    rt_free(c->d_zero_corr);
    rt_free(c->d_qsp_poly);
    rt_free(c->d_blocks); rt_free(c->d_flags);
    rt_free(c->d_tables); rt_free(c->d_mods); rt_free(c->d_behz); rt_free(c->d_slot_map);
    delete c;
}

extern "C" int hhe_ctx_set_stream(hhe_ctx *c, void *s)
{
    HHE_LOCK(c);
    if (!c) return HHE_ERR_INVALID;
    c->lanes[0].stream = (rt_stream)s;
    return HHE_OK;
}
extern "C" int hhe_ctx_sync(hhe_ctx *c)
{
    HHE_LOCK(c);
    if (!c) return HHE_ERR_INVALID;
    if (rt_sync(c->lanes[0].stream)) { hhe_set_error(rt_last_error()); return HHE_ERR_DEVICE; }
    return HHE_OK;
}

// Timing of the dominant kernel on the stream it is launched on (bench.py's roofline block): while enabled, every launch
// of ks_row_kernel is bracketed by a pair of timed HIP events.  Results are unaffected.
extern "C" int hhe_ctx_profile(hhe_ctx *c, int enable)
{
    HHE_LOCK(c);
    if (!c) return HHE_ERR_INVALID;
    sync_ctx(c);
    c->profile = enable ? 1 : 0;
    c->prof_items = 0;
    for (auto &ln : c->lanes) ln.prof_used = 0;
    return HHE_OK;
}
extern "C" int hhe_ctx_profile_read(hhe_ctx *c, char *kernel_name, size_t name_cap, uint64_t *launches, double *total_ms, uint64_t *items)
{
    HHE_LOCK(c);
    if (!c || !launches || !total_ms || !items) return HHE_ERR_INVALID;
    sync_ctx(c);
    uint64_t n = 0;
    double ms = 0;
    for (auto &ln : c->lanes) {
        for (size_t i = 0; i < ln.prof_used; ++i) {
            const float t = rt_event_elapsed_ms(ln.prof_ev[i].first, ln.prof_ev[i].second);
            if (t < 0) { hhe_set_error(rt_last_error()); return HHE_ERR_DEVICE; }
            ms += t;
            ++n;
        }
        ln.prof_used = 0;
    }
    *launches = n; *total_ms = ms; *items = c->prof_items;
    c->prof_items = 0;
    if (kernel_name && name_cap) {
        int n1, n2;
        ntt_split(c->logn, n1, n2);
        snprintf(kernel_name, name_cap, "void ks_row_kernel<%d>(NttArgs, KsRowArgs, NttArgs)", n2);  // as rocprofv3 prints it
    }
    return HHE_OK;
}

extern "C" int hhe_ctx_reserve(hhe_ctx *c, size_t B)
{
    HHE_LOCK(c);
    if (!c || B == 0) return HHE_ERR_INVALID;
    // generic ops run whole batches on lane 0; the transciphering path works in chunks on the internal lanes
    if (c->nstreams > 0) {
        const size_t per = std::min(B, c->chunk);
        for (int s = 1; s <= c->nstreams; ++s) {
            int rc = lane_reserve(c, c->lanes[s], per);
            if (rc) return rc;
        }
        return HHE_OK;
    }
    return lane_reserve(c, c->lanes[0], B);
}

extern "C" uint64_t hhe_ctx_query(const hhe_ctx *c, const char *what, int i)
{
    HHE_LOCK(c);
    if (!c || !what) return 0;
    const std::string w(what);
    if (w == "root" && i >= 0 && i < c->K) return c->roots[i];
    if (w == "bsk" && i >= 0 && i <= c->L) return c->bsk[i];
    if (w == "gamma") return c->gamma;
    if (w == "galois_elt") return galois_elt_from_step(c, i);
    if (w == "delta" && i >= 0 && i < c->L) return c->apl.delta[i];
    if (w == "slot_map" && i >= 0 && (size_t)i < c->n) return c->slot_map[i];
    if (w == "fc_fallbacks") return c->fc_fallbacks;
    if (w == "fc_csum_closes") return c->fc_csum_closes;
    if (w == "block_cache_bytes") return c->block_bytes;
    if (w == "block_cache_entries") return c->blocks.size();
    return 0;
}

// ------------------------------------------------------------------ key sets
// drop what was derived from the words at `key` (its Shoup table); every stream must be done with them first
static void forget_key(hhe_ctx *c, const u64 *key)
{
    auto sh = c->d_key_shoup.find(key);
    if (sh != c->d_key_shoup.end()) { rt_free(sh->second); c->d_key_shoup.erase(sh); }
}
// The kernels multiply key words through Shoup quotients, which are only right for words below their prime: a key that is not
// reduced is refused (SEAL's safe load does the same, is_data_valid_for).
static int check_key_words(const hhe_ctx *c, const uint64_t *ksk)
{
    const size_t n = c->n;
    for (size_t p = 0; p < (size_t)c->L * 2 * c->K; ++p) {
        const u64 qj = c->q[p % c->K];
        const uint64_t *w = ksk + p * n;
        u64 bad = 0;
        for (size_t i = 0; i < n; ++i) bad |= (u64)(w[i] >= qj);
        if (bad) { hhe_set_error("key-switch key data is not reduced modulo the coefficient primes"); return HHE_ERR_INVALID; }
    }
    return HHE_OK;
}
static int upload_key(hhe_ctx *c, u64 *&slot, const uint64_t *ksk)
{
    int rc = check_key_words(c, ksk);
    if (rc) return rc;
    sync_ctx(c);  // a resident key may still be read by work in flight on any lane
    if (slot) forget_key(c, slot);
    if (!slot) slot = (u64 *)rt_malloc(c->ksk_words() * 8);
    if (!slot || rt_h2d(slot, ksk, c->ksk_words() * 8, c->lanes[0].stream) || rt_sync(c->lanes[0].stream)) {
        hhe_set_error(std::string("key upload failed: ") + rt_last_error());
        return HHE_ERR_DEVICE;
    }
    return HHE_OK;
}
int keyset_put_relin(hhe_keyset *ks, const u64 *ksk) { return upload_key(ks->ctx, ks->rk, ksk); }
int keyset_put_galois(hhe_keyset *ks, u32 elt, const u64 *ksk)
{
    hhe_ctx *c = ks->ctx;
    if (!(elt & 1) || elt >= 2 * c->n) { hhe_set_error("invalid Galois element"); return HHE_ERR_INVALID; }
    int rc = check_key_words(c, ksk);
    if (rc) return rc;
    auto corr = ks->gk_corr.find(elt);  // derived from the key being replaced
    if (corr != ks->gk_corr.end()) { sync_ctx(c); rt_free(corr->second); ks->gk_corr.erase(corr); }
    auto it = ks->gk.find(elt);
    u64 *slot = it == ks->gk.end() ? nullptr : it->second;
    const bool fresh = slot == nullptr;
    rc = upload_key(c, slot, ksk);
    if (rc && fresh) { rt_free(slot); return rc; }  // a key that did not arrive is not registered
    ks->gk[elt] = slot;
    return rc;
}
void keyset_clear(hhe_keyset *ks)
{
    hhe_ctx *c = ks->ctx;
    if (ks->rk) { forget_key(c, ks->rk); rt_free(ks->rk); ks->rk = nullptr; }
    for (auto &kv : ks->gk) { forget_key(c, kv.second); rt_free(kv.second); }
    for (auto &kv : ks->gk_corr) rt_free(kv.second);
    ks->gk.clear();
    ks->gk_corr.clear();
}
extern "C" int hhe_keyset_create(hhe_ctx *c, hhe_keyset **out)
{
    HHE_LOCK(c);
    if (!c || !out) { hhe_set_error("hhe_keyset_create: null argument"); return HHE_ERR_INVALID; }
    hhe_keyset *ks = new hhe_keyset();
    ks->ctx = c;
    c->sets.push_back(ks);
    *out = ks;
    return HHE_OK;
}
extern "C" void hhe_keyset_destroy(hhe_keyset *ks)
{
    if (!ks) return;
    hhe_ctx *c = ks->ctx;
    HHE_LOCK(c);
    sync_ctx(c);
    keyset_clear(ks);
    c->sets.erase(std::remove(c->sets.begin(), c->sets.end(), ks), c->sets.end());
    delete ks;
}
extern "C" int hhe_keyset_set_relin(hhe_keyset *ks, const uint64_t *ksk)
{
    if (!ks || !ksk) { hhe_set_error("hhe_keyset_set_relin: null argument"); return HHE_ERR_INVALID; }
    HHE_LOCK(ks->ctx);
    return keyset_put_relin(ks, ksk);
}
extern "C" int hhe_keyset_set_galois(hhe_keyset *ks, uint32_t elt, const uint64_t *ksk)
{
    if (!ks || !ksk) { hhe_set_error("hhe_keyset_set_galois: null argument"); return HHE_ERR_INVALID; }
    HHE_LOCK(ks->ctx);
    return keyset_put_galois(ks, elt, ksk);
}
extern "C" int hhe_keyset_has_galois(const hhe_keyset *ks, uint32_t elt)
{
    if (!ks) return 0;
    HHE_LOCK(ks->ctx);
    return ks->gk.count(elt) ? 1 : 0;
}
extern "C" int hhe_keyset_has_relin(const hhe_keyset *ks)
{
    if (!ks) return 0;
    HHE_LOCK(ks->ctx);
    return ks->rk ? 1 : 0;
}
// the context's default set (the keys PASTA_SEAL was constructed with) and its further relin slots
extern "C" int hhe_set_relin_key_slot(hhe_ctx *c, int slot, const uint64_t *ksk)
{
    HHE_LOCK(c);
    if (!c || !ksk || slot < 0 || slot >= HHE_RELIN_SLOTS) return HHE_ERR_INVALID;
    return keyset_put_relin(c->relin_set(slot), ksk);
}
extern "C" int hhe_set_relin_key(hhe_ctx *c, const uint64_t *ksk) { HHE_LOCK(c); return hhe_set_relin_key_slot(c, 0, ksk); }
extern "C" int hhe_set_galois_key(hhe_ctx *c, uint32_t elt, const uint64_t *ksk)
{
    HHE_LOCK(c);
    if (!c || !ksk) { hhe_set_error("hhe_set_galois_key: null argument"); return HHE_ERR_INVALID; }
    return keyset_put_galois(&c->keys0, elt, ksk);
}
extern "C" int hhe_has_galois_key(const hhe_ctx *c, uint32_t elt) { HHE_LOCK(c); return c && c->keys0.gk.count(elt) ? 1 : 0; }

extern "C" void *hhe_malloc(size_t bytes) { return rt_malloc(bytes); }
extern "C" void hhe_free(void *p) { rt_free(p); }
extern "C" int hhe_copy_h2d(hhe_ctx *c, void *d, const void *h, size_t bytes)
{
    HHE_LOCK(c);
    if (rt_h2d(d, h, bytes, c ? c->lanes[0].stream : nullptr) || rt_sync(c ? c->lanes[0].stream : nullptr)) { hhe_set_error(rt_last_error()); return HHE_ERR_DEVICE; }
    return HHE_OK;
}
extern "C" int hhe_copy_d2h(hhe_ctx *c, void *h, const void *d, size_t bytes)
{
    HHE_LOCK(c);
    if (rt_d2h(h, d, bytes, c ? c->lanes[0].stream : nullptr) || rt_sync(c ? c->lanes[0].stream : nullptr)) { hhe_set_error(rt_last_error()); return HHE_ERR_DEVICE; }
    return HHE_OK;
}
