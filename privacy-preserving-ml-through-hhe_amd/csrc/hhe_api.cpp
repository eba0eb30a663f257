// hhe_api.cpp -- C ABI of libhhe_gfx950.so: batched BFV evaluator ops on device buffers
// and the PASTA-3 transciphering / FC schedules built from them.  Host code only decides
// the op order (the reference's schedule, src/pasta/pasta_3_seal.cpp); all arithmetic runs
// in the gfx950 kernels of hhe_kernels.hip.
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include "hhe_internal.h"
#include "../../include/hhe_gfx950.h"

namespace {

int fail(int code, const std::string &msg) { hhe_set_error(msg); return code; }
// the fused key-switch row kernel runs when the row pass has one of its sizes AND every key-level modulus has the pseudo-Mersenne
// form its arithmetic is written for (SEAL's own primes do); other contexts take the separate-kernel path (any N, any primes < 2^61)
bool use_row_kernel(const hhe_ctx *c) { return k_ks_row_supported(c->logn) && ntt_lazy8(c, 0, c->K); }
int dev_fail(const char *where) { return fail(HHE_ERR_DEVICE, std::string(where) + ": " + rt_last_error()); }

int need(hhe_ctx *c, size_t B)
{
    if (!c || B == 0) return fail(HHE_ERR_INVALID, "null context or empty batch");
    c->w = &c->lanes[0];
    return lane_reserve(c, c->lanes[0], B);
}

// host staging of a lane: wait for the previous staged copy before the buffers are overwritten; mark the new one
void stage_begin(Lane &ln)
{
    if (ln.stage_pending && ln.ev_stage) rt_event_sync(ln.ev_stage);
    ln.stage_pending = false;
}
void stage_end(Lane &ln)
{
    if (!ln.ev_stage) ln.ev_stage = rt_event_create();
    if (ln.ev_stage && !rt_event_record(ln.ev_stage, ln.stream)) ln.stage_pending = true;
    else rt_sync(ln.stream);
}

NttArgs ntt_args(const hhe_ctx *c, const u64 *src, u64 *dst, size_t count, int mod_base, int mod_cycle)
{
    NttArgs a;
    memset(&a, 0, sizeof(a));
    a.src = src; a.dst = dst; a.mods = c->d_mods; a.logn = c->logn; a.count = (int)count;
    a.mod_base = mod_base; a.mod_cycle = mod_cycle; a.src_div = 1; a.t = c->t;
    a.load_op = LOAD_PLAIN; a.store_op = STORE_PLAIN; a.mul_cycle = 1; a.mul_item_polys = 1;
    a.L = c->L; a.K = c->K; a.ks = c->ksc;
    a.lazy8 = ntt_lazy8(c, mod_base, mod_cycle);
    return a;
}
void op_ntt(hhe_ctx *c, u64 *polys, size_t count, int mod_base, int mod_cycle, bool inverse, int store_op = STORE_PLAIN)
{
    NttArgs a = ntt_args(c, polys, polys, count, mod_base, mod_cycle);
    a.store_op = store_op;
    k_ntt(a, inverse, c->w->stream);
}
void op_elt(hhe_ctx *c, int op, const u64 *x, const u64 *y, u64 *out, size_t count, int mod_base, int mod_cycle, int b_cycle = 0)
{
    EltArgs a;
    memset(&a, 0, sizeof(a));
    a.a = x; a.b = y; a.out = out; a.mods = c->d_mods; a.logn = c->logn; a.count = (int)count;
    a.mod_base = mod_base; a.mod_cycle = mod_cycle; a.b_cycle = b_cycle;
    k_elt(a, op, c->w->stream);
}
// ct (+) ct over [B][size][L][N]
void op_add(hhe_ctx *c, const u64 *x, const u64 *y, u64 *out, size_t B, int size) { op_elt(c, ELT_ADD, x, y, out, B * size * c->L, 0, c->L); }

void op_add_plain(hhe_ctx *c, const u64 *ct, const u64 *plain, const u64 *const *plain_ptrs, size_t shift, bool bcast,
                  bool subtract, bool negate, u64 *out, size_t B)
{
    AddPlainArgs a = c->apl;
    a.ct = ct; a.plain = plain; a.plain_ptrs = plain_ptrs; a.plain_shift = shift; a.out = out; a.B = (int)B;
    a.plain_bcast = bcast; a.subtract = subtract; a.negate_ct = negate;
    k_add_plain(a, c->w->stream);
}

// BatchEncoder::encode: vals [B][stride] (first `count` used) -> plain [B][N]
void op_encode(hhe_ctx *c, const u64 *vals, size_t B, int stride, int count, int second_off, u64 *plain)
{
    rt_memset(plain, 0, B * c->n * 8, c->w->stream);
    EncodeArgs e;
    memset(&e, 0, sizeof(e));
    e.vals = vals; e.out = plain; e.slot_map = c->d_slot_map; e.logn = c->logn; e.B = (int)B;
    e.stride = stride; e.count = count; e.second_off = second_off; e.t = c->t;
    k_encode_scatter(e, c->w->stream);
    op_ntt(c, plain, B, c->mod_t, 1, true);
}

// plain [P][N] (coefficients mod t) -> lifted NTT form [P][L][N]   
void op_lift_ntt(hhe_ctx *c, const u64 *plain, size_t P, u64 *out)
{
    NttArgs a = ntt_args(c, plain, out, P * c->L, 0, c->L);
    a.src_div = c->L; a.load_op = LOAD_LIFT;
    k_ntt(a, false, c->w->stream);
}

// out = INTT(NTT(ct) * D) with D an NTT-form lifted plaintext: shared [L][N] (ptrs null) or per item
void op_multiply_plain_ntt(hhe_ctx *c, const u64 *ct, const u64 *D, const u64 *const *D_ptrs, size_t shift, u64 *out, size_t B)
{
    NttArgs a = ntt_args(c, ct, out, B * 2 * c->L, 0, c->L);
    a.store_op = STORE_MUL; a.mul = D; a.mul_ptrs = D_ptrs; a.mul_shift = shift; a.mul_cycle = c->L; a.mul_item_polys = 2 * c->L;
    k_ntt(a, false, c->w->stream);
    op_ntt(c, out, B * 2 * c->L, 0, c->L, true);
}

// Shoup quotients floor(key * 2^64 / q_J) of a key-switch key ([L][2][K][N], as the key), built on first use by the
// fused row kernel and cached per key (by its device address; dropped when the key is replaced)
int ensure_key_shoup(hhe_ctx *c, const u64 *key, const u64 **out)
{
    auto it = c->d_key_shoup.find(key);
    if (it != c->d_key_shoup.end()) { *out = it->second; return HHE_OK; }
    DevBuf t(c->ksk_words() * 8);
    if (!t.p) return dev_fail("key Shoup table alloc");
    op_elt(c, ELT_SHOUP, key, nullptr, t.w(), (size_t)c->L * 2 * c->K, 0, c->K);
    if (rt_sync(c->w->stream)) return dev_fail("key Shoup table");
    *out = c->d_key_shoup[key] = t.release();
    return HHE_OK;
}

// Evaluator::switch_key_inplace core.  d: item b at d + b*d_stride, [L][N] coefficient form.
// out[b] = (base ? base polys selected by mask : 0) + key-switched pair.
// galois_einv > 0 (N >= 4096 only): d and base are the UN-rotated polynomials of a rotation; the digit loads and the base
// fetch of the fused mod-down read them through the Galois map, so no galois_kernel launch and no rotated copy is needed
int op_switch_key(hhe_ctx *c, const u64 *d, size_t d_stride, const u64 *key, const u64 *base, size_t base_stride,
                  int base_mask, u64 *out, size_t B, u32 galois_einv = 0)
{
    const int L = c->L, K = c->K;
    NttArgs a = ntt_args(c, d, c->w->ws_T, B * L * K, 0, K);
    a.src_div = K; a.src_item_polys = L * K; a.src_item_stride = d_stride; a.load_op = LOAD_DIGIT; a.digit_reduce = c->digit_reduce;
    a.store_op = STORE_LAZY;  // ks_mac reduces: digits may stay in [0,4q)
    a.load_einv = galois_einv;
    const u64 *key_s = nullptr;
    const bool rowk = use_row_kernel(c) && ensure_key_shoup(c, key, &key_s) == HHE_OK;
    if (galois_einv && !rowk) return fail(HHE_ERR_DEVICE, "switch_key: Shoup table of the key could not be built");
    if (rowk) {
        // N >= 4096: strided pass of the digit transforms, then ONE kernel for their row pass, the key inner product and
        // the inverse row pass of all 2K sums (ks_row_kernel, as in the matmul loop), then the strided inverse passes with
        // the mod-down fused into the store of the data limbs -- T and S never make a round trip
        const size_t n = c->n;
        u64 *W = c->w->ws_S, *Usp = c->w->ws_S + B * 2 * L * n;  // [B][2][L][N] | [B][2][N] inside [B][2][K][N]
        k_ntt_pass(a, false, false, c->w->stream);
        KsRowArgs x;
        memset(&x, 0, sizeof(x));
        x.key = key; x.key_s = key_s; x.U0 = W; x.U1 = W + (size_t)L * n; x.u_stride = (size_t)2 * L * n; x.Usp = Usp;
        x.B = (int)B; x.L = L; x.K = K;
        if (k_ks_row(a, x, nullptr, c->w->stream)) return dev_fail("switch_key");
        NttArgs as = ntt_args(c, Usp, Usp, B * 2, K - 1, 1);
        as.store_op = STORE_RSP;
        k_ntt_pass(as, true, true, c->w->stream);
        NttArgs ad = ntt_args(c, W, W, B * 2 * L, 0, L);
        ad.store_op = STORE_KSF; ad.aux_r = Usp; ad.aux_in = base; ad.base_stride = base_stride; ad.base_mask = base ? base_mask : 0;
        ad.aux_out = out; ad.gal_einv = galois_einv;
        k_ntt_pass(ad, true, true, c->w->stream);
        return HHE_OK;
    }
    k_ntt(a, false, c->w->stream);
    KsMacArgs m;
    memset(&m, 0, sizeof(m));
    m.T = c->w->ws_T; m.key = key; m.S = c->w->ws_S; m.mods = c->d_mods; m.logn = c->logn; m.B = (int)B; m.L = L; m.K = K;
    k_ks_mac(m, c->w->stream);
    op_ntt(c, c->w->ws_S, B * 2 * K, 0, K, true);
    KsFinishArgs f = c->ksf;
    f.S = c->w->ws_S; f.base = base; f.base_item_stride = base_stride; f.base_mask = base ? base_mask : 0; f.out = out; f.B = (int)B;
    k_ks_finish(f, c->w->stream);
    return HHE_OK;
}

int op_apply_galois(hhe_ctx *c, const u64 *ct, u32 elt, u64 *out, size_t B)
{
    auto it = c->gks->gk.find(elt);
    if (it == c->gks->gk.end()) return fail(HHE_ERR_NO_GALOIS_KEY, "Galois key not present");
    const int L = c->L;
    const size_t n = c->n;
    const u32 einv = (u32)nt_invmod(elt, 2 * n);
    const u64 *src = ct;
    if (ct == out) {  // gathers cannot run in place
        rt_d2d(c->w->ws_ct[3], ct, B * c->ct_words() * 8, c->w->stream);
        src = c->w->ws_ct[3];
    }
    if (use_row_kernel(c)) {
        // N >= 4096: no rotated copies -- the digit loads read c1 and the fused mod-down reads c0 through the Galois map
        const u64 *key_s = nullptr;
        int rc = ensure_key_shoup(c, it->second, &key_s);
        if (rc) return rc;
        return op_switch_key(c, src + L * n, 2 * L * n, it->second, src, 2 * L * n, 1, out, B, einv);
    }
    GaloisArgs g;
    memset(&g, 0, sizeof(g));
    g.mods = c->d_mods; g.logn = c->logn; g.count = (int)(B * L); g.L = L;
    g.einv = einv;
    // c0' = galois(c0) -> out poly 0 ; d = galois(c1) -> ws_d
    g.in = src; g.in_item_stride = 2 * L * n; g.out = out; g.out_item_stride = 2 * L * n;
    k_galois(g, c->w->stream);
    g.in = src + L * n; g.out = c->w->ws_d; g.out_item_stride = L * n;
    k_galois(g, c->w->stream);
    return op_switch_key(c, c->w->ws_d, L * n, it->second, out, 2 * L * n, 1, out, B);
}

// Evaluator::rotate_internal (seal/evaluator.h:1234)
int op_rotate_rows(hhe_ctx *c, const u64 *ct, int step, u64 *out, size_t B)
{
    if (step == 0) {
        if (ct != out) rt_d2d(out, ct, B * c->ct_words() * 8, c->w->stream);
        return HHE_OK;
    }
    const u32 elt = galois_elt_from_step(c, step);
    if (!elt) return fail(HHE_ERR_INVALID, "step count too large");
    if (c->gks->gk.count(elt)) return op_apply_galois(c, ct, elt, out, B);
    const std::vector<int> terms = nt_naf(step);
    if (terms.size() == 1) return fail(HHE_ERR_NO_GALOIS_KEY, "Galois key not present");
    const u64 *cur = ct;
    for (int term : terms) {
        if ((size_t)std::abs(term) == c->n / 2) continue;
        int rc = op_rotate_rows(c, cur, term, out, B);
        if (rc) return rc;
        cur = out;
    }
    if (cur == ct && ct != out) rt_d2d(out, ct, B * c->ct_words() * 8, c->w->stream);
    return HHE_OK;
}

// Evaluator::bfv_multiply (BEHZ): a, b [B][2][L][N] -> out3 [B][3][L][N]
void op_multiply(hhe_ctx *c, const u64 *x, const u64 *y, u64 *out3, size_t B)
{
    const int L = c->L, K = c->K;
    auto extend = [&](const u64 *in, u64 *oq, u64 *ob) {
        BehzExtendArgs e;
        memset(&e, 0, sizeof(e));
        e.x = in; e.xb = ob; e.mods = c->d_mods; e.bz = c->d_behz; e.logn = c->logn; e.P = (int)(B * 2); e.L = L; e.K = K;
        k_behz_extend(e, c->w->stream);
        op_ntt(c, ob, B * 2 * (L + 1), K, L + 1, false);
        NttArgs a = ntt_args(c, in, oq, B * 2 * L, 0, L);
        k_ntt(a, false, c->w->stream);
    };
    extend(x, c->w->bz_aq, c->w->bz_ab);
    const u64 *bq = c->w->bz_aq, *bb = c->w->bz_ab;
    if (y != x) { extend(y, c->w->bz_bq, c->w->bz_bb); bq = c->w->bz_bq; bb = c->w->bz_bb; }
    TensorArgs t;
    memset(&t, 0, sizeof(t));
    t.mods = c->d_mods; t.logn = c->logn; t.B = (int)B;
    t.a = c->w->bz_aq; t.b = bq; t.d = c->w->bz_dq; t.limbs = L; t.mod_base = 0;
    k_tensor(t, c->w->stream);
    t.a = c->w->bz_ab; t.b = bb; t.d = c->w->bz_db; t.limbs = L + 1; t.mod_base = K;
    k_tensor(t, c->w->stream);
    op_ntt(c, c->w->bz_dq, B * 3 * L, 0, L, true, STORE_SCALE_T);
    op_ntt(c, c->w->bz_db, B * 3 * (L + 1), K, L + 1, true, STORE_SCALE_T);
    BehzFloorArgs f;
    memset(&f, 0, sizeof(f));
    f.dq = c->w->bz_dq; f.db = c->w->bz_db; f.out = out3; f.mods = c->d_mods; f.bz = c->d_behz; f.logn = c->logn;
    f.P = (int)(B * 3); f.L = L; f.K = K;
    k_behz_floor(f, c->w->stream);
}

int op_relinearize(hhe_ctx *c, const u64 *a3, u64 *out, size_t B)
{
    if (!c->rks->rk) return fail(HHE_ERR_NO_RELIN_KEY, "relinearization key not set");
    const size_t ln = (size_t)c->L * c->n;
    return op_switch_key(c, a3 + 2 * ln, 3 * ln, c->rks->rk, a3, 3 * ln, 3, out, B);
}

// ------------------------------------------------------------------ PASTA public tables
int ensure_feistel_mask(hhe_ctx *c)
{
    if (c->d_feistel_mask) return HHE_OK;
    const size_t n = c->n, half = n / 2;
    // mask_vec: ones on [1,128) and [N/2+1, N/2+128) (pasta_3_seal.cpp:230-235)
    std::vector<u64> vals(2 * PASTA_T, 1);
    vals[0] = 0; vals[PASTA_T] = 0;
    DevBuf dv(vals.size() * 8), pl(n * 8), mask((size_t)c->L * n * 8);
    if (!dv.p || !pl.p || !mask.p) return dev_fail("feistel mask alloc");
    rt_h2d(dv.p, vals.data(), vals.size() * 8, c->w->stream);
    op_encode(c, dv.w(), 1, 2 * PASTA_T, PASTA_T, (int)half, pl.w());
    op_lift_ntt(c, pl.w(), 1, mask.w());
    if (rt_sync(c->w->stream)) return dev_fail("feistel mask");
    c->d_feistel_mask = mask.release();
    return HHE_OK;
}

void free_block(BlockTables &bt) { rt_free(bt.diag); rt_free(bt.pdiag); rt_free(bt.rc); rt_free(bt.bsgs); }
// make room for `need` more bytes of block tables: the least recently used counters that the running call (c->block_call) does not
// use are dropped while the cache would pass its limit.  A single call that needs more than the limit is served anyway.
void evict_blocks(hhe_ctx *c, size_t need)
{
    while (c->block_bytes + need > c->block_cache_limit) {
        auto victim = c->blocks.end();
        for (auto it = c->blocks.begin(); it != c->blocks.end(); ++it)
            if (it->second.last_call != c->block_call && (victim == c->blocks.end() || it->second.last_call < victim->second.last_call)) victim = it;
        if (victim == c->blocks.end()) return;
        sync_ctx(c);  // earlier calls have completed (every entry point waits for its work), but a caller's stream may lag
        c->block_bytes -= victim->second.bytes;
        free_block(victim->second);
        c->blocks.erase(victim);
    }
}
// Public tables of one block counter, built on the device on first use.  Footprint per counter: (4 x 128 x L) x N words of
// multipliers -- twice that for `pdiag` with its Shoup quotients (fused pipeline; 768 MiB at N = 2^15, L = 3), once for `diag`
// (op-by-op schedule, HHE_MATMUL=0) -- plus 4 N words of round constants; the babystep-giantstep variant adds (4 x 128 x L) x N on
// its first use.  The cache is bounded (hhe_pasta3_set_block_cache_limit); hhe_pasta3_clear_block_cache() releases everything.
int ensure_block(hhe_ctx *c, u64 block, BlockTables **out)
{
    auto it = c->blocks.find(block);
    if (it != c->blocks.end()) { it->second.last_call = c->block_call; *out = &it->second; return HHE_OK; }
    const size_t n = c->n, half = n / 2;
    const int L = c->L;
    const size_t ndiag = (size_t)(PASTA_R + 1) * PASTA_T;
    std::vector<u64> mats((size_t)(PASTA_R + 1) * 2 * PASTA_T * PASTA_T), rcs((size_t)(PASTA_R + 1) * 2 * PASTA_T);
    pasta3_block_randomness(c->t, PASTA_NONCE, block, mats.data(), rcs.data());
    const bool fused = c->matmul_mode == 1;
    const size_t entry_bytes = (fused ? 2 : 1) * ndiag * L * n * 8 + (size_t)(PASTA_R + 1) * n * 8;
    evict_blocks(c, entry_bytes);
    DevBuf d_mats(mats.size() * 8), d_rcs(rcs.size() * 8), slots(ndiag * n * 8), diag(ndiag * L * n * 8),
        rc((size_t)(PASTA_R + 1) * n * 8), pdiag(fused ? 2 * ndiag * L * n * 8 : 8);  // pdiag | its Shoup quotients
    if (!d_mats.p || !d_rcs.p || !slots.p || !diag.p || !rc.p || !pdiag.p) return dev_fail("block table alloc");
    rt_h2d(d_mats.p, mats.data(), mats.size() * 8, c->w->stream);
    rt_h2d(d_rcs.p, rcs.data(), rcs.size() * 8, c->w->stream);
    // 128 diagonals per affine layer -> slot image -> INTT mod t (= batch encode) -> lift + NTT per limb
    rt_memset(slots.p, 0, ndiag * n * 8, c->w->stream);
    DiagArgs d;
    memset(&d, 0, sizeof(d));
    d.mats = d_mats.w(); d.out = slots.w(); d.slot_map = c->d_slot_map; d.logn = c->logn;
    k_diag(d, c->w->stream);
    op_ntt(c, slots.w(), ndiag, c->mod_t, 1, true);
    op_lift_ntt(c, slots.w(), ndiag, diag.w());
    // round constants: rc1 -> slots [0,128), rc2 -> slots [N/2, N/2+128) (pasta_3_plain.cpp:286-295)
    op_encode(c, d_rcs.w(), PASTA_R + 1, 2 * PASTA_T, PASTA_T, (int)half, rc.w());
    if (fused) {
        // pdiag = diag o pi_g for g = elt(rotate_rows -1): multiplier tables in the rotated NTT frame
        PermArgs p;
        memset(&p, 0, sizeof(p));
        p.in = diag.w(); p.out = pdiag.w(); p.mods = c->d_mods; p.logn = c->logn; p.count = (int)(ndiag * L); p.L = L;
        p.out_item_stride = (size_t)L * n; p.elt = galois_elt_from_step(c, -1);
        k_perm(p, c->w->stream);
        op_elt(c, ELT_SHOUP, pdiag.w(), nullptr, pdiag.w() + ndiag * L * n, ndiag * L, 0, L);
    }
    if (rt_sync(c->w->stream)) return dev_fail("block tables");
    BlockTables bt;
    bt.rc = rc.release();
    if (fused) bt.pdiag = pdiag.release();  // the fused pipeline reads only pdiag: diag is dropped with this scope
    else bt.diag = diag.release();
    bt.bytes = entry_bytes;
    bt.last_call = c->block_call;
    c->block_bytes += entry_bytes;
    auto ins = c->blocks.emplace(block, bt);
    *out = &ins.first->second;
    return HHE_OK;
}

// babystep-giantstep multiplier tables of one block (lazy): same pipeline as `diag` with the rearranged diagonals
int ensure_bsgs_tables(hhe_ctx *c, u64 block, BlockTables *bt)
{
    if (bt->bsgs) return HHE_OK;
    const size_t n = c->n;
    const int L = c->L;
    const size_t ndiag = (size_t)(PASTA_R + 1) * PASTA_T;
    std::vector<u64> mats((size_t)(PASTA_R + 1) * 2 * PASTA_T * PASTA_T), rcs((size_t)(PASTA_R + 1) * 2 * PASTA_T);
    pasta3_block_randomness(c->t, PASTA_NONCE, block, mats.data(), rcs.data());
    evict_blocks(c, ndiag * L * n * 8);
    DevBuf d_mats(mats.size() * 8), slots(ndiag * n * 8), bsgs(ndiag * L * n * 8);
    if (!d_mats.p || !slots.p || !bsgs.p) return dev_fail("bsgs table alloc");
    rt_h2d(d_mats.p, mats.data(), mats.size() * 8, c->w->stream);
    rt_memset(slots.p, 0, ndiag * n * 8, c->w->stream);
    BsgsDiagArgs d;
    memset(&d, 0, sizeof(d));
    d.mats = d_mats.w(); d.out = slots.w(); d.slot_map = c->d_slot_map; d.logn = c->logn; d.n1 = 16;
    k_bsgs_diag(d, c->w->stream);
    op_ntt(c, slots.w(), ndiag, c->mod_t, 1, true);
    op_lift_ntt(c, slots.w(), ndiag, bsgs.w());
    if (rt_sync(c->w->stream)) return dev_fail("bsgs tables");
    bt->bsgs = bsgs.release();
    bt->bytes += ndiag * L * n * 8;
    c->block_bytes += ndiag * L * n * 8;
    return HHE_OK;
}

// PASTA_SEAL::babystep_giantstep (pasta_3_seal.cpp:267-366), N1 = 16, N2 = 8 (pasta_3_seal.h:35-36): 15 baby
// rotations by -1, 8 inner sums of 16 plain products (accumulated in the NTT domain), 7 giant
// rotations by -16k.  State in ws_ct[0].
int matmul_bsgs(hhe_ctx *c, int layer, const u64 *const *d_bsgs_ptrs, size_t B)
{
    constexpr int N1 = 16, N2 = 8;
    const int L = c->L;
    const size_t n = c->n, ctw = c->ct_words();
    Lane &w = *c->w;
    if (w.rot_cap < B) {
        rt_sync(w.stream);
        rt_free(w.ws_rot);
        w.ws_rot = (u64 *)rt_malloc(B * N1 * ctw * 8);
        if (!w.ws_rot) return dev_fail("bsgs workspace");
        w.rot_cap = B;
    }
    u64 *state = w.ws_ct[0], *inner = w.ws_ct[1], *scratch = w.ws_ct[2], *outer = w.ws_ct3;  // ct3 holds [B][3].. >= [B][2]
    if (n / 2 != PASTA_T) {
        int rc = op_rotate_rows(c, state, PASTA_T, scratch, B);
        if (rc) return rc;
        op_add(c, state, scratch, state, B, 2);
    }
    // rot[j] laid out [j][B][2][L][N]
    rt_d2d(w.ws_rot, state, B * ctw * 8, w.stream);
    for (int j = 1; j < N1; ++j) {
        int rc = op_rotate_rows(c, w.ws_rot + (size_t)(j - 1) * B * ctw, -1, w.ws_rot + (size_t)j * B * ctw, B);
        if (rc) return rc;
    }
    for (int k = 0; k < N2; ++k) {
        rt_memset(inner, 0, B * ctw * 8, w.stream);
        for (int j = 0; j < N1; ++j) {
            NttArgs a = ntt_args(c, w.ws_rot + (size_t)j * B * ctw, scratch, B * 2 * L, 0, L);
            a.store_op = STORE_MAC; a.mul_ptrs = d_bsgs_ptrs; a.mul_shift = ((size_t)layer * PASTA_T + k * N1 + j) * L * n;
            a.mul_cycle = L; a.mul_item_polys = 2 * L; a.acc = inner;
            k_ntt(a, false, w.stream);
        }
        op_ntt(c, inner, B * 2 * L, 0, L, true);
        if (k == 0) rt_d2d(outer, inner, B * ctw * 8, w.stream);
        else {
            int rc = op_rotate_rows(c, inner, -k * N1, inner, B);
            if (rc) return rc;
            op_add(c, outer, inner, outer, B, 2);
        }
    }
    rt_d2d(state, outer, B * ctw * 8, w.stream);
    return HHE_OK;
}

// PASTA_SEAL::diagonal (pasta_3_seal.cpp:370-413) for affine layer `layer`, state in ws_ct[0].
// The 128 products are accumulated in the NTT domain and inverse-transformed once, which
// yields the same words as SEAL's multiply_plain + add_inplace chain.
int matmul_diagonal(hhe_ctx *c, int layer, const u64 *const *d_diag_ptrs, size_t B)
{
    const int L = c->L;
    const size_t n = c->n;
    u64 *state = c->w->ws_ct[0], *acc = c->w->ws_ct[1], *scratch = c->w->ws_ct[2];
    if (n / 2 != PASTA_T) {
        int rc = op_rotate_rows(c, state, PASTA_T, scratch, B);
        if (rc) return rc;
        op_add(c, state, scratch, state, B, 2);
    }
    rt_memset(acc, 0, B * c->ct_words() * 8, c->w->stream);
    for (int i = 0; i < PASTA_T; ++i) {
        if (i) {
            int rc = op_rotate_rows(c, state, -1, state, B);
            if (rc) return rc;
        }
        NttArgs a = ntt_args(c, state, scratch, B * 2 * L, 0, L);
        a.store_op = STORE_MAC; a.mul_ptrs = d_diag_ptrs; a.mul_shift = ((size_t)layer * PASTA_T + i) * L * n;
        a.mul_cycle = L; a.mul_item_polys = 2 * L; a.acc = acc;
        k_ntt(a, false, c->w->stream);
    }
    op_ntt(c, acc, B * 2 * L, 0, L, true);
    rt_d2d(state, acc, B * c->ct_words() * 8, c->w->stream);
    return HHE_OK;
}

// hhe_ctx_profile: one timed event pair around a launch on the lane's stream (the stream the kernel runs on)
struct ProfScope {
    Lane *ln = nullptr;
    void *e1 = nullptr;
    ProfScope(hhe_ctx *c, Lane &lane, size_t items)
    {
        if (!c->profile) return;
        if (lane.prof_used == lane.prof_ev.size()) {
            void *a = rt_event_create_timed(), *b = rt_event_create_timed();
            if (!a || !b) { rt_event_destroy(a); rt_event_destroy(b); return; }
            lane.prof_ev.emplace_back(a, b);
        }
        auto &pr = lane.prof_ev[lane.prof_used++];
        ln = &lane; e1 = pr.second;
        c->prof_items += items;
        rt_event_record(pr.first, lane.stream);
    }
    ~ProfScope() { if (ln) rt_event_record(e1, ln->stream); }
};

// PASTA_SEAL::diagonal (pasta_3_seal.cpp:370-413) as a fused pipeline: same ciphertext words as the
// op-by-op schedule, 20 transforms per rotation step instead of 35 (DESIGN.md "fused matmul").
//  - c0 stays in NTT form across the 127 rotate_rows(-1); c1 is kept in coefficient form because the
//    key-switch digits need its canonical residues;
//  - the 128 plain products are accumulated in the NTT domain, in the frame rotated by the Galois map
//    (pdiag tables), so the NTT of galois(c1) computed for the key switch is reused for the product;
//  - mod-down of c0 is finished in the NTT domain: NTT_j(r_0 mod q_j - half) replaces INTT+NTT.
int matmul_diagonal_fused(hhe_ctx *c, int layer, const u64 *const *d_pdiag_ptrs, size_t B)
{
    const int L = c->L, K = c->K;
    const size_t n = c->n, ln = (size_t)L * n, bln = B * ln;
    u64 *state = c->w->ws_ct[0];
    if (n / 2 != PASTA_T) {
        int rc = op_rotate_rows(c, state, PASTA_T, c->w->ws_ct[2], B);
        if (rc) return rc;
        op_add(c, state, c->w->ws_ct[2], state, B, 2);
    }
    const u32 g = galois_elt_from_step(c, -1);
    auto it = c->gks->gk.find(g);
    if (it == c->gks->gk.end()) return fail(HHE_ERR_NO_GALOIS_KEY, "Galois key not present");
    const u64 *key = it->second;
    const u32 ginv = (u32)nt_invmod(g, 2 * n);
    u64 *accp0 = c->w->ws_ct[1], *accp1 = c->w->ws_ct[1] + bln;
    u64 *c0n[2] = {c->w->ws_ct[2], c->w->ws_ct[2] + bln};
    u64 *scr = c->w->ws_ct[3], *r = c->w->ws_ct[3] + bln, *scr2 = c->w->ws_ct3;
    rt_memset(c->w->ws_ct[1], 0, 2 * bln * 8, c->w->stream);
    {   // c0 -> NTT form ; d = galois(c1)
        NttArgs a = ntt_args(c, state, c0n[0], B * L, 0, L);
        a.src_item_polys = L; a.src_item_stride = 2 * ln;
        k_ntt(a, false, c->w->stream);
        GaloisArgs ga;
        memset(&ga, 0, sizeof(ga));
        ga.mods = c->d_mods; ga.logn = c->logn; ga.count = (int)(B * L); ga.L = L; ga.einv = ginv;
        ga.in = state + ln; ga.in_item_stride = 2 * ln; ga.out = c->w->ws_d; ga.out_item_stride = ln;
        k_galois(ga, c->w->stream);
    }
    int cur = 0;
    // The c0 branch of step i only feeds the c0 branch of step i+1, and like the digit transforms of step i+1 it depends on
    // nothing later than the inverse transforms of step i: it is held back (k5) and launched in the grids of step i+1.
    const bool rowk = use_row_kernel(c);
    const size_t pdiag_words = (size_t)(PASTA_R + 1) * PASTA_T * ln;  // the Shoup quotients of pdiag follow the table (ensure_block)
    const u64 *key_s = nullptr;
    if (rowk) {
        int rc = ensure_key_shoup(c, key, &key_s);
        if (rc) return rc;
    }
    NttArgs k5;
    bool k5_pending = false;
    // digit transforms of the current d: T[I][J] = NTT_J(d[I] mod q_J)
    auto digit_args = [&]() {
        NttArgs a = ntt_args(c, c->w->ws_d, c->w->ws_T, B * L * K, 0, K);
        a.src_div = K; a.src_item_polys = L * K; a.src_item_stride = ln; a.load_op = LOAD_DIGIT; a.digit_reduce = c->digit_reduce;
        a.store_op = STORE_LAZY;
        return a;
    };
    // N >= 4096 -- four launches per step: strided pass of the digit transforms (+ the held-back c0 branch's strided pass in
    // the same grid); ks_row_kernel = row pass + key inner product + inverse row pass (+ the c0 branch's row pass as extra
    // tiles; S_0 alternates between the two halves of ws_S, so that branch reads the previous step's while this step's is
    // written); strided inverse pass of the special limbs (r_k = INTT(S_k[special]) + floor(q_sp/2)); strided inverse pass
    // of the c1 limbs with the mod-down epilogue and the Galois map (the next digits' source)
    int krc = 0;
    auto step_row_kernel = [&](int i, size_t shift) {
        const NttArgs a = digit_args();
        if (k5_pending) k_ntt2_fwd_first(k5, a, c->w->stream);
        else k_ntt_pass(a, false, false, c->w->stream);
        KsRowArgs x;
        memset(&x, 0, sizeof(x));
        x.key = key; x.key_s = key_s; x.S = c->w->ws_S + (size_t)(i & 1) * K * n; x.U1 = scr; x.u_stride = ln; x.Usp = r;
        x.B = (int)B; x.L = L; x.K = K;
        x.acc = accp1; x.mul_ptrs = d_pdiag_ptrs; x.mul_shift = shift; x.mul_s_off = pdiag_words;
        {
            ProfScope prof(c, *c->w, B);
            krc |= k_ks_row(a, x, k5_pending ? &k5 : nullptr, c->w->stream);
        }
        k5_pending = false;
        NttArgs as = ntt_args(c, r, r, B * 2, K - 1, 1);
        as.store_op = STORE_RSP;
        k_ntt_pass(as, true, true, c->w->stream);
        NttArgs a1 = ntt_args(c, scr, scr, B * L, 0, L);
        a1.store_op = STORE_KS1; a1.aux_r = r; a1.aux_out = c->w->ws_d; a1.gal_elt = g;
        k_ntt_pass(a1, true, true, c->w->stream);
    };
    // N < 4096 (ragged tiles) -- the same step with the inner product as its own kernel: T and S are materialised
    auto step_separate = [&](size_t shift) {
        const NttArgs a = digit_args();
        KsMacArgs m;
        memset(&m, 0, sizeof(m));
        m.T = c->w->ws_T; m.key = key; m.S = c->w->ws_S; m.mods = c->d_mods; m.logn = c->logn; m.B = (int)B; m.L = L; m.K = K;
        m.acc = accp1; m.mul_ptrs = d_pdiag_ptrs; m.mul_shift = shift;  // the I = J digit also feeds the plain product
        if (k5_pending) { k_ntt2_fwd(k5, a, c->w->stream); k5_pending = false; }
        else k_ntt(a, false, c->w->stream);
        k_ks_mac(m, c->w->stream);
        NttArgs as = ntt_args(c, c->w->ws_S + (size_t)(K - 1) * n, r, B * 2, K - 1, 1);
        as.src_item_polys = 1; as.src_item_stride = (size_t)K * n; as.store_op = STORE_RSP;
        NttArgs a1 = ntt_args(c, c->w->ws_S + (size_t)K * n, scr, B * L, 0, L);
        a1.src_item_polys = L; a1.src_item_stride = (size_t)2 * K * n; a1.store_op = STORE_KS1;
        a1.aux_r = r; a1.aux_out = c->w->ws_d; a1.gal_elt = g;
        k_ntt2_inv(as, a1, c->w->stream);
    };
    for (int i = 0; i < PASTA_T - 1; ++i) {
        const size_t shift = ((size_t)layer * PASTA_T + i) * ln;
        if (rowk) step_row_kernel(i, shift);
        else step_separate(shift);
        {   // c0 of the next state in NTT form + permuted-frame product of the current c0
            NttArgs a = ntt_args(c, r, scr2, B * L, 0, L);
            a.src_item_polys = L; a.src_item_stride = 2 * n; a.src_div = L; a.load_op = LOAD_RNEG;
            a.store_op = STORE_KS0; a.aux_in = c0n[cur]; a.aux_out = c0n[cur ^ 1]; a.acc = accp0;
            a.aux_r = c->w->ws_S + (rowk ? (size_t)(i & 1) * K * n : 0);
            a.mul_ptrs = d_pdiag_ptrs; a.mul_shift = shift; a.gal_elt = g; a.mul_s_off = pdiag_words;
            k5 = a; k5_pending = true;  // launched in the grid of the next step's digit transforms
        }
        cur ^= 1;
    }
    if (krc) return dev_fail("matmul: key-switch row kernel");
    if (k5_pending) k_ntt(k5, false, c->w->stream);
    const size_t shift = ((size_t)layer * PASTA_T + (PASTA_T - 1)) * ln;
    {   // last state: products only (a "virtual" rotation keeps the frame uniform)
        NttArgs a = ntt_args(c, c->w->ws_d, scr, B * L, 0, L);
        a.store_op = STORE_MAC; a.mul_ptrs = d_pdiag_ptrs; a.mul_shift = shift; a.mul_cycle = L; a.mul_item_polys = L; a.acc = accp1;
        k_ntt(a, false, c->w->stream);
        PermArgs p;
        memset(&p, 0, sizeof(p));
        p.in = c0n[cur]; p.out = accp0; p.mods = c->d_mods; p.logn = c->logn; p.count = (int)(B * L); p.L = L;
        p.out_item_stride = ln; p.elt = g; p.mac = 1; p.mul_ptrs = d_pdiag_ptrs; p.mul_shift = shift;
        k_perm(p, c->w->stream);
        // back to the unrotated frame, then to coefficient form
        p.mac = 0; p.mul_ptrs = nullptr; p.elt = ginv; p.out_item_stride = 2 * ln;
        p.in = accp0; p.out = state;
        k_perm(p, c->w->stream);
        p.in = accp1; p.out = state + ln;
        k_perm(p, c->w->stream);
    }
    op_ntt(c, state, B * 2 * L, 0, L, true);
    return HHE_OK;
}

}  // namespace

// ====================================================================== C ABI
extern "C" int hhe_ntt(hhe_ctx *c, uint64_t *polys, size_t count, int mod_base, int mod_cycle, int inverse)
{
    HHE_LOCK(c);
    if (!c || !polys || mod_cycle < 1 || mod_base < 0 || mod_base + mod_cycle > c->nmod) return fail(HHE_ERR_INVALID, "hhe_ntt: bad arguments");
    op_ntt(c, polys, count, mod_base, mod_cycle, inverse != 0);
    return HHE_OK;
}
extern "C" int hhe_encode(hhe_ctx *c, const uint64_t *vals, size_t B, size_t count, uint64_t *plain)
{
    HHE_LOCK(c);
    if (!c || !vals || !plain || count > c->n) return fail(HHE_ERR_INVALID, "hhe_encode: bad arguments");
    op_encode(c, vals, B, (int)count, (int)count, -1, plain);
    return HHE_OK;
}
extern "C" int hhe_add(hhe_ctx *c, const uint64_t *a, const uint64_t *b, uint64_t *out, size_t B, int size)
{
    HHE_LOCK(c);
    if (!c || !a || !b || !out) return fail(HHE_ERR_INVALID, "hhe_add: bad arguments");
    op_add(c, a, b, out, B, size);
    return HHE_OK;
}
extern "C" int hhe_negate(hhe_ctx *c, const uint64_t *a, uint64_t *out, size_t B, int size)
{
    HHE_LOCK(c);
    if (!c || !a || !out) return fail(HHE_ERR_INVALID, "hhe_negate: bad arguments");
    op_elt(c, ELT_NEG, a, nullptr, out, B * size * c->L, 0, c->L);
    return HHE_OK;
}
extern "C" int hhe_add_plain(hhe_ctx *c, const uint64_t *ct, const uint64_t *plain, int bcast, int subtract, uint64_t *out, size_t B)
{
    HHE_LOCK(c);
    if (!c || !ct || !plain || !out) return fail(HHE_ERR_INVALID, "hhe_add_plain: bad arguments");
    op_add_plain(c, ct, plain, nullptr, 0, bcast != 0, subtract != 0, false, out, B);
    return HHE_OK;
}
extern "C" int hhe_multiply_plain(hhe_ctx *c, const uint64_t *ct, const uint64_t *plain, int bcast, uint64_t *out, size_t B)
{
    HHE_LOCK(c);
    if (!c || !ct || !plain || !out) return fail(HHE_ERR_INVALID, "hhe_multiply_plain: bad arguments");
    int rc = need(c, B);
    if (rc) return rc;
    const size_t P = bcast ? 1 : B;
    u64 *D = c->w->ws_ct3;  // [P][L][N] fits in [B][3][L][N]
    op_lift_ntt(c, plain, P, D);
    if (bcast) op_multiply_plain_ntt(c, ct, D, nullptr, 0, out, B);
    else {
        // per-item multiplier: items laid [B][L][N]; the pointer form of the fused store reads the lane's pointer table
        Lane &ln = *c->w;
        stage_begin(ln);
        ln.h_ptrs.assign(B, nullptr);
        for (size_t b = 0; b < B; ++b) ln.h_ptrs[b] = D + b * c->L * c->n;
        rt_h2d(ln.d_ptrs, ln.h_ptrs.data(), B * sizeof(u64 *), ln.stream);
        stage_end(ln);
        NttArgs a = ntt_args(c, ct, out, B * 2 * c->L, 0, c->L);
        a.store_op = STORE_MUL; a.mul_ptrs = ln.d_ptrs; a.mul_cycle = c->L; a.mul_item_polys = 2 * c->L;
        k_ntt(a, false, ln.stream);
        op_ntt(c, out, B * 2 * c->L, 0, c->L, true);
    }
    return HHE_OK;
}
// a key set handed to an entry point must belong to the context it is used with
static int check_sets(const hhe_ctx *c, const hhe_keyset *a, const hhe_keyset *b = nullptr, const hhe_keyset *d = nullptr)
{
    for (const hhe_keyset *ks : {a, b, d})
        if (ks && ks->ctx != c) return fail(HHE_ERR_INVALID, "key set belongs to another context");
    return HHE_OK;
}
extern "C" int hhe_apply_galois_ks(hhe_ctx *c, const hhe_keyset *gk, const uint64_t *ct, uint32_t elt, uint64_t *out, size_t B)
{
    HHE_LOCK(c);
    int rc = need(c, B);
    if (rc || (rc = check_sets(c, gk))) return rc;
    if (!ct || !out) return fail(HHE_ERR_INVALID, "hhe_apply_galois: null argument");
    KeyScope keys(c, gk, nullptr);
    return op_apply_galois(c, ct, elt, out, B);
}
extern "C" int hhe_apply_galois(hhe_ctx *c, const uint64_t *ct, uint32_t elt, uint64_t *out, size_t B) { return hhe_apply_galois_ks(c, nullptr, ct, elt, out, B); }
extern "C" int hhe_rotate_rows_ks(hhe_ctx *c, const hhe_keyset *gk, const uint64_t *ct, int step, uint64_t *out, size_t B)
{
    HHE_LOCK(c);
    int rc = need(c, B);
    if (rc || (rc = check_sets(c, gk))) return rc;
    if (!ct || !out) return fail(HHE_ERR_INVALID, "hhe_rotate_rows: null argument");
    KeyScope keys(c, gk, nullptr);
    return op_rotate_rows(c, ct, step, out, B);
}
extern "C" int hhe_rotate_rows(hhe_ctx *c, const uint64_t *ct, int step, uint64_t *out, size_t B) { return hhe_rotate_rows_ks(c, nullptr, ct, step, out, B); }
extern "C" int hhe_rotate_columns_ks(hhe_ctx *c, const hhe_keyset *gk, const uint64_t *ct, uint64_t *out, size_t B)
{
    HHE_LOCK(c);
    int rc = need(c, B);
    if (rc || (rc = check_sets(c, gk))) return rc;
    if (!ct || !out) return fail(HHE_ERR_INVALID, "hhe_rotate_columns: null argument");
    KeyScope keys(c, gk, nullptr);
    return op_apply_galois(c, ct, (u32)(2 * c->n - 1), out, B);
}
extern "C" int hhe_rotate_columns(hhe_ctx *c, const uint64_t *ct, uint64_t *out, size_t B) { return hhe_rotate_columns_ks(c, nullptr, ct, out, B); }
extern "C" int hhe_multiply(hhe_ctx *c, const uint64_t *a, const uint64_t *b, uint64_t *out3, size_t B)
{
    HHE_LOCK(c);
    int rc = need(c, B);
    if (rc) return rc;
    op_multiply(c, a, b, out3, B);
    return HHE_OK;
}
extern "C" int hhe_relinearize_ks(hhe_ctx *c, const hhe_keyset *rk, const uint64_t *a3, uint64_t *out, size_t B)
{
    HHE_LOCK(c);
    int rc = need(c, B);
    if (rc || (rc = check_sets(c, rk))) return rc;
    if (!a3 || !out) return fail(HHE_ERR_INVALID, "hhe_relinearize: null argument");
    KeyScope keys(c, nullptr, rk);
    return op_relinearize(c, a3, out, B);
}
extern "C" int hhe_relinearize(hhe_ctx *c, const uint64_t *a3, uint64_t *out, size_t B) { return hhe_relinearize_ks(c, nullptr, a3, out, B); }
extern "C" int hhe_relinearize_slot(hhe_ctx *c, int slot, const uint64_t *a3, uint64_t *out, size_t B)
{
    HHE_LOCK(c);
    if (!c || slot < 0 || slot >= HHE_RELIN_SLOTS) return fail(HHE_ERR_INVALID, "hhe_relinearize_slot: bad arguments");
    KeyScope keys(c, nullptr, c->relin_set(slot));
    if (!c->rks->rk) return fail(HHE_ERR_NO_RELIN_KEY, "relinearization key not set");
    int rc = need(c, B);
    if (rc) return rc;
    return op_relinearize(c, a3, out, B);
}

// one chunk of the batch on the current lane (c->w): the schedule of PASTA_SEAL::decomposition (pasta_3_seal.cpp:123-170)
static int transcipher_chunk(hhe_ctx *c, const u64 *enc_key, const u64 *const *d_diag, const u64 *const *d_rc,
                             const u64 *cw_padded_host, u64 *out, size_t B, bool bsgs)
{
    const size_t n = c->n;
    const int L = c->L;
    const bool fused = c->matmul_mode == 1;
    int rc = HHE_OK;
    rt_h2d(c->w->ws_vals, cw_padded_host, B * PASTA_T * 8, c->w->stream);
    u64 *state = c->w->ws_ct[0], *tmp = c->w->ws_ct[1], *t3 = c->w->ws_ct3;
    // state <- enc_ssk[0] for every item (pasta_3_seal.cpp:126)
    op_elt(c, ELT_BCAST, nullptr, enc_key, state, B * 2 * L, 0, L, 2 * L);
    for (int r = 0; r <= PASTA_R && !rc; ++r) {
        if ((rc = bsgs ? matmul_bsgs(c, r, d_diag, B) : fused ? matmul_diagonal_fused(c, r, d_diag, B) : matmul_diagonal(c, r, d_diag, B))) break;
        // add_rc (:205-211)
        op_add_plain(c, state, nullptr, d_rc, (size_t)r * n, false, false, false, state, B);
        // mix (:417-423)
        if ((rc = op_apply_galois(c, state, (u32)(2 * n - 1), tmp, B))) break;
        op_add(c, tmp, state, tmp, B, 2);
        op_add(c, state, tmp, state, B, 2);
        if (r == PASTA_R) break;
        if (r == PASTA_R - 1) {
            // sbox_cube (:215-218): exponentiate_inplace(x,3) == relin(mul(relin(mul(x,x)), x))
            op_multiply(c, state, state, t3, B);
            if ((rc = op_relinearize(c, t3, tmp, B))) break;
            op_multiply(c, tmp, state, t3, B);
            if ((rc = op_relinearize(c, t3, state, B))) break;
        } else {
            // sbox_feistel (:222-247)
            if ((rc = op_rotate_rows(c, state, -1, tmp, B))) break;
            op_multiply_plain_ntt(c, tmp, c->d_feistel_mask, nullptr, 0, tmp, B);
            op_multiply(c, tmp, tmp, t3, B);
            if ((rc = op_relinearize(c, t3, tmp, B))) break;
            op_add(c, state, tmp, state, B, 2);
        }
    }
    if (!rc) {
        // res = Enc(c_b) - KS : encode, negate, add_plain (:161-169)
        op_encode(c, c->w->ws_vals, B, PASTA_T, PASTA_T, -1, c->w->ws_plain);
        op_add_plain(c, state, c->w->ws_plain, nullptr, 0, false, false, true, out, B);
    }
    return rc;
}

// hhe_pasta3_transcipher with the key objects already named (c->rks / c->gks)
static int transcipher_impl(hhe_ctx *c, const uint64_t *enc_key, const uint64_t *cw, const uint32_t *ncw,
                            const uint64_t *block_index, size_t B, int use_bsgs, uint64_t *out)
{
    if (!c || !enc_key || !cw || !ncw || !block_index || !out || B == 0) return fail(HHE_ERR_INVALID, "hhe_pasta3_transcipher: null argument or empty batch");
    const size_t n = c->n, half = n / 2;
    // pasta_3_seal.cpp:376-377
    if ((size_t)PASTA_T * 2 != n && (size_t)PASTA_T * 4 > n) return fail(HHE_ERR_TOO_FEW_SLOTS, "too little slots for matmul implementation!");
    if (!c->rks->rk) return fail(HHE_ERR_NO_RELIN_KEY, "relinearization key not set");
    for (int step : {-1, half != PASTA_T ? PASTA_T : -1, 0})
        if (!c->gks->gk.count(galois_elt_from_step(c, step))) return fail(HHE_ERR_NO_GALOIS_KEY, "Galois key not present");
    if (use_bsgs)  // add_gk_indices (:196-200): -k*BSGS_N1, k = 1..7
        for (int k = 1; k < 8; ++k)
            if (!c->gks->gk.count(galois_elt_from_step(c, -16 * k))) return fail(HHE_ERR_NO_GALOIS_KEY, "Galois key not present");
    Lane &main = c->lanes[0];
    c->w = &main;
    int rc;
    if ((rc = ensure_feistel_mask(c))) return rc;
    ++c->block_call;
    // per-item public tables
    std::vector<const u64 *> ptrs(2 * B);
    std::vector<u64> cwp(B * PASTA_T, 0);
    for (size_t b = 0; b < B; ++b) {
        if (ncw[b] > PASTA_T) return fail(HHE_ERR_INVALID, "hhe_pasta3_transcipher: more than 128 words in a block");
        BlockTables *bt = nullptr;
        if ((rc = ensure_block(c, block_index[b], &bt))) return rc;
        if (use_bsgs && (rc = ensure_bsgs_tables(c, block_index[b], bt))) return rc;
        ptrs[b] = use_bsgs ? bt->bsgs : c->matmul_mode == 1 ? bt->pdiag : bt->diag;
        ptrs[B + b] = bt->rc;
        memcpy(&cwp[b * PASTA_T], cw + b * PASTA_T, ncw[b] * 8);
    }
    const int ns = c->nstreams;
    if (ns == 0) {
        if (!(rc = lane_reserve(c, main, B))) {
            std::vector<const u64 *> lp(2 * main.ptr_cap, nullptr);
            for (size_t b = 0; b < B; ++b) { lp[b] = ptrs[b]; lp[main.ptr_cap + b] = ptrs[B + b]; }
            rt_h2d(main.d_ptrs, lp.data(), lp.size() * sizeof(u64 *), main.stream);
            rc = transcipher_chunk(c, enc_key, main.d_ptrs, main.d_ptrs + main.ptr_cap, cwp.data(), out, B, use_bsgs != 0);
        }
    } else {
        // independent chunks round-robin over the internal streams: a chunk's working set stays cache resident and
        // concurrent streams de-phase the load / butterfly / store phases of the transforms
        // balanced chunks: as many as the chunk size demands, rounded up to a multiple of the stream count
        size_t nch = (B + c->chunk - 1) / c->chunk;
        if (nch > 1) nch = (nch + ns - 1) / ns * ns;
        const size_t per = (B + nch - 1) / nch;
        for (int s = 1; s <= ns && !rc; ++s) rc = lane_reserve(c, c->lanes[s], per);
        if (!rc) {
            rt_event_record(c->ev_fork, main.stream);
            for (int s = 1; s <= ns; ++s) rt_stream_wait_event(c->lanes[s].stream, c->ev_fork);
            size_t idx = 0;
            std::vector<std::vector<const u64 *>> keep;  // host staging must outlive the asynchronous copies
            for (size_t b0 = 0; b0 < B && !rc; b0 += per, ++idx) {
                const size_t bc = std::min(per, B - b0);
                Lane &ln = c->lanes[1 + idx % ns];
                c->w = &ln;
                keep.emplace_back(2 * ln.ptr_cap, nullptr);
                std::vector<const u64 *> &lp = keep.back();
                for (size_t b = 0; b < bc; ++b) { lp[b] = ptrs[b0 + b]; lp[ln.ptr_cap + b] = ptrs[B + b0 + b]; }
                rt_h2d(ln.d_ptrs, lp.data(), lp.size() * sizeof(u64 *), ln.stream);
                rc = transcipher_chunk(c, enc_key, ln.d_ptrs, ln.d_ptrs + ln.ptr_cap, &cwp[b0 * PASTA_T], out + b0 * c->ct_words(), bc, use_bsgs != 0);
            }
            for (int s = 1; s <= ns; ++s) {
                rt_event_record(c->lanes[s].ev_done, c->lanes[s].stream);
                rt_stream_wait_event(main.stream, c->lanes[s].ev_done);
            }
            if (rt_sync(main.stream) && !rc) rc = dev_fail("hhe_pasta3_transcipher");
        }
        c->w = &main;
    }
    if (rt_sync(main.stream) && !rc) rc = dev_fail("hhe_pasta3_transcipher");
    return rc;
}
extern "C" int hhe_pasta3_transcipher_ks(hhe_ctx *c, const hhe_keyset *rk, const hhe_keyset *gk, const uint64_t *enc_key, const uint64_t *cw,
                                         const uint32_t *ncw, const uint64_t *block_index, size_t B, int use_bsgs, uint64_t *out)
{
    HHE_LOCK(c);
    if (!c) return fail(HHE_ERR_INVALID, "hhe_pasta3_transcipher: null context");
    int rc = check_sets(c, rk, gk);
    if (rc) return rc;
    KeyScope keys(c, gk, rk);
    return transcipher_impl(c, enc_key, cw, ncw, block_index, B, use_bsgs, out);
}
extern "C" int hhe_pasta3_transcipher(hhe_ctx *c, const uint64_t *enc_key, const uint64_t *cw, const uint32_t *ncw,
                                      const uint64_t *block_index, size_t B, int use_bsgs, uint64_t *out)
{
    return hhe_pasta3_transcipher_ks(c, nullptr, nullptr, enc_key, cw, ncw, block_index, B, use_bsgs, out);
}

extern "C" int hhe_pasta3_set_block_cache_limit(hhe_ctx *c, size_t bytes)
{
    HHE_LOCK(c);
    if (!c) return fail(HHE_ERR_INVALID, "hhe_pasta3_set_block_cache_limit: null context");
    c->block_cache_limit = bytes;
    ++c->block_call;   // no transciphering call is running (the context is locked): nothing is pinned
    evict_blocks(c, 0);
    return HHE_OK;
}
extern "C" int hhe_mask(hhe_ctx *c, const uint64_t *ct, const uint64_t *mask_vals, size_t count, uint64_t *out, size_t B)
{
    HHE_LOCK(c);
    if (!c || !ct || !mask_vals || !out || count > c->n) return fail(HHE_ERR_INVALID, "hhe_mask: bad arguments");
    int rc = need(c, B);
    if (rc) return rc;
    // D = lifted NTT form of the mask plaintext in ws_ct3 [0, L*N); the mask values are staged behind it
    Lane &ln = *c->w;
    u64 *D = ln.ws_ct3, *dv = ln.ws_ct3 + (size_t)c->L * c->n;
    stage_begin(ln);
    ln.h_stage.assign(mask_vals, mask_vals + count);
    rt_h2d(dv, ln.h_stage.data(), count * 8, ln.stream);
    stage_end(ln);
    op_encode(c, dv, 1, (int)count, (int)count, -1, ln.ws_plain);
    op_lift_ntt(c, ln.ws_plain, 1, D);
    op_multiply_plain_ntt(c, ct, D, nullptr, 0, out, B);
    return HHE_OK;
}

static int flatten_impl(hhe_ctx *c, const uint64_t *blocks, size_t nblocks, uint64_t *out, size_t S)
{
    if (!c || !blocks || !out || nblocks == 0) return fail(HHE_ERR_INVALID, "hhe_flatten: bad arguments");
    int rc = need(c, S);
    if (rc) return rc;
    const size_t ctw = c->ct_words();
    // gather block i of every sample into a contiguous batch (one strided copy kernel), rotate by -128*i, accumulate
    CopyItemsArgs g;
    memset(&g, 0, sizeof(g));
    g.src = blocks; g.words = ctw; g.count = S; g.src_stride = nblocks; g.dst_stride = 1;
    g.dst = out;
    k_copy_items(g, c->w->stream);
    for (size_t i = 1; i < nblocks; ++i) {
        g.dst = c->w->ws_ct[0]; g.src_off = i;
        k_copy_items(g, c->w->stream);
        if ((rc = op_rotate_rows(c, c->w->ws_ct[0], -(int)(i * PASTA_T), c->w->ws_ct[1], S))) return rc;
        op_add(c, out, c->w->ws_ct[1], out, S, 2);
    }
    return HHE_OK;
}
extern "C" int hhe_flatten_ks(hhe_ctx *c, const hhe_keyset *gk, const uint64_t *blocks, size_t nblocks, uint64_t *out, size_t S)
{
    HHE_LOCK(c);
    if (!c) return fail(HHE_ERR_INVALID, "hhe_flatten: null context");
    int rc = check_sets(c, gk);
    if (rc) return rc;
    KeyScope keys(c, gk, nullptr);
    return flatten_impl(c, blocks, nblocks, out, S);
}
extern "C" int hhe_flatten(hhe_ctx *c, const uint64_t *blocks, size_t nblocks, uint64_t *out, size_t S) { return hhe_flatten_ks(c, nullptr, blocks, nblocks, out, S); }

// BaseCSP::decompose (src/examples/CSP/CSP.cpp:235-283) / hhe_pktnn_1fc_inference (hhe_pktnn_examples.cpp:578-630) on device:
// per record: decomposition of every 128-word block, mask of the ragged last block, flatten -- without leaving HBM.
extern "C" int hhe_decompose_ks(hhe_ctx *c, const hhe_keyset *rk, const hhe_keyset *gk, const hhe_keyset *flatten_gk, const uint64_t *enc_key,
                                const uint64_t *records, size_t S, size_t nwords, int mask_last, uint64_t *out)
{
    HHE_LOCK(c);
    if (c) { int rcs = check_sets(c, rk, gk, flatten_gk); if (rcs) return rcs; }
    if (!c || !enc_key || !records || !out || S == 0 || nwords == 0) return fail(HHE_ERR_INVALID, "hhe_decompose: bad arguments");
    const size_t nb = (nwords + PASTA_T - 1) / PASTA_T, rem = nwords % PASTA_T, ctw = c->ct_words();
    if (nb * PASTA_T > c->n / 2) return fail(HHE_ERR_INVALID, "hhe_decompose: record does not fit one batching row");
    std::vector<u64> cw(S * nb * PASTA_T, 0), bidx(S * nb);
    std::vector<uint32_t> ncw(S * nb);
    for (size_t s = 0; s < S; ++s)
        for (size_t b = 0; b < nb; ++b) {
            const size_t lo = b * PASTA_T, hi = std::min(lo + PASTA_T, nwords);
            memcpy(&cw[(s * nb + b) * PASTA_T], records + s * nwords + lo, (hi - lo) * 8);
            ncw[s * nb + b] = (uint32_t)(hi - lo);
            bidx[s * nb + b] = b;
        }
    if (c->blocks_cap < S * nb * ctw) {  // grow-only scratch for the decompositions of all blocks
        sync_ctx(c);
        rt_free(c->d_blocks);
        c->blocks_cap = 0;
        if (!(c->d_blocks = (u64 *)rt_malloc(S * nb * ctw * 8))) return dev_fail("hhe_decompose");
        c->blocks_cap = S * nb * ctw;
    }
    u64 *blocks = c->d_blocks;
    int rc;
    {   // PASTA_SEAL HHE(context, pk, sk, analyst rk, analyst gk).decomposition(...) (CSP.cpp:238-252)
        KeyScope keys(c, gk, rk);
        rc = transcipher_impl(c, enc_key, cw.data(), ncw.data(), bidx.data(), S * nb, 0, blocks);
    }
    if (!rc && mask_last && rem) {
        // hhe_pktnn_examples.cpp:620-625: ones on the first `rem` slots (CSP.cpp:264-269 intends the same)
        if (!(rc = need(c, S))) {
            u64 *last = c->lanes[0].ws_ct[2];
            rt_stream st = c->lanes[0].stream;
            CopyItemsArgs g;
            memset(&g, 0, sizeof(g));
            g.src = blocks; g.dst = last; g.words = ctw; g.count = S; g.src_stride = nb; g.src_off = nb - 1; g.dst_stride = 1;
            k_copy_items(g, st);
            std::vector<u64> ones(rem, 1);
            rc = hhe_mask(c, last, ones.data(), rem, last, S);
            g.src = last; g.dst = blocks; g.src_stride = 1; g.src_off = 0; g.dst_stride = nb; g.dst_off = nb - 1;
            if (!rc) k_copy_items(g, st);
        }
    }
    if (!rc) {  // HHE.flatten(record, tmp, csp_gk) (CSP.cpp:271-278): the Galois keys the call names
        KeyScope keys(c, flatten_gk, nullptr);
        rc = flatten_impl(c, blocks, nb, out, S);
    }
    if (!rc && rt_sync(c->lanes[0].stream)) rc = dev_fail("hhe_decompose");
    return rc;
}
extern "C" int hhe_decompose(hhe_ctx *c, const uint64_t *enc_key, const uint64_t *records, size_t S, size_t nwords,
                             int mask_last, uint64_t *out)
{
    return hhe_decompose_ks(c, nullptr, nullptr, nullptr, enc_key, records, S, nwords, mask_last, out);
}

// sealhelper::encrypted_vec_sum (sealhelper.cpp:379-392) adds rotate_rows(prod, -i) for i = 1..n-1, each rotation
// going through SEAL's NAF decomposition (evaluator.h:955-1060).  The NAF term sequences of different i
// share prefixes, and every key switch is a deterministic function of its input, so the rotations form a trie whose
// nodes are computed once: 2875 key switches become 1054 for n = 784 with bit-identical ciphertexts
// (modular addition is commutative, so the order of the final additions is free).
namespace {
struct NafNode {
    int term = 0;
    int mult = 0;                 // how many i end exactly here
    std::vector<int> kids;
};
struct FcLeafAcc {  // sums over the leaves of the trie (DESIGN.md "FC rotation trie"): accS in the NTT domain, accH in the
    u64 *accS, *accH, *rscr;  // coefficient domain (rounding terms + q_sp * galois(c0), added by leaf_round_kernel)
};
// rounding terms of m leaf key switches: inverse transform of their special-limb sums (S: [B][2][m][N] when dense, else the special slot
// of ws_S [B][2][K][N] for one leaf) into acc.rscr [B][2][m][N] (r = INTT(S_k[special]) + half), then the element-wise sums
// half_j - (r mod q_j) (+ q_sp * galois_l(c0 of leaf l's parent) for k = 0) into acc.accH
static void fc_leaf_round(hhe_ctx *c, const u64 *S, bool dense, const u64 *const *parents, const u32 *einv, int m, const FcLeafAcc &acc, size_t B)
{
    const int L = c->L, K = c->K;
    const size_t n = c->n, ln = (size_t)L * n;
    NttArgs r = ntt_args(c, dense ? S : S + (size_t)(K - 1) * n, acc.rscr, B * 2 * m, K - 1, 1);
    if (!dense) { r.src_item_polys = 1; r.src_item_stride = (size_t)K * n; }
    r.store_op = STORE_RSP;
    k_ntt(r, true, c->w->stream);
    LeafRoundArgs lr;
    memset(&lr, 0, sizeof(lr));
    lr.r = acc.rscr; lr.accH = acc.accH; lr.base_stride = 2 * ln; lr.mods = c->d_mods; lr.logn = c->logn;
    lr.B = (int)B; lr.L = L; lr.m = m; lr.ks = c->ksc;
    for (int l = 0; l < m; l++) { lr.base[l] = parents[l]; lr.gal_einv[l] = einv[l]; }
    k_leaf_round(lr, c->w->stream);
}
// A leaf's ciphertext is only ever added into the result.  Key switching is linear up to the rounding term, so for
// leaves the inverse transforms of S_k[j] are postponed: sum S_k[j] over all leaves in the NTT domain, sum the rounding
// terms half_j - (r_k mod q_j) and galois(c0) in the coefficient domain, inverse-transform once (14 instead of 20
// transforms per leaf; identical words because every step is exact modular arithmetic).
int fc_leaf(hhe_ctx *c, const u64 *parent, u32 elt, const FcLeafAcc &acc, size_t B)
{
    auto it = c->gks->gk.find(elt);
    if (it == c->gks->gk.end()) return fail(HHE_ERR_NO_GALOIS_KEY, "Galois key not present");
    const int L = c->L, K = c->K;
    const size_t n = c->n, ln = (size_t)L * n;
    GaloisArgs g;
    memset(&g, 0, sizeof(g));
    g.mods = c->d_mods; g.logn = c->logn; g.count = (int)(B * L); g.L = L; g.einv = (u32)nt_invmod(elt, 2 * n);
    g.in_item_stride = 2 * ln; g.out_item_stride = ln;
    g.in = parent + ln; g.out = c->w->ws_d; g.accumulate = 0;  // galois(c0) joins the sums inside leaf_round_kernel below
    k_galois(g, c->w->stream);
    NttArgs a = ntt_args(c, c->w->ws_d, c->w->ws_T, B * L * K, 0, K);
    a.src_div = K; a.src_item_polys = L * K; a.src_item_stride = ln; a.load_op = LOAD_DIGIT; a.digit_reduce = c->digit_reduce;
    a.store_op = STORE_LAZY;
    k_ntt(a, false, c->w->stream);
    KsMacArgs m;
    memset(&m, 0, sizeof(m));
    m.T = c->w->ws_T; m.key = it->second; m.S = c->w->ws_S; m.s_acc = acc.accS; m.mods = c->d_mods; m.logn = c->logn;
    m.B = (int)B; m.L = L; m.K = K;
    k_ks_mac(m, c->w->stream);
    const u32 einv = (u32)nt_invmod(elt, 2 * n);
    fc_leaf_round(c, c->w->ws_S, false, &parent, &einv, 1, acc, B);
    return HHE_OK;
}
// ---- shared digits (DESIGN.md "FC rotation trie", step 3) ----
// All children of a trie node rotate the SAME ciphertext.  The digit transforms NTT_J(c1_I) are computed once per node;
// for a child with Galois element g the transforms of the digits of galois_g(c1) are the NTT-domain index map of those
// (read through the map inside ks_mac_kernel) plus q_I * NTT_J(s_g) for I != J, where s_g marks the coefficients whose
// sign galois_g flips -- exact as long as no coefficient of c1 is 0 (a flipped 0 stays 0 instead of becoming q_I); the
// digit loads raise zero_flag in that case and the caller recomputes with per-child transforms.  The key-dependent part
// of the correction, NTT_J(s_g) * sum_{I != J} (q_I mod q_J) key_g[I][k][J], is tabulated once per Galois key.
int fc_corr(hhe_ctx *c, u32 elt, const u64 *key, const u64 **out)
{
    auto it = c->gks->gk_corr.find(elt);
    if (it != c->gks->gk_corr.end()) { *out = it->second; return HHE_OK; }
    const int L = c->L, K = c->K;
    const size_t n = c->n;
    std::vector<u64> s((size_t)K * n, 0), qmod((size_t)L * K);
    for (size_t i = 0; i < n; ++i) {  // GaloisTool::apply_galois (util/galois.h:32): i -> i*g mod 2N, negated when it wraps
        const u64 raw = (u64)i * elt;
        if ((raw >> c->logn) & 1) s[raw & (n - 1)] = 1;
    }
    for (int J = 1; J < K; ++J) memcpy(&s[(size_t)J * n], &s[0], n * 8);
    for (int I = 0; I < L; ++I)
        for (int J = 0; J < K; ++J) qmod[(size_t)I * K + J] = c->q[I] % c->q[J];
    u64 *shat = (u64 *)rt_malloc((size_t)K * n * 8), *dq = (u64 *)rt_malloc(qmod.size() * 8), *corr = (u64 *)rt_malloc((size_t)2 * K * n * 8);
    if (!shat || !dq || !corr) { rt_free(shat); rt_free(dq); rt_free(corr); return dev_fail("fc_corr"); }
    rt_stream st = c->w->stream;
    rt_h2d(shat, s.data(), s.size() * 8, st);
    rt_h2d(dq, qmod.data(), qmod.size() * 8, st);
    NttArgs a = ntt_args(c, shat, shat, K, 0, K);
    k_ntt(a, false, st);
    KsCorrArgs k;
    k.key = key; k.shat = shat; k.qmod = dq; k.corr = corr; k.mods = c->d_mods; k.logn = c->logn; k.L = L; k.K = K;
    k_ks_corr(k, st);
    const int bad = rt_sync(st);  // host staging buffers and the temporaries go out of scope
    rt_free(shat); rt_free(dq);
    if (bad) { rt_free(corr); return dev_fail("fc_corr"); }
    c->gks->gk_corr[elt] = corr;
    *out = corr;
    return HHE_OK;
}
// digit transforms of the un-rotated c1 of `parent` into tp ([B][L][K][N], lazy range)
// sp_only: the transforms modulo the special prime alone, tp [B][L][N] (a node whose children are all leaves of the c1-sum scheme)
void fc_parent_digits(hhe_ctx *c, const u64 *parent, u64 *tp, size_t B, bool sp_only = false)
{
    const int L = c->L, K = c->K;
    const size_t ln = (size_t)L * c->n;
    NttArgs a = sp_only ? ntt_args(c, parent + ln, tp, B * L, K - 1, 1) : ntt_args(c, parent + ln, tp, B * L * K, 0, K);
    a.src_div = sp_only ? 1 : K; a.src_item_polys = sp_only ? L : L * K; a.src_item_stride = 2 * ln; a.load_op = LOAD_DIGIT;
    a.digit_reduce = c->digit_reduce;
    a.store_op = STORE_LAZY; a.zero_flag = c->w->zero_flag;
    k_ntt(a, false, c->w->stream);
}
// one child of a node from the node's shared digit transforms: leaf (sums only) or full ciphertext into `cur`
int fc_child_shared(hhe_ctx *c, const u64 *parent, const u64 *tp, u32 elt, const FcLeafAcc *leaf, u64 *cur, size_t B, u64 *add_to = nullptr,
                    const u64 *c0hat = nullptr)
{
    auto it = c->gks->gk.find(elt);
    if (it == c->gks->gk.end()) return fail(HHE_ERR_NO_GALOIS_KEY, "Galois key not present");
    const u64 *corr = nullptr;
    int rc = fc_corr(c, elt, it->second, &corr);
    if (rc) return rc;
    const int L = c->L, K = c->K;
    const size_t n = c->n, ln = (size_t)L * n;
    const u32 einv = (u32)nt_invmod(elt, 2 * n);  // galois(c0) is gathered on the fly (leaf_round_kernel for leaves, the KSF epilogue otherwise)
    u64 *Usp = c->w->ws_S + B * 2 * L * n;  // split output [B][2][L][N] | [B][2][N] (non-leaf children)
    const u64 *key_s = nullptr;
    if (!leaf && c->fc_row_fused && use_row_kernel(c) && ensure_key_shoup(c, it->second, &key_s) == HHE_OK) {
        // N >= 4096: inner product over the shared digits and the inverse row pass of all 2K sums in one kernel (the sums never make a
        // round trip), then the strided inverse passes with the mod-down, the Galois-gathered c0 and the running sum in the data limbs' store
        KsRowArgs x;
        memset(&x, 0, sizeof(x));
        x.key = it->second; x.key_s = key_s; x.U0 = c->w->ws_S; x.U1 = c->w->ws_S + (size_t)L * n; x.u_stride = (size_t)2 * L * n; x.Usp = Usp;
        x.B = (int)B; x.L = L; x.K = K; x.T = tp; x.corr = corr; x.perm_elt = elt; x.c0hat = c0hat;
        NttArgs g = ntt_args(c, nullptr, nullptr, 0, 0, K);
        if (k_ks_perm_row(g, x, c->w->stream)) return dev_fail("fc: key-switch row kernel");
        NttArgs as = ntt_args(c, Usp, Usp, B * 2, K - 1, 1);
        as.store_op = STORE_RSP;
        k_ntt_pass(as, true, true, c->w->stream);
        NttArgs ad = ntt_args(c, c->w->ws_S, c->w->ws_S, B * 2 * L, 0, L);
        ad.store_op = STORE_KSF; ad.aux_r = Usp; ad.aux_in = parent; ad.base_stride = 2 * ln; ad.base_mask = 1; ad.gal_einv = einv; ad.aux_out = cur;
        if (c0hat) { ad.aux_in = nullptr; ad.base_mask = 0; ad.gal_einv = 0; }   // galois(c0) is already inside S_0 (KsRowArgs::c0hat)
        ad.acc = add_to;
        k_ntt_pass(ad, true, true, c->w->stream);
        return HHE_OK;
    }
    KsMacArgs m;
    memset(&m, 0, sizeof(m));
    m.T = tp; m.key = it->second; m.S = c->w->ws_S; m.mods = c->d_mods; m.logn = c->logn; m.B = (int)B; m.L = L; m.K = K;
    m.perm_elt = elt; m.corr = corr;
    if (leaf) m.s_acc = leaf->accS;
    else m.S_sp = Usp;
    k_ks_mac(m, c->w->stream);
    if (leaf) {
        fc_leaf_round(c, c->w->ws_S, false, &parent, &einv, 1, *leaf, B);
        return HHE_OK;
    }
    // all 2K sums are inverse-transformed (the row passes of the special and the data limbs in one grid); the mod-down rides in the
    // store of the data limbs' last pass (STORE_KSF), and so does the addition of a child that is itself a term of the sum (add_to)
    NttArgs as = ntt_args(c, Usp, Usp, B * 2, K - 1, 1);
    as.store_op = STORE_RSP;
    NttArgs ad = ntt_args(c, c->w->ws_S, c->w->ws_S, B * 2 * L, 0, L);
    ad.store_op = STORE_KSF; ad.aux_r = Usp; ad.aux_in = parent; ad.base_stride = 2 * ln; ad.base_mask = 1; ad.gal_einv = einv; ad.aux_out = cur;
    ad.acc = add_to;
    k_ntt2_inv(as, ad, c->w->stream);
    return HHE_OK;
}
// The walk over the trie with shared digits.  A node's digit transforms and ciphertext live in a slot of the lane's pool (Lane::FcSlot)
// for as long as something still reads them: the walk below the node, and its LEAF children, whose key switches are queued and run in
// groups of up to HHE_LEAF_GROUP -- leaves of different nodes together -- with one inner-product launch (the data-limb sums of all of them
// meet accS in one read-modify-write), one inverse transform of their 2 x m special-limb sums and one rounding kernel (accH read and
// written once).  The order of the additions into the sums is immaterial: exact modular arithmetic.
struct FcWalk {
    hhe_ctx *c;
    const std::vector<NafNode> &trie;
    u64 *out;
    const FcLeafAcc *acc;
    size_t B;
    int group, max_slots;
    bool csum;   // data limbs of the leaves through per-element sums of their parents' c1 (CsumArgs); the leaf queue then carries the special limb only
    struct Leaf { int slot; const u64 *parent; u32 elt; } q[HHE_LEAF_GROUP];
    int m = 0;
    struct ElemSum {   // one per Galois element among the leaves
        u32 elt;
        u64 *sums;      // [B][L][N] integer sums of c1 limbs
        u64 *sums0;     // [B][L][N] sums of c0 limbs mod q_j
        unsigned char *carry;  // [B][L][N] wraps of `sums`
        int count = 0;  // leaves summed into `sums`
        int npend = 0;  // parents queued for the next csum_add launch
        int pend_slot[HHE_CSUM_GROUP];
        const u64 *pend_c1[HHE_CSUM_GROUP];
    };
    std::vector<ElemSum> esums;
    static constexpr int CSUM_MAX = 2000;   // terms below 2^61: fewer than 255 wraps of the 64-bit word (CsumArgs::carry is a byte)

    int csum_add_pending(ElemSum &e)
    {
        if (!e.npend) return HHE_OK;
        Lane &ln = *c->w;
        CsumArgs a;
        memset(&a, 0, sizeof(a));
        a.sums = e.sums; a.carry = e.carry; a.sums0 = e.sums0; a.src_stride = 2 * (size_t)c->L * c->n; a.m = e.npend; a.mods = c->d_mods; a.logn = c->logn; a.B = (int)B; a.L = c->L; a.K = c->K;
        for (int l = 0; l < e.npend; l++) a.src[l] = e.pend_c1[l];
        k_csum_add(a, ln.stream);
        for (int l = 0; l < e.npend; l++) ln.fc_slots[e.pend_slot[l]].refs--;
        e.count += e.npend;
        e.npend = 0;
        return HHE_OK;
    }
    // closes a sum: digits of galois(sum) mod every key-level prime, their transforms, ONE inner product into accS
    int csum_close(ElemSum &e)
    {
        int rc = csum_add_pending(e);
        if (rc || !e.count) return rc;
        Lane &ln = *c->w;
        auto it = c->gks->gk.find(e.elt);
        if (it == c->gks->gk.end()) return fail(HHE_ERR_NO_GALOIS_KEY, "Galois key not present");
        const int L = c->L, K = c->K;
        CsumArgs a;
        memset(&a, 0, sizeof(a));
        a.sums = e.sums; a.carry = e.carry; a.out = ln.ws_T; a.mods = c->d_mods; a.logn = c->logn; a.B = (int)B; a.L = L; a.K = K;
        a.einv = (u32)nt_invmod(e.elt, 2 * c->n); a.count = (u32)e.count;
        k_csum_digits(a, ln.stream);
        a.sums0 = e.sums0; a.accH = acc->accH;
        for (int j = 0; j < L; j++) { a.qsp_mod[j] = c->ksc.qsp_mod[j]; a.qsp_mod_s[j] = c->ksc.qsp_mod_s[j]; }
        k_csum_c0(a, ln.stream);
        NttArgs t = ntt_args(c, ln.ws_T, ln.ws_T, B * L * K, 0, K);
        t.store_op = STORE_LAZY;
        k_ntt(t, false, ln.stream);
        KsMacArgs mm;
        memset(&mm, 0, sizeof(mm));
        mm.T = ln.ws_T; mm.key = it->second; mm.S = ln.ws_S; mm.mods = c->d_mods; mm.logn = c->logn; mm.B = (int)B; mm.L = L; mm.K = K;
        mm.s_acc = acc->accS; mm.perm_elt = 1; mm.corr = c->d_zero_corr;  // the digits are those of the ROTATED sum: identity map, no correction
        k_ks_mac(mm, ln.stream);
        rt_memset(e.sums, 0, B * (size_t)L * c->n * 17, ln.stream);   // sums | sums0 | carry
        e.count = 0;
        c->fc_csum_closes++;
        return HHE_OK;
    }
    int csum_leaf(int slot, const u64 *parent, u32 elt)
    {
        Lane &ln = *c->w;
        ElemSum *e = nullptr;
        for (auto &x : esums) if (x.elt == elt) e = &x;
        if (!e) {
            if (ln.csum_bufs.size() <= esums.size()) {
                u64 *p = (u64 *)rt_malloc(ln.fc_slot_cap * (size_t)c->L * c->n * 17);   // sums | sums0 | carry bytes
                if (!p) return dev_fail("hhe_fc_row workspace");
                ln.csum_bufs.push_back(p);
            }
            ElemSum x;
            x.elt = elt; x.sums = ln.csum_bufs[esums.size()]; x.sums0 = x.sums + B * (size_t)c->L * c->n;
            x.carry = (unsigned char *)(x.sums0 + B * (size_t)c->L * c->n);
            rt_memset(x.sums, 0, B * (size_t)c->L * c->n * 17, ln.stream);
            esums.push_back(x);
            e = &esums.back();
        }
        int rc;
        if (e->count + e->npend >= CSUM_MAX && (rc = csum_close(*e))) return rc;
        e->pend_slot[e->npend] = slot; e->pend_c1[e->npend] = parent;
        ++e->npend;
        ln.fc_slots[slot].refs++;
        if (e->npend == c->fc_csum_group) return csum_add_pending(*e);
        return HHE_OK;
    }

    int flush()
    {
        Lane &ln = *c->w;
        int rc = HHE_OK;
        if (m > 0 && !csum && (group == 1 || m == 1)) {
            for (int l = 0; l < m && !rc; l++) rc = fc_child_shared(c, q[l].parent, ln.fc_slots[q[l].slot].tp, q[l].elt, acc, nullptr, B);
        } else if (m > 0) {
            const int L = c->L, K = c->K;
            KsMacLeavesArgs a;
            memset(&a, 0, sizeof(a));
            u32 einv[HHE_LEAF_GROUP];
            const u64 *parents[HHE_LEAF_GROUP];
            for (int l = 0; l < m; l++) {
                auto it = c->gks->gk.find(q[l].elt);
                if (it == c->gks->gk.end()) return fail(HHE_ERR_NO_GALOIS_KEY, "Galois key not present");
                if ((rc = fc_corr(c, q[l].elt, it->second, &a.corr[l]))) return rc;
                a.T[l] = ln.fc_slots[q[l].slot].tp; a.t_polys[l] = ln.fc_slots[q[l].slot].tp_polys; a.key[l] = it->second; a.perm_elt[l] = q[l].elt;
                einv[l] = (u32)nt_invmod(q[l].elt, 2 * c->n);
                parents[l] = csum ? nullptr : q[l].parent;   // the c0 terms of the leaves come from the per-element sums
            }
            a.S_sp = ln.ws_leaf; a.s_acc = acc->accS; a.mods = c->d_mods; a.logn = c->logn; a.B = (int)B; a.L = L; a.K = K; a.m = m;
            a.sp_only = csum ? 1 : 0;
            if (k_ks_mac_leaves(a, ln.stream)) return fail(HHE_ERR_INVALID, "fc: leaf group");
            fc_leaf_round(c, ln.ws_leaf, true, parents, einv, m, *acc, B);
        }
        for (int l = 0; l < m; l++) ln.fc_slots[q[l].slot].refs--;
        m = 0;
        return rc;
    }
    // a free slot (grow-only pool; everything runs in stream order on the lane's stream, so a slot whose readers have been ENQUEUED is free)
    int acquire(int *slot)
    {
        Lane &ln = *c->w;
        for (int pass = 0; pass < 2; pass++) {
            for (size_t i = 0; i < ln.fc_slots.size(); i++)
                if (ln.fc_slots[i].refs == 0) { ln.fc_slots[i].refs = 1; *slot = (int)i; return HHE_OK; }
            if ((int)ln.fc_slots.size() < max_slots) {
                Lane::FcSlot sl;
                sl.tp = (u64 *)rt_malloc(ln.fc_slot_cap * (size_t)c->L * c->K * c->n * 8);   // sized for the pool's capacity, not this call's batch
                sl.ct = (u64 *)rt_malloc(ln.fc_slot_cap * c->ct_words() * 8);
                sl.c0hat = (u64 *)rt_malloc(ln.fc_slot_cap * (size_t)c->L * c->n * 8);
                if (!sl.tp || !sl.ct || !sl.c0hat) { rt_free(sl.tp); rt_free(sl.ct); rt_free(sl.c0hat); return dev_fail("hhe_fc_row workspace"); }
                sl.refs = 1;
                ln.fc_slots.push_back(sl);
                *slot = (int)ln.fc_slots.size() - 1;
                return HHE_OK;
            }
            int rc = flush();  // the queued leaves hold the remaining slots
            for (auto &e : esums) if (!rc) rc = csum_add_pending(e);
            if (rc) return rc;
        }
        return fail(HHE_ERR_INVALID, "hhe_fc_row: slot pool");
    }
    // node's ciphertext is `parent`; `slot` receives the digit transforms of its c1
    int walk(int node, const u64 *parent, int slot)
    {
        if (trie[node].kids.empty()) return HHE_OK;
        Lane &ln = *c->w;
        auto is_leaf = [&](int kid) { return acc && trie[kid].kids.empty() && trie[kid].mult == 1; };
        bool only_leaves = csum;
        for (int kid : trie[node].kids) only_leaves = only_leaves && is_leaf(kid);
        // a node whose children are all leaves needs its digit transforms modulo the special prime only (their data limbs come from the c1 sums)
        fc_parent_digits(c, parent, ln.fc_slots[slot].tp, B, only_leaves);
        ln.fc_slots[slot].tp_polys = only_leaves ? 1 : c->K;
        bool has_nonleaf = false;
        for (int kid : trie[node].kids) has_nonleaf = has_nonleaf || !is_leaf(kid);
        const u64 *c0hat = nullptr;
        if (has_nonleaf && c->fc_c0hat && c->fc_row_fused && use_row_kernel(c)) {
            // NTT form of the node's c0: its non-leaf children take galois(c0) through the NTT-domain map inside ks_perm_row_kernel
            NttArgs t = ntt_args(c, parent, ln.fc_slots[slot].c0hat, B * c->L, 0, c->L);
            t.src_item_polys = c->L; t.src_item_stride = c->ct_words();
            t.store_op = STORE_MUL; t.mul = c->d_qsp_poly; t.mul_cycle = c->L;   // times q_sp: the sum it joins is divided by q_sp in the mod-down
            k_ntt(t, false, ln.stream);
            c0hat = ln.fc_slots[slot].c0hat;
        }
        int rc;
        for (int kid : trie[node].kids) {
            if (!is_leaf(kid)) continue;
            const u32 elt = galois_elt_from_step(c, trie[kid].term);
            if (csum && (rc = csum_leaf(slot, parent, elt))) return rc;
            q[m].slot = slot; q[m].parent = parent; q[m].elt = elt;
            ++m;
            ln.fc_slots[slot].refs++;
            if (m >= group && (rc = flush())) return rc;
        }
        for (int kid : trie[node].kids) {
            if (acc && trie[kid].kids.empty() && trie[kid].mult == 1) continue;
            const u32 elt = galois_elt_from_step(c, trie[kid].term);
            int k = -1;
            if ((rc = acquire(&k))) return rc;
            u64 *cur = ln.fc_slots[k].ct;
            // a child that is itself a term of the sum (distinct steps have distinct term sequences: mult is 0 or 1) is added in its own epilogue
            if ((rc = fc_child_shared(c, parent, ln.fc_slots[slot].tp, elt, nullptr, cur, B, trie[kid].mult ? out : nullptr, c0hat))) return rc;
            for (int mm = 1; mm < trie[kid].mult; ++mm) op_add(c, out, cur, out, B, 2);
            if ((rc = walk(kid, cur, k))) return rc;
            ln.fc_slots[k].refs--;
        }
        return HHE_OK;
    }
};
int fc_dfs_shared(hhe_ctx *c, const std::vector<NafNode> &trie, int max_depth, const u64 *prod, u64 *out, const FcLeafAcc *acc, size_t B)
{
    Lane &ln = *c->w;
    if (ln.fc_slot_cap < B) {  // slots of a smaller batch: start over
        rt_sync(ln.stream);
        for (auto &sl : ln.fc_slots) { rt_free(sl.tp); rt_free(sl.ct); rt_free(sl.c0hat); }
        ln.fc_slots.clear();
        for (u64 *p : ln.csum_bufs) rt_free(p);
        ln.csum_bufs.clear();
        ln.fc_slot_cap = B;
    }
    for (auto &sl : ln.fc_slots) sl.refs = 0;
    const int group = c->L <= 4 ? std::min(HHE_LEAF_GROUP, c->fc_leaf_group) : 1;  // k_ks_mac_leaves is instantiated for L <= 4
    if (c->fc_c0hat && !c->d_qsp_poly) {
        std::vector<u64> h((size_t)c->L * c->n);
        for (int j = 0; j < c->L; j++) std::fill(h.begin() + (size_t)j * c->n, h.begin() + (size_t)(j + 1) * c->n, c->ksc.qsp_mod[j]);
        if (!(c->d_qsp_poly = (u64 *)rt_malloc(h.size() * 8))) return dev_fail("hhe_fc_row workspace");
        if (rt_h2d(c->d_qsp_poly, h.data(), h.size() * 8, ln.stream) || rt_sync(ln.stream)) return dev_fail("hhe_fc_row workspace");
    }
    // the c1-sum scheme serves leaf sums (acc) with a group kernel (L <= 4); its zero table stands in for the correction of the closing product
    bool csum = acc && c->fc_csum && c->L <= 4;
    if (csum && !c->d_zero_corr) {
        const size_t bytes = (size_t)2 * c->K * c->n * 8;
        if (!(c->d_zero_corr = (u64 *)rt_malloc(bytes))) return dev_fail("hhe_fc_row workspace");
        rt_memset(c->d_zero_corr, 0, bytes, ln.stream);
        if (rt_sync(ln.stream)) return dev_fail("hhe_fc_row workspace");   // once per context: chunks on other streams read the table too
    }
    FcWalk w{c, trie, out, acc, B, group, max_depth + 1 + group + 3 * HHE_CSUM_GROUP, csum};
    w.esums.reserve(64);
    int root = -1, rc = w.acquire(&root);
    if (!rc) rc = w.walk(0, prod, root);
    if (!rc) rc = w.flush();
    for (auto &e : w.esums) if (!rc) rc = w.csum_close(e);
    return rc;
}
int fc_dfs(hhe_ctx *c, const std::vector<NafNode> &trie, int node, int depth, const u64 *parent, u64 *bufs, u64 *out,
           const FcLeafAcc *acc, size_t B)
{
    const size_t ctw = c->ct_words();
    for (int kid : trie[node].kids) {
        const u32 elt = galois_elt_from_step(c, trie[kid].term);
        if (acc && trie[kid].kids.empty() && trie[kid].mult == 1) {
            int rc = fc_leaf(c, parent, elt, *acc, B);
            if (rc) return rc;
            continue;
        }
        u64 *cur = bufs + (size_t)depth * B * ctw;
        int rc = op_apply_galois(c, parent, elt, cur, B);
        if (rc) return rc;
        for (int m = 0; m < trie[kid].mult; ++m) op_add(c, out, cur, out, B, 2);
        if ((rc = fc_dfs(c, trie, kid, depth + 1, cur, bufs, out, acc, B))) return rc;
    }
    return HHE_OK;
}
}  // namespace

// one chunk on lane `ln`, enqueued asynchronously; shared = the shared-digit evaluation (raises *ln.zero_flag when it is not exact)
static int fc_row_chunk(hhe_ctx *c, Lane &ln, bool shared, const uint64_t *vi, const uint64_t *w, size_t W, size_t n_inputs,
                        int default_galois_only, uint64_t *out, size_t B);

// hhe_fc_row with the key objects already named: c->rks relinearizes the product, c->gks serves the rotation sum
static int fc_row_impl(hhe_ctx *c, const uint64_t *vi, const uint64_t *w, size_t W, size_t n_inputs, int default_galois_only, uint64_t *out, size_t B)
{
    if (!c || !vi || !w || !out || W == 0 || B == 0 || n_inputs == 0 || n_inputs > c->n / 2)
        return fail(HHE_ERR_INVALID, "hhe_fc_row: bad arguments");
    if (!c->rks->rk) return fail(HHE_ERR_NO_RELIN_KEY, "relinearization key not set");
    // chunks bound the key-switch working set (digit transforms per trie level); a chunk is a multiple of W so that item i
    // of a chunk still uses weight row i % W.  Chunks are independent: with more than one internal stream (HHE_STREAMS)
    // they run round-robin on the streams, as in hhe_pasta3_transcipher.
    size_t per = c->fc_chunk ? c->fc_chunk : B;
    if (per < B) per = std::max<size_t>(W, per / W * W);
    per = std::min(per, B);
    const size_t nch = (B + per - 1) / per, ctw = c->ct_words();
    const int ns = nch > 1 ? c->nstreams : 0;
    Lane &main = c->lanes[0];
    const bool shared = c->fc_shared != 0;
    u32 *flags = nullptr;
    if (shared) {
        if (c->flags_cap < nch) {  // grow-only per-chunk flags
            sync_ctx(c);
            rt_free(c->d_flags);
            c->flags_cap = 0;
            if (!(c->d_flags = (u32 *)rt_malloc(nch * 4))) return dev_fail("hhe_fc_row");
            c->flags_cap = nch;
        }
        flags = c->d_flags;
        rt_memset(flags, 0, nch * 4, main.stream);
    }
    int rc = HHE_OK;
    if (ns) {
        rt_event_record(c->ev_fork, main.stream);
        for (int s = 1; s <= ns; ++s) rt_stream_wait_event(c->lanes[s].stream, c->ev_fork);
    }
    size_t idx = 0;
    for (size_t b0 = 0; b0 < B && !rc; b0 += per, ++idx) {
        Lane &ln = ns ? c->lanes[1 + idx % ns] : main;
        ln.zero_flag = shared ? flags + idx : nullptr;
        rc = fc_row_chunk(c, ln, shared, vi + b0 * ctw, w, W, n_inputs, default_galois_only, out + b0 * ctw, std::min(per, B - b0));
    }
    for (int s = 1; s <= ns; ++s) {
        rt_event_record(c->lanes[s].ev_done, c->lanes[s].stream);
        rt_stream_wait_event(main.stream, c->lanes[s].ev_done);
    }
    c->w = &main;
    if (shared && !rc) {
        std::vector<u32> h(nch, 0);
        if (rt_d2h(h.data(), flags, nch * 4, main.stream) || rt_sync(main.stream)) rc = dev_fail("hhe_fc_row");
        // a zero coefficient in some c1 (probability ~ N/q per ciphertext): that chunk is recomputed with per-child transforms
        idx = 0;
        for (size_t b0 = 0; b0 < B && !rc; b0 += per, ++idx)
            if (h[idx] || c->fc_shared == 2) {
                c->fc_fallbacks++;
                main.zero_flag = nullptr;
                rc = fc_row_chunk(c, main, false, vi + b0 * ctw, w, W, n_inputs, default_galois_only, out + b0 * ctw, std::min(per, B - b0));
            }
    }
    if (rt_sync(main.stream) && !rc) rc = dev_fail("hhe_fc_row");
    return rc;
}
extern "C" int hhe_fc_row_ks(hhe_ctx *c, const hhe_keyset *rk, const hhe_keyset *gk, const uint64_t *vi, const uint64_t *w, size_t W, size_t n_inputs,
                             uint64_t *out, size_t B)
{
    HHE_LOCK(c);
    if (!c) return fail(HHE_ERR_INVALID, "hhe_fc_row: null context");
    int rc = check_sets(c, rk, gk);
    if (rc) return rc;
    KeyScope keys(c, gk, rk);
    return fc_row_impl(c, vi, w, W, n_inputs, 0, out, B);  // the set IS the GaloisKeys object: every key it holds is visible to rotate_rows
}
extern "C" int hhe_fc_row(hhe_ctx *c, const uint64_t *vi, const uint64_t *w, size_t W, size_t n_inputs, int relin_slot,
                          int default_galois_only, uint64_t *out, size_t B)
{
    HHE_LOCK(c);
    if (!c || relin_slot < 0 || relin_slot >= HHE_RELIN_SLOTS) return fail(HHE_ERR_INVALID, "hhe_fc_row: bad arguments");
    KeyScope keys(c, nullptr, c->relin_set(relin_slot));
    return fc_row_impl(c, vi, w, W, n_inputs, default_galois_only, out, B);
}

static int fc_row_chunk(hhe_ctx *c, Lane &lane, bool shared, const uint64_t *vi, const uint64_t *w, size_t W, size_t n_inputs,
                        int default_galois_only, uint64_t *out, size_t B)
{
    c->w = &lane;
    int rc = lane_reserve(c, lane, B);
    if (rc) return rc;
    const int L = c->L;
    const size_t ctw = c->ct_words();
    // trie of NAF term sequences (terms equal to +-N/2 are skipped, evaluator.h rotate_internal)
    std::vector<NafNode> trie(1);
    int max_depth = 0;
    for (size_t i = 1; i < n_inputs; ++i) {
        const int step = -(int)i;
        std::vector<int> terms;
        const u32 direct = galois_elt_from_step(c, step);
        const bool pow2 = (i & (i - 1)) == 0;
        if (c->gks->gk.count(direct) && (pow2 || !default_galois_only)) terms.push_back(step);  // has_key(elt): one key switch
        else
            for (int t : nt_naf(step))
                if ((size_t)std::abs(t) != c->n / 2) terms.push_back(t);
        if (terms.size() == 1 && !c->gks->gk.count(galois_elt_from_step(c, terms[0]))) return fail(HHE_ERR_NO_GALOIS_KEY, "Galois key not present");
        int node = 0;
        for (int t : terms) {
            if (!c->gks->gk.count(galois_elt_from_step(c, t))) return fail(HHE_ERR_NO_GALOIS_KEY, "Galois key not present");
            int next = -1;
            for (int kid : trie[node].kids)
                if (trie[kid].term == t) { next = kid; break; }
            if (next < 0) {
                next = (int)trie.size();
                trie.push_back(NafNode());
                trie[next].term = t;
                trie[node].kids.push_back(next);
            }
            node = next;
        }
        trie[node].mult++;
        max_depth = std::max(max_depth, (int)terms.size());
    }
    Lane &ln = *c->w;
    if (ln.rot_cap < B || max_depth + 1 > 16) {
        if (max_depth + 1 > 16) return fail(HHE_ERR_INVALID, "hhe_fc_row: NAF depth");
        rt_sync(ln.stream);
        rt_free(ln.ws_rot);
        ln.ws_rot = (u64 *)rt_malloc(B * 16 * ctw * 8);
        if (!ln.ws_rot) return dev_fail("hhe_fc_row workspace");
        ln.rot_cap = B;
    }
    u64 *wb = ln.ws_ct[0], *prod = ln.ws_rot;  // depth-0 buffer holds the product
    op_elt(c, ELT_BCAST, nullptr, w, wb, B * 2 * L, 0, L, (int)(W * 2 * L));
    op_multiply(c, vi, wb, ln.ws_ct3, B);                                  // packed_enc_multiply
    rc = op_relinearize(c, ln.ws_ct3, prod, B);                            // CSP.cpp:306 (the RelinKeys object the call names: c->rks)
    if (rc) return rc;
    const size_t bln = B * (size_t)L * c->n;
    // one evaluation of the rotation trie; shared = children of a node reuse the digit transforms of its c1
    auto run = [&](bool shared) -> int {
        rt_d2d(out, prod, B * ctw * 8, ln.stream);
        auto dfs = [&](const FcLeafAcc *acc) {
            return shared ? fc_dfs_shared(c, trie, max_depth, prod, out, acc, B) : fc_dfs(c, trie, 0, 1, prod, ln.ws_rot, out, acc, B);
        };
        if (!c->fc_leaf_sums) return dfs(nullptr);
        FcLeafAcc acc;
        const size_t leaf_words = B * 2 * (size_t)HHE_LEAF_GROUP * c->n;
        if (ln.leaf_cap < B) {
            rt_sync(ln.stream);
            rt_free(ln.ws_leaf);
            ln.leaf_cap = 0;
            if (!(ln.ws_leaf = (u64 *)rt_malloc(2 * leaf_words * 8))) return dev_fail("hhe_fc_row workspace");
            ln.leaf_cap = B;
        }
        acc.accS = ln.ws_ct[0]; acc.accH = ln.ws_ct[1]; acc.rscr = ln.ws_leaf + ln.leaf_cap * 2 * (size_t)HHE_LEAF_GROUP * c->n;
        rt_memset(acc.accS, 0, 2 * bln * 8, ln.stream);
        rt_memset(acc.accH, 0, 2 * bln * 8, ln.stream);
        int r = dfs(&acc);
        if (r) return r;
        op_ntt(c, acc.accS, B * 2 * L, 0, L, true);
        LeafSumArgs ls;
        memset(&ls, 0, sizeof(ls));
        ls.accS = acc.accS; ls.accH = acc.accH; ls.out = out; ls.mods = c->d_mods; ls.logn = c->logn;
        ls.B = (int)B; ls.L = L; ls.ks = c->ksc;
        k_leaf_sum(ls, ln.stream);
        return HHE_OK;
    };
    if (!shared || max_depth == 0) return run(false);
    return run(true);
}
