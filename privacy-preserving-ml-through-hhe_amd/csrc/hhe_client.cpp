// hhe_client.cpp -- C ABI of the client / analyst ends: the plain PASTA-3 cipher
// (pasta::PASTA::encrypt / decrypt, pasta::Pasta::keystream -- src/pasta/pasta_3_plain.cpp:9-46,156-178) and batched
// BFV result decryption (sealhelper::decrypting, src/util/sealhelper.cpp:252-266).  Host code only sets up constants and
// launches; the arithmetic is in hhe_client_bodies.h.
#include <cstring>
#include <vector>
#include "hhe_internal.h"
#include "../../include/hhe_gfx950.h"

namespace {
typedef unsigned __int128 u128;
int fail(int code, const std::string &msg) { hhe_set_error(msg); return code; }
int dev_fail(const char *where) { return fail(HHE_ERR_DEVICE, std::string(where) + ": " + rt_last_error()); }
void barrett_ratio(u64 q, u64 &lo, u64 &hi)
{
    const u128 two64 = (u128)1 << 64;
    hi = (u64)(two64 / q);
    lo = (u64)(((two64 % q) << 64) / q);
}

int keystream_into(hhe_ctx *c, const uint64_t *key, uint64_t first_block, size_t nblocks, u64 *ks, rt_stream st)
{
    for (int i = 0; i < 2 * PASTA_T; ++i)
        if (key[i] >= c->t) return fail(HHE_ERR_INVALID, "PASTA secret key word is not below the plain modulus");
    DevBuf rnd(nblocks * PASTA_RAND_PER_BLOCK * 8), dkey(2 * PASTA_T * 8);
    if (!rnd.p || !dkey.p) return dev_fail("hhe_pasta3_plain_keystream");
    if (rt_h2d(dkey.p, key, 2 * PASTA_T * 8, st)) return dev_fail("hhe_pasta3_plain_keystream");
    PastaXofArgs x;
    x.t = c->t; x.nonce = PASTA_NONCE; x.first_block = first_block; x.nblocks = (int)nblocks; x.rand = rnd.w();
    int bits = 0;
    for (u64 v = c->t; v; v >>= 1) ++bits;
    x.mask = bits >= 64 ? ~(u64)0 : (((u64)1 << bits) - 1);  // Pasta::Pasta(): max_prime_size (pasta_3_plain.cpp:150-153)
    k_pasta_xof(x, st);
    PastaPlainArgs p;
    p.t = c->t; barrett_ratio(c->t, p.r_lo, p.r_hi);
    p.rand = rnd.w(); p.key = dkey.w(); p.ks = ks; p.nblocks = (int)nblocks;
    k_pasta_plain(p, st);
    if (rt_sync(st)) return dev_fail("hhe_pasta3_plain_keystream");  // temporaries are released on return
    return HHE_OK;
}
}  // namespace

extern "C" int hhe_pasta3_plain_keystream(hhe_ctx *c, const uint64_t *key, uint64_t first_block, size_t nblocks, uint64_t *ks)
{
    HHE_LOCK(c);
    if (!c || !key || !ks || nblocks == 0 || nblocks > ((size_t)1 << 24))
        return fail(HHE_ERR_INVALID, "hhe_pasta3_plain_keystream: bad arguments");
    return keystream_into(c, key, first_block, nblocks, ks, c->lanes[0].stream);
}

extern "C" int hhe_pasta3_plain_crypt(hhe_ctx *c, const uint64_t *key, const uint64_t *in, size_t S, size_t nwords, int decrypt,
                                      uint64_t *out)
{
    HHE_LOCK(c);
    if (!c || !key || !in || !out || S == 0 || nwords == 0) return fail(HHE_ERR_INVALID, "hhe_pasta3_plain_crypt: bad arguments");
    const size_t nb = (nwords + PASTA_T - 1) / PASTA_T;  // ceil(size / plain_size) (pasta_3_plain.cpp:13)
    rt_stream st = c->lanes[0].stream;
    DevBuf ks(nb * PASTA_T * 8);
    if (!ks.p) return dev_fail("hhe_pasta3_plain_crypt");
    // every PASTA::encrypt call restarts at block counter 0 with the fixed nonce, so all records share one keystream
    int rc = keystream_into(c, key, 0, nb, ks.w(), st);
    if (rc) return rc;
    PastaCryptArgs a;
    u64 lo;
    a.t = c->t; barrett_ratio(c->t, lo, a.r_hi);
    a.in = in; a.ks = ks.w(); a.out = out; a.S = S; a.nwords = nwords; a.decrypt = decrypt != 0;
    k_pasta_crypt(a, st);
    if (rt_sync(st)) return dev_fail("hhe_pasta3_plain_crypt");
    return HHE_OK;
}

extern "C" int hhe_decrypt(hhe_ctx *c, const uint64_t *sk, const uint64_t *ct, size_t B, uint64_t *vals)
{
    HHE_LOCK(c);
    if (!c || !sk || !ct || !vals || B == 0) return fail(HHE_ERR_INVALID, "hhe_decrypt: bad arguments");
    const int L = c->L;
    const size_t n = c->n, ln = (size_t)L * n;
    rt_stream st = c->lanes[0].stream;
    DevBuf dsk(ln * 8), c1s(B * ln * 8), plain(B * n * 8);
    if (!dsk.p || !c1s.p || !plain.p) return dev_fail("hhe_decrypt");
    if (rt_h2d(dsk.p, sk, ln * 8, st)) return dev_fail("hhe_decrypt");  // data-level limbs of the key-level secret key

    // c1 * s: forward transform of c1 with the dyadic product fused into the store, inverse transform
    NttArgs a;
    memset(&a, 0, sizeof(a));
    a.src = ct + ln; a.dst = c1s.w(); a.mods = c->d_mods; a.logn = c->logn; a.count = (int)(B * L);
    a.mod_base = 0; a.mod_cycle = L; a.src_div = 1; a.src_item_polys = L; a.src_item_stride = 2 * ln; a.t = c->t;
    a.load_op = LOAD_PLAIN; a.store_op = STORE_MUL; a.mul = dsk.w(); a.mul_cycle = L; a.mul_item_polys = 1;
    a.L = L; a.K = c->K; a.ks = c->ksc; a.lazy8 = ntt_lazy8(c, 0, L);
    k_ntt(a, false, st);
    NttArgs inv = a;
    inv.src = c1s.w(); inv.src_item_polys = 0; inv.src_item_stride = 0; inv.store_op = STORE_PLAIN; inv.mul = nullptr;
    k_ntt(inv, true, st);

    DecryptArgs d;
    memset(&d, 0, sizeof(d));
    d.mods = c->d_mods; d.ct = ct; d.c1s = c1s.w(); d.plain = plain.w(); d.logn = c->logn; d.L = L; d.B = B;
    d.t = c->t; d.gamma = c->gamma;
    barrett_ratio(c->t, d.t_rlo, d.t_rhi);
    barrett_ratio(c->gamma, d.g_rlo, d.g_rhi);
    const u64 t = c->t, g = c->gamma;
    u64 q_t = 1 % t, q_g = 1 % g;
    for (int j = 0; j < L; ++j) {
        const u64 qj = c->q[j];
        u64 punct_q = 1 % qj, punct_t = 1 % t, punct_g = 1 % g;
        for (int i = 0; i < L; ++i)
            if (i != j) {
                punct_q = nt_mulmod(punct_q, c->q[i] % qj, qj);
                punct_t = nt_mulmod(punct_t, c->q[i] % t, t);
                punct_g = nt_mulmod(punct_g, c->q[i] % g, g);
            }
        const u64 tg = nt_mulmod(t % qj, g % qj, qj);
        d.cj[j] = nt_mulmod(tg, nt_invmod(punct_q, qj), qj);
        d.pt[j] = punct_t;
        d.pg[j] = punct_g;
        q_t = nt_mulmod(q_t, qj % t, t);
        q_g = nt_mulmod(q_g, qj % g, g);
    }
    d.neg_inv_q_t = (t - nt_invmod(q_t, t)) % t;
    d.neg_inv_q_g = (g - nt_invmod(q_g, g)) % g;
    d.inv_g_t = nt_invmod(g % t, t);
    k_decrypt_round(d, st);

    // BatchEncoder::decode: forward NTT mod t, then the slot index map
    NttArgs p;
    memset(&p, 0, sizeof(p));
    p.src = plain.w(); p.dst = plain.w(); p.mods = c->d_mods; p.logn = c->logn; p.count = (int)B;
    p.mod_base = c->mod_t; p.mod_cycle = 1; p.src_div = 1; p.t = c->t; p.mul_cycle = 1; p.mul_item_polys = 1;
    p.L = L; p.K = c->K; p.ks = c->ksc; p.lazy8 = ntt_lazy8(c, c->mod_t, 1);
    k_ntt(p, false, st);
    DecodeArgs g2;
    g2.in = plain.w(); g2.vals = vals; g2.slot_map = c->d_slot_map; g2.logn = c->logn; g2.B = B;
    k_decode_gather(g2, st);
    if (rt_sync(st)) return dev_fail("hhe_decrypt");
    return HHE_OK;
}
