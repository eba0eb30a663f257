// hhe_client_bodies.h -- client / analyst ends of the protocol on the device.
// (1) The plain PASTA-3 cipher:
// keystream blocks of pasta::Pasta::gen_keystream (src/pasta/pasta_3_plain.cpp:156-173) for many block counters at once,
// and PASTA::encrypt / decrypt (:9-46) over batches of records.  Two kernels:
//   pasta_xof_body    one lane per block counter: SHAKE128(BE64(nonce) || BE64(counter)) squeezed as big-endian 64-bit
//                     words, masked to bitlen(t), rejection-sampled (:56-84) into the 4 x 512 field elements a block
//                     consumes (per affine layer: first row of matrix 1, of matrix 2, rc 1, rc 2 -- the draw order of
//                     linear_layer, :205-211)
//   pasta_plain_phase one workgroup (256 lanes = 2 state halves x 128 columns) per block counter; the sequential
//                     matrix rows (calculate_row, :86-100) are rebuilt on the fly in LDS (two rows live), never stored.
// Bodies are functions of (block, lane, phase) with a workgroup barrier between phases, so the tests-only CPU emulator
// runs the same code.
#pragma once
#include "hhe_common.h"
#include "hhe_modarith.h"

constexpr int PASTA_RAND_PER_BLOCK = (PASTA_R + 1) * 4 * PASTA_T;  // 2048 field elements per block
constexpr int PASTA_PLAIN_THREADS = 2 * PASTA_T;

struct PastaXofArgs {
    u64 t, mask, nonce, first_block;
    int nblocks;
    u64 *rand;  // [nblocks][4][4][128]
};
struct PastaPlainArgs {
    u64 t, r_lo, r_hi;
    const u64 *rand;  // [nblocks][4][4][128]
    const u64 *key;   // [256] secret key words (device)
    u64 *ks;          // [nblocks][128] keystream blocks
    int nblocks;
};
struct PastaCryptArgs {  // records [S][nwords]; word i of a record uses keystream block i / 128, element i % 128
    u64 t, r_hi;  // r_hi = floor(2^64 / t)
    const u64 *in, *ks;
    u64 *out;
    size_t S, nwords;
    int decrypt;
};

// ------------------------------------------------------------------ SHAKE128 (FIPS 202), one sponge per lane
HD u64 pasta_rol(u64 x, int s) { return (x << s) | (x >> (64 - s)); }
HD u64 pasta_bswap(u64 x)
{
    x = ((x & 0x00ff00ff00ff00ffULL) << 8) | ((x >> 8) & 0x00ff00ff00ff00ffULL);
    x = ((x & 0x0000ffff0000ffffULL) << 16) | ((x >> 16) & 0x0000ffff0000ffffULL);
    return (x << 32) | (x >> 32);
}
// Keccak-f[1600] on 25 named lanes (constant indices only, so the state lives in registers)
HD void keccak_f1600(u64 *A)
{
    constexpr u64 RC[24] = {0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL,
                            0x000000000000808bULL, 0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL,
                            0x000000000000008aULL, 0x0000000000000088ULL, 0x0000000080008009ULL, 0x000000008000000aULL,
                            0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL, 0x8000000000008003ULL,
                            0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800aULL, 0x800000008000000aULL,
                            0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
    for (int round = 0; round < 24; round++) {
        u64 C0 = A[0] ^ A[5] ^ A[10] ^ A[15] ^ A[20], C1 = A[1] ^ A[6] ^ A[11] ^ A[16] ^ A[21];
        u64 C2 = A[2] ^ A[7] ^ A[12] ^ A[17] ^ A[22], C3 = A[3] ^ A[8] ^ A[13] ^ A[18] ^ A[23];
        u64 C4 = A[4] ^ A[9] ^ A[14] ^ A[19] ^ A[24];
        const u64 D0 = C4 ^ pasta_rol(C1, 1), D1 = C0 ^ pasta_rol(C2, 1), D2 = C1 ^ pasta_rol(C3, 1);
        const u64 D3 = C2 ^ pasta_rol(C4, 1), D4 = C3 ^ pasta_rol(C0, 1);
        // theta + rho + pi: B[y + 5*((2x+3y)%5)] = rol(A[x + 5y] ^ D[x], r[x][y])
        const u64 B0 = A[0] ^ D0, B10 = pasta_rol(A[1] ^ D1, 1), B20 = pasta_rol(A[2] ^ D2, 62);
        const u64 B5 = pasta_rol(A[3] ^ D3, 28), B15 = pasta_rol(A[4] ^ D4, 27);
        const u64 B16 = pasta_rol(A[5] ^ D0, 36), B1 = pasta_rol(A[6] ^ D1, 44), B11 = pasta_rol(A[7] ^ D2, 6);
        const u64 B21 = pasta_rol(A[8] ^ D3, 55), B6 = pasta_rol(A[9] ^ D4, 20);
        const u64 B7 = pasta_rol(A[10] ^ D0, 3), B17 = pasta_rol(A[11] ^ D1, 10), B2 = pasta_rol(A[12] ^ D2, 43);
        const u64 B12 = pasta_rol(A[13] ^ D3, 25), B22 = pasta_rol(A[14] ^ D4, 39);
        const u64 B23 = pasta_rol(A[15] ^ D0, 41), B8 = pasta_rol(A[16] ^ D1, 45), B18 = pasta_rol(A[17] ^ D2, 15);
        const u64 B3 = pasta_rol(A[18] ^ D3, 21), B13 = pasta_rol(A[19] ^ D4, 8);
        const u64 B14 = pasta_rol(A[20] ^ D0, 18), B24 = pasta_rol(A[21] ^ D1, 2), B9 = pasta_rol(A[22] ^ D2, 61);
        const u64 B19 = pasta_rol(A[23] ^ D3, 56), B4 = pasta_rol(A[24] ^ D4, 14);
        // chi
        A[0] = B0 ^ (~B1 & B2); A[1] = B1 ^ (~B2 & B3); A[2] = B2 ^ (~B3 & B4); A[3] = B3 ^ (~B4 & B0); A[4] = B4 ^ (~B0 & B1);
        A[5] = B5 ^ (~B6 & B7); A[6] = B6 ^ (~B7 & B8); A[7] = B7 ^ (~B8 & B9); A[8] = B8 ^ (~B9 & B5); A[9] = B9 ^ (~B5 & B6);
        A[10] = B10 ^ (~B11 & B12); A[11] = B11 ^ (~B12 & B13); A[12] = B12 ^ (~B13 & B14); A[13] = B13 ^ (~B14 & B10); A[14] = B14 ^ (~B10 & B11);
        A[15] = B15 ^ (~B16 & B17); A[16] = B16 ^ (~B17 & B18); A[17] = B17 ^ (~B18 & B19); A[18] = B18 ^ (~B19 & B15); A[19] = B19 ^ (~B15 & B16);
        A[20] = B20 ^ (~B21 & B22); A[21] = B21 ^ (~B22 & B23); A[22] = B22 ^ (~B23 & B24); A[23] = B23 ^ (~B24 & B20); A[24] = B24 ^ (~B20 & B21);
        A[0] ^= RC[round];  // iota
    }
}

// Pasta::init_shake + generate_random_field_element (pasta_3_plain.cpp:56-84) for the whole block: element p of the
// block is drawn with "zero allowed" iff it is a round constant (p % 512 >= 256); first-row elements reject zero (:286-295)
HD void pasta_xof_body(const PastaXofArgs &a, size_t gid)
{
    if (gid >= (size_t)a.nblocks) return;
    u64 A[25];
#pragma unroll
    for (int i = 0; i < 25; i++) A[i] = 0;
    // message = BE64(nonce) || BE64(counter); lanes are little-endian, so each lane is the byte swap of its word
    A[0] = pasta_bswap(a.nonce);
    A[1] = pasta_bswap(a.first_block + gid);
    A[2] = 0x1f;                    // SHAKE domain suffix right after the 16 message bytes
    A[20] = 0x8000000000000000ULL;  // final pad bit of the 168-byte rate
    keccak_f1600(A);
    u64 *out = a.rand + gid * PASTA_RAND_PER_BLOCK;
    int count = 0;
    while (count < PASTA_RAND_PER_BLOCK) {
#pragma unroll
        for (int k = 0; k < 21; k++) {  // one squeezed rate block = 21 big-endian 64-bit words
            const u64 e = pasta_bswap(A[k]) & a.mask;
            const bool rc = (count & 511) >= 256;
            if (count < PASTA_RAND_PER_BLOCK && e < a.t && (rc || e != 0)) out[count++] = e;
        }
        keccak_f1600(A);
    }
}

// ------------------------------------------------------------------ keystream block
// LDS words of one workgroup
constexpr int PP_X = 0, PP_ROW0 = 256, PP_ROW = 512, PP_PROD = 1024, PP_PART = 1024 + 8 * 256, PP_Y = PP_PART + 256;
constexpr int PASTA_PLAIN_LDS = PP_Y + 256;
// phases of one affine layer: 0 = first row, 1..127 = next rows, then after every 8th row two reduction phases;
// the schedule below is a flat list so that GPU kernel and emulator walk it identically
constexpr int PP_STEPS_PER_LAYER = PASTA_T + 2 * (PASTA_T / 8) + 3;  // rows + reductions + (affine/mix, sbox a, sbox b)

// One phase of layer `layer` for lane tau of the block's workgroup.  step: 0..127 row i; 128+2g / 129+2g reduction of
// row group g (rows 8g..8g+7) -- issued by the driver loop right after row 8g+7; PP_* finish phases at the end.
enum { PP_ROWSTEP = 0, PP_RED1 = 1, PP_RED2 = 2, PP_AFFINE = 3, PP_SBOX_A = 4, PP_SBOX_B = 5, PP_INIT = 6, PP_OUT = 7 };
HD void pasta_plain_phase(const PastaPlainArgs &a, int block, int tau, int layer, int kind, int i, u64 *lds)
{
    ModDev m;
    m.q = a.t; m.r_lo = a.r_lo; m.r_hi = a.r_hi;
    const u64 t = a.t;
    const int h = tau >> 7, j = tau & (PASTA_T - 1);
    const u64 *rnd = a.rand + ((size_t)block * (PASTA_R + 1) + layer) * 4 * PASTA_T;
    switch (kind) {
    case PP_INIT: lds[PP_X + tau] = a.key[tau]; break;  // state1_ = key[0..128), state2_ = key[128..256) (:160-163)
    case PP_ROWSTEP: {  // Pasta::matmul (:262-279): row i of both matrices and its products with the state
        u64 v;
        if (i == 0) {
            v = rnd[h * PASTA_T + j];
            lds[PP_ROW0 + tau] = v;
        } else {  // calculate_row (:86-100): first_row[j] * prev[127] + prev[j-1]
            const u64 *prev = lds + PP_ROW + ((i - 1) & 1) * 256 + h * PASTA_T;
            v = mulmod(lds[PP_ROW0 + tau], prev[PASTA_T - 1], m);
            if (j) v = addmod(v, prev[j - 1], t);
        }
        lds[PP_ROW + (i & 1) * 256 + tau] = v;
        lds[PP_PROD + (i & 7) * 256 + tau] = mulmod(v, lds[PP_X + tau], m);
        break;
    }
    case PP_RED1: {  // 16 sums (8 rows x 2 halves) of 128 products: 16 lanes per sum, 8 products each
        const int s = tau >> 4, p = tau & 15, r = s >> 1, hh = s & 1;
        const u64 *pr = lds + PP_PROD + r * 256 + hh * PASTA_T + p * 8;
        u64 acc = 0;
        for (int k = 0; k < 8; k++) acc = addmod(acc, pr[k], t);
        lds[PP_PART + tau] = acc;
        break;
    }
    case PP_RED2:
        if (tau < 16) {  // i = index of the group's first row
            const int r = tau >> 1, hh = tau & 1;
            u64 acc = 0;
            for (int k = 0; k < 16; k++) acc = addmod(acc, lds[PP_PART + tau * 16 + k], t);
            lds[PP_Y + hh * PASTA_T + i + r] = acc;
        }
        break;
    case PP_AFFINE: {  // add_rc (:205-211) + mix (:246-257)
        const u64 s1 = addmod(lds[PP_Y + j], rnd[2 * PASTA_T + j], t), s2 = addmod(lds[PP_Y + PASTA_T + j], rnd[3 * PASTA_T + j], t);
        const u64 sum = addmod(s1, s2, t);
        lds[PP_X + tau] = addmod(h ? s2 : s1, sum, t);
        break;
    }
    case PP_SBOX_A:  // sbox_feistel (:229-240) / sbox_cube (:218-225) into Y, copied back by PP_SBOX_B
        if (layer == PASTA_R - 1) {
            const u64 x = lds[PP_X + tau];
            lds[PP_Y + tau] = mulmod(mulmod(x, x, m), x, m);
        } else {
            const u64 x = lds[PP_X + tau];
            lds[PP_Y + tau] = j ? addmod(x, mulmod(lds[PP_X + tau - 1], lds[PP_X + tau - 1], m), t) : x;
        }
        break;
    case PP_SBOX_B: lds[PP_X + tau] = lds[PP_Y + tau]; break;
    case PP_OUT:
        if (tau < PASTA_T) a.ks[(size_t)block * PASTA_T + tau] = lds[PP_X + tau];  // gen_keystream returns state1_ (:172)
        break;
    }
}
// the phase list of one block, shared by the kernel and the emulator: F(kind, layer, i) is called once per phase and
// must run the phase for all 256 lanes followed by a barrier
template <class F> HD void pasta_plain_schedule(F &&f)
{
    f(PP_INIT, 0, 0);
    for (int layer = 0; layer <= PASTA_R; layer++) {
        for (int i = 0; i < PASTA_T; i++) {
            f(PP_ROWSTEP, layer, i);
            if ((i & 7) == 7) { f(PP_RED1, layer, i - 7); f(PP_RED2, layer, i - 7); }
        }
        f(PP_AFFINE, layer, 0);
        if (layer < PASTA_R) { f(PP_SBOX_A, layer, 0); f(PP_SBOX_B, layer, 0); }
    }
    f(PP_OUT, PASTA_R, 0);
}

// PASTA::encrypt / decrypt (pasta_3_plain.cpp:9-46), element-wise over [S][nwords], literally: encrypt
// (v + ks) % t on 64-bit words, decrypt v - ks with one conditional + t (no reduction of an input that is not below t)
HD void pasta_crypt_body(const PastaCryptArgs &a, size_t gid)
{
    if (gid >= a.S * a.nwords) return;
    const size_t i = gid % a.nwords;
    const u64 k = a.ks[i];
    u64 v = a.in[gid];
    if (a.decrypt) {
        if (k > v) v += a.t;
        a.out[gid] = v - k;
    } else {
        ModDev m;
        m.q = a.t; m.r_hi = a.r_hi;
        a.out[gid] = reduce64(v + k, m);
    }
}

// ------------------------------------------------------------------ analyst side: batched BFV decryption
// Decryptor::bfv_decrypt (seal/decryptor.h:70): phase = c0 + c1*s per data prime (NTT launches with the
// secret key as STORE_MUL operand), then RNSTool::decrypt_scale_and_round (seal/util/rns.h:230-243): multiply by
// t*gamma, fast base conversion q -> {t, gamma}, multiply by -Q^-1, correct by the centred gamma residue, times gamma^-1.
struct DecryptArgs {
    const ModDev *mods;
    const u64 *ct;   // [B][2][L][N]; c0 is read here
    const u64 *c1s;  // [B][L][N]  c1 * s, coefficient form
    u64 *plain;      // [B][N] coefficients mod t
    int logn, L;
    size_t B;
    u64 t, t_rlo, t_rhi, gamma, g_rlo, g_rhi;
    u64 neg_inv_q_t, neg_inv_q_g, inv_g_t;
    u64 cj[HHE_MAXL];  // t*gamma * (Q/q_j)^-1 mod q_j
    u64 pt[HHE_MAXL];  // Q/q_j mod t
    u64 pg[HHE_MAXL];  // Q/q_j mod gamma
};
HD void decrypt_round_body(const DecryptArgs &a, size_t gid)
{
    const size_t n = (size_t)1 << a.logn;
    if (gid >= a.B * n) return;
    const size_t b = gid >> a.logn, i = gid & (n - 1);
    ModDev mt, mg;
    mt.q = a.t; mt.r_lo = a.t_rlo; mt.r_hi = a.t_rhi;
    mg.q = a.gamma; mg.r_lo = a.g_rlo; mg.r_hi = a.g_rhi;
    Acc128 st = {0, 0}, sg = {0, 0};
    for (int j = 0; j < a.L; j++) {
        const ModDev m = mod_at(a.mods, j);
        const u64 ph = addmod(a.ct[(b * 2 * a.L + j) * n + i], a.c1s[(b * a.L + j) * n + i], m.q);
        const u64 v = mulmod(ph, a.cj[j], m);
        acc_mac(st, reduce64(v, mt), a.pt[j]);
        acc_mac(sg, reduce64(v, mg), a.pg[j]);
    }
    u64 vt = mulmod(barrett128(st.lo, st.hi, mt), a.neg_inv_q_t, mt);
    const u64 vg = mulmod(barrett128(sg.lo, sg.hi, mg), a.neg_inv_q_g, mg);
    u64 d;
    if (vg > (a.gamma >> 1)) d = addmod(vt, reduce64(a.gamma - vg, mt), a.t);
    else d = submod(vt, reduce64(vg, mt), a.t);
    a.plain[gid] = mulmod(d, a.inv_g_t, mt);
}
// BatchEncoder::decode (seal/batchencoder.h:282): after the forward NTT mod t, slot i = ntt[index_map[i]]
struct DecodeArgs {
    const u64 *in;  // [B][N] NTT mod t of the plaintext
    u64 *vals;      // [B][N]
    const u32 *slot_map;
    int logn;
    size_t B;
};
HD void decode_gather_body(const DecodeArgs &a, size_t gid)
{
    const size_t n = (size_t)1 << a.logn;
    if (gid >= a.B * n) return;
    const size_t b = gid >> a.logn, i = gid & (n - 1);
    a.vals[gid] = a.in[b * n + a.slot_map[i]];
}
