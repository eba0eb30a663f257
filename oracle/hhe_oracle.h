/*
 * hhe_oracle.h -- CPU ORACLE (test infrastructure, NOT the product).
 *
 * A plain-C restatement of the reference's CSP hot path: PASTA-3 -> BFV
 * transciphering (src/pasta/pasta_3_seal.cpp:106-172) followed by the packed
 * BFV linear layer (src/util/sealhelper.cpp:268-274, 379-392), including the
 * slice of Microsoft SEAL 4.0.0 it runs on.  SEAL 4.0.0 is present in the
 * reference only as headers + a prebuilt static library (libs/seal/); the
 * prebuilt library is never linked or loaded here.  Its algorithms are restated
 * from the SEAL 4.0.0 headers (file:line cited per function) and from the
 * arithmetic specification in SURVEY.md Appendix A, which the survey verified
 * word-for-word against that library.
 *
 * PARITY PINNING: the PASTA-3 public randomness / plain cipher are pinned by
 * known answers and by the reference's own pasta_3_plain.cpp
 * compiled from source (oracle/_ref).  Parameter derivation is pinned by the
 * concrete primes/roots/Galois elements of SURVEY.md A.1/A.3/A.7/A.10.  The BFV
 * ciphertext words are pinned only through decrypt-correctness (the reference's
 * own end-to-end checks, hhe_pktnn_examples.cpp:639-648, 692-699) --
 * ciphertext-bit parity against libseal is otherwise UNPINNED.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use
 * this library, and only as the checker / reported CPU baseline.
 */
#ifndef HHE_ORACLE_H
#define HHE_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_MAXK 32
#define PASTA_T 128
#define PASTA_R 3

typedef struct orc_ctx orc_ctx;

/* ---- number theory (seal/util/numth.h:136-157) ---- */
int orc_is_prime(uint64_t v);
/* get_primes(factor, bit_size, count): descending primes = 1 mod factor */
int orc_get_primes(uint64_t factor, int bit_size, size_t count, uint64_t *out);
/* CoeffModulus::Create(N, bit_sizes) ordering (seal/modulus.h) */
int orc_coeff_modulus_create(size_t n, const int *bit_sizes, size_t count, uint64_t *out);
uint64_t orc_minimal_primitive_root(uint64_t degree, uint64_t q);
/* util::naf (seal/util/numth.h:22-42); returns count */
int orc_naf(int value, int *out);

/* ---- context ---- */
orc_ctx *orc_ctx_create(int logn, int K, const uint64_t *q, uint64_t t);
void orc_ctx_destroy(orc_ctx *c);
size_t orc_ctx_n(const orc_ctx *c);
int orc_ctx_L(const orc_ctx *c);
int orc_ctx_K(const orc_ctx *c);
/* query derived constants for tests: what = "root"(i<K), "bsk"(i<=L: B.., m_sk), "gamma" */
uint64_t orc_ctx_query(const orc_ctx *c, const char *what, int i);
void orc_ctx_ntt_table(const orc_ctx *c, int mod_index, int inverse, int shoup, uint64_t *out);

/* ---- NTT (seal/util/dwthandler.h:94-356). mod_index: 0..K-1 coeff primes,
 *      K..K+L Bsk primes (B.., m_sk), -1 = plain modulus t ---- */
void orc_ntt_fwd(const orc_ctx *c, int mod_index, uint64_t *a);
void orc_ntt_inv(const orc_ctx *c, int mod_index, uint64_t *a);

/* ---- BatchEncoder (seal/batchencoder.h:80-217) ---- */
void orc_encode(const orc_ctx *c, const uint64_t *vals, size_t count, uint64_t *plain);
void orc_decode(const orc_ctx *c, const uint64_t *plain, uint64_t *vals);

/* ---- Galois (seal/util/galois.h:32,124,143-153) ---- */
uint32_t orc_galois_elt_from_step(const orc_ctx *c, int step);
int orc_galois_elts_all(const orc_ctx *c, uint32_t *out);
void orc_apply_galois_poly(const orc_ctx *c, int mod_index, uint32_t elt, const uint64_t *in, uint64_t *out);

/* ---- keys / encryption (oracle-side only; not bit-compatible with SEAL's PRNG) ----
 * sk: [K][N] NTT form. pk: [2][K][N] NTT form (key level).
 * kswitch key: [L][2][K][N] NTT form (seal/kswitchkeys.h:90-130). */
void orc_keygen_secret(const orc_ctx *c, uint64_t seed, uint64_t *sk);
void orc_keygen_public(const orc_ctx *c, const uint64_t *sk, uint64_t seed, uint64_t *pk);
void orc_keygen_relin(const orc_ctx *c, const uint64_t *sk, uint64_t seed, uint64_t *ksk);
void orc_keygen_galois(const orc_ctx *c, const uint64_t *sk, uint32_t elt, uint64_t seed, uint64_t *ksk);
/* ct: [2][L][N] coefficient form, data level */
void orc_encrypt(const orc_ctx *c, const uint64_t *pk, const uint64_t *plain, uint64_t seed, uint64_t *ct);
void orc_encrypt_symmetric(const orc_ctx *c, const uint64_t *sk, const uint64_t *plain, uint64_t seed, uint64_t *ct);
/* ct of `size` polys (2 or 3) -> plain[N] */
void orc_decrypt(const orc_ctx *c, const uint64_t *sk, const uint64_t *ct, int size, uint64_t *plain);
/* phase c0 + c1 s (+ c2 s^2) per data limb [L][N], coefficient form (for noise measurement in tests) */
void orc_phase(const orc_ctx *c, const uint64_t *sk, const uint64_t *ct, int size, uint64_t *ph);

/* ---- Evaluator ops (seal/evaluator.h). All cts coefficient form ---- */
void orc_add(const orc_ctx *c, const uint64_t *a, const uint64_t *b, int size, uint64_t *out);
void orc_negate(const orc_ctx *c, const uint64_t *a, int size, uint64_t *out);
void orc_add_plain(const orc_ctx *c, const uint64_t *a, const uint64_t *plain, uint64_t *out);
void orc_sub_plain(const orc_ctx *c, const uint64_t *a, const uint64_t *plain, uint64_t *out);
void orc_multiply_plain(const orc_ctx *c, const uint64_t *a, const uint64_t *plain, uint64_t *out);
/* key-switch: target poly d [L][N] coefficient; adds result into ct (size 2) */
void orc_switch_key(const orc_ctx *c, uint64_t *ct, const uint64_t *d, const uint64_t *ksk);
void orc_apply_galois(const orc_ctx *c, const uint64_t *a, uint32_t elt, const uint64_t *ksk, uint64_t *out);
void orc_multiply(const orc_ctx *c, const uint64_t *a, const uint64_t *b, uint64_t *out3);
void orc_relinearize(const orc_ctx *c, const uint64_t *a3, const uint64_t *rk, uint64_t *out2);

/* Galois key set: elts[nk], keys [nk][L][2][K][N] */
typedef struct {
    int nk;
    const uint32_t *elts;
    const uint64_t *keys;
} orc_gkeys;
/* Evaluator::rotate_rows with NAF fallback (seal/evaluator.h:955-1060). returns #key-switches or -1 */
int orc_rotate_rows(const orc_ctx *c, const uint64_t *a, int step, const orc_gkeys *gk, uint64_t *out);
int orc_rotate_columns(const orc_ctx *c, const uint64_t *a, const orc_gkeys *gk, uint64_t *out);

/* ---- PASTA-3 (src/pasta/pasta_3_plain.cpp) ---- */
typedef struct {
    uint64_t st[25];
    uint8_t buf[168];
    int pos;
} orc_shake;
void orc_shake128_init(orc_shake *s, const uint8_t *in, size_t len);
void orc_shake128_squeeze(orc_shake *s, uint8_t *out, size_t len);
/* one block's public randomness: mats [4][2][128][128], rcs [4][2][128] */
void orc_pasta_block_randomness(uint64_t t, uint64_t nonce, uint64_t block, uint64_t *mats, uint64_t *rcs);
void orc_pasta_keystream(uint64_t t, const uint64_t *key256, uint64_t nonce, uint64_t block, uint64_t *ks128);
void orc_pasta_encrypt(uint64_t t, const uint64_t *key256, const uint64_t *pt, size_t n, uint64_t *ct);
void orc_pasta_decrypt(uint64_t t, const uint64_t *key256, const uint64_t *ct, size_t n, uint64_t *pt);

/* ---- the hot path ---- */
/* PASTA_SEAL::encrypt_key_2 packing (pasta_3_seal.cpp:23-38): plain[N] from 256 key words */
void orc_pasta_pack_key(const orc_ctx *c, const uint64_t *key256, uint64_t *plain);
/* PASTA_SEAL::decomposition (pasta_3_seal.cpp:106-172) for ONE block `block_index`
 * (cipher words cw[ncw<=128]); enc_key ct [2][L][N]; out ct [2][L][N].
 * gk must hold elts for steps {-1,+128(if N/2!=128), columns}. returns 0 or -1. */
int orc_pasta_transcipher_block(const orc_ctx *c, const uint64_t *enc_key, const uint64_t *rk,
                                const orc_gkeys *gk, const uint64_t *cw, size_t ncw,
                                uint64_t block_index, int use_bsgs, uint64_t *out);
/* batch over blocks with OpenMP: cw [nb][128] (ncw each), block_index[nb], out [nb][2][L][N] */
int orc_pasta_transcipher_batch(const orc_ctx *c, const uint64_t *enc_key, const uint64_t *rk,
                                const orc_gkeys *gk, const uint64_t *cw, const uint32_t *ncw,
                                const uint64_t *block_index, size_t nb, int threads, uint64_t *out);
/* SEALZpCipher::mask / flatten (SEAL_Cipher.cpp:161-181) */
void orc_mask(const orc_ctx *c, const uint64_t *a, const uint64_t *mask_vals, size_t count, uint64_t *out);
int orc_flatten(const orc_ctx *c, const uint64_t *blocks, size_t nblocks, const orc_gkeys *gk, uint64_t *out);
/* FC row: packed_enc_multiply + relinearize + encrypted_vec_sum (sealhelper.cpp:268-274,379-392) */
int orc_fc_row(const orc_ctx *c, const uint64_t *vi, const uint64_t *w, const uint64_t *rk,
               const orc_gkeys *gk, size_t n_inputs, uint64_t *out);

#ifdef __cplusplus
}
#endif
#endif
