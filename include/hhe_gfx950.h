/*
 * hhe_gfx950.h -- C ABI of libhhe_gfx950.so: the MI355X (gfx950) implementation of the
 * CSP-side hot path of harpocrates-project/Privacy-Preserving-ML-through-HHE:
 * PASTA-3 -> BFV transciphering followed by the packed BFV linear layer.
 *
 * Plain C types only.  Every entry point returns 0 on success, non-zero on failure
 * (hhe_last_error() gives the message; INTEGRATION.md maps codes to the C++ exceptions
 * the reference throws).  "dptr" arguments are DEVICE pointers (HBM, uint64 words);
 * "hptr" arguments are host pointers.  Ciphertexts use SEAL's in-memory layout
 * [poly][limb][coeff] (seal/ciphertext.h:701-715) at the data level (L = K-1 limbs), in
 * coefficient (non-NTT) form, batches are contiguous: [B][2][L][N].
 * Key-switch keys use KSwitchKeys::data()[index][digit].data() layout, i.e.
 * [L digits][2][K][N] in NTT form (seal/kswitchkeys.h:90-130).
 *
 * Reference interface each entry replaces is cited as (file:line) relative to the
 * reference repository root.
 */
#ifndef HHE_GFX950_H
#define HHE_GFX950_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct hhe_ctx hhe_ctx;
/* One seal::RelinKeys / seal::GaloisKeys OBJECT on the device (keys with identity).  The reference's CSP holds several, made by
 * the same key generator with different randomness, and names one at every call: PASTA_SEAL(context, pk, sk, analyst rk, analyst gk)
 * (src/examples/CSP/CSP.cpp:238-242), flatten(record, tmp, csp_he_gk) (:271-278), relinearize_inplace(record, csp rk) (:306),
 * encrypted_vec_sum(.., analyst_he_gk, ..) (:312-316); the objects are created at src/examples/Analyst/Analyst.cpp:62-94.
 * Which keys a rotation finds decides its NAF decomposition (seal/evaluator.h:955-1060) and the ciphertext words. */
typedef struct hhe_keyset hhe_keyset;

enum {
    HHE_OK = 0,
    HHE_ERR_INVALID = 1,        /* std::invalid_argument in SEAL / the reference        */
    HHE_ERR_NO_GALOIS_KEY = 2,  /* SEAL: "Galois key not present" (evaluator.h:955-1060) */
    HHE_ERR_TOO_FEW_SLOTS = 3,  /* "too little slots for matmul implementation!" (src/pasta/pasta_3_seal.cpp:376-377) */
    HHE_ERR_DEVICE = 4,         /* HIP runtime failure                                    */
    HHE_ERR_NO_RELIN_KEY = 5,
    HHE_ERR_CAPACITY = 6
};

const char *hhe_last_error(void);
/* "hip-gfx950" for the product library */
const char *hhe_backend(void);

/* ---- context: replaces SEALZpCipher::create_context / sealhelper::get_seal_context
 *      (src/pasta/SEAL_Cipher.cpp:38-68, src/util/sealhelper.cpp:8-41) with an explicit
 *      prime list: q[K] = coeff_modulus (last = special key-switch prime), t = plain_modulus.
 *      Derives NTT tables, BatchEncoder map, BEHZ base exactly as SEAL 4.0.0 does. ---- */
int hhe_ctx_create(int logn, int K, const uint64_t *q_hptr, uint64_t t, int device, hhe_ctx **out);
/* the prime chain SEALZpCipher::create_context picks: CoeffModulus::BFVDefault(N) for N <= 32768 and the hard-coded
 * 29-prime chain for N = 65536 (src/pasta/SEAL_Cipher.cpp:47-65).  *count: in = capacity, out = number of primes. */
int hhe_bfv_default_coeff_modulus(size_t poly_modulus_degree, uint64_t *out_hptr, size_t *count);
void hhe_ctx_destroy(hhe_ctx *c);
/* run all work of this context on an existing HIP stream (hipStream_t); NULL = default stream */
int hhe_ctx_set_stream(hhe_ctx *c, void *hip_stream);
/* size per-batch workspaces for up to max_batch ciphertexts (allocated once, reused) */
int hhe_ctx_reserve(hhe_ctx *c, size_t max_batch);
int hhe_ctx_sync(hhe_ctx *c);
/* Thread safety: every entry point that takes a context serialises on a lock inside it (the reference's gRPC handlers call
 * the cipher concurrently, src/examples/CSP/CSPRPC.cpp:201-203); different contexts are independent.  hhe_ctx_destroy must
 * not race with other calls on the same context. */
/* Timing instrumentation of the dominant kernel (no reference counterpart; the reference brackets phases with std::chrono,
 * src/examples/hhe_pktnn_examples.cpp:73-76): while enabled, every launch of the fused key-switch row kernel is bracketed
 * by timed HIP events on the stream it is launched on.  hhe_ctx_profile_read waits for the work, returns the number of
 * launches, the sum of their durations and the ciphertexts they covered, and resets the counters.  Results of the ops are
 * unaffected.  kernel_name (optional, name_cap bytes) receives the kernel's name as rocprofv3 prints it. */
int hhe_ctx_profile(hhe_ctx *c, int enable);
int hhe_ctx_profile_read(hhe_ctx *c, char *kernel_name, size_t name_cap, uint64_t *launches, double *total_ms, uint64_t *items);
/* derived parameters, for cross-checking against SEAL's context: what in
 * {"root" i<K, "bsk" i<=L (B_0.., m_sk), "gamma", "galois_elt" i=step, "fc_fallbacks", "fc_csum_closes"} */
uint64_t hhe_ctx_query(const hhe_ctx *c, const char *what, int i);

/* ---- keys.  Key words must be reduced modulo their coefficient primes (what SEAL's safe load checks, is_data_valid_for):
 *      the kernels multiply them through Shoup quotients; an upload with a word >= its prime fails with HHE_ERR_INVALID.
 *      The DEFAULT set of a context = the by-value seal::RelinKeys / seal::GaloisKeys members of SEALZpCipher
 *      (src/pasta/SEAL_Cipher.h:28-31; ctor SEAL_Cipher.cpp:9-36): every entry point without a key-set argument uses it. ---- */
int hhe_set_relin_key(hhe_ctx *c, const uint64_t *ksk_hptr); /* slot 0: the key PASTA_SEAL was constructed with */
/* further RelinKeys objects (e.g. the CSP's own csp_rk used at CSP.cpp:306); slot < 4 */
int hhe_set_relin_key_slot(hhe_ctx *c, int slot, const uint64_t *ksk_hptr);
int hhe_set_galois_key(hhe_ctx *c, uint32_t galois_elt, const uint64_t *ksk_hptr);
int hhe_has_galois_key(const hhe_ctx *c, uint32_t galois_elt);
/* Key sets: one per RelinKeys / GaloisKeys object of the caller (a set may hold both kinds).  Uploaded once, resident in HBM
 * with everything derived from its keys (Shoup quotient tables, FC correction tables) until destroyed; using a set never
 * touches another set.  A set belongs to the context it was created on and must be destroyed before that context
 * (hhe_ctx_destroy releases forgotten ones).  In the *_ks entry points below a NULL set means the context's default set. */
int hhe_keyset_create(hhe_ctx *c, hhe_keyset **out);
void hhe_keyset_destroy(hhe_keyset *ks);
int hhe_keyset_set_relin(hhe_keyset *ks, const uint64_t *ksk_hptr);                        /* RelinKeys::key(2) */
int hhe_keyset_set_galois(hhe_keyset *ks, uint32_t galois_elt, const uint64_t *ksk_hptr);  /* GaloisKeys::key(galois_elt) */
int hhe_keyset_has_galois(const hhe_keyset *ks, uint32_t galois_elt);                      /* GaloisKeys::has_key */
int hhe_keyset_has_relin(const hhe_keyset *ks);

/* ---- device memory helpers for callers without their own HIP allocator ---- */
void *hhe_malloc(size_t bytes);
void hhe_free(void *dptr);
int hhe_copy_h2d(hhe_ctx *c, void *dptr, const void *hptr, size_t bytes);
int hhe_copy_d2h(hhe_ctx *c, void *hptr, const void *dptr, size_t bytes);

/* ---- BFV primitives on device batches (seal::Evaluator / BatchEncoder as used by the path) ---- */
/* util::ntt_negacyclic_harvey / inverse (seal/util/ntt.h:231,303): polys [count][N], modulus of
 * poly p = mod_base + p % mod_cycle (0..K-1 coeff primes, K..K+L Bsk, K+L+1 = t), in place */
int hhe_ntt(hhe_ctx *c, uint64_t *polys_dptr, size_t count, int mod_base, int mod_cycle, int inverse);
/* BatchEncoder::encode (seal/batchencoder.h:80): vals [B][count] -> plain [B][N] */
int hhe_encode(hhe_ctx *c, const uint64_t *vals_dptr, size_t B, size_t count, uint64_t *plain_dptr);
/* Evaluator::add_inplace / negate_inplace (seal/evaluator.h:92-132); size = polys per ct */
int hhe_add(hhe_ctx *c, const uint64_t *a_dptr, const uint64_t *b_dptr, uint64_t *out_dptr, size_t B, int size);
int hhe_negate(hhe_ctx *c, const uint64_t *a_dptr, uint64_t *out_dptr, size_t B, int size);
/* Evaluator::add_plain / sub_plain (seal/evaluator.h:665-680); plain [B][N] or [1][N] if bcast */
int hhe_add_plain(hhe_ctx *c, const uint64_t *ct_dptr, const uint64_t *plain_dptr, int plain_bcast,
                  int subtract, uint64_t *out_dptr, size_t B);
/* Evaluator::multiply_plain (seal/evaluator.h:729-747) */
int hhe_multiply_plain(hhe_ctx *c, const uint64_t *ct_dptr, const uint64_t *plain_dptr, int plain_bcast,
                       uint64_t *out_dptr, size_t B);
/* Evaluator::apply_galois (seal/evaluator.h:889) */
int hhe_apply_galois(hhe_ctx *c, const uint64_t *ct_dptr, uint32_t galois_elt, uint64_t *out_dptr, size_t B);
/* Evaluator::rotate_rows / rotate_columns incl. NAF fallback (seal/evaluator.h:955-1060) */
int hhe_rotate_rows(hhe_ctx *c, const uint64_t *ct_dptr, int step, uint64_t *out_dptr, size_t B);
int hhe_rotate_columns(hhe_ctx *c, const uint64_t *ct_dptr, uint64_t *out_dptr, size_t B);
/* the same with the GaloisKeys object the call names, as Evaluator::rotate_rows(ct, step, galois_keys, out) does: a step whose
 * element is not in THIS set is NAF-decomposed over the keys of this set, whatever other sets hold */
int hhe_apply_galois_ks(hhe_ctx *c, const hhe_keyset *gk, const uint64_t *ct_dptr, uint32_t galois_elt, uint64_t *out_dptr, size_t B);
int hhe_rotate_rows_ks(hhe_ctx *c, const hhe_keyset *gk, const uint64_t *ct_dptr, int step, uint64_t *out_dptr, size_t B);
int hhe_rotate_columns_ks(hhe_ctx *c, const hhe_keyset *gk, const uint64_t *ct_dptr, uint64_t *out_dptr, size_t B);
/* Evaluator::multiply (seal/evaluator.h:214-277; BEHZ) -> size-3 ct [B][3][L][N] */
int hhe_multiply(hhe_ctx *c, const uint64_t *a_dptr, const uint64_t *b_dptr, uint64_t *out3_dptr, size_t B);
/* Evaluator::relinearize_inplace (seal/evaluator.h:301-304) size 3 -> 2 */
int hhe_relinearize(hhe_ctx *c, const uint64_t *a3_dptr, uint64_t *out_dptr, size_t B);
/* the same with the RelinKeys object of `slot` (hhe_set_relin_key_slot): Evaluator::relinearize_inplace(record, csp_rk) at
 * src/examples/CSP/CSP.cpp:306 uses the CSP's keys, not the ones PASTA_SEAL was built with */
int hhe_relinearize_slot(hhe_ctx *c, int slot, const uint64_t *a3_dptr, uint64_t *out_dptr, size_t B);
/* ... or with the RelinKeys object as a key set */
int hhe_relinearize_ks(hhe_ctx *c, const hhe_keyset *rk, const uint64_t *a3_dptr, uint64_t *out_dptr, size_t B);

/* ---- the hot path ---- */
/* PASTA_SEAL::decomposition / HE_decrypt (src/pasta/pasta_3_seal.cpp:106-172 / :42-104), batched over
 * independent blocks: item i transciphers the <=128 symmetric-ciphertext words cw[i][0..ncw[i]) that
 * were PASTA-encrypted with block counter block_index[i] (nonce 123456789) under the key whose BFV
 * encryption is enc_key (ONE ciphertext [2][L][N], shared by the batch: enc_ssk[0]).
 * cw_hptr [B][128] uint64 host, ncw_hptr [B], block_index_hptr [B]; out [B][2][L][N] device.
 * Needs relin key + Galois keys for steps {-1, +128 (if N/2 != 128), columns} (add_gk_indices :190-201);
 * use_bsgs!=0 selects PASTA_SEAL::babystep_giantstep (:267-366) and additionally steps -16k, k=1..7. */
int hhe_pasta3_transcipher(hhe_ctx *c, const uint64_t *enc_key_dptr, const uint64_t *cw_hptr,
                           const uint32_t *ncw_hptr, const uint64_t *block_index_hptr, size_t B,
                           int use_bsgs, uint64_t *out_dptr);
/* the same for a PASTA_SEAL constructed with the RelinKeys `rk` and the GaloisKeys `gk` (CSP.cpp:238-242: the analyst's) */
int hhe_pasta3_transcipher_ks(hhe_ctx *c, const hhe_keyset *rk, const hhe_keyset *gk, const uint64_t *enc_key_dptr, const uint64_t *cw_hptr,
                              const uint32_t *ncw_hptr, const uint64_t *block_index_hptr, size_t B, int use_bsgs, uint64_t *out_dptr);
/* The per-block public tables (matrices depend only on (nonce, block index)) are built on first use and cached per block counter:
 * (4 x 128 x L) x N words of multipliers, twice that with their Shoup quotients in the fused pipeline -- 768 MiB per counter at
 * N = 2^15, L = 3; 1.07 / 2.1 GiB at the reference defaults N = 2^14, L = 8 -- and the babystep-giantstep variant adds (4 x 128 x L) x N.
 * The cache is bounded: beyond `bytes` (default 32 GiB; HHE_BLOCK_CACHE_MB) the least recently used counters are dropped, never
 * one the running call uses (a single call that needs more than the limit is served).  hhe_ctx_query("block_cache_bytes" /
 * "block_cache_entries") report its state; hhe_pasta3_clear_block_cache drops everything. */
int hhe_pasta3_set_block_cache_limit(hhe_ctx *c, size_t bytes);
void hhe_pasta3_clear_block_cache(hhe_ctx *c);
/* SEALZpCipher::mask (src/pasta/SEAL_Cipher.cpp:161-166): mask_vals_hptr[count], shared by the batch */
int hhe_mask(hhe_ctx *c, const uint64_t *ct_dptr, const uint64_t *mask_vals_hptr, size_t count,
             uint64_t *out_dptr, size_t B);
/* SEALZpCipher::flatten (src/pasta/SEAL_Cipher.cpp:170-181): blocks [S][nblocks][2][L][N] -> out [S][2][L][N] */
int hhe_flatten(hhe_ctx *c, const uint64_t *blocks_dptr, size_t nblocks, uint64_t *out_dptr, size_t S);
/* flatten(in, out, galois_keys) with the GaloisKeys object it is given (CSP.cpp:271-278 passes csp_he_gk, steps -128*i) */
int hhe_flatten_ks(hhe_ctx *c, const hhe_keyset *gk, const uint64_t *blocks_dptr, size_t nblocks, uint64_t *out_dptr, size_t S);
/* BaseCSP::decompose (src/examples/CSP/CSP.cpp:235-283): S records of nwords symmetric-ciphertext words (host, [S][nwords])
 * -> decomposition of every block + mask of the ragged last block (mask_last != 0: as hhe_pktnn_examples.cpp:620-626;
 * the CSP's own loop at CSP.cpp:264-269 masks a copy, i.e. has no effect: pass 0 to reproduce that) + flatten.
 * out [S][2][L][N] device.  Needs the PASTA keys plus Galois keys reaching steps -128*i (directly or through NAF). */
int hhe_decompose(hhe_ctx *c, const uint64_t *enc_key_dptr, const uint64_t *records_hptr, size_t S, size_t nwords,
                  int mask_last, uint64_t *out_dptr);
/* the same with the three key objects BaseCSP::decompose names: `rk` / `gk` construct the PASTA_SEAL (CSP.cpp:238-242), `flatten_gk`
 * is the GaloisKeys passed to flatten (CSP.cpp:271-278) */
int hhe_decompose_ks(hhe_ctx *c, const hhe_keyset *rk, const hhe_keyset *gk, const hhe_keyset *flatten_gk, const uint64_t *enc_key_dptr,
                     const uint64_t *records_hptr, size_t S, size_t nwords, int mask_last, uint64_t *out_dptr);
/* FC row: sealhelper::packed_enc_multiply + Evaluator::relinearize_inplace + sealhelper::encrypted_vec_sum
 * (src/util/sealhelper.cpp:268-274, src/examples/CSP/CSP.cpp:306, sealhelper.cpp:379-392).
 * vi [B][2][L][N]; w: weight-row ciphertexts [W][2][L][N]; item i uses w[i % W]. out [B][2][L][N];
 * the neuron's value is slot n_inputs-1 of the decryption.  relin_slot selects the RelinKeys object;
 * default_galois_only != 0 makes rotate_rows see only the power-of-two Galois elements, i.e. behave as if called
 * with a GaloisKeys made by create_galois_keys() without arguments (what the CSP passes, CSP.cpp:312-316), even when
 * the context also holds flatten / PASTA keys.  The n-1 NAF rotation chains share prefixes and are evaluated as a
 * trie (identical ciphertext words, ~2.7x fewer key switches for n = 784); the children of a trie node share the digit
 * transforms of its c1 (exact unless a c1 coefficient is 0, in which case the chunk is recomputed with per-child transforms;
 * hhe_ctx_query("fc_fallbacks") counts those).  Synchronous: returns when the results are in out. */
int hhe_fc_row(hhe_ctx *c, const uint64_t *vi_dptr, const uint64_t *w_dptr, size_t W, size_t n_inputs, int relin_slot,
               int default_galois_only, uint64_t *out_dptr, size_t B);
/* the same with the key objects the CSP names: relinearize_inplace(prod, rk) (CSP.cpp:306) and encrypted_vec_sum(prod, sum, evaluator,
 * gk, n) (CSP.cpp:312-316): a rotation step is one key switch exactly when `gk` holds its element, else its NAF terms over `gk` */
int hhe_fc_row_ks(hhe_ctx *c, const hhe_keyset *rk, const hhe_keyset *gk, const uint64_t *vi_dptr, const uint64_t *w_dptr, size_t W,
                  size_t n_inputs, uint64_t *out_dptr, size_t B);

/* PASTA-3 public randomness for one block as the kernels consume it (host; src/pasta/pasta_3_plain.cpp:56-119,286-295):
 * mats [4][2][128][128], rcs [4][2][128] */
int hhe_pasta3_block_randomness(uint64_t t, uint64_t block_index, uint64_t *mats_hptr, uint64_t *rcs_hptr);

/* ---- client / analyst ends ---- */
/* pasta::Pasta::keystream (src/pasta/pasta_3_plain.cpp:156-178) for block counters first_block .. first_block+nblocks-1
 * (nonce 123456789) under the 256-word secret key key_hptr (words < t): ks_dptr [nblocks][128] device. */
int hhe_pasta3_plain_keystream(hhe_ctx *c, const uint64_t *key_hptr, uint64_t first_block, size_t nblocks, uint64_t *ks_dptr);
/* pasta::PASTA::encrypt / decrypt (src/pasta/pasta_3_plain.cpp:9-46) over S records of nwords words each (device,
 * [S][nwords]); every record starts at block counter 0, as each PASTA::encrypt call does.  in == out is allowed. */
int hhe_pasta3_plain_crypt(hhe_ctx *c, const uint64_t *key_hptr, const uint64_t *in_dptr, size_t S, size_t nwords,
                           int decrypt, uint64_t *out_dptr);
/* Decryptor::decrypt + BatchEncoder::decode (seal/decryptor.h:70, seal/batchencoder.h:282) as used by
 * sealhelper::decrypting / Analyst::decrypt_result (src/util/sealhelper.cpp:252-266): B size-2 data-level
 * ciphertexts [B][2][L][N] -> slot values [B][N] (unsigned, < t; the signed view of decode_int64 is v > t/2 ? v - t : v).
 * sk_hptr: the secret key polynomial at the key level, NTT form [K][N] (SecretKey::data().data()). */
int hhe_decrypt(hhe_ctx *c, const uint64_t *sk_hptr, const uint64_t *ct_dptr, size_t B, uint64_t *vals_dptr);

/* ---- SEAL 4.0 binary serialization at the boundary: the blobs the reference moves over gRPC and spills to
 *      disk (src/examples/CSP/CSP.cpp:328-490 `*.load(*context, bytes, size)`, :495-547 spill file = size_t count followed by
 *      Ciphertext::save streams, :552-605 concatenated stream, protos/hhe.proto:21-24) decoded straight into HBM.
 *      Layout sources: SEALHeader (seal/serialization.h:60-93), DynArray::save_members (seal/dynarray.h:652-680),
 *      KSwitchKeys members (seal/kswitchkeys.h:161-178), PublicKey = Ciphertext (seal/publickey.h:89-93); Ciphertext's member
 *      order is SEAL 4.0.0's as published.  PARITY UNPINNED: the reference holds no serialized SEAL object and libseal is never
 *      run here, so these are checked by round trips and size arithmetic only.  compr_mode none / zlib / zstd are accepted (the
 *      latter two through the system's libz.so.1 / libzstd.so.1, looked up at run time); seeded (symmetric-key) objects are
 *      rejected.  `consumed` returns the size field of the object's header, i.e. where the next object of a stream starts. ---- */
/* Ciphertext::load: data-level, coefficient-form BFV ciphertext -> out_dptr [size][L][N]; *ct_size = 2 or 3;
 * parms_id_out (32 bytes, optional) receives the object's parms_id for later saves */
int hhe_seal_load_ciphertext(hhe_ctx *c, const uint8_t *bytes_hptr, size_t nbytes, uint64_t *out_dptr, size_t out_cap_words,
                             size_t *ct_size, uint8_t *parms_id_out, size_t *consumed);
/* Ciphertext::save(stream, compr_mode_type::none) of a device ciphertext; *written = bytes needed (returned also when out_cap is
 * too small).  parms_id: 32 bytes taken from a loaded object of the same context (SEAL hashes the parameters into it) */
int hhe_seal_save_ciphertext(hhe_ctx *c, const uint64_t *ct_dptr, size_t ct_size, const uint8_t *parms_id, uint8_t *out_hptr,
                             size_t out_cap, size_t *written);
/* RelinKeys::load / GaloisKeys::load: every key of the object is uploaded (relin key(2) into `slot`; Galois keys by element
 * 2*index+1, seal/galoiskeys.h:48-74) */
int hhe_seal_load_relin_keys(hhe_ctx *c, int slot, const uint8_t *bytes_hptr, size_t nbytes, size_t *consumed);
int hhe_seal_load_galois_keys(hhe_ctx *c, const uint8_t *bytes_hptr, size_t nbytes, size_t *consumed, uint32_t *n_keys);
/* the same into a key set (one set per loaded object).  All loads are all-or-nothing, as SEAL's load(context, ...) is: the whole
 * object is decoded and validated on the host first; a malformed, truncated or over-long object leaves the target untouched.
 * Inflated sizes are bounded by the largest legitimate object of the context (a few KB of zlib / zstd cannot expand into GBs). */
int hhe_seal_load_relin_keys_ks(hhe_keyset *ks, const uint8_t *bytes_hptr, size_t nbytes, size_t *consumed);
int hhe_seal_load_galois_keys_ks(hhe_keyset *ks, const uint8_t *bytes_hptr, size_t nbytes, size_t *consumed, uint32_t *n_keys);

#ifdef __cplusplus
}
#endif
#endif
