// pasta_seal_gfx950.hpp -- C++ host-side mirror of the reference's cipher-layer interface for the
// CSP hot path, implemented over the C ABI of libhhe_gfx950.so (include/hhe_gfx950.h).
//
// Mirrors (same names, argument meaning and error behaviour):
//   pasta::SEALZpCipher   src/pasta/SEAL_Cipher.h:11-129   (get_plain_size, mask, flatten, activate_bsgs, ...)
//   pasta::PASTA_SEAL     src/pasta/pasta_3_seal.h:8-54    (HE_decrypt, decomposition, add_gk_indices, ...)
//   sealhelper::packed_enc_multiply / encrypted_vec_sum   src/util/sealhelper.h:84-129
//   pasta::PASTA          src/pasta/pasta_3_plain.h:17-30  (client side: encrypt / decrypt)
//   sealhelper::decrypting                                 src/util/sealhelper.cpp:252-266 (analyst side)
// The reference passes seal:: objects; SEAL is not linked here, so the boundary types below are plain
// word containers with SEAL's in-memory layouts (what Ciphertext::data(), KSwitchKeys::data() hold).
// INTEGRATION.md shows the 1:1 conversion a SEAL-linking caller adds.  Errors surface as the C++
// exceptions the reference / SEAL throw (std::runtime_error, std::invalid_argument, std::logic_error).
#pragma once
#include <cmath>
#include <cstdint>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>
#include "hhe_gfx950.h"
#include "hhe_keyset_cache.hpp"

namespace pasta {

struct ZpCipherParams {  // src/pasta/Cipher.h:14-18
    size_t key_size, plain_size, cipher_size;
};
constexpr ZpCipherParams PASTA_PARAMS = {256, 128, 128};  // src/pasta/pasta_3_plain.h:15

// seal::Ciphertext stand-in: data() words, [size][L][N], data level, non-NTT form
struct Ciphertext {
    std::vector<uint64_t> words;
    size_t size = 0;
};
// one KSwitchKeys::data()[index] entry: [L digits][2][K][N], NTT form
typedef std::vector<uint64_t> KSwitchKey;
struct RelinKeys { KSwitchKey key; };                       // RelinKeys::key(2)
struct GaloisKeys { std::map<uint32_t, KSwitchKey> keys; }; // by Galois element (GaloisKeys::get_index = (elt-1)/2)
struct PublicKey { std::vector<uint64_t> words; };          // unused on the CSP path (kept for signature parity)
struct SecretKey { std::vector<uint64_t> words; };

// seal::SEALContext stand-in: (N, coeff_modulus incl. the special prime, plain_modulus)
class HheContext {
public:
    HheContext(int logn, std::vector<uint64_t> coeff_modulus, uint64_t plain_modulus, int device = 0)
        : logn_(logn), q_(std::move(coeff_modulus)), t_(plain_modulus)
    {
        if (hhe_ctx_create(logn, (int)q_.size(), q_.data(), t_, device, &h_) != HHE_OK)
            throw std::invalid_argument(std::string("encryption parameters are not set correctly: ") + hhe_last_error());
    }
    ~HheContext() { keys_.reset(); hhe_ctx_destroy(h_); }   // key sets go before their context
    HheContext(const HheContext &) = delete;
    HheContext &operator=(const HheContext &) = delete;
    hhe_ctx *handle() const { return h_; }
    // every RelinKeys / GaloisKeys object handed to a cipher object or to a call maps to ONE device key set, found again by its
    // contents (the reference copies the objects by value everywhere); persistent device buffers for the per-call operands
    hhe::KeySetCache &keys() { if (!keys_) keys_.reset(new hhe::KeySetCache(h_)); return *keys_; }
    hhe::DeviceArena &arena() { return arena_; }
    // enc_ssk[0] arrives by value with every call (CSP.cpp:249): it crosses PCIe only when its contents change.  Caller holds the
    // arena's lock; the resident copy lives in arena slot 2.
    uint64_t *encrypted_key(const uint64_t *words, size_t count)
    {
        hhe::ContentHash hsh;
        hsh.add(words, count);
        uint64_t *d = arena_.get(2, count * 8);
        if (!key_resident_ || key_hash_ < hsh || hsh < key_hash_) {
            if (hhe_copy_h2d(h_, d, words, count * 8) != HHE_OK) throw std::runtime_error(hhe_last_error());
            key_hash_ = hsh;
            key_resident_ = true;
            ++key_uploads;
        }
        return d;
    }
    uint64_t key_uploads = 0;   // instrumentation: how often an encrypted PASTA key was sent to the device
    size_t poly_modulus_degree() const { return (size_t)1 << logn_; }
    size_t data_limbs() const { return q_.size() - 1; }
    size_t ct_words() const { return 2 * data_limbs() * poly_modulus_degree(); }
    uint64_t plain_modulus() const { return t_; }

private:
    int logn_;
    std::vector<uint64_t> q_;
    uint64_t t_;
    hhe_ctx *h_ = nullptr;
    std::unique_ptr<hhe::KeySetCache> keys_;
    hhe::DeviceArena arena_;
    hhe::ContentHash key_hash_;
    bool key_resident_ = false;
};

namespace detail {
inline void check(int rc)
{
    if (rc == HHE_OK) return;
    const std::string msg = hhe_last_error();
    switch (rc) {
    case HHE_ERR_TOO_FEW_SLOTS: throw std::runtime_error(msg);   // pasta_3_seal.cpp:376-377
    case HHE_ERR_NO_GALOIS_KEY:
    case HHE_ERR_INVALID: throw std::invalid_argument(msg);      // SEAL: invalid_argument
    case HHE_ERR_NO_RELIN_KEY: throw std::invalid_argument(msg);
    default: throw std::runtime_error(msg);
    }
}
inline hhe_keyset *galois_set(HheContext &ctx, const GaloisKeys &gk)
{
    if (gk.keys.empty()) return nullptr;   // an empty object: the calls below then report "Galois key not present", as SEAL does
    std::vector<std::pair<uint32_t, const uint64_t *>> v;
    for (auto &kv : gk.keys) v.emplace_back(kv.first, kv.second.data());
    return ctx.keys().galois(v, gk.keys.begin()->second.size());
}
inline hhe_keyset *relin_set(HheContext &ctx, const RelinKeys &rk) { return rk.key.empty() ? nullptr : ctx.keys().relin(rk.key.data(), rk.key.size()); }
struct DevBuf {  // RAII device buffer
    void *p = nullptr;
    explicit DevBuf(size_t bytes) : p(hhe_malloc(bytes)) { if (!p) throw std::runtime_error("hhe_malloc failed"); }
    ~DevBuf() { hhe_free(p); }
    DevBuf(const DevBuf &) = delete;
    uint64_t *u64() const { return (uint64_t *)p; }
};
}  // namespace detail

class SEALZpCipher {
public:
    typedef std::vector<uint64_t> vector;

    SEALZpCipher(ZpCipherParams params, std::shared_ptr<HheContext> con, PublicKey pk, SecretKey sk, RelinKeys rk, GaloisKeys gk)
        : params(params), context(std::move(con)), he_pk(std::move(pk)), he_sk(std::move(sk))
    {
        // the by-value key members of the reference (SEAL_Cipher.h:28-31) become two device key sets, shared with every other cipher
        // object that was built from the same key objects (BaseCSP::decompose builds one per request, CSP.cpp:238-242)
        rk_set = detail::relin_set(*context, rk);
        gk_set = detail::galois_set(*context, gk);
        empty_set = rk_set && gk_set ? nullptr : make_empty_set();
        mod_degree = context->poly_modulus_degree();
        plain_mod = context->plain_modulus();
    }
    virtual ~SEALZpCipher() = default;

    size_t get_key_size() const { return params.key_size; }
    size_t get_plain_size() const { return params.plain_size; }
    size_t get_cipher_size() const { return params.cipher_size; }
    virtual std::string get_cipher_name() const = 0;
    virtual std::vector<Ciphertext> HE_decrypt(std::vector<uint64_t> &ciphertext, bool batch_encoder = false) = 0;
    virtual void add_gk_indices() = 0;

    // SEALZpCipher::create_context (src/pasta/SEAL_Cipher.cpp:38-68)
    static std::shared_ptr<HheContext> create_context(size_t mod_degree, uint64_t plain_mod, int seclevel = 128, int device = 0)
    {
        if (seclevel != 128) throw std::runtime_error("Security Level not supported");
        uint64_t q[64];
        size_t cnt = 64;
        if (hhe_bfv_default_coeff_modulus(mod_degree, q, &cnt) != HHE_OK) throw std::invalid_argument(hhe_last_error());
        int logn = 0;
        while (((size_t)1 << logn) < mod_degree) logn++;
        return std::make_shared<HheContext>(logn, std::vector<uint64_t>(q, q + cnt), plain_mod, device);
    }

    void activate_bsgs(bool activate) { use_bsgs = activate; }
    void set_bsgs_params(uint64_t n1, uint64_t n2) { bsgs_n1 = n1; bsgs_n2 = n2; }
    void add_some_gk_indices(std::vector<int> &gk_ind) { for (int i : gk_ind) gk_indices.push_back(i); }
    const std::vector<int> &get_gk_indices() const { return gk_indices; }

    // SEALZpCipher::mask (SEAL_Cipher.cpp:161-166)
    void mask(Ciphertext &cipher, std::vector<uint64_t> &mask_vec)
    {
        detail::DevBuf d(cipher.words.size() * 8);
        hhe_ctx *h = context->handle();
        detail::check(hhe_copy_h2d(h, d.p, cipher.words.data(), cipher.words.size() * 8));
        detail::check(hhe_mask(h, d.u64(), mask_vec.data(), mask_vec.size(), d.u64(), 1));
        detail::check(hhe_copy_d2h(h, cipher.words.data(), d.p, cipher.words.size() * 8));
    }
    // SEALZpCipher::flatten(in, out, galois_keys) (SEAL_Cipher.cpp:170-181): the rotations use the GaloisKeys object the CALL names
    // (CSP.cpp:271-278 passes csp_he_gk, not the keys the cipher object was built with)
    void flatten(std::vector<Ciphertext> &in, Ciphertext &out, const GaloisKeys &galois_keys)
    {
        flatten_with(in, out, or_empty(detail::galois_set(*context, galois_keys)));
    }
    // convenience: with the Galois keys held by this object
    void flatten(std::vector<Ciphertext> &in, Ciphertext &out) { flatten_with(in, out, or_empty(gk_set)); }

    // SEALZpCipher::packed_enc_mul / packed_enc_add / packed_square (SEAL_Cipher.cpp:547-566)
    void packed_enc_mul(const Ciphertext &e1, const Ciphertext &e2, Ciphertext &destination)
    {
        const size_t w = context->ct_words();
        detail::DevBuf a(w * 8), b(w * 8), o(w / 2 * 3 * 8);
        hhe_ctx *h = context->handle();
        detail::check(hhe_copy_h2d(h, a.p, e1.words.data(), w * 8));
        detail::check(hhe_copy_h2d(h, b.p, e2.words.data(), w * 8));
        detail::check(hhe_multiply(h, a.u64(), b.u64(), o.u64(), 1));   // Evaluator::multiply -> size 3
        destination.words.resize(w / 2 * 3);
        destination.size = 3;
        detail::check(hhe_copy_d2h(h, destination.words.data(), o.p, w / 2 * 3 * 8));
    }
    void packed_enc_add(const Ciphertext &e1, const Ciphertext &e2, Ciphertext &destination)
    {
        if (e1.size != e2.size || e1.words.size() != e2.words.size()) throw std::invalid_argument("encrypted1 and encrypted2 parameter mismatch");
        detail::DevBuf a(e1.words.size() * 8), b(e1.words.size() * 8);
        hhe_ctx *h = context->handle();
        detail::check(hhe_copy_h2d(h, a.p, e1.words.data(), e1.words.size() * 8));
        detail::check(hhe_copy_h2d(h, b.p, e2.words.data(), e2.words.size() * 8));
        detail::check(hhe_add(h, a.u64(), b.u64(), a.u64(), 1, (int)e1.size));
        destination = e1;
        detail::check(hhe_copy_d2h(h, destination.words.data(), a.p, e1.words.size() * 8));
    }
    void packed_square(Ciphertext &vo, const Ciphertext &vi)  // evaluator.square + relinearize_inplace(he_rk)
    {
        const size_t w = context->ct_words();
        detail::DevBuf a(w * 8), o3(w / 2 * 3 * 8);
        hhe_ctx *h = context->handle();
        detail::check(hhe_copy_h2d(h, a.p, vi.words.data(), w * 8));
        detail::check(hhe_multiply(h, a.u64(), a.u64(), o3.u64(), 1));
        detail::check(hhe_relinearize_ks(h, or_empty(rk_set), o3.u64(), a.u64(), 1));
        vo.words.resize(w);
        vo.size = 2;
        detail::check(hhe_copy_d2h(h, vo.words.data(), a.p, w * 8));
    }

protected:
    void flatten_with(std::vector<Ciphertext> &in, Ciphertext &out, const hhe_keyset *gk)
    {
        if (in.empty()) throw std::invalid_argument("flatten: empty input");
        const size_t w = context->ct_words();
        hhe_ctx *h = context->handle();
        std::lock_guard<std::mutex> lk(context->arena().mutex());
        uint64_t *d = context->arena().get(0, in.size() * w * 8), *o = context->arena().get(1, w * 8);
        for (size_t i = 0; i < in.size(); i++) detail::check(hhe_copy_h2d(h, d + i * w, in[i].words.data(), w * 8));
        detail::check(hhe_flatten_ks(h, gk, d, in.size(), o, 1));
        out.words.resize(w);
        out.size = 2;
        detail::check(hhe_copy_d2h(h, out.words.data(), o, w * 8));
    }
    // a cipher object built WITHOUT some key must not fall through to the context's default set: an empty set of its own
    hhe_keyset *make_empty_set()
    {
        hhe_keyset *ks = nullptr;
        detail::check(hhe_keyset_create(context->handle(), &ks));
        owned_empty.reset(ks, [](hhe_keyset *k) { hhe_keyset_destroy(k); });
        return ks;
    }
    const hhe_keyset *or_empty(const hhe_keyset *ks) { return ks ? ks : (empty_set ? empty_set : (empty_set = make_empty_set())); }
    ZpCipherParams params;
    uint64_t plain_mod = 0, mod_degree = 0;
    std::vector<Ciphertext> secret_key_encrypted;
    std::shared_ptr<HheContext> context;
    PublicKey he_pk;
    SecretKey he_sk;
    std::vector<int> gk_indices;
    bool use_bsgs = false;
    size_t bsgs_n1 = 0, bsgs_n2 = 0;
    // device key sets of this object's key members (owned by the context's cache); declared after `context` so that the empty set
    // this object may own is released while its context still exists
    hhe_keyset *rk_set = nullptr, *gk_set = nullptr, *empty_set = nullptr;
    std::shared_ptr<hhe_keyset> owned_empty;
};

class PASTA_SEAL : public SEALZpCipher {
public:
    PASTA_SEAL(std::shared_ptr<HheContext> con, PublicKey pk, SecretKey sk, RelinKeys rk, GaloisKeys gk)
        : SEALZpCipher(PASTA_PARAMS, std::move(con), std::move(pk), std::move(sk), std::move(rk), std::move(gk)),
          slots(mod_degree), halfslots(mod_degree >> 1) {}

    virtual std::string get_cipher_name() const { return "PASTA-SEAL (n=128,r=3)"; }

    // pasta_3_seal.cpp:190-201
    virtual void add_gk_indices()
    {
        gk_indices.push_back(0);
        gk_indices.push_back(-1);
        if (PASTA_PARAMS.plain_size * 2 != slots) gk_indices.push_back((int)PASTA_PARAMS.plain_size);
        if (use_bsgs)
            for (uint64_t k = 1; k < BSGS_N2; k++) gk_indices.push_back(-(int)(k * BSGS_N1));
    }

    // supply the BFV encryption of the PASTA key that HE_decrypt reads (secret_key_encrypted[0],
    // pasta_3_seal.cpp:58; filled by encrypt_key() on the client side of the reference)
    void set_encrypted_key(const Ciphertext &enc_key) { secret_key_encrypted.assign(1, enc_key); }

    // PASTA_SEAL::HE_decrypt (pasta_3_seal.cpp:42-104) == decomposition(ciphertexts, secret_key_encrypted)
    virtual std::vector<Ciphertext> HE_decrypt(std::vector<uint64_t> &ciphertexts, bool batch_encoder = false)
    {
        if (secret_key_encrypted.empty()) throw std::logic_error("HE_decrypt: encrypted key not set");
        return decomposition(ciphertexts, secret_key_encrypted, batch_encoder);
    }

    // PASTA_SEAL::decomposition (pasta_3_seal.cpp:106-172)
    virtual std::vector<Ciphertext> decomposition(std::vector<uint64_t> &ciphertexts, std::vector<Ciphertext> enc_ssk,
                                                  bool batch_encoder = false)
    {
        (void)batch_encoder;  // ignored by the reference as well (:113)
        if (enc_ssk.empty()) throw std::invalid_argument("decomposition: enc_ssk is empty");
        const size_t size = ciphertexts.size();
        const size_t num_block = (size_t)std::ceil((double)size / (double)params.cipher_size);
        std::vector<Ciphertext> res(num_block);
        if (num_block == 0) return res;
        const size_t w = context->ct_words();
        hhe_ctx *h = context->handle();
        std::vector<uint64_t> cw(num_block * 128, 0), bidx(num_block);
        std::vector<uint32_t> ncw(num_block);
        for (size_t b = 0; b < num_block; b++) {
            const size_t lo = b * params.cipher_size, hi = std::min(lo + params.cipher_size, size);
            for (size_t i = lo; i < hi; i++) cw[b * 128 + (i - lo)] = ciphertexts[i];
            ncw[b] = (uint32_t)(hi - lo);
            bidx[b] = b;
        }
        if (enc_ssk[0].words.size() != w) throw std::invalid_argument("decomposition: enc_ssk is not valid for encryption parameters");
        std::lock_guard<std::mutex> lk(context->arena().mutex());
        uint64_t *key = context->encrypted_key(enc_ssk[0].words.data(), w), *out = context->arena().get(3, num_block * w * 8);
        detail::check(hhe_pasta3_transcipher_ks(h, or_empty(rk_set), or_empty(gk_set), key, cw.data(), ncw.data(), bidx.data(), num_block, use_bsgs ? 1 : 0, out));
        for (size_t b = 0; b < num_block; b++) {
            res[b].words.resize(w);
            res[b].size = 2;
            detail::check(hhe_copy_d2h(h, res[b].words.data(), out + b * w, w * 8));
        }
        return res;
    }

    // BaseCSP::decompose's per-record loop (CSP.cpp:247-278) as ONE device call: decomposition of every record, the mask of the
    // ragged last block (mask_last: as hhe_pktnn_examples.cpp:620-626; the CSP's own loop masks a copy, i.e. pass false to reproduce
    // that) and flatten with the GaloisKeys object `flatten_gk` -- the blocks never leave HBM.  One flattened ciphertext per record.
    std::vector<Ciphertext> decompose(const std::vector<std::vector<uint64_t>> &records, std::vector<Ciphertext> enc_ssk,
                                      const GaloisKeys &flatten_gk, bool mask_last)
    {
        std::vector<Ciphertext> res(records.size());
        if (records.empty()) return res;
        if (enc_ssk.empty()) throw std::invalid_argument("decompose: enc_ssk is empty");
        const size_t nwords = records[0].size(), w = context->ct_words();
        std::vector<uint64_t> flat(records.size() * nwords);
        for (size_t s = 0; s < records.size(); s++) {
            if (records[s].size() != nwords) throw std::invalid_argument("decompose: records of different lengths");
            std::copy(records[s].begin(), records[s].end(), flat.begin() + s * nwords);
        }
        hhe_ctx *h = context->handle();
        std::lock_guard<std::mutex> lk(context->arena().mutex());
        if (enc_ssk[0].words.size() != w) throw std::invalid_argument("decompose: enc_ssk is not valid for encryption parameters");
        uint64_t *key = context->encrypted_key(enc_ssk[0].words.data(), w), *out = context->arena().get(3, records.size() * w * 8);
        detail::check(hhe_decompose_ks(h, or_empty(rk_set), or_empty(gk_set), or_empty(detail::galois_set(*context, flatten_gk)), key, flat.data(),
                                       records.size(), nwords, mask_last ? 1 : 0, out));
        for (size_t s = 0; s < records.size(); s++) {
            res[s].words.resize(w);
            res[s].size = 2;
            detail::check(hhe_copy_d2h(h, res[s].words.data(), out + s * w, w * 8));
        }
        return res;
    }

private:
    static constexpr uint64_t BSGS_N1 = 16, BSGS_N2 = 8;  // pasta_3_seal.h:35-36
    size_t slots, halfslots;
};

// pasta::PASTA (src/pasta/pasta_3_plain.h:17-30; base ZpCipher src/pasta/Cipher.h:20-58): the client's symmetric
// cipher, evaluated by the device kernels.  Differs from the reference ctor only by the context handle in front.
class PASTA {
public:
    PASTA(std::shared_ptr<HheContext> con, std::vector<uint64_t> secret_key, uint64_t modulus)
        : context(std::move(con)), secret_key(std::move(secret_key)), modulus(modulus), params(PASTA_PARAMS)
    {
        if (this->secret_key.size() != params.key_size) throw std::runtime_error("Invalid Key length");  // Cipher.h:30-31
        if (modulus != context->plain_modulus()) throw std::invalid_argument("PASTA modulus differs from the plain modulus of the context");
    }
    virtual ~PASTA() = default;
    virtual std::string get_cipher_name() const { return "PASTA (n=128,r=3)"; }
    size_t get_key_size() const { return params.key_size; }
    size_t get_plain_size() const { return params.plain_size; }
    size_t get_cipher_size() const { return params.cipher_size; }
    virtual std::vector<uint64_t> encrypt(std::vector<uint64_t> plaintext) const { return crypt(std::move(plaintext), 0); }
    virtual std::vector<uint64_t> decrypt(std::vector<uint64_t> ciphertext) const { return crypt(std::move(ciphertext), 1); }

private:
    std::vector<uint64_t> crypt(std::vector<uint64_t> v, int dec) const
    {
        if (v.empty()) return v;
        hhe_ctx *h = context->handle();
        detail::DevBuf d(v.size() * 8);
        detail::check(hhe_copy_h2d(h, d.p, v.data(), v.size() * 8));
        detail::check(hhe_pasta3_plain_crypt(h, secret_key.data(), d.u64(), 1, v.size(), dec, d.u64()));
        detail::check(hhe_copy_d2h(h, v.data(), d.p, v.size() * 8));
        return v;
    }
    std::shared_ptr<HheContext> context;
    std::vector<uint64_t> secret_key;
    uint64_t modulus;
    ZpCipherParams params;
};

}  // namespace pasta

namespace sealhelper {
// sealhelper::decrypting (src/util/sealhelper.cpp:252-266): Decryptor::decrypt + BatchEncoder::decode into signed values
// (SEAL: slot value v > (t+1)/2 reads as v - t), first `size` slots.  he_sk.words: SecretKey::data() [K][N], NTT form.
inline std::vector<int64_t> decrypting(const pasta::Ciphertext &enc_input, const pasta::SecretKey &he_sk, pasta::HheContext &ctx,
                                       size_t size)
{
    const size_t w = ctx.ct_words(), n = ctx.poly_modulus_degree();
    if (enc_input.words.size() != w || he_sk.words.size() < ctx.data_limbs() * n || size > n)
        throw std::invalid_argument("decrypting: ciphertext / secret key do not match the context");
    pasta::detail::DevBuf c(w * 8), v(n * 8);
    hhe_ctx *h = ctx.handle();
    pasta::detail::check(hhe_copy_h2d(h, c.p, enc_input.words.data(), w * 8));
    pasta::detail::check(hhe_decrypt(h, he_sk.words.data(), c.u64(), 1, v.u64()));
    std::vector<uint64_t> u(n);
    pasta::detail::check(hhe_copy_d2h(h, u.data(), v.p, n * 8));
    const uint64_t t = ctx.plain_modulus(), half = (t + 1) >> 1;
    std::vector<int64_t> out(size);
    for (size_t i = 0; i < size; i++) out[i] = u[i] > half ? (int64_t)u[i] - (int64_t)t : (int64_t)u[i];
    return out;
}

// packed_enc_multiply + relinearize_inplace(.., csp_rk) + encrypted_vec_sum(.., gal_keys, n) for one weight row, with the key objects
// the CSP names at those calls (sealhelper.cpp:268-274, 379-392; CSP.cpp:306, 312-316)
inline void fc_row(pasta::HheContext &ctx, const pasta::Ciphertext &vi, const pasta::Ciphertext &w_row, const pasta::RelinKeys &csp_rk,
                   const pasta::GaloisKeys &gal_keys, size_t vec_size, pasta::Ciphertext &destination)
{
    const size_t w = ctx.ct_words();
    hhe_keyset *rk = pasta::detail::relin_set(ctx, csp_rk), *gk = pasta::detail::galois_set(ctx, gal_keys);
    if (!rk || !gk) throw std::invalid_argument("fc_row: empty key object");
    hhe_ctx *h = ctx.handle();
    std::lock_guard<std::mutex> lk(ctx.arena().mutex());
    uint64_t *a = ctx.arena().get(0, w * 8), *b = ctx.arena().get(1, w * 8), *o = ctx.arena().get(3, w * 8);
    pasta::detail::check(hhe_copy_h2d(h, a, vi.words.data(), w * 8));
    pasta::detail::check(hhe_copy_h2d(h, b, w_row.words.data(), w * 8));
    pasta::detail::check(hhe_fc_row_ks(h, rk, gk, a, b, 1, vec_size, o, 1));
    destination.words.resize(w);
    destination.size = 2;
    pasta::detail::check(hhe_copy_d2h(h, destination.words.data(), o, w * 8));
}
}  // namespace sealhelper
