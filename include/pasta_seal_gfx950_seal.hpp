// pasta_seal_gfx950_seal.hpp -- the SEAL-typed drop-in: pasta::SEALZpCipher / pasta::PASTA_SEAL with the reference's exact
// signatures (src/pasta/SEAL_Cipher.h:11-129, src/pasta/pasta_3_seal.h:8-54) and sealhelper::packed_enc_multiply /
// encrypted_vec_sum (src/util/sealhelper.h:84-129), implemented over the C ABI of libhhe_gfx950.so.
//
// Use in the reference tree: include this header where src/Common.h:19-21 includes "pasta_3_seal.h" / "SEAL_Cipher.h"
// (it replaces both; it still includes the reference's Cipher.h and pasta_3_plain.h for ZpCipherParams / PASTA_PARAMS and
// SEAL's own headers for the boundary types) and link libhhe_gfx950.so.  src/examples/CSP/CSP.cpp:235-323 then compiles
// unchanged: decomposition / mask / flatten / packed_enc_multiply / encrypted_vec_sum run on the MI355X, everything else
// (key generation, encrypt_key, decrypt_result, serialization) stays on SEAL.
// SEAL objects cross the boundary as their own words: Ciphertext::data() is [size][L][N] at the data level in coefficient
// form, KSwitchKeys::data()[index][digit].data() is [2][K][N] in NTT form -- the layouts include/hhe_gfx950.h takes.
//
// tests/test_seal_adapter.py type-checks this file and the CSP's call sequence against the reference's SEAL 4.0.0 headers
// (g++ -fsyntax-only; the prebuilt libseal is never linked or loaded), so it cannot be executed in this repository.
#pragma once

#include <array>
#include <cmath>
#include <map>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

#include "Cipher.h"         // reference: pasta::ZpCipherParams
#include "pasta_3_plain.h"  // reference: PASTA_PARAMS, PASTA_T
#include "seal/seal.h"

#include "hhe_gfx950.h"
#include "hhe_keyset_cache.hpp"

namespace pasta {
namespace gfx950 {

inline void check(int rc)
{
    if (rc == HHE_OK) return;
    const std::string msg = hhe_last_error();
    switch (rc) {
    case HHE_ERR_TOO_FEW_SLOTS: throw std::runtime_error(msg);  // pasta_3_seal.cpp:376-377
    case HHE_ERR_NO_GALOIS_KEY:
    case HHE_ERR_NO_RELIN_KEY:
    case HHE_ERR_INVALID: throw std::invalid_argument(msg);     // what SEAL throws for these
    default: throw std::runtime_error(msg);
    }
}

struct DevBuf {  // RAII device buffer
    void *p = nullptr;
    explicit DevBuf(std::size_t bytes) : p(hhe_malloc(bytes)) { if (!p) throw std::runtime_error("hhe_malloc failed"); }
    ~DevBuf() { hhe_free(p); }
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    std::uint64_t *u64() const { return static_cast<std::uint64_t *>(p); }
};

// One device context per SEAL parameter set, shared by every cipher object of the process: BaseCSP::decompose builds a
// fresh PASTA_SEAL per request (CSP.cpp:238-242), the key-switch keys stay in HBM between requests.  Every RelinKeys /
// GaloisKeys OBJECT that reaches the adapter (a constructor argument, the galois_keys of flatten, the keys of the FC calls) maps
// to one device key set, recognised by a hash over all of its words (hhe::KeySetCache): analyst_he_gk and csp_he_gk, which
// share elements but not words (Analyst.cpp:62-94), live side by side and every call uses the one it names.
class DeviceContext {
public:
    explicit DeviceContext(const seal::SEALContext &context, int device = 0)
    {
        const auto &parms = context.key_context_data()->parms();
        std::vector<std::uint64_t> q;
        for (const auto &m : parms.coeff_modulus()) q.push_back(m.value());
        n_ = parms.poly_modulus_degree();
        L_ = q.size() - 1;
        t_ = parms.plain_modulus().value();
        first_parms_id_ = context.first_parms_id();
        check(hhe_ctx_create(seal::util::get_power_of_two(n_), static_cast<int>(q.size()), q.data(), t_, device, &h_));
    }
    ~DeviceContext() { keys_.reset(); hhe_ctx_destroy(h_); }   // key sets go before their context
    DeviceContext(const DeviceContext &) = delete;
    DeviceContext &operator=(const DeviceContext &) = delete;

    hhe_ctx *handle() const { return h_; }
    std::size_t n() const { return n_; }
    std::size_t ct_words(std::size_t size = 2) const { return size * L_ * n_; }
    const seal::parms_id_type &first_parms_id() const { return first_parms_id_; }

    // KSwitchKeys::data()[index] = one PublicKey per digit, each a size-2 K-limb NTT-form ciphertext -> [L][2][K][N]
    static std::vector<std::uint64_t> flatten_ksk(const std::vector<seal::PublicKey> &digits)
    {
        std::vector<std::uint64_t> out;
        for (const auto &pk : digits) {
            const seal::Ciphertext &ct = pk.data();
            out.insert(out.end(), ct.data(), ct.data() + ct.size() * ct.coeff_modulus_size() * ct.poly_modulus_degree());
        }
        return out;
    }
    // the device key set of a key object (nullptr for an empty object: the call then reports the missing key as SEAL would)
    hhe_keyset *relin_set(const seal::RelinKeys &rk)
    {
        if (rk.data().empty() || rk.data()[0].empty()) return nullptr;
        const auto words = flatten_ksk(rk.data()[seal::RelinKeys::get_index(2)]);
        return keys().relin(words.data(), words.size());
    }
    hhe_keyset *galois_set(const seal::GaloisKeys &gk)
    {
        std::vector<std::vector<std::uint64_t>> flat;
        std::vector<std::pair<std::uint32_t, const std::uint64_t *>> v;
        for (std::size_t idx = 0; idx < gk.data().size(); idx++) {
            if (gk.data()[idx].empty()) continue;
            flat.push_back(flatten_ksk(gk.data()[idx]));   // GaloisKeys::get_index(elt) = (elt - 1) / 2
            v.emplace_back(static_cast<std::uint32_t>(2 * idx + 1), nullptr);
        }
        if (v.empty()) return nullptr;
        for (std::size_t i = 0; i < v.size(); i++) v[i].second = flat[i].data();
        return keys().galois(v, flat[0].size());
    }
    hhe::KeySetCache &keys() { std::lock_guard<std::mutex> lk(mu_); if (!keys_) keys_.reset(new hhe::KeySetCache(h_)); return *keys_; }
    hhe::DeviceArena &arena() { return arena_; }
    // enc_ssk[0] arrives by value with every call (CSP.cpp:249): it crosses PCIe only when its contents change (arena slot 2;
    // the caller holds the arena's lock)
    std::uint64_t *encrypted_key(const seal::Ciphertext &ct)
    {
        if (ct.is_ntt_form() || ct.poly_modulus_degree() != n_ || ct.coeff_modulus_size() != L_ || ct.size() != 2)
            throw std::invalid_argument("encrypted is not valid for encryption parameters");
        hhe::ContentHash hsh;
        hsh.add(ct.data(), ct_words());
        std::uint64_t *d = arena_.get(2, ct_words() * 8);
        if (!key_resident_ || key_hash_ < hsh || hsh < key_hash_) {
            check(hhe_copy_h2d(h_, d, ct.data(), ct_words() * 8));
            key_hash_ = hsh;
            key_resident_ = true;
        }
        return d;
    }

    void to_device(const seal::Ciphertext &ct, std::uint64_t *dptr) const
    {
        if (ct.is_ntt_form() || ct.poly_modulus_degree() != n_ || ct.coeff_modulus_size() != L_)
            throw std::invalid_argument("encrypted is not valid for encryption parameters");
        check(hhe_copy_h2d(h_, dptr, ct.data(), ct_words(ct.size()) * 8));
    }
    void from_device(const seal::SEALContext &context, const std::uint64_t *dptr, std::size_t size, seal::Ciphertext &ct) const
    {
        ct.resize(context, first_parms_id_, size);
        ct.is_ntt_form() = false;
        ct.scale() = 1.0;
        check(hhe_copy_d2h(h_, ct.data(), dptr, ct_words(size) * 8));
    }

    // process-wide registry keyed by the data-level parms_id (what every ciphertext of the path carries)
    static std::shared_ptr<DeviceContext> get(const seal::SEALContext &context)
    {
        std::lock_guard<std::mutex> lk(registry_mutex());
        auto &reg = registry();
        auto it = reg.find(context.first_parms_id());
        if (it != reg.end()) return it->second;
        auto dc = std::make_shared<DeviceContext>(context);
        reg.emplace(context.first_parms_id(), dc);
        return dc;
    }
    static std::shared_ptr<DeviceContext> find(const seal::parms_id_type &id)
    {
        std::lock_guard<std::mutex> lk(registry_mutex());
        auto it = registry().find(id);
        if (it == registry().end()) throw std::invalid_argument("no gfx950 device context for these encryption parameters (construct a pasta::PASTA_SEAL first)");
        return it->second;
    }

private:
    static std::map<seal::parms_id_type, std::shared_ptr<DeviceContext>> &registry()
    {
        static std::map<seal::parms_id_type, std::shared_ptr<DeviceContext>> r;
        return r;
    }
    static std::mutex &registry_mutex()
    {
        static std::mutex m;
        return m;
    }
    hhe_ctx *h_ = nullptr;
    std::size_t n_ = 0, L_ = 0;
    std::uint64_t t_ = 0;
    seal::parms_id_type first_parms_id_{};
    std::mutex mu_;
    std::unique_ptr<hhe::KeySetCache> keys_;
    hhe::DeviceArena arena_;
    hhe::ContentHash key_hash_;
    bool key_resident_ = false;
};

}  // namespace gfx950

// ---------------------------------------------------------------------------------------------------------------------
// pasta::SEALZpCipher (src/pasta/SEAL_Cipher.h:11-129).  The members the CSP path uses run on the device; the SEAL objects
// the reference keeps as members are kept too (encrypt_key, decrypt_result and the analyst/user sides use them unchanged).
class SEALZpCipher {
public:
    typedef std::vector<uint64_t> vector;
    typedef std::vector<std::vector<uint64_t>> matrix;

protected:
    std::vector<uint64_t> secret_key;
    ZpCipherParams params;
    uint64_t plain_mod;
    uint64_t mod_degree;

    std::vector<seal::Ciphertext> secret_key_encrypted;

    std::shared_ptr<seal::SEALContext> context;
    seal::KeyGenerator keygen;

    seal::SecretKey he_sk;
    seal::PublicKey he_pk;
    seal::RelinKeys he_rk;
    seal::GaloisKeys he_gk;

    seal::Encryptor encryptor;
    seal::Evaluator evaluator;
    seal::Decryptor decryptor;
    seal::BatchEncoder batch_encoder;

    std::vector<int> gk_indices;

    bool use_bsgs = false;
    size_t bsgs_n1 = 0;
    size_t bsgs_n2 = 0;

    std::shared_ptr<gfx950::DeviceContext> device;  // HBM-resident keys and tables, shared per parameter set
    // the device key sets of he_rk / he_gk (owned by the device context's cache); an object built without a key gets an empty set of
    // its own instead of falling through to another object's keys (declared after `device`: released while the context exists)
    hhe_keyset *rk_set = nullptr, *gk_set = nullptr, *empty_set = nullptr;
    std::shared_ptr<hhe_keyset> owned_empty;
    const hhe_keyset *or_empty(const hhe_keyset *ks)
    {
        if (ks) return ks;
        if (!empty_set) {
            gfx950::check(hhe_keyset_create(device->handle(), &empty_set));
            owned_empty.reset(empty_set, [](hhe_keyset *k) { hhe_keyset_destroy(k); });
        }
        return empty_set;
    }

public:
    // src/pasta/SEAL_Cipher.cpp:9-36 (all arguments by value, as the reference takes them)
    SEALZpCipher(ZpCipherParams params, std::shared_ptr<seal::SEALContext> con, seal::PublicKey pk, seal::SecretKey sk,
                 seal::RelinKeys rk, seal::GaloisKeys gk)
        : params(params), context(con), keygen(*context, sk), he_sk(sk), he_pk(pk), he_rk(rk), he_gk(gk),
          encryptor(*context, pk), evaluator(*context), decryptor(*context, sk), batch_encoder(*context),
          device(gfx950::DeviceContext::get(*context))
    {
        encryptor.set_public_key(pk);
        mod_degree = context->first_context_data()->parms().poly_modulus_degree();
        plain_mod = context->first_context_data()->parms().plain_modulus().value();
        rk_set = device->relin_set(he_rk);
        gk_set = device->galois_set(he_gk);
    }
    virtual ~SEALZpCipher() = default;

    size_t get_key_size() const { return params.key_size; }
    size_t get_plain_size() const { return params.plain_size; }
    size_t get_cipher_size() const { return params.cipher_size; }

    void add_some_gk_indices(std::vector<int> &gk_ind)
    {
        for (auto &it : gk_ind) gk_indices.push_back(it);
    }
    void create_gk() { keygen.create_galois_keys(gk_indices, he_gk); gk_set = device->galois_set(he_gk); }

    virtual std::string get_cipher_name() const = 0;

    // src/pasta/SEAL_Cipher.cpp:38-68
    static std::shared_ptr<seal::SEALContext> create_context(size_t mod_degree, uint64_t plain_mod, int seclevel = 128)
    {
        if (seclevel != 128) throw std::runtime_error("Security Level not supported");
        seal::sec_level_type sec = seal::sec_level_type::tc128;
        seal::EncryptionParameters parms(seal::scheme_type::bfv);
        parms.set_poly_modulus_degree(mod_degree);
        if (mod_degree == 65536) {
            sec = seal::sec_level_type::none;
            uint64_t q[64];
            size_t cnt = 64;
            gfx950::check(hhe_bfv_default_coeff_modulus(mod_degree, q, &cnt));  // the reference's hard-coded 29-prime chain (:50-60)
            std::vector<seal::Modulus> mods;
            for (size_t i = 0; i < cnt; i++) mods.emplace_back(q[i]);
            parms.set_coeff_modulus(mods);
        } else {
            parms.set_coeff_modulus(seal::CoeffModulus::BFVDefault(mod_degree));
        }
        parms.set_plain_modulus(plain_mod);
        return std::make_shared<seal::SEALContext>(parms, true, sec);
    }

    virtual std::vector<seal::Ciphertext> HE_decrypt(std::vector<uint64_t> &ciphertext, bool batch_encoder = false) = 0;
    virtual std::vector<uint64_t> decrypt_result(std::vector<seal::Ciphertext> &ciphertext, bool batch_encoder = false) = 0;
    virtual void add_gk_indices() = 0;

    void activate_bsgs(bool activate) { use_bsgs = activate; }
    void set_bsgs_params(uint64_t bsgs_n1, uint64_t bsgs_n2) { this->bsgs_n1 = bsgs_n1; this->bsgs_n2 = bsgs_n2; }

    // SEALZpCipher::mask (SEAL_Cipher.cpp:161-166): batch_encoder.encode(mask) + multiply_plain_inplace
    void mask(seal::Ciphertext &cipher, std::vector<uint64_t> &mask)
    {
        std::lock_guard<std::mutex> lk(device->arena().mutex());
        std::uint64_t *d = device->arena().get(0, device->ct_words() * 8);
        device->to_device(cipher, d);
        gfx950::check(hhe_mask(device->handle(), d, mask.data(), mask.size(), d, 1));
        device->from_device(*context, d, 2, cipher);
    }
    // SEALZpCipher::flatten (SEAL_Cipher.cpp:170-181): out = sum_i rotate_rows(in[i], -i * plain_size, galois_keys)
    void flatten(std::vector<seal::Ciphertext> &in, seal::Ciphertext &out, const seal::GaloisKeys &galois_keys)
    {
        if (in.empty()) throw std::invalid_argument("flatten: empty input");
        const hhe_keyset *gk = or_empty(device->galois_set(galois_keys));   // the GaloisKeys object THIS call names (CSP.cpp:271-278: csp_he_gk)
        const size_t w = device->ct_words();
        std::lock_guard<std::mutex> lk(device->arena().mutex());
        std::uint64_t *d = device->arena().get(0, in.size() * w * 8), *o = device->arena().get(1, w * 8);
        for (size_t i = 0; i < in.size(); i++) device->to_device(in[i], d + i * w);
        gfx950::check(hhe_flatten_ks(device->handle(), gk, d, in.size(), o, 1));
        device->from_device(*context, o, 2, out);
    }

    // packed helpers of the FC (SEAL_Cipher.cpp:547-566)
    void packed_square(seal::Ciphertext &vo, const seal::Ciphertext &vi)
    {
        const size_t w = device->ct_words();
        gfx950::DevBuf a(w * 8), o3(device->ct_words(3) * 8);
        device->to_device(vi, a.u64());
        gfx950::check(hhe_multiply(device->handle(), a.u64(), a.u64(), o3.u64(), 1));
        gfx950::check(hhe_relinearize_ks(device->handle(), or_empty(rk_set), o3.u64(), a.u64(), 1));
        device->from_device(*context, a.u64(), 2, vo);
    }
    void packed_enc_mul(const seal::Ciphertext &encrypted1, const seal::Ciphertext &encrypted2, seal::Ciphertext &destination)
    {
        const size_t w = device->ct_words();
        gfx950::DevBuf a(w * 8), b(w * 8), o3(device->ct_words(3) * 8);
        device->to_device(encrypted1, a.u64());
        device->to_device(encrypted2, b.u64());
        gfx950::check(hhe_multiply(device->handle(), a.u64(), b.u64(), o3.u64(), 1));
        device->from_device(*context, o3.u64(), 3, destination);
    }
    void packed_enc_add(const seal::Ciphertext &encrypted1, const seal::Ciphertext &encrypted2, seal::Ciphertext &destination)
    {
        if (encrypted1.size() != encrypted2.size()) throw std::invalid_argument("encrypted1 and encrypted2 parameter mismatch");
        const size_t w = device->ct_words(encrypted1.size());
        gfx950::DevBuf a(w * 8), b(w * 8);
        device->to_device(encrypted1, a.u64());
        device->to_device(encrypted2, b.u64());
        gfx950::check(hhe_add(device->handle(), a.u64(), b.u64(), a.u64(), 1, static_cast<int>(encrypted1.size())));
        device->from_device(*context, a.u64(), encrypted1.size(), destination);
    }
};

// ---------------------------------------------------------------------------------------------------------------------
// pasta::PASTA_SEAL (src/pasta/pasta_3_seal.h:8-54)
class PASTA_SEAL : public SEALZpCipher {
public:
    typedef PASTA Plain;
    PASTA_SEAL(std::shared_ptr<seal::SEALContext> con, seal::PublicKey pk, seal::SecretKey sk, seal::RelinKeys rk, seal::GaloisKeys gk)
        : SEALZpCipher(PASTA_PARAMS, con, pk, sk, rk, gk), slots(this->batch_encoder.slot_count()), halfslots(slots >> 1) {}

    virtual ~PASTA_SEAL() = default;

    virtual std::string get_cipher_name() const { return "PASTA-SEAL (n=128,r=3)"; }

    // pasta_3_seal.cpp:8-21 (client-side key encryption stays on SEAL: one encode + one encrypt)
    virtual void encrypt_key(bool batch_encoder = false)
    {
        (void)batch_encoder;
        secret_key_encrypted = encrypt_key_2(secret_key, batch_encoder);
    }
    // pasta_3_seal.cpp:23-38
    virtual std::vector<seal::Ciphertext> encrypt_key_2(std::vector<uint64_t> ssk, bool batch_encoder = false)
    {
        (void)batch_encoder;
        std::vector<seal::Ciphertext> enc_sk(1);
        seal::Plaintext k;
        std::vector<uint64_t> key_tmp(halfslots + PASTA_T, 0);
        for (size_t i = 0; i < PASTA_T; i++) {
            key_tmp[i] = ssk[i];
            key_tmp[i + halfslots] = ssk[i + PASTA_T];
        }
        this->batch_encoder.encode(key_tmp, k);
        encryptor.encrypt(k, enc_sk[0]);
        return enc_sk;
    }

    // pasta_3_seal.cpp:42-104 == decomposition(ciphertexts, secret_key_encrypted) without the debug noise printing (:73)
    virtual std::vector<seal::Ciphertext> HE_decrypt(std::vector<uint64_t> &ciphertext, bool batch_encoder = false)
    {
        return decomposition(ciphertext, secret_key_encrypted, batch_encoder);
    }

    // pasta_3_seal.cpp:106-172: every 128-word block of the record on the device, one batched call
    virtual std::vector<seal::Ciphertext> decomposition(std::vector<uint64_t> &ciphertext, std::vector<seal::Ciphertext> enc_ssk,
                                                        bool batch_encoder = false)
    {
        (void)batch_encoder;  // ignored by the reference as well (:113)
        if (enc_ssk.empty()) throw std::invalid_argument("decomposition: enc_ssk is empty");
        const size_t size = ciphertext.size();
        const size_t num_block = static_cast<size_t>(std::ceil(static_cast<double>(size) / params.cipher_size));
        std::vector<seal::Ciphertext> res(num_block);
        if (num_block == 0) return res;
        std::vector<uint64_t> cw(num_block * PASTA_T, 0), bidx(num_block);
        std::vector<uint32_t> ncw(num_block);
        for (size_t b = 0; b < num_block; b++) {
            const size_t lo = b * params.cipher_size, hi = std::min(lo + params.cipher_size, size);
            for (size_t i = lo; i < hi; i++) cw[b * PASTA_T + (i - lo)] = ciphertext[i];
            ncw[b] = static_cast<uint32_t>(hi - lo);
            bidx[b] = b;  // pasta.init_shake(nonce, b) (:122)
        }
        const size_t w = device->ct_words();
        std::lock_guard<std::mutex> lk(device->arena().mutex());
        std::uint64_t *key = device->encrypted_key(enc_ssk[0]);  // state <- enc_ssk[0] (:126); uploaded when its contents change
        std::uint64_t *out = device->arena().get(3, num_block * w * 8);
        gfx950::check(hhe_pasta3_transcipher_ks(device->handle(), or_empty(rk_set), or_empty(gk_set), key, cw.data(), ncw.data(), bidx.data(),
                                                num_block, use_bsgs ? 1 : 0, out));
        for (size_t b = 0; b < num_block; b++) device->from_device(*context, out + b * w, 2, res[b]);
        return res;
    }

    // BaseCSP::decompose's per-record loop (CSP.cpp:247-278) as ONE device call: decomposition of every record, the mask of the
    // ragged last block (mask_last: as hhe_pktnn_examples.cpp:620-626; the CSP's own loop masks a copy, i.e. pass false to reproduce
    // that) and flatten with the GaloisKeys object `flatten_gk` -- the blocks never leave HBM.  One flattened ciphertext per record.
    std::vector<seal::Ciphertext> decompose(const std::vector<std::vector<uint64_t>> &records, std::vector<seal::Ciphertext> enc_ssk,
                                            const seal::GaloisKeys &flatten_gk, bool mask_last)
    {
        std::vector<seal::Ciphertext> res(records.size());
        if (records.empty()) return res;
        if (enc_ssk.empty()) throw std::invalid_argument("decompose: enc_ssk is empty");
        const size_t nwords = records[0].size(), w = device->ct_words();
        std::vector<uint64_t> flat(records.size() * nwords);
        for (size_t s = 0; s < records.size(); s++) {
            if (records[s].size() != nwords) throw std::invalid_argument("decompose: records of different lengths");
            std::copy(records[s].begin(), records[s].end(), flat.begin() + s * nwords);
        }
        const hhe_keyset *fgk = or_empty(device->galois_set(flatten_gk));
        std::lock_guard<std::mutex> lk(device->arena().mutex());
        std::uint64_t *key = device->encrypted_key(enc_ssk[0]), *out = device->arena().get(3, records.size() * w * 8);
        gfx950::check(hhe_decompose_ks(device->handle(), or_empty(rk_set), or_empty(gk_set), fgk, key, flat.data(), records.size(), nwords,
                                       mask_last ? 1 : 0, out));
        for (size_t s = 0; s < records.size(); s++) device->from_device(*context, out + s * w, 2, res[s]);
        return res;
    }

    // pasta_3_seal.cpp:176-188 (analyst side, one decrypt + decode: stays on SEAL)
    virtual std::vector<uint64_t> decrypt_result(std::vector<seal::Ciphertext> &ciphertext, bool batch_encoder = false)
    {
        (void)batch_encoder;
        seal::Plaintext p;
        std::vector<uint64_t> res;
        decryptor.decrypt(ciphertext[0], p);
        this->batch_encoder.decode(p, res);
        res.resize(params.plain_size);
        return res;
    }

    // pasta_3_seal.cpp:190-201
    virtual void add_gk_indices()
    {
        gk_indices.push_back(0);
        gk_indices.push_back(-1);
        if (PASTA_T * 2 != batch_encoder.slot_count()) gk_indices.push_back(static_cast<int>(PASTA_T));
        if (use_bsgs)
            for (uint64_t k = 1; k < BSGS_N2; k++) gk_indices.push_back(-static_cast<int>(k * BSGS_N1));
    }

private:
    static constexpr uint64_t BSGS_N1 = 16;
    static constexpr uint64_t BSGS_N2 = 8;
    size_t slots;
    size_t halfslots;
};

}  // namespace pasta

// ---------------------------------------------------------------------------------------------------------------------
// sealhelper::packed_enc_multiply / encrypted_vec_sum (src/util/sealhelper.h:84-129, sealhelper.cpp:268-274,379-392) with the
// reference's signatures; the Evaluator argument is unused (the device context is found through the ciphertext's parms_id).
namespace sealhelper {

inline void packed_enc_multiply(const seal::Ciphertext &encrypted1, const seal::Ciphertext &encrypted2, seal::Ciphertext &destination,
                                const seal::Evaluator &evaluator)
{
    (void)evaluator;
    auto dev = pasta::gfx950::DeviceContext::find(encrypted1.parms_id());
    const size_t w = dev->ct_words();
    pasta::gfx950::DevBuf a(w * 8), b(w * 8), o3(dev->ct_words(3) * 8);
    dev->to_device(encrypted1, a.u64());
    dev->to_device(encrypted2, b.u64());
    pasta::gfx950::check(hhe_multiply(dev->handle(), a.u64(), b.u64(), o3.u64(), 1));
    // the result ciphertext keeps the parameters of its inputs; resize needs the SEALContext, which a Ciphertext does not
    // carry: destination is sized from encrypted1 (same parms_id, pool) and grown to three polynomials
    destination = encrypted1;
    destination.resize(3);
    pasta::gfx950::check(hhe_copy_d2h(dev->handle(), destination.data(), o3.u64(), dev->ct_words(3) * 8));
}

// Evaluator::relinearize_inplace(record, csp_rk) as the CSP calls it between the two (CSP.cpp:306), on the device, with the
// RelinKeys object the call names (its key set is uploaded the first time the object is seen).
inline void relinearize_inplace(seal::Ciphertext &encrypted, const seal::RelinKeys &relin_keys)
{
    auto dev = pasta::gfx950::DeviceContext::find(encrypted.parms_id());
    hhe_keyset *rk = dev->relin_set(relin_keys);
    if (!rk) throw std::invalid_argument("relin_keys is not valid for encryption parameters");
    pasta::gfx950::DevBuf a3(dev->ct_words(3) * 8), o(dev->ct_words() * 8);
    if (encrypted.size() != 3) throw std::invalid_argument("encrypted is not valid for encryption parameters");
    pasta::gfx950::check(hhe_copy_h2d(dev->handle(), a3.u64(), encrypted.data(), dev->ct_words(3) * 8));
    pasta::gfx950::check(hhe_relinearize_ks(dev->handle(), rk, a3.u64(), o.u64(), 1));
    encrypted.resize(2);
    pasta::gfx950::check(hhe_copy_d2h(dev->handle(), encrypted.data(), o.u64(), dev->ct_words() * 8));
}

inline void encrypted_vec_sum(const seal::Ciphertext &encrypted_inp, seal::Ciphertext &destination, const seal::Evaluator &evaluator,
                              const seal::GaloisKeys &gal_keys, const size_t vec_size)
{
    (void)evaluator;
    auto dev = pasta::gfx950::DeviceContext::find(encrypted_inp.parms_id());
    hhe_keyset *gk = dev->galois_set(gal_keys);
    if (!gk) throw std::invalid_argument("Galois key not present");
    const size_t w = dev->ct_words();
    pasta::gfx950::DevBuf in(w * 8), acc(w * 8), rot(w * 8);
    dev->to_device(encrypted_inp, in.u64());
    // destination = encrypted_inp; for i = -1 .. -(vec_size-1): destination += rotate_rows(encrypted_inp, i)  (sealhelper.cpp:385-391)
    pasta::gfx950::check(hhe_rotate_rows_ks(dev->handle(), gk, in.u64(), 0, acc.u64(), 1));  // step 0: a copy, as in SEAL
    for (size_t i = 1; i < vec_size; i++) {
        pasta::gfx950::check(hhe_rotate_rows_ks(dev->handle(), gk, in.u64(), -static_cast<int>(i), rot.u64(), 1));
        pasta::gfx950::check(hhe_add(dev->handle(), acc.u64(), rot.u64(), acc.u64(), 1, 2));
    }
    destination = encrypted_inp;
    pasta::gfx950::check(hhe_copy_d2h(dev->handle(), destination.data(), acc.u64(), w * 8));
}

// The three FC calls of CSP_hhe_pktnn_1fc::evaluateModel (CSP.cpp:296-316) as ONE device call: multiply, relinearize with
// the CSP's RelinKeys, NAF-trie rotation sum with the analyst's default Galois keys -- identical ciphertext words.
inline void fc_row(const seal::Ciphertext &vi, const seal::Ciphertext &w_row, const seal::RelinKeys &csp_rk, const seal::GaloisKeys &gal_keys,
                   size_t vec_size, seal::Ciphertext &destination)
{
    auto dev = pasta::gfx950::DeviceContext::find(vi.parms_id());
    hhe_keyset *rk = dev->relin_set(csp_rk), *gk = dev->galois_set(gal_keys);
    if (!rk || !gk) throw std::invalid_argument("fc_row: empty key object");
    const size_t w = dev->ct_words();
    std::lock_guard<std::mutex> lk(dev->arena().mutex());
    std::uint64_t *a = dev->arena().get(0, w * 8), *b = dev->arena().get(1, w * 8), *o = dev->arena().get(3, w * 8);
    dev->to_device(vi, a);
    dev->to_device(w_row, b);
    pasta::gfx950::check(hhe_fc_row_ks(dev->handle(), rk, gk, a, b, 1, vec_size, o, 1));
    destination = vi;
    pasta::gfx950::check(hhe_copy_d2h(dev->handle(), destination.data(), o, w * 8));
}

}  // namespace sealhelper
