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
// fresh PASTA_SEAL per request (CSP.cpp:238-242), the key-switch keys stay in HBM between requests.
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
    ~DeviceContext() { hhe_ctx_destroy(h_); }
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
    static std::array<std::uint64_t, 4> fingerprint(const std::vector<std::uint64_t> &w)
    {
        return {w.size(), w.empty() ? 0 : w.front(), w.empty() ? 0 : w[w.size() / 2], w.empty() ? 0 : w.back()};
    }
    void upload_relin(const seal::RelinKeys &rk, int slot)
    {
        if (rk.data().empty() || rk.data()[0].empty()) return;
        const auto words = flatten_ksk(rk.data()[seal::RelinKeys::get_index(2)]);
        std::lock_guard<std::mutex> lk(mu_);
        if (relin_fp_[slot] == fingerprint(words)) return;
        check(hhe_set_relin_key_slot(h_, slot, words.data()));
        relin_fp_[slot] = fingerprint(words);
    }
    // every key of the object; a second upload of the same words is skipped
    void upload_galois(const seal::GaloisKeys &gk)
    {
        for (std::size_t idx = 0; idx < gk.data().size(); idx++) {
            if (gk.data()[idx].empty()) continue;
            const std::uint32_t elt = static_cast<std::uint32_t>(2 * idx + 1);  // GaloisKeys::get_index(elt) = (elt - 1) / 2
            const auto words = flatten_ksk(gk.data()[idx]);
            std::lock_guard<std::mutex> lk(mu_);
            auto it = galois_fp_.find(elt);
            if (it != galois_fp_.end() && it->second == fingerprint(words)) continue;
            check(hhe_set_galois_key(h_, elt, words.data()));
            galois_fp_[elt] = fingerprint(words);
        }
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
    std::array<std::uint64_t, 4> relin_fp_[4] = {};
    std::map<std::uint32_t, std::array<std::uint64_t, 4>> galois_fp_;
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
        device->upload_relin(he_rk, 0);
        device->upload_galois(he_gk);
    }
    virtual ~SEALZpCipher() = default;

    size_t get_key_size() const { return params.key_size; }
    size_t get_plain_size() const { return params.plain_size; }
    size_t get_cipher_size() const { return params.cipher_size; }

    void add_some_gk_indices(std::vector<int> &gk_ind)
    {
        for (auto &it : gk_ind) gk_indices.push_back(it);
    }
    void create_gk() { keygen.create_galois_keys(gk_indices, he_gk); device->upload_galois(he_gk); }

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
        gfx950::DevBuf d(device->ct_words() * 8);
        device->to_device(cipher, d.u64());
        gfx950::check(hhe_mask(device->handle(), d.u64(), mask.data(), mask.size(), d.u64(), 1));
        device->from_device(*context, d.u64(), 2, cipher);
    }
    // SEALZpCipher::flatten (SEAL_Cipher.cpp:170-181): out = sum_i rotate_rows(in[i], -i * plain_size, galois_keys)
    void flatten(std::vector<seal::Ciphertext> &in, seal::Ciphertext &out, const seal::GaloisKeys &galois_keys)
    {
        if (in.empty()) throw std::invalid_argument("flatten: empty input");
        device->upload_galois(galois_keys);
        const size_t w = device->ct_words();
        gfx950::DevBuf d(in.size() * w * 8), o(w * 8);
        for (size_t i = 0; i < in.size(); i++) device->to_device(in[i], d.u64() + i * w);
        gfx950::check(hhe_flatten(device->handle(), d.u64(), in.size(), o.u64(), 1));
        device->from_device(*context, o.u64(), 2, out);
    }

    // packed helpers of the FC (SEAL_Cipher.cpp:547-566)
    void packed_square(seal::Ciphertext &vo, const seal::Ciphertext &vi)
    {
        const size_t w = device->ct_words();
        gfx950::DevBuf a(w * 8), o3(device->ct_words(3) * 8);
        device->to_device(vi, a.u64());
        gfx950::check(hhe_multiply(device->handle(), a.u64(), a.u64(), o3.u64(), 1));
        gfx950::check(hhe_relinearize(device->handle(), o3.u64(), a.u64(), 1));
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
        gfx950::DevBuf key(w * 8), out(num_block * w * 8);
        device->to_device(enc_ssk[0], key.u64());  // state <- enc_ssk[0] (:126)
        gfx950::check(hhe_pasta3_transcipher(device->handle(), key.u64(), cw.data(), ncw.data(), bidx.data(), num_block,
                                             use_bsgs ? 1 : 0, out.u64()));
        for (size_t b = 0; b < num_block; b++) device->from_device(*context, out.u64() + b * w, 2, res[b]);
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

// Evaluator::relinearize_inplace(record, csp_rk) as the CSP calls it between the two (CSP.cpp:306), on the device.  The
// CSP's RelinKeys object lives in slot 1 (slot 0 holds the key PASTA_SEAL was constructed with).
inline void relinearize_inplace(seal::Ciphertext &encrypted, const seal::RelinKeys &relin_keys, int relin_slot = 1)
{
    auto dev = pasta::gfx950::DeviceContext::find(encrypted.parms_id());
    dev->upload_relin(relin_keys, relin_slot);
    pasta::gfx950::DevBuf a3(dev->ct_words(3) * 8), o(dev->ct_words() * 8);
    if (encrypted.size() != 3) throw std::invalid_argument("encrypted is not valid for encryption parameters");
    pasta::gfx950::check(hhe_copy_h2d(dev->handle(), a3.u64(), encrypted.data(), dev->ct_words(3) * 8));
    pasta::gfx950::check(hhe_relinearize_slot(dev->handle(), relin_slot, a3.u64(), o.u64(), 1));
    encrypted.resize(2);
    pasta::gfx950::check(hhe_copy_d2h(dev->handle(), encrypted.data(), o.u64(), dev->ct_words() * 8));
}

inline void encrypted_vec_sum(const seal::Ciphertext &encrypted_inp, seal::Ciphertext &destination, const seal::Evaluator &evaluator,
                              const seal::GaloisKeys &gal_keys, const size_t vec_size)
{
    (void)evaluator;
    auto dev = pasta::gfx950::DeviceContext::find(encrypted_inp.parms_id());
    dev->upload_galois(gal_keys);
    const size_t w = dev->ct_words();
    pasta::gfx950::DevBuf in(w * 8), acc(w * 8), rot(w * 8);
    dev->to_device(encrypted_inp, in.u64());
    // destination = encrypted_inp; for i = -1 .. -(vec_size-1): destination += rotate_rows(encrypted_inp, i)  (sealhelper.cpp:385-391)
    pasta::gfx950::check(hhe_rotate_rows(dev->handle(), in.u64(), 0, acc.u64(), 1));  // step 0: a copy, as in SEAL
    for (size_t i = 1; i < vec_size; i++) {
        pasta::gfx950::check(hhe_rotate_rows(dev->handle(), in.u64(), -static_cast<int>(i), rot.u64(), 1));
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
    dev->upload_relin(csp_rk, 1);
    dev->upload_galois(gal_keys);
    const size_t w = dev->ct_words();
    pasta::gfx950::DevBuf a(w * 8), b(w * 8), o(w * 8);
    dev->to_device(vi, a.u64());
    dev->to_device(w_row, b.u64());
    pasta::gfx950::check(hhe_fc_row(dev->handle(), a.u64(), b.u64(), 1, vec_size, 1, 1, o.u64(), 1));
    destination = vi;
    pasta::gfx950::check(hhe_copy_d2h(dev->handle(), destination.data(), o.u64(), w * 8));
}

}  // namespace sealhelper
