// TESTS ONLY -- type-checks include/pasta_seal_gfx950_seal.hpp against the reference's SEAL 4.0.0 headers and replays the
// calls BaseCSP::decompose (src/examples/CSP/CSP.cpp:238-278) and CSP_hhe_pktnn_1fc::evaluateModel (:296-316) make, with
// the same argument types.  Compiled with g++ -fsyntax-only (tests/test_seal_adapter.py); never linked or run.
#include "pasta_seal_gfx950_seal.hpp"

using namespace seal;
using namespace std;

// CSP.cpp:235-283
void csp_decompose(shared_ptr<SEALContext> context, PublicKey analyst_pk, SecretKey csp_sk, RelinKeys analyst_rk, GaloisKeys analyst_gk,
                   GaloisKeys csp_gk, vector<vector<uint64_t>> &enc_data, vector<Ciphertext> user_enc_sym_key, int inputLen,
                   vector<Ciphertext> &processed)
{
    pasta::PASTA_SEAL HHE(context, analyst_pk, csp_sk, analyst_rk, analyst_gk);
    vector<vector<Ciphertext>> he_enc_data;
    for (vector<uint64_t> record : enc_data) he_enc_data.push_back(HHE.decomposition(record, user_enc_sym_key, true));
    size_t rem = inputLen % HHE.get_plain_size();
    if (rem != 0) {
        vector<uint64_t> mask(rem, 1);
        for (vector<Ciphertext> record : he_enc_data) HHE.mask(record.back(), mask);
    }
    Ciphertext tmp;
    for (vector<Ciphertext> record : he_enc_data) {
        HHE.flatten(record, tmp, csp_gk);
        processed.push_back(tmp);
    }
    // the same loop as one batched device call (an addition of the adapter, not a reference signature)
    vector<Ciphertext> batched = HHE.decompose(enc_data, user_enc_sym_key, csp_gk, true);
    processed.insert(processed.end(), batched.begin(), batched.end());
}

// CSP.cpp:288-323
void csp_evaluate(shared_ptr<SEALContext> context, vector<Ciphertext> &processed, Ciphertext enc_weight_row, RelinKeys csp_rk,
                  GaloisKeys analyst_gk, int inputLen, vector<Ciphertext> &sums)
{
    Evaluator evaluator(*context);
    vector<Ciphertext> products;
    Ciphertext tmp;
    for (Ciphertext record : processed) {
        sealhelper::packed_enc_multiply(record, enc_weight_row, tmp, evaluator);
        products.push_back(tmp);
    }
    Ciphertext tmp1;
    for (Ciphertext record : products) {
        sealhelper::relinearize_inplace(record, csp_rk);  // CSP.cpp:306 calls getEvaluator()->relinearize_inplace(record, csp_rk)
        sealhelper::encrypted_vec_sum(record, tmp1, evaluator, analyst_gk, inputLen);
        sums.push_back(tmp1);
    }
    // the same three steps as one device call
    for (Ciphertext record : processed) {
        sealhelper::fc_row(record, enc_weight_row, csp_rk, analyst_gk, (size_t)inputLen, tmp1);
        sums.push_back(tmp1);
    }
}

// the other virtuals of the interface (pasta_3_seal.h:20-30) and SEALZpCipher's statics
void client_and_analyst_side(shared_ptr<SEALContext> context, PublicKey pk, SecretKey sk, RelinKeys rk, GaloisKeys gk, vector<uint64_t> ssk,
                             vector<uint64_t> &sym_ct)
{
    auto ctx2 = pasta::SEALZpCipher::create_context(16384, 65537, 128);
    (void)ctx2;
    pasta::PASTA_SEAL c(context, pk, sk, rk, gk);
    pasta::SEALZpCipher &base = c;
    string name = base.get_cipher_name();
    c.activate_bsgs(false);
    c.add_gk_indices();
    c.encrypt_key(true);
    vector<Ciphertext> enc = c.encrypt_key_2(ssk, true);
    vector<Ciphertext> he = c.HE_decrypt(sym_ct, true);
    vector<uint64_t> back = c.decrypt_result(he, true);
    (void)name; (void)back; (void)enc;
    size_t a = c.get_key_size() + c.get_plain_size() + c.get_cipher_size();
    (void)a;
}
