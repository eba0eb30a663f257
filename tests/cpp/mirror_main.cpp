// mirror_main.cpp -- drives include/pasta_seal_gfx950.hpp the way CSP.cpp:238-251 drives the reference
// classes: build PASTA_SEAL from context + keys, call decomposition on one record, flatten the blocks.
// Input/output are raw uint64 blobs written/read by tests/test_cpp_mirror.py.
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "pasta_seal_gfx950.hpp"

static std::vector<uint64_t> read_words(FILE *f, size_t n)
{
    std::vector<uint64_t> v(n);
    if (fread(v.data(), 8, n, f) != n) { fprintf(stderr, "short read\n"); exit(2); }
    return v;
}
int main(int argc, char **argv)
{
    if (argc < 3) return 2;
    FILE *f = fopen(argv[1], "rb");
    if (!f) return 2;
    auto hdr = read_words(f, 6);  // logn, K, t, n_galois, n_words, do_flatten
    const int logn = (int)hdr[0], K = (int)hdr[1];
    const size_t n = (size_t)1 << logn, L = K - 1, ksk = L * 2 * K * n, ctw = 2 * L * n;
    auto q = read_words(f, K);
    pasta::RelinKeys rk{read_words(f, ksk)};
    pasta::GaloisKeys gk;
    for (uint64_t i = 0; i < hdr[3]; i++) {
        uint32_t elt = (uint32_t)read_words(f, 1)[0];
        gk.keys[elt] = read_words(f, ksk);
    }
    pasta::Ciphertext enc_key{read_words(f, ctw), 2};
    auto record = read_words(f, hdr[4]);
    pasta::SecretKey he_sk{read_words(f, (size_t)K * n)};
    auto sym_key = read_words(f, 256);
    auto plain_record = read_words(f, hdr[4]);
    // second key objects of the same secret key (the CSP's own: csp_he_gk for flatten, csp rk for the FC; Analyst.cpp:70-94) + a weight row
    auto hdr2 = read_words(f, 1);
    pasta::GaloisKeys csp_gk;
    for (uint64_t i = 0; i < hdr2[0]; i++) {
        uint32_t elt = (uint32_t)read_words(f, 1)[0];
        csp_gk.keys[elt] = read_words(f, ksk);
    }
    pasta::RelinKeys csp_rk{read_words(f, ksk)};
    pasta::Ciphertext w_row{read_words(f, ctw), 2};
    fclose(f);
    try {
        auto ctx = std::make_shared<pasta::HheContext>(logn, q, hdr[2], 0);
        pasta::PASTA_SEAL HHE(ctx, pasta::PublicKey{}, pasta::SecretKey{}, rk, gk);
        HHE.add_gk_indices();
        std::vector<pasta::Ciphertext> blocks = HHE.decomposition(record, {enc_key}, true);
        FILE *o = fopen(argv[2], "wb");
        uint64_t nb = blocks.size();
        fwrite(&nb, 8, 1, o);
        for (auto &b : blocks) fwrite(b.words.data(), 8, b.words.size(), o);
        if (hdr[5]) {
            pasta::Ciphertext flat;
            HHE.flatten(blocks, flat);
            fwrite(flat.words.data(), 8, flat.words.size(), o);
        }
        // packed ops on the first two blocks (SEAL_Cipher.cpp:547-566)
        pasta::Ciphertext sq, prod, sum;
        HHE.packed_square(sq, blocks[0]);
        HHE.packed_enc_mul(blocks[0], blocks[1], prod);
        HHE.packed_enc_add(blocks[0], blocks[1], sum);
        fwrite(sq.words.data(), 8, sq.words.size(), o);
        fwrite(prod.words.data(), 8, prod.words.size(), o);
        fwrite(sum.words.data(), 8, sum.words.size(), o);
        // client and analyst ends (User.cpp / Analyst.cpp): PASTA::encrypt, PASTA::decrypt, sealhelper::decrypting
        pasta::PASTA sym(ctx, sym_key, hdr[2]);
        auto sym_ct = sym.encrypt(plain_record);
        auto sym_back = sym.decrypt(sym_ct);
        fwrite(sym_ct.data(), 8, sym_ct.size(), o);
        fwrite(sym_back.data(), 8, sym_back.size(), o);
        if (hdr[5]) {
            pasta::Ciphertext flat;
            HHE.flatten(blocks, flat);
            auto dec = sealhelper::decrypting(flat, he_sk, *ctx, 256);
            fwrite(dec.data(), 8, dec.size(), o);
        }
        // the CSP's request loop: a FRESH cipher object per request built from the same key objects by value (CSP.cpp:238-242), flatten
        // with the GaloisKeys the call names (CSP.cpp:271-278), then the per-record loop as one batched call and one FC row
        pasta::Ciphertext f2;
        for (int req = 0; req < 3; req++) {
            pasta::PASTA_SEAL per_request(ctx, pasta::PublicKey{}, pasta::SecretKey{}, rk, gk);
            std::vector<pasta::Ciphertext> b2 = per_request.decomposition(record, {enc_key}, true);
            per_request.flatten(b2, f2, csp_gk);
        }
        fwrite(f2.words.data(), 8, f2.words.size(), o);
        std::vector<std::vector<uint64_t>> recs = {record, record};
        std::vector<pasta::Ciphertext> batch = HHE.decompose(recs, {enc_key}, csp_gk, true);
        fwrite(batch[0].words.data(), 8, batch[0].words.size(), o);
        fwrite(batch[1].words.data(), 8, batch[1].words.size(), o);
        pasta::Ciphertext fc;
        sealhelper::fc_row(*ctx, batch[0], w_row, csp_rk, csp_gk, 3, fc);
        fwrite(fc.words.data(), 8, fc.words.size(), o);
        printf("key objects uploaded: %llu, resident sets: %zu, encrypted-key uploads: %llu\n", (unsigned long long)ctx->keys().uploads(),
               ctx->keys().resident(), (unsigned long long)ctx->key_uploads);
        fclose(o);
        try { pasta::PASTA bad(ctx, std::vector<uint64_t>(255, 1), hdr[2]); printf("NO THROW\n"); return 3; }
        catch (const std::runtime_error &e) { printf("throws: %s\n", e.what()); }
        // error behaviour: a context without Galois keys must throw like SEAL does
        pasta::PASTA_SEAL bare(std::make_shared<pasta::HheContext>(logn, q, hdr[2], 0), {}, {}, rk, {});
        try { bare.decomposition(record, {enc_key}); printf("NO THROW\n"); return 3; }
        catch (const std::invalid_argument &e) { printf("throws: %s\n", e.what()); }
    } catch (const std::exception &e) {
        fprintf(stderr, "exception: %s\n", e.what());
        return 1;
    }
    printf("mirror ok (%s)\n", hhe_backend());
    return 0;
}
