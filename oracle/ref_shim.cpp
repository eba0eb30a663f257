// ref_shim.cpp -- extern "C" driver around the REFERENCE's own plain PASTA-3
// classes (pasta::PASTA / pasta::Pasta, /root/reference/src/pasta/pasta_3_plain.{h,cpp}),
// compiled from the reference sources where they lie by `make -C oracle ref`.
// Test infrastructure only: used to pin the oracle's PASTA restatement and to
// generate tests/golden/pasta_plain.json (tests/golden/make_pasta_golden.py).
#include <cstdint>
#include <cstring>
#include <vector>
#include "pasta_3_plain.h"

extern "C" {

// mats [4][2][128][128], rcs [4][2][128]: draw order of PASTA_SEAL::decomposition
// (pasta_3_seal.cpp:131-133,145-147): get_random_matrix x2 then get_rc_vec.
void ref_pasta_block_randomness(uint64_t t, uint64_t nonce, uint64_t block, uint64_t *mats, uint64_t *rcs)
{
    pasta::Pasta p(t);
    p.init_shake(nonce, block);
    for (int r = 0; r < 4; r++) {
        for (int h = 0; h < 2; h++) {
            auto m = p.get_random_matrix();
            for (int i = 0; i < 128; i++)
                std::memcpy(mats + ((size_t)(r * 2 + h) * 128 + i) * 128, m[i].data(), 128 * 8);
        }
        auto rc = p.get_rc_vec(128);
        std::memcpy(rcs + (size_t)r * 256, rc.data(), 256 * 8);
    }
}

void ref_pasta_keystream(uint64_t t, const uint64_t *key256, uint64_t nonce, uint64_t block, uint64_t *ks)
{
    std::vector<uint64_t> key(key256, key256 + 256);
    pasta::Pasta p(key, t);
    auto b = p.keystream(nonce, block);
    std::memcpy(ks, b.data(), 128 * 8);
}

void ref_pasta_encrypt(uint64_t t, const uint64_t *key256, const uint64_t *pt, size_t n, uint64_t *ct)
{
    pasta::PASTA c(std::vector<uint64_t>(key256, key256 + 256), t);
    auto out = c.encrypt(std::vector<uint64_t>(pt, pt + n));
    std::memcpy(ct, out.data(), n * 8);
}

void ref_pasta_decrypt(uint64_t t, const uint64_t *key256, const uint64_t *ct, size_t n, uint64_t *pt)
{
    pasta::PASTA c(std::vector<uint64_t>(key256, key256 + 256), t);
    auto out = c.decrypt(std::vector<uint64_t>(ct, ct + n));
    std::memcpy(pt, out.data(), n * 8);
}
}
