// hhe_pasta_public.cpp -- PASTA-3 public per-block randomness on the host.
// Restates Pasta::init_shake / generate_random_field_element / get_random_matrix /
// calculate_row / get_rc_vec (src/pasta/pasta_3_plain.cpp:56-119, 286-295): a SHAKE128
// stream over BE64(nonce) || BE64(block), consumed as big-endian 64-bit words masked to
// bitlen(t) with rejection.  Tiny (ms per block) and identical for every sample, so it
// stays on the CPU; the kernels consume the matrices from HBM.
#include <array>
#include <cstring>
#include "hhe_internal.h"
#include "../../include/hhe_gfx950.h"

namespace {

class Shake128Stream {  // FIPS 202 sponge, rate 168 bytes, domain suffix 0x1F
public:
    explicit Shake128Stream(const uint8_t *msg, size_t len)
    {
        lanes_.fill(0);
        uint8_t block[kRate];
        while (len >= kRate) { absorb(msg); msg += kRate; len -= kRate; }
        memset(block, 0, sizeof(block));
        memcpy(block, msg, len);
        block[len] ^= 0x1f;
        block[kRate - 1] ^= 0x80;
        absorb(block);
        avail_ = 0;
    }
    u64 next_be64()
    {
        u64 v = 0;
        for (int i = 0; i < 8; ++i) v = (v << 8) | next_byte();
        return v;
    }

private:
    static constexpr size_t kRate = 168;
    std::array<u64, 25> lanes_;
    uint8_t out_[kRate];
    size_t avail_ = 0;

    static u64 rol(u64 x, unsigned s) { return s ? (x << s) | (x >> (64 - s)) : x; }
    void absorb(const uint8_t *blk)
    {
        for (size_t i = 0; i < kRate / 8; ++i) {
            u64 lane = 0;
            for (int b = 7; b >= 0; --b) lane = (lane << 8) | blk[8 * i + b];
            lanes_[i] ^= lane;
        }
        permute();
    }
    uint8_t next_byte()
    {
        if (avail_ == 0) {
            if (squeezed_once_) permute();
            squeezed_once_ = true;
            for (size_t i = 0; i < kRate; ++i) out_[i] = (uint8_t)(lanes_[i / 8] >> (8 * (i % 8)));
            avail_ = kRate;
        }
        return out_[kRate - avail_--];
    }
    bool squeezed_once_ = false;

    void permute()
    {  // Keccak-f[1600], x/y lane indexing A[x + 5y]
        static const unsigned rot[5][5] = {{0, 36, 3, 41, 18}, {1, 44, 10, 45, 2}, {62, 6, 43, 15, 61}, {28, 55, 25, 21, 56}, {27, 20, 39, 8, 14}};
        u64 rc = 1;  // round constants from the degree-8 LFSR
        auto lfsr_bit = [](uint8_t &r) { bool bit = r & 1; r = (uint8_t)((r & 0x80) ? (r << 1) ^ 0x71 : (r << 1)); return bit; };
        uint8_t lf = 1;
        (void)rc;
        for (int round = 0; round < 24; ++round) {
            u64 C[5], D[5], Bm[25];
            for (int x = 0; x < 5; ++x) C[x] = lanes_[x] ^ lanes_[x + 5] ^ lanes_[x + 10] ^ lanes_[x + 15] ^ lanes_[x + 20];
            for (int x = 0; x < 5; ++x) D[x] = C[(x + 4) % 5] ^ rol(C[(x + 1) % 5], 1);
            for (int i = 0; i < 25; ++i) lanes_[i] ^= D[i % 5];
            for (int x = 0; x < 5; ++x)
                for (int y = 0; y < 5; ++y) Bm[y + 5 * ((2 * x + 3 * y) % 5)] = rol(lanes_[x + 5 * y], rot[x][y]);
            for (int y = 0; y < 5; ++y)
                for (int x = 0; x < 5; ++x) lanes_[x + 5 * y] = Bm[x + 5 * y] ^ (~Bm[(x + 1) % 5 + 5 * y] & Bm[(x + 2) % 5 + 5 * y]);
            u64 iota = 0;
            for (int j = 0; j < 7; ++j)
                if (lfsr_bit(lf)) iota ^= (u64)1 << ((1u << j) - 1);
            lanes_[0] ^= iota;
        }
    }
};

struct FieldSampler {
    Shake128Stream xof;
    u64 p, mask;
    FieldSampler(u64 prime, u64 nonce, u64 block) : xof(seed(nonce, block).data(), 16), p(prime)
    {
        int bits = 0;
        for (u64 v = prime; v; v >>= 1) ++bits;
        mask = bits >= 64 ? ~(u64)0 : (((u64)1 << bits) - 1);
    }
    static std::array<uint8_t, 16> seed(u64 nonce, u64 block)
    {
        std::array<uint8_t, 16> s;
        for (int i = 0; i < 8; ++i) {
            s[i] = (uint8_t)(nonce >> (56 - 8 * i));
            s[8 + i] = (uint8_t)(block >> (56 - 8 * i));
        }
        return s;
    }
    u64 draw(bool allow_zero)
    {
        for (;;) {
            const u64 e = xof.next_be64() & mask;
            if (e >= p || (!allow_zero && e == 0)) continue;
            return e;
        }
    }
    // sequential invertible matrix: row_i = first_row * row_{i-1}[127] + shift(row_{i-1})
    void matrix(u64 *m)
    {
        for (int j = 0; j < PASTA_T; ++j) m[j] = draw(false);
        for (int i = 1; i < PASTA_T; ++i) {
            const u64 *prev = m + (size_t)(i - 1) * PASTA_T;
            u64 *row = m + (size_t)i * PASTA_T;
            const u64 last = prev[PASTA_T - 1];
            for (int j = 0; j < PASTA_T; ++j) {
                u64 v = nt_mulmod(m[j], last, p);
                if (j) v = (v + prev[j - 1]) % p;
                row[j] = v;
            }
        }
    }
};

}  // namespace

// draw order per affine layer (pasta_3_seal.cpp:131-133): matrix 1, matrix 2, rc (128 + 128)
void pasta3_block_randomness(u64 t, u64 nonce, u64 block, u64 *mats, u64 *rcs)
{
    FieldSampler fs(t, nonce, block);
    for (int layer = 0; layer <= PASTA_R; ++layer) {
        fs.matrix(mats + ((size_t)layer * 2 + 0) * PASTA_T * PASTA_T);
        fs.matrix(mats + ((size_t)layer * 2 + 1) * PASTA_T * PASTA_T);
        for (int i = 0; i < 2 * PASTA_T; ++i) rcs[(size_t)layer * 2 * PASTA_T + i] = fs.draw(true);
    }
}

extern "C" int hhe_pasta3_block_randomness(uint64_t t, uint64_t block_index, uint64_t *mats, uint64_t *rcs)
{
    if (!mats || !rcs || t < 2) { hhe_set_error("hhe_pasta3_block_randomness: invalid arguments"); return HHE_ERR_INVALID; }
    pasta3_block_randomness(t, PASTA_NONCE, block_index, mats, rcs);
    return HHE_OK;
}
