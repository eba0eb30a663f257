// hhe_keyset_cache.hpp -- host-side helpers shared by the two C++ adapters (pasta_seal_gfx950.hpp on word containers,
// pasta_seal_gfx950_seal.hpp on seal:: types) over the C ABI of libhhe_gfx950.so:
//   KeySetCache  maps every RelinKeys / GaloisKeys OBJECT the caller passes to one device key set (hhe_keyset), recognised by a
//                hash over ALL of its words.  The reference copies its key objects by value into every cipher object and every
//                call (SEAL_Cipher.cpp:9-36, CSP.cpp:238-242), so pointers say nothing; contents do.  An object is uploaded
//                once and stays resident with everything derived from it; the least recently used sets are dropped beyond
//                `max_sets` (a CSP that serves many analysts).
//   DeviceArena  grow-only device buffers reused across calls instead of a hipMalloc / hipFree pair per call.
#pragma once
#include <cstdint>
#include <list>
#include <map>
#include <mutex>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>
#include "hhe_gfx950.h"

namespace hhe {

// 128 bits over every word (two independent multiply-xorshift lanes) plus the length: a 4-word sample cannot tell two key
// objects of one key generator apart reliably, the whole content can
struct ContentHash {
    uint64_t a = 0x243F6A8885A308D3ULL, b = 0x13198A2E03707344ULL, n = 0;
    void add(const uint64_t *w, size_t count)
    {
        for (size_t i = 0; i < count; i++) {
            a = (a ^ w[i]) * 0x9E3779B97F4A7C15ULL; a ^= a >> 29;
            b = (b + w[i]) * 0xC2B2AE3D27D4EB4FULL; b ^= b >> 31;
        }
        n += count;
    }
    void add_tag(uint64_t t) { add(&t, 1); }
    bool operator<(const ContentHash &o) const { return a != o.a ? a < o.a : b != o.b ? b < o.b : n < o.n; }
};

class KeySetCache {
public:
    explicit KeySetCache(hhe_ctx *ctx, size_t max_sets = 16) : ctx_(ctx), max_sets_(max_sets) {}
    ~KeySetCache() { for (auto &e : lru_) hhe_keyset_destroy(e.second); }
    KeySetCache(const KeySetCache &) = delete;
    KeySetCache &operator=(const KeySetCache &) = delete;

    // keys: (Galois element, pointer to its [L][2][K][N] words); words = words per key
    hhe_keyset *galois(const std::vector<std::pair<uint32_t, const uint64_t *>> &keys, size_t words)
    {
        ContentHash h;
        h.add_tag(0x6b67);  // "gk"
        for (auto &kv : keys) { h.add_tag(kv.first); h.add(kv.second, words); }
        return lookup(h, [&](hhe_keyset *ks) {
            for (auto &kv : keys)
                if (int rc = hhe_keyset_set_galois(ks, kv.first, kv.second)) return rc;
            return (int)HHE_OK;
        });
    }
    hhe_keyset *relin(const uint64_t *key, size_t words)
    {
        ContentHash h;
        h.add_tag(0x726b);  // "rk"
        h.add(key, words);
        return lookup(h, [&](hhe_keyset *ks) { return hhe_keyset_set_relin(ks, key); });
    }
    size_t resident() const { return lru_.size(); }
    uint64_t uploads() const { return uploads_; }   // how many objects were sent to the device (a repeated object is not)

private:
    template <typename F> hhe_keyset *lookup(const ContentHash &h, F &&fill)
    {
        std::lock_guard<std::mutex> lk(mu_);
        auto it = index_.find(h);
        if (it != index_.end()) {
            lru_.splice(lru_.begin(), lru_, it->second);  // most recently used first
            return it->second->second;
        }
        hhe_keyset *ks = nullptr;
        if (hhe_keyset_create(ctx_, &ks) != HHE_OK) throw std::runtime_error(hhe_last_error());
        if (int rc = fill(ks)) {
            const std::string msg = hhe_last_error();
            hhe_keyset_destroy(ks);
            if (rc == HHE_ERR_INVALID) throw std::invalid_argument(msg);
            throw std::runtime_error(msg);
        }
        ++uploads_;
        lru_.emplace_front(h, ks);
        index_[h] = lru_.begin();
        while (lru_.size() > max_sets_) {
            hhe_keyset_destroy(lru_.back().second);
            index_.erase(lru_.back().first);
            lru_.pop_back();
        }
        return ks;
    }
    hhe_ctx *ctx_;
    size_t max_sets_;
    std::mutex mu_;
    std::list<std::pair<ContentHash, hhe_keyset *>> lru_;
    std::map<ContentHash, std::list<std::pair<ContentHash, hhe_keyset *>>::iterator> index_;
    uint64_t uploads_ = 0;
};

// A few grow-only device buffers ("slots") owned by the adapter's context object; a call takes the arena's lock for its duration
// (the library serialises calls on one context anyway) and gets buffers that survive the call.
class DeviceArena {
public:
    static constexpr int SLOTS = 6;
    ~DeviceArena() { for (auto &s : buf_) hhe_free(s.first); }
    std::mutex &mutex() { return mu_; }
    uint64_t *get(int slot, size_t bytes)  // caller holds mutex()
    {
        auto &s = buf_[slot];
        if (s.second < bytes) {
            hhe_free(s.first);
            s.first = hhe_malloc(bytes);
            s.second = s.first ? bytes : 0;
            if (!s.first) throw std::runtime_error("hhe_malloc failed");
        }
        return static_cast<uint64_t *>(s.first);
    }
private:
    std::mutex mu_;
    std::pair<void *, size_t> buf_[SLOTS] = {};
};

}  // namespace hhe
