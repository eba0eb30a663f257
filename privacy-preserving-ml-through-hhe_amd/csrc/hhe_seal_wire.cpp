// hhe_seal_wire.cpp -- SEAL 4.0 binary serialization at the boundary: Ciphertext / RelinKeys / GaloisKeys
// blobs as the reference moves them (src/examples/CSP/CSP.cpp:328-490 load of keys and ciphertexts, :495-547 spill file,
// :552-605 concatenated stream, protos/hhe.proto:21-24) are decoded straight into HBM, and result ciphertexts are encoded.
//
// What pins the layout: the SEALHeader struct (seal/serialization.h:60-93), DynArray<T>::save_members (seal/dynarray.h:652-680,
// inline), the member lists of KSwitchKeys::save_size (seal/kswitchkeys.h:161-178) and PublicKey = Ciphertext
// (seal/publickey.h:89-93) are in the reference's headers.  Ciphertext::save_members itself lives in the prebuilt libseal
// (never loaded here) and the reference holds no serialized object: the Ciphertext member order below is SEAL 4.0.0's as
// published (parms_id, is_ntt_form, size, poly_modulus_degree, coeff_modulus_size, scale, correction_factor, data); exactly that
// order is accepted (for BFV: the double 1.0, then the integer 1).
// PARITY UNPINNED against bytes produced by SEAL: the tests can only round-trip and check sizes.
// The blobs arrive over gRPC (untrusted): inflation is bounded by the largest legitimate object of the context, every entry point
// catches allocation failures, and key objects are decoded and validated completely before anything is uploaded.
// Compressed objects (SEAL's default is zstd, then zlib) are inflated with the system's libzstd.so.1 / libz.so.1, opened at
// run time; an object in a mode whose library is missing is rejected with an error, never guessed.
#include <dlfcn.h>
#include <zlib.h>  // z_stream layout only: the functions are looked up in libz.so.1 at run time
#include <algorithm>
#include <cstring>
#include <new>
#include <string>
#include <utility>
#include <vector>
#include "hhe_internal.h"
#include "../../include/hhe_gfx950.h"

namespace {

int wfail(int code, const std::string &msg) { hhe_set_error("SEAL stream: " + msg); return code; }

constexpr uint16_t SEAL_MAGIC = 0xA15E;  // seal/serialization.h:58
constexpr size_t HDR = 16;
enum { COMPR_NONE = 0, COMPR_ZLIB = 1, COMPR_ZSTD = 2 };  // seal/serialization.h compr_mode_type

struct Reader {
    const uint8_t *p;
    size_t n, pos = 0;
    Reader(const uint8_t *p_, size_t n_) : p(p_), n(n_) {}
    bool get(void *dst, size_t k)
    {
        if (k > n - pos) return false;
        memcpy(dst, p + pos, k);
        pos += k;
        return true;
    }
};

// ---- zlib (deflate with zlib wrapper, as SEAL's ztools writes it) through libz.so.1
int inflate_zlib(const uint8_t *in, size_t n, std::vector<uint8_t> &out, size_t max_out)
{
    void *h = dlopen("libz.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!h) return wfail(HHE_ERR_INVALID, "zlib-compressed object and libz.so.1 is not available");
    auto init = (int (*)(z_stream *, const char *, int))dlsym(h, "inflateInit_");
    auto run = (int (*)(z_stream *, int))dlsym(h, "inflate");
    auto end = (int (*)(z_stream *))dlsym(h, "inflateEnd");
    if (!init || !run || !end) return wfail(HHE_ERR_INVALID, "libz.so.1 lacks inflate");
    z_stream z;
    memset(&z, 0, sizeof(z));
    if (init(&z, ZLIB_VERSION, (int)sizeof(z_stream)) != Z_OK) return wfail(HHE_ERR_INVALID, "inflateInit failed");
    out.clear();
    std::vector<uint8_t> buf(1 << 20);
    size_t pos = 0;
    int rc = 0;
    do {
        const size_t chunk = std::min<size_t>(n - pos, 1u << 30);
        z.next_in = const_cast<Bytef *>(in + pos); z.avail_in = (unsigned)chunk;
        do {
            z.next_out = buf.data(); z.avail_out = (unsigned)buf.size();
            rc = run(&z, Z_NO_FLUSH);
            if (rc != Z_OK && rc != Z_STREAM_END) { end(&z); return wfail(HHE_ERR_INVALID, "zlib data is corrupt"); }
            if (out.size() + (buf.size() - z.avail_out) > max_out) { end(&z); return wfail(HHE_ERR_INVALID, "compressed object inflates beyond the largest object of this context"); }
            out.insert(out.end(), buf.data(), buf.data() + (buf.size() - z.avail_out));
        } while (z.avail_out == 0 && rc != Z_STREAM_END);
        pos += chunk - z.avail_in;
    } while (rc != Z_STREAM_END && pos < n);
    end(&z);
    if (rc != Z_STREAM_END) return wfail(HHE_ERR_INVALID, "zlib data is truncated");
    return HHE_OK;
}
// ---- Zstandard through libzstd.so.1 (streaming API; SEAL writes frames without a content size)
struct ZInBuf { const void *src; size_t size, pos; };
struct ZOutBuf { void *dst; size_t size, pos; };
int inflate_zstd(const uint8_t *in, size_t n, std::vector<uint8_t> &out, size_t max_out)
{
    void *h = dlopen("libzstd.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!h) return wfail(HHE_ERR_INVALID, "zstd-compressed object and libzstd.so.1 is not available");
    auto create = (void *(*)())dlsym(h, "ZSTD_createDStream");
    auto init = (size_t (*)(void *))dlsym(h, "ZSTD_initDStream");
    auto run = (size_t (*)(void *, ZOutBuf *, ZInBuf *))dlsym(h, "ZSTD_decompressStream");
    auto fre = (size_t (*)(void *))dlsym(h, "ZSTD_freeDStream");
    auto iserr = (unsigned (*)(size_t))dlsym(h, "ZSTD_isError");
    if (!create || !init || !run || !fre || !iserr) return wfail(HHE_ERR_INVALID, "libzstd.so.1 lacks the streaming decoder");
    void *ds = create();
    if (!ds || iserr(init(ds))) { if (ds) fre(ds); return wfail(HHE_ERR_INVALID, "ZSTD_initDStream failed"); }
    out.clear();
    std::vector<uint8_t> buf(1 << 20);
    ZInBuf ib{in, n, 0};
    size_t rc = 1;
    while (ib.pos < ib.size || rc != 0) {
        ZOutBuf ob{buf.data(), buf.size(), 0};
        const size_t before = ib.pos;
        rc = run(ds, &ob, &ib);
        if (iserr(rc)) { fre(ds); return wfail(HHE_ERR_INVALID, "zstd data is corrupt"); }
        if (out.size() + ob.pos > max_out) { fre(ds); return wfail(HHE_ERR_INVALID, "compressed object inflates beyond the largest object of this context"); }
        out.insert(out.end(), buf.data(), buf.data() + ob.pos);
        if (rc == 0 && ib.pos >= ib.size) break;             // frame complete, input consumed
        if (ob.pos == 0 && ib.pos == before) { fre(ds); return wfail(HHE_ERR_INVALID, "zstd data is truncated"); }
    }
    fre(ds);
    return HHE_OK;
}

// One serialized object: header at bytes[0..16), `size` bytes in all.  body/body_n = the members (inflated into `storage`
// when compressed, at most max_body bytes; max_body = 0: the object must not be compressed -- nested objects never are).
int open_object(const uint8_t *bytes, size_t nbytes, std::vector<uint8_t> &storage, const uint8_t *&body, size_t &body_n, size_t &consumed, size_t max_body)
{
    if (!bytes || nbytes < HDR) return wfail(HHE_ERR_INVALID, "shorter than a SEALHeader");
    uint16_t magic;
    memcpy(&magic, bytes, 2);
    const uint8_t header_size = bytes[2], vmajor = bytes[3], compr = bytes[5];
    uint64_t size;
    memcpy(&size, bytes + 8, 8);
    if (magic != SEAL_MAGIC || header_size != HDR) return wfail(HHE_ERR_INVALID, "bad magic / header size (not a SEAL >= 3.5 object)");
    if (vmajor != 3 && vmajor != 4) return wfail(HHE_ERR_INVALID, "unsupported SEAL major version " + std::to_string(vmajor));
    if (size < HDR || size > nbytes) return wfail(HHE_ERR_INVALID, "header size field exceeds the buffer (truncated object)");
    consumed = (size_t)size;
    if (compr == COMPR_NONE) { body = bytes + HDR; body_n = (size_t)size - HDR; return HHE_OK; }
    if (max_body == 0) return wfail(HHE_ERR_INVALID, "nested object is compressed");
    int rc;
    if (compr == COMPR_ZLIB) rc = inflate_zlib(bytes + HDR, (size_t)size - HDR, storage, max_body);
    else if (compr == COMPR_ZSTD) rc = inflate_zstd(bytes + HDR, (size_t)size - HDR, storage, max_body);
    else return wfail(HHE_ERR_INVALID, "unknown compression mode " + std::to_string(compr));
    if (rc) return rc;
    body = storage.data(); body_n = storage.size();
    return HHE_OK;
}

struct CtMembers {
    uint8_t parms_id[32];
    uint8_t is_ntt = 0;
    uint64_t size = 0, n = 0, cms = 0;
    const uint8_t *data = nullptr;  // size * cms * n words
    size_t words = 0;
};
// members of a Ciphertext at r (Ciphertext::save_members; the nested DynArray carries its own header, seal/dynarray.h:579)
int parse_ct_members(Reader &r, CtMembers &m)
{
    uint64_t f1, f2;
    if (!r.get(m.parms_id, 32) || !r.get(&m.is_ntt, 1) || !r.get(&m.size, 8) || !r.get(&m.n, 8) || !r.get(&m.cms, 8) ||
        !r.get(&f1, 8) || !r.get(&f2, 8))
        return wfail(HHE_ERR_INVALID, "truncated ciphertext members");
    // scale (double), then correction_factor (integer): SEAL 4.0.0's member order; BFV writes 1.0 and 1
    const uint64_t one_d = 0x3FF0000000000000ULL;
    if (f1 != one_d || f2 != 1) return wfail(HHE_ERR_INVALID, "scale / correction factor are not those of a BFV ciphertext (SEAL 4.0 member order)");
    if (m.size < 2 || m.size > 3 || m.n == 0 || m.cms == 0 || m.cms > HHE_MAXK) return wfail(HHE_ERR_INVALID, "ciphertext size fields out of range");
    // nested DynArray object, always uncompressed inside its parent
    if (r.n - r.pos < HDR) return wfail(HHE_ERR_INVALID, "truncated ciphertext data");
    std::vector<uint8_t> none;
    const uint8_t *body;
    size_t body_n, used;
    int rc = open_object(r.p + r.pos, r.n - r.pos, none, body, body_n, used, 0);  // rejects a compressed nested array before touching it
    if (rc) return rc;
    uint64_t count;
    if (body_n < 8) return wfail(HHE_ERR_INVALID, "truncated ciphertext data");
    memcpy(&count, body, 8);
    const uint64_t full = m.size * m.cms * m.n;
    if (count != full) {
        if (count < full) return wfail(HHE_ERR_INVALID, "seeded (symmetric-key) ciphertext serialization is not supported");
        return wfail(HHE_ERR_INVALID, "ciphertext data length does not match its size fields");
    }
    if (body_n < 8 + count * 8) return wfail(HHE_ERR_INVALID, "truncated ciphertext data");
    m.data = body + 8; m.words = (size_t)count;
    r.pos += used;
    return HHE_OK;
}

// members of one serialized ciphertext with `polys` polynomials of `limbs` limbs: parms_id, is_ntt_form, five 8-byte fields, the
// nested DynArray (header + count + words)
size_t ct_body_bytes(const hhe_ctx *c, size_t polys, size_t limbs) { return 32 + 1 + 8 * 5 + HDR + 8 + polys * limbs * c->n * 8; }
// largest KSwitchKeys body accepted: the index table plus HHE_WIRE_MAX_KEYS keys (a default GaloisKeys object holds 2 log2 N - 1)
constexpr size_t HHE_WIRE_MAX_KEYS = 64;
size_t ksk_body_bytes(const hhe_ctx *c) { return 32 + 8 + 8 * 2 * c->n + HHE_WIRE_MAX_KEYS * (size_t)c->L * (HDR + ct_body_bytes(c, 2, c->K)); }

// KSwitchKeys members (seal/kswitchkeys.h:161-178): parms_id, keys_dim1, then per entry keys_dim2 and that many PublicKey
// objects (= size-2 key-level NTT-form ciphertexts).  fn(index, words of [dim2][2][K][N]) is called for non-empty entries.
template <typename F> int parse_kswitch_keys(hhe_ctx *c, const uint8_t *bytes, size_t nbytes, size_t *consumed, F &&fn)
{
    std::vector<uint8_t> storage;
    const uint8_t *body;
    size_t body_n, used;
    int rc = open_object(bytes, nbytes, storage, body, body_n, used, ksk_body_bytes(c));
    if (rc) return rc;
    Reader r(body, body_n);
    uint8_t parms_id[32];
    uint64_t dim1;
    if (!r.get(parms_id, 32) || !r.get(&dim1, 8)) return wfail(HHE_ERR_INVALID, "truncated key members");
    if (dim1 > 2 * c->n) return wfail(HHE_ERR_INVALID, "key table larger than the Galois group");
    std::vector<u64> words;
    size_t nkeys = 0;
    for (uint64_t i = 0; i < dim1; ++i) {
        uint64_t dim2;
        if (!r.get(&dim2, 8)) return wfail(HHE_ERR_INVALID, "truncated key members");
        if (dim2 == 0) continue;
        if (++nkeys > HHE_WIRE_MAX_KEYS) return wfail(HHE_ERR_INVALID, "more than " + std::to_string(HHE_WIRE_MAX_KEYS) + " keys in one object");
        if (dim2 != (uint64_t)c->L) return wfail(HHE_ERR_INVALID, "key-switch key has " + std::to_string(dim2) + " digits, the context has " + std::to_string(c->L));
        words.clear();
        for (uint64_t d = 0; d < dim2; ++d) {
            std::vector<uint8_t> st2;
            const uint8_t *b2;
            size_t n2, u2;
            if ((rc = open_object(r.p + r.pos, r.n - r.pos, st2, b2, n2, u2, 0))) return rc;
            Reader rr(b2, n2);
            CtMembers m;
            if ((rc = parse_ct_members(rr, m))) return rc;
            if (m.size != 2 || m.n != c->n || m.cms != (uint64_t)c->K || !m.is_ntt)
                return wfail(HHE_ERR_INVALID, "key-switch key entry is not a size-2, key-level, NTT-form ciphertext of this context");
            const size_t at = words.size();
            words.resize(at + m.words);
            memcpy(words.data() + at, m.data, m.words * 8);   // the blob need not be 8-byte aligned
            for (size_t p = 0; p < 2 * (size_t)c->K; ++p)
                for (size_t j = 0; j < c->n; ++j)
                    if (words[at + p * c->n + j] >= c->q[p % c->K]) return wfail(HHE_ERR_INVALID, "key data is not reduced modulo the coefficient primes");
            r.pos += u2;
        }
        if ((rc = fn((size_t)i, words))) return rc;
    }
    if (consumed) *consumed = used;
    return HHE_OK;
}

// the loaders run on untrusted bytes: an allocation failure (or any other C++ exception) must not cross the C boundary
template <typename F> int guarded(F &&f)
{
    try { return f(); }
    catch (const std::bad_alloc &) { return wfail(HHE_ERR_CAPACITY, "out of host memory while decoding"); }
    catch (const std::exception &e) { return wfail(HHE_ERR_INVALID, e.what()); }
}

}  // namespace

// ====================================================================== C ABI
extern "C" int hhe_seal_load_ciphertext(hhe_ctx *c, const uint8_t *bytes, size_t nbytes, uint64_t *out_dptr, size_t out_cap_words,
                                        size_t *ct_size, uint8_t *parms_id_out, size_t *consumed)
{
    HHE_LOCK(c);
    if (!c || !bytes || !out_dptr) return wfail(HHE_ERR_INVALID, "null argument");
    return guarded([&]() -> int {
    std::vector<uint8_t> storage;
    const uint8_t *body;
    size_t body_n, used;
    int rc = open_object(bytes, nbytes, storage, body, body_n, used, ct_body_bytes(c, 3, c->L));
    if (rc) return rc;
    Reader r(body, body_n);
    CtMembers m;
    if ((rc = parse_ct_members(r, m))) return rc;
    if (m.n != c->n || m.cms != (uint64_t)c->L) return wfail(HHE_ERR_INVALID, "ciphertext is not at the data level of this context (encrypted is not valid for encryption parameters)");
    if (m.is_ntt) return wfail(HHE_ERR_INVALID, "BFV ciphertext in NTT form");
    if (m.words > out_cap_words) return wfail(HHE_ERR_CAPACITY, "output buffer too small");
    // the safe load's is_data_valid_for: every coefficient below its prime (a foreign blob must not reach the lazy kernels)
    for (size_t p = 0; p < m.size * m.cms; ++p) {
        const u64 qi = c->q[p % m.cms];
        for (size_t j = 0; j < c->n; ++j) {
            u64 v;
            memcpy(&v, m.data + (p * c->n + j) * 8, 8);
            if (v >= qi) return wfail(HHE_ERR_INVALID, "ciphertext data is not reduced modulo the coefficient primes");
        }
    }
    rt_stream st = c->lanes[0].stream;
    if (rt_h2d(out_dptr, m.data, m.words * 8, st) || rt_sync(st)) return wfail(HHE_ERR_DEVICE, rt_last_error());
    if (ct_size) *ct_size = (size_t)m.size;
    if (parms_id_out) memcpy(parms_id_out, m.parms_id, 32);
    if (consumed) *consumed = used;
    return HHE_OK;
    });
}

extern "C" int hhe_seal_save_ciphertext(hhe_ctx *c, const uint64_t *ct_dptr, size_t ct_size, const uint8_t *parms_id, uint8_t *out,
                                        size_t out_cap, size_t *written)
{
    HHE_LOCK(c);
    if (!c || !ct_dptr || !parms_id || !written || ct_size < 2 || ct_size > 3) return wfail(HHE_ERR_INVALID, "bad arguments");
    const size_t words = ct_size * c->L * c->n;
    const size_t inner = HDR + 8 + words * 8;                 // DynArray object
    const size_t total = HDR + 32 + 1 + 8 * 5 + inner;        // header + members
    *written = total;
    if (!out || out_cap < total) return wfail(HHE_ERR_CAPACITY, "output buffer too small (needed size returned)");
    auto header = [](uint8_t *p, uint64_t size) {
        memset(p, 0, HDR);
        const uint16_t magic = SEAL_MAGIC;
        memcpy(p, &magic, 2);
        p[2] = (uint8_t)HDR; p[3] = 4; p[4] = 0; p[5] = COMPR_NONE;
        memcpy(p + 8, &size, 8);
    };
    uint8_t *p = out;
    header(p, total); p += HDR;
    memcpy(p, parms_id, 32); p += 32;
    *p++ = 0;  // is_ntt_form: BFV ciphertexts are kept in coefficient form
    const uint64_t f[3] = {ct_size, c->n, (uint64_t)c->L};
    memcpy(p, f, 24); p += 24;
    const double scale = 1.0;
    const uint64_t corr = 1;
    memcpy(p, &scale, 8); p += 8;
    memcpy(p, &corr, 8); p += 8;
    header(p, inner); p += HDR;
    const uint64_t cnt = words;
    memcpy(p, &cnt, 8); p += 8;
    rt_stream st = c->lanes[0].stream;
    if (rt_d2h(p, ct_dptr, words * 8, st) || rt_sync(st)) return wfail(HHE_ERR_DEVICE, rt_last_error());
    return HHE_OK;
}

// every key of the object, decoded and validated on the host; nothing is uploaded here
static int decode_keys(hhe_ctx *c, const uint8_t *bytes, size_t nbytes, size_t *consumed, std::vector<std::pair<size_t, std::vector<u64>>> &keys)
{
    return parse_kswitch_keys(c, bytes, nbytes, consumed, [&](size_t index, const std::vector<u64> &w) {
        if (w.size() != c->ksk_words()) return wfail(HHE_ERR_INVALID, "key-switch key has the wrong size");
        keys.emplace_back(index, w);
        return (int)HHE_OK;
    });
}
static int load_relin(hhe_keyset *ks, const uint8_t *bytes, size_t nbytes, size_t *consumed)
{
    hhe_ctx *c = ks->ctx;
    std::vector<std::pair<size_t, std::vector<u64>>> keys;
    int rc = decode_keys(c, bytes, nbytes, consumed, keys);
    if (rc) return rc;
    if (keys.empty()) return wfail(HHE_ERR_INVALID, "RelinKeys object holds no key");
    if (keys.size() != 1 || keys[0].first != 0) return wfail(HHE_ERR_INVALID, "RelinKeys beyond key(2) are not used by the path");  // RelinKeys::get_index(2) = 0
    return keyset_put_relin(ks, keys[0].second.data());
}
static int load_galois(hhe_keyset *ks, const uint8_t *bytes, size_t nbytes, size_t *consumed, uint32_t *n_keys)
{
    hhe_ctx *c = ks->ctx;
    std::vector<std::pair<size_t, std::vector<u64>>> keys;
    int rc = decode_keys(c, bytes, nbytes, consumed, keys);
    if (n_keys) *n_keys = 0;
    if (rc) return rc;
    for (auto &kv : keys)   // GaloisKeys::get_index(elt) = (elt - 1) / 2 (seal/galoiskeys.h:48-74)
        if ((rc = keyset_put_galois(ks, (uint32_t)(2 * kv.first + 1), kv.second.data()))) return rc;  // only a device failure can stop this loop
    if (n_keys) *n_keys = (uint32_t)keys.size();
    return HHE_OK;
}
extern "C" int hhe_seal_load_relin_keys(hhe_ctx *c, int slot, const uint8_t *bytes, size_t nbytes, size_t *consumed)
{
    HHE_LOCK(c);
    if (!c || !bytes || slot < 0 || slot >= HHE_RELIN_SLOTS) return wfail(HHE_ERR_INVALID, "null argument / bad slot");
    return guarded([&] { return load_relin(c->relin_set(slot), bytes, nbytes, consumed); });
}
extern "C" int hhe_seal_load_relin_keys_ks(hhe_keyset *ks, const uint8_t *bytes, size_t nbytes, size_t *consumed)
{
    if (!ks || !bytes) return wfail(HHE_ERR_INVALID, "null argument");
    HHE_LOCK(ks->ctx);
    return guarded([&] { return load_relin(ks, bytes, nbytes, consumed); });
}
extern "C" int hhe_seal_load_galois_keys(hhe_ctx *c, const uint8_t *bytes, size_t nbytes, size_t *consumed, uint32_t *n_keys)
{
    HHE_LOCK(c);
    if (!c || !bytes) return wfail(HHE_ERR_INVALID, "null argument");
    return guarded([&] { return load_galois(&c->keys0, bytes, nbytes, consumed, n_keys); });
}
extern "C" int hhe_seal_load_galois_keys_ks(hhe_keyset *ks, const uint8_t *bytes, size_t nbytes, size_t *consumed, uint32_t *n_keys)
{
    if (!ks || !bytes) return wfail(HHE_ERR_INVALID, "null argument");
    HHE_LOCK(ks->ctx);
    return guarded([&] { return load_galois(ks, bytes, nbytes, consumed, n_keys); });
}
