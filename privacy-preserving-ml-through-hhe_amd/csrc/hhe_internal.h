// hhe_internal.h -- host-side context of libhhe_gfx950.so (not part of the C ABI).
#pragma once
#include <map>
#include <mutex>
#include <string>
#include <vector>
#include "hhe_common.h"
#include "hhe_launch.h"

struct BlockTables {   // public per-(nonce, block index) data of one PASTA block, device resident
    u64 *diag = nullptr;  // [4][128][L][N] lifted + NTT'd diagonals (multiply_plain operands)
    u64 *rc = nullptr;    // [4][N] round-constant plaintexts (coefficients mod t)
    u64 *pdiag = nullptr; // diag composed with the NTT-domain index map of rotate_rows(-1): pdiag[x] = diag[pi(x)]
    u64 *bsgs = nullptr;  // [4][128][L][N] babystep-giantstep variant of diag (lazy)
    size_t bytes = 0;     // device footprint of this entry
    u64 last_call = 0;    // the transciphering call that used it last (block_call): entries of the running call are never evicted
};

constexpr int HHE_MAX_STREAMS = 4;
constexpr int HHE_RELIN_SLOTS = 4;
struct hhe_ctx;
// One seal::RelinKeys / seal::GaloisKeys object with its identity: the reference's CSP holds several made by the same key
// generator with different randomness (analyst_he_gk with all default elements, csp_he_gk with the flatten steps, two RelinKeys;
// Analyst.cpp:62-94) and names the one it uses at every call (CSP.cpp:238-242, 271-278, 306, 312-316); which keys a rotation
// finds decides both its NAF decomposition and the ciphertext words.  Tables derived from a key live with the set (gk_corr) or
// are keyed by the key's device address (hhe_ctx::d_key_shoup), so two sets never share one.
struct hhe_keyset {
    hhe_ctx *ctx = nullptr;
    u64 *rk = nullptr;                 // RelinKeys::key(2): [L][2][K][N]
    std::map<u32, u64 *> gk;           // by Galois element: [L][2][K][N] each
    std::map<u32, u64 *> gk_corr;      // per Galois key of THIS set: shared-digit correction [2][K][N] (KsCorrArgs), built on first FC use
};
struct Lane {   // one stream + the per-batch workspaces of the ops (capacity `cap` ciphertexts)
    rt_stream stream = nullptr;
    void *ev_done = nullptr;
    bool own_stream = false;
    size_t cap = 0;
    u64 *ws_T = nullptr;     // [B][L][K][N]
    u64 *ws_S = nullptr;     // [B][2][K][N]
    u64 *ws_d = nullptr;     // [B][L][N]
    u64 *ws_ct[4] = {nullptr, nullptr, nullptr, nullptr};  // [B][2][L][N] each
    u64 *ws_ct3 = nullptr;   // [B][3][L][N]
    u64 *ws_plain = nullptr; // [B][N]
    u64 *ws_vals = nullptr;  // [B][128]
    u64 *bz_aq = nullptr, *bz_bq = nullptr;  // [B][2][L][N]
    u64 *bz_ab = nullptr, *bz_bb = nullptr;  // [B][2][L+1][N]
    u64 *bz_dq = nullptr;    // [B][3][L][N]
    u64 *bz_db = nullptr;    // [B][3][L+1][N]
    const u64 **d_ptrs = nullptr;  // [2*cap] per-item public-table pointers (diag | rc)
    size_t ptr_cap = 0;
    u64 *ws_rot = nullptr;   // [B][16][2][L][N] babystep rotations (allocated on first BSGS use)
    size_t rot_cap = 0;
    // FC shared digits: one slot per trie node that is still needed -- the digit transforms of its un-rotated c1 (tp [B][L][K][N]) and its
    // ciphertext (ct [B][2][L][N]); refs = 1 while the depth-first walk is below the node + 1 per queued leaf key switch that reads it
    struct FcSlot { u64 *tp = nullptr, *ct = nullptr, *c0hat = nullptr; int refs = 0; int tp_polys = 0; };  // c0hat [B][L][N]: NTT form of the node's c0 (nodes with a non-leaf child)  // tp_polys: K, or 1 when tp holds the special-prime transforms only
    std::vector<FcSlot> fc_slots;
    std::vector<u64 *> csum_bufs;  // FC leaves: integer sums of parents' c1 per Galois element, [B][L][N] each (sized like the slots)
    size_t fc_slot_cap = 0;  // items the slots were sized for
    u64 *ws_leaf = nullptr;  // FC leaf groups: special-limb sums [B][2][G][N] | their inverse transforms [B][2][G][N], G = HHE_LEAF_GROUP
    size_t leaf_cap = 0;
    u32 *zero_flag = nullptr; // device flag of the chunk this lane is evaluating (shared-digit FC)
    // host staging of small per-call inputs (mask values, pointer tables): it outlives the asynchronous copy, and the next
    // user waits for that copy (ev_stage) before overwriting it
    std::vector<u64> h_stage;
    std::vector<const u64 *> h_ptrs;
    void *ev_stage = nullptr;
    bool stage_pending = false;
    std::vector<std::pair<void *, void *>> prof_ev;  // hhe_ctx_profile: event pairs around the launches of ks_row_kernel on this lane's stream
    size_t prof_used = 0;
};

struct hhe_ctx {
    // every C-ABI entry point that takes the context holds this lock for its whole duration: concurrent callers (the
    // reference's gRPC handlers run concurrently, CSPRPC.cpp:201-203) are serialised per context; different contexts are
    // independent.  Recursive because hhe_decompose / the FC call other entry points.
    std::recursive_mutex mu;
    int profile = 0;               // hhe_ctx_profile: bracket the ks_row_kernel launches with timed events
    u64 prof_items = 0;            // ciphertexts covered by the bracketed launches since the last read
    int logn = 0, K = 0, L = 0, device = 0;
    size_t n = 0;
    u64 t = 0;
    std::vector<u64> q;            // K coefficient primes
    std::vector<u64> bsk;          // L+1: B_0..B_{L-1}, m_sk
    u64 gamma = 0;
    std::vector<u64> roots;        // psi per coefficient prime
    int nmod = 0;                  // K + (L+1) + 1
    int mod_t = 0;                 // index of the plain modulus
    int digit_reduce = 1;          // 0 when every data prime is below 4x every key prime (lazy NTT input range)
    std::vector<int> pm_ok;        // per ModDev index: the modulus has the pseudo-Mersenne form the lazy butterflies fold with (ModDev::pm_ok)
    u64 fc_fallbacks = 0;          // how often the shared-digit path had to be recomputed exactly
    int fc_shared = 1;             // FC rotation trie: children of a node share the digit transforms of its c1 (HHE_FC_SHARED; 2 = force the fallback, tests)
    int fc_leaf_group = HHE_LEAF_GROUP;  // FC rotation trie: leaf key switches per launch, across the nodes whose digits are resident (HHE_FC_LEAFGROUP; 1 = one leaf at a time)
    int fc_row_fused = 1;          // FC non-leaf children at N >= 4096: inner product + inverse row pass in one kernel (ks_perm_row_kernel; HHE_FC_ROWFUSED=0: separate launches)
    u64 fc_csum_closes = 0;        // how many c1 sums were closed (digits + transforms + one inner product); diagnostics, hhe_ctx_query("fc_csum_closes")
    int fc_csum_group = HHE_CSUM_GROUP;  // parents per csum_add launch (HHE_FC_CSUMGROUP, 1..HHE_CSUM_GROUP)
    int fc_c0hat = 1;              // FC non-leaf children through ks_perm_row_kernel: galois(c0) enters in the NTT domain (KsRowArgs::c0hat) instead of a gather in the KSF epilogue (HHE_FC_C0HAT=0)
    int fc_csum = 1;               // FC leaves: data-limb sums through per-element integer sums of the parents' c1 (one inner product per element instead of one per leaf; HHE_FC_CSUM=0: per leaf)
    u64 *d_qsp_poly = nullptr;     // [L][N]: the constant q_sp mod q_j in every slot (multiplier of the NTT form of a node's c0, FcSlot::c0hat)
    u64 *d_zero_corr = nullptr;    // [2][K][N] zeros: the correction table of the closing product of a c1 sum (its digits are already those of the rotated sum)
    int fc_leaf_sums = 1;          // FC rotation trie: postpone the inverse transforms of leaf key switches (linear part summed first)
    size_t fc_chunk = 160;         // items per internal chunk of hhe_fc_row (0 = whole batch); ms per MNIST sample (784x10, 16 samples): 64: 60.5, 80: 60.6, 96: 60.6, 128: 59.5, 160: 58.4
                                   // (round 1: 40: 67.1, 80: 64.9, 160: 64.0); the trie's small launches (2 polynomials per item) want more than one round of workgroups
    int matmul_mode = 1;           // 1: fused 20-transform pipeline (default), 0: op-by-op schedule
    KsConsts ksc{};

    // device tables
    ModDev *d_mods = nullptr;
    u64 *d_tables = nullptr;
    BehzDev *d_behz = nullptr;
    u32 *d_slot_map = nullptr;
    std::vector<u32> slot_map;

    // argument templates
    KsFinishArgs ksf{};
    AddPlainArgs apl{};

    // keys: the default set (hhe_set_galois_key, relin slot 0) and the further relin slots are key sets owned by the context;
    // `sets` are the ones callers created (hhe_keyset_create).  gks / rks = the sets the running entry point named (KeyScope).
    hhe_keyset keys0;
    hhe_keyset rk_slots[HHE_RELIN_SLOTS];      // [0] unused (slot 0 is keys0.rk)
    std::vector<hhe_keyset *> sets;
    hhe_keyset *gks = &keys0, *rks = &keys0;
    std::map<const u64 *, u64 *> d_key_shoup;  // per key-switch key (by device address): Shoup quotients of its words (fused row kernel), built on first use
    hhe_keyset *relin_set(int slot) { return slot == 0 ? &keys0 : &rk_slots[slot]; }

    // grow-only device scratch of hhe_decompose (all blocks of the records) and hhe_fc_row (per-chunk flags)
    u64 *d_blocks = nullptr;
    size_t blocks_cap = 0;   // words
    u32 *d_flags = nullptr;
    size_t flags_cap = 0;    // entries

    // PASTA public tables
    std::map<u64, BlockTables> blocks;
    size_t block_bytes = 0;                      // footprint of all cached block tables
    size_t block_cache_limit = (size_t)32 << 30; // bytes (hhe_pasta3_set_block_cache_limit, HHE_BLOCK_CACHE_MB): beyond it the least recently used counters go
    u64 block_call = 0;                          // running number of the transciphering calls
    u64 *d_feistel_mask = nullptr;  // [L][N] NTT form of the sbox_feistel mask plaintext

    // execution lanes: lane 0 runs on the caller's stream (generic ops); lanes 1.. are internal streams that
    // the transciphering path uses to process chunks of a batch concurrently (each with its own workspace)
    Lane lanes[1 + HHE_MAX_STREAMS];
    Lane *w = &lanes[0];
    int nstreams = 1;      // internal streams used by hhe_pasta3_transcipher (0 = caller's stream only).  One: since the row kernel lost its
                           // exposed round trips two overlapping chunks give the same throughput (289 vs 289 /s) and make every kernel's duration depend on its neighbour
    size_t chunk = 128;    // items per chunk (HHE_CHUNK); measured 211 /s at 32, 221 at 64, 225 at 128 items (round 1)
    void *ev_fork = nullptr;

    size_t ct_words() const { return (size_t)2 * L * n; }
    size_t ksk_words() const { return (size_t)L * 2 * K * n; }
};

// number theory (hhe_context.cpp)
bool nt_is_prime(u64 v);
bool nt_get_primes(u64 factor, int bits, size_t count, std::vector<u64> &out);
u64 nt_minimal_primitive_root(u64 degree, u64 q);
u64 nt_invmod(u64 a, u64 m);
u64 nt_mulmod(u64 a, u64 b, u64 m);
u64 nt_powmod(u64 a, u64 e, u64 m);
std::vector<int> nt_naf(int value);
u32 galois_elt_from_step(const hhe_ctx *c, int step);

// PASTA-3 public randomness (hhe_pasta_public.cpp)
void pasta3_block_randomness(u64 t, u64 nonce, u64 block, u64 *mats, u64 *rcs);

// 1 when every modulus of an NTT launch over ModDev indices [mod_base, mod_base + mod_cycle) has the pseudo-Mersenne form
// q = 2^b - c, 33 <= b <= 60 (ModDev::pm_ok; SEAL's own coefficient primes do, the 61-bit BEHZ base and t = 65537 do not): the
// kernels then use the truncated Shoup product and fold the butterfly ranges once per register round (NttArgs::lazy8)
inline int ntt_lazy8(const hhe_ctx *c, int mod_base, int mod_cycle)
{
    for (int i = mod_base; i < mod_base + mod_cycle; ++i)
        if (!c->pm_ok[i]) return 0;
    return 1;
}

struct DevBuf {  // scoped device allocation (freed on every exit path unless released)
    void *p = nullptr;
    explicit DevBuf(size_t bytes) { p = rt_malloc(bytes ? bytes : 8); }
    ~DevBuf() { if (p) rt_free(p); }
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    u64 *w() const { return (u64 *)p; }
    u64 *release() { u64 *r = (u64 *)p; p = nullptr; return r; }
};

struct CtxLock {  // null-safe scoped lock of a context (entry points check their arguments after taking it)
    std::unique_lock<std::recursive_mutex> l;
    explicit CtxLock(const hhe_ctx *c) { if (c) l = std::unique_lock<std::recursive_mutex>(const_cast<hhe_ctx *>(c)->mu); }
};
#define HHE_LOCK(c) CtxLock hhe_lock_guard_(c)

// names the key objects of one entry point for the ops below it (null = the context's default set); restored on exit
struct KeyScope {
    hhe_ctx *c;
    hhe_keyset *g0, *r0;
    KeyScope(hhe_ctx *c_, const hhe_keyset *gk, const hhe_keyset *rk) : c(c_), g0(c_->gks), r0(c_->rks)
    {
        c->gks = gk ? const_cast<hhe_keyset *>(gk) : &c->keys0;
        c->rks = rk ? const_cast<hhe_keyset *>(rk) : &c->keys0;
    }
    ~KeyScope() { c->gks = g0; c->rks = r0; }
};
// upload into a set (hhe_context.cpp); words are validated against the key-level primes first
int keyset_put_galois(hhe_keyset *ks, u32 elt, const u64 *ksk);
int keyset_put_relin(hhe_keyset *ks, const u64 *ksk);
void keyset_clear(hhe_keyset *ks);

void hhe_set_error(const std::string &msg);
int lane_reserve(hhe_ctx *c, Lane &ln, size_t B);
void sync_ctx(hhe_ctx *c);  // waits for every stream of the context
