// hhe_common.h -- shared types for the gfx950 BFV/PASTA-3 kernels and their host driver.
#pragma once
#include <cstddef>
#include <cstdint>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define HD __host__ __device__ __forceinline__
#else
#define HD inline
#endif

typedef uint64_t u64;
typedef uint32_t u32;

constexpr int HHE_MAXL = 32;          // data-level RNS limbs supported (the reference's N=65536 chain has 29 primes)
constexpr int HHE_MAXK = HHE_MAXL + 1;
constexpr int PASTA_T = 128;          // pasta_3_plain.h:18,32
constexpr int PASTA_R = 3;            // pasta_3_plain.h:33
constexpr u64 PASTA_NONCE = 123456789ULL;  // pasta_3_seal.cpp:47,115

// One RNS modulus as the kernels see it.  Tables live in device memory.
struct ModDev {
    u64 q;
    u64 r_lo, r_hi;        // floor(2^128 / q)
    u64 ninv, ninv_s;      // N^-1 mod q and its Shoup quotient
    u64 ninv_t, ninv_t_s;  // N^-1 * t mod q (INTT fused with the BEHZ "times t")
    const u64 *w, *ws;     // psi^bitrev(k), Shoup quotients [N] each (forward rounds: two 8-byte loads keep the 128-VGPR budget spill-free)
    const u64 *fw;         // [N][2]: the forward powers interleaved with their quotients as well (the row kernel's 8-points-per-lane rounds have the
                           // registers for one 16-byte load per twiddle; half the vector-memory instructions of w / ws)
    const u64 *iw;         // [N][2]: inverse powers interleaved with their Shoup quotients (inverse rounds: one 16-byte load, measured 3 % faster)
    u64 nq;                // 2^64 - q.  Read from the table, so the compiler cannot rewrite "+ h * nq" back into "- h * q": the lazy
                           // product becomes one multiply-add chain and conditional subtractions become add + sign select (no borrow chains)
    // Pseudo-Mersenne form q = 2^b - c (every prime SEAL generates descends from a power of two, seal/util/numth.h:138-139):
    // x mod q = (x mod 2^b) + (x >> b) * c  (mod q), one shift, one mask and one v_mad_u64_u32 -- no compare, no VCC.
    // pm_ok = 1 when 33 <= b <= 60, c < 2^32 and 2^b + 2^(64-b) c <= 2q, i.e. the fold takes ANY 64-bit value below 2q.
    u32 pm_sh;             // b - 32: the quotient x >> b is (high word) >> pm_sh
    u32 pm_mask;           // (1 << (b - 32)) - 1: mask of the high word
    u32 pm_c;              // c
    u32 pm_ok;
};

// The modulus table is written once at context creation.  Kernels read it through the CONSTANT address space: a plain global
// read of a workgroup-uniform entry cannot be proven unclobbered (the kernels store to global memory), so it compiles to a
// vector load that is repeated after every barrier and sits in front of every twiddle address as a second dependent memory
// round trip; from the constant address space it is an s_load into SGPRs (no VGPRs, scalar cache).
HD ModDev mod_at(const ModDev *mods, int i)
{
#if defined(__HIP_DEVICE_COMPILE__)
    const __attribute__((address_space(4))) ModDev *p = (const __attribute__((address_space(4))) ModDev *)(mods + i);
    ModDev m;
    m.q = p->q; m.r_lo = p->r_lo; m.r_hi = p->r_hi;
    m.ninv = p->ninv; m.ninv_s = p->ninv_s; m.ninv_t = p->ninv_t; m.ninv_t_s = p->ninv_t_s;
    m.w = p->w; m.ws = p->ws; m.fw = p->fw; m.iw = p->iw; m.nq = p->nq;
    m.pm_sh = p->pm_sh; m.pm_mask = p->pm_mask; m.pm_c = p->pm_c; m.pm_ok = p->pm_ok;
    return m;
#else
    return mods[i];
#endif
}

// the same for an index that is uniform over the workgroup (every transform kernel: one modulus per tile): stating the
// uniformity keeps the entry in SGPRs even where the compiler's divergence analysis gives up (inside the strided lane loops)
HD ModDev mod_at_u(const ModDev *mods, int i)
{
#if defined(__HIP_DEVICE_COMPILE__)
    i = __builtin_amdgcn_readfirstlane(i);
    const u64 p = (u64)mods;  // the table pointer too: the compiler sometimes keeps this kernel argument in VGPRs
    mods = (const ModDev *)(((u64)(unsigned)__builtin_amdgcn_readfirstlane((int)(p >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)p));
#endif
    return mod_at(mods, i);
}

// Modulus indices inside ModDev[]: 0..K-1 coefficient primes (K-1 = special),
// K..K+L Bsk = {B_0..B_{L-1}, m_sk}, K+L+1 = plain modulus t.

enum NttLoadOp { LOAD_PLAIN = 0, LOAD_DIGIT = 1, LOAD_LIFT = 2, LOAD_RNEG = 3 };
enum NttStoreOp {
    STORE_PLAIN = 0, STORE_MUL = 1, STORE_SCALE_T = 2, STORE_MAC = 3,
    STORE_RSP = 5,         // inv (special limb): v + floor(q_sp/2) mod q_sp
    STORE_KS1 = 6,         // inv: (v - r_1 + half) * q_sp^-1, written through the Galois map into aux_out
    STORE_KS0 = 7,         // fwd: NTT-domain key-switch finish of c0 + permuted-frame diagonal MAC
    STORE_KSF = 10,        // inv, polys [B][2][L]: generic key-switch finish (v - r_k + half) * q_sp^-1 (+ base poly k) into aux_out
    STORE_LAZY = 8         // fwd: leave the result in the lazy range [0,4q) (consumer reduces: key-switch inner product)
};

// key-switch mod-down constants, passed by value to the kernels that finish a key switch
struct KsConsts {
    u64 half;                 // floor(q_sp/2)
    u64 half_mod[HHE_MAXL];   // half mod q_j
    u64 qsp_inv[HHE_MAXL];    // q_sp^-1 mod q_j
    u64 qsp_inv_s[HHE_MAXL];
    u64 qsp_mod[HHE_MAXL];    // q_sp mod q_j (FC leaf sums: galois(c0) enters the sum that is later multiplied by q_sp^-1)
    u64 qsp_mod_s[HHE_MAXL];
    // KS1 epilogue in one expression: (v N^-1 - r + half) q_sp^-1 = v (N^-1 q_sp^-1) - r q_sp^-1 + half q_sp^-1  (mod q_j)
    u64 ninv_qinv[HHE_MAXL], ninv_qinv_s[HHE_MAXL];  // N^-1 q_sp^-1 mod q_j and its Shoup quotient
    u64 hq2[HHE_MAXL];                               // 2 q_j + (half q_sp^-1 mod q_j): keeps the lazy difference non-negative
};

struct NttArgs {
    const u64 *src;
    u64 *dst;
    const ModDev *mods;
    int logn;
    int logm;       // log2 of this pass's sub-transform size
    int logc;       // log2 lanes (independent sub-transforms) per tile
    int tiles_log;  // log2 tiles per polynomial in this pass (1-D grid decode)
    int count;      // polynomials in the batch
    int mod_base, mod_cycle;  // modulus of poly p = mod_base + p % mod_cycle
    // source poly of p: src + (p / src_item_polys) * src_item_stride + ((p % src_item_polys) / src_div) * N
    int src_div;          // DIGIT: K, LIFT: L, else 1
    int src_item_polys;   // polys of one batch item in p-space (0 => whole batch is one item)
    size_t src_item_stride;  // words between items in src
    int load_op, store_op;
    u32 load_einv;     // first pass: > 0 => the source is read through a Galois map (apply_galois as a gather): elt^-1 mod 2N;
                       // negation is modulo the SOURCE limb's prime (the digit source d_I lives mod q_I whatever the transform's modulus)
    u32 *zero_flag;    // LOAD_DIGIT: set to 1 when a coefficient equals 0 (the shared-digit FC path then falls back)
    int lazy8;         // every modulus of the launch has the pseudo-Mersenne form (ModDev::pm_ok, q < 2^60): the butterflies use the truncated
                       // Shoup product (results in [0,4q)) and fold with pm_fold once per register round; values stay below 16q <= 2^64
    int digit_reduce;  // DIGIT: 1 if some q_I >= 4*q_J (else the lazy butterflies absorb the unreduced residue)
    u64 t;          // LIFT: plain modulus
    // STORE_MUL / STORE_MAC multiplier (NTT form): (mul_ptrs ? mul_ptrs[p / mul_item_polys] : mul)
    //   + mul_shift + (p % mul_cycle) * N
    const u64 *mul;
    const u64 *const *mul_ptrs;
    size_t mul_shift;
    size_t mul_s_off;   // KS0: > 0 => the Shoup quotients of the multiplier table follow it at this word offset (pdiag of the fused matmul)
    int mul_cycle, mul_item_polys;
    u64 *acc;       // STORE_MAC / DIGIT_DIAG / KS0: accumulator polys
    // fused key-switch epilogues (matmul pipeline)
    int L, K;
    u32 gal_elt;        // KS1: coefficient-domain Galois element (0 = identity); KS0: NTT-domain element
    u32 gal_einv;       // KSF: > 0 => the base is galois(c0) gathered on the fly: elt^-1 mod 2N (aux_in = the un-rotated ciphertexts)
    const u64 *aux_r;   // KS1 / KSF: r [B][2][N] (KS1: poly 1);  KS0: S [B][2][K][N] (poly 0, limb j)
    const u64 *aux_in;  // KS0: c0 (NTT form) of the current state [B][L][N];  KSF: base ciphertexts (item b at aux_in + b * base_stride) or null
    u64 *aux_out;       // KS1: d [B][L][N];  KS0: c0 (NTT form) of the next state [B][L][N];  KSF: out [B][2][L][N]
    size_t base_stride; // KSF: words between the items of aux_in
    int base_mask;      // KSF: bit k set => add base poly k
    KsConsts ks;
};

enum EltOp { ELT_ADD = 0, ELT_SUB = 1, ELT_NEG = 2, ELT_MUL = 3, ELT_MAC = 4, ELT_COPY = 5, ELT_BCAST = 6, ELT_SHOUP = 7 };

struct EltArgs {  // element-wise kernels over [count][N] polys, modulus = mod_base + p % mod_cycle
    const u64 *a, *b;
    u64 *out;
    const ModDev *mods;
    int logn, count, mod_base, mod_cycle;
    int b_cycle;  // poly of b = p % b_cycle (broadcast over the batch), 0 => same index
};

struct CopyItemsArgs {  // dst item (s * dst_stride + dst_off) <- src item (s * src_stride + src_off), `words` words each, s < count
    const u64 *src;
    u64 *dst;
    size_t words, count;
    size_t src_stride, src_off, dst_stride, dst_off;  // in items
};

struct GaloisArgs {  // out[p][k] = +-in[p][k * einv mod 2N]; poly p = (item b, limb j), p = b * L + j
    const u64 *in;
    u64 *out;
    const ModDev *mods;
    int logn, count, L;
    size_t in_item_stride, out_item_stride;  // words between items
    u32 einv;  // elt^-1 mod 2N
    int accumulate;  // 1: out += gathered value (mod q) instead of out = value
};

struct KsMacArgs {  // S[b][k][J][n] = sum_I T[b][I][J][n] * key[I][k][J][n]
    const u64 *T;    // [B][L][K][N]
    const u64 *key;  // [L][2][K][N]
    u64 *S;          // [B][2][K][N]
    const ModDev *mods;
    int logn, B, L, K;
    // optional (FC leaf sums): for J < L the products are added into s_acc[b][k][J][n] instead of being written to S
    u64 *s_acc;
    // optional (fused matmul): acc[b][J][n] += T[b][J][J][n] * mul_ptrs[b][mul_shift + J*N + n] for J < L
    u64 *acc;
    const u64 *const *mul_ptrs;
    size_t mul_shift;
    // optional (FC shared digits): T holds the digit transforms of the UN-rotated c1; the rotation by perm_elt is applied
    // as the NTT-domain index map while reading T, and corr[k][J][n] (KsCorrArgs) is added to the sums
    u32 perm_elt;
    const u64 *corr;  // [2][K][N]
    // optional split output (key switch finished by the fused STORE_KSF pass): the data limbs go to S as [B][2][L][N] and the
    // special limb to S_sp [B][2][N] instead of S [B][2][K][N]
    u64 *S_sp;
};
// Fused key-switch row kernel (ks_row_kernel): for one (item b, key limb J, row tile) it runs the forward ROW pass of the
// L digit transforms T[b][I][J] (their strided pass has already run), multiplies each finished tile with key[I][0..1][J]
// (Shoup products, 64-bit lazy sums kept in registers), and runs the inverse ROW pass on the sums that are
// inverse-transformed next -- so neither T nor S_1 / S_k[special] makes a round trip through memory.
struct KsRowArgs {
    const u64 *key;      // [L][2][K][N]
    const u64 *key_s;    // Shoup quotients floor(key * 2^64 / q_J), same layout
    u64 *S;              // [B][2][K][N]: only S_0[j], j < L, is written (NTT form, canonical) -- the c0 branch reads it; null with U0
    u64 *U0;             // generic key switch: inverse row pass of S_0[j] (item b, limb j at U0 + b * u_stride + j * N); else null
    u64 *U1;             // inverse row pass of S_1[j], j < L (item b, limb j at U1 + b * u_stride + j * N)
    size_t u_stride;     // words between items in U0 / U1
    u64 *Usp;            // [B][2][N]: inverse row pass of S_0[special], S_1[special]
    int B, L, K;
    // fused matmul: acc[b][J][n] += T[b][J][J][n] * mul_ptrs[b][mul_shift + J*N + n] for J < L (the I = J digit)
    u64 *acc;
    const u64 *const *mul_ptrs;
    size_t mul_shift;
    size_t mul_s_off;    // > 0: Shoup quotients of the multiplier table at this word offset behind it
    // FC shared digits (ks_perm_row_kernel; U0 / U1 / Usp as in the generic key switch): T [B][L][K][N] = the COMPLETE digit transforms of the
    // node's un-rotated c1, read through the NTT-domain Galois map of perm_elt; corr [2][K][N] is added to the sums (KsCorrArgs)
    const u64 *T;
    const u64 *corr;
    u32 perm_elt;
    const u64 *c0hat;    // optional [B][L][N]: q_sp * NTT(the node's c0).  Read through the same map it joins S_0[j], so the mod-down
                         // returns galois(c0) + the key-switched part and the KSF epilogue needs no coefficient-domain gather of c0
};

// Correction of the shared-digit key switch (DESIGN.md "FC rotation trie"): the digit d_I of galois_g(c1) differs from
// galois_g applied mod q_J to the digit of c1 by q_I at every sign-flipped, non-zero coefficient, so
//   corr[k][J] = NTT_J(s_g) * sum_{I != J} (q_I mod q_J) * key_g[I][k][J]     (s_g = 0/1 polynomial of the flipped positions)
struct KsCorrArgs {
    const u64 *key;    // [L][2][K][N]
    const u64 *shat;   // [K][N] NTT_J(s_g)
    const u64 *qmod;   // [L][K]  q_I mod q_J
    u64 *corr;         // [2][K][N]
    const ModDev *mods;
    int logn, L, K;
};

struct PermArgs {  // NTT-domain Galois permutation: out[p][x] (op)= in[p][pi_elt(x)] (* mul)
    const u64 *in;
    u64 *out;
    const ModDev *mods;
    int logn, count, L;        // poly p = (item b, limb j)
    size_t out_item_stride;    // words between items in out (in is [count][N])
    u32 elt;
    int mac;                   // 1: out[p][x] += in[p][pi(x)] * mul_ptrs[b][shift + j*N + x]
    const u64 *const *mul_ptrs;
    size_t mul_shift;
};

struct LeafSumArgs {  // out[b][k][j] += qsp_inv_j * (accS[b][k][j] (INTT'd) + accH[b][k][j])
    const u64 *accS;  // [B][2][L][N] coefficient form (after INTT)
    const u64 *accH;  // [B][2][L][N] rounding terms, and for k = 0 q_sp * galois(c0) of the leaf parents
    u64 *out;         // [B][2][L][N]
    const ModDev *mods;
    int logn, B, L;
    KsConsts ks;
};

constexpr int HHE_LEAF_GROUP = 4;  // leaf key switches of the FC trie handled by one launch (leaves of ANY nodes whose digit transforms are still resident);
                                   // ms per MNIST sample at 1 / 2 / 4 / 8 per launch: 53.3 / 50.6 / 50.3 / 51.6
struct LeafRoundArgs {  // accH[b][k][j] += sum over the m leaves of: half_j - (r_l[b][k] mod q_j) (+ q_sp * galois_l(c0_b of leaf l's parent)[j] for k = 0)
    const u64 *r;      // [B][2][m][N]: INTT(S_k[special]) + floor(q_sp/2) mod q_sp of leaf l (the STORE_RSP result)
    u64 *accH;         // [B][2][L][N]
    const u64 *base[HHE_LEAF_GROUP];  // un-rotated parent ciphertexts of leaf l (item b at base[l] + b * base_stride, c0 limbs first); read through the Galois maps
    size_t base_stride;
    const ModDev *mods;
    int logn, B, L, m;
    u32 gal_einv[HHE_LEAF_GROUP];  // elt_l^-1 mod 2N
    KsConsts ks;
};
// Key-switch inner products of m leaf key switches from the shared digit transforms of their parents (KsMacArgs with perm_elt / corr,
// once per leaf): the data-limb sums of all m leaves go into s_acc with ONE read-modify-write, the special-limb sums of leaf l into
// S_sp[b][k][l].  Leaves of one parent read the same digit transforms back to back (cached).
struct KsMacLeavesArgs {
    u64 *S_sp;         // [B][2][m][N]
    u64 *s_acc;        // [B][2][L][N]
    const ModDev *mods;
    int logn, B, L, K, m;
    const u64 *T[HHE_LEAF_GROUP];     // [B][L][K][N] digit transforms of the un-rotated c1 of leaf l's parent ([B][L][N] when t_polys[l] == 1: special limb only)
    int t_polys[HHE_LEAF_GROUP];      // polynomials per digit in T[l]: K, or 1 for a parent whose children are all leaves
    int sp_only;                      // 1: only the special-limb sums (gid over [B][N/2]); the data-limb sums of the leaves come from the per-element c1 sums (CsumArgs)
    const u64 *key[HHE_LEAF_GROUP];   // [L][2][K][N]
    const u64 *corr[HHE_LEAF_GROUP];  // [2][K][N]
    u32 perm_elt[HHE_LEAF_GROUP];
};

// FC leaves, data limbs.  The inner product of a key switch is linear in the DIGITS as integers, and every leaf with the same Galois
// element g uses the same key: sum_l sum_I NTT_J(d_I(galois_g(c1_l))) key_g[I][k][J] = sum_I NTT_J(D_I) key_g[I][k][J] with D_I = the integer
// sum of the leaves' digits.  A digit of galois_g(c1) is c1_I at the mapped position, or q_I - c1_I where the map flips the sign (c1_I != 0:
// the shared-digit zero flag covers the rest), so D_I = galois_g applied to the integer sum of the UN-rotated limbs with `count` q_I - . at
// the flipped positions.  csum_add: sums[b][I][.] += c1 limbs of up to HHE_CSUM_GROUP parents (64-bit words + a byte counting the wraps);
// csum_digits: out[b][I][J][.] = D_I mod q_J for the K key-level primes -- the digit transforms and ONE inner product per element follow.
// The leaves' q_sp * galois_g(c0) terms are linear mod q_j outright: sums0 collects the un-rotated c0 limbs of the same parents (mod q_j),
// and csum_c0 adds q_sp * galois_g(sums0) into accH once per close -- the per-leaf coefficient-domain gather of c0 (8 bytes per cache
// line) leaves the rounding kernel.
constexpr int HHE_CSUM_GROUP = 8;  // parents per csum_add launch (the sums are read and written once per launch)
struct CsumArgs {
    u64 *sums;          // [B][L][N] integer sums of un-rotated c1 limbs, low 64 bits
    unsigned char *carry;  // [B][L][N] how often a sum wrapped 2^64 (terms below 2^61: 255 wraps cover > 2000 leaves per element)
    u64 *sums0;         // [B][L][N] sums of un-rotated c0 limbs mod q_j (null: not collected)
    const u64 *src[HHE_CSUM_GROUP];  // csum_add: parents' ciphertexts (item b at src[l] + b * src_stride: c0 limbs [L][N], then c1 limbs [L][N])
    size_t src_stride;
    u64 *accH;          // csum_c0: [B][2][L][N]
    u64 qsp_mod[HHE_MAXL], qsp_mod_s[HHE_MAXL];  // csum_c0: q_sp mod q_j and its Shoup quotient
    int m;              // csum_add: parents in this launch
    u64 *out;           // csum_digits: [B][L][K][N]
    const ModDev *mods;
    int logn, B, L, K;
    u32 einv;           // csum_digits: elt^-1 mod 2N
    u32 count;          // csum_digits: leaves summed
};

struct KsFinishArgs {  // SURVEY A.4 mod-down; S already INTT'd (coefficient form)
    const u64 *S;     // [B][2][K][N]
    const u64 *base;  // item b at base + b * base_item_stride, polys [2][L][N]; or null
    size_t base_item_stride;
    u64 *out;         // [B][2][L][N]
    const ModDev *mods;
    int logn, B, L, K;
    int base_mask;    // bit k set => add base poly k
    u64 half;                 // floor(q_sp/2)
    u64 half_mod[HHE_MAXL];   // half mod q_j
    u64 qsp_inv[HHE_MAXL];    // q_sp^-1 mod q_j
    u64 qsp_inv_s[HHE_MAXL];
    u64 qsp_mod[HHE_MAXL];    // q_sp mod q_j (FC leaf sums: galois(c0) enters the sum that is later multiplied by q_sp^-1)
    u64 qsp_mod_s[HHE_MAXL];
    // KS1 epilogue in one expression: (v N^-1 - r + half) q_sp^-1 = v (N^-1 q_sp^-1) - r q_sp^-1 + half q_sp^-1  (mod q_j)
    u64 ninv_qinv[HHE_MAXL], ninv_qinv_s[HHE_MAXL];  // N^-1 q_sp^-1 mod q_j and its Shoup quotient
    u64 hq2[HHE_MAXL];                               // 2 q_j + (half q_sp^-1 mod q_j): keeps the lazy difference non-negative
};

struct AddPlainArgs {
    const u64 *ct;     // [B][2][L][N]
    const u64 *plain;  // [B or 1][N] coefficients mod t
    const u64 *const *plain_ptrs;  // if set: item b reads plain_ptrs[b] + plain_shift
    size_t plain_shift;
    u64 *out;
    const ModDev *mods;
    int logn, B, L;
    int plain_bcast;   // 1: one plain for the whole batch
    int negate_ct;     // 1: out = -ct (+/-) plain
    int subtract;      // 1: sub_plain
    u64 t, q_mod_t, thr;
    u64 t_r_lo, t_r_hi;        // floor(2^128/t)
    u64 delta[HHE_MAXL];
};

struct EncodeArgs {  // slot scatter; INTT mod t follows as an NTT launch
    const u64 *vals;   // [B][stride]
    u64 *out;          // [B][N]
    const u32 *slot_map;
    int logn, B, stride, count;  // count values per item placed at slots [0,count)
    int second_off;    // >=0: values [count, 2*count) go to slots [second_off, second_off+count)
    u64 t;
};

struct DiagArgs {  // PASTA_SEAL::diagonal preparation (pasta_3_seal.cpp:390-401) into slot order
    const u64 *mats;   // [4][2][128][128]
    u64 *out;          // [4*128][N]  (pre-INTT slot image)
    const u32 *slot_map;
    int logn;
};

struct BsgsDiagArgs {  // PASTA_SEAL::babystep_giantstep diagonal preparation (pasta_3_seal.cpp:280-328) into slot order
    const u64 *mats;   // [4][2][128][128]
    u64 *out;          // [4*128][N]
    const u32 *slot_map;
    int logn, n1;      // n1 = BSGS_N1 (16)
};

struct BehzDev {  // SURVEY A.7 constants
    int L;
    u64 inv_punct_q[HHE_MAXL], inv_punct_q_s[HHE_MAXL];
    u64 mt_mod_q[HHE_MAXL];                      // 2^32 mod q_i
    u64 punct_q_bsk[HHE_MAXL][HHE_MAXL + 1];
    u64 punct_q_mt[HHE_MAXL];
    u64 neg_inv_q_mt;
    u64 q_mod_bsk[HHE_MAXL + 1], inv_mt_bsk[HHE_MAXL + 1], inv_q_bsk[HHE_MAXL + 1];
    u64 inv_punct_B[HHE_MAXL];
    u64 punct_B_q[HHE_MAXL][HHE_MAXL];
    u64 punct_B_msk[HHE_MAXL];
    u64 inv_B_msk, msk;
    u64 B_mod_q[HHE_MAXL], neg_B_mod_q[HHE_MAXL];
};

struct BehzExtendArgs {  // x [P][L][N] coeff -> xb [P][L+1][N] coeff (then NTT'd)
    const u64 *x;
    u64 *xb;
    const ModDev *mods;
    const BehzDev *bz;
    int logn, P, L, K;
};

struct TensorArgs {  // d[b][0..2][j] from a[b][0..1][j], b[b][0..1][j]; NTT domain
    const u64 *a, *b;
    u64 *d;
    const ModDev *mods;
    int logn, B, limbs, mod_base;
};

struct BehzFloorArgs {  // (dq [P][L][N], db [P][L+1][N]) coeff, already *t -> out [P][L][N]
    const u64 *dq, *db;
    u64 *out;
    const ModDev *mods;
    const BehzDev *bz;
    int logn, P, L, K;
};
