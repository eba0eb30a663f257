// hhe_launch.h -- launch + device-runtime interface between the host driver and the kernels.
// Implemented by hhe_kernels.hip (gfx950, the product) and, for CPU-side unit tests of
// the index arithmetic only, by tests/emu/hhe_launch_emu.cpp.
#pragma once
#include "hhe_common.h"
#include "hhe_client_bodies.h"

typedef void *rt_stream;

// device runtime
const char *rt_backend_name();
int rt_set_device(int device);
void *rt_malloc(size_t bytes);
void rt_free(void *p);
int rt_h2d(void *dst, const void *src, size_t bytes, rt_stream s);
int rt_d2h(void *dst, const void *src, size_t bytes, rt_stream s);
int rt_d2d(void *dst, const void *src, size_t bytes, rt_stream s);
int rt_memset(void *dst, int v, size_t bytes, rt_stream s);
int rt_sync(rt_stream s);
rt_stream rt_stream_create();
void rt_stream_destroy(rt_stream s);
void *rt_event_create();
void rt_event_destroy(void *ev);
void *rt_event_create_timed();                       // an event that hipEventElapsedTime accepts (profiling API only)
float rt_event_elapsed_ms(void *ev0, void *ev1);     // both recorded and complete; < 0 on failure
int rt_event_record(void *ev, rt_stream s);
int rt_event_sync(void *ev);                         // host waits for the event
int rt_stream_wait_event(rt_stream s, void *ev);
const char *rt_last_error();

// kernels (all asynchronous on `s`)
void k_ntt(const NttArgs &a, bool inverse, rt_stream s);  // runs both passes; a.logm/logc ignored
// forward transforms of two independent batches of the same degree in shared grids (the second one rides in the tail of the first)
void k_ntt2_fwd(const NttArgs &a1, const NttArgs &a2, rt_stream s);
// inverse transforms of two batches; the store epilogue of the second may read the results of the first (row passes share a grid)
void k_ntt2_inv(const NttArgs &a1, const NttArgs &a2, rt_stream s);
// one pass of a transform: second = false -> first pass (load ops), true -> second pass (store ops)
void k_ntt_pass(const NttArgs &a, bool inverse, bool second, rt_stream s);
void k_ntt2_fwd_first(const NttArgs &a1, const NttArgs &a2, rt_stream s);  // first forward pass of two batches in one grid
// fused key-switch row kernel (full 4096-point tiles only: N >= 4096); a = the digit transforms' NttArgs (dst = T);
// c0_row (optional) = a forward transform whose second (row) pass runs in the same grid
// supported: the row pass is 64, 128 or 256 points (N = 2^12 .. 2^16) -- the caller must also make sure that every key-level
// modulus has the pseudo-Mersenne form (ntt_lazy8(c, 0, K)): the kernel has no other arithmetic.  k_ks_row returns 0, or -1
// (nothing launched, rt_last_error() set) for an unsupported geometry.
inline bool k_ks_row_supported(int logn) { const int n2 = logn - logn / 2; return logn >= 12 && n2 >= 6 && n2 <= 8; }
int k_ks_row(const NttArgs &a, const KsRowArgs &x, const NttArgs *c0_row, rt_stream s);
// FC shared digits: key inner product over the complete digit transforms x.T read through the Galois map of x.perm_elt (+ x.corr) and the
// inverse row pass of all 2K sums into x.U0 / x.U1 / x.Usp; `a` carries the geometry and moduli (logn, mods, lazy8 = 1).  -1: unsupported size
int k_ks_perm_row(const NttArgs &a, const KsRowArgs &x, rt_stream s);
void k_elt(const EltArgs &a, int op, rt_stream s);
void k_copy_items(const CopyItemsArgs &a, rt_stream s);
void k_galois(const GaloisArgs &a, rt_stream s);
void k_perm(const PermArgs &a, rt_stream s);
void k_ks_mac(const KsMacArgs &a, rt_stream s);
void k_ks_corr(const KsCorrArgs &a, rt_stream s);
void k_ks_finish(const KsFinishArgs &a, rt_stream s);
void k_leaf_sum(const LeafSumArgs &a, rt_stream s);
void k_leaf_round(const LeafRoundArgs &a, rt_stream s);
void k_csum_add(const CsumArgs &a, rt_stream s);      // FC leaves: integer sums of un-rotated c1 limbs per Galois element
void k_csum_c0(const CsumArgs &a, rt_stream s);       // ... accH += q_sp * galois(sum of the parents' c0)
void k_csum_digits(const CsumArgs &a, rt_stream s);   // ... and the digits of galois(sum) mod every key-level prime
int k_ks_mac_leaves(const KsMacLeavesArgs &a, rt_stream s);  // -1: L > 4 (the caller takes the per-leaf path)
void k_add_plain(const AddPlainArgs &a, rt_stream s);
void k_encode_scatter(const EncodeArgs &a, rt_stream s);
void k_diag(const DiagArgs &a, rt_stream s);
void k_bsgs_diag(const BsgsDiagArgs &a, rt_stream s);
void k_behz_extend(const BehzExtendArgs &a, rt_stream s);
void k_tensor(const TensorArgs &a, rt_stream s);
void k_behz_floor(const BehzFloorArgs &a, rt_stream s);
// plain PASTA-3 (client side): XOF field elements, keystream blocks, record encryption / decryption
void k_pasta_xof(const PastaXofArgs &a, rt_stream s);
void k_pasta_plain(const PastaPlainArgs &a, rt_stream s);
void k_pasta_crypt(const PastaCryptArgs &a, rt_stream s);
void k_decrypt_round(const DecryptArgs &a, rt_stream s);
void k_decode_gather(const DecodeArgs &a, rt_stream s);

// split of logn into the two pass sizes (strided pass n1, row pass n2)
inline void ntt_split(int logn, int &n1, int &n2) { n1 = logn / 2; n2 = logn - n1; }
