// rsa.hip.h — RSA public-key operation + EMSA-PKCS1-v1_5 check, one signature per wavefront.
//
// Replaces rsa 0.9.6 (RsaPublicKey::verify with Pkcs1v15Sign::new::<Sha256>()) over
// num-bigint-dig 0.8.4, reached through cfdkim's verify_signature; call site
// core/src/email.rs:31-33.  RFC 8017 §8.2.2 / §9.2.
//
// Mapping.  A 2048-bit operand is exactly 64 x 32-bit limbs = ONE LIMB PER LANE of a
// wave64 (NL = 1); RSA-3072/4096 take two limbs per lane (NL = 2).  Montgomery
// multiplication runs limb-serial over b (v_readlane broadcast of b_i and of the
// quotient digit), limb-parallel over a and n: every lane does two v_mad_u64_u32 per
// step and keeps its column in a redundant (low word, small high word) form, so the only
// cross-lane traffic per step is one whole-wave shift of the low words (DPP wave_shl).
// Carries are resolved once per multiplication with the ballot carry-lookahead
// (generate/propagate masks added as 64-bit scalars).  Pure 32/64-bit integer VALU; no
// LDS, no MFMA.
//
// R^2 mod n is derived on the device per signature (no host precomputation):
// R mod n by subtraction (+ modular doublings when the modulus does not fill its
// container), then log2(bits) Montgomery squarings of 2R.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace zke {

struct RsaJob {                 // written by the parse kernel (or the host for the unit entry point)
  uint8_t  mod[512];            // modulus, big-endian, right-aligned in mod[512-k .. 512)
  uint8_t  sig[512];            // signature, big-endian, right-aligned in sig[512-k .. 512)
  uint64_t e;                   // public exponent
  uint32_t k;                   // modulus length in bytes (minimal)
  uint32_t flags;               // RSA_F_*
  uint32_t bits;                // modulus bit length
  uint32_t sig_len;             // decoded b= length in bytes
  uint32_t pad[2];
};
static_assert(sizeof(RsaJob) == 1056, "RsaJob layout");

enum : uint32_t {
  RSA_F_ACTIVE = 1,             // run the modexp
  RSA_F_SHA1 = 2,               // EMSA block carries the SHA-1 DigestInfo and a 20-byte hash (a=rsa-sha1)
  RSA_F_QUAD = 4,               // set by rsa_verify_kernel: rsa_group_kernel<4> (four lanes per signature) takes this job
  RSA_F_OCT = 8,                // ... rsa_group_kernel<8> (eight lanes per signature: moduli of 2049..4096 bits)
};

// Per-key Montgomery constants, cached across e-mails and batches (one table per engine = per device, shared by all
// submission slots).  Keyed by the MODULUS itself: the slot comes from a hash of its two low limbs and a hit needs all
// limbs equal, so a lookup is exact and needs nothing but the decoded key — the front end does it (parse.hip.h) and
// routes the signature to the lane-group kernel when the constants are there.  An entry is claimed once (state 0 -> 1 by
// atomicCAS), written, published (state 2) and never modified again, so readers need no lock.  A key whose slot is owned
// by another key is simply not cached.
struct KeyCacheEntry {
  uint32_t state;               // 0 empty, 1 being filled, 2 valid
  uint32_t ninv;                // -n^-1 mod 2^32
  uint32_t bits;                // modulus bit length
  uint32_t pad;
  uint32_t mod[128];            // the modulus, little-endian 32-bit limbs (zero above `bits`)
  uint32_t rr[128];             // R^2 mod n, limb q*64+lane
  uint32_t rr28[152];           // rsa_quad.hip.h: 2^4256 mod n as 76 limbs of 28 bits (512..2048 bits), 2^8512 mod n as 152 (..4096)
};
static_assert(sizeof(KeyCacheEntry) == 1648, "KeyCacheEntry layout");
constexpr uint32_t KEY_CACHE_SLOTS = 4096;
__host__ __device__ inline uint32_t key_cache_slot(uint32_t n0, uint32_t n1) { return (n0 * 0x9E3779B1u + n1 * 0x85EBCA77u) >> 20; }   // 12 bits
static_assert(KEY_CACHE_SLOTS == (1u << 12), "key_cache_slot yields 12 bits");

// agent-scope (sc1) accesses: served at the coherence point, not from this XCD's L2
__device__ __forceinline__ uint32_t ld_agent(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_agent(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// ---- cross-lane helpers -------------------------------------------------------------
// value of lane+1 (lane 63 gets 0): v_mov_b32_dpp wave_shl:1 bound_ctrl:0
__device__ __forceinline__ uint32_t lane_up(uint32_t x) {
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x130 /*wave_shl:1*/, 0xf, 0xf, false);
}
// value of lane-1 (lane 0 gets 0): wave_shr:1
__device__ __forceinline__ uint32_t lane_down(uint32_t x) {
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x138 /*wave_shr:1*/, 0xf, 0xf, false);
}
__device__ __forceinline__ uint64_t ballot64(bool p) { return __ballot(p); }

// A big number in "lane-major" layout: limb index = q*64 + lane for q in [0,NL).
template <int NL> struct Big { uint32_t v[NL]; };

// x >= y ?   (lexicographic from the top limb: the highest differing limb decides)
template <int NL>
__device__ __forceinline__ bool big_ge(const Big<NL>& x, const Big<NL>& y) {
#pragma unroll
  for (int q = NL - 1; q >= 0; q--) {
    uint64_t gt = ballot64(x.v[q] > y.v[q]), lt = ballot64(x.v[q] < y.v[q]);
    if (gt != lt) return gt > lt;
  }
  return true;
}

// r = x - y (mod 2^(2048*NL)); returns the borrow out.  Borrow-lookahead with ballots.
template <int NL>
__device__ __forceinline__ uint32_t big_sub(Big<NL>& r, const Big<NL>& x, const Big<NL>& y, int lane) {
  uint32_t bin = 0;
#pragma unroll
  for (int q = 0; q < NL; q++) {
    uint32_t d = x.v[q] - y.v[q];
    uint64_t g = ballot64(x.v[q] < y.v[q]);        // generates a borrow
    uint64_t p = ballot64(d == 0);                  // propagates an incoming borrow
    // borrow into lane j = bit j of ((g<<1 | bin) + p) ^ p
    uint64_t gs = (g << 1) | bin;
    uint64_t sum = gs + p;
    uint32_t carry_out = (uint32_t)((g >> 63) | ((sum < gs) ? 1u : 0u));
    uint64_t inc = sum ^ p;
    r.v[q] = d - (uint32_t)((inc >> lane) & 1);
    bin = carry_out;
  }
  return bin;
}

// Normalise columns (lo + hi*2^32 at limb position) into 32-bit limbs; returns the overflow
// word above the top limb.  hi is small (< 2^3).
template <int NL>
__device__ __forceinline__ uint32_t big_normalize(Big<NL>& r, const uint32_t (&lo)[NL], const uint32_t (&hi)[NL], int lane) {
  uint32_t cin = 0;      // carry word from the previous 64-limb group's top lane
  uint32_t cbit = 0;     // single-bit carry into lane 0 of this group
#pragma unroll
  for (int q = 0; q < NL; q++) {
    uint32_t up = lane_down(hi[q]);                 // hi of limb-1 lands on this limb
    if (lane == 0) up = cin;
    uint32_t x = lo[q] + up;
    uint64_t g = ballot64(x < up);
    uint64_t p = ballot64(x == 0xFFFFFFFFu);
    uint64_t gs = (g << 1) | cbit;
    uint64_t sum = gs + p;
    uint32_t cout = (uint32_t)((g >> 63) | ((sum < gs) ? 1u : 0u));
    uint64_t inc = sum ^ p;
    r.v[q] = x + (uint32_t)((inc >> lane) & 1);
    cin = __builtin_amdgcn_readlane(hi[q], 63);
    cbit = cout;
  }
  return cin + cbit;
}

// The 64 CIOS steps of a one-limb-per-lane Montgomery product, written out by hand (gfx950), vector unit only
// (the scalar unit is shared by the four SIMDs of a CU: one s_add per step is all it sees).
// Column state per lane: T = tl + th * 2^32 in a VGPR pair = the 64-bit addend of the first multiply.
//   P = a * b_i + T                 v_mad_u64_u32, carry-out c0 (65th bit; only when a = b_i = 2^32 - 1)
//   m = lo(P of lane 0) * ninv      v_mul_lo_u32 + v_readfirstlane
//   Q = n * m + P                   v_mad_u64_u32, carry-out c1
//   tl' = hi(Q) + lo(Q of lane+1)   one v_add_co_u32 with the DPP wave shift folded in, carry c2
//   th' = c2 + c0 + c1              three v_addc_co_u32
// 9 VALU instructions per step against 16 (seven of them register-pair moves) from the C++ below.
// Fixed temporaries v[64:71], s[60:67] (clobbers).  Wait states (gfx940 rules): one between a VALU write and a
// v_readfirstlane of the register, two before a DPP read.
__device__ __forceinline__ void mont_core_64(uint32_t& tl, uint32_t& th, uint32_t a, uint32_t b, uint32_t n, uint32_t ninv) {
  asm volatile(
      "v_mov_b32 v64, 0\n\t"
      "v_mov_b32 v65, 0\n\t"
      "v_mov_b32 v70, 0\n\t"
      "s_mov_b32 s60, 0\n\t"
      "v_readlane_b32 s61, %3, s60\n"              // b_0
      "1:\n\t"
      ".rept 4\n\t"
      "v_mad_u64_u32 v[66:67], s[64:65], %2, s61, v[64:65]\n\t"
      "v_mul_lo_u32 v71, v66, %5\n\t"
      "s_add_u32 s60, s60, 1\n\t"
      "v_readfirstlane_b32 s62, v71\n\t"
      "v_mad_u64_u32 v[68:69], s[66:67], %4, s62, v[66:67]\n\t"
      "v_readlane_b32 s61, %3, s60\n\t"           // next b limb (lane select 64 after the last step reads lane 0: unused)
      "v_addc_co_u32_e64 v65, s[62:63], v70, 0, s[64:65]\n\t"       // th' = c0
      "v_add_co_u32_dpp v64, vcc, v68, v69 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
      "v_addc_co_u32_e32 v65, vcc, 0, v65, vcc\n\t"                  // + c2
      "v_addc_co_u32_e64 v65, s[62:63], v65, 0, s[66:67]\n\t"       // + c1
      ".endr\n\t"
      "s_cmp_lt_u32 s60, 64\n\t"
      "s_cbranch_scc1 1b\n\t"
      "v_mov_b32 %0, v64\n\t"
      "v_mov_b32 %1, v65"
      : "=&v"(tl), "=&v"(th)
      : "v"(a), "v"(b), "v"(n), "s"(__builtin_amdgcn_readfirstlane(ninv))
      : "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "s60", "s61", "s62", "s63", "s64", "s65", "s66", "s67", "vcc", "scc");
}

// The two-limbs-per-lane counterpart (3072 / 4096-bit moduli: limb q*64 + lane, 128 steps).  Same step, two
// columns per lane; limb 64 (lane 0 of the upper group) hands its low word to limb 63 (lane 63 of the lower
// group) through v_readfirstlane + v_writelane.  19 VALU per step against ~35 from the C++ below.
// Operands: %4 %5 = a, %6 %7 = b, %8 %9 = n (lower, upper group), %10 = ninv.  Temporaries v[64:78], s[60:72].
#define ZKE_MONT128_STEP(BSRC)                                                                          \
      "v_mad_u64_u32 v[66:67], s[64:65], %4, s61, v[64:65]\n\t"                                         \
      "v_mad_u64_u32 v[74:75], s[68:69], %5, s61, v[72:73]\n\t"                                         \
      "v_mul_lo_u32 v71, v66, %10\n\t"                                                                  \
      "s_add_u32 s60, s60, 1\n\t"                                                                       \
      "v_readfirstlane_b32 s62, v71\n\t"                                                                \
      "v_mad_u64_u32 v[68:69], s[66:67], %8, s62, v[66:67]\n\t"                                         \
      "v_mad_u64_u32 v[76:77], s[70:71], %9, s62, v[74:75]\n\t"                                         \
      "s_and_b32 s63, s60, 63\n\t"                                                                      \
      "v_readlane_b32 s61, " BSRC ", s63\n\t"                                                           \
      "v_addc_co_u32_e64 v65, s[62:63], v70, 0, s[64:65]\n\t"                                           \
      "v_addc_co_u32_e64 v73, s[62:63], v70, 0, s[68:69]\n\t"                                           \
      "v_mov_b32_dpp v78, v68 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"                   \
      "v_readfirstlane_b32 s72, v76\n\t"                                                                \
      "v_writelane_b32 v78, s72, 63\n\t"                                                                \
      "v_add_co_u32_e32 v64, vcc, v78, v69\n\t"                                                         \
      "v_addc_co_u32_e32 v65, vcc, 0, v65, vcc\n\t"                                                     \
      "v_addc_co_u32_e64 v65, s[62:63], v65, 0, s[66:67]\n\t"                                           \
      "v_add_co_u32_dpp v72, vcc, v76, v77 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"      \
      "v_addc_co_u32_e32 v73, vcc, 0, v73, vcc\n\t"                                                     \
      "v_addc_co_u32_e64 v73, s[62:63], v73, 0, s[70:71]\n\t"

__device__ __forceinline__ void mont_core_128(uint32_t (&tl)[2], uint32_t (&th)[2], uint32_t a0, uint32_t a1, uint32_t b0,
                                              uint32_t b1, uint32_t n0, uint32_t n1, uint32_t ninv) {
  asm volatile(
      "v_mov_b32 v64, 0\n\t"
      "v_mov_b32 v65, 0\n\t"
      "v_mov_b32 v72, 0\n\t"
      "v_mov_b32 v73, 0\n\t"
      "v_mov_b32 v70, 0\n\t"
      "s_mov_b32 s60, 0\n\t"
      "v_readlane_b32 s61, %6, s60\n"              // b limb 0
      "1:\n\t"                                     // steps 0..63: b limbs of the lower group
      ".rept 4\n\t" ZKE_MONT128_STEP("%6") ".endr\n\t"
      "s_cmp_lt_u32 s60, 64\n\t"
      "s_cbranch_scc1 1b\n\t"
      "s_mov_b32 s63, 0\n\t"
      "v_readlane_b32 s61, %7, s63\n"              // step 63 prefetched a lower-group limb again: limb 64 is lane 0 of the upper group
      "2:\n\t"                                     // steps 64..127
      ".rept 4\n\t" ZKE_MONT128_STEP("%7") ".endr\n\t"
      "s_cmp_lt_u32 s60, 128\n\t"
      "s_cbranch_scc1 2b\n\t"
      "v_mov_b32 %0, v64\n\t"
      "v_mov_b32 %1, v65\n\t"
      "v_mov_b32 %2, v72\n\t"
      "v_mov_b32 %3, v73"
      : "=&v"(tl[0]), "=&v"(th[0]), "=&v"(tl[1]), "=&v"(th[1])
      : "v"(a0), "v"(a1), "v"(b0), "v"(b1), "v"(n0), "v"(n1), "s"(__builtin_amdgcn_readfirstlane(ninv))
      : "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78",
        "s60", "s61", "s62", "s63", "s64", "s65", "s66", "s67", "s68", "s69", "s70", "s71", "s72", "vcc", "scc");
}
#undef ZKE_MONT128_STEP

// Montgomery product r = a*b*R^-1 mod n, R = 2^(2048*NL); a, b < n; n odd; ninv = -n^-1 mod 2^32.
template <int NL>
__device__ __forceinline__ void mont_mul(Big<NL>& r, const Big<NL>& a, const Big<NL>& b, const Big<NL>& n,
                                         uint32_t ninv, int lane) {
  uint32_t tl[NL], th[NL];
#ifndef ZKE_MONT_CXX
  if constexpr (NL == 1) {
    mont_core_64(tl[0], th[0], a.v[0], b.v[0], n.v[0], ninv);
    Big<NL> t1;
    const uint32_t top1 = big_normalize<NL>(t1, tl, th, lane);
    if (top1 || big_ge<NL>(t1, n)) big_sub<NL>(r, t1, n, lane); else r = t1;
    return;
  }
  if constexpr (NL == 2) {
    mont_core_128(tl, th, a.v[0], a.v[1], b.v[0], b.v[1], n.v[0], n.v[1], ninv);
    Big<NL> t2;
    const uint32_t top2 = big_normalize<NL>(t2, tl, th, lane);
    if (top2 || big_ge<NL>(t2, n)) big_sub<NL>(r, t2, n, lane); else r = t2;
    return;
  }
#endif
#pragma unroll
  for (int q = 0; q < NL; q++) { tl[q] = 0; th[q] = 0; }
#pragma unroll 1
  for (int qi = 0; qi < NL; qi++) {
#pragma unroll 4
    for (int i = 0; i < 64; i++) {
      const uint32_t bi = __builtin_amdgcn_readlane(b.v[qi], i);
      uint32_t lo[NL];
      uint64_t H[NL];
#pragma unroll
      for (int q = 0; q < NL; q++) {
        uint64_t p = (uint64_t)a.v[q] * bi + tl[q];
        lo[q] = (uint32_t)p;
        H[q] = (p >> 32) + th[q];
      }
      const uint32_t m = __builtin_amdgcn_readfirstlane(lo[0] * ninv);
#pragma unroll
      for (int q = 0; q < NL; q++) {
        uint64_t p2 = (uint64_t)n.v[q] * m + lo[q];
        lo[q] = (uint32_t)p2;
        H[q] += (p2 >> 32);
      }
      // divide by 2^32: limb position l+1 becomes l.  lo[0] of lane 0 is zero by construction.
#pragma unroll
      for (int q = 0; q < NL; q++) {
        uint32_t nxt = lane_up(lo[q]);
        if (q + 1 < NL) {                                     // limb 64 (lane 0 of the next group) feeds limb 63
          const uint32_t first = __builtin_amdgcn_readfirstlane(lo[q + 1 < NL ? q + 1 : q]);
          if (lane == 63) nxt = first;
        }
        uint64_t t = H[q] + nxt;
        tl[q] = (uint32_t)t;
        th[q] = (uint32_t)(t >> 32);
      }
    }
  }
  Big<NL> t;
  uint32_t top = big_normalize<NL>(t, tl, th, lane);
  if (top || big_ge<NL>(t, n)) big_sub<NL>(r, t, n, lane); else r = t;
}

// x = 2x mod n (x < n)
template <int NL>
__device__ __forceinline__ void mod_double(Big<NL>& x, const Big<NL>& n, int lane) {
  uint32_t cin = 0;
  Big<NL> d;
#pragma unroll
  for (int q = 0; q < NL; q++) {
    uint32_t below = lane_down(x.v[q]);
    if (lane == 0) below = cin;
    cin = __builtin_amdgcn_readlane(x.v[q], 63);
    d.v[q] = (x.v[q] << 1) | (below >> 31);
  }
  uint32_t top = cin >> 31;
  if (top || big_ge<NL>(d, n)) big_sub<NL>(x, d, n, lane); else x = d;
}

// DigestInfo prefix for SHA-256 (RFC 8017 §9.2 note 1)
__device__ const uint8_t SHA256_DIGESTINFO[19] = {0x30, 0x31, 0x30, 0x0d, 0x06, 0x09, 0x60, 0x86, 0x48, 0x01,
                                                  0x65, 0x03, 0x04, 0x02, 0x01, 0x05, 0x00, 0x04, 0x20};

__device__ const uint8_t SHA1_DIGESTINFO[15] = {0x30, 0x21, 0x30, 0x09, 0x06, 0x05, 0x2b, 0x0e, 0x03, 0x02, 0x1a, 0x05, 0x00, 0x04, 0x14};

// EMSA-PKCS1-v1_5 byte at little-endian position q (q = 0 is the last byte of EM):
// EM = 0x00 0x01 FF..FF 0x00 | DigestInfo prefix | H      (SHA-256: 19 + 32 bytes; SHA-1: 15 + 20)
__device__ __forceinline__ uint32_t emsa_byte(uint32_t q, uint32_t k, const uint32_t* hash_words, bool sha1) {
  const uint32_t hl = sha1 ? 20u : 32u, pl = sha1 ? 15u : 19u;
  if (q >= k) return 0;
  if (q < hl) {
    const uint32_t bi = hl - 1 - q;                          // byte index in the digest as stored
    return (hash_words[bi >> 2] >> (8 * (bi & 3))) & 0xff;
  }
  if (q < hl + pl) return sha1 ? SHA1_DIGESTINFO[pl - 1 - (q - hl)] : SHA256_DIGESTINFO[pl - 1 - (q - hl)];
  if (q == hl + pl) return 0x00;
  if (q == k - 1) return 0x00;
  if (q == k - 2) return 0x01;
  return 0xff;
}

}  // namespace zke
