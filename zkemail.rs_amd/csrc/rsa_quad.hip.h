// rsa_quad.hip.h — RSA verification with FOUR LANES PER SIGNATURE (16 signatures per wavefront) for moduli of up
// to 2048 bits and e = 65537: the same rsa 0.9.6 / num-bigint-dig operation as rsa.hip.h (call site
// core/src/email.rs:31-33; RFC 8017 §8.2.2, §9.2), 4.6 k instead of 12.0 k VALU instructions per signature (measured).
//
// Why.  The path is VALU-issue bound and the one-limb-per-lane kernel spends 9 instructions per
// limb and CIOS step: two multiplies, and seven to read the multiplier digit, form the quotient digit, shift the
// accumulator one lane down and keep its carries.  Here a number is 76 limbs of 28 bits, 19 consecutive limbs per
// lane, four lanes (one DPP quad) per signature:
//   * products of 28-bit limbs are < 2^56, and a register holds at most 38 of them in its life as a column, so
//     every column is a plain 64-bit accumulator: ONE v_mad_u64_u32 per limb product, no carry instructions;
//   * the accumulator is a window of 38 columns per lane addressed at compile time (column k + r for limb k at
//     step r of a block of 19 steps), so nothing is shifted inside a block; the per-step cross-lane work is two
//     quad broadcasts (multiplier digit, quotient digit) for 38 multiplies;
//   * every 19 steps the reduced low half of a lane's window (zero in lane 0) moves one lane down and the high half
//     becomes the low half: 19 quad rotations + 19 additions per 722 multiplies;
//   * R = 2^2128 > 4n, so values stay in [0, 2n) without any conditional subtraction; carries are normalised once per
//     product (limbs <= 2^28), exactly only for the final result.
// 55 instructions per step for 16 signatures instead of 9 per step for one.  Moduli of 2049..4096 bits run the same
// code with eight lanes per signature (152 limbs, R = 2^4256, eight blocks of 19 steps).
//
// R^2 mod n for this radix (2^4256 mod n; 2^8512 mod n for eight lanes) comes from the key cache.  The front end looks a
// decoded key up there (by modulus: exact) and routes the e-mail's signatures here (RSA_F_QUAD / RSA_F_OCT) when the
// entry exists and e = 65537; a key seen for the first time — and other exponents, keys whose cache slot belongs to
// another key — goes to the one-signature-per-wave routine, which fills the entry (two more 32-bit-radix Montgomery
// products turn 2^4096 mod n into 2^4256 mod n).  Signatures rsa 0.9.6 rejects before the arithmetic are rejected here too.  The algorithm and its register bounds are modelled with
// Python integers in tests/test_rsa_group_model.py.
#pragma once
#include "rsa_kernel.hip.h"

namespace zke {

constexpr int QL = 19;                       // limbs per lane
constexpr uint32_t QMASK = 0x0FFFFFFFu;
struct QBig { uint32_t v[QL]; };             // lane p of the group: limbs 19p .. 19p+18

// Lane groups of G = 4 (one DPP quad: 76 limbs, moduli <= 2048 bits) or G = 8 (half a DPP row: 152 limbs, <= 4096 bits).
// (bound_ctrl on the full-mask moves: lanes without a source read 0 and the destination needs no initial value)
template <int G> __device__ __forceinline__ uint32_t g_bcast0(uint32_t x) {        // lane 0 of the group to all of it
  uint32_t q = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x00 /*quad_perm:[0,0,0,0]*/, 0xf, 0xf, true);
  if (G == 8) q = (uint32_t)__builtin_amdgcn_update_dpp((int)q, (int)q, 0x114 /*row_shr:4*/, 0xf, 0xA /*lanes 4-7, 12-15*/, false);
  return q;
}
template <int G> __device__ __forceinline__ uint32_t g_rotdown(uint32_t x, int p) {   // lane p <- lane p+1, lane G-1 <- lane 0
  if (G == 4) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x39 /*quad_perm:[1,2,3,0]*/, 0xf, 0xf, true);
  const uint32_t a = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x101 /*row_shl:1*/, 0xf, 0xf, true);
  const uint32_t b = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x117 /*row_shr:7*/, 0xf, 0xf, true);
  return p == 7 ? b : a;
}
template <int G> __device__ __forceinline__ uint32_t g_fromprev(uint32_t x) {      // lane p <- lane p-1 (lane 0: callers mask it)
  if (G == 4) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x90 /*quad_perm:[0,0,1,2]*/, 0xf, 0xf, true);
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111 /*row_shr:1*/, 0xf, 0xf, true);
}

// W = (a * b + sum_i m_i * n * 2^(28 i)) / 2^(532 G) as 19 lazy 64-bit columns per lane (column j of lane p: limb 19p + j).
// A register holds at most 38 products (< 2^56 each) in its life as a high and then a low column: no overflow.
template <int G>
__device__ __forceinline__ void qmont_columns(uint64_t (&W)[2 * QL], const QBig& a, const QBig& b, const QBig& n, uint32_t ninv, int p) {
#pragma unroll
  for (int j = 0; j < 2 * QL; j++) W[j] = 0;
  QBig B = b;
#pragma unroll 1
  for (int blk = 0; blk < G; blk++) {
    // 19 steps: multiplier digits 19 blk + r, held by lane 0 of the group after blk rotations of B
#pragma unroll
    for (int r = 0; r < QL; r++) {
      const uint32_t bd = g_bcast0<G>(B.v[r]);
#pragma unroll
      for (int k = 0; k < QL; k++)           // column r + 18 is touched here for the first time in this block (r > 0)
        W[k + r] = (uint64_t)a.v[k] * bd + ((k == QL - 1 && r > 0) ? 0ull : W[k + r]);
      const uint32_t m = g_bcast0<G>(((uint32_t)W[r] * ninv) & QMASK);       // lane 0's column r is the lowest live limb
#pragma unroll
      for (int k = 0; k < QL; k++) W[k + r] = (uint64_t)n.v[k] * m + W[k + r];
      W[r + 1] += W[r] >> 28;              // lane 0: the column is now a multiple of 2^28; other lanes: a partial carry
      W[r] &= QMASK;
    }
    // the window moves up 19 limbs: finished low columns go one lane down (lane 0's are zero and reach the top lane)
#pragma unroll
    for (int j = 0; j < QL; j++) {
      const uint32_t recv = g_rotdown<G>((uint32_t)W[j], p);
      W[j] = W[QL + j] + recv;
    }
#pragma unroll
    for (int r = 0; r < QL; r++) B.v[r] = g_rotdown<G>(B.v[r], p);
  }
}

// columns -> limbs.  CROSS cross-lane passes: 1 leaves limbs <= 2^28 (good enough as an operand), G - 1 is exact.
template <int G, int CROSS>
__device__ __forceinline__ void qnorm(QBig& out, const uint64_t (&W)[2 * QL], int p) {
  uint64_t c = 0;
#pragma unroll
  for (int j = 0; j < QL; j++) {
    const uint64_t t = W[j] + c;
    out.v[j] = (uint32_t)t & QMASK;
    c = t >> 28;
  }
  uint32_t clo = (uint32_t)c, chi = (uint32_t)(c >> 32);       // < 2^37 out of the local pass
#pragma unroll
  for (int pass = 0; pass < CROSS; pass++) {
    uint32_t ilo = g_fromprev<G>(clo), ihi = g_fromprev<G>(chi);
    if (p == 0) { ilo = 0; ihi = 0; }
    const uint64_t t0 = (uint64_t)out.v[0] + (((uint64_t)ihi << 32) | ilo);
    out.v[0] = (uint32_t)t0 & QMASK;
    uint32_t c32 = (uint32_t)(t0 >> 28);
#pragma unroll
    for (int j = 1; j < QL; j++) {
      const uint32_t t = out.v[j] + c32;
      out.v[j] = t & QMASK;
      c32 = t >> 28;
    }
    clo = c32; chi = 0;                                         // 0 or 1 from here on
  }
  uint32_t last = g_fromprev<G>(clo);
  if (p == 0) last = 0;
  out.v[0] += last;                                             // value-preserving; zero after G - 1 passes
}

// One wave: NG = 64 / G signatures.  job0 = the wave's first job.  Llimb: NG x (G * QL + 4) dwords of LDS private to the wave.
//   meta != nullptr: em_ok / em_tail of each signature go to EmailMeta for verdict_kernel (the batch pipeline);
//   hash_base != nullptr: the digest is at hand, ok_out[job] = full verification.
template <int G>
__device__ __forceinline__ void rsa_group_wave(const RsaJob* __restrict__ jobs, uint32_t n, uint32_t job0, uint32_t* Llimb_raw,
                                               const uint8_t* __restrict__ hash_base, size_t hash_stride,
                                               uint32_t* __restrict__ ok_out, uint8_t* __restrict__ em_out,
                                               const KeyCacheEntry* cache, EmailMeta* meta) {
  constexpr int NG = 64 / G, LIMBS = G * QL;
  constexpr uint32_t MY_FLAG = G == 4 ? RSA_F_QUAD : RSA_F_OCT;
  uint32_t (*Llimb)[LIMBS + 4] = reinterpret_cast<uint32_t (*)[LIMBS + 4]>(Llimb_raw);
#ifndef ZKE_QUAD_PRIO
#define ZKE_QUAD_PRIO 3
#endif
  const int lane = threadIdx.x & 63, p = lane & (G - 1), grp = lane / G;
  const uint32_t job = job0 + grp;
  const RsaJob* J = jobs + (job < n ? job : 0);
  uint32_t flags = 0;
  if (job < n) flags = J->flags;
  const bool act = (flags & MY_FLAG) != 0;
  if (__ballot(act) == 0) return;
  // NG signatures share one long dependency chain (~95 k instructions for G = 4): served round-robin with the short waves
  // of other batches it would stretch several times over and hold its whole batch back; the others have parallel slack.
  __builtin_amdgcn_s_setprio(ZKE_QUAD_PRIO);

  QBig nn, s, rr;
#pragma unroll
  for (int j = 0; j < QL; j++) { nn.v[j] = 0; s.v[j] = 0; rr.v[j] = 0; }
  uint32_t ninv = 0, kbytes = 0;
  if (act) {
    typedef uint64_t __attribute__((aligned(1))) u64_unaligned;
    auto limb_of = [&](const uint8_t* field, uint32_t t) -> uint32_t {     // bits [28 t, 28 t + 28) of a big-endian 512-byte field
      const uint32_t bit = 28u * t, o = bit >> 3;                          // little-endian byte o = field[511 - o]
      if (o >= 512) return 0;                                              // beyond 4096 bits
      const uint32_t oo = o > 504 ? 504u : o;                              // the top limbs: read the field's first 8 bytes and shift
      const uint64_t v = __builtin_bswap64(*(const u64_unaligned*)(field + 504 - oo)) >> (8 * (o - oo));
      return (uint32_t)(v >> (bit & 7)) & QMASK;
    };
    // the front end found this modulus in the cache (all limbs compared) before it set MY_FLAG; entries are immutable
    const uint32_t n0 = __builtin_bswap32(*(const uint32_t*)(J->mod + 508)), n1 = __builtin_bswap32(*(const uint32_t*)(J->mod + 504));
    const KeyCacheEntry* E = cache + key_cache_slot(n0, n1);
#pragma unroll
    for (int j = 0; j < QL; j++) {
      nn.v[j] = limb_of(J->mod, QL * p + j);
      s.v[j] = limb_of(J->sig, QL * p + j);
      rr.v[j] = ld_agent(&E->rr28[QL * p + j]);
    }
    ninv = ld_agent(&E->ninv) & QMASK;
    kbytes = J->k;
  }
  {
    // rsa 0.9.6 rejects a signature whose length is not the modulus length, or with s >= n, before any arithmetic:
    // such a job runs with s = 0, whose EM = 0 has no EMSA shape (em_ok = 0, an all-zero EM block, as the wave path leaves).
    // s >= n: per lane the sign of the highest differing limb; the highest lane of the group that differs decides.
    int c = 0;
#pragma unroll
    for (int j = QL - 1; j >= 0; j--) c = c != 0 ? c : (int)(s.v[j] > nn.v[j]) - (int)(s.v[j] < nn.v[j]);
    const uint64_t gmask0 = (G == 4 ? 0xFull : 0xFFull);
    const uint64_t gtg = (__ballot(c > 0) >> (G * grp)) & gmask0, ltg = (__ballot(c < 0) >> (G * grp)) & gmask0;
    const bool reject = act && (J->sig_len != kbytes || gtg >= ltg);
    if (reject) {
#pragma unroll
      for (int j = 0; j < QL; j++) s.v[j] = 0;
    }
  }

  // s^65537 in 18 products: s R (into the Montgomery domain), sixteen squarings -> s^65536 R, and the last product takes
  // the PLAIN s: (s^65536 R) s / R = s^65537 — out of the domain without a nineteenth product by one.  That value is
  // < n + n^2 / R (a < 2n, s < n, R > 2^80 n): one conditional subtraction makes it exact.
  QBig acc = s;
  uint64_t W[2 * QL];
#pragma unroll 1
  for (int step = 0; step < 17; step++) {
    QBig b;
#pragma unroll
    for (int j = 0; j < QL; j++) b.v[j] = step == 0 ? rr.v[j] : acc.v[j];
    qmont_columns<G>(W, acc, b, nn, ninv, p);
    qnorm<G, 1>(acc, W, p);
  }
  qmont_columns<G>(W, acc, s, nn, ninv, p);
  qnorm<G, G - 1>(acc, W, p);                 // exact limbs
  {
    // acc >= n?  Per lane the sign of the highest differing limb; the highest differing lane of the group decides, and the
    // lanes below a lane decide the borrow it starts with.
    int c = 0;
#pragma unroll
    for (int j = QL - 1; j >= 0; j--) c = c != 0 ? c : (int)(acc.v[j] > nn.v[j]) - (int)(acc.v[j] < nn.v[j]);
    const uint32_t gm = (G == 4 ? 0xFu : 0xFFu);
    const uint32_t gtg = (uint32_t)(__ballot(c > 0) >> (G * grp)) & gm, ltg = (uint32_t)(__ballot(c < 0) >> (G * grp)) & gm;
    if (gtg >= ltg) {                         // EM + n -> EM (never for a signature that verifies: EM < n / 2^15 there)
      const uint32_t low = (1u << p) - 1u;
      uint32_t borrow = (ltg & low) > (gtg & low) ? 1u : 0u;
#pragma unroll
      for (int j = 0; j < QL; j++) {
        const uint32_t t = acc.v[j] - nn.v[j] - borrow;
        acc.v[j] = t & QMASK;
        borrow = t >> 31;
      }
    }
  }

  // EMSA-PKCS1-v1_5 (rsa 0.9.6 pkcs1v15_sign_unpad), byte by byte through LDS: the structure in front of the digest is
  // checked here; the digest bytes are compared now (hash_base) or handed to verdict_kernel (meta)
#pragma unroll
  for (int j = 0; j < QL; j++) Llimb[grp][QL * p + j] = acc.v[j];
  if (p < 4) Llimb[grp][LIMBS + p] = 0;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  const bool sha1 = (flags & RSA_F_SHA1) != 0;
  const uint32_t hl = sha1 ? 20u : 32u;
  bool bad = false, tail_bad = false;
  if (act) {
    const uint32_t* hw = hash_base ? (const uint32_t*)(hash_base + (size_t)job * hash_stride) : nullptr;
    uint8_t* tail = meta ? reinterpret_cast<uint8_t*>(meta[job].em_tail) : nullptr;
    for (uint32_t u = 0; u < 64; u++) {
      const uint32_t i = G * u + (uint32_t)p;                    // little-endian byte index: 64 G bytes per group
      const uint32_t t = (8 * i) / 28, sh = 8 * i - 28 * t;
      const uint64_t two = (uint64_t)Llimb[grp][t] | ((uint64_t)Llimb[grp][t + 1] << 28);
      const uint32_t got = (uint32_t)(two >> sh) & 0xff;
      if (i < hl) {
        if (tail) tail[i] = (uint8_t)got;                        // little-endian limb image: byte i of EM counted from its end
        if (hw) tail_bad = tail_bad || got != emsa_byte(i, kbytes, hw, sha1);
      } else {
        bad = bad || got != emsa_byte(i, kbytes, nullptr, sha1);
      }
      if (em_out) em_out[(size_t)job * 512 + 511 - i] = (uint8_t)got;
    }
    if (em_out && G == 4) {
#pragma unroll
      for (int z = 0; z < 16; z++) *(uint32_t*)(em_out + (size_t)job * 512 + 64 * p + 4 * z) = 0;     // upper half of the slot
    }
  }
  const uint64_t badm = __ballot(bad), tbadm = __ballot(tail_bad);
  const uint64_t gmask = (G == 4 ? 0xFull : 0xFFull);
  const bool shape_ok = act && ((badm >> (G * grp)) & gmask) == 0 && kbytes >= (sha1 ? 46u : 62u);      // k >= tLen + 11
  if (act && p == 0) {
    if (ok_out) ok_out[job] = (hash_base && shape_ok && ((tbadm >> (G * grp)) & gmask) == 0) ? 1u : 0u;
    if (meta) meta[job].em_ok = shape_ok ? 1u : 0u;
  }
}

template <int G>
__global__ __launch_bounds__(64) void rsa_group_kernel(const RsaJob* __restrict__ jobs, uint32_t n,
                                                       const uint8_t* __restrict__ hash_base, size_t hash_stride,
                                                       uint32_t* __restrict__ ok_out, uint8_t* __restrict__ em_out,
                                                       const KeyCacheEntry* cache, EmailMeta* meta) {
  constexpr int NG = 64 / G, LIMBS = G * QL;
  __shared__ uint32_t Llimb[NG * (LIMBS + 4)];
  rsa_group_wave<G>(jobs, n, blockIdx.x * NG, Llimb, hash_base, hash_stride, ok_out, em_out, cache, meta);
}

}  // namespace zke
