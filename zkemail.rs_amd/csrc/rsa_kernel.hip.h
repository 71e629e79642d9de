// rsa_kernel.hip.h — the RSA public-key operation of one signature per wavefront (bigint core in rsa.hip.h), as a
// device routine (rsa_wave) and as a stand-alone kernel, and the verdict kernel that follows the hash / modexp stage.
//
// The modular exponentiation needs the key and b= only — not the hashes — so it runs BESIDE the SHA-256 launch of its
// batch (fused.hip.h), not behind it: it leaves "EM has the EMSA-PKCS1-v1_5 shape" and EM's trailing digest bytes in
// EmailMeta, and verdict_kernel compares those bytes with the header hash once both are there
// (rsa 0.9.6 pkcs1v15 verify, reached through cfdkim's verify_signature; call site core/src/email.rs:31-33).
#pragma once
#include "canon.hip.h"

namespace zke {

// Is E the published entry of modulus nn?  Every access to an entry is an agent-scope atomic (sc1: served at the
// coherence point, not from this XCD's L2), so no fence is needed on either side: an agent-scope acquire / release on
// this chip is an L2 invalidate / write-back, paid by every wave of the launch and by whatever else runs on the XCD.
// The entry is immutable once state == 2, and the state load is waited for before the other loads issue.
template <int NL>
__device__ __forceinline__ bool key_cache_hit(const KeyCacheEntry* E, const Big<NL>& nn, uint32_t bits) {
  // Every load issued before the first is looked at: one round trip to the coherence point instead of three (see rsa_route).
  // Limbs served before the entry's publication and a state served after it can only turn a hit into a miss (an unpublished
  // entry does not hold this modulus), never the reverse: an entry is written once and immutable from state == 2 on.
  const int lane = threadIdx.x & 63;
  const uint32_t st = ld_agent(&E->state), eb = ld_agent(&E->bits);
  uint32_t m[NL];
#pragma unroll
  for (int q = 0; q < NL; q++) m[q] = ld_agent(&E->mod[q * 64 + lane]);
  if (st != 2u || eb != bits) return false;
  bool same = true;
#pragma unroll
  for (int q = 0; q < NL; q++) same = same && m[q] == nn.v[q];
  return ballot64(!same) == 0;
}

// EMSA-PKCS1-v1_5 structure in front of the digest: do the limbs of em above the digest (limb >= hl / 4) spell
// 00 01 FF..FF 00 | DigestInfo?  (emsa_byte never touches the hash words for q >= hl.)
template <int NL>
__device__ __forceinline__ bool emsa_structure_ok(const Big<NL>& em, uint32_t k, bool sha1, int lane) {
  const uint32_t hl4 = sha1 ? 5u : 8u;
  bool match = k >= (sha1 ? 46u : 62u);                    // k >= tLen + 11
#pragma unroll
  for (int q = 0; q < NL; q++) {
    const uint32_t limb = q * 64 + lane;
    uint32_t expect = 0;
    if (limb >= hl4) {
#pragma unroll
      for (int b = 0; b < 4; b++) expect |= emsa_byte(4 * limb + b, k, nullptr, sha1) << (8 * b);
    }
    match = match && (ballot64(limb >= hl4 && em.v[q] != expect) == 0);
  }
  return match;
}
// EM's trailing digest, held as little-endian limbs tail[l] (l < hl / 4), against the digest words hw as stored in memory
__device__ __forceinline__ bool emsa_tail_ok(uint32_t tail_of_lane, const uint32_t* hw, bool sha1, int lane) {
  const uint32_t hl4 = sha1 ? 5u : 8u;
  bool bad = false;
  if ((uint32_t)lane < hl4) bad = tail_of_lane != __builtin_bswap32(hw[hl4 - 1 - lane]);
  return ballot64(bad) == 0;
}

// One wave per job.  Jobs routed to the lane-group kernels (RSA_F_QUAD / RSA_F_OCT, set by the front end) are left alone.
//   meta != nullptr: the batch pipeline — em_ok / em_tail go to EmailMeta[job] for verdict_kernel.
//   hash_base != nullptr: the building-block entry point — the digest is at hand, ok_out[job] = full verification.
// em_out (optional): EM big-endian, 512 B slots, right-aligned like RsaJob.sig.
template <int NL>
__device__ __forceinline__ bool rsa_wave(const RsaJob* __restrict__ jobs, uint32_t job,
                                         const uint8_t* __restrict__ hash_base, size_t hash_stride,
                                         uint32_t* __restrict__ ok_out, uint8_t* __restrict__ em_out,
                                         KeyCacheEntry* cache, EmailMeta* meta, uint32_t debug_skip) {
  const int lane = threadIdx.x & 63;
  const RsaJob* J = jobs + job;
  const uint32_t flags = J->flags, k = J->k, bits = J->bits;
  if (flags & (RSA_F_QUAD | RSA_F_OCT)) return false;

  Big<NL> nn, s;
#pragma unroll
  for (int q = 0; q < NL; q++) {
    const uint32_t limb = q * 64 + lane;                    // little-endian limb index
    const uint32_t boff = 512 - 4 * (limb + 1);             // its big-endian byte offset in the 512-byte field
    nn.v[q] = __builtin_bswap32(*(const uint32_t*)(J->mod + boff));
    s.v[q] = __builtin_bswap32(*(const uint32_t*)(J->sig + boff));
  }
  uint32_t ok = 0;
  bool shape_ok = false;
  const uint32_t n0 = __builtin_amdgcn_readfirstlane(nn.v[0]), n1 = __builtin_amdgcn_readlane(nn.v[0], 1);
  const bool odd = (n0 & 1) != 0;
  const bool lenok = J->sig_len == k;             // rsa 0.9.6 pkcs1v15::verify: sig_len != pub_key.size() -> Err
  const bool sha1 = (flags & RSA_F_SHA1) != 0;
  Big<NL> em;
#pragma unroll
  for (int q = 0; q < NL; q++) em.v[q] = 0;
  if ((flags & RSA_F_ACTIVE) && odd && lenok && bits >= 2 && !big_ge<NL>(s, nn) && !debug_skip) {
    Big<NL> rr;
    uint32_t ninv = 0;
    bool hit = false;
    KeyCacheEntry* E = nullptr;
    if (cache) {
      E = cache + key_cache_slot(n0, n1);
      if (key_cache_hit<NL>(E, nn, bits)) {
        // (only now: the constants are valid once state == 2 has been SEEN — loads issued beside the probe could be served
        // before the publication and the state after it)
#pragma unroll
        for (int q = 0; q < NL; q++) rr.v[q] = ld_agent(&E->rr[q * 64 + lane]);
        ninv = ld_agent(&E->ninv);
        hit = true;
      }
    }
    if (!hit) {
      // ninv = -n^-1 mod 2^32 (Newton; n odd)
      uint32_t x = n0;
#pragma unroll
      for (int i = 0; i < 5; i++) x *= 2 - n0 * x;
      ninv = 0u - x;
      // one = R mod n.  2^bits - n, then double (container_bits - bits) times.
      Big<NL> one;
      {
        Big<NL> pw;                                            // 2^bits mod 2^container (0 when bits == container)
#pragma unroll
        for (int q = 0; q < NL; q++) {
          const uint32_t limb = q * 64 + lane;
          pw.v[q] = (bits < 2048u * NL && (bits >> 5) == limb) ? (1u << (bits & 31)) : 0u;
        }
        big_sub<NL>(one, pw, nn, lane);                        // 2^bits - n  (mod 2^container): in [1, n)
        for (uint32_t i = bits; i < 2048u * NL; i++) mod_double<NL>(one, nn, lane);
      }
      rr = one;
      mod_double<NL>(rr, nn, lane);                            // 2R mod n
      // (2R)^(2^t) in the Montgomery domain = 2^(2^t) R; t = log2(container bits) -> R*R = R^2 mod n
      constexpr int T = (NL == 1) ? 11 : 12;
#pragma unroll 1
      for (int i = 0; i < T; i++) mont_mul<NL>(rr, rr, rr, nn, ninv, lane);
      if (cache) {
        uint32_t won = 0;
        if (lane == 0) won = atomicCAS(&E->state, 0u, 1u) == 0u ? 1u : 0u;
        won = __builtin_amdgcn_readfirstlane(won);
        if (won) {
          if (lane == 0) { st_agent(&E->ninv, ninv); st_agent(&E->bits, bits); }
#pragma unroll
          for (int q = 0; q < 2; q++) st_agent(&E->mod[q * 64 + lane], q < NL ? nn.v[q < NL ? q : 0] : 0u);
#pragma unroll
          for (int q = 0; q < NL; q++) st_agent(&E->rr[q * 64 + lane], rr.v[q]);
          if (bits >= 512) {
            // R'^2 mod n for the 28-bit radix of rsa_quad.hip.h (R' = 2^(532 G): 2^2128 for four lanes, 2^4256 for eight):
            // two more products with R = 2^(2048 NL): mont(R^2, 2^c) = 2^c R, mont(R^2, 2^c R) = 2^c R^2 with
            // c = 2 (532 G - 2048 NL) = 160 / 320; then limb t = bits [28 t, 28 t + 28)
            Big<NL> cc, z, r2;
#pragma unroll
            for (int q = 0; q < NL; q++) cc.v[q] = (q == 0 && lane == (NL == 1 ? 5 : 10)) ? 1u : 0u;
            mont_mul<NL>(z, rr, cc, nn, ninv, lane);
            mont_mul<NL>(r2, rr, z, nn, ninv, lane);
            for (uint32_t t0 = 0; t0 < 64u * (NL + 1); t0 += 64) {
              const uint32_t t = t0 + (uint32_t)lane, bit = 28u * t, w = bit >> 5;
              uint32_t lo = 0, hi = 0;
#pragma unroll
              for (int q = 0; q < NL; q++) {                       // 32-bit limb w lives in v[w >> 6] of lane w & 63
                const uint32_t l0 = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(4 * (w & 63)), (int)r2.v[q]);
                const uint32_t l1 = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(4 * ((w + 1) & 63)), (int)r2.v[q]);
                if ((w >> 6) == (uint32_t)q) lo = l0;
                if (((w + 1) >> 6) == (uint32_t)q) hi = l1;
              }
              const uint32_t v = (uint32_t)(((((uint64_t)hi) << 32) | lo) >> (bit & 31)) & 0x0FFFFFFFu;
              if (t < 76u * NL) st_agent(&E->rr28[t], v);
            }
          }
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the entry has reached the coherence point ...
          if (lane == 0) st_agent(&E->state, 2u);                  // ... before it is published
        }
      }
    }
    Big<NL> xm;
    mont_mul<NL>(xm, s, rr, nn, ninv, lane);                 // s in Montgomery form
    Big<NL> acc = xm;
    const uint64_t e = J->e | (J->e == 0);                   // e >= 1 (key decode enforces 2 <= e < 2^33)
    const int top = 63 - __builtin_clzll(e);
#pragma unroll 1
    for (int bit = top - 1; bit >= 0; bit--) {
      mont_mul<NL>(acc, acc, acc, nn, ninv, lane);
      if ((e >> bit) & 1) mont_mul<NL>(acc, acc, xm, nn, ninv, lane);
    }
    Big<NL> lit;
#pragma unroll
    for (int q = 0; q < NL; q++) lit.v[q] = (q == 0 && lane == 0) ? 1u : 0u;
    mont_mul<NL>(em, acc, lit, nn, ninv, lane);              // out of the Montgomery domain
    shape_ok = emsa_structure_ok<NL>(em, k, sha1, lane);     // rsa 0.9.6 pkcs1v15_sign_unpad, all but the digest
    if (hash_base)
      ok = (shape_ok && emsa_tail_ok(em.v[0], (const uint32_t*)(hash_base + (size_t)job * hash_stride), sha1, lane)) ? 1u : 0u;
  }
  if (lane == 0 && ok_out) ok_out[job] = ok;
  if (meta) {
    EmailMeta* M = meta + job;
    if (lane < 8) M->em_tail[lane] = em.v[0];
    if (lane == 0) M->em_ok = shape_ok ? 1u : 0u;
  }
  if (em_out && (flags & RSA_F_ACTIVE)) {     // inactive jobs (later signature rounds) leave the slot alone
#pragma unroll
    for (int q = 0; q < NL; q++) {
      const uint32_t limb = q * 64 + lane;
      *(uint32_t*)(em_out + (size_t)job * 512 + 512 - 4 * (limb + 1)) = __builtin_bswap32(em.v[q]);
    }
    if (NL == 1) *(uint32_t*)(em_out + (size_t)job * 512 + 4 * lane) = 0;   // upper half of the slot
  }
  return ok != 0;               // full verification: meaningful when hash_base was given
}

// a wave picks the one-limb-per-lane (<= 2048 bits) or two-limbs-per-lane path from its job's modulus size
__device__ __forceinline__ bool rsa_wave_any(const RsaJob* __restrict__ jobs, uint32_t job, const uint8_t* __restrict__ hash_base,
                                             size_t hash_stride, uint32_t* __restrict__ ok_out, uint8_t* __restrict__ em_out,
                                             KeyCacheEntry* cache, EmailMeta* meta, uint32_t debug_skip) {
  const uint32_t bits = __builtin_amdgcn_readfirstlane(jobs[job].bits);
  if (bits <= 2048) return rsa_wave<1>(jobs, job, hash_base, hash_stride, ok_out, em_out, cache, meta, debug_skip);
  return rsa_wave<2>(jobs, job, hash_base, hash_stride, ok_out, em_out, cache, meta, debug_skip);
}

// Stand-alone launch (launches too large for the fused kernel, the building-block entry point): grid = n, blockDim = 64.
__global__ __launch_bounds__(64, 6) void rsa_verify_kernel(const RsaJob* __restrict__ jobs, uint32_t n,
                                                         const uint8_t* __restrict__ hash_base, size_t hash_stride,
                                                         uint32_t* __restrict__ ok_out, uint8_t* __restrict__ em_out,
                                                         KeyCacheEntry* cache, EmailMeta* meta, uint32_t debug_skip) {
  const uint32_t job = blockIdx.x;      // one wave per workgroup: a single free wave slot is enough to place it
  if (job >= n) return;
  rsa_wave_any(jobs, job, hash_base, hash_stride, ok_out, em_out, cache, meta, debug_skip);
}

}  // namespace zke
