// rsa_kernel.hip.h — the RSA verification kernel (bigint core in rsa.hip.h) with the per-e-mail verdict fused in.
#pragma once
#include "canon.hip.h"

namespace zke {

// One wave per job.  blockDim = 64, grid = n.  When `fin.b.results` is set the wave also
// writes the e-mail's verdict (the former finalize kernel: one launch and ~20 us of serial loads less per batch).
// ok_out[i]: 1 = signature verifies (EM == EMSA(hash)), 0 = not.  em_out (optional): EM big-endian, 512 B slots,
// right-aligned like RsaJob.sig.  hash_base + i*hash_stride -> 32-byte SHA-256 of the header preimage.
template <int NL>
__device__ __forceinline__ void rsa_wave(const RsaJob* __restrict__ jobs, uint32_t job,
                                         const uint8_t* __restrict__ hash_base, size_t hash_stride,
                                         uint32_t* __restrict__ ok_out, uint8_t* __restrict__ em_out,
                                         KeyCacheEntry* cache, const uint8_t* __restrict__ key_hash_base,
                                         const FinArgs& fin, uint32_t quad) {
  const int lane = threadIdx.x & 63;
  const RsaJob* J = jobs + job;
  const uint32_t flags = J->flags, k = J->k, bits = J->bits;

  Big<NL> nn, s;
#pragma unroll
  for (int q = 0; q < NL; q++) {
    const uint32_t limb = q * 64 + lane;                    // little-endian limb index
    const uint32_t boff = 512 - 4 * (limb + 1);             // its big-endian byte offset in the 512-byte field
    nn.v[q] = __builtin_bswap32(*(const uint32_t*)(J->mod + boff));
    s.v[q] = __builtin_bswap32(*(const uint32_t*)(J->sig + boff));
  }
  uint32_t ok = 0;
  const bool odd = (__builtin_amdgcn_readfirstlane(nn.v[0]) & 1) != 0;
  const bool lenok = J->sig_len == k;             // rsa 0.9.6 pkcs1v15::verify: sig_len != pub_key.size() -> Err
  Big<NL> em;
#pragma unroll
  for (int q = 0; q < NL; q++) em.v[q] = 0;
  if ((flags & RSA_F_ACTIVE) && odd && lenok && bits >= 2 && !big_ge<NL>(s, nn) && !fin.debug_skip_rsa) {
    Big<NL> rr;
    uint32_t ninv = 0;
    bool hit = false;
    KeyCacheEntry* E = nullptr;
    const uint32_t* kh = nullptr;
    // Pre-pass for rsa_group_kernel: e = 65537, 512..2048 bits and the key's constants in the cache (found there, or put
    // there by this wave) -> the job is marked and left to that kernel, verdict included.
    const bool quad_ok = (quad & (NL == 1 ? 1u : 2u)) && cache && fin.b.results && bits >= 512 && J->e == 65537;
    constexpr uint32_t GROUP_FLAG = NL == 1 ? RSA_F_QUAD : RSA_F_OCT;
    if (cache) {
      kh = (const uint32_t*)(key_hash_base + (size_t)job * hash_stride);
      E = cache + (kh[0] % KEY_CACHE_SLOTS);
      // Every access to an entry is an agent-scope atomic (sc1: served at the coherence point, not from this XCD's
      // L2), so no fence is needed on either side: an agent-scope acquire / release on this chip is an L2
      // invalidate / write-back, paid by every wave of the launch and by whatever else runs on the XCD.
      // The entry is immutable once state == 2, and the state load is waited for before the other loads issue.
      auto ld = [](const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
      if (ld(&E->state) == 2u) {
        const bool same = (lane < 8) ? (ld(&E->hash[lane]) == kh[lane]) : true;
        if (ballot64(!same) == 0 && ld(&E->bits) == bits) {
#pragma unroll
          for (int q = 0; q < NL; q++) rr.v[q] = ld(&E->rr[q * 64 + lane]);
          ninv = ld(&E->ninv);
          hit = true;
        }
      }
    }
    if (hit && quad_ok) {
      if (lane == 0) const_cast<RsaJob*>(J)->flags = flags | GROUP_FLAG;
      return;
    }
    if (!hit) {
      // ninv = -n^-1 mod 2^32 (Newton; n odd)
      uint32_t n0 = __builtin_amdgcn_readfirstlane(nn.v[0]);
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
          auto st = [](uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
          if (lane < 8) st(&E->hash[lane], kh[lane]);
          if (lane == 0) { st(&E->ninv, ninv); st(&E->bits, bits); }
#pragma unroll
          for (int q = 0; q < NL; q++) st(&E->rr[q * 64 + lane], rr.v[q]);
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
              if (t < 76u * NL) st(&E->rr28[t], v);
            }
          }
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the entry has reached the coherence point ...
          if (lane == 0) st(&E->state, 2u);                        // ... before it is published
          if (quad_ok) {
            if (lane == 0) const_cast<RsaJob*>(J)->flags = flags | GROUP_FLAG;
            return;
          }
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
    // EMSA-PKCS1-v1_5 compare (rsa 0.9.6 pkcs1v15_sign_unpad); needs k >= 19 + 32 + 11
    const uint32_t* hw = (const uint32_t*)(hash_base + (size_t)job * hash_stride);
    const bool sha1 = (flags & RSA_F_SHA1) != 0;
    bool match = k >= (sha1 ? 46u : 62u);                    // k >= tLen + 11
#pragma unroll
    for (int q = 0; q < NL; q++) {
      const uint32_t limb = q * 64 + lane;
      uint32_t expect = 0;
#pragma unroll
      for (int b = 0; b < 4; b++) expect |= emsa_byte(4 * limb + b, k, hw, sha1) << (8 * b);
      match = match && (ballot64(em.v[q] != expect) == 0);
    }
    ok = match ? 1u : 0u;
  }
  if (lane == 0 && ok_out) ok_out[job] = ok;
  if (fin.b.results) verdict_wave(fin, job, ok != 0, lane);
  if (em_out && (flags & RSA_F_ACTIVE)) {     // inactive jobs (later signature rounds) leave the slot alone
#pragma unroll
    for (int q = 0; q < NL; q++) {
      const uint32_t limb = q * 64 + lane;
      *(uint32_t*)(em_out + (size_t)job * 512 + 512 - 4 * (limb + 1)) = __builtin_bswap32(em.v[q]);
    }
    if (NL == 1) *(uint32_t*)(em_out + (size_t)job * 512 + 4 * lane) = 0;   // upper half of the slot
  }
}


// One kernel for both containers: a wave picks the one-limb-per-lane (<= 2048 bits) or two-limbs-per-lane path
// from its job's modulus size (wave-uniform branch).
__global__ __launch_bounds__(64, 6) void rsa_verify_kernel(const RsaJob* __restrict__ jobs, uint32_t n,
                                                         const uint8_t* __restrict__ hash_base, size_t hash_stride,
                                                         uint32_t* __restrict__ ok_out, uint8_t* __restrict__ em_out,
                                                         KeyCacheEntry* cache, const uint8_t* __restrict__ key_hash_base,
                                                         FinArgs fin, uint32_t quad) {
  const uint32_t job = blockIdx.x;      // one wave per workgroup: a single free wave slot is enough to place it
  if (job >= n) return;
  if (fin.b.results && fin.b.meta[job].state == ST_PENDING) return;     // waits for a later signature round
  const uint32_t bits = __builtin_amdgcn_readfirstlane(jobs[job].bits);
  if (bits <= 2048) rsa_wave<1>(jobs, job, hash_base, hash_stride, ok_out, em_out, cache, key_hash_base, fin, quad);
  else rsa_wave<2>(jobs, job, hash_base, hash_stride, ok_out, em_out, cache, key_hash_base, fin, quad);
}

}  // namespace zke
