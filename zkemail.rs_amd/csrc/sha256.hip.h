// sha256.hip.h — batched SHA-256 for gfx950 (CDNA4).  FIPS 180-4.
//
// Replaces sha2 0.10.9 on the verify path: hash_bytes (core/src/crypto.rs:3-7, used at
// core/src/circuits.rs:16-17) and the body / header hashes inside cfdkim
// (call site core/src/email.rs:31-33).
//
// Mapping.  SHA-256 is a strictly sequential chain per message, so parallelism comes
// from the batch: ONE MESSAGE PER LANE for the compression (64 messages per wavefront,
// all 64 lanes doing integer VALU work), while the bytes are fetched WAVE-COOPERATIVELY:
// for each tile of T bytes the 64 lanes read each message's tile with 16-byte
// lane-contiguous global loads (T/16 lanes cover one message's tile -> 256-byte
// contiguous segments at T=256), park it in the wave's private LDS slab, and every lane
// then pulls its own message's 64-byte blocks back with ds_read_b128.  Row stride T+16
// bytes keeps both the ds_write_b128 and the ds_read_b128 phases conflict-free
// (lane stride = 4 banks mod 64).  No __syncthreads: a wave's LDS operations execute in
// order, and no other wave touches its slab.
//
// Roofline: integer-VALU bound (about 1.4k VALU per 64-byte block per lane), not HBM
// bound; DESIGN.md §3 has the arithmetic.  Two kernels: sha256_batch_kernel (one wave per 64 messages, for
// launches that fill the chip) and sha256_pair_kernel (two waves per 64 messages — message schedule and rounds
// — for launches whose duration is one wave's chain of compressions; it is the one BASELINE-sized batches use).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace zke {

struct ShaJob {          // one message
  uint64_t src;          // device address of the bytes
  uint64_t dst;          // device address of the 32-byte digest slot (0 = inactive job)
  uint32_t len;          // bytes
  uint32_t pad;          // 0: SHA-256; 1: SHA-1 (a=rsa-sha1 signatures; 20-byte digest, slot zero padded)
};

// ---- length buckets --------------------------------------------------------------------------------------------------
// A wave hashes 64 messages in lock-step: it runs as many compressions as its LONGEST message needs.  In a ragged batch
// (body lengths log-uniform over 0 .. 64 KB) 64 messages in arrival order span the whole range, so most lanes of
// most waves idle for most of the launch.  The front end therefore files every body (and header preimage) under a length
// class — its block count to three significant bits — with one atomic add: key[i] = class << 24 | position within the class.
// A hash group lays the classes out longest first (a prefix sum over the 192 counters), finds the 64 messages whose global
// position falls into its range by one pass over the keys, and hashes messages of one class: no wave waits for a message more
// than 12.5 % longer than its shortest.  No sort, no extra launch; the digests land by ShaJob::dst, so record order is untouched.
// Batches whose messages all fall into one or two neighbouring classes (the uniform BASELINE configs) keep the direct
// mapping lane -> job: the pass over the keys is skipped.
constexpr uint32_t SHA_CLASSES = 192;               // block counts 1..15 exactly, then 8 classes per octave up to 2^25 blocks
constexpr uint32_t SHA_ORDER_KIND_WORDS = 256;      // counters of one kind (192 used).  The slot's order buffer: the counters of kind 0,
                                                    // those of kind 1 — at fixed places, whatever the batch size: they carry over from
                                                    // batch to batch —, then n_pad keys of kind 0, n_pad keys of kind 1
__host__ __device__ inline size_t sha_order_cnt(uint32_t kind) { return (size_t)kind * SHA_ORDER_KIND_WORDS; }
__host__ __device__ inline size_t sha_order_key(uint32_t kind, uint32_t n_pad) { return 2 * (size_t)SHA_ORDER_KIND_WORDS + (size_t)kind * n_pad; }
constexpr uint32_t SHA_KEY_NONE = 0xFFFFFFFFu;
__host__ __device__ inline uint32_t sha_len_class(uint32_t nblk) {
  if (nblk < 16) return nblk;
  const uint32_t e = 31u - (uint32_t)__builtin_clz(nblk);                  // >= 4
  return 16u + (e - 4u) * 8u + ((nblk >> (e - 3u)) & 7u);
}
struct ShaOrder {                 // one kind's part of the slot's order buffer (nullptr cnt: no buckets, direct mapping)
  const uint32_t* cnt;            // [SHA_CLASSES] messages per class
  const uint32_t* key;            // [n] class << 24 | position, SHA_KEY_NONE for an e-mail without a message of this kind
  uint32_t n;                     // e-mails
};

__device__ __forceinline__ uint32_t rotr32(uint32_t x, int n) { return __builtin_amdgcn_alignbit(x, x, n); }
// gfx950 v_bitop3_b32: any 3-input boolean function in one VALU op (truth table in the immediate)
__device__ __forceinline__ uint32_t xor3(uint32_t a, uint32_t b, uint32_t c) { return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96); }
__device__ __forceinline__ uint32_t ch3(uint32_t e, uint32_t f, uint32_t g) { return __builtin_amdgcn_bitop3_b32(e, f, g, 0xCA); }   // e ? f : g
__device__ __forceinline__ uint32_t maj3(uint32_t a, uint32_t b, uint32_t c) { return __builtin_amdgcn_bitop3_b32(a, b, c, 0xE8); }

// One compression.  w[16] holds the big-endian message words and is consumed.
// The round constants are compile-time literals after full unrolling (no loads).
__device__ __forceinline__ void sha256_compress(uint32_t (&st)[8], uint32_t (&w)[16]) {
  constexpr uint32_t K[64] = {
      0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5,
      0xd807aa98, 0x12835b01, 0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174,
      0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da,
      0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967,
      0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
      0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070,
      0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3,
      0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};
  uint32_t a = st[0], b = st[1], c = st[2], d = st[3], e = st[4], f = st[5], g = st[6], h = st[7];
#pragma unroll
  for (int i = 0; i < 64; i++) {
    uint32_t wi;
    if (i < 16) {
      wi = w[i];
    } else {
      uint32_t w15 = w[(i - 15) & 15], w2 = w[(i - 2) & 15];
      uint32_t s0 = xor3(rotr32(w15, 7), rotr32(w15, 18), w15 >> 3);
      uint32_t s1 = xor3(rotr32(w2, 17), rotr32(w2, 19), w2 >> 10);
      wi = w[i & 15] + s0 + w[(i - 7) & 15] + s1;
      w[i & 15] = wi;
    }
    uint32_t S1 = xor3(rotr32(e, 6), rotr32(e, 11), rotr32(e, 25));
    uint32_t t1 = (h + S1 + ch3(e, f, g)) + (K[i] + wi);
    uint32_t S0 = xor3(rotr32(a, 2), rotr32(a, 13), rotr32(a, 22));
    uint32_t mj = maj3(a, b, c);
    h = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + S0 + mj;
  }
  st[0] += a; st[1] += b; st[2] += c; st[3] += d; st[4] += e; st[5] += f; st[6] += g; st[7] += h;
}

// SHA-1 (FIPS 180-4 §6.1), same block size and padding as SHA-256; used only for a=rsa-sha1 signatures
// (cfdkim HashAlgo::RsaSha1 over sha-1 0.10.1).
__device__ __forceinline__ void sha1_compress(uint32_t (&st)[8], uint32_t (&w)[16]) {
  uint32_t a = st[0], b = st[1], c = st[2], d = st[3], e = st[4];
#pragma unroll
  for (int i = 0; i < 80; i++) {
    uint32_t wi;
    if (i < 16) {
      wi = w[i];
    } else {
      const uint32_t x = xor3(w[(i - 3) & 15], w[(i - 8) & 15], w[(i - 14) & 15]) ^ w[i & 15];
      wi = rotr32(x, 31);
      w[i & 15] = wi;
    }
    uint32_t f, k;
    if (i < 20) { f = ch3(b, c, d); k = 0x5A827999u; }
    else if (i < 40) { f = xor3(b, c, d); k = 0x6ED9EBA1u; }
    else if (i < 60) { f = maj3(b, c, d); k = 0x8F1BBCDCu; }
    else { f = xor3(b, c, d); k = 0xCA62C1D6u; }
    const uint32_t t = rotr32(a, 27) + f + e + (k + wi);
    e = d; d = c; c = rotr32(b, 2); b = a; a = t;
  }
  st[0] += a; st[1] += b; st[2] += c; st[3] += d; st[4] += e;
}

typedef uint4 __attribute__((aligned(1))) uint4_unaligned;
// Message bytes are in HBM, but their addresses arrive as integers (ShaJob::src), so the compiler would emit FLAT
// loads: those count on lgkmcnt as well as vmcnt, and every wait for an LDS read behind one (the next row's
// descriptor) then waits for HBM too — a tile's loads went out one at a time.  Typed as address space 1 they are
// global_load_*: all of a tile's loads are in flight together and only commit() waits for them.
typedef uint32_t u32x4_raw __attribute__((ext_vector_type(4)));          // a plain vector: the host pass cannot bind HIP's uint4 class across address spaces
typedef const u32x4_raw __attribute__((aligned(1), address_space(1)))* gptr_u4;
typedef const uint8_t __attribute__((address_space(1)))* gptr_u8;
typedef uint32_t __attribute__((address_space(1)))* gptr_out32;

// Launch: blockDim = 256 (4 independent waves), grid = ceil(n / 256).
// LDS: 4 waves * 64 rows * (T + 16) bytes  (T = 256 -> 69,632 B per block).
template <int T>
__global__ __launch_bounds__(256) void sha256_batch_kernel(const ShaJob* __restrict__ jobs, uint32_t n) {
  static_assert(T % 64 == 0 && T >= 64 && T <= 1024, "tile must be whole SHA blocks");
  constexpr int ROW = T + 16;            // bytes
  constexpr int LPR = T / 16;            // lanes that cover one row's tile
  constexpr int RPI = 64 / LPR;          // rows per load instruction
  extern __shared__ __attribute__((aligned(16))) uint8_t lds_raw[];
  // A message is a serial chain of compressions, so this kernel's duration is set by its longest message, not by
  // the chip: with several batches in flight its few waves share SIMDs with thousands of front-end / RSA waves and
  // the chain stretches ~2x.  Raising the wave priority keeps the chain near its unloaded latency; the other
  // kernels have parallel slack to absorb it.
#ifndef ZKE_SHA_PRIO
#define ZKE_SHA_PRIO 3
#endif
  __builtin_amdgcn_s_setprio(ZKE_SHA_PRIO);
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  uint8_t* slab = lds_raw + (size_t)wave * (64 * ROW + 64 * 16);
  uint8_t* desc = slab + 64 * ROW;       // 64 x {src(8), len(4), pad(4)}

  const uint32_t m = (blockIdx.x * 4 + wave) * 64 + lane;
  uint64_t my_src = 0, my_dst = 0;
  uint32_t my_len = 0, my_nblk = 0, my_algo = 0;
  if (m < n) {
    ShaJob j = jobs[m];
    my_src = j.src; my_dst = j.dst; my_len = j.len; my_algo = j.pad;
    my_nblk = my_dst ? (my_len + 9 + 63) >> 6 : 0;      // dst == 0 marks an inactive job
  }
  *(uint64_t*)(desc + lane * 16) = my_src;
  *(uint32_t*)(desc + lane * 16 + 8) = my_len;
  *(uint32_t*)(desc + lane * 16 + 12) = my_nblk;
  uint32_t max_nblk = my_nblk;
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) max_nblk = max(max_nblk, (uint32_t)__shfl_xor((int)max_nblk, o));
  max_nblk = __builtin_amdgcn_readfirstlane(max_nblk);

  uint32_t st[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
  const bool any_sha1 = __ballot(my_algo != 0) != 0;        // wave-uniform: SHA-256-only waves skip the SHA-1 code
  if (my_algo) { st[0] = 0x67452301; st[1] = 0xEFCDAB89; st[2] = 0x98BADCFE; st[3] = 0x10325476; st[4] = 0xC3D2E1F0; st[5] = st[6] = st[7] = 0; }
  uint8_t* my_row = slab + lane * ROW;

  // Software pipeline: the global loads of tile t+1 are issued into registers BEFORE tile t is compressed and
  // land in LDS after it, so a wave's HBM latency hides under its own ~LPR/4 x 1.4k VALU of compression.
  uint4 stage[LPR];
  uint32_t live = 0;                       // bit g: stage[g] holds a row chunk that must be written to LDS
  auto fetch = [&](uint32_t blk0) {
    const uint32_t tile_off = blk0 * 64;
    live = 0;
    uint4 dsc[LPR];                              // every row descriptor first (one ds_read_b128 each), then the loads
#pragma unroll
    for (int g = 0; g < LPR; g++) dsc[g] = *(const uint4*)(desc + (g * RPI + lane / LPR) * 16);
#pragma unroll
    for (int g = 0; g < LPR; g++) {
      const int chunk = lane % LPR;
      const uint64_t rsrc = ((uint64_t)dsc[g].y << 32) | dsc[g].x;
      const uint32_t rlen = dsc[g].z;
      const uint32_t rnblk = dsc[g].w;
      const uint32_t off = tile_off + chunk * 16;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (blk0 < rnblk) {                      // rows that are already finished are skipped
        live |= 1u << g;
        if (off + 16 <= rlen) {
          { const u32x4_raw t4 = *(gptr_u4)(rsrc + off); v = make_uint4(t4.x, t4.y, t4.z, t4.w); }
        } else if (off < rlen) {               // last partial chunk of the message: byte loads, never past the end
          gptr_u8 p = (gptr_u8)(rsrc + off);
          uint32_t rem = rlen - off, t[4] = {0, 0, 0, 0};
          for (uint32_t b = 0; b < rem; b++) t[b >> 2] |= (uint32_t)p[b] << (8 * (b & 3));
          v = make_uint4(t[0], t[1], t[2], t[3]);
        }
      }
      stage[g] = v;
    }
  };
  auto commit = [&](uint32_t blk0) {           // registers -> LDS slab, then the padding of this tile
    const uint32_t tile_off = blk0 * 64;
#pragma unroll
    for (int g = 0; g < LPR; g++) {
      const int row = g * RPI + lane / LPR;
      const int chunk = lane % LPR;
      if (live & (1u << g)) *(uint4*)(slab + row * ROW + chunk * 16) = stage[g];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // ---- padding, patched into the lane's own row (FIPS 180-4 §5.1.1)
    if (blk0 < my_nblk) {
      if (my_len >= tile_off && my_len < tile_off + T) my_row[my_len - tile_off] = 0x80;
      const uint32_t last = my_nblk - 1;
      if (last >= blk0 && last < blk0 + T / 64) {
        const uint64_t bits = (uint64_t)my_len * 8;
        uint32_t* tail = (uint32_t*)(my_row + (last - blk0) * 64 + 56);
        tail[0] = __builtin_bswap32((uint32_t)(bits >> 32));
        tail[1] = __builtin_bswap32((uint32_t)bits);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  };

  if (max_nblk) { fetch(0); commit(0); }
  for (uint32_t blk0 = 0; blk0 < max_nblk; blk0 += T / 64) {
    const uint32_t next = blk0 + T / 64;
    const bool more = next < max_nblk;
    if (more) fetch(next);                     // in flight during the compression below
    // ---- compress: one message per lane
#pragma unroll 1
    for (int b = 0; b < T / 64; b++) {
      if (blk0 + b < my_nblk) {
        uint32_t w[16];
        const uint4* src = (const uint4*)(my_row + b * 64);
#pragma unroll
        for (int q = 0; q < 4; q++) {
          uint4 v = src[q];
          w[4 * q + 0] = __builtin_bswap32(v.x); w[4 * q + 1] = __builtin_bswap32(v.y);
          w[4 * q + 2] = __builtin_bswap32(v.z); w[4 * q + 3] = __builtin_bswap32(v.w);
        }
        if (any_sha1 && my_algo) sha1_compress(st, w); else sha256_compress(st, w);
      }
    }
    __builtin_amdgcn_wave_barrier();
    if (more) commit(next);
  }
  if (m < n && my_dst) {
    gptr_out32 out = (gptr_out32)my_dst;      // digests are 4-byte aligned (result records / engine buffers)
#pragma unroll
    for (int i = 0; i < 8; i++) out[i] = (my_algo && i >= 5) ? 0u : __builtin_bswap32(st[i]);
  }
}

template <int T>
constexpr size_t sha256_lds_bytes() { return 4 * (64 * (T + 16) + 64 * 16); }

// ---------------------------------------------------------------------------------------------------------------
// Two waves per 64 messages, for launches with few messages (the BASELINE batch: 1 024 bodies = 16 groups on a
// chip with 1 024 SIMDs).  Such a launch lasts as long as ONE wave needs for the longest message's chain of
// compressions, and a lone wave issues one instruction every ~4.3 cycles however independent its instructions
// are — so the only way to shorten the chain is to put fewer instructions on it.  The message schedule
// (W[16..63], ~480 VALU per block) does not depend on the chaining state: wave 0 (feeder) fetches the bytes,
// builds K[t] + W[t] for block k+1 and leaves the 64 words per lane in LDS while wave 1 (rounds) runs the 64
// rounds of block k from registers it filled from that buffer.  Two s_barriers per block; ~900 instead of ~1 400
// instructions on the critical path.  Groups that contain a SHA-1 job fall back to the one-wave routine.
// LDS per group: slab 64 x (T+16) + descriptors 1 KB + 64 lanes x 68 words of K+W (row stride 68 words = 4 banks
// mod 64: conflict-free 16-byte writes and reads, a quarter of the LDS instructions of word accesses).
constexpr uint32_t SHA_K[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5,
    0xd807aa98, 0x12835b01, 0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174,
    0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da,
    0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967,
    0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
    0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070,
    0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3,
    0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};

constexpr int SHA_KW_ROW = 68;        // dwords per lane and buffer

template <int T>
constexpr size_t sha256_pair_lds_bytes() { return 64 * (T + 16) + 64 * 16 + 64 * SHA_KW_ROW * 4 + 1024; }      // slab, descriptors, K+W, job pick

// The job of each of a group's 64 lanes under length buckets (see ShaOrder): lane l of group g takes the message at global
// position 64 g + l of the longest-first order.  `pick`: 1 KB of LDS (192 class bases + 64 picks), written and read by the
// calling wave only.  Returns the lane's job index within the kind (SHA_KEY_NONE: no message), or `direct` when the batch is
// uniform enough for the direct mapping.  *skip = the whole group has nothing to do.
__device__ __forceinline__ uint32_t sha_pick_job(const ShaOrder& O, uint32_t g, uint32_t direct, uint32_t* pick, bool& skip) {
  const uint32_t lane = threadIdx.x & 63;
  skip = false;
  // class bases, longest class first: base[c] = messages in classes above c
  uint32_t c3[3], tot3 = 0;
#pragma unroll
  for (int k = 0; k < 3; k++) { c3[k] = O.cnt[3 * lane + k]; tot3 += c3[k]; }
  const uint64_t nonempty = __ballot(tot3 != 0);
  if (!nonempty) { skip = true; return SHA_KEY_NONE; }
  {
    // uniform enough?  the non-empty classes span at most two neighbouring ones
    const uint32_t lo_lane = (uint32_t)__builtin_ctzll(nonempty), hi_lane = 63u - (uint32_t)__builtin_clzll(nonempty);
    uint32_t lo_c = 0xFFFFFFFFu, hi_c = 0;
#pragma unroll
    for (int k = 0; k < 3; k++) if (c3[k]) { lo_c = min(lo_c, 3 * lane + k); hi_c = max(hi_c, 3 * lane + k); }
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)lo_c, (int)lo_lane), hi = (uint32_t)__builtin_amdgcn_readlane((int)hi_c, (int)hi_lane);
    if (hi - lo <= 1) return direct;
  }
  uint32_t above = tot3;                                   // inclusive suffix sum over the lanes: messages in this lane's classes and above
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { const uint32_t t = (uint32_t)__shfl_down((int)above, o); if (lane + o < 64) above += t; }
  const uint32_t total = (uint32_t)__builtin_amdgcn_readfirstlane((int)above);
  if (64 * g >= total) { skip = true; return SHA_KEY_NONE; }
  uint32_t b = above - tot3;                               // messages in the classes above this lane's
  pick[3 * lane + 2] = b; b += c3[2];
  pick[3 * lane + 1] = b; b += c3[1];
  pick[3 * lane + 0] = b;
  pick[SHA_CLASSES + lane] = SHA_KEY_NONE;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  const uint32_t lo = 64 * g;
  for (uint32_t t0 = 0; t0 < O.n; t0 += 256) {             // four keys per lane and step: the loads of a step are in flight together
    uint32_t k4[4];
#pragma unroll
    for (int q = 0; q < 4; q++) { const uint32_t i = t0 + 64 * q + lane; k4[q] = i < O.n ? O.key[i] : SHA_KEY_NONE; }
#pragma unroll
    for (int q = 0; q < 4; q++) {
      if (k4[q] != SHA_KEY_NONE) {
        const uint32_t gp = pick[k4[q] >> 24] + (k4[q] & 0xFFFFFFu) - lo;
        if (gp < 64) pick[SHA_CLASSES + gp] = t0 + 64 * q + lane;
      }
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  return pick[SHA_CLASSES + lane];
}

// The two waves of workgroup-local threads 0..127 hash messages [64 group, 64 group + 64); lds_raw: sha256_pair_lds_bytes<T>()
// of 16-byte aligned LDS.  A device routine so that other work can share the launch (fused.hip.h); both waves of the
// workgroup must call it (it uses __syncthreads()).
// `order`: length buckets of the jobs [0, n) this call sees (then `jobs` / `n` / `group` are the kind's), or nullptr.
template <int T>
__device__ __forceinline__ void sha256_pair_group(const ShaJob* __restrict__ jobs, uint32_t n, uint32_t group, uint8_t* lds_raw,
                                                  const ShaOrder* order = nullptr) {
  static_assert(T % 64 == 0 && T >= 64 && T <= 1024, "tile must be whole SHA blocks");
  constexpr int ROW = T + 16;
  constexpr int LPR = T / 16;
  constexpr int RPI = 64 / LPR;
#ifndef ZKE_SHA_PRIO
#define ZKE_SHA_PRIO 3
#endif
  __builtin_amdgcn_s_setprio(ZKE_SHA_PRIO);
  const int lane = threadIdx.x & 63;
  const int role = threadIdx.x >> 6;               // 0 feeder, 1 rounds
  uint8_t* slab = lds_raw;
  uint8_t* desc = slab + 64 * ROW;
  uint32_t* kw = (uint32_t*)(desc + 64 * 16);      // [64 lanes][SHA_KW_ROW]: a lane's 64 words are contiguous (b128 accesses)
  uint32_t* pick = kw + 64 * SHA_KW_ROW;           // 1 KB: sha_pick_job

  uint32_t m = group * 64 + lane;                  // both waves look at the same 64 jobs
  if (order && order->cnt) {
    // length buckets: the feeder works the picks out, the rounds wave reads them (one barrier; every thread of the group
    // arrives here, and a group with nothing to do leaves as a whole)
    bool skip = false;
    uint32_t mine = 0;
    if (role == 0) {
      mine = sha_pick_job(*order, group, m, pick, skip);
      if (lane == 0) pick[SHA_CLASSES - 1] = skip ? 1u : 0u;          // (class 191 = 2^25 blocks and more: never populated)
      pick[SHA_CLASSES + lane] = mine;
    }
    __syncthreads();
    if (pick[SHA_CLASSES - 1]) return;
    m = pick[SHA_CLASSES + lane];
    __syncthreads();                               // the pick area is dead from here (nobody writes it again)
  }
  uint64_t my_src = 0, my_dst = 0;
  uint32_t my_len = 0, my_nblk = 0, my_algo = 0;
  if (m < n) {
    ShaJob j = jobs[m];
    my_src = j.src; my_dst = j.dst; my_len = j.len; my_algo = j.pad;
    my_nblk = my_dst ? (my_len + 9 + 63) >> 6 : 0;
  }
  uint32_t max_nblk = my_nblk;
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) max_nblk = max(max_nblk, (uint32_t)__shfl_xor((int)max_nblk, o));
  max_nblk = __builtin_amdgcn_readfirstlane(max_nblk);
  const bool any_sha1 = __ballot(my_algo != 0) != 0;          // the same in both waves: they read the same jobs

  uint32_t st[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
  uint8_t* my_row = slab + lane * ROW;

  if (role == 0) {
    *(uint64_t*)(desc + lane * 16) = my_src;
    *(uint32_t*)(desc + lane * 16 + 8) = my_len;
    *(uint32_t*)(desc + lane * 16 + 12) = my_nblk;
  }
  uint4 stage[LPR];
  uint32_t live = 0;
  auto fetch = [&](uint32_t blk0) {
    const uint32_t tile_off = blk0 * 64;
    live = 0;
    uint4 dsc[LPR];                              // every row descriptor first (one ds_read_b128 each), then the loads
#pragma unroll
    for (int g = 0; g < LPR; g++) dsc[g] = *(const uint4*)(desc + (g * RPI + lane / LPR) * 16);
#pragma unroll
    for (int g = 0; g < LPR; g++) {
      const int chunk = lane % LPR;
      const uint64_t rsrc = ((uint64_t)dsc[g].y << 32) | dsc[g].x;
      const uint32_t rlen = dsc[g].z;
      const uint32_t rnblk = dsc[g].w;
      const uint32_t off = tile_off + chunk * 16;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (blk0 < rnblk) {
        live |= 1u << g;
        if (off + 16 <= rlen) {
          { const u32x4_raw t4 = *(gptr_u4)(rsrc + off); v = make_uint4(t4.x, t4.y, t4.z, t4.w); }
        } else if (off < rlen) {
          gptr_u8 p = (gptr_u8)(rsrc + off);
          uint32_t rem = rlen - off, t[4] = {0, 0, 0, 0};
          for (uint32_t b = 0; b < rem; b++) t[b >> 2] |= (uint32_t)p[b] << (8 * (b & 3));
          v = make_uint4(t[0], t[1], t[2], t[3]);
        }
      }
      stage[g] = v;
    }
  };
  auto commit = [&](uint32_t blk0) {
    const uint32_t tile_off = blk0 * 64;
#pragma unroll
    for (int g = 0; g < LPR; g++) {
      const int row = g * RPI + lane / LPR;
      const int chunk = lane % LPR;
      if (live & (1u << g)) *(uint4*)(slab + row * ROW + chunk * 16) = stage[g];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (blk0 < my_nblk) {
      if (my_len >= tile_off && my_len < tile_off + T) my_row[my_len - tile_off] = 0x80;
      const uint32_t last = my_nblk - 1;
      if (last >= blk0 && last < blk0 + T / 64) {
        const uint64_t bits = (uint64_t)my_len * 8;
        uint32_t* tail = (uint32_t*)(my_row + (last - blk0) * 64 + 56);
        tail[0] = __builtin_bswap32((uint32_t)(bits >> 32));
        tail[1] = __builtin_bswap32((uint32_t)bits);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  };

  if (any_sha1) {
    // one-wave routine (as sha256_batch_kernel) by the feeder; the other wave has nothing to do.  No barrier below.
    if (role != 0) return;
    if (my_algo) { st[0] = 0x67452301; st[1] = 0xEFCDAB89; st[2] = 0x98BADCFE; st[3] = 0x10325476; st[4] = 0xC3D2E1F0; st[5] = st[6] = st[7] = 0; }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (max_nblk) { fetch(0); commit(0); }
    for (uint32_t blk0 = 0; blk0 < max_nblk; blk0 += T / 64) {
      const uint32_t next = blk0 + T / 64;
      const bool more = next < max_nblk;
      if (more) fetch(next);
#pragma unroll 1
      for (int b = 0; b < T / 64; b++) {
        if (blk0 + b < my_nblk) {
          uint32_t w[16];
          const uint4* src = (const uint4*)(my_row + b * 64);
#pragma unroll
          for (int q = 0; q < 4; q++) {
            uint4 v = src[q];
            w[4 * q + 0] = __builtin_bswap32(v.x); w[4 * q + 1] = __builtin_bswap32(v.y);
            w[4 * q + 2] = __builtin_bswap32(v.z); w[4 * q + 3] = __builtin_bswap32(v.w);
          }
          if (my_algo) sha1_compress(st, w); else sha256_compress(st, w);
        }
      }
      __builtin_amdgcn_wave_barrier();
      if (more) commit(next);
    }
  } else {
    // feeder: K+W of block kb into kw[kb & 1]
    auto sched = [&](uint32_t kb) {
      uint32_t w[16];
      const uint4* src = (const uint4*)(my_row + (kb % (T / 64)) * 64);
#pragma unroll
      for (int q = 0; q < 4; q++) {
        uint4 v = src[q];
        w[4 * q + 0] = __builtin_bswap32(v.x); w[4 * q + 1] = __builtin_bswap32(v.y);
        w[4 * q + 2] = __builtin_bswap32(v.z); w[4 * q + 3] = __builtin_bswap32(v.w);
      }
      uint4* dst = (uint4*)(kw + lane * SHA_KW_ROW);
      uint32_t o4[4];
#pragma unroll
      for (int i = 0; i < 64; i++) {
        uint32_t wi;
        if (i < 16) {
          wi = w[i];
        } else {
          const uint32_t w15 = w[(i - 15) & 15], w2 = w[(i - 2) & 15];
          const uint32_t s0 = xor3(rotr32(w15, 7), rotr32(w15, 18), w15 >> 3);
          const uint32_t s1 = xor3(rotr32(w2, 17), rotr32(w2, 19), w2 >> 10);
          wi = w[i & 15] + s0 + w[(i - 7) & 15] + s1;
          w[i & 15] = wi;
        }
        o4[i & 3] = wi + SHA_K[i];
        if ((i & 3) == 3) dst[i >> 2] = make_uint4(o4[0], o4[1], o4[2], o4[3]);
      }
    };
    if (role == 0 && max_nblk) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      fetch(0); commit(0);
      if (T / 64 < max_nblk) fetch(T / 64);
      sched(0);
    }
    __syncthreads();
    // Both waves run exactly max_nblk steps (same jobs, same maximum).  ONE K+W buffer (a group then needs 28 KB of
    // LDS instead of 45 KB and finds room on a CU that front-end waves of other batches have nearly filled): the
    // rounds wave pulls the block's 64 words into registers, barrier, then the feeder overwrites the buffer with the
    // next block's while the rounds run from registers, barrier.
    for (uint32_t kb = 0; kb < max_nblk; kb++) {
      uint4 kq[16];
      if (role == 1) {
        const uint4* src = (const uint4*)(kw + lane * SHA_KW_ROW);
#pragma unroll
        for (int q = 0; q < 16; q++) kq[q] = src[q];
      }
      __syncthreads();                                // the buffer has been read
      if (role == 0) {
        const uint32_t nb = kb + 1;
        if (nb < max_nblk) {
          if (nb % (T / 64) == 0) {                  // first block of the next tile: its bytes were fetched a tile ago
            commit(nb);
            if (nb + T / 64 < max_nblk) fetch(nb + T / 64);
          }
          sched(nb);
        }
      } else {
        uint32_t a = st[0], b = st[1], c = st[2], d = st[3], e = st[4], f = st[5], g = st[6], h = st[7];
#pragma unroll
        for (int i = 0; i < 64; i++) {
          const uint32_t S1 = xor3(rotr32(e, 6), rotr32(e, 11), rotr32(e, 25));
          const uint4 k4 = kq[i >> 2];
          const uint32_t kwi = (i & 3) == 0 ? k4.x : (i & 3) == 1 ? k4.y : (i & 3) == 2 ? k4.z : k4.w;
          const uint32_t t1 = (h + S1 + ch3(e, f, g)) + kwi;
          const uint32_t S0 = xor3(rotr32(a, 2), rotr32(a, 13), rotr32(a, 22));
          const uint32_t mj = maj3(a, b, c);
          h = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + S0 + mj;
        }
        if (kb < my_nblk) { st[0] += a; st[1] += b; st[2] += c; st[3] += d; st[4] += e; st[5] += f; st[6] += g; st[7] += h; }
      }
      __syncthreads();                                // the next block's words are in the buffer
    }
    if (role == 0) return;
  }
  if (m < n && my_dst) {
    gptr_out32 out = (gptr_out32)my_dst;
#pragma unroll
    for (int i = 0; i < 8; i++) out[i] = (my_algo && i >= 5) ? 0u : __builtin_bswap32(st[i]);
  }
}

template <int T>
__global__ __launch_bounds__(128) void sha256_pair_kernel(const ShaJob* __restrict__ jobs, uint32_t n) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds_raw[];
  sha256_pair_group<T>(jobs, n, blockIdx.x, lds_raw);
}

}  // namespace zke
