// ed25519.hip.h — the `k=ed25519` / `a=ed25519-sha256` branch of cfdkim's verify_signature (ed25519-dalek 2.1.1,
// Cargo.lock:778; key bytes from helpers/src/dkim.rs:53-56,103-108), on the device.
//
// Mapping: FOUR LANES PER SIGNATURE (ed25519_verify_quad below).  Unlike RSA (one 2048-bit Montgomery product = 64 x 64
// limb products, spread over a wave) a curve25519 field product is 8 x 8 limbs — too small to spread — so field
// arithmetic is per lane and the four independent products of each level of the point formulas go to the four lanes
// of a DPP quad.  The wave never diverges on data: the scalar loop is bit-serial over two fixed 256-bit scalars with
// complete (unified) addition formulas, selections are v_cndmask, and the only branch is the early-out a wave with
// nothing to verify takes as a whole.
// Integer work throughout (v_mad_u64_u32); no memory traffic beyond the 32 + 64 + 32 input bytes per signature.
//
// Field elements: eight 32-bit limbs holding ANY 256-bit value, read modulo p = 2^255 - 19 (2^256 = 38 mod p);
// reduced fully only to encode or to compare.
//
// Acceptance rule = dalek `verify_strict` (restated in oracle/zke_ed25519.c, checked against RFC 8032 vectors):
//   A decompresses (y taken mod p, x = 0 with the sign bit set accepted as 0), S < L, R decompresses,
//   neither A nor R of small order, compress([S]B - [k]A) == R bytes, k = SHA-512(R || A || M) mod L.
#pragma once
#include <stdint.h>
#include <hip/hip_runtime.h>
#include "canon.hip.h"

namespace zke {

struct Fe { uint32_t v[8]; };

#define ZKE_ED __device__ __forceinline__
#define ZKE_ED_CALL __device__ __noinline__      // the two field products are real calls: ~60 call sites otherwise

ZKE_ED Fe fe_small(uint32_t x) { Fe r; r.v[0] = x; for (int i = 1; i < 8; i++) r.v[i] = 0; return r; }

// r + 38*c folded back in (c < 2^32)
ZKE_ED void fe_fold(Fe& r, uint32_t c) {
  uint64_t x = (uint64_t)c * 38u + r.v[0];
  r.v[0] = (uint32_t)x;
  uint32_t cy = (uint32_t)(x >> 32);
#pragma unroll
  for (int j = 1; j < 8; j++) { const uint64_t y = (uint64_t)r.v[j] + cy; r.v[j] = (uint32_t)y; cy = (uint32_t)(y >> 32); }
  // a second wrap leaves a value below 2^38 (limbs 2..7 zero, limb 1 < 64): one more 38 stays inside limbs 0 and 1
  const uint64_t z = (uint64_t)r.v[0] + 38u * cy;
  r.v[0] = (uint32_t)z; r.v[1] += (uint32_t)(z >> 32);
}

ZKE_ED Fe fe_add(const Fe& a, const Fe& b) {
  Fe r;
  uint32_t cy = 0;
#pragma unroll
  for (int j = 0; j < 8; j++) { const uint64_t x = (uint64_t)a.v[j] + b.v[j] + cy; r.v[j] = (uint32_t)x; cy = (uint32_t)(x >> 32); }
  fe_fold(r, cy);
  return r;
}

ZKE_ED Fe fe_sub(const Fe& a, const Fe& b) {
  Fe r;
  uint32_t bw = 0;
#pragma unroll
  for (int j = 0; j < 8; j++) {
    const uint64_t x = (uint64_t)a.v[j] - b.v[j] - bw;
    r.v[j] = (uint32_t)x; bw = (uint32_t)(x >> 63);
  }
  // a - b + 2^256 = a - b + 38 (mod p): take the 38 back out, twice at most
  uint32_t bw2 = 0;
  {
    const uint64_t x = (uint64_t)r.v[0] - 38u * bw;
    r.v[0] = (uint32_t)x; bw2 = (uint32_t)(x >> 63);
#pragma unroll
    for (int j = 1; j < 8; j++) { const uint64_t y = (uint64_t)r.v[j] - bw2; r.v[j] = (uint32_t)y; bw2 = (uint32_t)(y >> 63); }
  }
  r.v[0] -= 38u * bw2;          // after a second wrap the value is within 38 of 2^256: no further borrow
  return r;
}

// (the bodies inline: the scalar-multiplication loop of ed25519_verify_quad uses them without a call; everything else —
// ~60 sites in the square-root and inversion chains — goes through the two real calls below)
ZKE_ED Fe fe_mul_i(const Fe& a, const Fe& b) {
  uint32_t t[16];
#pragma unroll
  for (int i = 0; i < 16; i++) t[i] = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    uint32_t cy = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const uint64_t x = (uint64_t)a.v[i] * b.v[j] + t[i + j] + cy;      // <= 2^64 - 1
      t[i + j] = (uint32_t)x; cy = (uint32_t)(x >> 32);
    }
    t[i + 8] = cy;
  }
  Fe r;
  uint32_t cy = 0;
#pragma unroll
  for (int j = 0; j < 8; j++) {
    const uint64_t x = (uint64_t)t[8 + j] * 38u + t[j] + cy;
    r.v[j] = (uint32_t)x; cy = (uint32_t)(x >> 32);
  }
  fe_fold(r, cy);
  return r;
}

// a^2: the 28 cross products once, doubled by a one-bit shift of the 512-bit sum, plus the 8 squares — 36 multiplies
// instead of 64
ZKE_ED Fe fe_sq_i(const Fe& a) {
  uint32_t t[16];
#pragma unroll
  for (int i = 0; i < 16; i++) t[i] = 0;
#pragma unroll
  for (int i = 0; i < 7; i++) {
    uint32_t cy = 0;
#pragma unroll
    for (int j = i + 1; j < 8; j++) {
      const uint64_t x = (uint64_t)a.v[i] * a.v[j] + t[i + j] + cy;
      t[i + j] = (uint32_t)x; cy = (uint32_t)(x >> 32);
    }
    t[i + 8] = cy;
  }
#pragma unroll
  for (int k = 15; k > 0; k--) t[k] = (t[k] << 1) | (t[k - 1] >> 31);      // the sum of cross products is < 2^511
  t[0] <<= 1;
  uint32_t cy = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    const uint64_t x = (uint64_t)a.v[i] * a.v[i] + t[2 * i] + cy;
    t[2 * i] = (uint32_t)x;
    const uint64_t y = (x >> 32) + t[2 * i + 1];
    t[2 * i + 1] = (uint32_t)y; cy = (uint32_t)(y >> 32);
  }
  Fe r;
  cy = 0;
#pragma unroll
  for (int j = 0; j < 8; j++) {
    const uint64_t x = (uint64_t)t[8 + j] * 38u + t[j] + cy;
    r.v[j] = (uint32_t)x; cy = (uint32_t)(x >> 32);
  }
  fe_fold(r, cy);
  return r;
}
ZKE_ED_CALL Fe fe_mul(Fe a, Fe b) { return fe_mul_i(a, b); }
ZKE_ED_CALL Fe fe_sq(Fe a) { return fe_sq_i(a); }
ZKE_ED Fe fe_sqn(Fe a, int n) {
#pragma unroll 1
  for (int i = 0; i < n; i++) a = fe_sq(a);
  return a;
}

// fully reduced value in [0, p)
ZKE_ED Fe fe_canon(const Fe& a) {
  Fe r = a;
  // at most two subtractions of p bring any 256-bit value below p
#pragma unroll 1
  for (int pass = 0; pass < 2; pass++) {
    // t = r - p = r + 19 - 2^255
    Fe t;
    uint32_t cy = 19;
#pragma unroll
    for (int j = 0; j < 8; j++) { const uint64_t x = (uint64_t)r.v[j] + cy; t.v[j] = (uint32_t)x; cy = (uint32_t)(x >> 32); }
    // r >= p  <=>  r + 19 >= 2^255  <=>  carry out or bit 255 of t
    const bool ge = cy || (t.v[7] >> 31);
    // r + 19 - 2^255: with a carry out (r + 19 >= 2^256) bit 255 of t is the true bit 255 minus... handle by value:
    // r < 2^256, so r + 19 - 2^255 fits 256 bits; subtracting 2^255 flips bit 255 and, if it was clear, borrows from the carry
    t.v[7] ^= 0x80000000u;
    if (ge) r = t;
  }
  return r;
}
ZKE_ED bool fe_is_zero(const Fe& a) {
  const Fe c = fe_canon(a);
  uint32_t o = 0;
#pragma unroll
  for (int j = 0; j < 8; j++) o |= c.v[j];
  return o == 0;
}
ZKE_ED bool fe_eq(const Fe& a, const Fe& b) { return fe_is_zero(fe_sub(a, b)); }
ZKE_ED bool fe_is_neg(const Fe& a) { return (fe_canon(a).v[0] & 1u) != 0; }
ZKE_ED Fe fe_neg(const Fe& a) { return fe_sub(fe_small(0), a); }
ZKE_ED Fe fe_select(bool c, const Fe& a, const Fe& b) {
  Fe r;
#pragma unroll
  for (int j = 0; j < 8; j++) r.v[j] = c ? a.v[j] : b.v[j];
  return r;
}

// z^(2^252 - 3)
ZKE_ED Fe fe_pow22523(const Fe& z) {
  Fe t0 = fe_sq(z);                       // 2
  Fe t1 = fe_mul(z, fe_sqn(t0, 2));       // 9
  t0 = fe_mul(t0, t1);                    // 11
  t0 = fe_mul(t1, fe_sq(t0));             // 31 = 2^5 - 1
  t0 = fe_mul(fe_sqn(t0, 5), t0);         // 2^10 - 1
  t1 = fe_mul(fe_sqn(t0, 10), t0);        // 2^20 - 1
  t1 = fe_mul(fe_sqn(t1, 20), t1);        // 2^40 - 1
  t0 = fe_mul(fe_sqn(t1, 10), t0);        // 2^50 - 1
  t1 = fe_mul(fe_sqn(t0, 50), t0);        // 2^100 - 1
  t1 = fe_mul(fe_sqn(t1, 100), t1);       // 2^200 - 1
  t0 = fe_mul(fe_sqn(t1, 50), t0);        // 2^250 - 1
  return fe_mul(fe_sqn(t0, 2), z);        // 2^252 - 3
}
ZKE_ED Fe fe_invert(const Fe& z) {        // z^(p-2) = (z^(2^252-3))^8 * z^3
  const Fe t = fe_sqn(fe_pow22523(z), 3);
  return fe_mul(t, fe_mul(fe_sq(z), z));
}

ZKE_ED Fe fe_from_bytes(const uint8_t* s) {      // bit 255 dropped; NOT required to be < p
  Fe r;
#pragma unroll
  for (int j = 0; j < 8; j++)
    r.v[j] = (uint32_t)s[4 * j] | ((uint32_t)s[4 * j + 1] << 8) | ((uint32_t)s[4 * j + 2] << 16) | ((uint32_t)s[4 * j + 3] << 24);
  r.v[7] &= 0x7fffffffu;
  return r;
}

// d, 2d, sqrt(-1)
ZKE_ED Fe fe_d() { return Fe{{0x135978a3u, 0x75eb4dcau, 0x4141d8abu, 0x00700a4du, 0x7779e898u, 0x8cc74079u, 0x2b6ffe73u, 0x52036ceeu}}; }
ZKE_ED Fe fe_2d() { return Fe{{0x26b2f159u, 0xebd69b94u, 0x8283b156u, 0x00e0149au, 0xeef3d130u, 0x198e80f2u, 0x56dffce7u, 0x2406d9dcu}}; }
ZKE_ED Fe fe_sqrtm1() { return Fe{{0x4a0ea0b0u, 0xc4ee1b27u, 0xad2fe478u, 0x2f431806u, 0x3dfbd7a7u, 0x2b4d0099u, 0x4fc1df0bu, 0x2b832480u}}; }

struct Ge { Fe X, Y, Z, T; };
ZKE_ED Ge ge_identity() { return Ge{fe_small(0), fe_small(1), fe_small(1), fe_small(0)}; }

// unified addition, a = -1 (add-2008-hwcd-3): complete on this curve — also doubles, also adds the identity
ZKE_ED Ge ge_add(const Ge& p, const Ge& q) {
  const Fe a = fe_mul(fe_sub(p.Y, p.X), fe_sub(q.Y, q.X));
  const Fe b = fe_mul(fe_add(p.Y, p.X), fe_add(q.Y, q.X));
  const Fe c = fe_mul(fe_mul(p.T, q.T), fe_2d());
  const Fe zz = fe_mul(p.Z, q.Z);
  const Fe d = fe_add(zz, zz);
  const Fe e = fe_sub(b, a), f = fe_sub(d, c), g = fe_add(d, c), h = fe_add(b, a);
  return Ge{fe_mul(e, f), fe_mul(g, h), fe_mul(f, g), fe_mul(e, h)};
}
// doubling, a = -1 (dbl-2008-hwcd): 4 squarings + 4 products
ZKE_ED Ge ge_dbl(const Ge& p) {
  const Fe a = fe_sq(p.X), b = fe_sq(p.Y);
  const Fe zz = fe_sq(p.Z);
  const Fe c = fe_add(zz, zz);
  const Fe xy = fe_add(p.X, p.Y);
  const Fe e = fe_sub(fe_sub(fe_sq(xy), a), b);
  const Fe g = fe_sub(b, a);                 // D + B with D = -A
  const Fe f = fe_sub(g, c);
  const Fe h = fe_sub(fe_neg(a), b);         // D - B
  return Ge{fe_mul(e, f), fe_mul(g, h), fe_mul(f, g), fe_mul(e, h)};
}
// addition of a table point whose 2d*T is known (one product fewer)
ZKE_ED Ge ge_add_cached(const Ge& p, const Ge& q, const Fe& q_t2d) {
  const Fe a = fe_mul(fe_sub(p.Y, p.X), fe_sub(q.Y, q.X));
  const Fe b = fe_mul(fe_add(p.Y, p.X), fe_add(q.Y, q.X));
  const Fe c = fe_mul(p.T, q_t2d);
  const Fe zz = fe_mul(p.Z, q.Z);
  const Fe d = fe_add(zz, zz);
  const Fe e = fe_sub(b, a), f = fe_sub(d, c), g = fe_add(d, c), h = fe_add(b, a);
  return Ge{fe_mul(e, f), fe_mul(g, h), fe_mul(f, g), fe_mul(e, h)};
}
ZKE_ED Ge ge_neg(const Ge& p) { return Ge{fe_neg(p.X), p.Y, p.Z, fe_neg(p.T)}; }
ZKE_ED Ge ge_select(bool c, const Ge& a, const Ge& b) {
  return Ge{fe_select(c, a.X, b.X), fe_select(c, a.Y, b.Y), fe_select(c, a.Z, b.Z), fe_select(c, a.T, b.T)};
}

// curve25519-dalek CompressedEdwardsY::decompress
ZKE_ED bool ge_decompress(Ge& p, const uint8_t* s) {
  const Fe y = fe_from_bytes(s);
  const Fe yy = fe_sq(y);
  const Fe u = fe_sub(yy, fe_small(1));
  const Fe v = fe_add(fe_mul(yy, fe_d()), fe_small(1));
  const Fe v3 = fe_mul(fe_sq(v), v);
  Fe r = fe_mul(fe_mul(fe_sq(v3), v), u);          // u v^7
  r = fe_mul(fe_mul(fe_pow22523(r), v3), u);       // u v^3 (u v^7)^((p-5)/8)
  const Fe chk = fe_mul(fe_sq(r), v);
  bool ok = true;
  if (!fe_eq(chk, u)) {
    ok = fe_eq(chk, fe_neg(u));
    r = fe_mul(r, fe_sqrtm1());
  }
  if (fe_is_neg(r)) r = fe_neg(r);
  if (s[31] >> 7) r = fe_neg(r);
  p = Ge{r, y, fe_small(1), fe_mul(r, y)};
  return ok;
}
ZKE_ED void ge_compress(uint32_t out[8], const Ge& p) {
  const Fe zi = fe_invert(p.Z);
  const Fe x = fe_mul(p.X, zi);
  const Fe y = fe_canon(fe_mul(p.Y, zi));
#pragma unroll
  for (int j = 0; j < 8; j++) out[j] = y.v[j];
  out[7] |= fe_is_neg(x) ? 0x80000000u : 0u;
}
ZKE_ED bool ge_is_small_order(const Ge& p) {
  Ge q = ge_dbl(p);
  q = ge_dbl(q);
  q = ge_dbl(q);
  return fe_is_zero(q.X) && fe_eq(q.Y, q.Z);
}

// ------------------------------------------------------------------ scalars mod L
#define ZKE_ED_L {0x5cf5d3edu, 0x5812631au, 0xa2f79cd6u, 0x14def9deu, 0u, 0u, 0u, 0x10000000u}

ZKE_ED bool sc_lt_L(const uint32_t s[8]) {
  constexpr uint32_t ED_L[8] = ZKE_ED_L;
  bool lt = false, decided = false;
#pragma unroll
  for (int j = 7; j >= 0; j--) {
    const uint32_t l = ED_L[j];
    if (!decided && s[j] != l) { lt = s[j] < l; decided = true; }
  }
  return lt;
}
// 512-bit little-endian h mod L by binary long division (speed is irrelevant next to the scalar multiplication)
ZKE_ED void sc_reduce512(uint32_t out[8], const uint32_t h[16]) {
  constexpr uint32_t ED_L[8] = ZKE_ED_L;
  uint32_t r[8];
#pragma unroll
  for (int j = 0; j < 8; j++) r[j] = 0;
  uint32_t q[16];
#pragma unroll
  for (int j = 0; j < 16; j++) q[j] = h[j];
#pragma unroll 1
  for (int bit = 0; bit < 512; bit++) {
    {
      uint32_t cy = q[15] >> 31;                 // next bit of h, most significant first; q shifts left by one
#pragma unroll
      for (int j = 15; j > 0; j--) q[j] = (q[j] << 1) | (q[j - 1] >> 31);
      q[0] <<= 1;
#pragma unroll
      for (int j = 0; j < 8; j++) { const uint32_t nv = (r[j] << 1) | cy; cy = r[j] >> 31; r[j] = nv; }
      // r < 2L < 2^254 here (no bit leaves the eight words); subtract L when r >= L
      uint32_t t[8];
      uint32_t bw = 0;
#pragma unroll
      for (int j = 0; j < 8; j++) { const uint64_t x = (uint64_t)r[j] - ED_L[j] - bw; t[j] = (uint32_t)x; bw = (uint32_t)(x >> 63); }
      if (!bw) {
#pragma unroll
        for (int j = 0; j < 8; j++) r[j] = t[j];
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 8; j++) out[j] = r[j];
}

// ------------------------------------------------------------------ SHA-512 of one 128-byte block (FIPS 180-4 §6.4)
__device__ const uint64_t ED_K512[80] = {
  0x428a2f98d728ae22ULL, 0x7137449123ef65cdULL, 0xb5c0fbcfec4d3b2fULL, 0xe9b5dba58189dbbcULL, 0x3956c25bf348b538ULL,
  0x59f111f1b605d019ULL, 0x923f82a4af194f9bULL, 0xab1c5ed5da6d8118ULL, 0xd807aa98a3030242ULL, 0x12835b0145706fbeULL,
  0x243185be4ee4b28cULL, 0x550c7dc3d5ffb4e2ULL, 0x72be5d74f27b896fULL, 0x80deb1fe3b1696b1ULL, 0x9bdc06a725c71235ULL,
  0xc19bf174cf692694ULL, 0xe49b69c19ef14ad2ULL, 0xefbe4786384f25e3ULL, 0x0fc19dc68b8cd5b5ULL, 0x240ca1cc77ac9c65ULL,
  0x2de92c6f592b0275ULL, 0x4a7484aa6ea6e483ULL, 0x5cb0a9dcbd41fbd4ULL, 0x76f988da831153b5ULL, 0x983e5152ee66dfabULL,
  0xa831c66d2db43210ULL, 0xb00327c898fb213fULL, 0xbf597fc7beef0ee4ULL, 0xc6e00bf33da88fc2ULL, 0xd5a79147930aa725ULL,
  0x06ca6351e003826fULL, 0x142929670a0e6e70ULL, 0x27b70a8546d22ffcULL, 0x2e1b21385c26c926ULL, 0x4d2c6dfc5ac42aedULL,
  0x53380d139d95b3dfULL, 0x650a73548baf63deULL, 0x766a0abb3c77b2a8ULL, 0x81c2c92e47edaee6ULL, 0x92722c851482353bULL,
  0xa2bfe8a14cf10364ULL, 0xa81a664bbc423001ULL, 0xc24b8b70d0f89791ULL, 0xc76c51a30654be30ULL, 0xd192e819d6ef5218ULL,
  0xd69906245565a910ULL, 0xf40e35855771202aULL, 0x106aa07032bbd1b8ULL, 0x19a4c116b8d2d0c8ULL, 0x1e376c085141ab53ULL,
  0x2748774cdf8eeb99ULL, 0x34b0bcb5e19b48a8ULL, 0x391c0cb3c5c95a63ULL, 0x4ed8aa4ae3418acbULL, 0x5b9cca4f7763e373ULL,
  0x682e6ff3d6b2b8a3ULL, 0x748f82ee5defb2fcULL, 0x78a5636f43172f60ULL, 0x84c87814a1f0ab72ULL, 0x8cc702081a6439ecULL,
  0x90befffa23631e28ULL, 0xa4506cebde82bde9ULL, 0xbef9a3f7b2c67915ULL, 0xc67178f2e372532bULL, 0xca273eceea26619cULL,
  0xd186b8c721c0c207ULL, 0xeada7dd6cde0eb1eULL, 0xf57d4f7fee6ed178ULL, 0x06f067aa72176fbaULL, 0x0a637dc5a2c898a6ULL,
  0x113f9804bef90daeULL, 0x1b710b35131c471bULL, 0x28db77f523047d84ULL, 0x32caab7b40c72493ULL, 0x3c9ebe0a15c9bebcULL,
  0x431d67c49c100d4cULL, 0x4cc5d4becb3e42b6ULL, 0x597f299cfc657e2aULL, 0x5fcb6fab3ad6faecULL, 0x6c44198c4a475817ULL};

ZKE_ED uint64_t ror64(uint64_t x, int n) { return (x >> n) | (x << (64 - n)); }

// digest of R(32) || A(32) || M(mlen <= 32): one block, since 64 + 32 + 17 <= 128.  out: 16 little-endian words
ZKE_ED void sha512_ram(uint32_t out[16], const uint8_t* R, const uint8_t* A, const uint8_t* M, uint32_t mlen) {
  uint64_t w[16];
  const uint32_t total = 64 + mlen;
#pragma unroll
  for (int i = 0; i < 16; i++) {
    uint64_t v = 0;
#pragma unroll
    for (int b = 0; b < 8; b++) {
      const uint32_t pos = 8 * i + b;
      uint32_t byte;
      if (pos < 32) byte = R[pos];
      else if (pos < 64) byte = A[pos - 32];
      else if (pos < total) byte = M[pos - 64];
      else byte = (pos == total) ? 0x80u : 0u;
      v = (v << 8) | byte;
    }
    w[i] = v;
  }
  w[15] = (uint64_t)total * 8;
  uint64_t st[8] = {0x6a09e667f3bcc908ULL, 0xbb67ae8584caa73bULL, 0x3c6ef372fe94f82bULL, 0xa54ff53a5f1d36f1ULL,
                    0x510e527fade682d1ULL, 0x9b05688c2b3e6c1fULL, 0x1f83d9abfb41bd6bULL, 0x5be0cd19137e2179ULL};
  uint64_t a = st[0], b = st[1], c = st[2], d = st[3], e = st[4], f = st[5], g = st[6], h = st[7];
#pragma unroll 1
  for (int i0 = 0; i0 < 80; i0 += 16) {
#pragma unroll
    for (int k = 0; k < 16; k++) {
      if (i0) {
        const uint64_t w15 = w[(k + 1) & 15], w2 = w[(k + 14) & 15];
        const uint64_t s0 = ror64(w15, 1) ^ ror64(w15, 8) ^ (w15 >> 7);
        const uint64_t s1 = ror64(w2, 19) ^ ror64(w2, 61) ^ (w2 >> 6);
        w[k] = w[k] + s0 + w[(k + 9) & 15] + s1;
      }
      const uint64_t t1 = h + (ror64(e, 14) ^ ror64(e, 18) ^ ror64(e, 41)) + ((e & f) ^ (~e & g)) + ED_K512[i0 + k] + w[k];
      const uint64_t t2 = (ror64(a, 28) ^ ror64(a, 34) ^ ror64(a, 39)) + ((a & b) ^ (a & c) ^ (b & c));
      h = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
    }
  }
  st[0] += a; st[1] += b; st[2] += c; st[3] += d; st[4] += e; st[5] += f; st[6] += g; st[7] += h;
  // the digest is a byte string (big-endian words); the scalar reads it little-endian
#pragma unroll
  for (int i = 0; i < 8; i++) {
    const uint64_t v = st[i];
    out[2 * i] = __builtin_bswap32((uint32_t)(v >> 32));
    out[2 * i + 1] = __builtin_bswap32((uint32_t)v);
  }
}

// ------------------------------------------------------------------ verification of one signature by FOUR LANES
// A DPP quad per signature, 16 signatures per wave.  Field arithmetic stays per lane (an 8 x 8-limb product is too small
// to spread); what is spread is the POINT arithmetic: the extended-coordinate formulas are two levels of four independent
// products each, so lane q of the quad holds ONE coordinate (0 X, 1 Y, 2 Z, 3 T) and computes one product per level:
//   doubling   level 1: X^2 | Y^2 | Z^2 | (X+Y)^2          level 2: E*F | G*H | F*G | E*H
//   addition   level 1: (Y1-X1)(Y2-X2) | (Y1+X1)(Y2+X2) | T1*2dT2 | Z1*Z2     level 2: the same four
// with quad broadcasts (8 DPP moves per field element) and a few additions between the levels.  A signature's critical
// path is 4 products per scalar bit instead of 15, and a 1 024-signature batch is 64 waves instead of 16: the stage is
// latency-bound (the chip holds 1 024 waves of this size), so this is what shortens it.
//   * lane 0 decompresses A while lane 1 decompresses R (same instruction stream), each checks its point's order;
//   * the table entries B, -A, B - A are kept as the per-lane factor of the addition's level 1
//     (Y-X | Y+X | 2dT | Z), so a table point with any Z costs nothing extra;
//   * compress([S]B - [k]A) == R bytes is decided without the inversion: the R bytes must be the canonical encoding of
//     the decompressed R (y < p; the sign bit clear when x = 0 — compress never produces anything else), and then the
//     byte strings are equal iff the points are: X = x_R Z and Y = y_R Z.
// Same acceptance rule as above (dalek verify_strict); returns 0 = key does not decode, 1 = rejected, 2 = valid, the
// same value in the four lanes.  Lanes of a quad pass the same pointers.
template <int S> ZKE_ED uint32_t qb(uint32_t x) {                    // lane S of the quad to all four
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, S * 0x55 /*quad_perm:[S,S,S,S]*/, 0xf, 0xf, true);
}
template <int S> ZKE_ED Fe q_bcast(const Fe& a) {
  Fe r;
#pragma unroll
  for (int j = 0; j < 8; j++) r.v[j] = qb<S>(a.v[j]);
  return r;
}
ZKE_ED Fe q_swap23(const Fe& a) {                                    // lanes 2 and 3 exchange, 0 and 1 keep
  Fe r;
#pragma unroll
  for (int j = 0; j < 8; j++) r.v[j] = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)a.v[j], 0xB4 /*quad_perm:[0,1,3,2]*/, 0xf, 0xf, true);
  return r;
}
// level 2 of both formulas: lane 0 E*F, lane 1 G*H, lane 2 F*G, lane 3 E*H
ZKE_ED Fe q_finish(uint32_t q, const Fe& E, const Fe& F, const Fe& G, const Fe& H) {
  const Fe o1 = fe_select(q == 1, G, fe_select(q == 2, F, E));
  const Fe o2 = fe_select(q == 0, F, fe_select(q == 2, G, H));
  return fe_mul_i(o1, o2);
}
ZKE_ED Fe q_dbl(const Fe& c, uint32_t q) {                           // dbl-2008-hwcd, a = -1 (ge_dbl above)
  const Fe X = q_bcast<0>(c), Y = q_bcast<1>(c);
  const Fe s = fe_sq_i(fe_select(q == 3, fe_add(X, Y), c));
  const Fe A = q_bcast<0>(s), B = q_bcast<1>(s), ZZ = q_bcast<2>(s), S = q_bcast<3>(s);
  const Fe Hn = fe_add(A, B);
  const Fe G = fe_sub(B, A);
  return q_finish(q, fe_sub(S, Hn), fe_sub(G, fe_add(ZZ, ZZ)), G, fe_neg(Hn));
}
ZKE_ED Fe q_add(const Fe& c, const Fe& tab, uint32_t q) {            // ge_add_cached above; tab: this lane's level-1 factor
  const Fe X = q_bcast<0>(c), Y = q_bcast<1>(c);
  const Fe op = fe_select(q == 0, fe_sub(Y, X), fe_select(q == 1, fe_add(Y, X), q_swap23(c)));
  const Fe m = fe_mul_i(op, tab);
  const Fe a = q_bcast<0>(m), b = q_bcast<1>(m), cc = q_bcast<2>(m), zz = q_bcast<3>(m);
  const Fe d = fe_add(zz, zz);
  return q_finish(q, fe_sub(b, a), fe_sub(d, cc), fe_add(d, cc), fe_add(b, a));
}
ZKE_ED Fe q_table(const Ge& p, uint32_t q) {                         // Y - X | Y + X | 2d T | Z
  return fe_select(q == 0, fe_sub(p.Y, p.X), fe_select(q == 1, fe_add(p.Y, p.X), fe_select(q == 2, fe_mul(p.T, fe_2d()), p.Z)));
}
ZKE_ED uint32_t ed25519_verify_quad(const uint8_t* key, const uint8_t* msg, uint32_t mlen, const uint8_t* sig, bool have_sig) {
  const uint32_t q = threadIdx.x & 3u;
  const uint8_t* src = (q == 1) ? sig : key;
  Ge P;                                                              // lane 0 (2, 3): A; lane 1: R
  const bool okP = ge_decompress(P, src);
  const bool smallP = ge_is_small_order(P);
  // canonical encoding of this lane's point: y < p, and no sign bit on x = 0
  const Fe yc = fe_canon(P.Y);
  bool canonP = !(fe_is_zero(P.X) && (src[31] >> 7));
#pragma unroll
  for (int j = 0; j < 8; j++) canonP = canonP && yc.v[j] == P.Y.v[j];
  uint32_t S[8];
#pragma unroll
  for (int j = 0; j < 8; j++)
    S[j] = (uint32_t)sig[32 + 4 * j] | ((uint32_t)sig[33 + 4 * j] << 8) | ((uint32_t)sig[34 + 4 * j] << 16) | ((uint32_t)sig[35 + 4 * j] << 24);
  const bool okA = qb<0>(okP) != 0;
  const bool reject = !have_sig || !sc_lt_L(S) || !qb<1>(okP) || qb<0>(smallP) || qb<1>(smallP) || !qb<1>(canonP);
  const uint32_t early = !okA ? 0u : reject ? 1u : 3u;
  if (__ballot(early == 3u) == 0) return early;                      // nothing left to multiply in this wave
  uint32_t hw[16], k[8];
  sha512_ram(hw, sig, key, msg, mlen);
  sc_reduce512(k, hw);
  const Fe Ax = q_bcast<0>(P.X), Ay = q_bcast<0>(P.Y), At = q_bcast<0>(P.T);
  const Fe Rx = q_bcast<1>(P.X), Ry = q_bcast<1>(P.Y);
  // base point: y = 4/5, the even x
  const Fe by = Fe{{0x66666658u, 0x66666666u, 0x66666666u, 0x66666666u, 0x66666666u, 0x66666666u, 0x66666666u, 0x66666666u}};
  const Fe bx = Fe{{0x8f25d51au, 0xc9562d60u, 0x9525a7b2u, 0x692cc760u, 0xfdd6dc5cu, 0xc0a4e231u, 0xcd6e53feu, 0x216936d3u}};
  const Ge B = Ge{bx, by, fe_small(1), fe_mul(bx, by)};
  const Ge nA = Ge{fe_neg(Ax), Ay, fe_small(1), fe_neg(At)};
  const Fe tB = q_table(B, q), tnA = q_table(nA, q), tBnA = q_table(ge_add(B, nA), q);
  const Fe tId = fe_small(q == 2 ? 0u : 1u);
  Fe c = fe_small((q == 1 || q == 2) ? 1u : 0u);                     // the identity (0 : 1 : 1 : 0)
#pragma unroll 1
  for (int bit = 0; bit < 256; bit++) {
    c = q_dbl(c, q);
    const bool sb = (S[7] >> 31) != 0, kb = (k[7] >> 31) != 0;      // most significant bit first; both scalars shift left
#pragma unroll
    for (int j = 7; j > 0; j--) { S[j] = (S[j] << 1) | (S[j - 1] >> 31); k[j] = (k[j] << 1) | (k[j - 1] >> 31); }
    S[0] <<= 1; k[0] <<= 1;
    c = q_add(c, fe_select(kb, fe_select(sb, tBnA, tnA), fe_select(sb, tB, tId)), q);
  }
  const Fe Z = q_bcast<2>(c);
  const bool eq = fe_eq(fe_mul(fe_select(q == 0, Rx, Ry), Z), c);   // lane 0: x_R Z == X, lane 1: y_R Z == Y
  const bool same = qb<0>(eq) != 0 && qb<1>(eq) != 0;
  return early == 3u ? (same ? 2u : 1u) : early;
}

// building-block kernel: n independent (key, message, signature) triples, 32-byte messages, packed arrays; 16 per wave
__global__ __launch_bounds__(64) void ed25519_verify_kernel(const uint8_t* keys, const uint8_t* msgs, uint32_t msg_len,
                                                            const uint8_t* sigs, uint32_t n, uint32_t* out) {
  uint32_t i = blockIdx.x * 16 + (threadIdx.x >> 2);
  const bool live = i < n;
  if (!live) i = n - 1;                                              // whole quads stay in step; the result is dropped
  const uint32_t r = ed25519_verify_quad(keys + (size_t)i * 32, msgs + (size_t)i * msg_len, msg_len, sigs + (size_t)i * 64, true);
  if (live && (threadIdx.x & 3) == 0) out[i] = r;
}

}  // namespace zke
