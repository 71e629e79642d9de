/*
 * zke_ed25519.c — CPU oracle for the `k=ed25519` / `a=ed25519-sha256` branch.  TEST INFRASTRUCTURE ONLY
 * (see zke_oracle.h: only tests/, smoke() and bench.py's cpu_baseline leg may load the oracle library).
 *
 * PARITY UNPINNED by the reference (no fixture, no buildable Rust).  The branch lives in un-vendored crates:
 * cfdkim@75af99fb calls ed25519-dalek 2.1.1 (Cargo.lock:778) with the SHA-256 header hash as the message;
 * key bytes come from helpers/src/dkim.rs:53-56,103-108 (raw 32 bytes, `VerifyingKey::from_bytes`).
 * This file restates RFC 8032 §5.1 with the acceptance rule of dalek's `verify_strict`:
 *   - A = decompress(key) must succeed at key-decode time (y taken mod p; x = 0 with the sign bit set is -0 = 0);
 *   - S < L; R decompresses; neither A nor R has small order;
 *   - compress([S]B - [k]A) equals the 32 R bytes as transmitted, k = SHA-512(R || A || M) mod L.
 * Pinned by RFC 8032 §7.1 TEST 1, openssl-generated vectors (tests/golden/ed25519.json) and the Python-integer
 * implementation in zkemail.rs_amd/ed25519_ref.py, including small-order / non-canonical edge cases.
 */
#include "zke_oracle.h"

#include <string.h>

/* ------------------------------------------------------------------ SHA-512 (FIPS 180-4 §6.4) */
static const uint64_t K512[80] = {
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

static inline uint64_t ror64(uint64_t x, int n) { return (x >> n) | (x << (64 - n)); }

static void sha512_block(uint64_t st[8], const uint8_t* p) {
  uint64_t w[80];
  for (int i = 0; i < 16; i++) {
    uint64_t v = 0;
    for (int b = 0; b < 8; b++) v = (v << 8) | p[8 * i + b];
    w[i] = v;
  }
  for (int i = 16; i < 80; i++) {
    uint64_t s0 = ror64(w[i - 15], 1) ^ ror64(w[i - 15], 8) ^ (w[i - 15] >> 7);
    uint64_t s1 = ror64(w[i - 2], 19) ^ ror64(w[i - 2], 61) ^ (w[i - 2] >> 6);
    w[i] = w[i - 16] + s0 + w[i - 7] + s1;
  }
  uint64_t a = st[0], b = st[1], c = st[2], d = st[3], e = st[4], f = st[5], g = st[6], h = st[7];
  for (int i = 0; i < 80; i++) {
    uint64_t t1 = h + (ror64(e, 14) ^ ror64(e, 18) ^ ror64(e, 41)) + ((e & f) ^ (~e & g)) + K512[i] + w[i];
    uint64_t t2 = (ror64(a, 28) ^ ror64(a, 34) ^ ror64(a, 39)) + ((a & b) ^ (a & c) ^ (b & c));
    h = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
  }
  st[0] += a; st[1] += b; st[2] += c; st[3] += d; st[4] += e; st[5] += f; st[6] += g; st[7] += h;
}

void zko_sha512(const uint8_t* data, size_t len, uint8_t out[64]) {
  uint64_t st[8] = {0x6a09e667f3bcc908ULL, 0xbb67ae8584caa73bULL, 0x3c6ef372fe94f82bULL, 0xa54ff53a5f1d36f1ULL,
                    0x510e527fade682d1ULL, 0x9b05688c2b3e6c1fULL, 0x1f83d9abfb41bd6bULL, 0x5be0cd19137e2179ULL};
  size_t full = len / 128;
  for (size_t i = 0; i < full; i++) sha512_block(st, data + 128 * i);
  uint8_t tail[256];
  size_t rem = len - 128 * full;
  memset(tail, 0, sizeof tail);
  if (rem) memcpy(tail, data + 128 * full, rem);
  tail[rem] = 0x80;
  size_t tl = (rem + 17 <= 128) ? 128 : 256;
  uint64_t bits = (uint64_t)len * 8;                 /* lengths < 2^61 bytes: the upper 64 bits of the count are 0 */
  for (int b = 0; b < 8; b++) tail[tl - 1 - b] = (uint8_t)(bits >> (8 * b));
  sha512_block(st, tail);
  if (tl == 256) sha512_block(st, tail + 128);
  for (int i = 0; i < 8; i++)
    for (int b = 0; b < 8; b++) out[8 * i + b] = (uint8_t)(st[i] >> (56 - 8 * b));
}

/* ------------------------------------------------------------------ GF(2^255 - 19), five 51-bit limbs */
typedef struct { uint64_t v[5]; } fe;
typedef unsigned __int128 u128;
#define M51 ((1ULL << 51) - 1)

static const fe FE_D = {{0x34dca135978a3ULL, 0x1a8283b156ebdULL, 0x5e7a26001c029ULL, 0x739c663a03cbbULL, 0x52036cee2b6ffULL}};
static const fe FE_2D = {{0x69b9426b2f159ULL, 0x35050762add7aULL, 0x3cf44c0038052ULL, 0x6738cc7407977ULL, 0x2406d9dc56dffULL}};
static const fe FE_SQRTM1 = {{0x61b274a0ea0b0ULL, 0x0d5a5fc8f189dULL, 0x7ef5e9cbd0c60ULL, 0x78595a6804c9eULL, 0x2b8324804fc1dULL}};

static void fe_set(fe* r, uint64_t x) { r->v[0] = x; r->v[1] = r->v[2] = r->v[3] = r->v[4] = 0; }
static void fe_add(fe* r, const fe* a, const fe* b) { for (int i = 0; i < 5; i++) r->v[i] = a->v[i] + b->v[i]; }
/* a - b with 2p added first; inputs must have limbs < 2^52 */
static void fe_sub(fe* r, const fe* a, const fe* b) {
  r->v[0] = a->v[0] + 0xfffffffffffdaULL - b->v[0];
  for (int i = 1; i < 5; i++) r->v[i] = a->v[i] + 0xffffffffffffeULL - b->v[i];
}
static void fe_carry(fe* r) {
  uint64_t c;
  c = r->v[0] >> 51; r->v[0] &= M51; r->v[1] += c;
  c = r->v[1] >> 51; r->v[1] &= M51; r->v[2] += c;
  c = r->v[2] >> 51; r->v[2] &= M51; r->v[3] += c;
  c = r->v[3] >> 51; r->v[3] &= M51; r->v[4] += c;
  c = r->v[4] >> 51; r->v[4] &= M51; r->v[0] += 19 * c;
  c = r->v[0] >> 51; r->v[0] &= M51; r->v[1] += c;
}
static void fe_mul(fe* r, const fe* a, const fe* b) {
  u128 t[5];
  const uint64_t *x = a->v, *y = b->v;
  uint64_t y1 = 19 * y[1], y2 = 19 * y[2], y3 = 19 * y[3], y4 = 19 * y[4];
  t[0] = (u128)x[0] * y[0] + (u128)x[1] * y4 + (u128)x[2] * y3 + (u128)x[3] * y2 + (u128)x[4] * y1;
  t[1] = (u128)x[0] * y[1] + (u128)x[1] * y[0] + (u128)x[2] * y4 + (u128)x[3] * y3 + (u128)x[4] * y2;
  t[2] = (u128)x[0] * y[2] + (u128)x[1] * y[1] + (u128)x[2] * y[0] + (u128)x[3] * y4 + (u128)x[4] * y3;
  t[3] = (u128)x[0] * y[3] + (u128)x[1] * y[2] + (u128)x[2] * y[1] + (u128)x[3] * y[0] + (u128)x[4] * y4;
  t[4] = (u128)x[0] * y[4] + (u128)x[1] * y[3] + (u128)x[2] * y[2] + (u128)x[3] * y[1] + (u128)x[4] * y[0];
  uint64_t c;
  c = (uint64_t)(t[0] >> 51); r->v[0] = (uint64_t)t[0] & M51; t[1] += c;
  c = (uint64_t)(t[1] >> 51); r->v[1] = (uint64_t)t[1] & M51; t[2] += c;
  c = (uint64_t)(t[2] >> 51); r->v[2] = (uint64_t)t[2] & M51; t[3] += c;
  c = (uint64_t)(t[3] >> 51); r->v[3] = (uint64_t)t[3] & M51; t[4] += c;
  c = (uint64_t)(t[4] >> 51); r->v[4] = (uint64_t)t[4] & M51;
  r->v[0] += 19 * c;
  c = r->v[0] >> 51; r->v[0] &= M51; r->v[1] += c;
}
static void fe_sq(fe* r, const fe* a) { fe_mul(r, a, a); }
static void fe_sqn(fe* r, const fe* a, int n) { fe_sq(r, a); for (int i = 1; i < n; i++) fe_sq(r, r); }

static void fe_frombytes(fe* r, const uint8_t s[32]) {      /* bit 255 ignored; value NOT required to be < p */
  uint64_t w[4];
  for (int i = 0; i < 4; i++) { w[i] = 0; for (int b = 7; b >= 0; b--) w[i] = (w[i] << 8) | s[8 * i + b]; }
  r->v[0] = w[0] & M51;
  r->v[1] = ((w[0] >> 51) | (w[1] << 13)) & M51;
  r->v[2] = ((w[1] >> 38) | (w[2] << 26)) & M51;
  r->v[3] = ((w[2] >> 25) | (w[3] << 39)) & M51;
  r->v[4] = (w[3] >> 12) & M51;
}
static void fe_tobytes(uint8_t s[32], const fe* a) {        /* canonical (fully reduced) encoding */
  fe t = *a;
  fe_carry(&t); fe_carry(&t);
  /* t < 2^255 + small; subtract p when t >= p: q = (t + 19) >> 255 */
  uint64_t q = (t.v[0] + 19) >> 51;
  q = (t.v[1] + q) >> 51; q = (t.v[2] + q) >> 51; q = (t.v[3] + q) >> 51; q = (t.v[4] + q) >> 51;
  t.v[0] += 19 * q;
  uint64_t c;
  c = t.v[0] >> 51; t.v[0] &= M51; t.v[1] += c;
  c = t.v[1] >> 51; t.v[1] &= M51; t.v[2] += c;
  c = t.v[2] >> 51; t.v[2] &= M51; t.v[3] += c;
  c = t.v[3] >> 51; t.v[3] &= M51; t.v[4] += c;
  t.v[4] &= M51;
  uint64_t w[4] = {t.v[0] | (t.v[1] << 51), (t.v[1] >> 13) | (t.v[2] << 38), (t.v[2] >> 26) | (t.v[3] << 25),
                   (t.v[3] >> 39) | (t.v[4] << 12)};
  for (int i = 0; i < 4; i++) for (int b = 0; b < 8; b++) s[8 * i + b] = (uint8_t)(w[i] >> (8 * b));
}
static int fe_iszero(const fe* a) { uint8_t s[32]; fe_tobytes(s, a); uint8_t o = 0; for (int i = 0; i < 32; i++) o |= s[i]; return o == 0; }
static int fe_isneg(const fe* a) { uint8_t s[32]; fe_tobytes(s, a); return s[0] & 1; }
static int fe_eq(const fe* a, const fe* b) { fe d; fe_sub(&d, a, b); return fe_iszero(&d); }
static void fe_neg(fe* r, const fe* a) { fe z; fe_set(&z, 0); fe t = *a; fe_carry(&t); fe_sub(r, &z, &t); fe_carry(r); }

/* z^(2^252 - 3) = z^((p-5)/8) */
static void fe_pow22523(fe* r, const fe* z) {
  fe t0, t1, t2;
  fe_sq(&t0, z); fe_sqn(&t1, &t0, 2); fe_mul(&t1, z, &t1); fe_mul(&t0, &t0, &t1);      /* t0 = z^11, t1 = z^9 */
  fe_sq(&t0, &t0); fe_mul(&t0, &t1, &t0);                                               /* z^31 = 2^5 - 1 */
  fe_sqn(&t1, &t0, 5); fe_mul(&t0, &t1, &t0);                                           /* 2^10 - 1 */
  fe_sqn(&t1, &t0, 10); fe_mul(&t1, &t1, &t0);                                          /* 2^20 - 1 */
  fe_sqn(&t2, &t1, 20); fe_mul(&t1, &t2, &t1);                                          /* 2^40 - 1 */
  fe_sqn(&t1, &t1, 10); fe_mul(&t0, &t1, &t0);                                          /* 2^50 - 1 */
  fe_sqn(&t1, &t0, 50); fe_mul(&t1, &t1, &t0);                                          /* 2^100 - 1 */
  fe_sqn(&t2, &t1, 100); fe_mul(&t1, &t2, &t1);                                         /* 2^200 - 1 */
  fe_sqn(&t1, &t1, 50); fe_mul(&t0, &t1, &t0);                                          /* 2^250 - 1 */
  fe_sqn(&t0, &t0, 2); fe_mul(r, &t0, z);                                               /* 2^252 - 3 */
}
/* z^(p-2) = z^(2^255 - 21) = (z^(2^252-3))^8 * z^3 */
static void fe_invert(fe* r, const fe* z) {
  fe t, z3;
  fe_pow22523(&t, z);
  fe_sqn(&t, &t, 3);
  fe_sq(&z3, z); fe_mul(&z3, &z3, z);
  fe_mul(r, &t, &z3);
}

/* ------------------------------------------------------------------ the curve, extended coordinates */
typedef struct { fe X, Y, Z, T; } ge;

static void ge_identity(ge* p) { fe_set(&p->X, 0); fe_set(&p->Y, 1); fe_set(&p->Z, 1); fe_set(&p->T, 0); }
/* unified addition (add-2008-hwcd-3, a = -1); complete on this curve, so it also doubles and adds the identity */
static void ge_add(ge* r, const ge* p, const ge* q) {
  fe a, b, c, d, e, f, g, h, t;
  fe_sub(&a, &p->Y, &p->X); fe_carry(&a); fe_sub(&t, &q->Y, &q->X); fe_carry(&t); fe_mul(&a, &a, &t);
  fe_add(&b, &p->Y, &p->X); fe_carry(&b); fe_add(&t, &q->Y, &q->X); fe_carry(&t); fe_mul(&b, &b, &t);
  fe_mul(&c, &p->T, &q->T); fe_mul(&c, &c, &FE_2D);
  fe_mul(&d, &p->Z, &q->Z); fe_add(&d, &d, &d); fe_carry(&d);
  fe_sub(&e, &b, &a); fe_carry(&e);
  fe_sub(&f, &d, &c); fe_carry(&f);
  fe_add(&g, &d, &c); fe_carry(&g);
  fe_add(&h, &b, &a); fe_carry(&h);
  fe_mul(&r->X, &e, &f); fe_mul(&r->Y, &g, &h); fe_mul(&r->Z, &f, &g); fe_mul(&r->T, &e, &h);
}
static void ge_neg(ge* r, const ge* p) { fe_neg(&r->X, &p->X); r->Y = p->Y; r->Z = p->Z; fe_neg(&r->T, &p->T); }

/* curve25519-dalek CompressedEdwardsY::decompress.  1 = ok */
static int ge_decompress(ge* p, const uint8_t s[32]) {
  fe y, u, v, v3, r, chk, one;
  fe_set(&one, 1);
  fe_frombytes(&y, s);
  fe_sq(&u, &y); fe_mul(&v, &u, &FE_D);
  fe_sub(&u, &u, &one); fe_carry(&u);                       /* u = y^2 - 1 */
  fe_add(&v, &v, &one); fe_carry(&v);                       /* v = d y^2 + 1 */
  fe_sq(&v3, &v); fe_mul(&v3, &v3, &v);                     /* v^3 */
  fe_sq(&r, &v3); fe_mul(&r, &r, &v); fe_mul(&r, &r, &u);   /* u v^7 */
  fe_pow22523(&r, &r);
  fe_mul(&r, &r, &v3); fe_mul(&r, &r, &u);                  /* r = u v^3 (u v^7)^((p-5)/8) */
  fe_sq(&chk, &r); fe_mul(&chk, &chk, &v);                  /* v r^2 */
  if (!fe_eq(&chk, &u)) {
    fe nu; fe_neg(&nu, &u);
    if (!fe_eq(&chk, &nu)) return 0;
    fe_mul(&r, &r, &FE_SQRTM1);
  }
  if (fe_isneg(&r)) fe_neg(&r, &r);
  if (s[31] >> 7) fe_neg(&r, &r);
  p->X = r; p->Y = y; fe_set(&p->Z, 1); fe_mul(&p->T, &r, &y);
  return 1;
}
static void ge_compress(uint8_t s[32], const ge* p) {
  fe zi, x, y;
  fe_invert(&zi, &p->Z);
  fe_mul(&x, &p->X, &zi); fe_mul(&y, &p->Y, &zi);
  fe_tobytes(s, &y);
  s[31] ^= (uint8_t)(fe_isneg(&x) << 7);
}
static int ge_is_small_order(const ge* p) {
  ge q;
  ge_add(&q, p, p); ge_add(&q, &q, &q); ge_add(&q, &q, &q);
  return fe_iszero(&q.X) && fe_eq(&q.Y, &q.Z);
}

/* ------------------------------------------------------------------ scalars */
static const uint8_t ORDER_L[32] = {0xed, 0xd3, 0xf5, 0x5c, 0x1a, 0x63, 0x12, 0x58, 0xd6, 0x9c, 0xf7, 0xa2, 0xde, 0xf9, 0xde, 0x14,
                                    0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0x10};
static int sc_lt_L(const uint8_t s[32]) {       /* little-endian compare */
  for (int i = 31; i >= 0; i--) { if (s[i] < ORDER_L[i]) return 1; if (s[i] > ORDER_L[i]) return 0; }
  return 0;
}
/* h (64 bytes little-endian) mod L by binary long division: plain and obviously right, speed is irrelevant here */
static void sc_reduce512(uint8_t out[32], const uint8_t h[64]) {
  uint8_t r[33];
  memset(r, 0, sizeof r);
  for (int bit = 511; bit >= 0; bit--) {
    unsigned c = (h[bit >> 3] >> (bit & 7)) & 1;                 /* r = 2r + bit */
    for (int i = 0; i < 33; i++) { unsigned v = ((unsigned)r[i] << 1) | c; r[i] = (uint8_t)v; c = v >> 8; }
    int ge_l = r[32] != 0;
    if (!ge_l) ge_l = !sc_lt_L(r);
    if (ge_l) {
      int borrow = 0;
      for (int i = 0; i < 33; i++) {
        int v = (int)r[i] - (i < 32 ? ORDER_L[i] : 0) - borrow;
        borrow = v < 0; r[i] = (uint8_t)(v + (borrow << 8));
      }
    }
  }
  memcpy(out, r, 32);
}

static const uint8_t BASE_Y[32] = {0x58, 0x66, 0x66, 0x66, 0x66, 0x66, 0x66, 0x66, 0x66, 0x66, 0x66, 0x66, 0x66, 0x66, 0x66, 0x66,
                                   0x66, 0x66, 0x66, 0x66, 0x66, 0x66, 0x66, 0x66, 0x66, 0x66, 0x66, 0x66, 0x66, 0x66, 0x66, 0x66};

/* [s]B + [k]Q, bit-serial (Shamir) */
static void ge_double_scalarmult(ge* r, const uint8_t s[32], const ge* B, const uint8_t k[32], const ge* Q) {
  ge BQ;
  ge_add(&BQ, B, Q);
  ge_identity(r);
  for (int bit = 255; bit >= 0; bit--) {
    ge_add(r, r, r);
    int sb = (s[bit >> 3] >> (bit & 7)) & 1, kb = (k[bit >> 3] >> (bit & 7)) & 1;
    if (sb && kb) ge_add(r, r, &BQ);
    else if (sb) ge_add(r, r, B);
    else if (kb) ge_add(r, r, Q);
  }
}

/* VerifyingKey::from_bytes: does the 32-byte key decompress? */
int zko_ed25519_key_decodes(const uint8_t key[32]) { ge a; return ge_decompress(&a, key); }

/* ed25519-dalek 2.1.1 verify_strict(msg, sig).  1 = valid */
int zko_ed25519_verify_strict(const uint8_t key[32], const uint8_t* msg, size_t msg_len, const uint8_t sig[64]) {
  ge A, R, nA, Bp, Rp;
  if (!ge_decompress(&A, key)) return 0;
  if (!sc_lt_L(sig + 32)) return 0;
  if (!ge_decompress(&R, sig)) return 0;
  if (ge_is_small_order(&R) || ge_is_small_order(&A)) return 0;
  if (msg_len > 192) return 0;                                   /* oracle use: the message is a 20/32-byte hash */
  uint8_t buf[64 + 192], h[64], k[32], out[32];
  memcpy(buf, sig, 32); memcpy(buf + 32, key, 32); memcpy(buf + 64, msg, msg_len);
  zko_sha512(buf, 64 + msg_len, h);
  sc_reduce512(k, h);
  ge_neg(&nA, &A);
  ge_decompress(&Bp, BASE_Y);
  ge_double_scalarmult(&Rp, sig + 32, &Bp, k, &nA);
  ge_compress(out, &Rp);
  return memcmp(out, sig, 32) == 0;
}
