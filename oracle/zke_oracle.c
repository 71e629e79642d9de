/*
 * zke_oracle.c — CPU restatement of the zkemail_core verify path.  TEST INFRASTRUCTURE.
 * See zke_oracle.h for who may use it and for the "parity unpinned" statement.
 *
 * Every section cites the reference call site it follows (paths relative to
 * /root/reference) and, where the arithmetic lives in an un-vendored crate, the pinned
 * crate version (Cargo.lock) and the public spec it implements.
 */
#define _GNU_SOURCE
#include "zke_oracle.h"

#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#if defined(__x86_64__)
#include <cpuid.h>
#include <immintrin.h>
#endif

/* ===================================================================== SHA-256 ==
 * core/src/crypto.rs:3-7 (hash_bytes) and the body / header hashes inside cfdkim
 * (call site core/src/email.rs:31-33).  sha2 0.10.9 (Cargo.lock:2532) = FIPS 180-4.
 * sha2 picks SHA-NI at run time when the CPU has it; so does this restatement. */
static const uint32_t K256[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5,
    0xd807aa98, 0x12835b01, 0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174,
    0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da,
    0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967,
    0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
    0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070,
    0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3,
    0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};

#define ROR(x, n) (((x) >> (n)) | ((x) << (32 - (n))))

static void sha256_blocks_c(uint32_t st[8], const uint8_t* p, size_t nblk) {
  uint32_t w[64];
  while (nblk--) {
    for (int i = 0; i < 16; i++)
      w[i] = ((uint32_t)p[4 * i] << 24) | ((uint32_t)p[4 * i + 1] << 16) | ((uint32_t)p[4 * i + 2] << 8) | p[4 * i + 3];
    for (int i = 16; i < 64; i++) {
      uint32_t s0 = ROR(w[i - 15], 7) ^ ROR(w[i - 15], 18) ^ (w[i - 15] >> 3);
      uint32_t s1 = ROR(w[i - 2], 17) ^ ROR(w[i - 2], 19) ^ (w[i - 2] >> 10);
      w[i] = w[i - 16] + s0 + w[i - 7] + s1;
    }
    uint32_t a = st[0], b = st[1], c = st[2], d = st[3], e = st[4], f = st[5], g = st[6], h = st[7];
    for (int i = 0; i < 64; i++) {
      uint32_t S1 = ROR(e, 6) ^ ROR(e, 11) ^ ROR(e, 25);
      uint32_t ch = (e & f) ^ (~e & g);
      uint32_t t1 = h + S1 + ch + K256[i] + w[i];
      uint32_t S0 = ROR(a, 2) ^ ROR(a, 13) ^ ROR(a, 22);
      uint32_t mj = (a & b) ^ (a & c) ^ (b & c);
      uint32_t t2 = S0 + mj;
      h = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
    }
    st[0] += a; st[1] += b; st[2] += c; st[3] += d; st[4] += e; st[5] += f; st[6] += g; st[7] += h;
    p += 64;
  }
}

#if defined(__x86_64__)
__attribute__((target("sha,sse4.1,ssse3"))) static void sha256_blocks_ni(uint32_t st[8], const uint8_t* p, size_t nblk) {
  const __m128i MASK = _mm_set_epi64x(0x0c0d0e0f08090a0bULL, 0x0405060700010203ULL);
  __m128i tmp = _mm_loadu_si128((const __m128i*)&st[0]);
  __m128i s1 = _mm_loadu_si128((const __m128i*)&st[4]);
  tmp = _mm_shuffle_epi32(tmp, 0xB1);
  s1 = _mm_shuffle_epi32(s1, 0x1B);
  __m128i s0 = _mm_alignr_epi8(tmp, s1, 8);
  s1 = _mm_blend_epi16(s1, tmp, 0xF0);
  while (nblk--) {
    __m128i save0 = s0, save1 = s1, v[4];
    for (int j = 0; j < 16; j++) {
      if (j < 4) {
        v[j] = _mm_shuffle_epi8(_mm_loadu_si128((const __m128i*)(p + 16 * j)), MASK);
      } else {
        __m128i t = _mm_sha256msg1_epu32(v[(j - 4) & 3], v[(j - 3) & 3]);
        t = _mm_add_epi32(t, _mm_alignr_epi8(v[(j - 1) & 3], v[(j - 2) & 3], 4));
        v[j & 3] = _mm_sha256msg2_epu32(t, v[(j - 1) & 3]);
      }
      __m128i msg = _mm_add_epi32(v[j & 3], _mm_loadu_si128((const __m128i*)&K256[4 * j]));
      s1 = _mm_sha256rnds2_epu32(s1, s0, msg);
      msg = _mm_shuffle_epi32(msg, 0x0E);
      s0 = _mm_sha256rnds2_epu32(s0, s1, msg);
    }
    s0 = _mm_add_epi32(s0, save0);
    s1 = _mm_add_epi32(s1, save1);
    p += 64;
  }
  tmp = _mm_shuffle_epi32(s0, 0x1B);
  s1 = _mm_shuffle_epi32(s1, 0xB1);
  s0 = _mm_blend_epi16(tmp, s1, 0xF0);
  s1 = _mm_alignr_epi8(s1, tmp, 8);
  _mm_storeu_si128((__m128i*)&st[0], s0);
  _mm_storeu_si128((__m128i*)&st[4], s1);
}
#endif

static int g_shani = -1;
int zko_sha256_uses_shani(void) {
  if (g_shani < 0) {
    g_shani = 0;
#if defined(__x86_64__)
    unsigned a, b, c, d;
    if (__get_cpuid_count(7, 0, &a, &b, &c, &d) && (b & (1u << 29))) {
      unsigned a1, b1, c1, d1;
      if (__get_cpuid(1, &a1, &b1, &c1, &d1) && (c1 & (1u << 19)) && (c1 & (1u << 9))) g_shani = 1;
    }
    if (getenv("ZKO_NO_SHANI")) g_shani = 0;
#endif
  }
  return g_shani;
}

static void sha256_blocks(uint32_t st[8], const uint8_t* p, size_t nblk) {
#if defined(__x86_64__)
  if (zko_sha256_uses_shani()) { sha256_blocks_ni(st, p, nblk); return; }
#endif
  sha256_blocks_c(st, p, nblk);
}

void zko_sha256(const uint8_t* data, size_t len, uint8_t out[32]) {
  uint32_t st[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
  size_t nfull = len / 64;
  if (nfull) sha256_blocks(st, data, nfull);
  uint8_t tail[128];
  size_t rem = len - nfull * 64;
  memset(tail, 0, sizeof tail);
  if (rem) memcpy(tail, data + nfull * 64, rem);
  tail[rem] = 0x80;
  size_t tl = (rem + 9 <= 64) ? 64 : 128;
  uint64_t bits = (uint64_t)len * 8;
  for (int i = 0; i < 8; i++) tail[tl - 1 - i] = (uint8_t)(bits >> (8 * i));
  sha256_blocks(st, tail, tl / 64);
  for (int i = 0; i < 8; i++) {
    out[4 * i] = (uint8_t)(st[i] >> 24); out[4 * i + 1] = (uint8_t)(st[i] >> 16);
    out[4 * i + 2] = (uint8_t)(st[i] >> 8); out[4 * i + 3] = (uint8_t)st[i];
  }
}

/* ========================================================================= SHA-1 ==
 * a=rsa-sha1 signatures (cfdkim HashAlgo::RsaSha1 over sha-1 0.10.1, Cargo.lock:2521).  FIPS 180-4 §6.1. */
#define ROL(x, n) (((x) << (n)) | ((x) >> (32 - (n))))
void zko_sha1(const uint8_t* data, size_t len, uint8_t out[20]) {
  uint32_t st[5] = {0x67452301, 0xEFCDAB89, 0x98BADCFE, 0x10325476, 0xC3D2E1F0};
  const size_t total = ((len + 9 + 63) / 64) * 64;
  for (size_t off = 0; off < total; off += 64) {
    uint8_t blk[64];
    for (size_t i = 0; i < 64; i++) {
      const size_t p = off + i;
      uint8_t b = 0;
      if (p < len) b = data[p];
      else if (p == len) b = 0x80;
      else if (p >= total - 8) b = (uint8_t)(((uint64_t)len * 8) >> (8 * (total - 1 - p)));
      blk[i] = b;
    }
    uint32_t w[80];
    for (int i = 0; i < 16; i++) w[i] = ((uint32_t)blk[4 * i] << 24) | ((uint32_t)blk[4 * i + 1] << 16) | ((uint32_t)blk[4 * i + 2] << 8) | blk[4 * i + 3];
    for (int i = 16; i < 80; i++) w[i] = ROL(w[i - 3] ^ w[i - 8] ^ w[i - 14] ^ w[i - 16], 1);
    uint32_t a = st[0], b = st[1], c = st[2], d = st[3], e = st[4];
    for (int i = 0; i < 80; i++) {
      uint32_t f, k;
      if (i < 20) { f = (b & c) | (~b & d); k = 0x5A827999; }
      else if (i < 40) { f = b ^ c ^ d; k = 0x6ED9EBA1; }
      else if (i < 60) { f = (b & c) | (b & d) | (c & d); k = 0x8F1BBCDC; }
      else { f = b ^ c ^ d; k = 0xCA62C1D6; }
      const uint32_t t = ROL(a, 5) + f + e + k + w[i];
      e = d; d = c; c = ROL(b, 30); b = a; a = t;
    }
    st[0] += a; st[1] += b; st[2] += c; st[3] += d; st[4] += e;
  }
  for (int i = 0; i < 5; i++) { out[4 * i] = (uint8_t)(st[i] >> 24); out[4 * i + 1] = (uint8_t)(st[i] >> 16); out[4 * i + 2] = (uint8_t)(st[i] >> 8); out[4 * i + 3] = (uint8_t)st[i]; }
}

/* ====================================================================== base64 ==
 * cfdkim compares base64(body hash) with bh= as strings and decodes b= with
 * base64::engine::general_purpose::STANDARD (padded, canonical, no trailing bits). */
static const char B64[] = "ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789+/";
size_t zko_b64_encode(const uint8_t* in, size_t n, char* out) {
  size_t o = 0, i = 0;
  for (; i + 3 <= n; i += 3) {
    uint32_t v = ((uint32_t)in[i] << 16) | ((uint32_t)in[i + 1] << 8) | in[i + 2];
    out[o++] = B64[v >> 18]; out[o++] = B64[(v >> 12) & 63]; out[o++] = B64[(v >> 6) & 63]; out[o++] = B64[v & 63];
  }
  if (n - i == 1) {
    uint32_t v = (uint32_t)in[i] << 16;
    out[o++] = B64[v >> 18]; out[o++] = B64[(v >> 12) & 63]; out[o++] = '='; out[o++] = '=';
  } else if (n - i == 2) {
    uint32_t v = ((uint32_t)in[i] << 16) | ((uint32_t)in[i + 1] << 8);
    out[o++] = B64[v >> 18]; out[o++] = B64[(v >> 12) & 63]; out[o++] = B64[(v >> 6) & 63]; out[o++] = '=';
  }
  return o;
}
static int b64val(uint8_t c) {
  if (c >= 'A' && c <= 'Z') return c - 'A';
  if (c >= 'a' && c <= 'z') return c - 'a' + 26;
  if (c >= '0' && c <= '9') return c - '0' + 52;
  if (c == '+') return 62;
  if (c == '/') return 63;
  return -1;
}
long zko_b64_decode(const uint8_t* in, size_t n, uint8_t* out) {
  if (n % 4 != 0) return -1;
  size_t o = 0;
  for (size_t i = 0; i < n; i += 4) {
    int last = (i + 4 == n);
    int a = b64val(in[i]), b = b64val(in[i + 1]);
    if (a < 0 || b < 0) return -1;
    uint8_t c3 = in[i + 2], c4 = in[i + 3];
    if (last && c4 == '=') {
      if (c3 == '=') {
        if (b & 15) return -1; /* non-zero trailing bits */
        out[o++] = (uint8_t)((a << 2) | (b >> 4));
      } else {
        int c = b64val(c3);
        if (c < 0 || (c & 3)) return -1;
        out[o++] = (uint8_t)((a << 2) | (b >> 4));
        out[o++] = (uint8_t)((b << 4) | (c >> 2));
      }
    } else {
      int c = b64val(c3), d = b64val(c4);
      if (c < 0 || d < 0) return -1;
      out[o++] = (uint8_t)((a << 2) | (b >> 4));
      out[o++] = (uint8_t)((b << 4) | (c >> 2));
      out[o++] = (uint8_t)((c << 6) | d);
    }
  }
  return (long)o;
}

/* ========================================================================= RSA ==
 * rsa 0.9.6 (Cargo.lock:2231) over num-bigint-dig 0.8.4 (Cargo.lock:1683), reached
 * through cfdkim's verify_signature; key built at core/src/email.rs:28-29.
 * RFC 8017 §8.2.2 (RSASSA-PKCS1-V1_5-VERIFY), §9.2 (EMSA-PKCS1-v1_5), A.1.1 (RSAPublicKey).
 * Montgomery CIOS with 64-bit limbs (num-bigint-dig's modpow is Montgomery for odd n). */
typedef unsigned __int128 u128;
#define MAXL 66

static uint64_t mont_n0inv(uint64_t n0) {
  uint64_t x = n0; /* Newton: x = n0^-1 mod 2^64 */
  for (int i = 0; i < 6; i++) x *= 2 - n0 * x;
  return (uint64_t)0 - x;
}
static int big_cmp(const uint64_t* a, const uint64_t* b, int L) {
  for (int i = L - 1; i >= 0; i--) {
    if (a[i] != b[i]) return a[i] < b[i] ? -1 : 1;
  }
  return 0;
}
static uint64_t big_sub(uint64_t* r, const uint64_t* a, const uint64_t* b, int L) {
  uint64_t br = 0;
  for (int i = 0; i < L; i++) {
    u128 d = (u128)a[i] - b[i] - br;
    r[i] = (uint64_t)d;
    br = (uint64_t)(d >> 64) & 1;
  }
  return br;
}
static void mont_mul(uint64_t* r, const uint64_t* a, const uint64_t* b, const uint64_t* n, uint64_t ni, int L) {
  uint64_t t[MAXL + 2];
  memset(t, 0, sizeof(uint64_t) * (L + 2));
  for (int i = 0; i < L; i++) {
    u128 c = 0;
    for (int j = 0; j < L; j++) {
      c += (u128)a[j] * b[i] + t[j];
      t[j] = (uint64_t)c;
      c >>= 64;
    }
    c += t[L];
    t[L] = (uint64_t)c;
    t[L + 1] = (uint64_t)(c >> 64);
    uint64_t m = t[0] * ni;
    c = (u128)m * n[0] + t[0];
    c >>= 64;
    for (int j = 1; j < L; j++) {
      c += (u128)m * n[j] + t[j];
      t[j - 1] = (uint64_t)c;
      c >>= 64;
    }
    c += t[L];
    t[L - 1] = (uint64_t)c;
    t[L] = t[L + 1] + (uint64_t)(c >> 64);
  }
  if (t[L] || big_cmp(t, n, L) >= 0) big_sub(t, t, n, L);
  memcpy(r, t, sizeof(uint64_t) * L);
}
/* x = 2x mod n, x < n */
static void mod_double(uint64_t* x, const uint64_t* n, int L) {
  uint64_t top = x[L - 1] >> 63;
  for (int i = L - 1; i > 0; i--) x[i] = (x[i] << 1) | (x[i - 1] >> 63);
  x[0] <<= 1;
  if (top || big_cmp(x, n, L) >= 0) big_sub(x, x, n, L);
}

int zko_rsa_modexp(const uint8_t* sig, const uint8_t* mod, uint32_t bytes, uint64_t e, uint8_t* em) {
  int L = (int)((bytes + 7) / 8);
  if (L < 1 || L > MAXL - 2) return -1;
  uint64_t n[MAXL] = {0}, s[MAXL] = {0};
  for (uint32_t i = 0; i < bytes; i++) {
    n[i / 8] |= (uint64_t)mod[bytes - 1 - i] << (8 * (i % 8));
    s[i / 8] |= (uint64_t)sig[bytes - 1 - i] << (8 * (i % 8));
  }
  if (!(n[0] & 1)) return -1;
  if (big_cmp(s, n, L) >= 0) return -1;
  uint64_t ni = mont_n0inv(n[0]);
  /* one = R mod n by 64L modular doublings of 1 */
  uint64_t one[MAXL] = {0}, two[MAXL], rr[MAXL];
  one[0] = 1;
  if (big_cmp(one, n, L) >= 0) return -1; /* n == 1 */
  for (int i = 0; i < 64 * L; i++) mod_double(one, n, L);
  /* rr = R^2 mod n: Montgomery form of 2 raised to 64L (MM(2^a R, 2^b R) = 2^(a+b) R) */
  memcpy(two, one, sizeof(uint64_t) * L);
  mod_double(two, n, L);
  memcpy(rr, one, sizeof(uint64_t) * L);
  unsigned ex = (unsigned)(64 * L);
  for (int bit = 31; bit >= 0; bit--) {
    mont_mul(rr, rr, rr, n, ni, L);
    if ((ex >> bit) & 1) mont_mul(rr, rr, two, n, ni, L);
  }
  uint64_t x[MAXL], acc[MAXL], lit[MAXL] = {0};
  mont_mul(x, s, rr, n, ni, L);
  memcpy(acc, one, sizeof(uint64_t) * L);
  for (int bit = 63; bit >= 0; bit--) {
    mont_mul(acc, acc, acc, n, ni, L);
    if ((e >> bit) & 1) mont_mul(acc, acc, x, n, ni, L);
  }
  lit[0] = 1;
  mont_mul(acc, acc, lit, n, ni, L);
  for (uint32_t i = 0; i < bytes; i++) em[bytes - 1 - i] = (uint8_t)(acc[i / 8] >> (8 * (i % 8)));
  return 0;
}

/* DER length; returns bytes consumed or 0 on error (der crate: definite, minimal) */
static size_t der_len(const uint8_t* p, size_t avail, size_t* out) {
  if (avail < 1) return 0;
  if (p[0] < 0x80) { *out = p[0]; return 1; }
  int nb = p[0] & 0x7f;
  if (nb == 0 || nb > 4 || (size_t)nb + 1 > avail) return 0;
  size_t v = 0;
  for (int i = 0; i < nb; i++) v = (v << 8) | p[1 + i];
  if (p[1] == 0) return 0;             /* leading zero octet: non-minimal */
  if (nb == 1 && v < 0x80) return 0;   /* should have used the short form */
  *out = v;
  return (size_t)nb + 1;
}
static int der_uint(const uint8_t* p, size_t avail, const uint8_t** val, size_t* vlen, size_t* used) {
  if (avail < 2 || p[0] != 0x02) return -1;
  size_t l, c = der_len(p + 1, avail - 1, &l);
  if (!c || l == 0 || 1 + c + l > avail) return -1;
  const uint8_t* v = p + 1 + c;
  if (v[0] & 0x80) return -1;                         /* negative */
  if (l > 1 && v[0] == 0 && !(v[1] & 0x80)) return -1; /* non-minimal */
  if (l > 1 && v[0] == 0) { v++; l--; }
  *val = v; *vlen = l; *used = 1 + c + (size_t)(v - (p + 1 + c)) + l;
  return 0;
}
int zko_parse_rsa_pkcs1(const uint8_t* der, size_t len, uint8_t* mod_out, uint32_t* mod_len, uint64_t* e_out) {
  if (len < 2 || der[0] != 0x30) return ZKE_D_KEY_DER;
  size_t sl, c = der_len(der + 1, len - 1, &sl);
  if (!c || 1 + c + sl != len) return ZKE_D_KEY_DER;
  const uint8_t* p = der + 1 + c;
  size_t avail = sl, used;
  const uint8_t *nv, *ev;
  size_t nl, el;
  if (der_uint(p, avail, &nv, &nl, &used)) return ZKE_D_KEY_DER;
  p += used; avail -= used;
  if (der_uint(p, avail, &ev, &el, &used)) return ZKE_D_KEY_DER;
  if (used != avail) return ZKE_D_KEY_DER;
  /* rsa 0.9.6 RsaPublicKey::new -> check_public: n.bits() <= 4096, 2 <= e <= 2^33-1 */
  size_t nbits = 0;
  if (!(nl == 1 && nv[0] == 0)) {
    nbits = nl * 8;
    for (uint8_t t = nv[0]; !(t & 0x80); t <<= 1) nbits--;
  }
  if (nbits > 4096) return ZKE_D_KEY_RANGE;
  if (el > 8) return ZKE_D_KEY_RANGE;
  uint64_t e = 0;
  for (size_t i = 0; i < el; i++) e = (e << 8) | ev[i];
  if (e < 2 || e > ((1ull << 33) - 1)) return ZKE_D_KEY_RANGE;
  memcpy(mod_out, nv, nl);
  *mod_len = (uint32_t)nl;
  *e_out = e;
  return 0;
}

static const uint8_t SHA256_PREFIX[19] = {0x30, 0x31, 0x30, 0x0d, 0x06, 0x09, 0x60, 0x86, 0x48, 0x01,
                                          0x65, 0x03, 0x04, 0x02, 0x01, 0x05, 0x00, 0x04, 0x20};

static const uint8_t SHA1_PREFIX[15] = {0x30, 0x21, 0x30, 0x09, 0x06, 0x05, 0x2b, 0x0e, 0x03, 0x02, 0x1a, 0x05, 0x00, 0x04, 0x14};

/* rsa 0.9.6 pkcs1v15::verify + pkcs1v15_sign_unpad, DigestInfo prefix of SHA-256 or SHA-1 */
static int rsa_pkcs1v15_verify(const uint8_t* mod, uint32_t k, uint64_t e, const uint8_t* sig, uint32_t sig_len,
                               const uint8_t* hash, uint32_t hlen, const uint8_t* prefix, uint32_t plen, uint8_t* em_out) {
  uint8_t em[ZKE_MAX_RSA_BYTES + 8];
  if (em_out) memset(em_out, 0, k);
  if (k == 0 || k > ZKE_MAX_RSA_BYTES) return 0;
  if (sig_len != k) return 0;
  if (zko_rsa_modexp(sig, mod, k, e, em)) return 0; /* sig >= n */
  if (em_out) memcpy(em_out, em, k);
  const uint32_t tlen = plen + hlen;
  if (k < tlen + 11) return 0;
  int ok = em[0] == 0 && em[1] == 1;
  ok &= memcmp(em + k - hlen, hash, hlen) == 0;
  ok &= memcmp(em + k - tlen, prefix, plen) == 0;
  ok &= em[k - tlen - 1] == 0;
  for (uint32_t i = 2; i < k - tlen - 1; i++) ok &= em[i] == 0xff;
  return ok;
}
int zko_rsa_pkcs1v15_sha256_verify(const uint8_t* mod, uint32_t k, uint64_t e, const uint8_t* sig,
                                   uint32_t sig_len, const uint8_t hash[32], uint8_t* em_out) {
  return rsa_pkcs1v15_verify(mod, k, e, sig, sig_len, hash, 32, SHA256_PREFIX, 19, em_out);
}

/* =================================================================== mailparse ==
 * mailparse 0.15.0 (Cargo.lock:1597) parse_mail, call site core/src/email.rs:26: parse_headers / parse_header for the
 * header list, then parse_mail_recursive's walk over the MIME subparts, whose only effect on this path is an Err (a panic
 * at email.rs:26) when a subpart's header block is malformed.  Restated from recollection of the crate (its source is
 * not in the container). */

/* parse_header at raw[ix] (ix < len): spans of one header and the index after it; 0 or -ZKE_D_HDR_* */
static int parse_one_header(const uint8_t* raw, size_t len, size_t ix, size_t* key_end, size_t* vs_out, size_t* ve_out, size_t* next) {
  if (raw[ix] == ' ') return -(int)ZKE_D_HDR_LEADING_SPACE;
  size_t p = ix, vs, ve;
  while (p < len && raw[p] != ':' && raw[p] != '\n') p++;
  if (p >= len) {             /* ran off the end inside the key: key = rest, empty value */
    *key_end = len; vs = ve = len; p = len;
  } else if (raw[p] == '\n') { /* key line without colon */
    *key_end = p; vs = ve = p; p = p + 1;
  } else {
    *key_end = p;
    p++;
    while (p < len && raw[p] == ' ') p++;
    vs = ve = p;
    /* Value / ValueNewline states */
    for (;;) {
      if (p >= len) break;
      uint8_t c = raw[p];
      if (c == '\n') {
        if (p + 1 < len && (raw[p + 1] == ' ' || raw[p + 1] == '\t')) { p++; continue; }
        p++;
        break;
      }
      if (c != '\r') ve = p + 1;
      p++;
    }
    if (vs > len) vs = ve = len;
  }
  *vs_out = vs; *ve_out = ve; *next = p;
  return 0;
}
/* parse_headers: 0 = a header starts at ix, 1 = the list ends (ix moved past the empty line), <0 = error */
static int header_list_step(const uint8_t* raw, size_t len, size_t* ix) {
  if (*ix >= len) return 1;
  if (raw[*ix] == '\n') { *ix += 1; return 1; }
  if (raw[*ix] == '\r') {
    if (*ix + 1 < len && raw[*ix + 1] == '\n') { *ix += 2; return 1; }
    return -(int)ZKE_D_HDR_LONE_CR;
  }
  return 0;
}
long zko_parse_headers(const uint8_t* raw, size_t len, uint32_t* spans, size_t max_headers, size_t* body_ix) {
  size_t ix = 0, nh = 0;
  for (;;) {
    int st = header_list_step(raw, len, &ix);
    if (st < 0) return st;
    if (st) break;
    size_t key_end, vs, ve, next;
    int r = parse_one_header(raw, len, ix, &key_end, &vs, &ve, &next);
    if (r) return r;
    if (nh >= max_headers) return -(long)ZKE_D_U_TOO_MANY_HEADERS;
    spans[4 * nh] = (uint32_t)ix; spans[4 * nh + 1] = (uint32_t)key_end;
    spans[4 * nh + 2] = (uint32_t)vs; spans[4 * nh + 3] = (uint32_t)ve;
    nh++;
    ix = next;
  }
  if (body_ix) *body_ix = ix;
  return (long)nh;
}

/* ---- parse_mail_recursive: the subpart walk.
 *   (headers, ix_body) = parse_headers(part)?                       an Err here is the panic of email.rs:26
 *   ctype = parse_content_type(headers.get_first_value("Content-Type"))
 *   if ctype.mimetype.starts_with("multipart/") && ctype.params has "boundary" && part.len() > ix_body:
 *       boundary = "--" + params["boundary"]
 *       the body ends at the first line that starts with the boundary; after each such line the next part runs from the
 *       byte after the next LF to the next line that starts with the boundary (no such line: the rest is not a part);
 *       each part is parsed recursively; "--" right after a boundary ends the walk.
 * get_first_value: the first header whose key is "Content-Type" (eq_ignore_ascii_case); get_value() unfolds the raw
 * value (lines(), each trim_start()ed, joined with one SP) and decodes RFC 2047 words; parse_param_content splits at
 * every ';' (quotes do not protect one), trims, lower-cases the first token (the mimetype) and the parameter names,
 * strips one pair of double quotes from a value, last duplicate wins; RFC 2231 forms (boundary*, boundary*0 ...) only
 * matter when no plain "boundary" exists.
 * The engine decides on the raw bytes, which is exact for ASCII values without encoded words and with the boundary on
 * one line; what is not — bytes >= 0x80 or "=?" in a value that decides (str::trim() and to_lowercase() are Unicode-aware,
 * decoded words can hold anything), a folded boundary value, RFC 2231 boundary forms, nesting beyond 8 multiparts — is
 * ZKE_UNSUPPORTED, never a guess. */
#define MIME_MAX_DEPTH 8
static int rust_ws(uint8_t c) { return c == ' ' || (c >= 9 && c <= 13); }      /* ASCII members of White_Space */
static void trim_span(const uint8_t* v, size_t* s, size_t* e) {
  while (*s < *e && rust_ws(v[*s])) (*s)++;
  while (*e > *s && rust_ws(v[*e - 1])) (*e)--;
}
static int ieq_lit(const uint8_t* v, size_t n, const char* lit) {
  for (size_t i = 0; i < n; i++) {
    uint8_t c = v[i];
    if (c >= 'A' && c <= 'Z') c = (uint8_t)(c + 32);
    if (c != (uint8_t)lit[i]) return 0;
  }
  return 1;
}
static int undecidable(const uint8_t* v, size_t s, size_t e, size_t n) {    /* v[s,e) holds a byte >= 0x80 or "=?" */
  for (size_t i = s; i < e; i++)
    if (v[i] >= 0x80 || (v[i] == '=' && i + 1 < n && v[i + 1] == '?')) return 1;
  return 0;
}
/* 0: not a multipart with a boundary; 1: boundary value = v[*bs, *be); < 0: -ZKE_D_U_MIME_* */
static int mime_content_type(const uint8_t* v, size_t n, size_t* bs, size_t* be) {
  size_t t0e = 0;
  while (t0e < n && v[t0e] != ';') t0e++;
  if (undecidable(v, 0, t0e, n)) return -(int)ZKE_D_U_MIME_CTYPE;
  size_t s = 0, e = t0e;
  trim_span(v, &s, &e);
  if (e - s < 10 || !ieq_lit(v + s, 10, "multipart/")) return 0;
  if (undecidable(v, 0, n, n)) return -(int)ZKE_D_U_MIME_CTYPE;
  int have = 0, starred = 0;
  for (size_t p = t0e + 1; p <= n;) {
    size_t q = p;
    while (q < n && v[q] != ';') q++;
    size_t eq = p;
    while (eq < q && v[eq] != '=') eq++;
    if (eq < q) {
      size_t ks = p, ke = eq, vs = eq + 1, ve = q;
      trim_span(v, &ks, &ke);
      trim_span(v, &vs, &ve);
      if (ve - vs > 1 && v[vs] == '"' && v[ve - 1] == '"') { vs++; ve--; }
      if (ke - ks == 8 && ieq_lit(v + ks, 8, "boundary")) { have = 1; *bs = vs; *be = ve; }
      else if (ke - ks >= 9 && ieq_lit(v + ks, 9, "boundary*")) starred = 1;
    }
    if (q >= n) break;
    p = q + 1;
  }
  if (have) {
    for (size_t i = *bs; i < *be; i++) if (v[i] == '\n') return -(int)ZKE_D_U_MIME_BOUNDARY;
    return 1;
  }
  return starred ? -(int)ZKE_D_U_MIME_BOUNDARY : 0;
}
/* find_from_u8_line_prefix within raw[a, b): first pos >= from where "--" + raw[bs, be) starts a line */
static size_t find_boundary_line(const uint8_t* raw, size_t a, size_t b, size_t from, size_t bs, size_t be) {
  const size_t L = 2 + (be - bs);
  for (size_t pos = from; pos + L <= b; pos++)
    if ((pos == a || raw[pos - 1] == '\n') && raw[pos] == '-' && raw[pos + 1] == '-' && memcmp(raw + pos + 2, raw + bs, L - 2) == 0) return pos;
  return (size_t)-1;
}
/* parse_mail_recursive over raw[a, b).  0, or ZKE_PARSE_FAIL / ZKE_UNSUPPORTED with *detail */
static uint32_t mime_walk(const uint8_t* raw, size_t a, size_t b, int depth, uint32_t* detail) {
  size_t ix = a, ct_s = 0, ct_e = 0;
  int has_ct = 0;
  for (;;) {
    int st = header_list_step(raw, b, &ix);
    if (st < 0) { *detail = depth ? ZKE_D_SUBPART_LONE_CR : (uint32_t)-st; return ZKE_PARSE_FAIL; }
    if (st) break;
    size_t key_end, vs, ve, next;
    int r = parse_one_header(raw, b, ix, &key_end, &vs, &ve, &next);
    if (r) { *detail = depth ? ZKE_D_SUBPART_LEADING_SPACE : (uint32_t)-r; return ZKE_PARSE_FAIL; }
    if (!has_ct && key_end - ix == 12 && ieq_lit(raw + ix, 12, "content-type")) { has_ct = 1; ct_s = vs; ct_e = ve; }
    ix = next;
  }
  if (!has_ct) return 0;
  size_t bs = 0, be = 0;
  int m = mime_content_type(raw + ct_s, ct_e - ct_s, &bs, &be);
  if (m < 0) { *detail = (uint32_t)-m; return ZKE_UNSUPPORTED; }
  if (m == 0 || !(b > ix)) return 0;
  if (depth >= MIME_MAX_DEPTH) { *detail = ZKE_D_U_MIME_DEPTH; return ZKE_UNSUPPORTED; }
  bs += ct_s; be += ct_s;
  const size_t L = 2 + (be - bs);
  size_t pos = find_boundary_line(raw, a, b, ix, bs, be);
  if (pos == (size_t)-1) return 0;
  size_t bend = pos + L;
  for (;;) {
    size_t nl = bend;
    while (nl < b && raw[nl] != '\n') nl++;
    if (nl >= b) break;
    const size_t ps = nl + 1;
    const size_t pe = find_boundary_line(raw, a, b, ps, bs, be);
    if (pe == (size_t)-1) break;
    uint32_t r = mime_walk(raw, ps, pe, depth + 1, detail);
    if (r) return r;
    bend = pe + L;
    if (bend + 1 < b && raw[bend] == '-' && raw[bend + 1] == '-') break;
  }
  return 0;
}
uint32_t zko_mime_walk(const uint8_t* raw, size_t len, uint32_t* detail) {
  uint32_t d = 0;
  uint32_t r = mime_walk(raw, 0, len, 0, &d);
  if (detail) *detail = d;
  return r;
}

/* ====================================================================== cfdkim ==
 * cfdkim 0.3.3 @ zkemail/cfdkim#75af99fb (Cargo.lock:475-477).  Call sites:
 * core/src/email.rs:31-33 (verify_email_with_key), core/src/circuits.rs:34-35
 * (canonicalize_signed_email).  RFC 6376 §3.4 (canonicalisation), §3.5 (tags),
 * §3.7 (hash computation), §6.1 (verifier actions). */
static int is_fws(uint8_t c) { return c == ' ' || c == '\t' || c == '\r' || c == '\n'; }
static int is_valchar(uint8_t c) { return (c >= 0x21 && c <= 0x3a) || (c >= 0x3c && c <= 0x7e); }
static int is_alpha(uint8_t c) { return (c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z'); }
static int is_alnumpunc(uint8_t c) { return is_alpha(c) || (c >= '0' && c <= '9') || c == '_'; }
static uint8_t lower(uint8_t c) { return (c >= 'A' && c <= 'Z') ? (uint8_t)(c + 32) : c; }

typedef struct {
  uint32_t name_s, name_e; /* in the header value */
  uint32_t raw_s, raw_e;   /* raw tag value (inner FWS kept) */
  uint32_t val_off, val_len; /* FWS-stripped value in tagbuf */
} tag_t;

typedef struct {
  tag_t tags[ZKE_MAX_TAGS];
  int ntags;
  uint8_t* tagbuf; /* stripped values, capacity = value length */
  size_t tagbuf_len;
} taglist_t;

/* tag-spec = [FWS] tag-name [FWS] "=" [FWS] tag-value [FWS]; returns new pos or -1 */
static long parse_tag_spec(const uint8_t* s, size_t n, size_t pos, taglist_t* tl, int* overflow) {
  size_t p = pos;
  while (p < n && is_fws(s[p])) p++;
  if (p >= n || !is_alpha(s[p])) return -1;
  size_t ns = p;
  while (p < n && is_alnumpunc(s[p])) p++;
  size_t ne = p;
  while (p < n && is_fws(s[p])) p++;
  if (p >= n || s[p] != '=') return -1;
  p++;
  while (p < n && is_fws(s[p])) p++;
  size_t rs = p, re = p;
  while (p < n && (is_valchar(s[p]) || is_fws(s[p]))) {
    if (is_valchar(s[p])) re = p + 1;
    p++;
  }
  if (tl->ntags >= (int)ZKE_MAX_TAGS) { *overflow = 1; return (long)p; }
  tag_t* t = &tl->tags[tl->ntags++];
  t->name_s = (uint32_t)ns; t->name_e = (uint32_t)ne;
  t->raw_s = (uint32_t)rs; t->raw_e = (uint32_t)re;
  t->val_off = (uint32_t)tl->tagbuf_len;
  for (size_t i = rs; i < re; i++)
    if (!is_fws(s[i])) tl->tagbuf[tl->tagbuf_len++] = s[i];
  t->val_len = (uint32_t)(tl->tagbuf_len - t->val_off);
  if (tl->tagbuf_len > ZKE_MAX_TAGBUF) { *overflow = 2; return (long)p; }   /* engine limit, mirrored so parity is defined */
  return (long)p;
}
/* tag-list = tag-spec *( ";" tag-spec ) [ ";" ]; trailing garbage is ignored (nom remainder dropped) */
static int parse_tag_list(const uint8_t* s, size_t n, taglist_t* tl, int* overflow) {
  tl->ntags = 0; tl->tagbuf_len = 0;
  long p = parse_tag_spec(s, n, 0, tl, overflow);
  if (p < 0) return -1;
  while (!*overflow && (size_t)p < n && s[p] == ';') {
    long q = parse_tag_spec(s, n, (size_t)p + 1, tl, overflow);
    if (q < 0) break;
    p = q;
  }
  return 0;
}
/* IndexMap insert: the last tag of a name wins */
static const tag_t* get_tag(const taglist_t* tl, const uint8_t* s, const char* name) {
  size_t nl = strlen(name);
  const tag_t* r = NULL;
  for (int i = 0; i < tl->ntags; i++) {
    const tag_t* t = &tl->tags[i];
    if (t->name_e - t->name_s == nl && memcmp(s + t->name_s, name, nl) == 0) r = t;
  }
  return r;
}
static int tag_eq(const taglist_t* tl, const tag_t* t, const char* lit) {
  size_t l = strlen(lit);
  return t->val_len == l && memcmp(tl->tagbuf + t->val_off, lit, l) == 0;
}

/* cfdkim validate_header (RFC 6376 §6.1.1).  `strict`: ZKE_STRICT_* — the readings of cfdkim that could not be verified offline
 * (zke_options' strictness flags); each is one named site here and the same site in csrc/parse.hip.h. */
static int validate_header(const uint8_t* s, size_t n, taglist_t* tl, uint32_t strict, uint64_t now) {
  int overflow = 0;
  if (parse_tag_list(s, n, tl, &overflow)) return ZKE_D_SIG_SYNTAX;
  if (overflow == 1) return ZKE_D_U_TOO_MANY_TAGS;
  if (overflow == 2) return ZKE_D_U_SIG_TOO_LONG;
  static const char* req[] = {"v", "a", "b", "bh", "d", "h", "s"};
  for (int i = 0; i < 7; i++)
    if (!get_tag(tl, s, req[i])) return ZKE_D_MISSING_TAG;
  if (!tag_eq(tl, get_tag(tl, s, "v"), "1")) return ZKE_D_INCOMPATIBLE_VERSION;
  const tag_t* ti = get_tag(tl, s, "i");
  const tag_t* td = get_tag(tl, s, "d");
  if (ti) {
    /* STRICTNESS SITE i_must_be_subdomain.  Default: user.ends_with(signing_domain), a plain suffix test on the bytes.
     * ZKE_STRICT_I_SUBDOMAIN: the domain of i= (behind its last '@'; the whole value without one) equals d= or ends with
     * "." d=, ASCII case folded (RFC 6376 §3.5). */
    const uint8_t* iv = tl->tagbuf + ti->val_off;
    const uint8_t* dv = tl->tagbuf + td->val_off;
    if (!(strict & ZKE_STRICT_I_SUBDOMAIN)) {
      if (ti->val_len < td->val_len || memcmp(iv + ti->val_len - td->val_len, dv, td->val_len)) return ZKE_D_DOMAIN_MISMATCH;
    } else {
      size_t ds = 0;
      for (size_t k = 0; k < ti->val_len; k++) if (iv[k] == '@') ds = k + 1;
      const size_t il = ti->val_len - ds;
      if (il < td->val_len) return ZKE_D_DOMAIN_MISMATCH;
      for (size_t k = 0; k < td->val_len; k++)
        if (lower(iv[ti->val_len - td->val_len + k]) != lower(dv[k])) return ZKE_D_DOMAIN_MISMATCH;
      if (il > td->val_len && iv[ti->val_len - td->val_len - 1] != '.') return ZKE_D_DOMAIN_MISMATCH;
    }
  }
  {
    const tag_t* th = get_tag(tl, s, "h");
    const uint8_t* h = tl->tagbuf + th->val_off;
    int found = 0;
    size_t st = 0;
    for (size_t i = 0; i <= th->val_len; i++) {
      if (i == th->val_len || h[i] == ':') {
        if (i - st == 4 && lower(h[st]) == 'f' && lower(h[st + 1]) == 'r' && lower(h[st + 2]) == 'o' && lower(h[st + 3]) == 'm')
          found = 1;
        st = i + 1;
      }
    }
    if (!found) return ZKE_D_FROM_NOT_SIGNED;
  }
  const tag_t* tq = get_tag(tl, s, "q");
  if (tq && !tag_eq(tl, tq, "dns/txt")) return ZKE_D_BAD_QUERY_METHOD;
  const tag_t* tx = (strict & ZKE_STRICT_EXPIRY_X) ? get_tag(tl, s, "x") : NULL;
  if (tx) {
    /* STRICTNESS SITE enforce_expiry_x.  Default: x= is ignored — a zkVM guest has no clock.  ZKE_STRICT_EXPIRY_X:
     * cloudflare/dkim's rule — x= parsed as i64 (str::parse: optional sign, digits, no overflow; anything else counts as
     * 0), fifteen minutes of drift allowed, expired when now > x + 900. */
    const uint8_t* xs = tl->tagbuf + tx->val_off;
    size_t k = 0, xn = tx->val_len;
    int neg = 0, okx = xn > 0;
    if (okx && (xs[0] == '+' || xs[0] == '-')) { neg = xs[0] == '-'; k = 1; okx = xn > 1; }
    uint64_t mag = 0;
    const uint64_t lim = neg ? (1ull << 63) : (1ull << 63) - 1;
    for (; okx && k < xn; k++) {
      if (xs[k] < '0' || xs[k] > '9') { okx = 0; break; }
      const uint64_t dgt = (uint64_t)(xs[k] - '0');
      if (mag > (lim - dgt) / 10) { okx = 0; break; }
      mag = mag * 10 + dgt;
    }
    const int64_t x = okx ? (neg ? (int64_t)(0 - mag) : (int64_t)mag) : 0;
    const int64_t deadline = x > INT64_MAX - 900 ? INT64_MAX : x + 900;
    if ((int64_t)now > deadline) return ZKE_D_SIG_EXPIRED;
  }
  return 0;
}

/* cfdkim canonicalize_body_{simple,relaxed} (RFC 6376 §3.4.3 / §3.4.4) */
size_t zko_canon_body(const uint8_t* body, size_t len, int relaxed, uint8_t* out) {
  size_t o = 0;
  if (!relaxed) {
    if (len == 0) { out[0] = '\r'; out[1] = '\n'; return 2; }
    while (len >= 4 && memcmp(body + len - 4, "\r\n\r\n", 4) == 0) len -= 2;
    memcpy(out, body, len);
    return len;
  }
  /* tabs -> SP, collapse SP runs, drop the SP in front of CRLF */
  int prev_sp = 0;
  for (size_t i = 0; i < len; i++) {
    uint8_t c = body[i] == '\t' ? ' ' : body[i];
    if (c == ' ') {
      if (prev_sp) continue;
      prev_sp = 1;
      size_t j = i + 1; /* end of this WSP run */
      while (j < len && (body[j] == ' ' || body[j] == '\t')) j++;
      if (j + 1 < len && body[j] == '\r' && body[j + 1] == '\n') continue;
      out[o++] = ' ';
    } else {
      prev_sp = 0;
      out[o++] = c;
    }
  }
  while (o >= 4 && memcmp(out + o - 4, "\r\n\r\n", 4) == 0) o -= 2;
  if (o > 0 && !(o >= 2 && out[o - 2] == '\r' && out[o - 1] == '\n')) { out[o++] = '\r'; out[o++] = '\n'; }
  return o;
}

/* cfdkim canonicalize_header_{simple,relaxed}(key, value) (RFC 6376 §3.4.1 / §3.4.2).
 * simple rebuilds "key: value CRLF" from mailparse's (key, raw value) pair. */
size_t zko_canon_header(const uint8_t* key, size_t klen, const uint8_t* val, size_t vlen, int relaxed, uint8_t* out) {
  size_t o = 0;
  if (!relaxed) {
    memcpy(out, key, klen); o = klen;
    out[o++] = ':'; out[o++] = ' ';
    memcpy(out + o, val, vlen); o += vlen;
    out[o++] = '\r'; out[o++] = '\n';
    return o;
  }
  size_t kl = klen;
  while (kl > 0 && (key[kl - 1] == ' ' || key[kl - 1] == '\t')) kl--;
  for (size_t i = 0; i < kl; i++) out[o++] = lower(key[i]);
  out[o++] = ':';
  size_t vstart = o;
  int prev_sp = 0;
  for (size_t i = 0; i < vlen; i++) {
    uint8_t c = val[i];
    if (c == '\r' && i + 1 < vlen && val[i + 1] == '\n') { i++; continue; } /* unfold */
    if (c == '\t') c = ' ';
    if (c == ' ') {
      if (prev_sp) continue;
      prev_sp = 1;
    } else {
      prev_sp = 0;
    }
    out[o++] = c;
  }
  while (o > vstart && out[o - 1] == ' ') o--;              /* trim end */
  if (o > vstart && out[vstart] == ' ') {                    /* trim start (one SP after collapsing) */
    memmove(out + vstart, out + vstart + 1, o - vstart - 1);
    o--;
  }
  out[o++] = '\r'; out[o++] = '\n';
  return o;
}

static int key_ieq(const uint8_t* a, size_t al, const uint8_t* b, size_t bl) {
  if (al != bl) return 0;
  for (size_t i = 0; i < al; i++)
    if (lower(a[i]) != lower(b[i])) return 0;
  return 1;
}

typedef struct {
  const uint8_t* raw; size_t len;
  uint32_t* spans; long nh;
  size_t body_off, body_len; /* cfdkim get_body: after the first CRLFCRLF */
} parsed_t;

typedef struct {
  int hdr_relaxed, body_relaxed, has_len, sha1;
  uint64_t len_tag;
  uint8_t* preimage; size_t preimage_len;
  uint8_t* cbody; size_t cbody_full, cbody_len;
  uint8_t sig[ZKE_MAX_RSA_BYTES + 4]; long sig_len;
} canon_t;

static void find_body(parsed_t* pm) {
  pm->body_off = pm->len; pm->body_len = 0;
  for (size_t i = 0; i + 4 <= pm->len; i++)
    if (pm->raw[i] == '\r' && pm->raw[i + 1] == '\n' && pm->raw[i + 2] == '\r' && pm->raw[i + 3] == '\n') {
      pm->body_off = i + 4; pm->body_len = pm->len - pm->body_off;
      return;
    }
}

/* cfdkim hash::select_headers + compute_headers_hash preimage */
static size_t build_preimage(const parsed_t* pm, const uint8_t* sv, size_t svl, const taglist_t* tl, int relaxed, uint8_t* out, uint32_t strict) {
  size_t o = 0;
  const tag_t* th = get_tag(tl, sv, "h");
  const uint8_t* h = tl->tagbuf + th->val_off;
  long* last = (long*)malloc(sizeof(long) * (th->val_len + 2));
  uint32_t* nm = (uint32_t*)malloc(sizeof(uint32_t) * 2 * (th->val_len + 2));
  int nn = 0;
  size_t st = 0;
  for (size_t i = 0; i <= th->val_len; i++) {
    if (i == th->val_len || h[i] == ':') {
      /* last_index is keyed by lowercased name */
      long start = pm->nh;
      for (int k = 0; k < nn; k++)
        if (key_ieq(h + nm[2 * k], nm[2 * k + 1], h + st, i - st)) start = last[k];
      long found = -1;
      for (long x = start - 1; x >= 0; x--) {
        const uint32_t* sp = pm->spans + 4 * x;
        if (key_ieq(pm->raw + sp[0], sp[1] - sp[0], h + st, i - st)) { found = x; break; }
      }
      if (found >= 0) {
        const uint32_t* sp = pm->spans + 4 * found;
        o += zko_canon_header(pm->raw + sp[0], sp[1] - sp[0], pm->raw + sp[2], sp[3] - sp[2], relaxed, out + o);
      }
      nm[2 * nn] = (uint32_t)st; nm[2 * nn + 1] = (uint32_t)(i - st);
      last[nn] = found >= 0 ? found : 0;
      nn++;
      st = i + 1;
    }
  }
  free(last); free(nm);
  /* DKIM-Signature itself with the raw b= value removed (String::replace of every occurrence) */
  const tag_t* tb = get_tag(tl, sv, "b");
  size_t bl = tb->raw_e - tb->raw_s;
  uint8_t* tmp = (uint8_t*)malloc(svl + 1);
  size_t tn = 0;
  /* STRICTNESS SITE b_removes_own_span_only.  Default: String::replace — every occurrence of the raw b= value goes.
   * ZKE_STRICT_B_OWN_SPAN: only the tag's own span is emptied. */
  if (bl == 0) {
    memcpy(tmp, sv, svl); tn = svl;
  } else if (strict & ZKE_STRICT_B_OWN_SPAN) {
    memcpy(tmp, sv, tb->raw_s); tn = tb->raw_s;
    memcpy(tmp + tn, sv + tb->raw_e, svl - tb->raw_e); tn += svl - tb->raw_e;
  } else {
    for (size_t i = 0; i < svl;) {
      if (i + bl <= svl && memcmp(sv + i, sv + tb->raw_s, bl) == 0) { i += bl; continue; }
      tmp[tn++] = sv[i++];
    }
  }
  o += zko_canon_header((const uint8_t*)"DKIM-Signature", 14, tmp, tn, relaxed, out + o);
  o -= 2; /* without the trailing CRLF */
  free(tmp);
  return o;
}

/* usize::from_str: optional '+', then decimal digits, no overflow */
static int parse_usize(const uint8_t* s, size_t n, uint64_t* out) {
  size_t i = 0;
  if (n && s[0] == '+') i = 1;
  if (i >= n) return -1;
  uint64_t v = 0;
  for (; i < n; i++) {
    if (s[i] < '0' || s[i] > '9') return -1;
    uint64_t d = s[i] - '0';
    if (v > (UINT64_MAX - d) / 10) return -1;
    v = v * 10 + d;
  }
  *out = v;
  return 0;
}

/* c= / a= / l= handling + both canonicalisations for one validated signature.
 * returns 0 or a ZKE_D_* detail */
/* *algo_unsupported (historic name) = 1 when a=ed25519-sha256, 0 for the rsa-* algorithms */
static int canon_for_sig(const parsed_t* pm, const uint8_t* sv, size_t svl, const taglist_t* tl, canon_t* c, int* algo_unsupported, uint32_t strict) {
  const tag_t* tc = get_tag(tl, sv, "c");
  c->hdr_relaxed = c->body_relaxed = 0;
  if (tc) {
    if (tag_eq(tl, tc, "simple/simple") || tag_eq(tl, tc, "simple")) {}
    else if (tag_eq(tl, tc, "relaxed/simple") || tag_eq(tl, tc, "relaxed")) c->hdr_relaxed = 1;
    else if (tag_eq(tl, tc, "simple/relaxed")) c->body_relaxed = 1;
    else if (tag_eq(tl, tc, "relaxed/relaxed")) { c->hdr_relaxed = 1; c->body_relaxed = 1; }
    else return ZKE_D_BAD_CANON;
  }
  if (algo_unsupported) {
    const tag_t* ta = get_tag(tl, sv, "a");
    *algo_unsupported = 0;
    c->sha1 = 0;
    if (tag_eq(tl, ta, "rsa-sha256")) {}
    else if (tag_eq(tl, ta, "rsa-sha1")) c->sha1 = 1;
    else if (tag_eq(tl, ta, "ed25519-sha256")) *algo_unsupported = 1;      /* RFC 8463: SHA-256 hashes, Ed25519 signature */
    else return ZKE_D_BAD_ALGO;
  }
  c->cbody_full = zko_canon_body(pm->raw + pm->body_off, pm->body_len, c->body_relaxed, c->cbody);
  c->cbody_len = c->cbody_full;
  const tag_t* tlen = get_tag(tl, sv, "l");
  c->has_len = 0;
  if (tlen) {
    if (parse_usize(tl->tagbuf + tlen->val_off, tlen->val_len, &c->len_tag)) return ZKE_D_BAD_LENGTH;
    c->has_len = 1;
    if (c->len_tag < c->cbody_len) c->cbody_len = (size_t)c->len_tag;
  }
  c->preimage_len = build_preimage(pm, sv, svl, tl, c->hdr_relaxed, c->preimage, strict);
  return 0;
}

static int value_has_non_ascii(const uint8_t* s, size_t n) {
  for (size_t i = 0; i < n; i++)
    if (s[i] >= 0x80) return 1;
  return 0;
}

/* ======================================================= QP soft line breaks ==
 * core/src/email.rs:61-86 remove_quoted_printable_soft_breaks */
void zko_remove_qp_soft_breaks(const uint8_t* body, size_t len, uint8_t* out) {
  size_t o = 0;
  for (size_t i = 0; i < len;) {
    if (body[i] == '=' && i + 2 < len && body[i + 1] == '\r' && body[i + 2] == '\n') { i += 3; continue; }
    out[o++] = body[i++];
  }
  memset(out + o, 0, len - o); /* email.rs:79 */
}

/* ============================================================= dense DFA =====
 * regex-automata 0.4.9 (Cargo.lock:2130): dense::DFA::from_bytes (core/src/regex.rs:32-33),
 * dfa::regex::Regex::find_iter (core/src/regex.rs:36).  Wire format as written by
 * to_bytes_little_endian (helpers/src/regex.rs:8-13): SURVEY.md Appendix A.3. */
typedef struct {
  int valid;
  uint32_t has_empty, is_utf8, always_anchored;
  uint32_t state_len, stride2, alphabet_len;
  uint8_t classes[256];
  uint32_t* table; size_t table_len;
  uint32_t start_kind; uint8_t start_map[256]; uint32_t start_stride; uint32_t start_pattern_len;
  uint32_t* starts; size_t starts_len;
  uint32_t sp_max, quit_id, min_match, max_match, min_accel, max_accel, min_start, max_start;
  uint8_t quitset[32]; int quitset_nonempty;
} dfa_t;

static uint32_t rd32(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }

/* 0 when the blob deserialises, else the section at which from_bytes gives up (ZKE_D_DFA_LABEL .. ZKE_D_DFA_QUITSET): the
 * `detail` of ZKE_DFA_DECODE_FAIL, the same section names as the engine's parser (csrc/dfa_registry.hip.h) */
static int dfa_parse(const uint8_t* b, size_t n, dfa_t* d) {
  int sec = ZKE_D_DFA_LABEL;
  static const char LABEL[] = "rust-regex-automata-dfa-dense";
  memset(d, 0, sizeof *d);
  size_t p = 0;
  while (p < n && p < 7 && b[p] == 0) p++; /* wire::skip_initial_padding */
#define NEED(k) do { if (n - p < (size_t)(k)) return sec; } while (0)
  NEED(32);
  if (memcmp(b + p, LABEL, 29) || b[p + 29] != 0) return sec;
  p += 32;
  sec = ZKE_D_DFA_ENDIAN_VERSION;
  NEED(4); if (rd32(b + p) != 0xFEFF) return sec; p += 4;
  NEED(4); if (rd32(b + p) != 2) return sec; p += 4;
  NEED(4); p += 4; /* unused */
  /* Flags::from_bytes: ONE u32 bit set — bit 0 has_empty, bit 1 is_utf8, bit 2 is_always_start_anchored (other bits ignored).
   * SURVEY Appendix A.3 recalled three u32s here; blobs written by regex-automata itself (tests/golden/regex_automata_*.dfa)
   * show one, and everything behind it as A.3 has it. */
  sec = ZKE_D_DFA_FLAGS;
  NEED(4);
  { uint32_t fl = rd32(b + p); d->has_empty = fl & 1u; d->is_utf8 = (fl >> 1) & 1u; d->always_anchored = (fl >> 2) & 1u; }
  p += 4;
  /* transition table */
  sec = ZKE_D_DFA_TRANSITIONS;
  NEED(8 + 256);
  d->state_len = rd32(b + p); d->stride2 = rd32(b + p + 4); p += 8;
  memcpy(d->classes, b + p, 256); p += 256;
  if (d->stride2 < 1 || d->stride2 > 9) return sec;
  d->alphabet_len = (uint32_t)d->classes[255] + 2;
  if (d->alphabet_len > (1u << d->stride2)) return sec;
  for (int i = 0; i < 256; i++)                    /* ByteClasses::from_bytes: no class beyond the alphabet (a walk would leave the checked columns) */
    if (d->classes[i] >= d->alphabet_len) return sec;
  if (d->state_len > (1u << 26)) return sec;
  d->table_len = (size_t)d->state_len << d->stride2;
  NEED(d->table_len * 4);
  d->table = (uint32_t*)malloc(d->table_len * 4 + 4);
  for (size_t i = 0; i < d->table_len; i++) d->table[i] = rd32(b + p + 4 * i);
  p += d->table_len * 4;
  uint32_t stride = 1u << d->stride2;
  for (size_t s = 0; s < d->state_len; s++)
    for (uint32_t c = 0; c < d->alphabet_len; c++) {
      uint32_t id = d->table[(s << d->stride2) + c];
      if (id >= d->table_len || (id & (stride - 1))) return sec; /* tt.is_valid */
    }
  /* start table */
  sec = ZKE_D_DFA_START_TABLE;
  NEED(4 + 256 + 16);
  d->start_kind = rd32(b + p); p += 4;
  if (d->start_kind > 2) return sec;
  memcpy(d->start_map, b + p, 256); p += 256;
  for (int i = 0; i < 256; i++) if (d->start_map[i] >= 6) return sec;
  d->start_stride = rd32(b + p); p += 4;
  if (d->start_stride != 6) return sec;
  d->start_pattern_len = rd32(b + p); p += 4;
  p += 8; /* universal unanchored / anchored start (not needed for search) */
  size_t npat = d->start_pattern_len == 0xFFFFFFFFu ? 0 : d->start_pattern_len;
  if (npat > (1u << 20)) return sec;
  d->starts_len = 2 * 6 + 6 * npat;
  NEED(d->starts_len * 4);
  d->starts = (uint32_t*)malloc(d->starts_len * 4);
  for (size_t i = 0; i < d->starts_len; i++) {
    d->starts[i] = rd32(b + p + 4 * i);
    if (d->starts[i] >= d->table_len || (d->starts[i] & (stride - 1))) return sec;
  }
  p += d->starts_len * 4;
  /* match states */
  sec = ZKE_D_DFA_MATCH_STATES;
  NEED(4);
  uint32_t ms_len = rd32(b + p); p += 4;
  if (ms_len > d->state_len) return sec;
  NEED((size_t)ms_len * 8 + 8);
  p += (size_t)ms_len * 8;
  p += 4; /* pattern_len */
  uint32_t idlen = rd32(b + p); p += 4;
  if (idlen > (1u << 24)) return sec;
  NEED((size_t)idlen * 4);
  p += (size_t)idlen * 4;
  /* special */
  sec = ZKE_D_DFA_SPECIAL;
  NEED(32);
  d->sp_max = rd32(b + p); d->quit_id = rd32(b + p + 4); d->min_match = rd32(b + p + 8); d->max_match = rd32(b + p + 12);
  d->min_accel = rd32(b + p + 16); d->max_accel = rd32(b + p + 20); d->min_start = rd32(b + p + 24); d->max_start = rd32(b + p + 28);
  p += 32;
  if (d->min_match > d->max_match || d->min_accel > d->max_accel || d->min_start > d->max_start) return sec;
  if ((d->min_match == 0) != (d->max_match == 0)) return sec;
  if (d->max_match > d->sp_max || d->max_accel > d->sp_max || d->max_start > d->sp_max) return sec;
  if (d->sp_max >= d->table_len && d->table_len) return sec;
  {
    uint32_t nm = d->max_match ? ((d->max_match - d->min_match) >> d->stride2) + 1 : 0;
    if (nm != ms_len) return ZKE_D_DFA_MATCH_STATES;
  }
  /* accelerators: u32 count then 8 bytes each */
  sec = ZKE_D_DFA_ACCELS;
  NEED(4);
  uint32_t acc = rd32(b + p); p += 4;
  if (acc > d->state_len) return sec;
  NEED((size_t)acc * 8);
  p += (size_t)acc * 8;
  sec = ZKE_D_DFA_QUITSET;
  NEED(32);
  memcpy(d->quitset, b + p, 32); p += 32;
  for (int i = 0; i < 32; i++) if (d->quitset[i]) d->quitset_nonempty = 1;
#undef NEED
  d->valid = 1;
  return 0;
}

static int dfa_is_match(const dfa_t* d, uint32_t s) { return s != 0 && d->min_match <= s && s <= d->max_match; }
static int dfa_is_quit(const dfa_t* d, uint32_t s) { return s != 0 && s == d->quit_id; }
static uint32_t dfa_next(const dfa_t* d, uint32_t s, uint8_t b) { return d->table[s + d->classes[b]]; }
static uint32_t dfa_eoi(const dfa_t* d, uint32_t s) { return d->table[s + d->alphabet_len - 1]; }

/* Automaton::start_state: returns 0 ok, -1 quit, -2 unsupported anchored mode */
static int dfa_start(const dfa_t* d, int anchored, int have_look, uint8_t look, uint32_t* sid) {
  uint32_t st = 2; /* Start::Text */
  if (have_look) {
    if (d->quitset_nonempty && (d->quitset[look >> 3] >> (look & 7) & 1)) return -1;
    st = d->start_map[look];
  }
  if (!anchored) {
    if (d->start_kind == 2) return -2;
    *sid = d->starts[st];
  } else {
    if (d->start_kind == 1) return -2;
    *sid = d->starts[6 + st];
  }
  return 0;
}

/* dfa/search.rs find_fwd (leftmost, earliest=false). 1 match / 0 none / -1 quit */
static int dfa_find_fwd(const dfa_t* d, const uint8_t* hay, size_t hlen, size_t start, size_t end, int anchored, size_t* mend) {
  if (start > end) return 0;
  uint32_t sid;
  int r = dfa_start(d, anchored, start > 0, start > 0 ? hay[start - 1] : 0, &sid);
  if (r) return -1;
  int have = 0;
  for (size_t at = start; at < end; at++) {
    sid = dfa_next(d, sid, hay[at]);
    if (sid <= d->sp_max) {
      if (dfa_is_match(d, sid)) { have = 1; *mend = at; }
      else if (sid == 0) return have;
      else if (dfa_is_quit(d, sid)) return -1;
    }
  }
  if (end < hlen) {
    sid = dfa_next(d, sid, hay[end]);
    if (dfa_is_match(d, sid)) { have = 1; *mend = end; }
    else if (dfa_is_quit(d, sid)) return -1;
  } else {
    sid = dfa_eoi(d, sid);
    if (dfa_is_match(d, sid)) { have = 1; *mend = hlen; }
  }
  return have;
}
/* find_rev: anchored reverse search over [start,end) */
static int dfa_find_rev(const dfa_t* d, const uint8_t* hay, size_t hlen, size_t start, size_t end, size_t* mstart) {
  uint32_t sid;
  int r = dfa_start(d, 1, end < hlen, end < hlen ? hay[end] : 0, &sid);
  if (r) return -1;
  int have = 0;
  if (start < end) {
    for (size_t at = end; at-- > start;) {
      sid = dfa_next(d, sid, hay[at]);
      if (sid <= d->sp_max) {
        if (dfa_is_match(d, sid)) { have = 1; *mstart = at + 1; }
        else if (sid == 0) return have;
        else if (dfa_is_quit(d, sid)) return -1;
      }
    }
  }
  if (start > 0) {
    sid = dfa_next(d, sid, hay[start - 1]);
    if (dfa_is_match(d, sid)) { have = 1; *mstart = start; }
    else if (dfa_is_quit(d, sid)) return -1;
  } else {
    sid = dfa_eoi(d, sid);
    if (dfa_is_match(d, sid)) { have = 1; *mstart = 0; }
  }
  return have;
}
static int is_char_boundary(const uint8_t* hay, size_t hlen, size_t off) {
  if (off >= hlen) return off == hlen;
  return (int8_t)hay[off] >= -0x40;
}
/* Automaton::try_search_fwd incl. util::empty::skip_splits_fwd */
static int dfa_search_fwd(const dfa_t* d, const uint8_t* hay, size_t hlen, size_t start, size_t end, size_t* mend) {
  int r = dfa_find_fwd(d, hay, hlen, start, end, 0, mend);
  if (r <= 0) return r;
  if (!(d->has_empty && d->is_utf8)) return 1;
  while (!is_char_boundary(hay, hlen, *mend)) {
    start++;
    r = dfa_find_fwd(d, hay, hlen, start, end, 0, mend);
    if (r <= 0) return r;
  }
  return 1;
}
typedef struct { dfa_t fwd, rev; uint32_t detail; } regex_t_;      /* detail: 0, or the ZKE_D_DFA_* section that does not parse */
/* dfa::regex::Regex::try_search */
static int regex_search(const regex_t_* re, const uint8_t* hay, size_t hlen, size_t start, size_t end, size_t* ms, size_t* me) {
  size_t e;
  int r = dfa_search_fwd(&re->fwd, hay, hlen, start, end, &e);
  if (r <= 0) return r;
  *me = e;
  if (start == e) { *ms = e; return 1; }
  if (re->fwd.always_anchored) { *ms = start; return 1; }
  size_t s;
  r = dfa_find_rev(&re->rev, hay, hlen, start, e, &s);
  if (r <= 0) return -1; /* .expect("reverse search must match if forward search does") */
  *ms = s;
  return 1;
}

#define MAX_DFAS 1024
static regex_t_* g_re[MAX_DFAS];
static uint32_t g_nre = 0;
static pthread_mutex_t g_re_mu = PTHREAD_MUTEX_INITIALIZER;

int zko_dfa_register(const uint8_t* fwd, size_t fwd_len, const uint8_t* bwd, size_t bwd_len, uint32_t* out_id) {
  pthread_mutex_lock(&g_re_mu);
  if (g_nre >= MAX_DFAS) { pthread_mutex_unlock(&g_re_mu); return -1; }
  regex_t_* re = (regex_t_*)calloc(1, sizeof *re);
  int det = dfa_parse(fwd, fwd_len, &re->fwd);
  if (!det) { det = dfa_parse(bwd, bwd_len, &re->rev); if (det) det += ZKE_D_DFA_BWD_OFFSET; }
  else dfa_parse(bwd, bwd_len, &re->rev);
  re->detail = (uint32_t)det;
  g_re[g_nre] = re;
  *out_id = g_nre++;
  pthread_mutex_unlock(&g_re_mu);
  return 0;
}
/* 0, or the section at which the pair does not deserialise (the engine's zke_dfa_status); -1: no such id */
long zko_dfa_status(uint32_t id) {
  long r = -1;
  pthread_mutex_lock(&g_re_mu);
  if (id < g_nre) r = (long)g_re[id]->detail;
  pthread_mutex_unlock(&g_re_mu);
  return r;
}
void zko_dfa_reset(void) {
  pthread_mutex_lock(&g_re_mu);
  for (uint32_t i = 0; i < g_nre; i++) {
    free(g_re[i]->fwd.table); free(g_re[i]->fwd.starts); free(g_re[i]->rev.table); free(g_re[i]->rev.starts);
    free(g_re[i]);
  }
  g_nre = 0;
  pthread_mutex_unlock(&g_re_mu);
}

/* util::iter::Searcher (find_iter): non-overlapping, an empty match abutting the previous
 * match end restarts one byte later */
long zko_regex_find_iter(uint32_t id, const uint8_t* hay, size_t len, uint32_t* spans, size_t max_spans) {
  if (id >= g_nre || !g_re[id]->fwd.valid || !g_re[id]->rev.valid) return -2;
  const regex_t_* re = g_re[id];
  size_t start = 0, n = 0;
  int have_last = 0; size_t last_end = 0;
  for (;;) {
    size_t ms, me;
    int r = regex_search(re, hay, len, start, len, &ms, &me);
    if (r < 0) return -1;
    if (r == 0) break;
    if (ms == me && have_last && me == last_end) {
      start += 1;
      r = regex_search(re, hay, len, start, len, &ms, &me);
      if (r < 0) return -1;
      if (r == 0) break;
    }
    if (n < max_spans) { spans[2 * n] = (uint32_t)ms; spans[2 * n + 1] = (uint32_t)me; }
    n++;
    start = me; have_last = 1; last_end = me;
  }
  return (long)n;
}

static int utf8_valid(const uint8_t* s, size_t n) {
  size_t i = 0;
  while (i < n) {
    uint8_t c = s[i];
    if (c < 0x80) { i++; continue; }
    size_t need; uint32_t lo;
    if (c >= 0xC2 && c <= 0xDF) { need = 1; lo = 0x80; }
    else if (c >= 0xE0 && c <= 0xEF) { need = 2; lo = 0x800; }
    else if (c >= 0xF0 && c <= 0xF4) { need = 3; lo = 0x10000; }
    else return 0;
    if (n - i <= need) return 0;
    uint32_t cp = c & (0x3Fu >> need);
    for (size_t k = 1; k <= need; k++) {
      if ((s[i + k] & 0xC0) != 0x80) return 0;
      cp = (cp << 6) | (s[i + k] & 0x3F);
    }
    if (cp < lo || cp > 0x10FFFF || (cp >= 0xD800 && cp <= 0xDFFF)) return 0;
    i += need + 1;
  }
  return 1;
}
static int contains(const uint8_t* h, size_t hl, const uint8_t* nd, size_t nl) {
  if (nl == 0) return 1;
  if (nl > hl) return 0;
  for (size_t i = 0; i + nl <= hl; i++)
    if (h[i] == nd[0] && memcmp(h + i, nd, nl) == 0) return 1;
  return 0;
}

/* core/src/regex.rs:15-53 process_regex_parts for the parts [p0,p1) of email i.
 * returns 0 ok, else a ZKE_D_* detail; fills the match fields of `out` */
static int process_parts(const zke_batch* in, uint32_t i, const uint32_t* ids, uint32_t np, uint32_t pbase,
                         const uint8_t* hay, size_t hlen, zke_result* out, int* decode_fail) {
  uint32_t P = in->n_header_parts + in->n_body_parts;
  for (uint32_t k = 0; k < np; k++) {
    uint32_t id = ids[k];
    out->regex_part = pbase + k;
    out->match_count = 0; out->match_start = 0; out->match_end = 0;
    if (id >= g_nre) { *decode_fail = ZKE_D_DFA_UNREGISTERED; return ZKE_D_NONE; }
    if (g_re[id]->detail) { *decode_fail = (int)g_re[id]->detail; return ZKE_D_NONE; }
    uint32_t sp[4];
    long n = zko_regex_find_iter(id, hay, hlen, sp, 2);
    if (n == -1) return ZKE_D_RE_QUIT;
    out->match_count = n > 2 ? 2 : (uint32_t)n;
    if (n >= 1) { out->match_start = sp[0]; out->match_end = sp[1]; }
    if (n != 1) return ZKE_D_RE_MATCH_COUNT;
    if (in->cap_off) {
      uint32_t c0 = in->cap_off[(size_t)i * P + pbase + k], c1 = in->cap_off[(size_t)i * P + pbase + k + 1];
      const uint8_t* m = hay + sp[0]; size_t ml = sp[1] - sp[0];
      int mvalid = -1;
      for (uint32_t c = c0; c < c1; c++) {
        const uint8_t* cs = in->cap_blob + in->cap_str_off[c];
        size_t cl = in->cap_str_off[c + 1] - in->cap_str_off[c];
        static const uint8_t FFFD[3] = {0xEF, 0xBF, 0xBD};
        if (contains(cs, cl, FFFD, 3)) {
          if (mvalid < 0) mvalid = utf8_valid(m, ml);
          if (!mvalid) return ZKE_D_U_CAPTURE_FFFD;
        }
        /* String::from_utf8_lossy(match).contains(capture) == byte containment when the capture
         * holds no U+FFFD (a valid UTF-8 needle cannot straddle a replaced sequence) */
        if (!contains(m, ml, cs, cl)) return ZKE_D_RE_CAPTURE_MISSING;
      }
    }
  }
  return 0;
}

/* ================================================= verify_email[_with_regex] ==
 * core/src/circuits.rs:9-68 + core/src/email.rs:25-36, one email */
typedef struct {
  uint32_t* spans; uint8_t* tagbuf; uint8_t* preimage; uint8_t* cbody; uint8_t* clean;
  size_t cap;
} scratch_t;

static void scratch_fit(scratch_t* s, size_t raw_len) {
  size_t need = raw_len * 2 + 4096;
  if (need <= s->cap) return;
  free(s->spans); free(s->tagbuf); free(s->preimage); free(s->cbody); free(s->clean);
  s->cap = need;
  s->spans = (uint32_t*)malloc(sizeof(uint32_t) * 4 * ZKE_MAX_HEADERS);
  s->tagbuf = (uint8_t*)malloc(need);
  s->preimage = (uint8_t*)malloc(need * 2);
  s->cbody = (uint8_t*)malloc(need);
  s->clean = (uint8_t*)malloc(need);
}

static void dbg_copy(uint8_t* base, size_t stride, uint32_t i, const uint8_t* src, size_t n) {
  if (!base) return;
  if (n > stride) n = stride;
  memset(base + (size_t)i * stride, 0, stride);
  memcpy(base + (size_t)i * stride, src, n);
}

static void verify_one(const zke_batch* in, uint32_t i, zke_result* out, zke_debug_out* dbg, scratch_t* sc, uint32_t strict, uint64_t now) {
  memset(out, 0, sizeof *out);
  out->regex_part = 0xFFFFFFFFu;
  const uint8_t* raw = in->raw_blob + in->raw_off[i];
  size_t raw_len = (size_t)(in->raw_off[i + 1] - in->raw_off[i]);
  const uint8_t* dom = in->domain_blob + in->domain_off[i];
  size_t dom_len = (size_t)(in->domain_off[i + 1] - in->domain_off[i]);
  const uint8_t* key = in->key_blob + in->key_off[i];
  size_t key_len = (size_t)(in->key_off[i + 1] - in->key_off[i]);
  if (raw_len >= (1ull << 31)) { out->status = ZKE_UNSUPPORTED; out->detail = ZKE_D_U_EMAIL_TOO_LARGE; return; }
  scratch_fit(sc, raw_len);

  /* --- verify_dkim: core/src/email.rs:25-36 */
  parsed_t pm = {raw, raw_len, sc->spans, 0, 0, 0};
  size_t mp_body;
  pm.nh = zko_parse_headers(raw, raw_len, sc->spans, ZKE_MAX_HEADERS, &mp_body);   /* email.rs:26 */
  if (pm.nh < 0) {
    out->detail = (uint32_t)(-pm.nh);
    out->status = out->detail == ZKE_D_U_TOO_MANY_HEADERS ? ZKE_UNSUPPORTED : ZKE_PARSE_FAIL;
    return;
  }
  out->n_headers = (uint32_t)pm.nh;
  find_body(&pm);
  out->body_offset = (uint32_t)pm.body_off;
  {
    uint32_t md = 0;
    const uint32_t mr = mime_walk(raw, 0, raw_len, 0, &md);      /* still email.rs:26: the subparts */
    if (mr) { out->status = mr; out->detail = md; return; }
  }

  uint8_t mod[ZKE_MAX_RSA_BYTES + 8]; uint32_t mod_len = 0; uint64_t e = 0;      /* email.rs:28-29 */
  const int ed_key = in->key_type[i] == ZKE_KEY_ED25519;
  if (ed_key) {
    /* VerifyingKey::from_bytes (helpers/src/dkim.rs:103-108 hands over the raw 32 bytes) */
    if (key_len != 32) { out->status = ZKE_KEY_DECODE_FAIL; out->detail = ZKE_D_KEY_DER; return; }
    if (!zko_ed25519_key_decodes(key)) { out->status = ZKE_KEY_DECODE_FAIL; out->detail = ZKE_D_KEY_ED25519_POINT; return; }
  } else if (in->key_type[i] != ZKE_KEY_RSA) {
    out->status = ZKE_KEY_DECODE_FAIL; out->detail = ZKE_D_KEY_TYPE; return;
  }
  int kr = ed_key ? 0 : zko_parse_rsa_pkcs1(key, key_len, mod, &mod_len, &e);
  if (kr) { out->status = ZKE_KEY_DECODE_FAIL; out->detail = (uint32_t)kr; return; }
  if (!ed_key) {
    uint32_t bits = mod_len * 8;
    for (uint8_t t = mod[0]; mod_len && !(t & 0x80) && bits; t <<= 1) bits--;
    out->rsa_bits = (mod_len == 1 && mod[0] == 0) ? 0 : bits;
  }

  /* cfdkim compares signing_domain.to_lowercase() with from_domain.to_lowercase() (Unicode).  d= is ASCII whenever a
   * signature gets that far (non-ASCII DKIM-Signature values are reported above), so folding ASCII alone is exact unless
   * from_domain holds a non-ASCII character whose lower case is ASCII: U+212A KELVIN SIGN -> "k" is the only one. */
  for (size_t q = 0; q + 2 < dom_len; q++)
    if (dom[q] == 0xE2 && dom[q + 1] == 0x84 && dom[q + 2] == 0xAA) { out->status = ZKE_UNSUPPORTED; out->detail = ZKE_D_U_DOMAIN_FOLD; return; }

  /* cfdkim::verify_email_with_key: email.rs:31-33 */
  taglist_t tl; tl.tagbuf = sc->tagbuf;
  canon_t cn; cn.preimage = sc->preimage; cn.cbody = sc->cbody;
  uint32_t last_err = ZKE_D_NEUTRAL, unsupported = 0, sig_ix = 0;
  int passed = 0;
  const uint8_t* pass_sv = NULL; size_t pass_svl = 0;     /* the signature that verified */
  for (long hx = 0; hx < pm.nh && !passed; hx++) {
    const uint32_t* sp = pm.spans + 4 * hx;
    if (!key_ieq(raw + sp[0], sp[1] - sp[0], (const uint8_t*)"DKIM-Signature", 14)) continue;
    uint32_t this_ix = sig_ix++;
    const uint8_t* sv = raw + sp[2]; size_t svl = sp[3] - sp[2];
    if (value_has_non_ascii(sv, svl)) { unsupported = ZKE_D_U_SIG_NON_ASCII; out->sig_index = this_ix; continue; }
    int v = validate_header(sv, svl, &tl, strict, now);
    if (v == ZKE_D_U_TOO_MANY_TAGS || v == ZKE_D_U_SIG_TOO_LONG) { unsupported = v; out->sig_index = this_ix; continue; }
    if (v) { last_err = (uint32_t)v; out->sig_index = this_ix; continue; }
    const tag_t* td = get_tag(&tl, sv, "d");
    if (!key_ieq(tl.tagbuf + td->val_off, td->val_len, dom, dom_len)) continue;
    out->sig_index = this_ix;
    int algo_uns = 0;
    int c = canon_for_sig(&pm, sv, svl, &tl, &cn, &algo_uns, strict);
    if (c == ZKE_D_BAD_CANON || c == ZKE_D_BAD_ALGO) { last_err = (uint32_t)c; continue; }
    /* a= and the key type must name the same scheme; cfdkim's behaviour for a mixed pair is not restated */
    if ((algo_uns != 0) != ed_key) { unsupported = ZKE_D_U_ALGO_ED25519; continue; }
    if (c) { last_err = (uint32_t)c; continue; }
    out->flags = (cn.hdr_relaxed ? ZKE_F_HDR_RELAXED : 0) | (cn.body_relaxed ? ZKE_F_BODY_RELAXED : 0) | (cn.has_len ? ZKE_F_HAS_LENGTH : 0) |
                 (cn.sha1 ? ZKE_F_SHA1 : 0) | (ed_key ? ZKE_F_ED25519 : 0);
    out->canon_header_len = (uint32_t)cn.preimage_len;
    out->canon_body_len = (uint32_t)cn.cbody_len;
    const uint32_t hlen = cn.sha1 ? 20 : 32;
    memset(out->body_hash, 0, 32); memset(out->header_hash, 0, 32);
    if (cn.sha1) { zko_sha1(cn.cbody, cn.cbody_len, out->body_hash); zko_sha1(cn.preimage, cn.preimage_len, out->header_hash); }
    else { zko_sha256(cn.cbody, cn.cbody_len, out->body_hash); zko_sha256(cn.preimage, cn.preimage_len, out->header_hash); }
    if (dbg) {
      dbg_copy(dbg->canon_header, dbg->canon_header_stride, i, cn.preimage, cn.preimage_len);
      dbg_copy(dbg->canon_body, dbg->canon_body_stride, i, cn.cbody, cn.cbody_full);
      if (dbg->canon_body_full_len) dbg->canon_body_full_len[i] = (uint32_t)cn.cbody_full;
    }
    char b64[48];
    size_t bl = zko_b64_encode(out->body_hash, hlen, b64);
    const tag_t* tbh = get_tag(&tl, sv, "bh");
    if (tbh->val_len != bl || memcmp(tl.tagbuf + tbh->val_off, b64, bl)) { last_err = ZKE_D_BODY_HASH_MISMATCH; continue; }
    const tag_t* tb = get_tag(&tl, sv, "b");
    uint8_t* sigbuf = (uint8_t*)malloc(tb->val_len + 4);
    long sl = zko_b64_decode(tl.tagbuf + tb->val_off, tb->val_len, sigbuf);
    if (sl < 0) { free(sigbuf); last_err = ZKE_D_SIG_B64; continue; }
    if (ed_key) {
      /* ed25519-dalek verify_strict over the SHA-256 header hash; a b= that is not 64 bytes cannot be a Signature */
      int ok = sl == 64 && zko_ed25519_verify_strict(key, out->header_hash, 32, sigbuf);
      free(sigbuf);
      if (!ok) { last_err = ZKE_D_SIG_MISMATCH; continue; }
      passed = 1; pass_sv = sv; pass_svl = svl;
      continue;
    }
    if (!(mod[mod_len - 1] & 1) && !(mod_len == 1 && mod[0] == 0)) { free(sigbuf); unsupported = ZKE_D_U_EVEN_MODULUS; continue; }
    uint8_t em[ZKE_MAX_RSA_BYTES + 8];
    int ok = (mod_len >= 1 && !(mod_len == 1 && mod[0] == 0)) &&
             rsa_pkcs1v15_verify(mod, mod_len, e, sigbuf, (uint32_t)sl, out->header_hash, hlen,
                                 cn.sha1 ? SHA1_PREFIX : SHA256_PREFIX, cn.sha1 ? 15 : 19, em);
    if (dbg && dbg->em) dbg_copy(dbg->em, dbg->em_stride, i, em, mod_len);
    free(sigbuf);
    if (!ok) { last_err = ZKE_D_SIG_MISMATCH; continue; }
    passed = 1; pass_sv = sv; pass_svl = svl;
  }
  if (!passed) {                                                       /* circuits.rs:13 */
    if (unsupported) { out->status = ZKE_UNSUPPORTED; out->detail = unsupported; }
    else { out->status = ZKE_DKIM_NOT_PASS; out->detail = last_err; }
    return;
  }
  zko_sha256(dom, dom_len, out->from_domain_hash);                     /* circuits.rs:16 */
  zko_sha256(key, key_len, out->public_key_hash);                      /* circuits.rs:17 */
  if (in->ext_null && in->ext_null[i]) { out->status = ZKE_EXTERNAL_INPUT_NULL; return; } /* circuits.rs:24 */
  if (!in->with_regex) return;

  /* --- verify_email_with_regex: circuits.rs:34-62.  canonicalize_signed_email takes the FIRST
   * DKIM-Signature header and has no from_domain argument. */
  const uint8_t* sv = NULL; size_t svl = 0;
  for (long hx = 0; hx < pm.nh; hx++) {
    const uint32_t* sp = pm.spans + 4 * hx;
    if (key_ieq(raw + sp[0], sp[1] - sp[0], (const uint8_t*)"DKIM-Signature", 14)) { sv = raw + sp[2]; svl = sp[3] - sp[2]; break; }
  }
  /* STRICTNESS SITE canon_takes_verified_signature.  Default: the FIRST DKIM-Signature header, whatever its d=.
   * ZKE_STRICT_CANON_VERIFIED: the signature verify_dkim accepted. */
  if (strict & ZKE_STRICT_CANON_VERIFIED) { sv = pass_sv; svl = pass_svl; }
  if (!sv) { out->status = ZKE_CANON_FAIL; out->detail = ZKE_D_NO_SIGNATURE; return; }
  if (value_has_non_ascii(sv, svl)) { out->status = ZKE_UNSUPPORTED; out->detail = ZKE_D_U_SIG_NON_ASCII; return; }
  int v = validate_header(sv, svl, &tl, strict, now);
  if (v == ZKE_D_U_TOO_MANY_TAGS || v == ZKE_D_U_SIG_TOO_LONG) { out->status = ZKE_UNSUPPORTED; out->detail = (uint32_t)v; return; }
  if (v) { out->status = ZKE_CANON_FAIL; out->detail = (uint32_t)v; return; }
  v = canon_for_sig(&pm, sv, svl, &tl, &cn, NULL, strict);
  if (v) { out->status = ZKE_CANON_FAIL; out->detail = (uint32_t)v; return; }
  /* STRICTNESS SITE canon_ignores_l.  Default: the canonical body is truncated to l=, as on the verify path.
   * ZKE_STRICT_CANON_IGNORES_L: canonicalize_signed_email returns the whole canonical body. */
  if (strict & ZKE_STRICT_CANON_IGNORES_L) cn.cbody_len = cn.cbody_full;
  zko_remove_qp_soft_breaks(cn.cbody, cn.cbody_len, sc->clean);       /* circuits.rs:37 */
  if (dbg) dbg_copy(dbg->clean_body, dbg->clean_body_stride, i, sc->clean, cn.cbody_len);
  int decode_fail = 0;
  int d = process_parts(in, i, in->header_part_ids, in->n_header_parts, 0, cn.preimage, cn.preimage_len, out, &decode_fail);
  if (decode_fail) { out->status = ZKE_DFA_DECODE_FAIL; out->detail = (uint32_t)decode_fail; return; }
  if (d) { out->status = (d == ZKE_D_U_CAPTURE_FFFD) ? ZKE_UNSUPPORTED : ZKE_HEADER_REGEX_FAIL; out->detail = (uint32_t)d; return; }
  d = process_parts(in, i, in->body_part_ids, in->n_body_parts, in->n_header_parts, sc->clean, cn.cbody_len, out, &decode_fail);
  if (decode_fail) { out->status = ZKE_DFA_DECODE_FAIL; out->detail = (uint32_t)decode_fail; return; }
  if (d) { out->status = (d == ZKE_D_U_CAPTURE_FFFD) ? ZKE_UNSUPPORTED : ZKE_BODY_REGEX_FAIL; out->detail = (uint32_t)d; return; }
}

typedef struct { const zke_batch* in; zke_result* out; zke_debug_out* dbg; uint32_t lo, hi; uint32_t strict; uint64_t now; } job_t;
static void* worker(void* a) {
  job_t* j = (job_t*)a;
  scratch_t sc; memset(&sc, 0, sizeof sc);
  for (uint32_t i = j->lo; i < j->hi; i++) verify_one(j->in, i, &j->out[i], j->dbg, &sc, j->strict, j->now);
  free(sc.spans); free(sc.tagbuf); free(sc.preimage); free(sc.cbody); free(sc.clean);
  return NULL;
}
int zko_verify_batch(const zke_batch* in, zke_result* out, zke_debug_out* dbg, int threads) {
  return zko_verify_batch_strict(in, out, dbg, threads, 0, 0);
}
int zko_verify_batch_strict(const zke_batch* in, zke_result* out, zke_debug_out* dbg, int threads, uint32_t strict, uint64_t now) {
  if (!in || !out) return ZKE_E_ARG;
  if (threads < 1) threads = 1;
  if ((uint32_t)threads > in->n) threads = in->n ? (int)in->n : 1;
  if (threads == 1) { job_t j = {in, out, dbg, 0, in->n, strict, now}; worker(&j); return 0; }
  pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * threads);
  job_t* jobs = (job_t*)malloc(sizeof(job_t) * threads);
  for (int t = 0; t < threads; t++) {
    jobs[t].in = in; jobs[t].out = out; jobs[t].dbg = dbg; jobs[t].strict = strict; jobs[t].now = now;
    jobs[t].lo = (uint32_t)((uint64_t)in->n * t / threads);
    jobs[t].hi = (uint32_t)((uint64_t)in->n * (t + 1) / threads);
    pthread_create(&th[t], NULL, worker, &jobs[t]);
  }
  for (int t = 0; t < threads; t++) pthread_join(th[t], NULL);
  free(th); free(jobs);
  return 0;
}
