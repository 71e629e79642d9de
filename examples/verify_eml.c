/* verify_eml.c — the reference's call, from C: one e-mail in, the two witness hashes out.
 *
 *     zkemail_core::verify_email(&Email { from_domain, raw_email, public_key: PublicKey { key, key_type }, external_inputs })
 *                                                                                   core/src/circuits.rs:9-29
 * through the C-ABI of this repository (include/zkemail_amd.h): zke_engine_create, zke_verify_email, zke_abi_encode.
 * Nothing but the header and the shared library is needed — no Python, no torch.
 *
 *     gcc -O2 -I include -o verify_eml examples/verify_eml.c -L zkemail.rs_amd -lzkemail_amd -Wl,-rpath,$PWD/zkemail.rs_amd
 *     ./verify_eml message.eml football.example.com ed25519 d75a980182b10ab7d54bfed3c964073a0ee172f3daa62325af021a68f707511a
 *
 * key: hex of PublicKey.key — PKCS#1 RSAPublicKey DER for "rsa", the 32 raw bytes for "ed25519" (helpers/src/dkim.rs:53-56,103-108).
 * Exit code 0 = the e-mail verifies (the reference returns), 1 = it does not (the reference panics; the site is printed), 2 = usage / I/O.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "zkemail_amd.h"

static unsigned char* slurp(const char* path, size_t* n) {
  FILE* f = fopen(path, "rb");
  if (!f) return NULL;
  fseek(f, 0, SEEK_END);
  long sz = ftell(f);
  fseek(f, 0, SEEK_SET);
  unsigned char* b = (unsigned char*)malloc(sz > 0 ? (size_t)sz : 1);
  *n = b ? fread(b, 1, (size_t)sz, f) : 0;
  fclose(f);
  return b;
}

static int unhex(const char* s, unsigned char* out, size_t cap, size_t* n) {
  size_t l = strlen(s);
  if (l % 2 || l / 2 > cap) return -1;
  for (size_t i = 0; i < l / 2; i++) {
    unsigned v;
    if (sscanf(s + 2 * i, "%2x", &v) != 1) return -1;
    out[i] = (unsigned char)v;
  }
  *n = l / 2;
  return 0;
}

static void hex(const char* label, const unsigned char* b, size_t n) {
  printf("%s", label);
  for (size_t i = 0; i < n; i++) printf("%02x", b[i]);
  printf("\n");
}

int main(int argc, char** argv) {
  if (argc != 5) {
    fprintf(stderr, "usage: %s message.eml from_domain rsa|ed25519 key_hex\n", argv[0]);
    return 2;
  }
  size_t raw_len = 0, key_len = 0;
  unsigned char* raw = slurp(argv[1], &raw_len);
  static unsigned char key[1024];
  if (!raw || unhex(argv[4], key, sizeof key, &key_len)) {
    fprintf(stderr, "cannot read the message or the key\n");
    return 2;
  }
  const unsigned key_type = !strcmp(argv[3], "rsa") ? ZKE_KEY_RSA : !strcmp(argv[3], "ed25519") ? ZKE_KEY_ED25519 : ZKE_KEY_OTHER;

  zke_options opt;
  memset(&opt, 0, sizeof opt);              /* all zeros = the defaults (ABI 0.3) */
  opt.device = -1;                          /* the current device */
  zke_engine* e = NULL;
  if (zke_engine_create(&opt, &e)) {
    fprintf(stderr, "zke_engine_create: %s\n", zke_last_error(NULL));
    return 2;
  }
  zke_result r;
  if (zke_verify_email(e, raw, raw_len, argv[2], strlen(argv[2]), key, key_len, key_type, 0, &r)) {
    fprintf(stderr, "zke_verify_email: %s\n", zke_last_error(e));
    zke_engine_destroy(e);
    return 2;
  }
  printf("status %u detail %u (signature header %u)\n", r.status, r.detail, r.sig_index);
  int rc = 1;
  if (r.status == ZKE_OK) {
    hex("from_domain_hash ", r.from_domain_hash, 32);
    hex("public_key_hash  ", r.public_key_hash, 32);
    /* what a zkVM guest commits: VerificationOutput::from_parts(email, None).abi_encode()   core/src/io.rs:28-44 */
    unsigned char abi[512];
    size_t abi_len = 0;
    if (zke_abi_encode(r.from_domain_hash, r.public_key_hash, NULL, NULL, 0, 0, NULL, NULL, 0, abi, sizeof abi, &abi_len) == 0)
      hex("abi_encode       ", abi, abi_len);
    rc = 0;
  } else {
    printf("the reference panics here: %s\n", zke_status_name(r.status));
  }
  zke_engine_destroy(e);
  free(raw);
  return rc;
}
