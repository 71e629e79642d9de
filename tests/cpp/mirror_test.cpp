// Exercises include/zkemail_core.hpp (the C++ mirror of zkemail_core's API) end to end.
// usage: mirror_test <raw.eml> <key.der> <from_domain> [fwd.dfa bwd.dfa capture]
// prints "OK <from_domain_hash hex> <public_key_hash hex> [matches...]" or "PANIC <status> <detail>", and on success a
// second line "ABI <hex>": VerificationOutput::from_parts(..).abi_encode() (core/src/io.rs:28-44)
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iterator>

#include "zkemail_core.hpp"

static std::vector<uint8_t> slurp(const char* p) {
  std::ifstream f(p, std::ios::binary);
  return std::vector<uint8_t>(std::istreambuf_iterator<char>(f), std::istreambuf_iterator<char>());
}
static void hex(const std::vector<uint8_t>& v) { for (uint8_t b : v) std::printf("%02x", b); }

int main(int argc, char** argv) {
  if (argc < 4) return 2;
  zkemail::Email em;
  em.raw_email = slurp(argv[1]);
  em.public_key = {slurp(argv[2]), "rsa"};
  em.from_domain = argv[3];
  try {
    if (argc >= 7) {
      zkemail::EmailWithRegex in{em, {}};
      zkemail::CompiledRegex cr{{slurp(argv[4]), slurp(argv[5])}, std::vector<std::string>{argv[6]}};
      in.regex_info.header_parts = std::vector<zkemail::CompiledRegex>{cr};
      auto out = zkemail::verify_email_with_regex(in);
      std::printf("OK "); hex(out.email.from_domain_hash); std::printf(" "); hex(out.email.public_key_hash);
      for (auto& m : out.regex_matches) std::printf(" [%s]", m.c_str());
      std::printf("\nABI ");
      hex(zkemail::VerificationOutput::from_parts(out.email, out.regex_matches).abi_encode());
      std::printf("\n");
      // the same input twice and once with a capture the match does not contain, as a vector (zke_verify_emails_with_regex)
      std::vector<zkemail::EmailWithRegex> many{in, in, in};
      (*many[1].regex_info.header_parts)[0].captures = std::vector<std::string>{"not-in-the-match"};
      zkemail::Engine eng;
      std::printf("BATCH");
      for (const auto& r : eng.verify_emails_with_regex(many)) std::printf(" %u/%u", r.status, r.detail);
      std::printf("\n");
    } else {
      auto out = zkemail::verify_email(em);
      std::printf("OK "); hex(out.from_domain_hash); std::printf(" "); hex(out.public_key_hash); std::printf("\nABI ");
      hex(zkemail::VerificationOutput::from_parts(out, std::nullopt).abi_encode());
      std::printf("\n");
      // the same e-mail three times and once with its last body byte changed, as a vector of Email values (zke_verify_emails)
      std::vector<zkemail::Email> many{em, em, em, em};
      many[2].raw_email.back() ^= 1;
      zkemail::Engine eng;
      const auto recs = eng.verify_emails(many);
      std::printf("BATCH");
      for (const auto& r : recs) std::printf(" %u/%u", r.status, r.detail);
      std::printf(" %d\n", (int)(std::memcmp(recs[0].public_key_hash, out.public_key_hash.data(), 32) == 0));
    }
  } catch (const zkemail::VerifyPanic& p) {
    std::printf("PANIC %u %u\n", p.status, p.detail);
    return 1;
  }
  return 0;
}
