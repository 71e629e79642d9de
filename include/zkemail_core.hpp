// zkemail_core.hpp — C++ mirror of zkemail_core's public surface (core/src/lib.rs:1-13) over the
// C-ABI in zkemail_amd.h.  Header-only; link with libzkemail_amd.so.
//
// Same type names, field names and function names as the reference (core/src/structs.rs:8-75,
// core/src/circuits.rs:9,31).  Where the reference panics (assert!/unwrap/expect, an abort under
// its release profile, Cargo.toml:35) these functions throw zkemail::VerifyPanic carrying the
// status that names the panic site; a caller that wants drop-in abort semantics lets it propagate
// to std::terminate.
#pragma once
#include <cstdint>
#include <memory>
#include <optional>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "zkemail_amd.h"

namespace zkemail {

struct PublicKey {                         // structs.rs:8-11
  std::vector<uint8_t> key;
  std::string key_type;                    // "rsa" | "ed25519"
};
struct DFA {                               // structs.rs:16-19
  std::vector<uint8_t> fwd, bwd;
};
struct CompiledRegex {                     // structs.rs:24-27
  DFA verify_re;
  std::optional<std::vector<std::string>> captures;
};
struct RegexInfo {                         // structs.rs:32-35
  std::optional<std::vector<CompiledRegex>> header_parts, body_parts;
};
struct ExternalInput {                     // structs.rs:40-44
  std::string name;
  std::optional<std::string> value;
  size_t max_length = 0;
};
struct Email {                             // structs.rs:49-54
  std::string from_domain;
  std::vector<uint8_t> raw_email;
  PublicKey public_key;
  std::vector<ExternalInput> external_inputs;
};
struct EmailWithRegex {                    // structs.rs:59-62
  Email email;
  RegexInfo regex_info;
};
struct EmailVerifierOutput {               // structs.rs:65-69
  std::vector<uint8_t> from_domain_hash, public_key_hash;
  std::vector<std::string> external_inputs;
};
struct EmailWithRegexVerifierOutput {      // structs.rs:72-75
  EmailVerifierOutput email;
  std::vector<std::string> regex_matches;
};

struct EngineError : std::runtime_error { using std::runtime_error::runtime_error; };

// core/src/io.rs:18-44: the on-the-wire form of the witness (Solidity abi.encode of SolEmailOutput /
// SolEmailWithRegexOutput), through the C entry point zke_abi_encode.
struct VerificationOutput {
  EmailVerifierOutput email;
  std::optional<std::vector<std::string>> matches;      // nullopt: EmailOnly; a value: WithRegex
  static VerificationOutput from_parts(EmailVerifierOutput email, std::optional<std::vector<std::string>> matches) {   // io.rs:28-33
    return VerificationOutput{std::move(email), std::move(matches)};
  }
  std::vector<uint8_t> abi_encode() const {                                                                                // io.rs:35-44
    if (email.from_domain_hash.size() != 32 || email.public_key_hash.size() != 32)
      throw std::length_error("hashes must be 32 bytes");                                                                // io.rs:49-50 try_into().unwrap()
    auto table = [](const std::vector<std::string>& v, std::vector<const uint8_t*>& p, std::vector<size_t>& l) {
      for (const auto& s : v) { p.push_back(reinterpret_cast<const uint8_t*>(s.data())); l.push_back(s.size()); }
    };
    std::vector<const uint8_t*> p1, p2;
    std::vector<size_t> l1, l2;
    table(email.external_inputs, p1, l1);
    if (matches) table(*matches, p2, l2);
    size_t need = 0;
    auto call = [&](uint8_t* out, size_t cap) {
      return zke_abi_encode(email.from_domain_hash.data(), email.public_key_hash.data(), p1.data(), l1.data(), (uint32_t)p1.size(),
                            matches ? 1u : 0u, p2.data(), l2.data(), (uint32_t)p2.size(), out, cap, &need);
    };
    if (int rc = call(nullptr, 0)) throw EngineError("zke_abi_encode: " + std::to_string(rc));
    std::vector<uint8_t> out(need);
    if (int rc = call(out.data(), out.size())) throw EngineError("zke_abi_encode: " + std::to_string(rc));
    return out;
  }
};

// The reference would have panicked here.
struct VerifyPanic : std::runtime_error {
  uint32_t status, detail;
  VerifyPanic(uint32_t s, uint32_t d)
      : std::runtime_error("zkemail_core panic site status=" + std::to_string(s) + " detail=" + std::to_string(d)),
        status(s), detail(d) {}
};

class Engine {
 public:
  explicit Engine(int device = -1) {
    zke_options o{};
    o.device = device;
    if (int r = zke_engine_create(&o, &e_)) throw EngineError("zke_engine_create failed: " + std::to_string(r) + " " + zke_last_error(nullptr));
  }
  // every field of zke_options by name (ABI 0.3): slots, host threads, kernel variants, the strictness flags
  explicit Engine(const zke_options& o) {
    if (int r = zke_engine_create(&o, &e_)) throw EngineError("zke_engine_create failed: " + std::to_string(r) + " " + zke_last_error(nullptr));
  }
  ~Engine() { zke_engine_destroy(e_); }
  Engine(const Engine&) = delete;
  Engine& operator=(const Engine&) = delete;
  zke_engine* raw() { return e_; }

  // core/src/circuits.rs:9-29
  EmailVerifierOutput verify_email(const Email& email) {
    zke_result r = run(email, nullptr);
    if (r.status != ZKE_OK) throw VerifyPanic(r.status, r.detail);
    return output(email, r);
  }
  // core/src/circuits.rs:31-68
  EmailWithRegexVerifierOutput verify_email_with_regex(const EmailWithRegex& in) {
    zke_result r = run(in.email, &in.regex_info);
    if (r.status != ZKE_OK) throw VerifyPanic(r.status, r.detail);
    EmailWithRegexVerifierOutput out{output(in.email, r), {}};
    for (const auto* parts : {&in.regex_info.header_parts, &in.regex_info.body_parts})    // circuits.rs:58-62
      if (*parts)
        for (const auto& p : **parts)
          if (p.captures) out.regex_matches.insert(out.regex_matches.end(), p.captures->begin(), p.captures->end());
    return out;
  }

  // verify_email over a vector of e-mails, each where it is (zke_verify_emails: the engine gathers the buffers itself):
  // one record per e-mail, never a throw for a bad e-mail — `status` / `detail` say what the reference would have done.
  std::vector<zke_result> verify_emails(const std::vector<Email>& emails) {
    std::vector<zke_email_ref> refs(emails.size());
    for (size_t i = 0; i < emails.size(); i++) {
      const Email& em = emails[i];
      uint32_t ext = 0;
      for (const auto& x : em.external_inputs) if (!x.value) ext = 1;                    // circuits.rs:24
      refs[i] = zke_email_ref{em.raw_email.data(), em.raw_email.size(), em.from_domain.data(), em.from_domain.size(),
                              em.public_key.key.data(), em.public_key.key.size(), key_type_code(em.public_key.key_type), ext};
    }
    std::vector<zke_result> out(emails.size());
    if (int r = zke_verify_emails(e_, refs.data(), (uint32_t)refs.size(), out.data()))
      throw EngineError("zke_verify_emails failed: " + std::to_string(r) + " " + zke_last_error(e_));
    return out;
  }

  // verify_email_with_regex over a vector that shares one part list (one regex_config per batch; the captures are per e-mail):
  // the pairs are registered once (zke_dfa_register), the e-mails stay where they are (zke_verify_emails_with_regex).
  std::vector<zke_result> verify_emails_with_regex(const std::vector<EmailWithRegex>& in) {
    if (in.empty()) return {};
    std::vector<uint32_t> hids, bids;
    auto ids_of = [&](const std::optional<std::vector<CompiledRegex>>& parts) {
      std::vector<uint32_t> v;
      if (parts)
        for (const auto& p : *parts) {
          uint32_t id = 0;
          if (int r = zke_dfa_register(e_, p.verify_re.fwd.data(), p.verify_re.fwd.size(), p.verify_re.bwd.data(), p.verify_re.bwd.size(), &id))
            throw EngineError("zke_dfa_register failed: " + std::to_string(r) + " " + zke_last_error(e_));
          v.push_back(id);                       // an equal pair registered again gets the id it already has
        }
      return v;
    };
    hids = ids_of(in[0].regex_info.header_parts);
    bids = ids_of(in[0].regex_info.body_parts);
    std::vector<zke_email_ref> refs(in.size());
    std::vector<uint32_t> cap_off{0}, cap_str_off{0};
    std::vector<uint8_t> cap_blob;
    for (size_t i = 0; i < in.size(); i++) {
      const Email& em = in[i].email;
      if (ids_of(in[i].regex_info.header_parts) != hids || ids_of(in[i].regex_info.body_parts) != bids)
        throw EngineError("a batch must share one part list; split it per regex_config");
      uint32_t ext = 0;
      for (const auto& x : em.external_inputs) if (!x.value) ext = 1;
      refs[i] = zke_email_ref{em.raw_email.data(), em.raw_email.size(), em.from_domain.data(), em.from_domain.size(),
                              em.public_key.key.data(), em.public_key.key.size(), key_type_code(em.public_key.key_type), ext};
      for (const auto* parts : {&in[i].regex_info.header_parts, &in[i].regex_info.body_parts})
        if (*parts)
          for (const auto& p : **parts) {
            if (p.captures)
              for (const auto& s : *p.captures) { cap_blob.insert(cap_blob.end(), s.begin(), s.end()); cap_str_off.push_back((uint32_t)cap_blob.size()); }
            cap_off.push_back((uint32_t)cap_str_off.size() - 1);
          }
    }
    if (cap_blob.empty()) cap_blob.push_back(0);
    zke_regex_lists lists{(uint32_t)hids.size(), hids.data(), (uint32_t)bids.size(), bids.data(),
                          hids.size() + bids.size() ? cap_off.data() : nullptr, cap_str_off.data(), cap_blob.data()};
    std::vector<zke_result> out(in.size());
    if (int r = zke_verify_emails_with_regex(e_, refs.data(), (uint32_t)refs.size(), &lists, out.data()))
      throw EngineError("zke_verify_emails_with_regex failed: " + std::to_string(r) + " " + zke_last_error(e_));
    return out;
  }

 private:
  static uint32_t key_type_code(const std::string& t) {
    return t == "rsa" ? ZKE_KEY_RSA : (t == "ed25519" ? ZKE_KEY_ED25519 : ZKE_KEY_OTHER);
  }
  // One e-mail through the single-e-mail entry points of the C-ABI (zke_verify_email / zke_verify_email_with_regex).
  zke_result run(const Email& em, const RegexInfo* ri) {
    uint32_t ext = 0;
    for (const auto& x : em.external_inputs) if (!x.value) ext = 1;                      // circuits.rs:24
    const uint32_t kt = key_type_code(em.public_key.key_type);
    zke_result r{};
    int rc;
    if (!ri) {
      rc = zke_verify_email(e_, em.raw_email.data(), em.raw_email.size(), em.from_domain.data(), em.from_domain.size(),
                            em.public_key.key.data(), em.public_key.key.size(), kt, ext, &r);
    } else {
      // RegexInfo -> two zke_regex_part lists; the pointer tables live until the call returns
      std::vector<zke_regex_part> hp, bp;
      std::vector<std::vector<const uint8_t*>> ptrs;
      std::vector<std::vector<size_t>> lens;
      size_t total = 0;
      for (const auto* parts : {&ri->header_parts, &ri->body_parts}) if (*parts) total += (*parts)->size();
      ptrs.reserve(total); lens.reserve(total);
      for (const auto* parts : {&ri->header_parts, &ri->body_parts}) {
        if (!*parts) continue;
        for (const auto& p : **parts) {
          ptrs.emplace_back(); lens.emplace_back();
          if (p.captures)
            for (const auto& s : *p.captures) { ptrs.back().push_back(reinterpret_cast<const uint8_t*>(s.data())); lens.back().push_back(s.size()); }
          zke_regex_part q{};
          q.fwd = p.verify_re.fwd.data(); q.fwd_len = p.verify_re.fwd.size();
          q.bwd = p.verify_re.bwd.data(); q.bwd_len = p.verify_re.bwd.size();
          q.n_captures = (uint32_t)ptrs.back().size();
          q.captures = ptrs.back().data(); q.capture_lens = lens.back().data();
          (parts == &ri->header_parts ? hp : bp).push_back(q);
        }
      }
      rc = zke_verify_email_with_regex(e_, em.raw_email.data(), em.raw_email.size(), em.from_domain.data(), em.from_domain.size(),
                                       em.public_key.key.data(), em.public_key.key.size(), kt, ext, hp.data(), (uint32_t)hp.size(),
                                       bp.data(), (uint32_t)bp.size(), &r);
    }
    if (rc) throw EngineError(std::string("zke_verify_email: ") + zke_last_error(e_) + " (" + std::to_string(rc) + ")");
    return r;
  }
  static EmailVerifierOutput output(const Email& em, const zke_result& r) {
    EmailVerifierOutput o;
    o.from_domain_hash.assign(r.from_domain_hash, r.from_domain_hash + 32);            // circuits.rs:16
    o.public_key_hash.assign(r.public_key_hash, r.public_key_hash + 32);               // circuits.rs:17
    for (const auto& x : em.external_inputs) { o.external_inputs.push_back(x.name); o.external_inputs.push_back(*x.value); }
    return o;
  }
  zke_engine* e_ = nullptr;
};

inline Engine& default_engine() {
  static Engine e;
  return e;
}
inline EmailVerifierOutput verify_email(const Email& email) { return default_engine().verify_email(email); }
inline EmailWithRegexVerifierOutput verify_email_with_regex(const EmailWithRegex& in) {
  return default_engine().verify_email_with_regex(in);
}

}  // namespace zkemail
