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
#include <map>
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
    if (int r = zke_engine_create(&o, &e_)) throw EngineError("zke_engine_create failed: " + std::to_string(r));
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

 private:
  uint32_t dfa_id(const DFA& d) {
    auto key = std::make_pair(d.fwd, d.bwd);
    auto it = ids_.find(key);
    if (it != ids_.end()) return it->second;
    uint32_t id = 0;
    if (int r = zke_dfa_register(e_, d.fwd.data(), d.fwd.size(), d.bwd.data(), d.bwd.size(), &id))
      throw EngineError(std::string("zke_dfa_register: ") + zke_last_error(e_));
    ids_.emplace(std::move(key), id);
    return id;
  }
  static uint8_t key_type_code(const std::string& t) {
    return t == "rsa" ? ZKE_KEY_RSA : (t == "ed25519" ? ZKE_KEY_ED25519 : ZKE_KEY_OTHER);
  }
  zke_result run(const Email& em, const RegexInfo* ri) {
    const uint64_t ro[2] = {0, em.raw_email.size()}, dofs[2] = {0, em.from_domain.size()}, ko[2] = {0, em.public_key.key.size()};
    const uint8_t kt = key_type_code(em.public_key.key_type);
    uint8_t ext = 0;
    for (const auto& x : em.external_inputs) if (!x.value) ext = 1;                      // circuits.rs:24
    static const uint8_t dummy = 0;
    zke_batch b{};
    b.n = 1;
    b.raw_blob = em.raw_email.empty() ? &dummy : em.raw_email.data(); b.raw_off = ro;
    b.domain_blob = em.from_domain.empty() ? &dummy : reinterpret_cast<const uint8_t*>(em.from_domain.data()); b.domain_off = dofs;
    b.key_blob = em.public_key.key.empty() ? &dummy : em.public_key.key.data(); b.key_off = ko;
    b.key_type = &kt; b.ext_null = &ext;
    std::vector<uint32_t> hids, bids, cap_off{0}, str_off{0};
    std::vector<uint8_t> blob;
    if (ri) {
      b.with_regex = 1;
      for (const auto* parts : {&ri->header_parts, &ri->body_parts}) {
        if (!*parts) continue;
        for (const auto& p : **parts) {
          (parts == &ri->header_parts ? hids : bids).push_back(dfa_id(p.verify_re));
          if (p.captures)
            for (const auto& s : *p.captures) { blob.insert(blob.end(), s.begin(), s.end()); str_off.push_back((uint32_t)blob.size()); }
          cap_off.push_back((uint32_t)str_off.size() - 1);
        }
      }
      if (blob.empty()) blob.push_back(0);
      b.n_header_parts = (uint32_t)hids.size(); b.n_body_parts = (uint32_t)bids.size();
      b.header_part_ids = hids.data(); b.body_part_ids = bids.data();
      b.cap_off = cap_off.data(); b.cap_str_off = str_off.data(); b.cap_blob = blob.data();
    }
    zke_result r{};
    if (int rc = zke_verify_batch(e_, &b, &r, nullptr)) throw EngineError(std::string("zke_verify_batch: ") + zke_last_error(e_) + " (" + std::to_string(rc) + ")");
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
  std::map<std::pair<std::vector<uint8_t>, std::vector<uint8_t>>, uint32_t> ids_;
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
