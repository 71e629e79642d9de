#pragma once
extern "C" {
int zke_dfa_register(zke_engine* e, const uint8_t*, size_t, const uint8_t*, size_t, uint32_t*) { return fail(e, ZKE_E_DEVICE, "not built yet"); }
int zke_verify_batch(zke_engine* e, const zke_batch*, zke_result*, zke_debug_out*) { return fail(e, ZKE_E_DEVICE, "not built yet"); }
int zke_verify_batch_device(zke_engine* e, const zke_batch*, uint64_t, uint64_t, uint64_t, zke_result*, void*) { return fail(e, ZKE_E_DEVICE, "not built yet"); }
int zke_verify_email(zke_engine* e, const uint8_t*, size_t, const char*, size_t, const uint8_t*, size_t, uint32_t, zke_result*) { return fail(e, ZKE_E_DEVICE, "not built yet"); }
}
