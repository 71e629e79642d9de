#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "sha256.hip.h"
namespace zke {
// CSR (blob, off[n+1]) -> ShaJob list with digests packed 32 B apart
__global__ void sha_jobs_from_csr_kernel(const uint8_t* blob, const uint64_t* off, uint32_t n, uint8_t* digests, ShaJob* jobs) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  ShaJob j;
  j.src = (uint64_t)(blob + off[i]);
  j.dst = (uint64_t)(digests + (size_t)i * 32);
  j.len = (uint32_t)(off[i + 1] - off[i]);
  j.pad = 0;
  jobs[i] = j;
}
}
