// ccr_gemm_common.h -- definitions shared by the GEMM + top-k kernels (ccr_fused.hip, experimental/gemm4w_proto.hip).
#pragma once
#include "ccr_common.h"

namespace ccr {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

enum { EPI_FILTER = 0, EPI_GMAX = 1, EPI_STORE = 2 };
// EPI_FILTER       : candidate record = one corpus row        {MFMA score, row}

// LDS-DMA of 16 bytes per lane: the LDS destination is the wave-uniform base + lane * 16
__device__ __forceinline__ void glds16(const void *gsrc, char *lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)gsrc,
                                     (__attribute__((address_space(3))) void *)lds_wave_base, 16, 0, 0);
}

// the same with a cache-policy immediate (aux: 1 = sc0, 2 = nt, 16 = sc1): diagnostic instantiations only
template <int AUX>
__device__ __forceinline__ void glds16_aux(const void *gsrc, char *lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)gsrc,
                                     (__attribute__((address_space(3))) void *)lds_wave_base, 16, 0, AUX);
}

// K is walked in 32-element sub-stages through a ring of four 32-KiB LDS buffers:
// [256 corpus rows x 64 B][256 query rows x 64 B] per sub-stage
constexpr int SUB_K = 32;
constexpr int SUB_BYTES = (TILE_DOCS + TILE_Q) * SUB_K * 2;  // 32768
constexpr int SUB_Q_REGION = TILE_DOCS * SUB_K * 2;           // 16384
constexpr int RING = 4;

// hipcc neither waits for LDS-DMA in __syncthreads() nor counts it for us: every wait is explicit
#define CCR_BARRIER() asm volatile("s_barrier" ::: "memory")
#define CCR_WAIT_VM(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
#define CCR_WAIT_LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")

}  // namespace ccr
