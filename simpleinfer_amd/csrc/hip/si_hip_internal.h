// si_hip_internal.h -- shared helpers for the HIP translation units (not part of the C-ABI).
#ifndef SI_HIP_INTERNAL_H_
#define SI_HIP_INTERNAL_H_

#include <hip/hip_runtime.h>

#define SI_HIP_TRY(expr)                      \
    do {                                      \
        hipError_t _e = (expr);               \
        if (_e != hipSuccess) return (int)_e; \
    } while (0)

// memory-bound kernels: cap the grid at ~8 workgroups per CU and grid-stride the rest
static inline unsigned si_grid_for(size_t work_items, unsigned block = 256) {
    size_t blocks = (work_items + block - 1) / block;
    const size_t cap = 256u * 8u;
    if (blocks > cap) blocks = cap;
    if (blocks == 0) blocks = 1;
    return (unsigned)blocks;
}

// fp32 -> storage type. For _Float16 the empty asm keeps the value opaque so the compiler cannot fold the multiply / add
// that produced it into v_fma_mix{lo,hi}_f16 for some unrolled elements and not others: the fused form rounds once, the
// separate form twice, and an image's result would then depend on which accumulator slot (= batch position) it used.
template <typename T>
__device__ __forceinline__ T si_store_cast(float v) {
    if constexpr (sizeof(T) == 2) asm("" : "+v"(v));
    return (T)v;
}

#endif  // SI_HIP_INTERNAL_H_
