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

// Lanes 2k and 2k+1 of a 32x32 MFMA C/D tile hold neighbouring channels of the same pixels.  The pair trades one value
// each (DPP quad_perm [1,0,3,2]) so that every lane stores ONE dword -- two channels of one pixel -- instead of two 2-byte
// values: half the store instructions for fp16 activations.  v0 / v1: this lane's channel at two pixels; afterwards the
// even lane owns the first pixel and the odd lane the second.  Returns {channel 2k, channel 2k+1} as packed halves.
__device__ __forceinline__ unsigned si_pair_halves(float v0, float v1, bool odd) {
    typedef _Float16 si_h2 __attribute__((ext_vector_type(2)));
    const float send = odd ? v0 : v1;
    const float recv = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, send), 0xB1, 0xF, 0xF, true));
    si_h2 p;
    p[0] = si_store_cast<_Float16>(odd ? recv : v0);
    p[1] = si_store_cast<_Float16>(odd ? v1 : recv);
    return __builtin_bit_cast(unsigned, p);
}

#endif  // SI_HIP_INTERNAL_H_
