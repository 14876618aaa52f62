// si_hip_internal.h -- shared helpers for the HIP translation units (not part of the C-ABI).
#ifndef SI_HIP_INTERNAL_H_
#define SI_HIP_INTERNAL_H_

#include <hip/hip_runtime.h>

#include <map>
#include <mutex>
#include <utility>

// Environment switches (ablations, sweeps) exist in the EXPERIMENT build of the library only (-DSI_EXPERIMENT: tools/hip_variant.sh,
// simpleinfer_amd/build.py build_hip(defines=...)); in the product build the expression is its default and nothing reads the environment
// (round 6).  Choices a caller may legitimately make per call travel in SiConv2dDesc::plan (include/si_hip.h).
#ifdef SI_EXPERIMENT
#include <cstdlib>
#define SI_ENV_INT(name, dflt) ([] { const char* e_ = getenv(name); return e_ ? atoi(e_) : (dflt); }())
#else
#define SI_ENV_INT(name, dflt) (dflt)
#endif

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

// Workgroups of `kern` that are resident per CU at this block size and dynamic-LDS size: the grid of a persistent kernel
// (a workgroup that has to wait for a slot starts its share of the work when the others are finishing theirs).  Cached
// per (kernel, LDS size); engines may launch from several host threads, hence the lock.
template <typename Kern>
static inline int si_resident_blocks(Kern kern, int threads, size_t lds) {
    static std::mutex mu;
    static std::map<std::pair<const void*, size_t>, int> cache;
    const std::pair<const void*, size_t> key(reinterpret_cast<const void*>(kern), lds);
    std::lock_guard<std::mutex> lock(mu);
    auto it = cache.find(key);
    if (it != cache.end()) return it->second;
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kern, threads, lds) != hipSuccess || nb < 1) nb = 1;
    cache[key] = nb;
    return nb;
}

// Dynamic LDS above the 64 KB default needs the per-function opt-in once (not again during a stream capture).
template <typename Kern>
static inline hipError_t si_allow_dynamic_lds(Kern kern, size_t lds) {
    if (lds <= 64 * 1024) return hipSuccess;
    static std::mutex mu;
    static std::map<const void*, size_t> allowed;
    std::lock_guard<std::mutex> lock(mu);
    size_t& have = allowed[reinterpret_cast<const void*>(kern)];
    if (have >= lds) return hipSuccess;
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e == hipSuccess) have = lds;
    return e;
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

// UnaryOp codes of expand_expression.cpp:146-165 (ncnn's numbering): 0 abs 1 neg 2 floor 3 ceil 4 square 5 sqrt 6 rsqrt 7 exp
// 8 log 9 sin 10 cos 11 tan 12 asin 13 acos 14 atan 15 reciprocal 16 tanh 17 log10.  Library-accurate functions, IEEE sqrt and
// division: this is a standalone arithmetic operator, not a fused epilogue.
__device__ __forceinline__ float si_unary_apply(int op, float x) {
    switch (op) {
        case 0: return fabsf(x);
        case 1: return -x;
        case 2: return floorf(x);
        case 3: return ceilf(x);
        case 4: return x * x;
        case 5: return sqrtf(x);
        case 6: return 1.0f / sqrtf(x);
        case 7: return expf(x);
        case 8: return logf(x);
        case 9: return sinf(x);
        case 10: return cosf(x);
        case 11: return tanf(x);
        case 12: return asinf(x);
        case 13: return acosf(x);
        case 14: return atanf(x);
        case 15: return 1.0f / x;
        case 16: return tanhf(x);
        case 17: return log10f(x);
        default: return x;
    }
}

// YOLOv5 Detect decode of one workgroup's conv tile, straight-line form for a tile that lies inside ONE image and below M
// (reference src/layer/yolo_detect.cpp:223-266: sigmoid, xy = (2s + grid) * stride, wh = (2s)^2 * anchor, rows
// [img][row_off + pix*na + anchor][ne]).  Per lane (= output channel) the element kind, the grid / anchor pointer and
// the output pointer are fixed before the 16-element loop; per element: sigmoid and one store, for the waves that hold box
// columns also one masked 4-byte load and a bit-select.  Args: the conv kernels' argument structs (ocg, bias, yna, yne, ohow, ygrid,
// yanchor, ystride, yrows_total, yrow_off).  C/D map of the 32x32 MFMA tile as everywhere: col = lane&31 (channel),
// row = (e&3) + 8*(e>>2) + 4*(lane>>5) (pixel).
// MT: the MFMA tile the accumulators come from -- 32 (16 registers, rows (e&3) + 8*(e>>2)) or 16 (v_mfma_f32_16x16x4_f32: 4
// registers, rows e; col = lane&15, + 4*(lane>>4) already in mrow0)
typedef float si_f32x16 __attribute__((ext_vector_type(16)));
template <int TM, int TN, typename Args, int MT = 32, typename AccT = si_f32x16>
__device__ __forceinline__ void si_yolo_tile_one_image(const Args& a, float* out, AccT (&acc)[TM][TN], int mrow0, int ocol0, int img) {
#pragma clang fp contract(off)
    const int per_pix = a.yna * a.yne;
#pragma unroll
    for (int u = 0; u < TN; ++u) {
        const int o = ocol0 + u * MT;
        const bool live = o < a.ocg;
        const int oo = live ? o : 0;
        const float bv = a.bias ? a.bias[oo] : 0.0f;
        const int anc = oo / a.yne;
        const int e_ = oo - anc * a.yne;
        const bool is_xy = e_ < 2, is_box = e_ < 4;
        const float* const auxp = (is_xy ? a.ygrid + e_ : a.yanchor + (is_box ? e_ - 2 : 0)) + anc * 2;
        // a wave whose columns hold no box entry (x, y, w, h of an anchor: 12 of 255 columns, in three of eight 32-column blocks)
        // never sees the box arithmetic (wave-uniform branch)
        const bool any_box = __builtin_amdgcn_ballot_w64(is_box && live) != 0ull;
        const unsigned mxy = is_xy ? ~0u : 0u, mwh = (is_box && !is_xy) ? ~0u : 0u, msg = is_box ? 0u : ~0u;
#pragma unroll
        for (int t = 0; t < TM; ++t) {
            const int pix0 = mrow0 + t * MT - img * a.ohow;
            float* const op = out + ((size_t)img * a.yrows_total + a.yrow_off) * a.yne + (size_t)pix0 * per_pix + oo;
            constexpr int NE = MT == 32 ? 16 : 4;
            // the sigmoid for every element, nothing else (round 5: the select chain below used to sit in this loop, and hipcc turned
            // it into two exec-masked branches per element plus one around every store: ~25 instructions per element)
            float v[NE];
#pragma unroll
            for (int e = 0; e < NE; ++e) v[e] = __builtin_amdgcn_rcpf(1.0f + __expf(-(acc[t][u][e] + bv)));
            if (any_box) {
                const float* const ap = auxp + (size_t)pix0 * a.yna * 2;
                // the box lanes' grid / anchor values, ALL requested before the first one is used: with the load inside the element
                // loop hipcc waits for each one in turn (16 L2 round trips per 32x32 block on the waves that own a box column)
                float auxv[NE];
#pragma unroll
                for (int e = 0; e < NE; ++e) auxv[e] = 0.0f;
                if (is_box) {
#pragma unroll
                    for (int e = 0; e < NE; ++e) auxv[e] = ap[(MT == 32 ? (e & 3) + 8 * (e >> 2) : e) * a.yna * 2];
                }
#pragma unroll
                for (int e = 0; e < NE; ++e) {
                    const float t2 = v[e] * 2.0f;
                    const float xy = (t2 + auxv[e]) * a.ystride;
                    const float wh = t2 * t2 * auxv[e];
                    // (the same three values as `is_xy ? xy : (is_box ? wh : sg)`, selected by bit masks: no branches)
                    v[e] = __builtin_bit_cast(float, (__builtin_bit_cast(unsigned, xy) & mxy) | (__builtin_bit_cast(unsigned, wh) & mwh) |
                                                         (__builtin_bit_cast(unsigned, v[e]) & msg));
                }
            }
            if (live) {
#pragma unroll
                for (int e = 0; e < NE; ++e) op[(MT == 32 ? (e & 3) + 8 * (e >> 2) : e) * per_pix] = v[e];
            }
        }
    }
}

// ---- diagnostic build only (-DSI_DIAG_STAMPS; tools/conv_diag.py): per-workgroup s_memtime / s_memrealtime stamps at the phase
// boundaries of a kernel, written to a __device__ array of their own that nothing else reads.  In the product build no stamp
// executes and every macro expands to nothing.  Slots: [0] realtime at start, [1..5] cycle stamps, [6] realtime at end, [7] hw id.
#ifdef SI_DIAG_STAMPS
#define SI_STAMP_ARRAY(name) __device__ unsigned long long name[65536 * 8]
#define SI_STAMP_DECL unsigned long long st_[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define SI_STAMP(i) st_[i] = __builtin_amdgcn_s_memtime()
#define SI_STAMP_RT(i) st_[i] = __builtin_amdgcn_s_memrealtime()
#define SI_STAMP_FLUSH(name)                                                                                    \
    if (threadIdx.x == 0 && blockIdx.y == 0 && blockIdx.x < 65536) {                                            \
        unsigned hw_, xcc_;                                                                                     \
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_));                                       \
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_));                                     \
        st_[7] = ((unsigned long long)xcc_ << 32) | hw_;                                                        \
        for (int i_ = 0; i_ < 8; ++i_) name[(size_t)blockIdx.x * 8 + i_] = st_[i_];                             \
    }
#define SI_STAMP_ACCESSORS(name, read_fn, clear_fn)                                                                              \
    extern "C" int read_fn(unsigned long long* host, size_t count) {                                                             \
        return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(name), count * sizeof(unsigned long long), 0, hipMemcpyDeviceToHost);   \
    }                                                                                                                            \
    extern "C" int clear_fn(void) {                                                                                              \
        void* p_ = nullptr;                                                                                                      \
        hipError_t e_ = hipGetSymbolAddress(&p_, HIP_SYMBOL(name));                                                              \
        if (e_ != hipSuccess) return (int)e_;                                                                                    \
        return (int)hipMemset(p_, 0, sizeof(unsigned long long) * 65536 * 8);                                                    \
    }
#else
#define SI_STAMP_ARRAY(name) static_assert(true, "")
#define SI_STAMP_DECL
#define SI_STAMP(i)
#define SI_STAMP_RT(i)
#define SI_STAMP_FLUSH(name)
#endif

#endif  // SI_HIP_INTERNAL_H_
