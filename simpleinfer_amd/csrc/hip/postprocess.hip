// postprocess.hip -- the steps either side of Engine::Forward() in the reference's only real application
// (test/test_yolo/test_yolo.cpp): letterbox packing of a resized image (:234-259) and the detection post-processing
// (:337-428: confidence filter, descending sort, per-class greedy NMS, un-letterbox + clip), all on the device so that
// only a few KB of boxes per image have to leave HBM instead of the 8.57 MB/img prediction slab.
//
// Everything here is compare / index work plus a handful of fp32 operations whose results must equal the reference's
// bit for bit (the picks depend on them), so floating-point contraction is switched off for this file.
#include "si_hip.h"
#include "si_hip_internal.h"

#pragma clang fp contract(off)

namespace {

// ---- letterbox --------------------------------------------------------------------------------
// dst[y][x][c] = inside ? src[y-pt][x-pl][2-c] / 255 : 114 / 255        (bgr -> rgb reverse, pad(114), cast, /255)
__global__ void letterbox_kernel(const unsigned char* __restrict__ src, int hr, int wr, float* __restrict__ dst, int H,
                                 int W, int pt, int pl) {
    const int total = H * W * 3;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int c = i % 3;
        const int x = (i / 3) % W;
        const int y = i / (3 * W);
        const int sy = y - pt, sx = x - pl;
        float v = 114.0f;
        if (sy >= 0 && sy < hr && sx >= 0 && sx < wr) v = (float)src[((size_t)sy * wr + sx) * 3 + (2 - c)];
        dst[i] = v / 255.0f;
    }
}

// the same for all images of a batch that share one geometry (frames of one camera): blockIdx.y = image, four consecutive
// output floats per thread (one 16-byte store; H * W * 3 a multiple of 4)
__global__ void letterbox_batch_kernel(const unsigned char* __restrict__ src, size_t src_stride, int hr, int wr, float* __restrict__ dst,
                                       int H, int W, int pt, int pl) {
    const int quads = H * W * 3 / 4;
    const unsigned char* s = src + (size_t)blockIdx.y * src_stride;
    float4* d = reinterpret_cast<float4*>(dst + (size_t)blockIdx.y * H * W * 3);
    for (int q = blockIdx.x * blockDim.x + threadIdx.x; q < quads; q += gridDim.x * blockDim.x) {
        float v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int i = q * 4 + k;
            const int c = i % 3;
            const int x = (i / 3) % W;
            const int y = i / (3 * W);
            const int sy = y - pt, sx = x - pl;
            float t = 114.0f;
            if (sy >= 0 && sy < hr && sx >= 0 && sx < wr) t = (float)s[((size_t)sy * wr + sx) * 3 + (2 - c)];
            v[k] = t / 255.0f;
        }
        d[q] = make_float4(v[0], v[1], v[2], v[3]);
    }
}

// ---- bilinear resize (the cv::resize of PreProcess, test_yolo.cpp:213-216) ------------------------------------------------
// The reference calls cv::resize of its simpleocv submodule, whose source is ABSENT from this checkout (empty 3rdparty
// directory), so this is pinned to the PUBLISHED algorithm that library and OpenCV share for 8-bit INTER_LINEAR -- not to the
// reference: half-pixel centres, sx = floor((dx + 0.5) * (src / dst) - 0.5) clamped to the image (left edge: fx = 0; right edge:
// sx = src - 2, fx = 1), 11-bit fixed-point coefficients a = round(f * 2048) (half away from zero), the horizontal pass kept
// as (S[sx] * a0 + S[sx + 1] * a1) >> 4 in 16 bits, the vertical pass ((b0 * r0) >> 16) + ((b1 * r1) >> 16), + 2, >> 2.
// The test suite holds the device to a numpy restatement of exactly this arithmetic (tests/test_gpu_ops.py).
struct ResizeAxis {
    int s0;        // first source index
    short a0, a1;  // 11-bit weights of s0 and s0 + 1
};
__device__ __forceinline__ ResizeAxis resize_axis(int d, double scale, int src) {
    float f = (float)(((double)d + 0.5) * scale - 0.5);
    int s = (int)floorf(f);
    f -= (float)s;
    if (s < 0) { s = 0; f = 0.0f; }
    if (s >= src - 1) { s = src - 2; f = 1.0f; }
    if (src == 1) { s = 0; f = 0.0f; }
    ResizeAxis r;
    r.s0 = s;
    r.a0 = (short)(int)((1.0f - f) * 2048.0f + 0.5f);
    r.a1 = (short)(int)(f * 2048.0f + 0.5f);
    return r;
}
__device__ __forceinline__ int resize_pixel(const unsigned char* __restrict__ src, int sw, int c, const ResizeAxis& ax, const ResizeAxis& ay, int sh) {
    const int x1 = sw > 1 ? ax.s0 + 1 : ax.s0, y1 = sh > 1 ? ay.s0 + 1 : ay.s0;
    const unsigned char* r0 = src + ((size_t)ay.s0 * sw) * 3 + c;
    const unsigned char* r1 = src + ((size_t)y1 * sw) * 3 + c;
    const int h0 = ((int)r0[ax.s0 * 3] * ax.a0 + (int)r0[x1 * 3] * ax.a1) >> 4;
    const int h1 = ((int)r1[ax.s0 * 3] * ax.a0 + (int)r1[x1 * 3] * ax.a1) >> 4;
    return ((((int)ay.a0 * (int)(short)h0) >> 16) + (((int)ay.a1 * (int)(short)h1) >> 16) + 2) >> 2;
}

// plain resize, u8 BGR (or any 3-channel) image -> u8, blockIdx.y = image
__global__ void resize_bilinear_u8c3_kernel(const unsigned char* __restrict__ src, size_t src_stride, int sh, int sw,
                                            unsigned char* __restrict__ dst, size_t dst_stride, int dh, int dw) {
    const double scx = (double)sw / (double)dw, scy = (double)sh / (double)dh;
    const unsigned char* s = src + (size_t)blockIdx.y * src_stride;
    unsigned char* d = dst + (size_t)blockIdx.y * dst_stride;
    const int total = dh * dw;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int y = i / dw, x = i - y * dw;
        const ResizeAxis ax = resize_axis(x, scx, sw), ay = resize_axis(y, scy, sh);
#pragma unroll
        for (int c = 0; c < 3; ++c) d[(size_t)i * 3 + c] = (unsigned char)resize_pixel(s, sw, c, ax, ay, sh);
    }
}

// resize + letterbox in one pass (PreProcess whole, test_yolo.cpp:194-259): camera frame [sh][sw][3] BGR u8 -> the aspect-
// preserving [hr][wr] bilinear resize (never written) -> RGB, pad(114), float, / 255 into the [H][W][3] slot of the input tensor
__global__ void resize_letterbox_batch_kernel(const unsigned char* __restrict__ src, size_t src_stride, int sh, int sw, int hr, int wr,
                                              float* __restrict__ dst, int H, int W, int pt, int pl) {
    const double scx = (double)sw / (double)wr, scy = (double)sh / (double)hr;
    const unsigned char* s = src + (size_t)blockIdx.y * src_stride;
    float* d = dst + (size_t)blockIdx.y * H * W * 3;
    const int total = H * W;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int y = i / W, x = i - y * W;
        const int ry = y - pt, rx = x - pl;
        float v[3] = {114.0f, 114.0f, 114.0f};
        if (ry >= 0 && ry < hr && rx >= 0 && rx < wr) {
            const ResizeAxis ax = resize_axis(rx, scx, sw), ay = resize_axis(ry, scy, sh);
#pragma unroll
            for (int c = 0; c < 3; ++c) v[c] = (float)resize_pixel(s, sw, 2 - c, ax, ay, sh);   // bgr -> rgb
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) d[(size_t)i * 3 + c] = v[c] / 255.0f;
    }
}

// ---- post-processing workspace ----------------------------------------------------------------
// per image, cap = rows candidates, nbins = classes + 1 (bin = label + 1; label -1 = "no class"):
//   count[n] | bin_count[n][nbins] | keep[n][cap]            (zeroed at the start of every call, one memset)
//   key[n][cap] u64 | box, label [n][cap]                    candidates in arrival order
//   sbox, slabel, sprob [n][cap]                             sorted by confidence (the reference's order)
//   gbox[n][cap] float4, grank[n][cap]                       sorted by (label, confidence): one segment per label
//   pbox, plabel, parea [n][cap]                             boxes picked so far (network coordinates)
struct PostWs {
    int* count;
    int* bin_count;
    int* keep;
    int nbins;
    size_t zero_bytes;
    float4* gbox;
    int* grank;
    unsigned long long* key;
    float4* box;
    int* label;
    float4* sbox;
    int* slabel;
    float* sprob;
    float4* pbox;
    int* plabel;
    float* parea;
};

__host__ __device__ inline size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

inline size_t post_ws_layout(int n, int cap, int nbins, char* base, PostWs* ws) {
    size_t off = 0;
    auto take = [&](size_t bytes) {
        char* p = base ? base + off : nullptr;
        off += align256(bytes);
        return p;
    };
    const size_t nc = (size_t)n * cap;
    char* p;
    p = take(sizeof(int) * n);                     if (ws) ws->count = (int*)p;
    p = take(sizeof(int) * (size_t)n * nbins);     if (ws) ws->bin_count = (int*)p;
    p = take(sizeof(int) * nc);                    if (ws) ws->keep = (int*)p;
    if (ws) { ws->zero_bytes = off; ws->nbins = nbins; }
    p = take(sizeof(float4) * nc);                 if (ws) ws->gbox = (float4*)p;
    p = take(sizeof(int) * nc);                    if (ws) ws->grank = (int*)p;
    p = take(sizeof(unsigned long long) * nc);     if (ws) ws->key = (unsigned long long*)p;
    p = take(sizeof(float4) * nc);                 if (ws) ws->box = (float4*)p;
    p = take(sizeof(int) * nc);                    if (ws) ws->label = (int*)p;
    p = take(sizeof(float4) * nc);                 if (ws) ws->sbox = (float4*)p;
    p = take(sizeof(int) * nc);                    if (ws) ws->slabel = (int*)p;
    p = take(sizeof(float) * nc);                  if (ws) ws->sprob = (float*)p;
    p = take(sizeof(float4) * nc);                 if (ws) ws->pbox = (float4*)p;
    p = take(sizeof(int) * nc);                    if (ws) ws->plabel = (int*)p;
    p = take(sizeof(float) * nc);                  if (ws) ws->parea = (float*)p;
    return off;
}

// order-preserving map float -> u32 (larger float <=> larger integer), so the sort key is one integer compare
__device__ __forceinline__ unsigned ordered_bits(float f) {
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// ---- 1. confidence filter (test_yolo.cpp:341-377) ----------------------------------------------
// RPB prediction rows are staged in LDS with coalesced loads; one thread then scans one row: box score, first-maximum
// class (strict '>' as :349-353), confidence = box * class, kept when >= prob_threshold.  Survivors are appended to the
// image's candidate list (order is irrelevant: the sort key carries the element index).
constexpr int RPB = 128;

__global__ __launch_bounds__(RPB) void yolo_filter_kernel(const float* __restrict__ pred, int rows, int ne,
                                                          float prob_threshold, PostWs ws) {
    extern __shared__ float stage[];
    const int b = blockIdx.y;
    const int r0 = blockIdx.x * RPB;
    const int nrows = min(RPB, rows - r0);
    const float* src = pred + ((size_t)b * rows + r0) * ne;
    const int total = nrows * ne;
    for (int i = threadIdx.x; i < total; i += RPB) stage[i] = src[i];
    __syncthreads();
    if ((int)threadIdx.x >= nrows) return;
    const float* r = stage + threadIdx.x * ne;
    const float box_score = r[4];
    int class_index = -1;
    float class_score = -3.402823466e+38f;  // -FLT_MAX
    for (int k = 0; k < ne - 5; ++k) {
        const float s = r[5 + k];
        if (s > class_score) {
            class_index = k;
            class_score = s;
        }
    }
    const float confidence = box_score * class_score;
    if (confidence >= prob_threshold) {
        const float cx = r[0], cy = r[1], w = r[2], h = r[3];
        const float x0 = cx - w * 0.5f, y0 = cy - h * 0.5f;
        const float x1 = cx + w * 0.5f, y1 = cy + h * 0.5f;
        const int e = r0 + threadIdx.x;
        const int slot = atomicAdd(&ws.count[b], 1);
        atomicAdd(&ws.bin_count[(size_t)b * ws.nbins + class_index + 1], 1);
        const size_t o = (size_t)b * rows + slot;
        ws.box[o] = make_float4(x0, y0, x1 - x0, y1 - y0);
        ws.label[o] = class_index;
        // descending (prob, then earlier element first): larger key sorts first
        ws.key[o] = ((unsigned long long)ordered_bits(confidence) << 32) | (unsigned)(0xffffffffu - (unsigned)e);
    }
}

// ---- 2. sort by rank counting -------------------------------------------------------------------
// rank(i) = #{j : key_j > key_i}; keys are unique, so ranks are a permutation.  O(count^2) compares spread over the
// whole chip with the keys tiled through LDS: exact, deterministic, and ~0.5 ms even if all 25200 rows of all 32 images
// survive; real images leave a few hundred candidates.  (The reference's order among EQUAL confidences is whatever its
// unstable quicksort produces, :28-66; here ties are broken by element index.)
constexpr int SORT_T = 256;
constexpr int SORT_TILE = 1024;

struct alignas(16) SortItem {
    unsigned long long key;
    int label;
    int pad;
};

// Two ranks per candidate from one pass over the keys: `rank` in the confidence order (the order the reference's
// picks come out in) and `seg` = position inside the candidate's own label segment of the (label, confidence) order,
// which is what per-class NMS walks.
__global__ __launch_bounds__(SORT_T) void yolo_rank_sort_kernel(int rows, PostWs ws) {
    __shared__ SortItem tile[SORT_TILE];
    const int b = blockIdx.y;
    const int count = ws.count[b];
    const int i0 = blockIdx.x * SORT_T;
    if (i0 >= count) return;
    const size_t base = (size_t)b * rows;
    const int i = i0 + threadIdx.x;
    const unsigned long long mine = i < count ? ws.key[base + i] : 0ull;
    const int mylabel = i < count ? ws.label[base + i] : -2;
    int rank = 0, seg = 0;
    for (int t0 = 0; t0 < count; t0 += SORT_TILE) {
        const int tn = min(SORT_TILE, count - t0);
        __syncthreads();
        for (int j = threadIdx.x; j < tn; j += SORT_T) {
            tile[j].key = ws.key[base + t0 + j];
            tile[j].label = ws.label[base + t0 + j];
        }
        __syncthreads();
#pragma unroll 4
        for (int j = 0; j < tn; ++j) {
            const SortItem it = tile[j];
            const int gt = it.key > mine;
            rank += gt;
            seg += gt & (it.label == mylabel);
        }
    }
    // start of my label's segment = candidates with a smaller label (exclusive prefix over the bins)
    int start = 0;
    {
        const int* bc = ws.bin_count + (size_t)b * ws.nbins;
        for (int k = 0; k < mylabel + 1; ++k) start += bc[k];
    }
    if (i < count) {
        const float4 bx = ws.box[base + i];
        const size_t o = base + rank;
        ws.sbox[o] = bx;
        ws.slabel[o] = mylabel;
        const unsigned ob = (unsigned)(mine >> 32);
        const unsigned u = (ob & 0x80000000u) ? (ob & 0x7fffffffu) : ~ob;
        ws.sprob[o] = __uint_as_float(u);
        ws.gbox[base + start + seg] = bx;
        ws.grank[base + start + seg] = rank;
    }
}

// ---- 3. greedy NMS + un-letterbox (test_yolo.cpp:68-104, 379-416) -------------------------------
// intersection of two rects as simpleocv's `a & b` gives it (3rdparty/simpleocv, from ncnn; an absent submodule):
// empty (area 0) when width <= 0 or height <= 0.
__device__ __forceinline__ float inter_area(const float4 a, const float4 b) {
    const float x1 = fmaxf(a.x, b.x), y1 = fmaxf(a.y, b.y);
    const float w = fminf(a.x + a.z, b.x + b.z) - x1;
    const float h = fminf(a.y + a.w, b.y + b.w) - y1;
    if (w <= 0.0f || h <= 0.0f) return 0.0f;
    return w * h;
}

__device__ __forceinline__ float clipf(float v, float lo, float hi) { return fmaxf(lo, fminf(v, hi)); }

// One workgroup (4 waves) per image walks the sorted candidates 64 at a time.  Phase 1: every wave tests the 64
// candidates against its quarter of the boxes picked so far.  Phase 2 (wave 0): the in-chunk dependency chain is
// resolved with ballots -- candidate i, if still alive, suppresses later same-label candidates it overlaps.
__global__ __launch_bounds__(256) void yolo_nms_kernel(int rows, float nms_threshold, int agnostic,
                                                       const float* __restrict__ adjust, float* __restrict__ dets,
                                                       int* __restrict__ counts, int max_det, PostWs ws) {
    __shared__ unsigned long long dead[4];
    __shared__ int picked_n;
    const int b = blockIdx.x;
    const int count = ws.count[b];
    const size_t base = (size_t)b * rows;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (threadIdx.x == 0) picked_n = 0;
    __syncthreads();

    float pad_l = 0.0f, pad_t = 0.0f, scale = 1.0f, xmax = 0.0f, ymax = 0.0f;
    const bool adj = adjust != nullptr;
    if (adj) {
        pad_l = adjust[b * 5 + 0];
        pad_t = adjust[b * 5 + 1];
        scale = adjust[b * 5 + 2];
        xmax = adjust[b * 5 + 3] - 1.0f;
        ymax = adjust[b * 5 + 4] - 1.0f;
    }

    for (int c0 = 0; c0 < count; c0 += 64) {
        const int idx = c0 + lane;
        const bool valid = idx < count;
        float4 box = make_float4(0.f, 0.f, 0.f, 0.f);
        int label = -2;
        if (valid) {
            box = ws.sbox[base + idx];
            label = ws.slabel[base + idx];
        }
        const float area = box.z * box.w;
        const int P = picked_n;
        bool supp = !valid;
        for (int p = wave; p < P; p += 4) {
            const int pl = ws.plabel[base + p];
            if (!agnostic && pl != label) continue;
            const float4 pb = ws.pbox[base + p];
            const float ia = inter_area(box, pb);
            const float ua = area + ws.parea[base + p] - ia;
            if (ia / ua > nms_threshold) supp = true;
        }
        const unsigned long long m = __ballot(supp);
        if (lane == 0) dead[wave] = m;
        __syncthreads();
        if (wave == 0) {
            unsigned long long alive = ~(dead[0] | dead[1] | dead[2] | dead[3]);
            for (int i = 0; i < 63; ++i) {
                if (!((alive >> i) & 1ull)) continue;  // wave-uniform
                float4 bi;
                bi.x = __shfl(box.x, i); bi.y = __shfl(box.y, i); bi.z = __shfl(box.z, i); bi.w = __shfl(box.w, i);
                const int li = __shfl(label, i);
                const float ai = __shfl(area, i);
                bool s = false;
                if (lane > i && (agnostic || li == label)) {
                    const float ia = inter_area(box, bi);
                    const float ua = area + ai - ia;
                    s = ia / ua > nms_threshold;
                }
                alive &= ~__ballot(s);
            }
            const bool keep = (alive >> lane) & 1ull;
            const int pos = P + __popcll(alive & ((1ull << lane) - 1ull));
            if (keep) {
                ws.pbox[base + pos] = box;
                ws.plabel[base + pos] = label;
                ws.parea[base + pos] = area;
                if (pos < max_det) {
                    float x0 = box.x, y0 = box.y, x1 = box.x + box.z, y1 = box.y + box.w;
                    if (adj) {
                        x0 = clipf((x0 - pad_l) / scale, 0.0f, xmax);
                        y0 = clipf((y0 - pad_t) / scale, 0.0f, ymax);
                        x1 = clipf((x1 - pad_l) / scale, 0.0f, xmax);
                        y1 = clipf((y1 - pad_t) / scale, 0.0f, ymax);
                    }
                    float* d = dets + ((size_t)b * max_det + pos) * 6;
                    d[0] = x0; d[1] = y0;
                    d[2] = adj ? x1 - x0 : box.z;
                    d[3] = adj ? y1 - y0 : box.w;
                    d[4] = ws.sprob[base + idx];
                    d[5] = (float)label;
                }
            }
            if (lane == 0) picked_n = P + __popcll(alive);
        }
        __threadfence_block();
        __syncthreads();
    }
    if (threadIdx.x == 0) counts[b] = picked_n;
}

// Per-class NMS (the default, agnostic = false): candidates of different labels never interact (:86-87), so every
// (image, label) segment is an independent problem and gets its own wave; no workgroup barriers.  A pick is recorded as
// keep[confidence rank] = 1 and the output order is restored by yolo_compact_kernel.
__global__ __launch_bounds__(256) void yolo_nms_segments_kernel(int rows, float nms_threshold, PostWs ws) {
    const int b = blockIdx.y;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int bin = blockIdx.x * 4 + wave;
    if (bin >= ws.nbins) return;
    const int* bc = ws.bin_count + (size_t)b * ws.nbins;
    const int len = bc[bin];
    if (len == 0) return;
    int start = 0;
    for (int k = lane; k < bin; k += 64) start += bc[k];
    for (int o = 32; o > 0; o >>= 1) start += __shfl_xor(start, o);
    const size_t base = (size_t)b * rows + start;
    int P = 0;
    for (int c0 = 0; c0 < len; c0 += 64) {
        const int idx = c0 + lane;
        const bool valid = idx < len;
        float4 box = make_float4(0.f, 0.f, 0.f, 0.f);
        int grank = 0;
        if (valid) {
            box = ws.gbox[base + idx];
            grank = ws.grank[base + idx];
        }
        const float area = box.z * box.w;
        bool supp = !valid;
        for (int p = 0; p < P; ++p) {
            const float4 pb = ws.pbox[base + p];
            const float ia = inter_area(box, pb);
            const float ua = area + ws.parea[base + p] - ia;
            if (ia / ua > nms_threshold) supp = true;
        }
        unsigned long long alive = ~__ballot(supp);
        for (int i = 0; i < 63; ++i) {
            if (!((alive >> i) & 1ull)) continue;  // wave-uniform
            float4 bi;
            bi.x = __shfl(box.x, i); bi.y = __shfl(box.y, i); bi.z = __shfl(box.z, i); bi.w = __shfl(box.w, i);
            const float ai = __shfl(area, i);
            bool s = false;
            if (lane > i) {
                const float ia = inter_area(box, bi);
                const float ua = area + ai - ia;
                s = ia / ua > nms_threshold;
            }
            alive &= ~__ballot(s);
        }
        if ((alive >> lane) & 1ull) {
            const int pos = P + __popcll(alive & ((1ull << lane) - 1ull));
            ws.pbox[base + pos] = box;
            ws.parea[base + pos] = area;
            ws.keep[(size_t)b * rows + grank] = 1;
        }
        P += __popcll(alive);
        __threadfence_block();  // this wave's picks are read back by all of its lanes in the next chunk
    }
}

// keep flags (indexed by confidence rank) -> dense output in confidence order + un-letterbox / clip (:386-416)
__global__ __launch_bounds__(256) void yolo_compact_kernel(int rows, const float* __restrict__ adjust,
                                                           float* __restrict__ dets, int* __restrict__ counts,
                                                           int max_det, PostWs ws) {
    __shared__ int wsum[4];
    const int b = blockIdx.x;
    const int count = ws.count[b];
    const size_t base = (size_t)b * rows;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float pad_l = 0.0f, pad_t = 0.0f, scale = 1.0f, xmax = 0.0f, ymax = 0.0f;
    const bool adj = adjust != nullptr;
    if (adj) {
        pad_l = adjust[b * 5 + 0];
        pad_t = adjust[b * 5 + 1];
        scale = adjust[b * 5 + 2];
        xmax = adjust[b * 5 + 3] - 1.0f;
        ymax = adjust[b * 5 + 4] - 1.0f;
    }
    int total = 0;
    for (int g0 = 0; g0 < count; g0 += 256) {
        const int g = g0 + threadIdx.x;
        const bool f = g < count && ws.keep[base + g] != 0;
        const unsigned long long m = __ballot(f);
        __syncthreads();
        if (lane == 0) wsum[wave] = __popcll(m);
        __syncthreads();
        int pos = total + __popcll(m & ((1ull << lane) - 1ull));
        for (int w = 0; w < wave; ++w) pos += wsum[w];
        total += wsum[0] + wsum[1] + wsum[2] + wsum[3];
        if (f && pos < max_det) {
            const float4 box = ws.sbox[base + g];
            float x0 = box.x, y0 = box.y, x1 = box.x + box.z, y1 = box.y + box.w;
            if (adj) {
                x0 = clipf((x0 - pad_l) / scale, 0.0f, xmax);
                y0 = clipf((y0 - pad_t) / scale, 0.0f, ymax);
                x1 = clipf((x1 - pad_l) / scale, 0.0f, xmax);
                y1 = clipf((y1 - pad_t) / scale, 0.0f, ymax);
            }
            float* d = dets + ((size_t)b * max_det + pos) * 6;
            d[0] = x0; d[1] = y0;
            d[2] = adj ? x1 - x0 : box.z;
            d[3] = adj ? y1 - y0 : box.w;
            d[4] = ws.sprob[base + g];
            d[5] = (float)ws.slabel[base + g];
        }
    }
    if (threadIdx.x == 0) counts[b] = total;
}

}  // namespace

extern "C" {

void si_letterbox_geometry(int height_origin, int width_origin, int height_new, int width_new, int* height_resize,
                           int* width_resize, float* scale, int* padding_t, int* padding_l) {
    int hr = height_new, wr = width_new;
    float s = 1.0f;
    if ((long long)height_new * width_origin < (long long)width_new * height_origin) {
        s = (float)height_new / (float)height_origin;
        wr = (int)(width_origin * s);
    } else {
        s = (float)width_new / (float)width_origin;
        hr = (int)(height_origin * s);
    }
    if (height_resize) *height_resize = hr;
    if (width_resize) *width_resize = wr;
    if (scale) *scale = s;
    if (padding_t) *padding_t = (height_new - hr) / 2;
    if (padding_l) *padding_l = (width_new - wr) / 2;
}

int si_hip_letterbox_u8_f32(const unsigned char* resized_bgr, int height_resize, int width_resize, float* out,
                            int height_new, int width_new, int padding_t, int padding_l, si_stream_t stream) {
    if (!out || height_new <= 0 || width_new <= 0 || height_resize < 0 || width_resize < 0) return SI_E_BADARG;
    if ((height_resize > 0 && width_resize > 0) && !resized_bgr) return SI_E_BADARG;
    if ((long long)height_new * width_new * 3 > 0x7fffffffLL) return SI_E_UNSUPPORTED;
    hipLaunchKernelGGL(letterbox_kernel, dim3(si_grid_for((size_t)height_new * width_new * 3)), dim3(256), 0,
                       (hipStream_t)stream, resized_bgr, height_resize, width_resize, out, height_new, width_new,
                       padding_t, padding_l);
    return (int)hipGetLastError();
}

int si_hip_letterbox_batch_u8_f32(const unsigned char* resized_bgr, int n, size_t image_stride_bytes, int height_resize, int width_resize,
                                  float* out, int height_new, int width_new, int padding_t, int padding_l, si_stream_t stream) {
    if (n < 0 || !out || height_new <= 0 || width_new <= 0 || height_resize < 0 || width_resize < 0) return SI_E_BADARG;
    if ((height_resize > 0 && width_resize > 0) && !resized_bgr) return SI_E_BADARG;
    if ((long long)height_new * width_new * 3 > 0x7fffffffLL || n > 65535) return SI_E_UNSUPPORTED;
    if (n == 0) return 0;
    if (((long long)height_new * width_new * 3) % 4 != 0 || (reinterpret_cast<uintptr_t>(out) & 15) != 0) {
        for (int b = 0; b < n; ++b) {   // odd sizes: one launch per image
            const int rc = si_hip_letterbox_u8_f32(resized_bgr + (size_t)b * image_stride_bytes, height_resize, width_resize,
                                                   out + (size_t)b * height_new * width_new * 3, height_new, width_new, padding_t, padding_l, stream);
            if (rc != 0) return rc;
        }
        return 0;
    }
    const unsigned gx = si_grid_for((size_t)height_new * width_new * 3 / 4);
    hipLaunchKernelGGL(letterbox_batch_kernel, dim3(gx > 512 ? 512 : gx, n), dim3(256), 0, (hipStream_t)stream, resized_bgr, image_stride_bytes,
                       height_resize, width_resize, out, height_new, width_new, padding_t, padding_l);
    return (int)hipGetLastError();
}

int si_hip_resize_bilinear_u8c3(const unsigned char* src, int n, size_t src_stride_bytes, int src_h, int src_w, unsigned char* dst,
                                size_t dst_stride_bytes, int dst_h, int dst_w, si_stream_t stream) {
    if (n < 0 || src_h <= 0 || src_w <= 0 || dst_h <= 0 || dst_w <= 0 || (n > 0 && (!src || !dst))) return SI_E_BADARG;
    if ((long long)dst_h * dst_w > 0x7fffffffLL / 3 || (long long)src_h * src_w > 0x7fffffffLL / 3 || n > 65535) return SI_E_UNSUPPORTED;
    if (n == 0) return 0;
    const unsigned gx = si_grid_for((size_t)dst_h * dst_w);
    hipLaunchKernelGGL(resize_bilinear_u8c3_kernel, dim3(gx > 1024 ? 1024 : gx, n), dim3(256), 0, (hipStream_t)stream, src, src_stride_bytes,
                       src_h, src_w, dst, dst_stride_bytes, dst_h, dst_w);
    return (int)hipGetLastError();
}

int si_hip_resize_letterbox_batch_u8_f32(const unsigned char* frames_bgr, int n, size_t image_stride_bytes, int height_origin,
                                         int width_origin, float* out, int height_new, int width_new, si_stream_t stream) {
    if (n < 0 || !out || height_new <= 0 || width_new <= 0 || height_origin <= 0 || width_origin <= 0 || (n > 0 && !frames_bgr)) return SI_E_BADARG;
    if ((long long)height_new * width_new * 3 > 0x7fffffffLL || (long long)height_origin * width_origin > 0x7fffffffLL / 3 || n > 65535)
        return SI_E_UNSUPPORTED;
    if (n == 0) return 0;
    int hr = 0, wr = 0, pt = 0, pl = 0;
    float scale = 0.f;
    si_letterbox_geometry(height_origin, width_origin, height_new, width_new, &hr, &wr, &scale, &pt, &pl);
    if (hr <= 0 || wr <= 0) return SI_E_BADARG;
    const unsigned gx = si_grid_for((size_t)height_new * width_new);
    hipLaunchKernelGGL(resize_letterbox_batch_kernel, dim3(gx > 1024 ? 1024 : gx, n), dim3(256), 0, (hipStream_t)stream, frames_bgr,
                       image_stride_bytes, height_origin, width_origin, hr, wr, out, height_new, width_new, pt, pl);
    return (int)hipGetLastError();
}

size_t si_hip_yolo_postprocess_workspace_bytes(int n, int rows, int ne) {
    if (n <= 0 || rows <= 0 || ne < 5) return 0;
    return post_ws_layout(n, rows, ne - 5 + 1, nullptr, nullptr);
}

int si_hip_yolo_postprocess_f32(const float* pred, int n, int rows, int ne, float prob_threshold, float nms_threshold,
                                int agnostic, const float* adjust, float* dets, int* counts, int max_det,
                                void* workspace, size_t workspace_bytes, si_stream_t stream) {
    if (n < 0 || rows < 0 || ne < 6 || max_det < 0) return SI_E_BADARG;
    if (n == 0) return 0;
    if (!counts || (max_det > 0 && !dets)) return SI_E_BADARG;
    hipStream_t s = (hipStream_t)stream;
    if (rows == 0) {
        SI_HIP_TRY(hipMemsetAsync(counts, 0, sizeof(int) * n, s));
        return 0;
    }
    if (!pred || !workspace) return SI_E_BADARG;
    if ((size_t)RPB * ne * sizeof(float) > 64 * 1024) return SI_E_UNSUPPORTED;
    PostWs ws;
    const size_t need = post_ws_layout(n, rows, ne - 5 + 1, (char*)workspace, &ws);
    if (workspace_bytes < need) return SI_E_BADARG;
    SI_HIP_TRY(hipMemsetAsync(workspace, 0, ws.zero_bytes, s));
    hipLaunchKernelGGL(yolo_filter_kernel, dim3((rows + RPB - 1) / RPB, n), dim3(RPB), (size_t)RPB * ne * sizeof(float), s,
                       pred, rows, ne, prob_threshold, ws);
    hipLaunchKernelGGL(yolo_rank_sort_kernel, dim3((rows + SORT_T - 1) / SORT_T, n), dim3(SORT_T), 0, s, rows, ws);
    if (agnostic) {
        // one chain per image: a workgroup walks the confidence order against everything picked so far
        hipLaunchKernelGGL(yolo_nms_kernel, dim3(n), dim3(256), 0, s, rows, nms_threshold, 1, adjust, dets, counts,
                           max_det, ws);
    } else {
        hipLaunchKernelGGL(yolo_nms_segments_kernel, dim3((ws.nbins + 3) / 4, n), dim3(256), 0, s, rows, nms_threshold,
                           ws);
        hipLaunchKernelGGL(yolo_compact_kernel, dim3(n), dim3(256), 0, s, rows, adjust, dets, counts, max_det, ws);
    }
    return (int)hipGetLastError();
}

}  // extern "C"
