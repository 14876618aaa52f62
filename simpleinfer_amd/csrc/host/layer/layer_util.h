// layer_util.h -- small helpers shared by the concrete layers.
#ifndef SIMPLE_INFER_SRC_LAYER_UTIL_H_
#define SIMPLE_INFER_SRC_LAYER_UTIL_H_

#include <vector>

#include "si_hip.h"
#include "tensor.h"

namespace SimpleInfer {

struct Dims4 {
    int n = 0, h = 0, w = 0, c = 0;
    size_t pixels() const { return (size_t)n * h * w; }
};

// NHWC view of a rank-4 tensor
inline bool GetDims4(const Tensor& t, Dims4& d) {
    const std::vector<int>& s = t.Shape();
    if (s.size() != 4) return false;
    d.n = s[0]; d.h = s[1]; d.w = s[2]; d.c = s[3];
    return d.n > 0 && d.h > 0 && d.w > 0 && d.c > 0;
}

// any rank as [pixels, channels]: last dim = channels, the rest folded
inline bool GetPixelsChannels(const Tensor& t, size_t& pixels, int& c) {
    const std::vector<int>& s = t.Shape();
    if (s.empty()) return false;
    c = s.back();
    pixels = 1;
    for (size_t i = 0; i + 1 < s.size(); ++i) pixels *= (size_t)s[i];
    return c > 0 && pixels > 0;
}

// HBM buffer for layer parameters (weights, bias, grids): uploaded once, freed with the layer.
class DeviceBuffer {
public:
    DeviceBuffer() {}
    ~DeviceBuffer() { Free(); }
    DeviceBuffer(const DeviceBuffer&) = delete;
    DeviceBuffer& operator=(const DeviceBuffer&) = delete;

    // (re)allocates and copies synchronously; returns a C-ABI code
    int Upload(const void* host, size_t bytes) {
        if (bytes != bytes_ || !ptr_) {
            Free();
            const int rc = si_hip_malloc(&ptr_, bytes);
            if (rc != 0) return rc;
            bytes_ = bytes;
        }
        int rc = si_hip_memcpy_h2d(ptr_, host, bytes, nullptr);
        if (rc != 0) return rc;
        return si_hip_stream_sync(nullptr);
    }
    int Alloc(size_t bytes) {
        Free();
        const int rc = si_hip_malloc(&ptr_, bytes);
        if (rc == 0) bytes_ = bytes;
        return rc;
    }
    void Free() {
        if (ptr_) si_hip_free(ptr_);
        ptr_ = nullptr;
        bytes_ = 0;
    }
    template<typename T>
    T* As() const { return static_cast<T*>(ptr_); }
    size_t bytes() const { return bytes_; }

private:
    void* ptr_ = nullptr;
    size_t bytes_ = 0;
};

}  // namespace SimpleInfer

#endif
