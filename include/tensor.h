// tensor.h -- SimpleInfer::Tensor, source compatible with the reference's include/tensor.h:13-69
// for everything that does not name Eigen: same constructors, Allocate/Deallocate, GetDataType,
// Shape, and the same ownership rules (copy / assignment make a NON-OWNING alias, reference
// src/tensor.cpp:28-45; Allocate() owns, :47-69).
//
// Differences, all forced by the target:
//  * Eigen is not a dependency.  SetEigenTensor / GetEigenTensor (tensor.h:39-61) become
//    SetData() / Data<T>() -- a raw pointer plus Shape(); the user-facing layout is still NHWC.
//  * A tensor may live in HBM (MemoryType::kDevice); then Data() is a device pointer.
//  * A tensor may be a channel slice of a wider buffer: PixelStride() is the distance in
//    elements between consecutive pixels (== Shape().back() when dense).
#pragma once

#include <cassert>
#include <cstddef>
#include <vector>

#include "types.h"

namespace SimpleInfer {

class Tensor {
public:
    Tensor();                                                                                        // empty, kNone
    Tensor(const DataType data_type, const std::vector<int>& shape, const bool allocate = false);   // host memory
    Tensor(const DataType data_type, const std::vector<int>& shape, const MemoryType memory_type,   // host or HBM
           const bool allocate);
    ~Tensor();                                  // frees only what this object allocated itself
    Tensor(const Tensor& tensor);               // copies ALIAS: non-owning view of the same bytes (reference Q6)
    Tensor& operator=(const Tensor& tensor);    // same

public:
    Status Allocate();                                                          // malloc / hipMalloc for the current shape
    Status Allocate(const DataType data_type, const std::vector<int>& shape);  // no-op when nothing changes
    Status Deallocate();
    const DataType GetDataType() const;
    const std::vector<int>& Shape() const;      // NHWC for rank 4 (the reference's in-memory layout)

public:
    // borrow caller memory (replaces SetEigenTensor; fails if this tensor owns its buffer)
    Status SetData(void* data, const MemoryType memory_type = MemoryType::kHost);

    template<typename T>
    T* Data() const {
        assert(IsSameDataType<T>(data_type_));
        return static_cast<T*>(data_);
    }

    void* RawData() const { return data_; }

    MemoryType GetMemoryType() const { return memory_type_; }

    bool OwnsData() const { return use_internal_data_; }

    size_t NumElements() const;

    size_t ByteSize() const;

    // elements between consecutive pixels (last-dim rows); dense tensors return Shape().back()
    int PixelStride() const;

    // engine-internal: make this tensor a non-owning view with an explicit pixel stride
    void SetView(void* data, const MemoryType memory_type, const int pixel_stride);

    // rank-adapting shape: leading dims folded (or padded with 1) to exactly `rank` dims, the rule
    // of reference include/eigen_helper.h:32-63
    std::vector<int> ShapeAs(const int rank) const;

protected:
    DataType data_type_ = DataType::kNone;

    std::vector<int> shape_;

    bool use_internal_data_ = false;
    void* data_             = nullptr;

    MemoryType memory_type_ = MemoryType::kHost;
    int pixel_stride_       = 0;  // 0 = dense
};

}  // namespace SimpleInfer


