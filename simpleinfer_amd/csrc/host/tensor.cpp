// tensor.cpp -- Tensor storage.  Ownership rules follow reference src/tensor.cpp:11-105
// (copy/assign alias, Allocate owns, Allocate(dt, shape) is a no-op when nothing changed);
// device tensors allocate HBM through the C-ABI instead of malloc.
#include "tensor.h"

#include <cstdlib>

#include "logger.h"
#include "si_hip.h"

namespace SimpleInfer {

Tensor::Tensor() {}

Tensor::Tensor(const DataType data_type, const std::vector<int>& shape, const bool allocate)
    : data_type_(data_type), shape_(shape) {
    if (allocate) Allocate(data_type, shape);
}

Tensor::Tensor(const DataType data_type, const std::vector<int>& shape, const MemoryType memory_type,
               const bool allocate)
    : data_type_(data_type), shape_(shape), memory_type_(memory_type) {
    if (allocate) Allocate();
}

Tensor::~Tensor() {
    Deallocate();
    shape_.clear();
    data_type_ = DataType::kNone;
}

Tensor::Tensor(const Tensor& t)
    : data_type_(t.data_type_), shape_(t.shape_), use_internal_data_(false), data_(t.data_),
      memory_type_(t.memory_type_), pixel_stride_(t.pixel_stride_) {}

Tensor& Tensor::operator=(const Tensor& t) {
    if (this == &t) return *this;
    Deallocate();
    data_type_ = t.data_type_;
    shape_ = t.shape_;
    use_internal_data_ = false;
    data_ = t.data_;
    memory_type_ = t.memory_type_;
    pixel_stride_ = t.pixel_stride_;
    return *this;
}

size_t Tensor::NumElements() const {
    if (shape_.empty()) return 0;
    size_t n = 1;
    for (int s : shape_) n *= (s > 0 ? (size_t)s : 0);
    return n;
}

size_t Tensor::ByteSize() const { return NumElements() * (size_t)ElementSize(data_type_); }

int Tensor::PixelStride() const {
    if (pixel_stride_ > 0) return pixel_stride_;
    return shape_.empty() ? 0 : shape_.back();
}

Status Tensor::Allocate() {
    const size_t bytes = ByteSize();
    if (bytes > 0) {
        if (memory_type_ == MemoryType::kDevice) {
            const int rc = si_hip_malloc(&data_, bytes);
            if (rc != 0) {
                LOG(ERROR) << "device allocation of " << bytes << " bytes failed: " << si_hip_error_string(rc);
                data_ = nullptr;
            }
        } else {
            data_ = malloc(bytes);
        }
        if (nullptr != data_) {
            use_internal_data_ = true;
            pixel_stride_ = 0;
            return Status::kSuccess;
        }
    }
    LOG(ERROR) << "Tensor Allocate Fail, size " << bytes;
    return Status::kFail;
}

Status Tensor::Allocate(const DataType data_type, const std::vector<int>& shape) {
    if (data_type_ == data_type && IsSameShape(shape_, shape) && use_internal_data_ && nullptr != data_) {
        return Status::kSuccess;
    }
    Deallocate();
    data_type_ = data_type;
    shape_ = shape;
    return Allocate();
}

Status Tensor::Deallocate() {
    if (!use_internal_data_) return Status::kFail;  // nothing owned (same return as the reference)
    if (nullptr != data_) {
        if (memory_type_ == MemoryType::kDevice)
            si_hip_free(data_);
        else
            free(data_);
        data_ = nullptr;
    }
    use_internal_data_ = false;
    return Status::kSuccess;
}

const DataType Tensor::GetDataType() const { return data_type_; }

const std::vector<int>& Tensor::Shape() const { return shape_; }

Status Tensor::SetData(void* data, const MemoryType memory_type) {
    if (use_internal_data_) return Status::kFail;
    data_ = data;
    memory_type_ = memory_type;
    pixel_stride_ = 0;
    return Status::kSuccess;
}

void Tensor::SetView(void* data, const MemoryType memory_type, const int pixel_stride) {
    Deallocate();
    data_ = data;
    memory_type_ = memory_type;
    pixel_stride_ = pixel_stride;
}

std::vector<int> Tensor::ShapeAs(const int rank) const {
    std::vector<int> out(rank > 0 ? rank : 0, 1);
    const int have = (int)shape_.size();
    if (rank <= 0) return out;
    if (rank <= have) {
        for (int i = 1; i < rank; ++i) out[rank - i] = shape_[have - i];
        int lead = 1;
        for (int i = 0; i <= have - rank; ++i) lead *= shape_[i];
        out[0] = lead;
    } else {
        for (int i = 0; i < have; ++i) out[rank - have + i] = shape_[i];
    }
    return out;
}

}  // namespace SimpleInfer
