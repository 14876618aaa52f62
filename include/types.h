// types.h -- public enums and helpers of the SimpleInfer API, kept source compatible with the
// reference's include/types.h:8-59 (DataType, Status, CHECK_BOOL / CHECK_STATUS, IsSameDataType,
// PnnxToDataType, ElementSize, IsSameShape).  MemoryType is new: tensors may live in HBM.
#pragma once

#include <vector>

namespace SimpleInfer {

// enumerators in the order of pnnx's operand type codes 0..12 (src/pnnx/ir.cpp:24-60), so PnnxToDataType is a cast
enum class DataType {
    kNone = 0,    // "null"
    kFloat32,     // 1  f32   the arithmetic of the whole path (and the only type the reference's layers accept)
    kFloat64,     // 2  f64
    kFloat16,     // 3  f16   internal storage of the fp16 engine option
    kInt32,       // 4  i32
    kInt64,       // 5  i64
    kInt16,       // 6  i16
    kInt8,        // 7  i8
    kUint8,       // 8  u8
    kBool,        // 9  bool
    kComplex64,   // 10 cp64
    kComplex128,  // 11 cp128
    kComplex32    // 12 cp32
};

enum class Status { kSuccess = 0, kFail, kEmpty, kErrorShape, kErrorContext, kUnsupport };

// where a Tensor's bytes live (extension; the reference is host-only)
enum class MemoryType { kHost = 0, kDevice = 1 };

#define CHECK_BOOL(b)                                          \
    {                                                          \
        const bool _b = (b);                                   \
        if (!_b) { return ::SimpleInfer::Status::kFail; }      \
    }

#define CHECK_STATUS(s)                                        \
    {                                                          \
        const ::SimpleInfer::Status _status = (s);             \
        if (::SimpleInfer::Status::kSuccess != _status) {      \
            return _status;                                    \
        }                                                      \
    }

template<typename T>
bool IsSameDataType(const DataType data_type);

DataType PnnxToDataType(int type);

int ElementSize(const DataType data_type);

bool IsSameShape(const std::vector<int>& shape0, const std::vector<int>& shape1);

const char* StatusString(Status s);

}  // namespace SimpleInfer
