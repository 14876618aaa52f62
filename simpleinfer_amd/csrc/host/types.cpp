// types.cpp -- dtype tables (contract: reference src/types.cpp:48-105).
#include "types.h"

#include <cstdint>

namespace SimpleInfer {

#define SI_SAME_TYPE(ctype, tag)                                  \
    template<>                                                    \
    bool IsSameDataType<ctype>(const DataType data_type) {        \
        return DataType::tag == data_type;                        \
    }
SI_SAME_TYPE(float, kFloat32)
SI_SAME_TYPE(double, kFloat64)
SI_SAME_TYPE(int32_t, kInt32)
SI_SAME_TYPE(int64_t, kInt64)
SI_SAME_TYPE(int16_t, kInt16)
SI_SAME_TYPE(int8_t, kInt8)
SI_SAME_TYPE(uint8_t, kUint8)
SI_SAME_TYPE(bool, kBool)
#undef SI_SAME_TYPE

DataType PnnxToDataType(int type) {
    // pnnx codes 1..12 follow the DataType enumerators in order
    return (type >= 1 && type <= 12) ? static_cast<DataType>(type) : DataType::kNone;
}

int ElementSize(const DataType data_type) {
    switch (data_type) {
        case DataType::kInt8: case DataType::kUint8: case DataType::kBool: return 1;
        case DataType::kFloat16: case DataType::kInt16: return 2;
        case DataType::kFloat32: case DataType::kInt32: case DataType::kComplex32: return 4;
        case DataType::kFloat64: case DataType::kInt64: case DataType::kComplex64: return 8;
        case DataType::kComplex128: return 16;
        default: return 0;
    }
}

bool IsSameShape(const std::vector<int>& shape0, const std::vector<int>& shape1) { return shape0 == shape1; }

const char* StatusString(Status s) {
    switch (s) {
        case Status::kSuccess: return "kSuccess";
        case Status::kFail: return "kFail";
        case Status::kEmpty: return "kEmpty";
        case Status::kErrorShape: return "kErrorShape";
        case Status::kErrorContext: return "kErrorContext";
        case Status::kUnsupport: return "kUnsupport";
    }
    return "?";
}

}  // namespace SimpleInfer
