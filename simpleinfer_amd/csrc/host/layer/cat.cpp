#include "cat.h"

#include "layer_util.h"
#include "si_hip.h"

namespace SimpleInfer {

DEFINE_LAYER_REGISTRY(Cat);

Status Cat::Init(const pnnx::Operator* op) {
    CHECK_STATUS(Layer::Init(op));
    CHECK_BOOL(CheckParam(op, "dim", 2));
    dim_ = op->params.at("dim").i;
    return Status::kSuccess;
}

Status Cat::Validate() {
    CHECK_STATUS(Layer::Validate());
    CHECK_STATUS(ValidateShape(-1, 1));
    if (Status::kSuccess != ValidateFloat()) {
        LOG(ERROR) << "Cat::Validate fail [unsupport data type]";
        return Status::kUnsupport;
    }
    return Status::kSuccess;
}

int Cat::NhwcAxis() const {
    switch (dim_) {
        case 1: return 3;
        case 2: return 1;
        case 3: return 2;
        default: return dim_;
    }
}

Status Cat::Forward(const std::vector<Tensor>& inputs, Tensor& output) {
    if (4 != (int)output.Shape().size()) {
        LOG(ERROR) << "Cat::Forward fail [unsupport output shape]";
        return Status::kUnsupport;
    }
    if (inputs.size() < 2) {
        LOG(ERROR) << "Cat::Forward fail [unsupport inputs size]";
        return Status::kUnsupport;
    }
    const int axis = NhwcAxis();
    if (axis < 0 || axis > 3) return Status::kUnsupport;

    std::vector<const Tensor*> ins;
    for (auto& t : inputs) ins.push_back(&t);
    return RunOnDevice(ins, {&output}, [this, axis](const std::vector<Tensor>& in, std::vector<Tensor>& out) {
        Dims4 od;
        if (!GetDims4(out[0], od)) return Status::kErrorShape;
        const std::vector<int>& os = out[0].Shape();
        int offset = 0;
        int copies = 0;
        const bool half = IsHalf(out[0]);
        // fp16 tensors are copied as 4-byte words (pure data movement): channel counts / strides / offsets must be even
        const int wd = half ? 2 : 1;
        for (const Tensor& t : in) {
            Dims4 id;
            if (!GetDims4(t, id)) return Status::kErrorShape;
            if (IsHalf(t) != half) return Status::kUnsupport;
            const std::vector<int>& is = t.Shape();
            if (axis == 3) {
                char* dst = static_cast<char*>(out[0].RawData()) + (size_t)offset * (half ? 2 : 4);
                // already in place: the engine pointed this input at its slice of our output
                if (t.RawData() != dst || t.PixelStride() != out[0].PixelStride()) {
                    ++copies;
                    if (id.c % wd || t.PixelStride() % wd || out[0].PixelStride() % wd || offset % wd) return Status::kUnsupport;
                    CHECK_STATUS(CheckHip(si_hip_copy_channels_f32(static_cast<const float*>(t.RawData()), id.pixels(), id.c / wd,
                                                                   t.PixelStride() / wd, reinterpret_cast<float*>(dst),
                                                                   out[0].PixelStride() / wd, Stream()),
                                          "Cat"));
                }
            } else {
                ++copies;
                if (t.PixelStride() != id.c || out[0].PixelStride() != od.c) return Status::kUnsupport;
                if (half) {
                    if (id.c % 2) return Status::kUnsupport;
                    int isw[4] = {is[0], is[1], is[2], is[3] / 2}, osw[4] = {os[0], os[1], os[2], os[3] / 2};
                    CHECK_STATUS(CheckHip(si_hip_cat_axis_f32(static_cast<const float*>(t.RawData()), isw,
                                                              static_cast<float*>(out[0].RawData()), osw, axis, offset, Stream()),
                                          "Cat"));
                } else {
                    CHECK_STATUS(CheckHip(si_hip_cat_axis_f32(t.Data<float>(), is.data(), out[0].Data<float>(), os.data(), axis,
                                                              offset, Stream()),
                                          "Cat"));
                }
            }
            offset += is[axis];
        }
        last_copies_ = copies;
        return offset == os[axis] ? Status::kSuccess : Status::kErrorShape;
    });
}

}  // namespace SimpleInfer
