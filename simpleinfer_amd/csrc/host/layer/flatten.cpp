#include "flatten.h"

#include "layer_util.h"
#include "si_hip.h"

namespace SimpleInfer {

DEFINE_LAYER_REGISTRY(Flatten);

Status Flatten::Init(const pnnx::Operator* op) {
    CHECK_STATUS(Layer::Init(op));
    CHECK_BOOL(CheckParam(op, "start_dim", 2));
    start_dim_ = op->params.at("start_dim").i;
    CHECK_BOOL(CheckParam(op, "end_dim", 2));
    end_dim_ = op->params.at("end_dim").i;
    return Status::kSuccess;
}

Status Flatten::Validate() {
    CHECK_STATUS(Layer::Validate());
    CHECK_STATUS(ValidateShape(1, 1));
    if (Status::kSuccess != ValidateFloat()) {
        LOG(ERROR) << "Flatten::Validate fail [unsupport input/output data type]";
        return Status::kUnsupport;
    }
    return Status::kSuccess;
}

Status Flatten::Forward(const Tensor& input, Tensor& output) {
    return RunOnDevice({&input}, {&output}, [this](const std::vector<Tensor>& in, std::vector<Tensor>& out) {
        if (in[0].NumElements() != out[0].NumElements()) return Status::kErrorShape;
        Dims4 d;
        if (IsHalf(in[0]) || IsHalf(out[0])) {
            // fp16 path: only the permutation-free case (H = W = 1, what a classifier head has), copied as 4-byte words
            size_t pixels = 0;
            int c = 0;
            if (!(IsHalf(in[0]) && IsHalf(out[0])) || !GetPixelsChannels(in[0], pixels, c)) return Status::kUnsupport;
            if (GetDims4(in[0], d) && d.h * d.w != 1) return Status::kUnsupport;
            if (c % 2 || in[0].PixelStride() % 2) return Status::kUnsupport;
            return CheckHip(si_hip_copy_channels_f32(static_cast<const float*>(in[0].RawData()), pixels, c / 2,
                                                     in[0].PixelStride() / 2, static_cast<float*>(out[0].RawData()), c / 2, Stream()),
                            "Flatten");
        }
        if (GetDims4(in[0], d)) {
            return CheckHip(si_hip_nhwc_to_nchw_f32(in[0].Data<float>(), d.n, d.h, d.w, d.c, in[0].PixelStride(),
                                                    out[0].Data<float>(), Stream()),
                            "Flatten");
        }
        size_t pixels = 0;
        int c = 0;
        if (!GetPixelsChannels(in[0], pixels, c)) return Status::kErrorShape;
        return CheckHip(si_hip_copy_channels_f32(in[0].Data<float>(), pixels, c, in[0].PixelStride(), out[0].Data<float>(),
                                                 c, Stream()),
                        "Flatten");
    });
}

}  // namespace SimpleInfer
