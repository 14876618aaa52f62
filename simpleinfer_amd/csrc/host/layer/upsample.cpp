#include "upsample.h"

#include "layer_util.h"
#include "si_hip.h"

namespace SimpleInfer {

DEFINE_LAYER_REGISTRY(Upsample);

Status Upsample::Init(const pnnx::Operator* op) {
    CHECK_STATUS(Layer::Init(op));
    CHECK_BOOL(CheckParam(op, "mode", 4));
    if ("nearest" != op->params.at("mode").s) {
        LOG(ERROR) << "Upsample::Init fail [unsupport upsample mode]";
        return Status::kUnsupport;
    }
    upsample_mode_ = UpsampleMode::kNearest;
    CHECK_BOOL(CheckParam(op, "scale_factor", 6) || CheckParam(op, "size", 5));
    if (CheckParam(op, "scale_factor", 6)) {
        const std::vector<float>& v = op->params.at("scale_factor").af;
        CHECK_BOOL(2 == v.size());
        scale_factor_h_ = v[0];
        scale_factor_w_ = v[1];
    }
    return Status::kSuccess;
}

Status Upsample::Validate() {
    CHECK_STATUS(Layer::Validate());
    CHECK_STATUS(ValidateShape(1, 1));
    if (Status::kSuccess != ValidateFloat()) {
        LOG(ERROR) << "Upsample::Validate fail [unsupport input/output data type]";
        return Status::kUnsupport;
    }
    return Status::kSuccess;
}

Status Upsample::Forward(const Tensor& input, Tensor& output) {
    return RunOnDevice({&input}, {&output}, [this](const std::vector<Tensor>& in, std::vector<Tensor>& out) {
        Dims4 id, od;
        if (!GetDims4(in[0], id) || !GetDims4(out[0], od) || id.c != od.c || id.n != od.n) return Status::kErrorShape;
        if (IsHalf(in[0]) != IsHalf(out[0])) return Status::kUnsupport;
        if (IsHalf(in[0])) {
            // pure data movement: an fp16 tensor is copied as half as many 4-byte words
            if (id.c % 2 || in[0].PixelStride() % 2 || out[0].PixelStride() % 2) return Status::kUnsupport;
            return CheckHip(si_hip_upsample_nearest_f32(static_cast<const float*>(in[0].RawData()), id.n, id.h, id.w, id.c / 2,
                                                        in[0].PixelStride() / 2, scale_factor_h_, scale_factor_w_,
                                                        static_cast<float*>(out[0].RawData()), od.h, od.w,
                                                        out[0].PixelStride() / 2, Stream()),
                            "Upsample");
        }
        return CheckHip(si_hip_upsample_nearest_f32(in[0].Data<float>(), id.n, id.h, id.w, id.c, in[0].PixelStride(),
                                                    scale_factor_h_, scale_factor_w_, out[0].Data<float>(), od.h, od.w,
                                                    out[0].PixelStride(), Stream()),
                        "Upsample");
    });
}

}  // namespace SimpleInfer
