#include "adaptive_avg_pool_2d.h"

#include "layer_util.h"
#include "si_hip.h"

namespace SimpleInfer {

DEFINE_LAYER_REGISTRY(AdaptiveAvgPool2d);

Status AdaptiveAvgPool2d::Init(const pnnx::Operator* op) {
    CHECK_STATUS(Layer::Init(op));
    CHECK_BOOL(CheckParam(op, "output_size", 5));
    const std::vector<int>& v = op->params.at("output_size").ai;
    CHECK_BOOL(2 == v.size());
    output_h_ = v[0];
    output_w_ = v[1];
    return Status::kSuccess;
}

Status AdaptiveAvgPool2d::Validate() {
    CHECK_STATUS(Layer::Validate());
    CHECK_STATUS(ValidateShape(1, 1));
    if (Status::kSuccess != ValidateFloat()) {
        LOG(ERROR) << "AdaptiveAvgPool2d::Validate fail [unsupport input/output data type]";
        return Status::kUnsupport;
    }
    return Status::kSuccess;
}

Status AdaptiveAvgPool2d::Forward(const Tensor& input, Tensor& output) {
    return RunOnDevice({&input}, {&output}, [this](const std::vector<Tensor>& in, std::vector<Tensor>& out) {
        Dims4 id, od;
        if (!GetDims4(in[0], id) || !GetDims4(out[0], od) || id.c != od.c || id.n != od.n) return Status::kErrorShape;
        if (0 != id.h % od.h || 0 != id.w % od.w) {
            LOG(ERROR) << "AdaptiveAvgPool2d::Forward fail [unsupport input/output shape]";
            return Status::kUnsupport;
        }
        if (IsHalf(in[0]) != IsHalf(out[0])) return Status::kUnsupport;
        if (IsHalf(in[0]))
            return CheckHip(si_hip_adaptive_avgpool2d_f16(in[0].RawData(), id.n, id.h, id.w, id.c, in[0].PixelStride(),
                                                          out[0].RawData(), od.h, od.w, out[0].PixelStride(), Stream()),
                            "AdaptiveAvgPool2d");
        return CheckHip(si_hip_adaptive_avgpool2d_f32(in[0].Data<float>(), id.n, id.h, id.w, id.c, in[0].PixelStride(),
                                                      out[0].Data<float>(), od.h, od.w, out[0].PixelStride(), Stream()),
                        "AdaptiveAvgPool2d");
    });
}

}  // namespace SimpleInfer
