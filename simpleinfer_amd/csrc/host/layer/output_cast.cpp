// layer/output_cast.cpp -- fp16 storage only: the step the engine appends after a layer that produced a graph output in
// half precision, so that Extract() hands back fp32 exactly like the reference's Engine::Extract (src/engine_impl.cpp:548).
#include "operators.h"

#include "layer_util.h"
#include "si_hip.h"

namespace SimpleInfer {

OutputCast::OutputCast(const std::string& producer, const char* suffix) {
    op_storage_.type = "si.OutputCast";
    op_storage_.name = producer + suffix;
    op_ = &op_storage_;
}

Status OutputCast::Validate() {
    CHECK_STATUS(Layer::Validate());
    CHECK_STATUS(ValidateShape(1, 1));
    const Tensor& in = InputNodes()[0]->tensor;
    const Tensor& out = OutputNodes()[0]->tensor;
    const bool down = in.GetDataType() == DataType::kFloat32 && IsHalf(out);
    const bool up = IsHalf(in) && out.GetDataType() == DataType::kFloat32;
    if (!(up || down) || in.NumElements() != out.NumElements()) {
        LOG(ERROR) << "OutputCast::Validate fail [expects a half and an fp32 tensor of the same size]";
        return Status::kUnsupport;
    }
    to_half_ = down;
    return Status::kSuccess;
}

Status OutputCast::Forward(const Tensor& input, Tensor& output) {
    return RunOnDevice({&input}, {&output}, [this](const std::vector<Tensor>& in, std::vector<Tensor>& out) {
        size_t pixels = 0;
        int c = 0;
        if (!GetPixelsChannels(in[0], pixels, c)) return Status::kErrorShape;
        const int out_ld = out[0].PixelStride() > 0 ? out[0].PixelStride() : c;
        if (to_half_)
            return CheckHip(si_hip_convert_f32_f16(in[0].Data<float>(), pixels, c, in[0].PixelStride(), out[0].RawData(), out_ld, Stream()),
                            "OutputCast (fp32 -> fp16)");
        return CheckHip(si_hip_convert_f16_f32(in[0].RawData(), pixels, c, in[0].PixelStride(), out[0].Data<float>(), out_ld,
                                               Stream()),
                        "OutputCast");
    });
}

}  // namespace SimpleInfer
