#include "max_pool_2d.h"

#include <cstring>

#include "layer_util.h"
#include "si_hip.h"

namespace SimpleInfer {

DEFINE_LAYER_REGISTRY(MaxPool2d);

Status MaxPool2d::Init(const pnnx::Operator* op) {
    CHECK_STATUS(Layer::Init(op));
    CHECK_BOOL(CheckParam(op, "ceil_mode", 1));
    ceil_mode_ = op->params.at("ceil_mode").b;
    CHECK_BOOL(CheckParam(op, "return_indices", 1));
    return_indices_ = op->params.at("return_indices").b;

    struct IntPair { const char* key; int* a; int* b; };
    const IntPair pairs[] = {{"padding", &padding_t_, &padding_l_},
                             {"kernel_size", &kernel_h_, &kernel_w_},
                             {"stride", &stride_h_, &stride_w_},
                             {"dilation", &dilation_h_, &dilation_w_}};
    for (const IntPair& p : pairs) {
        CHECK_BOOL(CheckParam(op, p.key, 5));
        const std::vector<int>& v = op->params.at(p.key).ai;
        CHECK_BOOL(2 == v.size());
        *p.a = v[0];
        *p.b = v[1];
    }
    padding_b_ = padding_t_;
    padding_r_ = padding_l_;
    return Status::kSuccess;
}

Status MaxPool2d::Validate() {
    CHECK_STATUS(Layer::Validate());
    CHECK_STATUS(ValidateShape(1, 1));
    if (Status::kSuccess != ValidateFloat()) {
        LOG(ERROR) << "MaxPool2d::Validate fail [unsupport input/output data type]";
        return Status::kUnsupport;
    }
    return Status::kSuccess;
}

Status MaxPool2d::Forward(const Tensor& input, Tensor& output) {
    return RunOnDevice({&input}, {&output}, [this](const std::vector<Tensor>& in, std::vector<Tensor>& out) {
        Dims4 id, od;
        if (!GetDims4(in[0], id) || !GetDims4(out[0], od) || id.c != od.c || id.n != od.n) return Status::kErrorShape;
        SiPool2dDesc d;
        memset(&d, 0, sizeof(d));
        d.n = id.n; d.ih = id.h; d.iw = id.w; d.c = id.c; d.in_ld = in[0].PixelStride();
        d.oh = od.h; d.ow = od.w; d.out_ld = out[0].PixelStride();
        d.kh = kernel_h_; d.kw = kernel_w_; d.sh = stride_h_; d.sw = stride_w_;
        d.dh = dilation_h_; d.dw = dilation_w_; d.pt = padding_t_; d.pl = padding_l_;
        if (IsHalf(in[0]) != IsHalf(out[0])) return Status::kUnsupport;
        if (IsHalf(in[0])) return CheckHip(si_hip_maxpool2d_f16(&d, in[0].RawData(), out[0].RawData(), Stream()), "MaxPool2d");
        return CheckHip(si_hip_maxpool2d_f32(&d, in[0].Data<float>(), out[0].Data<float>(), Stream()), "MaxPool2d");
    });
}

}  // namespace SimpleInfer
