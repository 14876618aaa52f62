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

bool MaxPool2d::ChainHead(const MaxPool2d& next) const {
    auto sppf = [](const MaxPool2d& p) {
        return p.kernel_h_ == 5 && p.kernel_w_ == 5 && p.stride_h_ == 1 && p.stride_w_ == 1 && p.dilation_h_ == 1 &&
               p.dilation_w_ == 1 && p.padding_t_ == 2 && p.padding_l_ == 2 && !p.return_indices_;
    };
    return sppf(*this) && sppf(next);
}

Status MaxPool2d::Forward(const Tensor& input, std::vector<Tensor>& outputs) {
    if (chain_.size() != 2 || outputs.size() != 3) return Status::kUnsupport;
    Status st = RunOnDevice({&input}, {&outputs[0], &outputs[1], &outputs[2]},
                            [this](const std::vector<Tensor>& in, std::vector<Tensor>& out) {
        Dims4 id;
        if (!GetDims4(in[0], id)) return Status::kErrorShape;
        for (const Tensor& o : out) {
            Dims4 od;
            if (!GetDims4(o, od) || od.n != id.n || od.h != id.h || od.w != id.w || od.c != id.c) return Status::kErrorShape;
            if (IsHalf(o) != IsHalf(in[0])) return Status::kUnsupport;
        }
        const int rc = IsHalf(in[0])
                           ? si_hip_maxpool5_chain3_f16(in[0].RawData(), id.n, id.h, id.w, id.c, in[0].PixelStride(), out[0].RawData(),
                                                        out[0].PixelStride(), out[1].RawData(), out[1].PixelStride(), out[2].RawData(),
                                                        out[2].PixelStride(), Stream())
                           : si_hip_maxpool5_chain3_f32(in[0].Data<float>(), id.n, id.h, id.w, id.c, in[0].PixelStride(),
                                                        out[0].Data<float>(), out[0].PixelStride(), out[1].Data<float>(),
                                                        out[1].PixelStride(), out[2].Data<float>(), out[2].PixelStride(), Stream());
        if (rc == SI_E_UNSUPPORTED) return Status::kEmpty;  // not an error: the caller runs the three pools one by one
        return CheckHip(rc, "MaxPool2d chain");
    });
    if (st != Status::kEmpty) return st;
    CHECK_STATUS(Forward(input, outputs[0]));
    CHECK_STATUS(chain_[0]->Forward(outputs[0], outputs[1]));
    return chain_[1]->Forward(outputs[1], outputs[2]);
}

}  // namespace SimpleInfer
