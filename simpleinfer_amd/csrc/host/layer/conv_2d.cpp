#include "conv_2d.h"

#include <cstring>

#include "layer_util.h"

namespace SimpleInfer {

DEFINE_LAYER_REGISTRY(Conv2d);

Conv2d::Conv2d() {}

Conv2d::~Conv2d() {}

Status Conv2d::Init(const pnnx::Operator* op) {
    CHECK_STATUS(Layer::Init(op));
    return Init(op->params, op->attrs);
}

// keys and their pnnx types: reference src/layer/conv_2d.cpp:27-70 (a missing key is kFail)
Status Conv2d::Init(const std::map<std::string, pnnx::Parameter>& params,
                    const std::map<std::string, pnnx::Attribute>& attrs) {
    CHECK_BOOL(CheckParam(params, "padding_mode", 4));
    const std::string& mode = params.at("padding_mode").s;
    if (mode == "zeros") {
        padding_mode_ = PaddingMode::kZeros;
    } else if (mode == "replicate") {
        padding_mode_ = PaddingMode::kReplicate;
    } else if (mode == "reflect") {
        padding_mode_ = PaddingMode::kReflect;
    } else {
        LOG(ERROR) << "Conv2d::Init fail [unsupport padding mode " << mode << "]";
        return Status::kUnsupport;
    }

    struct IntPair { const char* key; int* a; int* b; };
    const IntPair pairs[] = {{"padding", &padding_t_, &padding_l_},
                             {"kernel_size", &kernel_h_, &kernel_w_},
                             {"stride", &stride_h_, &stride_w_},
                             {"dilation", &dilation_h_, &dilation_w_}};
    for (const IntPair& p : pairs) {
        CHECK_BOOL(CheckParam(params, p.key, 5));
        const std::vector<int>& v = params.at(p.key).ai;
        CHECK_BOOL(2 == v.size());
        *p.a = v[0];
        *p.b = v[1];
    }
    padding_b_ = padding_t_;
    padding_r_ = padding_l_;

    CHECK_BOOL(CheckParam(params, "groups", 2));
    groups_ = params.at("groups").i;
    CHECK_BOOL(CheckParam(params, "in_channels", 2));
    in_channels_ = params.at("in_channels").i;
    CHECK_BOOL(CheckParam(params, "out_channels", 2));
    out_channels_ = params.at("out_channels").i;
    CHECK_BOOL(groups_ > 0);

    // weights (reference InitWeightAndBias, conv_2d.cpp:120-180)
    CHECK_BOOL(CheckAttr(attrs, "weight", 1));
    const pnnx::Attribute& w = attrs.at("weight");
    CHECK_BOOL(4 == w.shape.size());
    CHECK_BOOL(w.shape[0] == out_channels_ && w.shape[2] == kernel_h_ && w.shape[3] == kernel_w_);
    const size_t w_count = (size_t)w.shape[0] * w.shape[1] * w.shape[2] * w.shape[3];
    CHECK_BOOL(w.data.size() == w_count * sizeof(float));
    weight_.resize(w_count);
    memcpy(weight_.data(), w.data.data(), w.data.size());

    CHECK_BOOL(CheckParam(params, "bias", 1));
    use_bias_ = params.at("bias").b;
    bias_.clear();
    if (use_bias_) {
        CHECK_BOOL(CheckAttr(attrs, "bias", 1));
        const pnnx::Attribute& b = attrs.at("bias");
        CHECK_BOOL(1 == b.shape.size() && b.shape[0] == out_channels_);
        CHECK_BOOL(b.data.size() == (size_t)out_channels_ * sizeof(float));
        bias_.resize(out_channels_);
        memcpy(bias_.data(), b.data.data(), b.data.size());
    }
    device_ready_ = false;

    if (padding_mode_ != PaddingMode::kZeros) {
        // the reference parses replicate/reflect but every path pads with zeros (conv_2d.cpp:260)
        LOG(WARNING) << "Conv2d: padding_mode " << mode << " computes with zero padding (as the reference does)";
    }
    return Status::kSuccess;
}

Status Conv2d::SetWeights(const std::vector<float>& weight_oihw, const std::vector<float>& bias) {
    weight_ = weight_oihw;
    bias_ = bias;
    use_bias_ = !bias.empty();
    device_ready_ = false;
    return Status::kSuccess;
}

void Conv2d::SetFusion(int act1, TensorNode* residual, int act2, float act_param) {
    act1_ = act1;
    residual_node_ = residual;
    act2_ = act2;
    act_param_ = act_param;
}

bool Conv2d::CanFuseSibling(const Conv2d& o) const {
    auto plain1x1 = [](const Conv2d& c) {
        return c.kernel_h_ == 1 && c.kernel_w_ == 1 && c.stride_h_ == 1 && c.stride_w_ == 1 && c.dilation_h_ == 1 &&
               c.dilation_w_ == 1 && c.padding_t_ == 0 && c.padding_l_ == 0 && c.groups_ == 1 && c.residual_node_ == nullptr &&
               c.act2_ == SI_ACT_NONE && c.sibling_ == nullptr;
    };
    return this != &o && plain1x1(*this) && plain1x1(o) && in_channels_ == o.in_channels_ && act1_ == o.act1_ &&
           act_param_ == o.act_param_ && use_bias_ == o.use_bias_ && out_channels_ % 32 == 0 && in_channels_ % 32 == 0;
}

bool Conv2d::CanFuseStemProducer(const Conv2d& stem) const {
    if (residual_node_ || sibling_ || up_node_ || stem_producer_ || stem.residual_node_ || stem.sibling_ || stem.up_node_) return false;
    if (input_tensor_nodes_.size() != 1 || output_tensor_nodes_.size() != 1 || stem.input_tensor_nodes_.size() != 1 ||
        stem.output_tensor_nodes_.size() != 1 || stem.output_tensor_nodes_[0] != input_tensor_nodes_[0])
        return false;
    const Tensor& img = stem.input_tensor_nodes_[0]->tensor;
    const Tensor& mid = input_tensor_nodes_[0]->tensor;
    const Tensor& out = output_tensor_nodes_[0]->tensor;
    if (IsHalf(img) || !IsHalf(mid) || !IsHalf(out)) return false;
    if (img.Shape().size() != 4 || mid.Shape().size() != 4 || out.Shape().size() != 4) return false;
    SiConv2dDesc d0 = stem.MakeDesc(img, mid), d1 = MakeDesc(mid, out);
    d0.in_ld = d0.ic; d0.out_ld = d0.oc; d1.in_ld = d1.ic;   // (dense views: the image as handed over, the intermediate nowhere)
    return si_hip_conv2d_stem_s2c32_f16_supported(&d0, &d1) == 1;
}

void Conv2d::SetStemProducer(Conv2d* stem) {
    stem_producer_ = stem;
    stem_mid_ = input_tensor_nodes_.empty() ? nullptr : input_tensor_nodes_[0];
    if (stem) SetInputNodes(stem->InputNodes());
    device_ready_ = false;
}

bool Conv2d::CanFuseStemPairProducer(const Conv2d& c1) const {
    if (!c1.stem_producer_ || !c1.stem_mid_ || c1.residual_node_ || c1.sibling_ || c1.up_node_ || c1.pw_producer_ || c1.stem_pair_) return false;
    if (residual_node_ || up_node_ || stem_producer_ || pw_producer_ || stem_pair_ || act2_ != SI_ACT_NONE) return false;
    if (input_tensor_nodes_.size() != 1 || c1.output_tensor_nodes_.size() != 1 || c1.output_tensor_nodes_[0] != input_tensor_nodes_[0]) return false;
    if (output_tensor_nodes_.size() != (sibling_ ? 2u : 1u) || c1.input_tensor_nodes_.size() != 1) return false;
    const Tensor& img = c1.input_tensor_nodes_[0]->tensor;   // (c1 reads the image since it took the stem)
    const Tensor& smid = c1.stem_mid_->tensor;
    const Tensor& mid = input_tensor_nodes_[0]->tensor;
    if (IsHalf(img) || !IsHalf(mid)) return false;
    for (const TensorNode* o : output_tensor_nodes_)
        if (!IsHalf(o->tensor) || o->tensor.Shape().size() != 4) return false;
    if (img.Shape().size() != 4 || mid.Shape().size() != 4) return false;
    SiConv2dDesc d0 = c1.stem_producer_->MakeDesc(img, smid), d1 = c1.MakeDesc(smid, mid), d2 = MakeDesc(mid, output_tensor_nodes_[0]->tensor);
    d0.in_ld = d0.ic; d0.out_ld = d0.oc; d1.in_ld = d1.ic; d1.out_ld = d1.oc; d2.in_ld = d2.ic;
    int split = 0;
    if (sibling_) {
        split = out_channels_;
        d2.oc = out_channels_ + sibling_->out_channels_;
    }
    return si_hip_conv2d_stem_s2c32_pw_f16_supported(&d0, &d1, &d2, split) == 1;
}

void Conv2d::SetStemPairProducer(Conv2d* c1) {
    stem_pair_ = c1;
    stem_pair_mid_ = input_tensor_nodes_.empty() ? nullptr : input_tensor_nodes_[0];
    if (c1) SetInputNodes(c1->InputNodes());
    device_ready_ = false;
}

// the RGB stem, the 3x3 stride-2 conv behind it and this 1x1 conv (+ its sibling) in one launch: `image` is the fp32 image
Status Conv2d::LaunchStemTriple(const Tensor& image, Tensor& out0, Tensor* out1) {
    Conv2d* const c1 = stem_pair_;
    Conv2d* const stem = c1 ? c1->stem_producer_ : nullptr;
    if (!stem || !c1->stem_mid_ || !stem_pair_mid_ || IsHalf(image) || !IsHalf(out0) || (out1 && !IsHalf(*out1)) || (sibling_ != nullptr) != (out1 != nullptr))
        return Status::kUnsupport;
    CHECK_STATUS(PrepareDevice(1));
    CHECK_STATUS(c1->PrepareDevice(1));
    CHECK_STATUS(stem->PrepareDevice(2));
    SiConv2dDesc d0 = stem->MakeDesc(image, c1->stem_mid_->tensor), d1 = c1->MakeDesc(c1->stem_mid_->tensor, stem_pair_mid_->tensor);
    SiConv2dDesc d2 = MakeDesc(stem_pair_mid_->tensor, out0);
    d0.out_ld = d0.oc; d1.in_ld = d1.ic; d1.out_ld = d1.oc; d2.in_ld = d2.ic;
    int split = 0;
    if (sibling_) {
        split = out_channels_;
        d2.oc = out_channels_ + sibling_->out_channels_;
    }
    return CheckHip(si_hip_conv2d_stem_s2c32_pw_f16(&d0, &d1, &d2, image.Data<float>(), stem->weight_dev_.As<void>(),
                                                    stem->use_bias_ ? stem->bias_dev_.As<float>() : nullptr, c1->weight_dev_.As<void>(),
                                                    c1->use_bias_ ? c1->bias_dev_.As<float>() : nullptr, weight_dev_.As<void>(),
                                                    use_bias_ ? bias_dev_.As<float>() : nullptr, out0.RawData(), split,
                                                    out1 ? out1->RawData() : nullptr, out1 ? out1->PixelStride() : 0, Stream()),
                    "conv2d stem + 3x3 s2 + 1x1 (fp16, one launch)");
}

bool Conv2d::CanFuseCv3Pair(const Conv2d& pair, const TensorNode* z) const {
    if (!pair.pw_producer_ || !pair.pw_mid_ || pair.sibling_ || pair.up_node_ || pair.stem_producer_ || pair.stem_pair_ || pair.cv3_pair_) return false;
    if (residual_node_ || sibling_ || up_node_ || stem_producer_ || stem_pair_ || pw_producer_ || cv3_pair_ || act2_ != SI_ACT_NONE) return false;
    if (!z || input_tensor_nodes_.size() != 1 || output_tensor_nodes_.size() != 1 || pair.output_tensor_nodes_.size() != 1 || pair.input_tensor_nodes_.size() != 1) return false;
    const Tensor& x = pair.input_tensor_nodes_[0]->tensor;     // (the pair reads its 1x1 conv's input)
    const Tensor& y = pair.output_tensor_nodes_[0]->tensor;
    const Tensor& cat = input_tensor_nodes_[0]->tensor;
    const Tensor& out = output_tensor_nodes_[0]->tensor;
    if (!IsHalf(x) || !IsHalf(y) || !IsHalf(z->tensor) || !IsHalf(cat) || !IsHalf(out)) return false;
    if (pair.residual_node_ && !IsHalf(pair.residual_node_->tensor)) return false;
    Dims4 dy, dz, dc;
    if (!GetDims4(y, dy) || !GetDims4(z->tensor, dz) || !GetDims4(cat, dc)) return false;
    if (dy.c != 64 || dz.c != 64 || dc.c != 128 || dy.pixels() != dz.pixels() || dy.pixels() != dc.pixels() || dy.n != dz.n) return false;
    SiConv2dDesc d0 = pair.pw_producer_->MakeDesc(x, pair.pw_mid_->tensor), d1 = pair.MakeDesc(pair.pw_mid_->tensor, y), d2 = MakeDesc(cat, out);
    d0.in_ld = d0.ic; d0.out_ld = d0.oc; d1.in_ld = d1.ic; d1.out_ld = d1.oc; d2.in_ld = d2.ic; d2.out_ld = d2.oc;
    d1.has_residual = pair.residual_node_ ? 1 : 0;
    d1.res_ld = d1.oc;
    return si_hip_conv2d_pw_cv3_f16_supported(&d0, &d1, &d2) == 1;
}

void Conv2d::SetCv3Pair(Conv2d* pair, TensorNode* z) {
    cv3_pair_ = pair;
    cv3_z_ = z;
    cv3_cat_ = input_tensor_nodes_.empty() ? nullptr : input_tensor_nodes_[0];
    if (pair) SetInputNodes(pair->InputNodes());
    device_ready_ = false;
}

// the C3's last bottleneck pair and this closing 1x1 conv in one launch: `x` is the pair's input
Status Conv2d::LaunchCv3(const Tensor& x, Tensor& output) {
    Conv2d* const pair = cv3_pair_;
    Conv2d* const pw = pair ? pair->pw_producer_ : nullptr;
    if (!pw || !pair->pw_mid_ || !cv3_z_ || !cv3_cat_ || pair->output_tensor_nodes_.empty() || !IsHalf(x) || !IsHalf(output)) return Status::kUnsupport;
    CHECK_STATUS(PrepareDevice(1));
    CHECK_STATUS(pair->PrepareDevice(1));
    CHECK_STATUS(pw->PrepareDevice(1));
    const Tensor& y = pair->output_tensor_nodes_[0]->tensor;   // (shape only: never allocated)
    SiConv2dDesc d0 = pw->MakeDesc(x, pair->pw_mid_->tensor), d1 = pair->MakeDesc(pair->pw_mid_->tensor, y), d2 = MakeDesc(cv3_cat_->tensor, output);
    d0.out_ld = d0.oc; d1.in_ld = d1.ic; d1.out_ld = d1.oc; d2.in_ld = d2.ic;
    const Tensor* res = pair->residual_node_ ? &pair->residual_node_->tensor : nullptr;
    if (res) {
        d1.has_residual = 1;
        d1.res_ld = res->PixelStride();
    }
    const Tensor& z = cv3_z_->tensor;
    return CheckHip(si_hip_conv2d_pw_cv3_f16(&d0, &d1, &d2, x.RawData(), pw->weight_dev_.As<void>(), pw->use_bias_ ? pw->bias_dev_.As<float>() : nullptr,
                                             pair->weight_dev_.As<void>(), pair->use_bias_ ? pair->bias_dev_.As<float>() : nullptr,
                                             res ? res->RawData() : nullptr, z.RawData(), z.PixelStride(), weight_dev_.As<void>(),
                                             use_bias_ ? bias_dev_.As<float>() : nullptr, output.RawData(), Stream()),
                    "conv2d 1x1 + 3x3 + cat + 1x1 (fp16, one launch)");
}

bool Conv2d::CanFusePointwiseProducer(const Conv2d& pw) const {
    if (sibling_ || up_node_ || stem_producer_ || pw_producer_ || pw.residual_node_ || pw.sibling_ || pw.up_node_ || pw.stem_producer_ || pw.pw_producer_) return false;
    if (input_tensor_nodes_.size() != 1 || output_tensor_nodes_.size() != 1 || pw.input_tensor_nodes_.size() != 1 ||
        pw.output_tensor_nodes_.size() != 1 || pw.output_tensor_nodes_[0] != input_tensor_nodes_[0])
        return false;
    const Tensor& x = pw.input_tensor_nodes_[0]->tensor;
    const Tensor& mid = input_tensor_nodes_[0]->tensor;
    const Tensor& out = output_tensor_nodes_[0]->tensor;
    if (!IsHalf(x) || !IsHalf(mid) || !IsHalf(out)) return false;
    if (residual_node_ && !IsHalf(residual_node_->tensor)) return false;
    if (x.Shape().size() != 4 || mid.Shape().size() != 4 || out.Shape().size() != 4) return false;
    SiConv2dDesc d0 = pw.MakeDesc(x, mid), d1 = MakeDesc(mid, out);
    d0.in_ld = d0.ic; d0.out_ld = d0.oc; d1.in_ld = d1.ic; d1.out_ld = d1.oc;   // (dense views: strides are checked again at the launch)
    d1.has_residual = residual_node_ ? 1 : 0;
    d1.res_ld = d1.oc;
    // 2: the pair runs fused under the plan the 3x3 conv would take alone, on a grid that covers the chip (where the fused form pays)
    return si_hip_conv2d_pw_slab_f16_supported(&d0, &d1) == 2;
}

void Conv2d::SetPointwiseProducer(Conv2d* pw) {
    pw_producer_ = pw;
    pw_mid_ = input_tensor_nodes_.empty() ? nullptr : input_tensor_nodes_[0];
    if (pw) SetInputNodes(pw->InputNodes());
    device_ready_ = false;
}

void Conv2d::SetSibling(Conv2d* other) {
    sibling_ = other;
    device_ready_ = false;
}

Status Conv2d::Deinit() {
    weight_dev_.Free();
    bias_dev_.Free();
    device_ready_ = false;
    return Status::kSuccess;
}

Status Conv2d::Validate() {
    CHECK_STATUS(Layer::Validate());
    CHECK_STATUS(ValidateShape(1, 1));
    if (Status::kSuccess != ValidateFloat()) {
        LOG(ERROR) << "Conv2d::Validate fail [unsupport input/output data type]";
        return Status::kUnsupport;
    }
    return Status::kSuccess;
}

SiConv2dDesc Conv2d::MakeDesc(const Tensor& input, const Tensor& output) const {
    SiConv2dDesc d;
    memset(&d, 0, sizeof(d));
    Dims4 in, out;
    GetDims4(input, in);
    GetDims4(output, out);
    d.n = in.n; d.ih = in.h; d.iw = in.w; d.ic = in.c; d.in_ld = input.PixelStride();
    d.oh = out.h; d.ow = out.w; d.oc = out.c; d.out_ld = output.PixelStride();
    d.kh = kernel_h_; d.kw = kernel_w_; d.sh = stride_h_; d.sw = stride_w_;
    d.dh = dilation_h_; d.dw = dilation_w_; d.pt = padding_t_; d.pl = padding_l_;
    d.groups = groups_;
    d.has_bias = use_bias_ ? 1 : 0;
    d.act1 = act1_; d.act2 = act2_; d.act_param = act_param_;
    d.plan = has_plan_ ? &plan_ : nullptr;
    return d;
}

// mode 0: fp32 kernels; 1: fp16 implicit GEMM (fp16 weights); 2: fp16 stem kernel, fp32 image in / fp16 out (fp16 B fragments)
Status Conv2d::PrepareDevice(int mode) {
    if (device_ready_ && mode == prepared_mode_) return Status::kSuccess;
    device_ready_ = false;
    prepared_mode_ = mode;
    CHECK_BOOL(groups_ > 0 && in_channels_ > 0 && out_channels_ > 0 && kernel_h_ > 0 && kernel_w_ > 0);
    CHECK_BOOL(in_channels_ % groups_ == 0 && out_channels_ % groups_ == 0);
    const size_t expect = (size_t)out_channels_ * (in_channels_ / groups_) * kernel_h_ * kernel_w_;
    if (weight_.size() != expect) {
        LOG(ERROR) << "Conv2d: weight has " << weight_.size() << " elements, expected " << expect;
        return Status::kErrorShape;
    }
    if (use_bias_ && bias_.size() != (size_t)out_channels_) return Status::kErrorShape;

    // the weight layout is chosen from the layer's static shape (channels, kernel, stride, dilation, groups), so the
    // descriptor used for packing must carry all of those, exactly as the launch descriptor will
    SiConv2dDesc d;
    memset(&d, 0, sizeof(d));
    d.ic = in_channels_; d.oc = out_channels_; d.kh = kernel_h_; d.kw = kernel_w_; d.groups = groups_;
    d.sh = stride_h_; d.sw = stride_w_; d.dh = dilation_h_; d.dw = dilation_w_; d.pt = padding_t_; d.pl = padding_l_;
    if (mode == 1) return PrepareDeviceHalf(d);
    if (mode == 3 && si_hip_conv2d_f16_supported(&d) != 3) {
        LOG(ERROR) << "Conv2d: no fp16 depthwise kernel for " << in_channels_ << " channels (needs a multiple of 8)";
        return Status::kUnsupport;
    }   // (mode 3 then packs the fp32 depthwise image below, exactly as mode 0 does)
    if (mode == 4) {
        wino_tile_ = 0;
        use_winograd_ = false;
        // (sibling fusion: ONE conv over the concatenated filters -- OIHW: the sibling's rows follow this layer's -- as PrepareDeviceHalf)
        std::vector<float> w_all = weight_, bias_all = bias_;
        if (sibling_) {
            CHECK_BOOL(sibling_->weight_.size() == (size_t)sibling_->out_channels_ * in_channels_);
            d.oc = out_channels_ + sibling_->out_channels_;
            w_all.insert(w_all.end(), sibling_->weight_.begin(), sibling_->weight_.end());
            if (use_bias_) bias_all.insert(bias_all.end(), sibling_->bias_.begin(), sibling_->bias_.end());
        }
        std::vector<uint16_t> packed(si_hip_conv2d_split3_weight_elems(&d));
        CHECK_BOOL(!packed.empty());
        if (si_hip_conv2d_split3_pack_weight_host(&d, w_all.data(), packed.data()) == SI_E_UNSUPPORTED) {
            // a weight outside fp16's range (or not finite): this layer cannot be split -- the true-fp32 kernels, decided here, at load
            LOG(WARNING) << "f32_split: conv " << in_channels_ << " -> " << out_channels_ << " has a weight outside fp16's range; the layer stays on the fp32 kernels";
            DemoteSplit();
            return Status::kUnsupport;
        }
        CHECK_STATUS(CheckHip(si_hip_conv2d_split3_pack_weight_host(&d, w_all.data(), packed.data()), "split weights (hi / lo halves)"));
        CHECK_STATUS(CheckHip(weight_dev_.Upload(packed.data(), packed.size() * sizeof(uint16_t)), "upload weight"));
        if (use_bias_) CHECK_STATUS(CheckHip(bias_dev_.Upload(bias_all.data(), bias_all.size() * sizeof(float)), "upload bias"));
        device_ready_ = true;
        return Status::kSuccess;
    }
    if (mode == 5) {
        wino_tile_ = 0;
        use_winograd_ = false;
        std::vector<uint16_t> packed(si_hip_conv2d_wino23_split_weight_elems(&d));
        CHECK_BOOL(!packed.empty());
        if (si_hip_conv2d_wino23_split_pack_weight_host(&d, weight_.data(), packed.data()) == SI_E_UNSUPPORTED) {
            LOG(WARNING) << "f32_split: conv " << in_channels_ << " -> " << out_channels_ << " has a transformed filter value outside fp16's range; the layer stays on the fp32 kernels";
            DemoteSplit();
            return Status::kUnsupport;
        }
        CHECK_STATUS(CheckHip(si_hip_conv2d_wino23_split_pack_weight_host(&d, weight_.data(), packed.data()), "winograd filter transform (hi / lo halves)"));
        CHECK_STATUS(CheckHip(weight_dev_.Upload(packed.data(), packed.size() * sizeof(uint16_t)), "upload weight"));
        if (use_bias_) CHECK_STATUS(CheckHip(bias_dev_.Upload(bias_.data(), bias_.size() * sizeof(float)), "upload bias"));
        device_ready_ = true;
        return Status::kSuccess;
    }
    if (mode == 6) {
        wino_tile_ = 0;
        use_winograd_ = false;
        std::vector<uint16_t> packed(si_hip_conv2d_stem_split3_weight_elems(&d));
        CHECK_BOOL(!packed.empty());
        if (si_hip_conv2d_stem_split3_pack_weight_host(&d, weight_.data(), packed.data()) == SI_E_UNSUPPORTED) {
            LOG(WARNING) << "f32_split: stem conv " << in_channels_ << " -> " << out_channels_ << " has a weight outside fp16's range; the layer stays on the fp32 kernels";
            DemoteSplit();
            return Status::kUnsupport;
        }
        CHECK_STATUS(CheckHip(si_hip_conv2d_stem_split3_pack_weight_host(&d, weight_.data(), packed.data()), "split stem weights (hi / lo halves)"));
        CHECK_STATUS(CheckHip(weight_dev_.Upload(packed.data(), packed.size() * sizeof(uint16_t)), "upload weight"));
        if (use_bias_) CHECK_STATUS(CheckHip(bias_dev_.Upload(bias_.data(), bias_.size() * sizeof(float)), "upload bias"));
        device_ready_ = true;
        return Status::kSuccess;
    }
    if (mode == 2) {
        wino_tile_ = 0;
        use_winograd_ = false;
        std::vector<uint16_t> packed(si_hip_conv2d_stem_f16_weight_elems(&d));
        CHECK_BOOL(!packed.empty());
        CHECK_STATUS(CheckHip(si_hip_conv2d_stem_f16_pack_weight_host(&d, weight_.data(), packed.data()), "pack stem weight"));
        CHECK_STATUS(CheckHip(weight_dev_.Upload(packed.data(), packed.size() * sizeof(uint16_t)), "upload weight"));
        if (use_bias_) CHECK_STATUS(CheckHip(bias_dev_.Upload(bias_.data(), bias_.size() * sizeof(float)), "upload bias"));
        device_ready_ = true;
        return Status::kSuccess;
    }
    wino_tile_ = WinogradTile(d);
    use_winograd_ = wino_tile_ != 0;
    if ((algo_ == Algo::kWinograd23 || algo_ == Algo::kWinograd43) && !use_winograd_) {
        LOG(ERROR) << "Conv2d: Winograd requested for a shape it does not support";
        return Status::kUnsupport;
    }
    std::vector<float> packed(wino_tile_ == 4   ? si_hip_conv2d_wino43_weight_elems(&d)
                              : wino_tile_ == 2 ? si_hip_conv2d_wino23_weight_elems(&d)
                                                : si_hip_conv2d_weight_elems(&d));
    if (wino_tile_ == 4) {
        CHECK_STATUS(CheckHip(si_hip_conv2d_wino43_pack_weight_host(&d, weight_.data(), packed.data()), "winograd filter transform"));
    } else if (wino_tile_ == 2) {
        CHECK_STATUS(CheckHip(si_hip_conv2d_wino23_pack_weight_host(&d, weight_.data(), packed.data()), "winograd filter transform"));
    } else {
        CHECK_STATUS(CheckHip(si_hip_conv2d_pack_weight_host(&d, weight_.data(), packed.data()), "pack weight"));
    }
    std::vector<float> bias_all = bias_;
    if (sibling_) {
        // [oc][K] layouts with the same K: the fused weight is the two images one after the other
        SiConv2dDesc ds = d;
        ds.oc = sibling_->out_channels_;
        CHECK_BOOL(sibling_->weight_.size() == (size_t)ds.oc * in_channels_);
        std::vector<float> packed2(si_hip_conv2d_weight_elems(&ds));
        CHECK_STATUS(CheckHip(si_hip_conv2d_pack_weight_host(&ds, sibling_->weight_.data(), packed2.data()), "pack sibling weight"));
        packed.insert(packed.end(), packed2.begin(), packed2.end());
        if (use_bias_) bias_all.insert(bias_all.end(), sibling_->bias_.begin(), sibling_->bias_.end());
    }
    CHECK_STATUS(CheckHip(weight_dev_.Upload(packed.data(), packed.size() * sizeof(float)), "upload weight"));
    if (use_bias_) CHECK_STATUS(CheckHip(bias_dev_.Upload(bias_all.data(), bias_all.size() * sizeof(float)), "upload bias"));
    device_ready_ = true;
    return Status::kSuccess;
}

Status Conv2d::PrepareDeviceHalf(const SiConv2dDesc& d) {
    wino_tile_ = 0;
    use_winograd_ = false;
    if (si_hip_conv2d_f16_supported(&d) != 1) {
        LOG(ERROR) << "Conv2d: no fp16 kernel for " << in_channels_ << " -> " << out_channels_ << " channels, groups " << groups_
                   << " (needs ic/groups % 32 == 0)";
        return Status::kUnsupport;
    }
    // sibling fusion: ONE conv over the concatenated filters (OIHW: the sibling's rows follow this layer's), packed as a whole --
    // the packed buffer holds two images (rows, then MFMA lane order, include/si_hip.h), so two packed buffers cannot be appended
    SiConv2dDesc df = d;
    std::vector<float> w_all = weight_;
    std::vector<float> bias_all = bias_;
    if (sibling_) {
        CHECK_BOOL(sibling_->weight_.size() == (size_t)sibling_->out_channels_ * in_channels_);
        df.oc = d.oc + sibling_->out_channels_;
        w_all.insert(w_all.end(), sibling_->weight_.begin(), sibling_->weight_.end());
        if (use_bias_) bias_all.insert(bias_all.end(), sibling_->bias_.begin(), sibling_->bias_.end());
    }
    std::vector<uint16_t> packed(si_hip_conv2d_f16_weight_elems(&df));
    CHECK_BOOL(!packed.empty());
    CHECK_STATUS(CheckHip(si_hip_conv2d_f16_pack_weight_host(&df, w_all.data(), packed.data()), "pack fp16 weight"));
    CHECK_STATUS(CheckHip(weight_dev_.Upload(packed.data(), packed.size() * sizeof(uint16_t)), "upload weight"));
    if (use_bias_) CHECK_STATUS(CheckHip(bias_dev_.Upload(bias_all.data(), bias_all.size() * sizeof(float)), "upload bias"));
    device_ready_ = true;
    return Status::kSuccess;
}

// which kernel family serves this (input, output) storage pair; a non-stem conv fed an fp32 tensor inside an fp16 graph
// converts its input first (in_half_)
// (4: fp32 tensors, the contraction on the fp16 matrix cores by operand splitting -- engine option f32_split, opt-in)
// Detect levels under f32_split: from this channel count on (tools/split3_check.py --detect; profiles/r05_f32_split_detect.txt)
static const int kSplit3DetectMinChannels = 128;

bool Conv2d::UseSplit3() const {
    if (!f32_split_ || stem_producer_ || groups_ != 1 || dilation_h_ != 1 || dilation_w_ != 1) return false;
    // (round 6: the dual-source form -- upsample + concat read at the source -- exists on the split kernel too, for 64-channel granularity)
    if (up_node_ && (f32_split_level_ < 3 || residual_node_ || up_c0_ % 64 != 0 || in_channels_ % 64 != 0)) return false;
    if (up_node_) {
        Dims4 lo;
        if (!GetDims4(up_node_->tensor, lo) || lo.c % 64 != 0) return false;
    }
    // (round 6: a sibling-fused conv -- C3's cv1 | cv2 -- is ONE conv over the concatenated filters here too: si_hip_conv2d_split3_split_f32)
    if (sibling_ && (f32_split_level_ < 2 || residual_node_ || out_channels_ % 32 != 0)) return false;
    const int oc_all = out_channels_ + (sibling_ ? sibling_->out_channels_ : 0);
    SiConv2dDesc d;
    memset(&d, 0, sizeof(d));
    d.ic = in_channels_; d.oc = oc_all; d.kh = kernel_h_; d.kw = kernel_w_; d.groups = groups_;
    d.sh = stride_h_; d.sw = stride_w_; d.dh = dilation_h_; d.dw = dilation_w_; d.pt = padding_t_; d.pl = padding_l_;
    d.in_ld = in_channels_;
    if (!si_hip_conv2d_split3_supported(&d)) return false;
    // Where it pays (YOLOv5s batch 32, per layer, profiles/r05_f32_split.txt): K >= 512 and >= 128 output channels -- the 3x3 stride-2
    // layers (1.4-2.1x), the wide 1x1 layers (1.3-1.8x) -- and the strided spatial convs from K = 288 and 64 output channels (YOLOv5s
    // conv_1, 320x320x32 -> 64: 1.34x on the 128 x 64 tile of 2 x 2 waves over 32-channel K-tiles).  Thin-K 1x1 layers are HBM-bound
    // (the fp32 -> hi / lo conversion only costs: 0.54-0.62x at 64 output channels), and the 3x3 stride-1 layers below 256 channels are
    // faster on the fused Winograd kernel (2.25x fewer multiplies).  A property of the layer, not of the batch: an image's bits do not
    // depend on it.
    const long long K = (long long)kernel_h_ * kernel_w_ * in_channels_;
    const bool strided_spatial = kernel_h_ * kernel_w_ > 1 && (stride_h_ > 1 || stride_w_ > 1);
    // round 6: ... and the wide 1x1 layers from K = 256 with >= 256 output columns (C3's cv1 | cv2 and cv3 over 256
    // channels at 40x40: 1.29x standalone in round 5, left on the fp32 template then because the policy asked for K >= 512)
    // (thin 1x1 layers, K = 128 over 128 columns, measured flat at batch 32 and -3 % at batch 4 on the split kernel: profiles/r06_f32_split.txt)
    const bool wide_pw = f32_split_level_ >= 2 && kernel_h_ * kernel_w_ == 1 && K >= 256 && (oc_all >= 256 || (up_node_ && oc_all >= 128));
    if (strided_spatial ? (K < 288 || oc_all < 64) : ((K < 512 || oc_all < 128) && !wide_pw)) return false;
    const bool wino_shape = kernel_h_ == 3 && kernel_w_ == 3 && stride_h_ == 1 && stride_w_ == 1;
    return !(wino_shape && in_channels_ < 256 && algo_ != Algo::kImplicitGemm);
}

// (5: f32_split on a layer the fused Winograd F(2,3) kernel serves -- the same kernel around a channel loop on the fp16 matrix cores,
// conv_wino23_split.hip)
bool Conv2d::UseWinoSplit() const {
    if (!f32_split_ || sibling_ || up_node_ || stem_producer_ || groups_ != 1) return false;
    SiConv2dDesc d;
    memset(&d, 0, sizeof(d));
    d.ic = in_channels_; d.oc = out_channels_; d.kh = kernel_h_; d.kw = kernel_w_; d.groups = groups_;
    d.sh = stride_h_; d.sw = stride_w_; d.dh = dilation_h_; d.dw = dilation_w_; d.pt = padding_t_; d.pl = padding_l_;
    return WinogradTile(d) == 2 && si_hip_conv2d_wino23_split_supported(&d) != 0;
}

// (6: f32_split on the RGB stem -- the fp16 stem kernel's staging and MFMA loop on operands split hi + 2^-11 lo, fp32 out: conv_stem_f16.hip)
bool Conv2d::UseStemSplit() const {
    if (!f32_split_ || f32_split_level_ < 4 || sibling_ || up_node_ || stem_producer_ || residual_node_) return false;
    SiConv2dDesc d;
    memset(&d, 0, sizeof(d));
    d.ic = in_channels_; d.oc = out_channels_; d.kh = kernel_h_; d.kw = kernel_w_; d.groups = groups_;
    d.sh = stride_h_; d.sw = stride_w_; d.dh = dilation_h_; d.dw = dilation_w_; d.pt = padding_t_; d.pl = padding_l_;
    return si_hip_conv2d_f16_supported(&d) == 2;
}

int Conv2d::PrecisionMode(const Tensor& input, const Tensor& output) const {
    if (!IsHalf(input) && !IsHalf(output)) return UseSplit3() ? 4 : (UseWinoSplit() ? 5 : (UseStemSplit() ? 6 : 0));
    if (groups_ > 1 && groups_ == in_channels_ && in_channels_ == out_channels_ && IsHalf(input) && IsHalf(output)) return 3;   // depthwise, fp16 storage
    if (!IsHalf(input)) {
        SiConv2dDesc d;
        memset(&d, 0, sizeof(d));
        d.ic = in_channels_; d.oc = out_channels_; d.kh = kernel_h_; d.kw = kernel_w_; d.groups = groups_;
        d.sh = stride_h_; d.sw = stride_w_; d.dh = dilation_h_; d.dw = dilation_w_; d.pt = padding_t_; d.pl = padding_l_;
        if (si_hip_conv2d_f16_supported(&d) == 2 && !residual_node_ && !sibling_) return 2;
    }
    return 1;
}

Status Conv2d::HalfInput(const Tensor& input, Tensor& half) {
    if (IsHalf(input)) {
        half = input;
        return Status::kSuccess;
    }
    size_t pixels = 0;
    int c = 0;
    if (!GetPixelsChannels(input, pixels, c)) return Status::kErrorShape;
    CHECK_STATUS(in_half_.Allocate(DataType::kFloat16, input.Shape()));
    CHECK_STATUS(CheckHip(si_hip_convert_f32_f16(input.Data<float>(), pixels, c, input.PixelStride(), in_half_.RawData(), c, Stream()),
                          "conv2d input fp32 -> fp16"));
    half = in_half_;
    return Status::kSuccess;
}

Status Conv2d::Forward(const Tensor& input, std::vector<Tensor>& outputs) {
    if (!sibling_ || outputs.size() != 2) return Status::kUnsupport;
    return RunOnDevice({&input}, {&outputs[0], &outputs[1]}, [this](const std::vector<Tensor>& in, std::vector<Tensor>& out) {
        if (stem_pair_) return LaunchStemTriple(in[0], out[0], &out[1]);
        int mode = PrecisionMode(in[0], out[0]);
        if (IsHalf(out[0]) != IsHalf(out[1])) return Status::kUnsupport;
        {
            Status st = PrepareDevice(mode);
            if (Status::kUnsupport == st && mode == 4 && !f32_split_) {   // (a weight out of fp16's range: see PrepareDevice)
                mode = PrecisionMode(in[0], out[0]);
                st = PrepareDevice(mode);
            }
            CHECK_STATUS(st);
        }
        Dims4 di, d0, d1;
        if (!GetDims4(in[0], di) || !GetDims4(out[0], d0) || !GetDims4(out[1], d1)) return Status::kErrorShape;
        if (di.c != in_channels_ || d0.c != out_channels_ || d1.c != sibling_->out_channels_ || d0.pixels() != d1.pixels()) return Status::kErrorShape;
        SiConv2dDesc d = MakeDesc(in[0], out[0]);
        d.oc = out_channels_ + sibling_->out_channels_;
        if (mode == 4) {
            d.range_flag = range_flag_;
            int rc;
            if (up_node_) {
                SiConv2dUpsampledSource up;
                CHECK_STATUS(MakeUpsampledSource(up));
                rc = si_hip_conv2d_split3_upcat_f32(&d, in[0].Data<float>(), &up, weight_dev_.As<void>(), use_bias_ ? bias_dev_.As<float>() : nullptr,
                                                    out[0].Data<float>(), out_channels_, out[1].Data<float>(), out[1].PixelStride(), Stream());
            } else
            rc = si_hip_conv2d_split3_split_f32(&d, in[0].Data<float>(), weight_dev_.As<void>(), use_bias_ ? bias_dev_.As<float>() : nullptr,
                                                          out[0].Data<float>(), out_channels_, out[1].Data<float>(), out[1].PixelStride(), Stream());
            if (rc != SI_E_UNSUPPORTED) return CheckHip(rc, "conv2d (fused siblings, fp32 by three fp16 products)");
            DemoteSplit();   // an unaligned / oversized view: the true-fp32 kernels from here on
            d.range_flag = nullptr;
            mode = 0;
            CHECK_STATUS(PrepareDevice(0));
        }
        if (mode == 1 && up_node_) {
            if (!IsHalf(in[0])) return Status::kUnsupport;
            SiConv2dUpsampledSource up;
            CHECK_STATUS(MakeUpsampledSource(up));
            return CheckHip(si_hip_conv2d_upcat_f16(&d, in[0].RawData(), &up, weight_dev_.As<void>(), use_bias_ ? bias_dev_.As<float>() : nullptr,
                                                    out[0].RawData(), out_channels_, out[1].RawData(), out[1].PixelStride(), Stream()),
                            "conv2d fp16 (fused siblings, upsampled source)");
        }
        if (mode == 1) {
            Tensor xin;
            CHECK_STATUS(HalfInput(in[0], xin));
            d.in_ld = xin.PixelStride();
            return CheckHip(si_hip_conv2d_split_f16(&d, xin.RawData(), weight_dev_.As<void>(),
                                                    use_bias_ ? bias_dev_.As<float>() : nullptr, out[0].RawData(), out_channels_,
                                                    out[1].RawData(), out[1].PixelStride(), Stream()),
                            "conv2d fp16 (fused siblings)");
        }
        if (up_node_) {
            if (mode != 0) return Status::kUnsupport;
            SiConv2dUpsampledSource up;
            CHECK_STATUS(MakeUpsampledSource(up));
            return CheckHip(si_hip_conv2d_upcat_f32(&d, in[0].Data<float>(), &up, weight_dev_.As<float>(), use_bias_ ? bias_dev_.As<float>() : nullptr,
                                                    out[0].Data<float>(), out_channels_, out[1].Data<float>(), out[1].PixelStride(), Stream()),
                            "conv2d (fused siblings, upsampled source)");
        }
        return CheckHip(si_hip_conv2d_split_f32(&d, in[0].Data<float>(), weight_dev_.As<float>(),
                                                use_bias_ ? bias_dev_.As<float>() : nullptr, out[0].Data<float>(), out_channels_,
                                                out[1].Data<float>(), out[1].PixelStride(), Stream()),
                        "conv2d (fused siblings)");
    });
}

Status Conv2d::Launch(const Tensor& input, const Tensor* residual, Tensor& output) {
    if (stem_pair_) return residual ? Status::kUnsupport : LaunchStemTriple(input, output, nullptr);
    if (cv3_pair_) return residual ? Status::kUnsupport : LaunchCv3(input, output);
    if (stem_producer_) {
        // the stem conv and this one in one launch: `input` is the fp32 image
        if (residual || !stem_mid_ || IsHalf(input) || !IsHalf(output)) return Status::kUnsupport;
        CHECK_STATUS(PrepareDevice(1));
        CHECK_STATUS(stem_producer_->PrepareDevice(2));
        SiConv2dDesc d0 = stem_producer_->MakeDesc(input, stem_mid_->tensor), d1 = MakeDesc(stem_mid_->tensor, output);
        d0.out_ld = d0.oc; d1.in_ld = d1.ic;
        return CheckHip(si_hip_conv2d_stem_s2c32_f16(&d0, &d1, input.Data<float>(), stem_producer_->weight_dev_.As<void>(),
                                                     stem_producer_->use_bias_ ? stem_producer_->bias_dev_.As<float>() : nullptr,
                                                     weight_dev_.As<void>(), use_bias_ ? bias_dev_.As<float>() : nullptr, output.RawData(),
                                                     Stream()),
                        "conv2d stem + 3x3 s2 (fp16, one launch)");
    }
    if (pw_producer_) {
        // the bottleneck's 1x1 conv and this 3x3 one in one launch: `input` is the 1x1 conv's input
        if (!pw_mid_ || !IsHalf(input) || !IsHalf(output) || (residual && !IsHalf(*residual))) return Status::kUnsupport;
        CHECK_STATUS(PrepareDevice(1));
        CHECK_STATUS(pw_producer_->PrepareDevice(1));
        SiConv2dDesc d0 = pw_producer_->MakeDesc(input, pw_mid_->tensor), d1 = MakeDesc(pw_mid_->tensor, output);
        d0.out_ld = d0.oc; d1.in_ld = d1.ic;
        if (residual) {
            d1.has_residual = 1;
            d1.res_ld = residual->PixelStride();
        }
        const int rc = si_hip_conv2d_pw_slab_f16(&d0, &d1, input.RawData(), pw_producer_->weight_dev_.As<void>(),
                                                 pw_producer_->use_bias_ ? pw_producer_->bias_dev_.As<float>() : nullptr, weight_dev_.As<void>(),
                                                 use_bias_ ? bias_dev_.As<float>() : nullptr, residual ? residual->RawData() : nullptr,
                                                 output.RawData(), Stream());
        if (rc != SI_E_UNSUPPORTED) return CheckHip(rc, "conv2d 1x1 + 3x3 (fp16, one launch)");
        // The pair was fused on dense strides, before the concat aliasing gave the tensors their real views (CanFusePointwiseProducer); the
        // launch re-checks the views it gets (16-byte aligned rows, strides in whole 16-byte vectors) and may refuse a channel slice at an odd
        // offset.  The fused-away intermediate is never planned, so the two ordinary launches run through a private one (this path
        // returned kUnsupport until round 6; YOLOv5s never takes it -- all its strides are multiples of 8).
        if (!pw_scratch_.RawData()) CHECK_STATUS(pw_scratch_.Allocate(DataType::kFloat16, pw_mid_->tensor.Shape()));
        Conv2d* const pw = pw_producer_;
        CHECK_STATUS(pw->Launch(input, nullptr, pw_scratch_));
        pw_producer_ = nullptr;
        const Status st = Launch(pw_scratch_, residual, output);
        pw_producer_ = pw;
        return st;
    }
    int mode = PrecisionMode(input, output);
    {
        Status st = PrepareDevice(mode);
        if (Status::kUnsupport == st && (mode == 4 || mode == 5 || mode == 6) && !f32_split_) {   // (a weight out of fp16's range: see PrepareDevice)
            mode = PrecisionMode(input, output);
            st = PrepareDevice(mode);
        }
        CHECK_STATUS(st);
    }
    Dims4 in, out;
    if (!GetDims4(input, in) || !GetDims4(output, out)) return Status::kErrorShape;
    if (in.c != in_channels_ || out.c != out_channels_ || in.n != out.n) {
        LOG(ERROR) << "Conv2d: tensor channels/batch do not match the layer";
        return Status::kErrorShape;
    }
    SiConv2dDesc d = MakeDesc(input, output);
    if (residual) {
        d.has_residual = 1;
        d.res_ld = residual->PixelStride();
    }
    if (mode == 4) {
        d.range_flag = range_flag_;
        int rc;
        if (up_node_) {
            if (residual) return Status::kUnsupport;
            SiConv2dUpsampledSource up;
            CHECK_STATUS(MakeUpsampledSource(up));
            rc = si_hip_conv2d_split3_upcat_f32(&d, input.Data<float>(), &up, weight_dev_.As<void>(), use_bias_ ? bias_dev_.As<float>() : nullptr,
                                                output.Data<float>(), 0, nullptr, 0, Stream());
        } else
        rc = si_hip_conv2d_split3_f32(&d, input.Data<float>(), weight_dev_.As<void>(), use_bias_ ? bias_dev_.As<float>() : nullptr,
                                                residual ? residual->Data<float>() : nullptr, output.Data<float>(), Stream());
        if (rc != SI_E_UNSUPPORTED) return CheckHip(rc, "conv2d (fp32 by three fp16 products)");
        // an unaligned / oversized view: the true-fp32 kernels from here on.  (The re-pack uploads weights: inside a hipGraph capture that
        // fails, and EngineImpl::ForwardAsync then discards the capture and runs the step eagerly.)
        DemoteSplit();
        d.range_flag = nullptr;
        return Launch(input, residual, output);
    }
    if (mode == 5) {
        d.range_flag = range_flag_;
        const int rc = si_hip_conv2d_wino23_split_f32(&d, input.Data<float>(), weight_dev_.As<void>(), use_bias_ ? bias_dev_.As<float>() : nullptr,
                                                      residual ? residual->Data<float>() : nullptr, output.Data<float>(), Stream());
        if (rc != SI_E_UNSUPPORTED) return CheckHip(rc, "conv2d (winograd, fp32 by three fp16 products)");
        DemoteSplit();   // an unaligned / oversized view: the true-fp32 kernels from here on
        return Launch(input, residual, output);
    }
    if (mode == 6) {
        d.range_flag = range_flag_;
        const int rc = si_hip_conv2d_stem_split3_f32(&d, input.Data<float>(), weight_dev_.As<void>(), use_bias_ ? bias_dev_.As<float>() : nullptr,
                                                     output.Data<float>(), Stream());
        if (rc != SI_E_UNSUPPORTED) return CheckHip(rc, "conv2d stem (fp32 by three fp16 products)");
        DemoteSplit();   // a strided / unaligned image view: the true-fp32 stem kernel from here on
        return Launch(input, residual, output);
    }
    if (mode == 2)
        return CheckHip(si_hip_conv2d_stem_f16(&d, input.Data<float>(), weight_dev_.As<void>(),
                                               use_bias_ ? bias_dev_.As<float>() : nullptr, output.RawData(), Stream()),
                        "conv2d stem (fp16 out)");
    if (mode == 3) {
        if (residual && !IsHalf(*residual)) return Status::kUnsupport;
        return CheckHip(si_hip_conv2d_depthwise_f16(&d, input.RawData(), weight_dev_.As<float>(), use_bias_ ? bias_dev_.As<float>() : nullptr,
                                                    residual ? residual->RawData() : nullptr, output.RawData(), Stream()),
                        "conv2d depthwise fp16");
    }
    if (mode == 1 && up_node_) {
        if (residual || !IsHalf(input) || !IsHalf(output)) return Status::kUnsupport;
        SiConv2dUpsampledSource up;
        CHECK_STATUS(MakeUpsampledSource(up));
        return CheckHip(si_hip_conv2d_upcat_f16(&d, input.RawData(), &up, weight_dev_.As<void>(), use_bias_ ? bias_dev_.As<float>() : nullptr,
                                                output.RawData(), 0, nullptr, 0, Stream()),
                        "conv2d fp16 (upsampled source)");
    }
    if (mode == 1) {
        if (residual && !IsHalf(*residual)) return Status::kUnsupport;
        Tensor xin;
        CHECK_STATUS(HalfInput(input, xin));
        d.in_ld = xin.PixelStride();
        return CheckHip(si_hip_conv2d_f16(&d, xin.RawData(), weight_dev_.As<void>(), use_bias_ ? bias_dev_.As<float>() : nullptr,
                                          residual ? residual->RawData() : nullptr, output.RawData(), IsHalf(output) ? 0 : 1,
                                          Stream()),
                        "conv2d fp16");
    }
    if (use_winograd_) {
        const auto fn = wino_tile_ == 4 ? si_hip_conv2d_wino43_f32 : si_hip_conv2d_wino23_f32;
        const int rc = fn(&d, input.Data<float>(), weight_dev_.As<float>(), use_bias_ ? bias_dev_.As<float>() : nullptr,
                          residual ? residual->Data<float>() : nullptr, output.Data<float>(), Stream());
        if (rc != SI_E_UNSUPPORTED) return CheckHip(rc, "conv2d (winograd)");
        // an unaligned / oversized tensor view: re-pack for the implicit-GEMM kernel once and stay there
        algo_ = Algo::kImplicitGemm;
        device_ready_ = false;
        CHECK_STATUS(PrepareDevice(0));
    }
    if (up_node_) {
        if (residual) return Status::kUnsupport;
        SiConv2dUpsampledSource up;
        CHECK_STATUS(MakeUpsampledSource(up));
        return CheckHip(si_hip_conv2d_upcat_f32(&d, input.Data<float>(), &up, weight_dev_.As<float>(), use_bias_ ? bias_dev_.As<float>() : nullptr,
                                                output.Data<float>(), 0, nullptr, 0, Stream()),
                        "conv2d (upsampled source)");
    }
    return CheckHip(si_hip_conv2d_f32(&d, input.Data<float>(), weight_dev_.As<float>(),
                                      use_bias_ ? bias_dev_.As<float>() : nullptr,
                                      residual ? residual->Data<float>() : nullptr, output.Data<float>(), Stream()),
                    "conv2d");
}

bool Conv2d::CanReadUpsampled(int c0, int c) const {
    return kernel_h_ == 1 && kernel_w_ == 1 && stride_h_ == 1 && stride_w_ == 1 && padding_t_ == 0 && padding_b_ == 0 && padding_l_ == 0 &&
           padding_r_ == 0 && groups_ == 1 && in_channels_ % 32 == 0 && c0 % 32 == 0 && c % 32 == 0 && c0 + c <= in_channels_ &&
           residual_node_ == nullptr && algo_ != Algo::kWinograd23 && algo_ != Algo::kWinograd43;
}

// ... and the kernel's own preconditions on THIS problem (tensor sizes below 4 GiB, strides), asked with the shapes the bound
// nodes have: once the engine drops the upsample launch there is no unfused schedule to fall back to at Forward() time
bool Conv2d::CanReadUpsampledFrom(const TensorNode* low, int c0, float scale_h, float scale_w) const {
    Dims4 lo, di, dout;
    if (!low || input_tensor_nodes_.size() != 1 || output_tensor_nodes_.empty()) return false;
    if (!GetDims4(low->tensor, lo) || !GetDims4(input_tensor_nodes_[0]->tensor, di) || !GetDims4(output_tensor_nodes_[0]->tensor, dout)) return false;
    const bool half = IsHalf(low->tensor);
    // fp16 storage (round 4): both sources and the output half (a graph-output conv writes fp32 through the plain kernel only)
    if (half != IsHalf(input_tensor_nodes_[0]->tensor) || (half && !IsHalf(output_tensor_nodes_[0]->tensor))) return false;
    if (!CanReadUpsampled(c0, lo.c) || scale_h <= 0.f || scale_w <= 0.f) return false;
    SiConv2dDesc d = MakeDesc(input_tensor_nodes_[0]->tensor, output_tensor_nodes_[0]->tensor);
    d.in_ld = di.c;   // (the concat buffer is dense: the conv's input IS the concat output)
    if (sibling_) d.oc = out_channels_ + sibling_->out_channels_;
    SiConv2dUpsampledSource up;
    memset(&up, 0, sizeof(up));
    up.ih = lo.h; up.iw = lo.w; up.c = lo.c; up.ld = lo.c; up.c0 = c0;
    up.inv_scale_h = 1.0f / scale_h; up.inv_scale_w = 1.0f / scale_w;
    if (half) {
        for (const TensorNode* o : output_tensor_nodes_)
            if (!IsHalf(o->tensor)) return false;
        return si_hip_conv2d_upcat_f16_supported(&d, &up) == 1;
    }
    return si_hip_conv2d_upcat_supported(&d, &up) == 1;
}

void Conv2d::SetUpsampledSource(TensorNode* low, int c0, float scale_h, float scale_w) {
    up_node_ = low;
    up_c0_ = c0;
    up_scale_h_ = scale_h;
    up_scale_w_ = scale_w;
}

Status Conv2d::MakeUpsampledSource(SiConv2dUpsampledSource& up) const {
    Dims4 lo;
    if (!up_node_ || !GetDims4(up_node_->tensor, lo)) return Status::kUnsupport;
    up.src = static_cast<const float*>(up_node_->tensor.RawData());   // (half data for si_hip_conv2d_upcat_f16: include/si_hip.h)
    up.ih = lo.h; up.iw = lo.w; up.c = lo.c; up.ld = up_node_->tensor.PixelStride();
    up.c0 = up_c0_;
    up.inv_scale_h = 1.0f / up_scale_h_;   // as si_hip_upsample_nearest_f32 forms it (reference upsample.cpp:85-92)
    up.inv_scale_w = 1.0f / up_scale_w_;
    return Status::kSuccess;
}

// Detect-head variant: device tensors only (called by YoloDetect inside its own RunOnDevice scope)
Status Conv2d::ForwardYolo(const Tensor& input, const SiYoloLevel& level, const float* grid_dev, const float* anchor_dev,
                           Tensor& detect_out) {
    // (f32_split: the Detect levels take the three-fp16-products kernel too -- YOLOv5s batch 32: 147 -> 118, 64 -> 43, 33 -> 26 us per
    // level, +3 % on the network; profiles/r05_f32_split_detect.txt)
    int mode = IsHalf(input) ? 1 : ((f32_split_ && in_channels_ % 64 == 0 && in_channels_ >= kSplit3DetectMinChannels && out_channels_ > 64) ? 4 : 0);
    {
        Status st = PrepareDevice(mode);
        if (Status::kUnsupport == st && mode == 4 && !f32_split_) {   // (a weight out of fp16's range)
            mode = 0;
            st = PrepareDevice(mode);
        }
        CHECK_STATUS(st);
    }
    Dims4 in;
    if (!GetDims4(input, in) || in.c != in_channels_) return Status::kErrorShape;
    if (input.GetMemoryType() != MemoryType::kDevice || detect_out.GetMemoryType() != MemoryType::kDevice) return Status::kUnsupport;
    Tensor conv_out(DataType::kFloat32, {in.n, in.h, in.w, out_channels_}, MemoryType::kDevice, false);  // shape only
    SiConv2dDesc d = MakeDesc(input, conv_out);
    d.act1 = d.act2 = SI_ACT_NONE;
    if (mode == 4) {
        d.range_flag = range_flag_;
        const int rc = si_hip_conv2d_split3_yolo_f32(&d, input.Data<float>(), weight_dev_.As<void>(), use_bias_ ? bias_dev_.As<float>() : nullptr,
                                                     &level, grid_dev, anchor_dev, detect_out.Data<float>(), Stream());
        if (rc != SI_E_UNSUPPORTED) return CheckHip(rc, "conv2d+yolo (fp32 by three fp16 products)");
        DemoteSplit();   // an unaligned view: the true-fp32 kernel from here on
        return ForwardYolo(input, level, grid_dev, anchor_dev, detect_out);
    }
    if (mode == 1) {
        const int rc = si_hip_conv2d_yolo_f16(&d, input.RawData(), weight_dev_.As<void>(), use_bias_ ? bias_dev_.As<float>() : nullptr,
                                              &level, grid_dev, anchor_dev, detect_out.Data<float>(), Stream());
        if (rc == SI_E_UNSUPPORTED) return Status::kUnsupport;
        return CheckHip(rc, "conv2d+yolo fp16");
    }
    const int rc = si_hip_conv2d_yolo_f32(&d, input.Data<float>(), weight_dev_.As<float>(), use_bias_ ? bias_dev_.As<float>() : nullptr,
                                          &level, grid_dev, anchor_dev, detect_out.Data<float>(), Stream());
    if (rc == SI_E_UNSUPPORTED) return Status::kUnsupport;
    return CheckHip(rc, "conv2d+yolo");
}

Status Conv2d::Forward(const Tensor& input, Tensor& output) {
    std::vector<const Tensor*> ins{&input};
    if (residual_node_) ins.push_back(&residual_node_->tensor);
    return RunOnDevice(ins, {&output}, [this](const std::vector<Tensor>& in, std::vector<Tensor>& out) {
        return Launch(in[0], in.size() > 1 ? &in[1] : nullptr, out[0]);
    });
}

const char* Conv2d::KernelName() const {
    if (input_tensor_nodes_.empty() || output_tensor_nodes_.empty()) return "conv_igemm_f32";
    const Tensor& in = input_tensor_nodes_[0]->tensor;
    const Tensor& out = output_tensor_nodes_[0]->tensor;
    if (in.Shape().size() != 4 || out.Shape().size() != 4) return "conv_igemm_f32";
    if (stem_pair_) return "conv_stem_s2c32_f16_kernel<true>";
    if (cv3_pair_) return "conv_pw_patch_f16_kernel<pw + 3x3 + cv3>";
    if (stem_producer_) return "conv_stem_s2c32_f16_kernel<false>";
    if (pw_producer_) return in_channels_ <= 64 ? "conv_pw_patch_f16_kernel<pw + 3x3>" : "conv3x3s1_slab_f16_kernel<pw + 3x3>";
    SiConv2dDesc d = MakeDesc(in, out);
    if (sibling_) d.oc += sibling_->out_channels_;
    const int mode = PrecisionMode(in, out);
    if (mode == 1) {
        const char* name = si_hip_conv2d_f16_kernel_name(&d, up_node_ ? 1 : 0);   // the tile follows the launch size (round 4)
        return name && name[0] ? name : "conv_igemm_f16_kernel";
    }
    if (mode == 2) return "conv_stem_f16_kernel";
    if (mode == 3) return "conv_depthwise_f16_kernel<2>";
    if (mode == 4) return "conv_split3_f32_kernel";
    if (mode == 5) return "conv_wino23s_kernel";
    if (mode == 6) return "conv_stem_split_f32_kernel";
    const int tile = WinogradTile(d);
    if (tile) return tile == 4 ? "conv_wino43_kernel" : "conv_wino23_kernel";
    return si_hip_conv2d_kernel_name_form(&d, in.Data<float>(), up_node_ ? 1 : 0);
}

// 0 = implicit GEMM, 2 = fused Winograd F(2,3), 4 = fused Winograd F(4,3) for this layer's static shape
int Conv2d::WinogradTile(const SiConv2dDesc& d) const {
    if (sibling_) return 0;
    switch (algo_) {
        case Algo::kWinograd43: return si_hip_conv2d_wino43_eligible(&d) ? 4 : 0;
        case Algo::kWinograd23: return si_hip_conv2d_wino23_eligible(&d) ? 2 : 0;
        case Algo::kAuto:
            if (si_hip_conv2d_wino43_preferred(&d)) return 4;
            if (si_hip_conv2d_wino23_preferred(&d)) return (prefer_wino43_ && si_hip_conv2d_wino43_eligible(&d)) ? 4 : 2;
            return 0;
        default: return 0;
    }
}

double Conv2d::Flops() const {
    if (output_tensor_nodes_.empty() || groups_ <= 0) return 0.0;
    double elems = 0.0;
    for (auto* n : output_tensor_nodes_) elems += (double)n->tensor.NumElements();
    double f = 2.0 * elems * kernel_h_ * kernel_w_ * (in_channels_ / groups_);
    if (pw_producer_ && pw_mid_) f += 2.0 * (double)pw_mid_->tensor.NumElements() * pw_producer_->in_channels_;   // the fused-away 1x1 conv's multiplies
    if (stem_producer_ && stem_mid_)   // the fused-away stem's own multiplies (its recomputed seam not counted)
        f += 2.0 * (double)stem_mid_->tensor.NumElements() * stem_producer_->kernel_h_ * stem_producer_->kernel_w_ * stem_producer_->in_channels_;
    if (cv3_pair_) f += cv3_pair_->Flops();   // the fused-away bottleneck pair (its own 1x1 included)
    if (stem_pair_ && stem_pair_mid_) {   // the fused-away 3x3 stride-2 conv and the stem inside it
        f += 2.0 * (double)stem_pair_mid_->tensor.NumElements() * 9.0 * stem_pair_->in_channels_;
        if (stem_pair_->stem_producer_ && stem_pair_->stem_mid_)
            f += 2.0 * (double)stem_pair_->stem_mid_->tensor.NumElements() * stem_pair_->stem_producer_->kernel_h_ * stem_pair_->stem_producer_->kernel_w_ *
                 stem_pair_->stem_producer_->in_channels_;
    }
    return f;
}

double Conv2d::Bytes() const {
    double b = Layer::Bytes() + (double)weight_.size() * sizeof(float);
    if (sibling_) b += (double)sibling_->weight_.size() * sizeof(float);
    if (residual_node_) b += (double)residual_node_->tensor.ByteSize();
    if (stem_producer_) b += (double)stem_producer_->weight_.size() * sizeof(float);
    if (cv3_pair_) {
        b += (double)cv3_pair_->weight_.size() * sizeof(float);
        if (cv3_pair_->pw_producer_) b += (double)cv3_pair_->pw_producer_->weight_.size() * sizeof(float);
        if (cv3_pair_->residual_node_) b += (double)cv3_pair_->residual_node_->tensor.ByteSize();
        if (cv3_z_) b += (double)cv3_z_->tensor.ByteSize();
    }
    if (stem_pair_) {
        b += (double)stem_pair_->weight_.size() * sizeof(float);
        if (stem_pair_->stem_producer_) b += (double)stem_pair_->stem_producer_->weight_.size() * sizeof(float);
    }
    if (pw_producer_) b += (double)pw_producer_->weight_.size() * sizeof(float);
    return b;
}

bool Conv2d::HalfStorageOk(std::string& why) const {
    if (input_tensor_nodes_.empty() || output_tensor_nodes_.empty()) return true;
    const Tensor& in = input_tensor_nodes_[0]->tensor;
    const Tensor& out = output_tensor_nodes_[0]->tensor;
    if (!IsHalf(in) && !IsHalf(out)) return true;
    if (stem_producer_ || pw_producer_ || stem_pair_ || cv3_pair_) return true;   // (asked of the kernel when the pair was fused: CanFuseStemProducer / CanFusePointwiseProducer)
    SiConv2dDesc d;
    memset(&d, 0, sizeof(d));
    d.ic = in_channels_; d.oc = out_channels_; d.kh = kernel_h_; d.kw = kernel_w_; d.groups = groups_;
    d.sh = stride_h_; d.sw = stride_w_; d.dh = dilation_h_; d.dw = dilation_w_; d.pt = padding_t_; d.pl = padding_l_;
    if (si_hip_conv2d_f16_supported(&d) != 0) return true;
    why = "no fp16 conv kernel for " + std::to_string(in_channels_) + " -> " + std::to_string(out_channels_) + " channels, groups " +
          std::to_string(groups_) + " (needs ic / groups % 32 == 0, an ungrouped conv with ic % 8 == 0, a depthwise conv with ic % 8 == 0, or an RGB stem)";
    return false;
}

}  // namespace SimpleInfer
