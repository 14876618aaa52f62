// layer/conv_2d.h -- nn.Conv2d on the MI355X.  Public configuration fields keep the reference's
// names (src/layer/conv_2d.h:39-57) so layer tests that poke them still read the same; the three
// CPU paths (Winograd23 / im2col / grouped im2col, src/layer/conv_2d.cpp:108-118) collapse into one
// implicit-GEMM MFMA kernel (si_hip_conv2d_f32) with bias / activation / residual in its epilogue.
#ifndef SIMPLE_INFER_SRC_LAYER_CONV_2D_H_
#define SIMPLE_INFER_SRC_LAYER_CONV_2D_H_

#include "layer.h"
#include "layer_util.h"
#include "si_hip.h"

namespace SimpleInfer {

class Conv2d : public Layer {
public:
    Conv2d();
    virtual ~Conv2d() override;

public:
    virtual Status Init(const pnnx::Operator* op) override;

    virtual Status Init(const std::map<std::string, pnnx::Parameter>& params,
                        const std::map<std::string, pnnx::Attribute>& attrs) override;

    virtual Status Deinit() override;

    virtual Status Validate() override;

    virtual Status Forward(const Tensor& input, Tensor& output) override;

    // two outputs: this conv and its fused sibling (see SetSibling)
    virtual Status Forward(const Tensor& input, std::vector<Tensor>& outputs) override;

    virtual const char* KernelName() const override;
    virtual bool HalfStorageOk(std::string& why) const override;
    virtual void ExtraReads(std::vector<TensorNode*>& nodes) const override {
        if (residual_node_) nodes.push_back(residual_node_);
        if (up_node_) nodes.push_back(up_node_);
        if (cv3_pair_ && cv3_pair_->residual_node_) nodes.push_back(cv3_pair_->residual_node_);
        if (cv3_z_) nodes.push_back(cv3_z_);
    }
    virtual bool ReplaceExtraRead(TensorNode* from, TensorNode* to) override {
        if (from && from == residual_node_) { residual_node_ = to; return true; }
        return false;   // (an upsampled source is only ever set on a conv that has a kernel for its storage types)
    }
    int WinogradTile(const SiConv2dDesc& d) const;
    virtual double Flops() const override;
    virtual double Bytes() const override;

public:
    // OIHW weights (the pnnx attribute layout) + optional bias; re-laid out to the kernel's
    // [oc][kh][kw][ic/g padded] order and uploaded to HBM on the next Forward
    Status SetWeights(const std::vector<float>& weight_oihw, const std::vector<float>& bias);

    // engine fusion hook: y = act2(act1(conv + bias) + residual)
    void SetFusion(int act1, TensorNode* residual, int act2, float act_param = 0.0f);

    // engine fusion hook: `other` reads the same input with the same 1x1 geometry (YOLOv5 C3 cv1 / cv2); this layer
    // then computes both (weights concatenated along oc) and writes other's result to its second output node
    bool CanFuseSibling(const Conv2d& other) const;
    void SetSibling(Conv2d* other);
    Conv2d* Sibling() const { return sibling_; }

    // engine fusion hook: channels [c0, c0 + low->C) of this 1x1 conv's input (a torch.cat output) are nn.Upsample(nearest) of
    // `low`; the conv reads them from `low` at the source pixel and the upsample / concat copy never run
    bool CanReadUpsampled(int c0, int c) const;
    bool CanReadUpsampledFrom(const TensorNode* low, int c0, float scale_h, float scale_w) const;
    void SetUpsampledSource(TensorNode* low, int c0, float scale_h, float scale_w);
    TensorNode* UpsampledSource() const { return up_node_; }

    // engine fusion hook (fp16 storage, round 4): `stem` is the RGB stem conv whose only reader is this 3x3 stride-2 conv over 32
    // channels (YOLOv5's first two layers); this layer then reads the fp32 IMAGE and computes both in one launch
    // (si_hip_conv2d_stem_s2c32_f16), the 32-channel intermediate never leaves the CU.  Same bits as the two launches.
    bool CanFuseStemProducer(const Conv2d& stem) const;
    void SetStemProducer(Conv2d* stem);
    Conv2d* StemProducer() const { return stem_producer_; }

    // engine fusion hook (fp16 storage, round 5): `pw` is a 1x1 conv + SiLU over the same channel count whose only reader is this 3x3
    // stride-1 conv (the C3 bottleneck's pair); this layer then reads pw's INPUT and computes both in one launch
    // (si_hip_conv2d_pw_slab_f16: the 1x1 goes straight into the slab kernel's LDS patch), the intermediate is never written.  Same bits
    // as the two launches.
    bool CanFusePointwiseProducer(const Conv2d& pw) const;
    void SetPointwiseProducer(Conv2d* pw);
    Conv2d* PointwiseProducer() const { return pw_producer_; }

    // engine fusion hook (fp16 storage, round 6): `conv` is the 3x3 stride-2 conv that already carries the RGB stem (SetStemProducer) and whose
    // only readers are this 1x1 conv over its 64 channels (and its fused sibling: the first C3's cv1 | cv2); this layer then reads the fp32
    // IMAGE and computes all three in one launch (si_hip_conv2d_stem_s2c32_pw_f16), conv's output is never written.  Same bits.
    bool CanFuseStemPairProducer(const Conv2d& conv) const;
    void SetStemPairProducer(Conv2d* conv);
    Conv2d* StemPairProducer() const { return stem_pair_; }

    // engine fusion hook (fp16 storage, round 6): this 1x1 conv is a YOLOv5 C3's closing conv over cat(y, z); `pair` is the C3's LAST bottleneck
    // (a 3x3 conv that carries its 1x1 producer: SetPointwiseProducer) whose output y only the concat reads, `z` the C3's other branch.  This
    // layer then reads the pair's INPUT (and shortcut) and z and computes all three convs in one launch (si_hip_conv2d_pw_cv3_f16); neither y
    // nor the concat buffer exists.  Same bits as the launches it replaces.
    bool CanFuseCv3Pair(const Conv2d& pair, const TensorNode* z) const;
    void SetCv3Pair(Conv2d* pair, TensorNode* z);
    Conv2d* Cv3Pair() const { return cv3_pair_; }

    Status PrepareDevice(int mode = 0);
    Status PrepareDeviceHalf(const SiConv2dDesc& d);
    int PrecisionMode(const Tensor& input, const Tensor& output) const;
    Status HalfInput(const Tensor& input, Tensor& half);

    // conv + YOLOv5 decode epilogue, writing into the Detect output (kUnsupport: shape not eligible)
    Status ForwardYolo(const Tensor& input, const SiYoloLevel& level, const float* grid_dev, const float* anchor_dev,
                       Tensor& detect_out);

public:
    enum class PaddingMode { kZeros = 0, kReplicate, kReflect } padding_mode_ = PaddingMode::kZeros;
    int padding_t_    = 0;
    int padding_b_    = 0;
    int padding_l_    = 0;
    int padding_r_    = 0;
    int kernel_h_     = 0;
    int kernel_w_     = 0;
    int stride_h_     = 1;
    int stride_w_     = 1;
    int dilation_h_   = 1;
    int dilation_w_   = 1;
    int groups_       = 1;
    int in_channels_  = 0;
    int out_channels_ = 0;

    bool use_bias_ = false;
    std::vector<float> weight_;  // OIHW, host copy
    std::vector<float> bias_;

    // algorithm: the reference picks Winograd F(2,3) for every eligible 3x3 s1 conv (InitWinograd, conv_2d.cpp:182-205);
    // here kAuto does the same when the fused Winograd kernel supports the channel counts, else implicit GEMM
    enum class Algo { kAuto = 0, kImplicitGemm, kWinograd23, kWinograd43 } algo_ = Algo::kAuto;
    // engine option f32_split (opt-in, round 5): fp32 tensors, the contraction on the fp16 matrix cores from three fp16 products per
    // fp32 product (si_hip_conv2d_split3_f32) for the dense layers where that is faster (UseSplit3); everything else as without it
    bool f32_split_ = false;
    // ... its range guard (round 6): the word the split kernels set to 1 when an operand left fp16's range (engine-owned pinned host memory;
    // nullptr: unguarded), and what the engine does when it finds it set -- back to the true-fp32 kernels for this layer, for good
    int f32_split_level_ = 4;   // which layers the option takes (A/B: engine option f32_split_policy): 1 round 5's; 2 + sibling-fused and wide 1x1 layers
                                // from K = 256; 3 + the dual-source layers (upsample + concat read at the source); 4 (default) + the RGB stem
    unsigned* range_flag_ = nullptr;
    bool split_demoted_ = false;
    void DemoteSplit() { f32_split_ = false; split_demoted_ = true; device_ready_ = false; }
    // kernel-form choices handed to every launch of this layer (engine options f32_tile, f16_slab, ...; SiConvPlan in include/si_hip.h)
    void SetPlan(const SiConvPlan& plan) { plan_ = plan; has_plan_ = true; }
    bool UseSplit3() const;
    bool UseStemSplit() const;   // ... the RGB stem on the split form of the fp16 stem kernel (si_hip_conv2d_stem_split3_f32; level 4)
    bool UseWinoSplit() const;   // ... and the Winograd layers on the split form of the fused Winograd kernel (si_hip_conv2d_wino23_split_f32)
    bool prefer_wino43_ = false;  // kAuto: take F(4,3) instead of F(2,3) wherever F(2,3) would have been chosen
    bool use_winograd_ = false;  // resolved at PrepareDevice (same name as the reference's flag, conv_2d.h:60)
    int wino_tile_ = 0;          // 2 = F(2,3), 4 = F(4,3) when use_winograd_

    // fused epilogue
    int act1_ = SI_ACT_NONE;
    int act2_ = SI_ACT_NONE;
    float act_param_ = 0.0f;
    TensorNode* residual_node_ = nullptr;
    Conv2d* sibling_ = nullptr;
    TensorNode* up_node_ = nullptr;   // see SetUpsampledSource
    Conv2d* stem_producer_ = nullptr; // see SetStemProducer
    TensorNode* stem_mid_ = nullptr;  // the fused-away intermediate (shape only: it is never allocated)
    Conv2d* stem_pair_ = nullptr;     // see SetStemPairProducer
    TensorNode* stem_pair_mid_ = nullptr;
    Conv2d* cv3_pair_ = nullptr;      // see SetCv3Pair
    TensorNode* cv3_z_ = nullptr;
    TensorNode* cv3_cat_ = nullptr;   // the concat operand this conv used to read (shape only)
    Conv2d* pw_producer_ = nullptr;   // see SetPointwiseProducer
    TensorNode* pw_mid_ = nullptr;    // its fused-away output (shape only)
    int up_c0_ = 0;
    float up_scale_h_ = 1.0f, up_scale_w_ = 1.0f;

private:
    Status Launch(const Tensor& input, const Tensor* residual, Tensor& output);
    Status LaunchStemTriple(const Tensor& image, Tensor& out0, Tensor* out1);
    Status LaunchCv3(const Tensor& x, Tensor& output);
    SiConv2dDesc MakeDesc(const Tensor& input, const Tensor& output) const;
    Status MakeUpsampledSource(SiConv2dUpsampledSource& up) const;

    SiConvPlan plan_ = SI_CONV_PLAN_DEFAULT;
    bool has_plan_ = false;
    DeviceBuffer weight_dev_;
    DeviceBuffer bias_dev_;
    bool device_ready_ = false;
    int prepared_mode_ = 0;      // which weight image weight_dev_ holds (see PrepareDevice)
    Tensor in_half_;             // fp16 copy of an fp32 input consumed by the fp16 kernel
    Tensor pw_scratch_;          // the fused-away 1x1 conv's output, when the fused launch refuses the views it is handed (Launch)
};

}  // namespace SimpleInfer

#endif
