// layer/yolo_detect.h -- models.yolo.Detect (YOLOv5 head).  Attribute names, level order and the
// [H][W][anchor] row order are the reference's (src/layer/yolo_detect.h:27-41, yolo_detect.cpp:11-165
// Init, :204-272 Forward).  Per level: 1x1 conv on the MFMA kernel, then one decode kernel
// that applies sigmoid, the grid/anchor transforms and the concat into [N, rows, 85] in a single
// pass (the reference makes a sigmoid pass plus two slice passes per level).
#ifndef SIMPLE_INFER_SRC_LAYER_YOLO_DETECT_H_
#define SIMPLE_INFER_SRC_LAYER_YOLO_DETECT_H_

#include "conv_2d.h"
#include "layer.h"
#include "layer_util.h"

namespace SimpleInfer {

class YoloDetect : public Layer {
public:
    virtual Status Init(const pnnx::Operator* op) override;
    virtual void SetContext(Context* context) override;
    virtual Status Deinit() override;
    virtual Status Validate() override;
    virtual Status Forward(const std::vector<Tensor>& inputs, Tensor& output) override;
    // Engine option "detect_stream": the levels in `mask` launch on `side`'s stream as soon as the engine has their input
    // (ForwardLevel, called by the engine right after the producing step); Forward() then launches only the others.
    void SetEarlyLevels(Context* side, unsigned mask);
    unsigned EarlyLevels() const { return early_mask_; }
    Status ForwardLevel(int level);   // on the bound (device) tensors
    virtual bool HalfStorageOk(std::string& why) const override;
    virtual const char* KernelName() const override {
        const bool half = !input_tensor_nodes_.empty() && IsHalf(input_tensor_nodes_[0]->tensor);
        if (half) return fuse_decode_ ? "conv_igemm_f16(yolo epilogue)" : "conv_igemm_f16+yolo_decode";
        if (fuse_decode_ && conv_2d_layer_[0].f32_split_) return "conv_igemm_f32 / conv_split3_f32(yolo epilogue)";
        return fuse_decode_ ? "conv_igemm_f32(yolo epilogue)" : "conv_igemm_f32+yolo_decode";
    }
    virtual double Flops() const override;

public:
    static const int num_spatial_sizes = 3;
    static constexpr int anchor_index[num_spatial_sizes]{4, 2, 0};
    static constexpr int grid_index[num_spatial_sizes]{6, 3, 1};

    Conv2d conv_2d_layer_[num_spatial_sizes];
    Tensor spatial_output[num_spatial_sizes];  // device, [N][H][W][na*ne]

    std::vector<float> anchor_grids_[num_spatial_sizes];  // [H*W*na][2]
    std::vector<float> grids_[num_spatial_sizes];
    int level_h_[num_spatial_sizes] = {0, 0, 0};
    int level_w_[num_spatial_sizes] = {0, 0, 0};
    float strides_[num_spatial_sizes] = {8.f, 16.f, 32.f};

    int num_elements_           = 255;
    int num_anchor_grid_levels_ = 3;
    int num_classes_info_       = 85;

    bool fuse_decode_ = true;  // decode + concat in the conv epilogue (engine option "fuse")

private:
    DeviceBuffer grids_dev_[num_spatial_sizes], anchor_grids_dev_[num_spatial_sizes];
    bool device_ready_ = false;
    unsigned early_mask_ = 0;
    Status PrepareDevice();
    Status LaunchLevel(int i, const Tensor& in, Tensor& out, int rows_total, int row_off);
};

}  // namespace SimpleInfer

#endif
