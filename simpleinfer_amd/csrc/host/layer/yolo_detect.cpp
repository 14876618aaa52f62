#include "yolo_detect.h"

#include <cstring>

#include "si_hip.h"

namespace SimpleInfer {

DEFINE_LAYER_REGISTRY(YoloDetect);

constexpr int YoloDetect::anchor_index[];
constexpr int YoloDetect::grid_index[];

// [1][na][H][W][2] (pnnx attribute) -> [H][W][na][2]: the shuffle (0,2,3,1,4) of reference
// yolo_detect.cpp:75-79,104-106
static bool ShuffleGrid(const pnnx::Attribute& a, std::vector<float>& dst, int& na, int& h, int& w) {
    if (5 != a.shape.size() || 1 != a.shape[0] || 2 != a.shape[4]) return false;
    na = a.shape[1]; h = a.shape[2]; w = a.shape[3];
    const size_t count = (size_t)na * h * w * 2;
    if (a.data.size() != count * sizeof(float)) return false;
    const float* src = reinterpret_cast<const float*>(a.data.data());
    dst.resize(count);
    for (int k = 0; k < na; ++k)
        for (int y = 0; y < h; ++y)
            for (int x = 0; x < w; ++x)
                for (int e = 0; e < 2; ++e)
                    dst[(((size_t)y * w + x) * na + k) * 2 + e] = src[(((size_t)k * h + y) * w + x) * 2 + e];
    return true;
}

Status YoloDetect::Init(const pnnx::Operator* op) {
    CHECK_STATUS(Layer::Init(op));
    CHECK_BOOL((int)op->inputs.size() == num_spatial_sizes);

    CHECK_BOOL(CheckAttr(op->attrs, "pnnx_5", 1));
    const pnnx::Attribute& st = op->attrs.at("pnnx_5");
    CHECK_BOOL(1 == st.shape.size() && 3 == st.shape[0] && st.data.size() == 3 * sizeof(float));
    memcpy(strides_, st.data.data(), 3 * sizeof(float));

    for (int i = 0; i < num_spatial_sizes; ++i) {
        const std::string wname = "m." + std::to_string(i) + ".weight";
        const std::string bname = "m." + std::to_string(i) + ".bias";
        CHECK_BOOL(CheckAttr(op->attrs, wname, 1));
        CHECK_BOOL(CheckAttr(op->attrs, bname, 1));
        const std::vector<int>& ws = op->attrs.at(wname).shape;
        CHECK_BOOL(4 == ws.size() && 1 == ws[2] && 1 == ws[3]);

        // the embedded 1x1 Conv2d is configured through the same params/attrs Init as a graph conv
        std::map<std::string, pnnx::Parameter> params;
        std::map<std::string, pnnx::Attribute> attrs;
        attrs["weight"] = op->attrs.at(wname);
        attrs["bias"] = op->attrs.at(bname);
        params["bias"] = pnnx::Parameter(true);
        params["padding_mode"] = pnnx::Parameter("zeros");
        params["padding"] = pnnx::Parameter({0, 0});
        params["kernel_size"] = pnnx::Parameter({1, 1});
        params["stride"] = pnnx::Parameter({1, 1});
        params["dilation"] = pnnx::Parameter({1, 1});
        params["groups"] = pnnx::Parameter(1);
        params["in_channels"] = pnnx::Parameter(ws[1]);
        params["out_channels"] = pnnx::Parameter(ws[0]);
        CHECK_STATUS(conv_2d_layer_[i].Init(params, attrs));

        if (0 == i) {
            num_elements_ = ws[0];
        } else {
            CHECK_BOOL(num_elements_ == ws[0]);
        }

        const std::string aname = "pnnx_" + std::to_string(anchor_index[i]);
        const std::string gname = "pnnx_" + std::to_string(grid_index[i]);
        CHECK_BOOL(CheckAttr(op->attrs, aname, 1));
        CHECK_BOOL(CheckAttr(op->attrs, gname, 1));
        int na_a = 0, ha = 0, wa = 0, na_g = 0, hg = 0, wg = 0;
        CHECK_BOOL(ShuffleGrid(op->attrs.at(aname), anchor_grids_[i], na_a, ha, wa));
        CHECK_BOOL(ShuffleGrid(op->attrs.at(gname), grids_[i], na_g, hg, wg));
        CHECK_BOOL(na_a == na_g && ha == hg && wa == wg);
        level_h_[i] = ha;
        level_w_[i] = wa;
        if (0 == i) {
            num_anchor_grid_levels_ = na_a;
        } else {
            CHECK_BOOL(num_anchor_grid_levels_ == na_a);
        }

        // conv output buffer [N][H][W][na*ne]; input operand shape is NCHW in the file
        const std::vector<int>& in_shape = op->inputs[i]->shape;
        CHECK_BOOL(4 == in_shape.size());
        spatial_output[i] = Tensor(DataType::kFloat32, {in_shape[0], in_shape[2], in_shape[3], ws[0]}, MemoryType::kDevice, false);
    }
    CHECK_BOOL(num_anchor_grid_levels_ > 0 && 0 == num_elements_ % num_anchor_grid_levels_);
    num_classes_info_ = num_elements_ / num_anchor_grid_levels_;
    device_ready_ = false;
    return Status::kSuccess;
}

void YoloDetect::SetContext(Context* context) {
    Layer::SetContext(context);
    for (int i = 0; i < num_spatial_sizes; ++i) conv_2d_layer_[i].SetContext(context);
}

Status YoloDetect::Deinit() {
    for (int i = 0; i < num_spatial_sizes; ++i) {
        conv_2d_layer_[i].Deinit();
        spatial_output[i].Deallocate();
        grids_dev_[i].Free();
        anchor_grids_dev_[i].Free();
    }
    device_ready_ = false;
    return Status::kSuccess;
}

Status YoloDetect::Validate() {
    CHECK_STATUS(Layer::Validate());
    CHECK_STATUS(ValidateShape(3, 1));
    if (Status::kSuccess != ValidateFloat()) {
        LOG(ERROR) << "YoloDetect::Validate fail [unsupport input/output data type]";
        return Status::kUnsupport;
    }
    return Status::kSuccess;
}

void YoloDetect::SetEarlyLevels(Context* side, unsigned mask) {
    early_mask_ = side ? mask : 0u;
    for (int i = 0; i < num_spatial_sizes; ++i) conv_2d_layer_[i].SetContext(((early_mask_ >> i) & 1u) ? side : context_);
}

Status YoloDetect::PrepareDevice() {
    if (device_ready_) return Status::kSuccess;
    for (int i = 0; i < num_spatial_sizes; ++i) {
        CHECK_STATUS(CheckHip(grids_dev_[i].Upload(grids_[i].data(), grids_[i].size() * sizeof(float)), "upload grid"));
        CHECK_STATUS(CheckHip(anchor_grids_dev_[i].Upload(anchor_grids_[i].data(), anchor_grids_[i].size() * sizeof(float)), "upload anchors"));
    }
    device_ready_ = true;
    return Status::kSuccess;
}

// one level: 1x1 conv with the decode + concat in its epilogue (one launch), or conv + decode kernel where that form is not available
Status YoloDetect::LaunchLevel(int i, const Tensor& in, Tensor& out, int rows_total, int row_off) {
    Dims4 d;
    if (!GetDims4(in, d) || d.h != level_h_[i] || d.w != level_w_[i]) return Status::kErrorShape;
    if (fuse_decode_) {
        SiYoloLevel lv;
        lv.na = num_anchor_grid_levels_; lv.ne = num_classes_info_; lv.rows_total = rows_total; lv.row_off = row_off;
        lv.stride = strides_[i];
        const Status fs = conv_2d_layer_[i].ForwardYolo(in, lv, grids_dev_[i].As<float>(), anchor_grids_dev_[i].As<float>(), out);
        if (fs != Status::kUnsupport) return fs;
    }
    CHECK_STATUS(spatial_output[i].Allocate(DataType::kFloat32, {d.n, d.h, d.w, num_elements_}));
    CHECK_STATUS(conv_2d_layer_[i].Forward(in, spatial_output[i]));
    return CheckHip(si_hip_yolo_decode_f32(spatial_output[i].Data<float>(), d.n, d.h, d.w, num_anchor_grid_levels_, num_classes_info_,
                                           grids_dev_[i].As<float>(), anchor_grids_dev_[i].As<float>(), strides_[i], out.Data<float>(),
                                           rows_total, row_off, conv_2d_layer_[i].LaunchStream()),
                    "YoloDetect decode");
}

Status YoloDetect::ForwardLevel(int level) {
    if (level < 0 || level >= num_spatial_sizes || (int)input_tensor_nodes_.size() != num_spatial_sizes || output_tensor_nodes_.size() != 1)
        return Status::kErrorShape;
    const Tensor& in = input_tensor_nodes_[level]->tensor;
    Tensor& out = output_tensor_nodes_[0]->tensor;
    if (in.GetMemoryType() != MemoryType::kDevice || out.GetMemoryType() != MemoryType::kDevice || IsHalf(out)) return Status::kUnsupport;
    CHECK_STATUS(PrepareDevice());
    const std::vector<int> os = out.ShapeAs(3);
    if (os[2] != num_classes_info_) return Status::kErrorShape;
    int row_off = 0;
    for (int i = 0; i < level; ++i) row_off += level_h_[i] * level_w_[i] * num_anchor_grid_levels_;
    if (row_off + level_h_[level] * level_w_[level] * num_anchor_grid_levels_ > os[1]) return Status::kErrorShape;
    return LaunchLevel(level, in, out, os[1], row_off);
}

Status YoloDetect::Forward(const std::vector<Tensor>& inputs, Tensor& output) {
    if ((int)inputs.size() != num_spatial_sizes) return Status::kErrorShape;
    return RunOnDevice({&inputs[0], &inputs[1], &inputs[2]}, {&output},
                       [this](const std::vector<Tensor>& in, std::vector<Tensor>& out) {
        CHECK_STATUS(PrepareDevice());
        if (IsHalf(out[0])) return Status::kUnsupport;  // detections are always fp32 (features may be fp16)
        const std::vector<int> os = out[0].ShapeAs(3);
        const int rows_total = os[1];
        if (os[2] != num_classes_info_) return Status::kErrorShape;
        // the output's row count comes from the model file: check it against what the levels will write BEFORE any launch (a
        // file whose Detect operand is too small would otherwise get out-of-bounds device writes, then an error)
        {
            long long expect = 0;
            for (int i = 0; i < num_spatial_sizes; ++i) {
                Dims4 d;
                if (!GetDims4(in[i], d) || d.h != level_h_[i] || d.w != level_w_[i] || d.n != os[0]) return Status::kErrorShape;
                expect += (long long)d.h * d.w * num_anchor_grid_levels_;
            }
            if (expect != rows_total) {
                LOG(ERROR) << "YoloDetect: output has " << rows_total << " rows per image, the levels produce " << expect;
                return Status::kErrorShape;
            }
        }
        int row_off = 0;
        for (int i = 0; i < num_spatial_sizes; ++i) {
            // (a level the engine has already launched on the side stream this forward is skipped here)
            if (!((early_mask_ >> i) & 1u)) CHECK_STATUS(LaunchLevel(i, in[i], out[0], rows_total, row_off));
            row_off += level_h_[i] * level_w_[i] * num_anchor_grid_levels_;
        }
        return row_off == rows_total ? Status::kSuccess : Status::kErrorShape;
    });
}

double YoloDetect::Flops() const {
    double f = 0.0;
    for (int i = 0; i < num_spatial_sizes && i < (int)input_tensor_nodes_.size(); ++i)
        f += 2.0 * (double)input_tensor_nodes_[i]->tensor.NumElements() * num_elements_;
    return f;
}

bool YoloDetect::HalfStorageOk(std::string& why) const {
    if (!output_tensor_nodes_.empty() && IsHalf(output_tensor_nodes_[0]->tensor)) { why = "Detect writes fp32 only"; return false; }
    for (int i = 0; i < num_spatial_sizes && i < (int)input_tensor_nodes_.size(); ++i) {
        const Tensor& t = input_tensor_nodes_[i]->tensor;
        if (IsHalf(t) && !t.Shape().empty() && t.Shape().back() % 32 != 0) { why = "Detect's fp16 1x1 convs need channel counts that are multiples of 32"; return false; }
    }
    return true;
}

}  // namespace SimpleInfer
