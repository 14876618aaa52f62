// python/pybind11_main.cpp -- the reference's pybind11 module (zpye/SimpleInfer python/pybind11_main.cpp:13-68) as a COMPILED extension over
// this repo's C++ Engine / Tensor (include/engine.h, include/tensor.h): the same functions, classes, method and enumerator names, so a script
// written against the reference's module runs unchanged (the pure-ctypes mirror python/simpleinfer.py exposes the same surface without a
// compiler; `import simpleinfer_pybind as infer` is this one).
//
// What differs from the reference's file, and why: the reference binds Tensor::SetEigenTensor / GetEigenTensor through pybind11's Eigen TensorMap
// caster; Eigen is not a dependency here (include/tensor.h: SetData / Data<T>), so the two methods take / return numpy arrays directly --
//   SetTensorDim4(array)  a C-contiguous float32 array of rank 4 is BORROWED (its pointer is stored and the array kept alive by the Tensor;
//                         nothing is copied and Shape() stays as constructed: Tensor::SetEigenTensor, reference include/tensor.h:39-52)
//   GetTensorDim4()       a numpy VIEW of the tensor's host memory with the leading dimensions folded -- or padded with 1 -- to exactly four
//                         (ToEigenDSize, reference include/eigen_helper.h:32-63); zero-copy: the array's base object is the Tensor
// which is what the Eigen caster does under the hood, with py::return_value_policy::reference semantics for both.
// Built by `python -m simpleinfer_amd.build --pybind` (g++, pybind11 headers from the wheel; links libsimpleinfer_amd.so by rpath).
#include <pybind11/numpy.h>
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include <memory>
#include <stdexcept>

#include "engine.h"
#include "tensor.h"
#include "types.h"

using namespace SimpleInfer;

namespace py = pybind11;

namespace {

// Tensor plus the Python object whose memory it borrows (SetTensorDim4 keeps the array alive, as a Python user expects of a borrow)
struct PyTensor : public Tensor {
    using Tensor::Tensor;
    py::object keep;
};

Status SetTensorDim4(PyTensor& self, py::array array) {
    if (self.GetDataType() != DataType::kFloat32) return Status::kFail;
    if (!py::isinstance<py::array_t<float>>(array) || array.ndim() != 4 || !(array.flags() & py::array::c_style))
        throw py::type_error("SetTensorDim4(): incompatible function arguments: expected a C-contiguous float32 array of rank 4");
    if (self.OwnsData()) return Status::kFail;
    const Status st = self.SetData(array.mutable_data(), MemoryType::kHost);
    if (st == Status::kSuccess) self.keep = array;
    return st;
}

py::array GetTensorDim4(py::object self_obj) {
    PyTensor& self = self_obj.cast<PyTensor&>();
    if (self.RawData() == nullptr || self.GetMemoryType() != MemoryType::kHost || self.GetDataType() != DataType::kFloat32)
        throw std::runtime_error("GetTensorDim4(): the tensor holds no host float32 data");
    const std::vector<int> s4 = self.ShapeAs(4);
    std::vector<py::ssize_t> shape(s4.begin(), s4.end());
    // a view: the Tensor object is the array's base, so the memory outlives the array as long as the Tensor (or what it borrows) does
    return py::array_t<float>(shape, self.Data<float>(), self_obj);
}

}  // namespace

PYBIND11_MODULE(simpleinfer_pybind, m) {
    m.doc() = "pybind11 SimpleInfer (MI355X engine)";

    m.def("InitializeContext", &InitializeContext);

    py::enum_<DataType>(m, "DataType")
        .value("None", DataType::kNone)
        .value("Float32", DataType::kFloat32);

    py::enum_<Status>(m, "Status")
        .value("Success", Status::kSuccess)
        .value("Fail", Status::kFail)
        .value("Empty", Status::kEmpty)
        .value("ErrorShape", Status::kErrorShape)
        .value("ErrorContext", Status::kErrorContext)
        .value("Unsupport", Status::kUnsupport);

    py::class_<PyTensor>(m, "Tensor")
        .def(py::init<>())
        .def(py::init([](DataType dt, std::vector<int> shape) { return std::make_unique<PyTensor>(dt, shape); }))
        .def("GetDataType", [](const PyTensor& t) { return t.GetDataType(); })
        .def("Shape", [](const PyTensor& t) { return t.Shape(); })
        .def("SetTensorDim4", &SetTensorDim4)
        .def("GetTensorDim4", &GetTensorDim4);

    py::class_<Engine>(m, "Engine")
        .def(py::init<>())
        .def("LoadModel", static_cast<Status (Engine::*)(const std::string&, const std::string&)>(&Engine::LoadModel))
        .def("Release", static_cast<Status (Engine::*)()>(&Engine::Release))
        .def("InputNames", static_cast<const std::vector<std::string> (Engine::*)()>(&Engine::InputNames))
        .def("OutputNames", static_cast<const std::vector<std::string> (Engine::*)()>(&Engine::OutputNames))
        .def("Input", [](Engine& e, const std::string& name, const PyTensor& t) { return e.Input(name, t); }, py::keep_alive<1, 3>())
        .def("Forward", static_cast<Status (Engine::*)()>(&Engine::Forward), py::call_guard<py::gil_scoped_release>())
        .def("Extract", [](Engine& e, const std::string& name, PyTensor& t) {
            // (Engine::Extract assigns a non-owning view of engine memory: src/engine_impl.cpp:546-555)
            Tensor view;
            const Status st = e.Extract(name, view);
            if (st == Status::kSuccess) {
                static_cast<Tensor&>(t) = view;
                t.keep = py::none();
            }
            return st;
        }, py::keep_alive<3, 1>());
}
