// pnnx/ir.h -- in-memory form of a pnnx graph (the file format SimpleInfer loads).
//
// Type names and public fields mirror the reference's src/pnnx/ir.h (Parameter :36-148,
// Attribute :152-173, Operand :179-201, Operator :203-223, Graph :225-259) because they ARE the
// plugin surface: Layer::Init receives `const pnnx::Operator*` and
// `std::map<std::string, pnnx::Parameter>` (reference src/layer.h:22-26).  The implementation
// (ir.cpp, storezip.cpp, expand_expression.cpp) is written from the file-format description in
// SURVEY.md 8(b), not from the reference sources.  save() writes the same two files back (SURVEY.md 8(f4):
// reference src/pnnx/ir.cpp:817-1008 minus its Python / onnx exporters, which stay out of scope).
#ifndef SIMPLEINFER_AMD_PNNX_IR_H_
#define SIMPLEINFER_AMD_PNNX_IR_H_

#include <initializer_list>
#include <map>
#include <string>
#include <vector>

namespace pnnx {

class Parameter {
public:
    Parameter() : type(0) {}
    Parameter(bool v) : type(1), b(v) {}
    Parameter(int v) : type(2), i(v) {}
    Parameter(long v) : type(2), i((int)v) {}
    Parameter(float v) : type(3), f(v) {}
    Parameter(double v) : type(3), f((float)v) {}
    Parameter(const char* v) : type(4), s(v) {}
    Parameter(const std::string& v) : type(4), s(v) {}
    Parameter(const std::initializer_list<int>& v) : type(5), ai(v) {}
    Parameter(const std::vector<int>& v) : type(5), ai(v) {}
    Parameter(const std::initializer_list<float>& v) : type(6), af(v) {}
    Parameter(const std::vector<float>& v) : type(6), af(v) {}
    Parameter(const std::vector<std::string>& v) : type(7), as(v) {}

    // text -> typed value; the classification rules of the .param syntax
    static Parameter parse_from_string(const std::string& value);

    // 0=null 1=b 2=i 3=f 4=s 5=ai 6=af 7=as
    int type;
    bool b = false;
    int i = 0;
    float f = 0.f;
    std::vector<int> ai;
    std::vector<float> af;
    std::string s;
    std::vector<std::string> as;
};

class Attribute {
public:
    Attribute() : type(0) {}
    // 0=null 1=f32 2=f64 3=f16 4=i32 5=i64 6=i16 7=i8 8=u8 9=bool
    int type;
    std::vector<int> shape;
    std::vector<char> data;
};

class Operator;

class Operand {
public:
    void remove_consumer(const Operator* c);

    Operator* producer = nullptr;
    std::vector<Operator*> consumers;
    // 0=null 1=f32 2=f64 3=f16 4=i32 5=i64 6=i16 7=i8 8=u8 9=bool 10=cp64 11=cp128 12=cp32
    int type = 0;
    std::vector<int> shape;
    std::string name;
    std::map<std::string, Parameter> params;
};

class Operator {
public:
    std::vector<Operand*> inputs;
    std::vector<Operand*> outputs;
    std::string type;
    std::string name;
    std::vector<std::string> inputnames;
    std::map<std::string, Parameter> params;
    std::map<std::string, Attribute> attrs;
};

class Graph {
public:
    Graph() = default;
    ~Graph();
    Graph(const Graph&) = delete;
    Graph& operator=(const Graph&) = delete;

    // returns 0 on success
    int load(const std::string& parampath, const std::string& binpath);
    // writes .pnnx.param / .pnnx.bin that load() (and the reference's loader) read back to the same graph
    int save(const std::string& parampath, const std::string& binpath) const;

    Operator* new_operator(const std::string& type, const std::string& name);
    Operator* new_operator_before(const std::string& type, const std::string& name, const Operator* cur);
    Operand* new_operand(const std::string& name);
    Operand* get_operand(const std::string& name);
    const Operand* get_operand(const std::string& name) const;

    std::vector<Operator*> ops;
    std::vector<Operand*> operands;
};

int type_from_string(const std::string& s);
size_t type_elemsize(int type);

}  // namespace pnnx

#endif  // SIMPLEINFER_AMD_PNNX_IR_H_
