#include "pnnx_helper.h"

namespace SimpleInfer {

bool CheckParam(const std::map<std::string, pnnx::Parameter>& params, const std::string& name, const int type) {
    auto it = params.find(name);
    return it != params.end() && it->second.type == type;
}

bool CheckParam(const pnnx::Operator* op, const std::string& name, const int type) {
    return op && CheckParam(op->params, name, type);
}

bool CheckAttr(const std::map<std::string, pnnx::Attribute>& attrs, const std::string& name, const int type) {
    auto it = attrs.find(name);
    return it != attrs.end() && it->second.type == type;
}

bool CheckAttr(const pnnx::Operator* op, const std::string& name, const int type) {
    return op && CheckAttr(op->attrs, name, type);
}

}  // namespace SimpleInfer
