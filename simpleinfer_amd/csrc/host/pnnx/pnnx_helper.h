// pnnx/pnnx_helper.h -- typed presence checks used by every Layer::Init
// (same four signatures as reference src/pnnx/pnnx_helper.h:10-24).
#ifndef SIMPLEINFER_AMD_PNNX_HELPER_H_
#define SIMPLEINFER_AMD_PNNX_HELPER_H_

#include <map>
#include <string>

#include "ir.h"

namespace SimpleInfer {

bool CheckParam(const std::map<std::string, pnnx::Parameter>& params, const std::string& name, const int type);
bool CheckParam(const pnnx::Operator* op, const std::string& name, const int type);
bool CheckAttr(const std::map<std::string, pnnx::Attribute>& attrs, const std::string& name, const int type);
bool CheckAttr(const pnnx::Operator* op, const std::string& name, const int type);

}  // namespace SimpleInfer

#endif
