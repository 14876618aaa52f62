// pnnx/expand_expression.h -- lowers pnnx.Expression operators into BinaryOp / UnaryOp chains.
// Behaviour contract: reference src/pnnx/expand_expression.cpp:65-307 (one expression) and
// :309-387 (graph rewrite): prefix expression such as "add(@0,mul(@1,2.0))"; each call becomes an
// operator named "<fn>_<running index>" inserted before the expression op, whose output operand is
// named "<exprop>_<fn>(<arg>,<arg>)" (SURVEY.md Q8); a literal argument turns into params
// "1"=1 (with_scalar) and "2"=<value>; BinaryOp codes add 0, sub 1, mul 2, div 3, pow 6, atan2 10
// (reversed-scalar forms 7, 8, 9, 11); pow(x, 2) becomes UnaryOp square; size/int/list are
// unsupported and leave the expression in place.
#ifndef SIMPLEINFER_AMD_PNNX_EXPAND_EXPRESSION_H_
#define SIMPLEINFER_AMD_PNNX_EXPAND_EXPRESSION_H_

#include "ir.h"

namespace pnnx {

void expand_expression(Graph& graph);

}  // namespace pnnx

#endif
