#include "expand_expression.h"

#include <algorithm>
#include <map>
#include <set>
#include <sstream>

namespace pnnx {

namespace {

const std::map<std::string, int>& unary_codes() {
    static const std::map<std::string, int> m = {
        {"abs", 0}, {"neg", 1}, {"floor", 2}, {"ceil", 3}, {"square", 4}, {"sqrt", 5}, {"rsqrt", 6}, {"exp", 7},
        {"log", 8}, {"sin", 9}, {"cos", 10}, {"tan", 11}, {"asin", 12}, {"acos", 13}, {"atan", 14},
        {"reciprocal", 15}, {"tanh", 16}, {"log10", 17}};
    return m;
}

const std::map<std::string, int>& binary_codes() {
    static const std::map<std::string, int> m = {{"add", 0}, {"sub", 1}, {"mul", 2}, {"div", 3}, {"pow", 6}, {"atan2", 10}};
    return m;
}

bool is_argument(const std::string& t) {
    if (t.size() < 2 || t[0] != '@') return false;
    return std::all_of(t.begin() + 1, t.end(), [](char c) { return c >= '0' && c <= '9'; });
}

bool is_literal(const std::string& t) {
    std::istringstream iss(t);
    float f;
    iss >> std::noskipws >> f;
    return iss.eof() && !iss.fail();
}

std::vector<std::string> tokenize(const std::string& expr) {
    std::vector<std::string> toks;
    std::string cur;
    for (char ch : expr) {
        if (ch == '[') {
            cur += ch;
            toks.push_back(cur);
            cur.clear();
        } else if (ch == '(' || ch == ')' || ch == ',' || ch == ']') {
            if (!cur.empty()) toks.push_back(cur);
            cur.clear();
        } else {
            cur += ch;
        }
    }
    if (!cur.empty()) toks.push_back(cur);
    return toks;
}

// Expands one expression operator; returns the printed form of the result ("" = unsupported).
std::string expand_one(Graph& g, const Operator* op, int& counter) {
    const std::vector<std::string> toks = tokenize(op->params.at("expr").s);
    std::vector<std::string> stack;  // printed sub-expressions, top at back

    // "@k" -> input k.  A malformed file may name an input that does not exist (the reference indexes blindly):
    // such an expression is left unexpanded, and the engine then reports the operator type as unregistered.
    auto arg_input = [&](const std::string& t) -> Operand* {
        if (t.size() < 2 || t.size() > 9) return nullptr;
        int k = 0;
        for (size_t i = 1; i < t.size(); ++i) {
            if (t[i] < '0' || t[i] > '9') return nullptr;
            k = k * 10 + (t[i] - '0');
        }
        return k < (int)op->inputs.size() ? op->inputs[k] : nullptr;
    };
    bool bad_argument = false;
    auto printed = [&](const std::string& t) -> std::string {
        if (!is_argument(t)) return t;
        Operand* r = arg_input(t);
        if (!r) {
            bad_argument = true;
            return t;
        }
        return r->name;
    };
    auto operand_of = [&](const std::string& t) -> Operand* {
        return is_argument(t) ? arg_input(t) : g.get_operand(op->name + "_" + t);
    };
    auto make_out = [&](Operator* nop, const std::string& r, const std::vector<int>& shape, int type) {
        Operand* out = g.new_operand(op->name + "_" + r);
        out->producer = nop;
        out->shape = shape;
        out->type = type;
        nop->outputs.push_back(out);
    };

    for (int i = (int)toks.size() - 1; i >= 0; --i) {
        const std::string& t = toks[i];
        if (t == "size" || t == "int" || t == "[") return std::string();

        if (unary_codes().count(t)) {
            if (stack.empty()) return std::string();
            const std::string a = stack.back();
            stack.pop_back();
            const std::string r = t + "(" + printed(a) + ")";
            if (bad_argument) return std::string();
            stack.push_back(r);
            Operator* nop = g.new_operator_before("UnaryOp", t + "_" + std::to_string(counter++), op);
            nop->params["0"] = unary_codes().at(t);
            Operand* in = operand_of(a);
            if (!in) return std::string();
            in->consumers.push_back(nop);
            nop->inputs.push_back(in);
            make_out(nop, r, in->shape, in->type);
        } else if (binary_codes().count(t)) {
            if (stack.size() < 2) return std::string();
            const std::string a = stack.back();
            stack.pop_back();
            const std::string b = stack.back();
            stack.pop_back();
            const std::string r = t + "(" + printed(a) + "," + printed(b) + ")";
            if (bad_argument) return std::string();
            stack.push_back(r);
            Operator* nop = g.new_operator_before("BinaryOp", t + "_" + std::to_string(counter++), op);
            nop->params["0"] = binary_codes().at(t);

            if (is_literal(a)) {  // scalar on the left: reversed forms for the non-commutative ops
                if (t == "sub") nop->params["0"] = 7;
                if (t == "div") nop->params["0"] = 8;
                if (t == "pow") nop->params["0"] = 9;
                if (t == "atan2") nop->params["0"] = 11;
                Operand* in = operand_of(b);
                if (!in) return std::string();
                in->consumers.push_back(nop);
                nop->params["1"] = 1;
                nop->params["2"] = std::stof(a);
                nop->inputs.push_back(in);
                make_out(nop, r, in->shape, in->type);
            } else if (is_literal(b)) {
                Operand* in = operand_of(a);
                if (!in) return std::string();
                in->consumers.push_back(nop);
                nop->params["1"] = 1;
                nop->params["2"] = std::stof(b);
                if (t == "pow" && std::stof(b) == 2.0f) {
                    nop->type = "UnaryOp";
                    nop->params.clear();
                    nop->params["0"] = 4;
                }
                nop->inputs.push_back(in);
                make_out(nop, r, in->shape, in->type);
            } else {
                Operand* ia = operand_of(a);
                Operand* ib = operand_of(b);
                if (!ia || !ib) return std::string();
                ia->consumers.push_back(nop);
                ib->consumers.push_back(nop);
                std::vector<int> sa = ia->shape, sb = ib->shape;
                const size_t rank = std::max(sa.size(), sb.size());
                sa.insert(sa.begin(), rank - sa.size(), 1);
                sb.insert(sb.begin(), rank - sb.size(), 1);
                std::vector<int> so(rank);
                for (size_t k = 0; k < rank; ++k) so[k] = std::max(sa[k], sb[k]);
                nop->inputs.push_back(ia);
                nop->inputs.push_back(ib);
                make_out(nop, r, so, ia->type);
            }
        } else {
            stack.push_back(t);  // argument or literal
        }
    }
    return stack.empty() ? std::string() : stack.back();
}

}  // namespace

void expand_expression(Graph& graph) {
    int counter = 0;
    std::set<Operator*> unsupported;
    for (;;) {
        Operator* target = nullptr;
        for (Operator* op : graph.ops)
            if (op->type == "pnnx.Expression" && !unsupported.count(op)) {
                target = op;
                break;
            }
        if (!target) break;

        if (target->outputs.empty() || !target->params.count("expr")) {  // malformed: leave it alone
            unsupported.insert(target);
            continue;
        }
        const std::string result = expand_one(graph, target, counter);
        Operand* new_out = result.empty() ? nullptr : graph.get_operand(target->name + "_" + result);
        if (!new_out) {
            unsupported.insert(target);
            continue;
        }
        Operand* old_out = target->outputs[0];
        for (Operand* r : target->inputs) r->remove_consumer(target);
        for (Operator* user : old_out->consumers) {
            new_out->consumers.push_back(user);
            for (Operand*& slot : user->inputs)
                if (slot == old_out) slot = new_out;
        }
        new_out->type = old_out->type;
        new_out->shape = old_out->shape;
        new_out->params = old_out->params;

        graph.ops.erase(std::find(graph.ops.begin(), graph.ops.end(), target));
        delete target;
        graph.operands.erase(std::find(graph.operands.begin(), graph.operands.end(), old_out));
        delete old_out;
    }
}

}  // namespace pnnx
