// pnnx/ir.cpp -- .pnnx.param text parser + attribute loading.
// Format contract (reference src/pnnx/ir.cpp:479-550 value syntax, :597-707
// shapes/attributes, :709-815 file walk): line 1 magic, line 2 "<ops> <operands>", then one line
// per operator "type name n_in n_out in.. out.. key=value..", where a key starting with
//   '@' is an attribute "(shape)dtype" whose bytes are the zip entry "<opname>.<key>",
//   '$' names an input slot, '#' gives an operand's "(shape)dtype", anything else is a parameter.
#include "ir.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <sstream>

#include "storezip.h"

namespace pnnx {

int type_from_string(const std::string& s) {
    static const char* names[] = {"", "f32", "f64", "f16", "i32", "i64", "i16", "i8", "u8", "bool", "cp64", "cp128", "cp32"};
    for (int i = 1; i <= 12; ++i)
        if (s == names[i]) return i;
    return 0;
}

size_t type_elemsize(int type) {
    switch (type) {
        case 1: case 4: case 12: return 4;
        case 2: case 5: case 10: return 8;
        case 3: case 6: return 2;
        case 7: case 8: case 9: return 1;
        case 11: return 16;
        default: return 0;
    }
}

namespace {

// "looks numeric": first char a digit, or '-' followed by a digit
bool starts_number(const std::string& t) {
    if (t.empty()) return false;
    if (t[0] >= '0' && t[0] <= '9') return true;
    return t[0] == '-' && t.size() > 1 && t[1] >= '0' && t[1] <= '9';
}
bool looks_float(const std::string& t) { return t.find('.') != std::string::npos || t.find('e') != std::string::npos; }

std::vector<std::string> split(const std::string& s, char sep) {
    std::vector<std::string> out;
    size_t b = 0;
    while (true) {
        const size_t e = s.find(sep, b);
        out.push_back(s.substr(b, e == std::string::npos ? std::string::npos : e - b));
        if (e == std::string::npos) break;
        b = e + 1;
    }
    return out;
}

// "(a,b,c)suffix" -> {a,b,c}, suffix
bool split_shape(const std::string& v, std::vector<std::string>& dims, std::string& suffix) {
    const size_t close = v.find_last_of(')');
    if (v.empty() || v[0] != '(' || close == std::string::npos) return false;
    suffix = v.substr(close + 1);
    const std::string inner = v.substr(1, close - 1);
    dims = inner.empty() ? std::vector<std::string>() : split(inner, ',');
    return true;
}

}  // namespace

Parameter Parameter::parse_from_string(const std::string& value) {
    Parameter p;
    if (value.empty() || value == "None" || value == "()" || value == "[]") return p;
    if (value == "True" || value == "False") {
        p.type = 1;
        p.b = (value == "True");
        return p;
    }
    if (value[0] == '(' || value[0] == '[') {
        // list: the element kind is decided per element, the last element's kind wins the tag
        for (const std::string& e : split(value.substr(1, value.size() - 2), ',')) {
            if (!starts_number(e)) {
                p.type = 7;
                p.as.push_back(e);
            } else if (looks_float(e)) {
                p.type = 6;
                p.af.push_back(std::stof(e));
            } else {
                p.type = 5;
                p.ai.push_back(std::stoi(e));
            }
        }
        return p;
    }
    if (!starts_number(value)) {
        p.type = 4;
        p.s = value;
    } else if (looks_float(value)) {
        p.type = 3;
        p.f = std::stof(value);
    } else {
        p.type = 2;
        p.i = std::stoi(value);
    }
    return p;
}

void Operand::remove_consumer(const Operator* c) {
    auto it = std::find(consumers.begin(), consumers.end(), c);
    if (it != consumers.end()) consumers.erase(it);
}

Graph::~Graph() {
    for (auto* x : ops) delete x;
    for (auto* x : operands) delete x;
}

Operator* Graph::new_operator(const std::string& type, const std::string& name) {
    Operator* op = new Operator;
    op->type = type;
    op->name = name;
    ops.push_back(op);
    return op;
}

Operator* Graph::new_operator_before(const std::string& type, const std::string& name, const Operator* cur) {
    Operator* op = new Operator;
    op->type = type;
    op->name = name;
    ops.insert(std::find(ops.begin(), ops.end(), cur), op);
    return op;
}

Operand* Graph::new_operand(const std::string& name) {
    Operand* r = new Operand;
    r->name = name;
    operands.push_back(r);
    return r;
}

Operand* Graph::get_operand(const std::string& name) {
    for (Operand* r : operands)
        if (r->name == name) return r;
    return nullptr;
}

const Operand* Graph::get_operand(const std::string& name) const {
    for (const Operand* r : operands)
        if (r->name == name) return r;
    return nullptr;
}

int Graph::load(const std::string& parampath, const std::string& binpath) {
    std::ifstream is(parampath, std::ios::in | std::ios::binary);
    if (!is.good()) {
        fprintf(stderr, "pnnx: cannot open %s\n", parampath.c_str());
        return -1;
    }
    StoreZipReader zip;
    if (zip.open(binpath) != 0) return -1;

    std::string line;
    int magic = 0, n_ops = 0, n_operands = 0;
    if (std::getline(is, line)) std::istringstream(line) >> magic;
    if (std::getline(is, line)) std::istringstream(line) >> n_ops >> n_operands;
    (void)magic;
    (void)n_operands;

    for (int i = 0; i < n_ops; ++i) {
        if (!std::getline(is, line)) break;
        std::istringstream ls(line);
        std::string type, name;
        int n_in = 0, n_out = 0;
        ls >> type >> name >> n_in >> n_out;
        Operator* op = new_operator(type, name);

        for (int j = 0; j < n_in; ++j) {
            std::string rn;
            ls >> rn;
            Operand* r = get_operand(rn);
            if (!r) {
                fprintf(stderr, "pnnx: operator %s consumes unknown operand %s\n", name.c_str(), rn.c_str());
                return -1;
            }
            r->consumers.push_back(op);
            op->inputs.push_back(r);
        }
        for (int j = 0; j < n_out; ++j) {
            std::string rn;
            ls >> rn;
            Operand* r = new_operand(rn);
            r->producer = op;
            op->outputs.push_back(r);
        }

        std::string tok;
        while (ls >> tok) {
            const size_t eq = tok.find('=');
            const std::string key = tok.substr(0, eq);
            const std::string val = eq == std::string::npos ? std::string() : tok.substr(eq + 1);
            if (key.empty()) continue;

            if (key[0] == '@') {
                Attribute& a = op->attrs[key.substr(1)];
                std::vector<std::string> dims;
                std::string suffix;
                if (!split_shape(val, dims, suffix)) continue;
                a.type = type_from_string(suffix);
                if (a.type == 0) continue;
                a.shape.clear();
                for (const auto& d : dims) a.shape.push_back(std::stoi(d));
                if (a.shape.empty()) continue;
                const std::string entry = op->name + "." + key.substr(1);
                const size_t have = zip.get_file_size(entry);
                if (have == 0) continue;  // no such entry: attribute keeps shape, no data
                // the byte count the shape implies must be the entry's, checked BEFORE anything is sized by it (a corrupt
                // shape must not drive an allocation; the layer that needs the data then fails its own size check at Init)
                size_t count = 1;
                bool sane = true;
                for (int d : a.shape) {
                    if (d <= 0 || count > have) { sane = false; break; }
                    count *= (size_t)d;
                }
                if (!sane || count * type_elemsize(a.type) != have) {
                    fprintf(stderr, "pnnx: %s holds %zu bytes, its shape says otherwise: ignored\n", entry.c_str(), have);
                    continue;
                }
                a.data.resize(have);
                zip.read_file(entry, a.data.data());
            } else if (key[0] == '$') {
                op->inputnames.resize(op->inputs.size());
                for (size_t j = 0; j < op->inputs.size(); ++j)
                    if (op->inputs[j]->name == val) {
                        op->inputnames[j] = key.substr(1);
                        break;
                    }
            } else if (key[0] == '#') {
                const std::string rn = key.substr(1);
                Operand* target = nullptr;
                for (Operand* r : op->inputs)
                    if (r->name == rn) { target = r; break; }
                if (!target)
                    for (Operand* r : op->outputs)
                        if (r->name == rn) { target = r; break; }
                if (!target) {
                    fprintf(stderr, "pnnx: no operand %s on operator %s\n", rn.c_str(), op->name.c_str());
                    continue;
                }
                std::vector<std::string> dims;
                std::string suffix;
                if (!split_shape(val, dims, suffix)) continue;
                target->type = type_from_string(suffix);
                target->shape.clear();
                for (const auto& d : dims) target->shape.push_back(d == "?" ? -1 : std::stoi(d));
            } else {
                op->params[key] = Parameter::parse_from_string(val);
            }
        }
    }
    return 0;
}

// ---- writer ---------------------------------------------------------------------------------------
namespace {

const char* type_to_string(int type) {
    static const char* names[] = {"null", "f32", "f64", "f16", "i32", "i64", "i16", "i8", "u8", "bool", "cp64", "cp128", "cp32"};
    return type >= 0 && type <= 12 ? names[type] : "null";
}

// shortest text that parses back to the same float AND is classified as a float by parse_from_string
std::string float_text(float v) {
    char buf[48];
    for (int digits = 6; digits <= 9; ++digits) {
        snprintf(buf, sizeof(buf), "%.*e", digits, (double)v);
        if (std::strtof(buf, nullptr) == v) break;
    }
    return buf;
}

std::string shape_text(const std::vector<int>& shape, int type) {
    std::string t = "(";
    for (size_t i = 0; i < shape.size(); ++i) {
        if (i) t += ",";
        t += shape[i] < 0 ? std::string("?") : std::to_string(shape[i]);
    }
    return t + ")" + type_to_string(type);
}

std::string param_text(const Parameter& p) {
    std::string t;
    switch (p.type) {
        case 1: return p.b ? "True" : "False";
        case 2: return std::to_string(p.i);
        case 3: return float_text(p.f);
        case 4: return p.s;
        case 5:
            for (size_t i = 0; i < p.ai.size(); ++i) t += (i ? "," : "") + std::to_string(p.ai[i]);
            return "(" + t + ")";
        case 6:
            for (size_t i = 0; i < p.af.size(); ++i) t += (i ? "," : "") + float_text(p.af[i]);
            return "(" + t + ")";
        case 7:
            for (size_t i = 0; i < p.as.size(); ++i) t += (i ? "," : "") + p.as[i];
            return "(" + t + ")";
        default: return "None";
    }
}

}  // namespace

int Graph::save(const std::string& parampath, const std::string& binpath) const {
    FILE* fp = fopen(parampath.c_str(), "wb");
    if (!fp) {
        fprintf(stderr, "pnnx: cannot create %s\n", parampath.c_str());
        return -1;
    }
    StoreZipWriter zip;
    if (zip.open(binpath) != 0) {
        fclose(fp);
        return -1;
    }
    int rc = 0;
    fprintf(fp, "7767517\n%d %d\n", (int)ops.size(), (int)operands.size());
    for (const Operator* op : ops) {
        fprintf(fp, "%-24s %-24s %d %d", op->type.c_str(), op->name.c_str(), (int)op->inputs.size(), (int)op->outputs.size());
        for (const Operand* r : op->inputs) fprintf(fp, " %s", r->name.c_str());
        for (const Operand* r : op->outputs) fprintf(fp, " %s", r->name.c_str());
        for (const auto& kv : op->params) fprintf(fp, " %s=%s", kv.first.c_str(), param_text(kv.second).c_str());
        for (const auto& kv : op->attrs) {
            const Attribute& a = kv.second;
            fprintf(fp, " @%s=%s", kv.first.c_str(), shape_text(a.shape, a.type).c_str());
            if (!a.data.empty() && zip.write_file(op->name + "." + kv.first, a.data.data(), a.data.size()) != 0) rc = -1;
        }
        for (size_t j = 0; j < op->inputnames.size() && j < op->inputs.size(); ++j)
            if (!op->inputnames[j].empty()) fprintf(fp, " $%s=%s", op->inputnames[j].c_str(), op->inputs[j]->name.c_str());
        // operand shapes ride on both the consuming and the producing line, as pnnx writes them
        for (const Operand* r : op->inputs)
            if (r->type != 0 || !r->shape.empty()) fprintf(fp, " #%s=%s", r->name.c_str(), shape_text(r->shape, r->type).c_str());
        for (const Operand* r : op->outputs)
            if (r->type != 0 || !r->shape.empty()) fprintf(fp, " #%s=%s", r->name.c_str(), shape_text(r->shape, r->type).c_str());
        fprintf(fp, "\n");
    }
    if (ferror(fp)) rc = -1;
    if (fclose(fp) != 0) rc = -1;
    if (zip.close() != 0) rc = -1;
    return rc;
}

}  // namespace pnnx
