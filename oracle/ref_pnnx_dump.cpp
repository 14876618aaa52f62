// ref_pnnx_dump.cpp -- driver (our code) around the REFERENCE's own pnnx loader.
//
// TEST INFRASTRUCTURE ONLY.  Built by oracle/Makefile into oracle/_ref/ from
// the reference sources where they lie (/root/reference/src/pnnx/{ir,storezip,
// expand_expression}.cpp compile with plain g++, std-only); no reference source
// is copied into this repository.  It loads a .pnnx.param/.bin pair with
// pnnx::Graph::load (ir.cpp:709-815), optionally runs pnnx::expand_expression
// (expand_expression.cpp:309-387) and prints a canonical text dump that
// tests/test_pnnx_loader.py compares with the dump of our own loader
// (si_pnnx_dump in include/si_engine.h).
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>

#include "expand_expression.h"
#include "ir.h"

static uint64_t fnv1a(const std::vector<char>& d) {
    uint64_t h = 1469598103934665603ull;
    for (char c : d) {
        h ^= (unsigned char)c;
        h *= 1099511628211ull;
    }
    return h;
}

static void dump_param(const pnnx::Parameter& p) {
    printf("type=%d", p.type);
    switch (p.type) {
        case 1: printf(" b=%d", p.b ? 1 : 0); break;
        case 2: printf(" i=%d", p.i); break;
        case 3: printf(" f=%.9g", p.f); break;
        case 4: printf(" s=%s", p.s.c_str()); break;
        case 5: printf(" ai="); for (int v : p.ai) printf("%d,", v); break;
        case 6: printf(" af="); for (float v : p.af) printf("%.9g,", v); break;
        case 7: printf(" as="); for (auto& v : p.as) printf("%s,", v.c_str()); break;
        default: break;
    }
}

int main(int argc, char** argv) {
    if (argc < 3) {
        fprintf(stderr, "usage: %s model.pnnx.param model.pnnx.bin [--expand]\n", argv[0]);
        return 2;
    }
    pnnx::Graph g;
    if (g.load(argv[1], argv[2]) != 0) return 1;
    if (argc > 3 && strcmp(argv[3], "--expand") == 0) pnnx::expand_expression(g);
    printf("ops %zu operands %zu\n", g.ops.size(), g.operands.size());
    for (auto* op : g.ops) {
        printf("op %s %s in=", op->type.c_str(), op->name.c_str());
        for (auto* r : op->inputs) printf("%s,", r->name.c_str());
        printf(" out=");
        for (auto* r : op->outputs) printf("%s,", r->name.c_str());
        printf("\n");
        for (auto& kv : op->params) {
            printf("  param %s ", kv.first.c_str());
            dump_param(kv.second);
            printf("\n");
        }
        for (auto& kv : op->attrs) {
            printf("  attr %s type=%d shape=", kv.first.c_str(), kv.second.type);
            for (int s : kv.second.shape) printf("%d,", s);
            printf(" bytes=%zu fnv=%016llx\n", kv.second.data.size(),
                   (unsigned long long)fnv1a(kv.second.data));
        }
    }
    for (auto* r : g.operands) {
        printf("operand %s type=%d shape=", r->name.c_str(), r->type);
        for (int s : r->shape) printf("%d,", s);
        printf(" producer=%s consumers=", r->producer ? r->producer->name.c_str() : "-");
        for (auto* c : r->consumers) printf("%s,", c->name.c_str());
        printf("\n");
    }
    return 0;
}
