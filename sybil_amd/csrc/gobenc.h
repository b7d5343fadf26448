// gobenc.h -- a minimal encoding/gob ENCODER (the inverse of gob.cpp): type descriptors, Go's type-id
// allocation order (type.go: a struct takes its id before its fields are visited, slices and maps after
// their element types) and type-definition messages before first use (encoder.go:sendActualType).
// Used by writer.cpp to write tables in the reference's on-disk format.
#pragma once
#include <stdint.h>
#include <string.h>

#include <memory>
#include <set>
#include <string>
#include <utility>
#include <vector>

namespace sybl {
namespace gobenc {

enum Builtin : int { kBool = 1, kInt = 2, kUint = 3, kFloat = 4, kBytes = 5, kString = 6 };

struct Type {
    enum Kind { Basic, Slice, Map, Struct } kind = Basic;
    int id = 0;
    std::string name;
    Type *elem = nullptr, *key = nullptr;
    std::vector<std::pair<std::string, Type *>> fields;
};

// owns the descriptors of one schema
struct Schema {
    std::vector<std::unique_ptr<Type>> all;
    Type *basic(int id) {
        all.emplace_back(new Type());
        all.back()->id = id;
        return all.back().get();
    }
    Type *slice(Type *elem, const char *name) {
        all.emplace_back(new Type());
        Type *t = all.back().get();
        t->kind = Type::Slice;
        t->elem = elem;
        t->name = name;
        return t;
    }
    Type *map(Type *key, Type *elem, const char *name) {
        all.emplace_back(new Type());
        Type *t = all.back().get();
        t->kind = Type::Map;
        t->key = key;
        t->elem = elem;
        t->name = name;
        return t;
    }
    Type *strukt(const char *name, std::vector<std::pair<std::string, Type *>> fields) {
        all.emplace_back(new Type());
        Type *t = all.back().get();
        t->kind = Type::Struct;
        t->name = name;
        t->fields = std::move(fields);
        return t;
    }
};

struct Buf {
    std::string b;
    void u(uint64_t x) {
        if (x < 128) {
            b.push_back((char)x);
            return;
        }
        char tmp[8];
        int n = 0;
        while (x) {
            tmp[n++] = (char)(x & 0xFF);
            x >>= 8;
        }
        b.push_back((char)(256 - n));
        while (n) b.push_back(tmp[--n]);
    }
    void i(int64_t x) { u(x < 0 ? ((~(uint64_t)x) << 1) | 1 : (uint64_t)x << 1); }
    void f(double d) {
        uint64_t bits, rev = 0;
        memcpy(&bits, &d, 8);
        for (int k = 0; k < 8; k++) rev |= ((bits >> (8 * k)) & 0xFF) << (8 * (7 - k));
        u(rev);
    }
    void s(const std::string &x) {
        u(x.size());
        b += x;
    }
    void s(const char *p, size_t n) {
        u(n);
        b.append(p, n);
    }
};

// struct value writer: field deltas, zero values are simply not put (gob omits them)
struct Fields {
    Buf &w;
    int prev = -1;
    explicit Fields(Buf &w_) : w(w_) {}
    void at(int ix) {
        w.u((uint64_t)(ix - prev));
        prev = ix;
    }
    void put_int(int ix, int64_t v) {
        if (v == 0) return;
        at(ix);
        w.i(v);
    }
    void put_bool(int ix, bool v) {
        if (!v) return;
        at(ix);
        w.u(1);
    }
    void put_float(int ix, double v) {
        if (v == 0.0) return;
        at(ix);
        w.f(v);
    }
    void put_str(int ix, const std::string &v) {
        if (v.empty()) return;
        at(ix);
        w.s(v);
    }
    void end() { w.u(0); }
};

// one gob.Encoder: ids, type definitions, then the value message
struct Encoder {
    int next_id = 65;
    std::set<int> sent;
    std::string out;

    void assign(Type *t) {
        if (t->kind == Type::Basic || t->id) return;
        if (t->kind == Type::Struct) {
            t->id = next_id++;
            for (auto &f : t->fields) assign(f.second);
        } else if (t->kind == Type::Slice) {
            assign(t->elem);
            t->id = next_id++;
        } else {
            assign(t->key);
            assign(t->elem);
            t->id = next_id++;
        }
    }
    void message(const std::string &payload) {
        Buf h;
        h.u(payload.size());
        out += h.b;
        out += payload;
    }
    static void common(Buf &w, const Type *t) {  // CommonType{Name, Id}
        if (!t->name.empty()) {
            w.u(1);
            w.s(t->name);
            w.u(1);
        } else {
            w.u(2);
        }
        w.i(t->id);
        w.u(0);
    }
    void send_type(Type *t) {
        if (t->kind == Type::Basic || sent.count(t->id)) return;
        sent.insert(t->id);
        Buf w;
        w.i(-t->id);
        if (t->kind == Type::Slice) {  // wireType.SliceT (field 1): sliceType{CommonType, Elem}
            w.u(2);
            w.u(1);
            common(w, t);
            w.u(1);
            w.i(t->elem->id);
            w.u(0);
            w.u(0);
        } else if (t->kind == Type::Struct) {  // wireType.StructT (field 2): structType{CommonType, Field []fieldType}
            w.u(3);
            w.u(1);
            common(w, t);
            if (!t->fields.empty()) {
                w.u(1);
                w.u(t->fields.size());
                for (auto &f : t->fields) {
                    w.u(1);
                    w.s(f.first);
                    w.u(1);
                    w.i(f.second->id);
                    w.u(0);
                }
            }
            w.u(0);
            w.u(0);
        } else {  // wireType.MapT (field 3): mapType{CommonType, Key, Elem}
            w.u(4);
            w.u(1);
            common(w, t);
            w.u(1);
            w.i(t->key->id);
            w.u(1);
            w.i(t->elem->id);
            w.u(0);
            w.u(0);
        }
        message(w.b);
        if (t->kind == Type::Struct) {
            for (auto &f : t->fields) send_type(f.second);
        } else if (t->kind == Type::Slice) {
            send_type(t->elem);
        } else {
            send_type(t->key);
            send_type(t->elem);
        }
    }
    // top-level struct value: `body` is its field encoding incl. the terminating 0
    std::string finish(Type *top, const std::string &body) {
        assign(top);
        send_type(top);
        Buf w;
        w.i(top->id);
        message(w.b + body);
        return out;
    }
};

}  // namespace gobenc
}  // namespace sybl
