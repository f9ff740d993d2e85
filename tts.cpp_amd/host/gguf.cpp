#include "gguf.h"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>

size_t gguf_type_row_bytes(int type, int64_t n) {
    switch (type) {
        case 0: return (size_t) n * 4;
        case 1: return (size_t) n * 2;
        case 2: return (size_t) (n / 32) * 18;
        case 6: return (size_t) (n / 32) * 22;
        case 8: return (size_t) (n / 32) * 34;
        default: return 0;
    }
}

gguf_file::~gguf_file() {
    if (map) munmap(map, map_size);
}

namespace {
struct cursor {
    const uint8_t * p;
    const uint8_t * end;
    bool            ok = true;
    template <typename T> T rd() {
        T v{};
        if ((size_t) (end - p) < sizeof(T)) { ok = false; return v; }
        memcpy(&v, p, sizeof(T));
        p += sizeof(T);
        return v;
    }
    std::string str() {
        const uint64_t n = rd<uint64_t>();
        if (!ok || n > (uint64_t) (end - p)) { ok = false; return {}; }   // compared as sizes: a hostile length must not wrap the pointer
        std::string s((const char *) p, (size_t) n);
        p += n;
        return s;
    }
};

size_t scalar_size(gguf_vtype t) {
    switch (t) {
        case GGUF_U8: case GGUF_I8: case GGUF_BOOL: return 1;
        case GGUF_U16: case GGUF_I16: return 2;
        case GGUF_U32: case GGUF_I32: case GGUF_F32: return 4;
        case GGUF_U64: case GGUF_I64: case GGUF_F64: return 8;
        default: return 0;
    }
}

bool read_scalar(cursor & c, gguf_vtype t, gguf_value & v) {
    switch (t) {
        case GGUF_U8: v.u = c.rd<uint8_t>(); break;
        case GGUF_I8: v.u = (uint64_t) (int64_t) c.rd<int8_t>(); break;
        case GGUF_U16: v.u = c.rd<uint16_t>(); break;
        case GGUF_I16: v.u = (uint64_t) (int64_t) c.rd<int16_t>(); break;
        case GGUF_U32: v.u = c.rd<uint32_t>(); break;
        case GGUF_I32: v.u = (uint64_t) (int64_t) c.rd<int32_t>(); break;
        case GGUF_BOOL: v.u = c.rd<uint8_t>() != 0; break;
        case GGUF_U64: v.u = c.rd<uint64_t>(); break;
        case GGUF_I64: v.u = (uint64_t) c.rd<int64_t>(); break;
        case GGUF_F32: v.f = c.rd<float>(); break;
        case GGUF_F64: v.f = c.rd<double>(); break;
        default: return false;
    }
    if (t == GGUF_F32 || t == GGUF_F64) v.u = (uint64_t) v.f; else v.f = (double) v.u;
    return c.ok;
}
}  // namespace

static std::shared_ptr<gguf_file> gguf_open_impl(const char * path, std::string & err);

std::shared_ptr<gguf_file> gguf_file::open(const char * path, std::string & err) {
    try {
        return gguf_open_impl(path, err);
    } catch (const std::exception & e) {   // bad_alloc / length_error from a hostile count that passed the size checks
        err = std::string("GGUF parse failed: ") + e.what();
        return nullptr;
    }
}

static std::shared_ptr<gguf_file> gguf_open_impl(const char * path, std::string & err) {
    auto f = std::make_shared<gguf_file>();
    f->path = path;
    const int fd = ::open(path, O_RDONLY);
    if (fd < 0) { err = std::string("cannot open ") + path; return nullptr; }
    struct stat st {};
    if (fstat(fd, &st) != 0) { close(fd); err = "fstat failed"; return nullptr; }
    const uint8_t * base = nullptr;
    if (getenv("OLLAMA_NO_MMAP")) {  // same opt-out switch as the reference (loaders.cpp:45)
        f->owned.resize((size_t) st.st_size);
        size_t off = 0;
        while (off < f->owned.size()) {
            const ssize_t n = read(fd, f->owned.data() + off, f->owned.size() - off);
            if (n <= 0) { close(fd); err = "read failed"; return nullptr; }
            off += (size_t) n;
        }
        base = f->owned.data();
    } else {
        f->map_size = (size_t) st.st_size;
        f->map = mmap(nullptr, f->map_size, PROT_READ, MAP_PRIVATE, fd, 0);
        if (f->map == MAP_FAILED) { f->map = nullptr; close(fd); err = "mmap failed"; return nullptr; }
        base = (const uint8_t *) f->map;
    }
    close(fd);
    cursor c{base, base + st.st_size};
    if (st.st_size < 24 || memcmp(base, "GGUF", 4) != 0) { err = "not a GGUF file"; return nullptr; }
    c.p += 4;
    f->version = c.rd<uint32_t>();
    if (f->version != 2 && f->version != 3) { err = "unsupported GGUF version " + std::to_string(f->version); return nullptr; }
    const uint64_t n_tensors = c.rd<uint64_t>();
    const uint64_t n_kv = c.rd<uint64_t>();
    for (uint64_t i = 0; i < n_kv && c.ok; i++) {
        const size_t rec_begin = (size_t) (c.p - base);
        std::string  key = c.str();
        gguf_value   v;
        v.type = (gguf_vtype) c.rd<uint32_t>();
        if (v.type == GGUF_STR) {
            v.s = c.str();
        } else if (v.type == GGUF_ARR) {
            v.elem_type = (gguf_vtype) c.rd<uint32_t>();
            v.arr_n = c.rd<uint64_t>();
            if (v.elem_type == GGUF_STR) {
                // a string is at least its 8-byte length: a count the rest of the file cannot hold is corrupt (and must not reach reserve)
                if (v.arr_n > (uint64_t) (c.end - c.p) / 8) { c.ok = false; break; }
                v.arr_s.reserve((size_t) v.arr_n);
                for (uint64_t j = 0; j < v.arr_n && c.ok; j++) v.arr_s.push_back(c.str());
            } else {
                const size_t es = scalar_size(v.elem_type);
                if (es == 0 || v.arr_n > (uint64_t) (c.end - c.p) / es) { c.ok = false; break; }
                v.arr_data = c.p;
                c.p += es * v.arr_n;
            }
        } else if (!read_scalar(c, v.type, v)) {
            c.ok = false;
        }
        f->kv_order.push_back({key, rec_begin, (size_t) (c.p - base)});
        f->kv.emplace(std::move(key), std::move(v));
    }
    if (!c.ok) { err = "truncated or corrupt GGUF metadata"; return nullptr; }
    struct info { std::string name; int n_dims; int64_t ne[4]; int type; uint64_t off; };
    std::vector<info> infos;
    // a tensor record is at least 8 (name length) + 4 (n_dims) + 8 (one dimension) + 4 (type) + 8 (offset) bytes
    if (n_tensors > (uint64_t) (c.end - c.p) / 32) { err = "truncated or corrupt GGUF tensor table"; return nullptr; }
    infos.reserve((size_t) n_tensors);
    for (uint64_t i = 0; i < n_tensors && c.ok; i++) {
        info t{};
        t.name = c.str();
        t.n_dims = (int) c.rd<uint32_t>();
        if (t.n_dims < 1 || t.n_dims > 4) { c.ok = false; break; }
        for (int d = 0; d < 4; d++) t.ne[d] = 1;
        for (int d = 0; d < t.n_dims; d++) t.ne[d] = (int64_t) c.rd<uint64_t>();
        t.type = (int) c.rd<uint32_t>();
        t.off = c.rd<uint64_t>();
        infos.push_back(std::move(t));
    }
    if (!c.ok) { err = "truncated or corrupt GGUF tensor table"; return nullptr; }
    uint64_t align = 32;
    if (auto a = f->get("general.alignment")) align = a->u ? a->u : 32;
    f->data_offset = ((size_t) (c.p - base) + align - 1) / align * align;
    f->tensor_names.reserve(infos.size());
    for (auto & t : infos) f->tensor_names.push_back(t.name);
    for (size_t i = 0; i < infos.size(); i++) {
        const info & t = infos[i];
        gguf_tensor_view v{};
        v.name = f->tensor_names[i].c_str();
        v.type = t.type;
        v.n_dims = t.n_dims;
        int64_t rows = 1;
        bool    dims_ok = true;
        for (int d = 0; d < 4; d++) {
            v.ne[d] = t.ne[d];
            // no tensor of the file can hold more elements than the file has bits; products are checked, not trusted
            if (t.ne[d] <= 0 || t.ne[d] > (int64_t) st.st_size * 8) dims_ok = false;
            else if (d > 0 && __builtin_mul_overflow(rows, t.ne[d], &rows)) dims_ok = false;
        }
        // block formats hold 32 consecutive elements of ne[0] per block (Q4_0 / Q5_0 / Q8_0): a ragged row has no encoding
        if (dims_ok && (t.type == 2 || t.type == 6 || t.type == 8) && t.ne[0] % 32 != 0) {
            err = "tensor '" + t.name + "' has a quantised type with ne[0] = " + std::to_string(t.ne[0]) + " not a multiple of 32";
            return nullptr;
        }
        size_t row_bytes = dims_ok ? gguf_type_row_bytes(t.type, t.ne[0]) : 0;
        if (dims_ok && row_bytes && __builtin_mul_overflow(row_bytes, (size_t) rows, &v.nbytes)) dims_ok = false;
        if (!dims_ok) { err = "tensor '" + t.name + "' has impossible dimensions"; return nullptr; }
        v.nbytes = row_bytes * (size_t) rows;
        if (v.nbytes == 0) { err = "tensor '" + t.name + "' has unsupported type " + std::to_string(t.type); return nullptr; }
        if (f->data_offset > (size_t) st.st_size || t.off > (size_t) st.st_size - f->data_offset || v.nbytes > (size_t) st.st_size - f->data_offset - t.off) { err = "tensor '" + t.name + "' runs past the end of the file"; return nullptr; }
        v.data = base + f->data_offset + t.off;
        f->tensors.push_back(v);
    }
    return f;
}

const gguf_value * gguf_file::get_any(std::initializer_list<const char *> keys) const {
    for (const char * k : keys)
        if (auto v = get(k)) return v;
    return nullptr;
}

bool gguf_file::get_u32(std::initializer_list<const char *> keys, uint32_t & out) const {
    if (auto v = get_any(keys)) { out = (uint32_t) v->u; return true; }
    return false;
}
