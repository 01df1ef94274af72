#include "safetensors.hpp"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cerrno>
#include <cstdint>
#include <cstring>

namespace t3mi {

namespace {

// Tiny JSON cursor: enough for the safetensors header (objects, arrays, strings,
// integers).  Values we do not care about are skipped structurally.
struct Cur {
    const char* p;
    const char* e;
    std::string* err;
    bool fail(const char* m) {
        if (err->empty()) *err = std::string("safetensors header: ") + m;
        return false;
    }
    void ws() {
        while (p < e && (*p == ' ' || *p == '\n' || *p == '\t' || *p == '\r')) ++p;
    }
    bool eat(char c) {
        ws();
        if (p < e && *p == c) {
            ++p;
            return true;
        }
        return false;
    }
    bool str(std::string& out) {
        ws();
        if (p >= e || *p != '"') return fail("expected string");
        ++p;
        out.clear();
        while (p < e && *p != '"') {
            if (*p == '\\') {
                if (++p >= e) return fail("bad escape");
                switch (*p) {
                    case 'n': out.push_back('\n'); break;
                    case 't': out.push_back('\t'); break;
                    case 'u':
                        // keep \uXXXX verbatim; tensor names are ASCII in practice
                        out.append("\\u");
                        break;
                    default: out.push_back(*p);
                }
                ++p;
            } else {
                out.push_back(*p++);
            }
        }
        if (p >= e) return fail("unterminated string");
        ++p;
        return true;
    }
    bool integer(int64_t& v) {
        ws();
        bool neg = false;
        if (p < e && *p == '-') {
            neg = true;
            ++p;
        }
        if (p >= e || *p < '0' || *p > '9') return fail("expected integer");
        v = 0;
        while (p < e && *p >= '0' && *p <= '9') {
            if (v > (INT64_MAX - 9) / 10) return fail("integer out of range");
            v = v * 10 + (*p++ - '0');
        }
        if (neg) v = -v;
        return true;
    }
    bool skip() {  // skip any JSON value
        ws();
        if (p >= e) return fail("truncated");
        if (*p == '"') {
            std::string s;
            return str(s);
        }
        if (*p == '{' || *p == '[') {
            char open = *p, close = (open == '{') ? '}' : ']';
            ++p;
            if (eat(close)) return true;
            for (;;) {
                if (open == '{') {
                    std::string k;
                    if (!str(k)) return false;
                    if (!eat(':')) return fail("expected ':'");
                }
                if (!skip()) return false;
                if (eat(',')) continue;
                if (eat(close)) return true;
                return fail("expected ',' or close");
            }
        }
        while (p < e && *p != ',' && *p != '}' && *p != ']') ++p;  // number / literal
        return true;
    }
};

StDtype parse_dtype(const std::string& s) {
    if (s == "F32") return StDtype::F32;
    if (s == "F16") return StDtype::F16;
    if (s == "BF16") return StDtype::BF16;
    if (s == "I64") return StDtype::I64;
    if (s == "I32") return StDtype::I32;
    return StDtype::OTHER;
}

size_t dtype_size(StDtype d) {
    switch (d) {
        case StDtype::F32: return 4;
        case StDtype::F16: return 2;
        case StDtype::BF16: return 2;
        case StDtype::I64: return 8;
        case StDtype::I32: return 4;
        default: return 0;
    }
}

float half_to_float(uint16_t h) {
    uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    uint32_t exp = (h >> 10) & 0x1Fu;
    uint32_t man = h & 0x3FFu;
    uint32_t bits;
    if (exp == 0) {
        if (man == 0) {
            bits = sign;
        } else {  // subnormal
            int sh = 0;
            while (!(man & 0x400u)) {
                man <<= 1;
                ++sh;
            }
            man &= 0x3FFu;
            bits = sign | ((uint32_t)(127 - 15 - sh + 1) << 23) | (man << 13);
        }
    } else if (exp == 31) {
        bits = sign | 0x7F800000u | (man << 13);
    } else {
        bits = sign | ((exp - 15 + 127) << 23) | (man << 13);
    }
    float f;
    std::memcpy(&f, &bits, 4);
    return f;
}

}  // namespace

SafeTensorsFile::~SafeTensorsFile() {
    if (map_ && map_ != MAP_FAILED) munmap(map_, map_len_);
    if (fd_ >= 0) close(fd_);
}

bool SafeTensorsFile::open(const std::string& path, std::string& err) {
    fd_ = ::open(path.c_str(), O_RDONLY);
    if (fd_ < 0) {
        err = "cannot open '" + path + "': " + std::strerror(errno);
        return false;
    }
    struct stat st;
    if (fstat(fd_, &st) != 0 || st.st_size < 8) {
        err = "'" + path + "' is not a safetensors file (too small)";
        return false;
    }
    map_len_ = (size_t)st.st_size;
    mtime_ns_ = (int64_t)st.st_mtim.tv_sec * 1000000000LL + (int64_t)st.st_mtim.tv_nsec;
    map_ = mmap(nullptr, map_len_, PROT_READ, MAP_PRIVATE, fd_, 0);
    if (map_ == MAP_FAILED) {
        err = "mmap failed for '" + path + "'";
        map_ = nullptr;
        return false;
    }
    const uint8_t* base = static_cast<const uint8_t*>(map_);
    uint64_t hlen = 0;
    std::memcpy(&hlen, base, 8);
    if (hlen == 0 || hlen > map_len_ - 8 || hlen > (100u << 20)) {
        err = "'" + path + "': bad safetensors header length";
        return false;
    }
    const uint8_t* data0 = base + 8 + hlen;
    const size_t data_len = map_len_ - 8 - (size_t)hlen;

    Cur c{reinterpret_cast<const char*>(base + 8), reinterpret_cast<const char*>(base + 8 + hlen), &err};
    if (!c.eat('{')) return c.fail("expected '{'");
    if (c.eat('}')) return true;
    for (;;) {
        std::string name;
        if (!c.str(name)) return false;
        if (!c.eat(':')) return c.fail("expected ':'");
        if (name == "__metadata__") {
            if (!c.skip()) return false;
        } else {
            if (!c.eat('{')) return c.fail("expected tensor object");
            StTensor t;
            int64_t off0 = -1, off1 = -1;
            for (;;) {
                std::string k;
                if (!c.str(k)) return false;
                if (!c.eat(':')) return c.fail("expected ':'");
                if (k == "dtype") {
                    std::string d;
                    if (!c.str(d)) return false;
                    t.dtype = parse_dtype(d);
                } else if (k == "shape") {
                    if (!c.eat('[')) return c.fail("expected '['");
                    if (!c.eat(']')) {
                        for (;;) {
                            int64_t v;
                            if (!c.integer(v)) return false;
                            if (v < 0) return c.fail("negative extent in a shape");
                            t.shape.push_back(v);
                            if (c.eat(',')) continue;
                            if (c.eat(']')) break;
                            return c.fail("bad shape");
                        }
                    }
                } else if (k == "data_offsets") {
                    if (!c.eat('[')) return c.fail("expected '['");
                    if (!c.integer(off0)) return false;
                    if (!c.eat(',')) return c.fail("bad data_offsets");
                    if (!c.integer(off1)) return false;
                    if (!c.eat(']')) return c.fail("bad data_offsets");
                } else {
                    if (!c.skip()) return false;
                }
                if (c.eat(',')) continue;
                if (c.eat('}')) break;
                return c.fail("bad tensor object");
            }
            if (off0 < 0 || off1 < off0 || (size_t)off1 > data_len) {
                err = "tensor '" + name + "': data_offsets out of range";
                return false;
            }
            t.data = data0 + off0;
            t.nbytes = (size_t)(off1 - off0);
            {
                int64_t n = 1;
                for (int64_t ext : t.shape) {
                    if (ext != 0 && n > (int64_t(1) << 62) / ext) {
                        err = "tensor '" + name + "': shape overflows";
                        return false;
                    }
                    n *= ext;
                }
            }
            const size_t es = dtype_size(t.dtype);
            if (es && (size_t)t.numel() * es != t.nbytes) {
                err = "tensor '" + name + "': shape does not match byte size";
                return false;
            }
            tensors_.emplace(std::move(name), std::move(t));
        }
        if (c.eat(',')) continue;
        if (c.eat('}')) break;
        return c.fail("expected ',' or '}'");
    }
    return true;
}

const StTensor* SafeTensorsFile::find(const std::string& name) const {
    auto it = tensors_.find(name);
    return it == tensors_.end() ? nullptr : &it->second;
}

bool SafeTensorsFile::to_f32(const StTensor& t, float* dst, std::string& err) {
    const int64_t n = t.numel();
    switch (t.dtype) {
        case StDtype::F32:
            std::memcpy(dst, t.data, (size_t)n * 4);
            return true;
        case StDtype::BF16: {
            const uint16_t* s = reinterpret_cast<const uint16_t*>(t.data);
            for (int64_t i = 0; i < n; ++i) {
                uint32_t b = (uint32_t)s[i] << 16;
                std::memcpy(dst + i, &b, 4);
            }
            return true;
        }
        case StDtype::F16: {
            const uint16_t* s = reinterpret_cast<const uint16_t*>(t.data);
            for (int64_t i = 0; i < n; ++i) dst[i] = half_to_float(s[i]);
            return true;
        }
        default:
            err = "unsupported tensor dtype for float conversion";
            return false;
    }
}

}  // namespace t3mi
