/*
 * ss_msgpack.h -- the small subset of MessagePack the SEND-SLAM wire protocol uses
 * (SURVEY.md section 8(b)), dependency-free (msgpack-c is not in this image).
 *
 * Decoder: nil, bool, ints, float32/64, str, bin, array, map -> a value tree; maps keep
 * arrival order and lookups are by key, so parsing is order-insensitive exactly like the
 * reference's ParseMessage (/root/reference/slam_backends/orb_slam_3/
 * orbslam3_mono_networked.cc:295-336).  Encoder: what SendPosePacket (:225-282) emits --
 * fixmap, fixstr/str8, float64 and integers in msgpack-c's smallest-encoding rule.
 */
#ifndef SS_MSGPACK_H
#define SS_MSGPACK_H

#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace ssmp {

enum class type { NIL, BOOL, INT, UINT, FLOAT, STR, BIN, ARRAY, MAP };

struct value {
    type t = type::NIL;
    bool b = false;
    int64_t i = 0;
    uint64_t u = 0;
    double f = 0;
    const uint8_t *data = nullptr; /* STR / BIN: view into the payload */
    size_t size = 0;
    std::vector<value> arr;
    std::vector<std::pair<value, value>> map;

    std::string str() const { return std::string((const char *)data, size); }
    const value *find(const char *key) const
    {
        const size_t n = strlen(key);
        for (const auto &kv : map)
            if (kv.first.t == type::STR && kv.first.size == n && memcmp(kv.first.data, key, n) == 0) return &kv.second;
        return nullptr;
    }
    /* msgpack-c object::as<T>() semantics: ints convert among themselves, floats accept ints,
     * anything else throws (the reference turns that into "Failed to parse key") */
    double as_double() const
    {
        if (t == type::FLOAT) return f;
        if (t == type::INT) return (double)i;
        if (t == type::UINT) return (double)u;
        throw std::runtime_error("std::bad_cast");
    }
    int as_int() const
    {
        if (t == type::INT) {
            if (i < INT32_MIN || i > INT32_MAX) throw std::runtime_error("std::bad_cast");
            return (int)i;
        }
        if (t == type::UINT) {
            if (u > (uint64_t)INT32_MAX) throw std::runtime_error("std::bad_cast");
            return (int)u;
        }
        throw std::runtime_error("std::bad_cast");
    }
    std::string as_string() const
    {
        if (t == type::STR || t == type::BIN) return str();
        throw std::runtime_error("std::bad_cast");
    }
};

class decoder {
  public:
    decoder(const uint8_t *p, size_t n) : p_(p), end_(p + n) {}
    value parse()
    {
        value v = one(0);
        return v;
    }
    size_t consumed(const uint8_t *start) const { return (size_t)(p_ - start); }

  private:
    const uint8_t *p_, *end_;
    void need(size_t n)
    {
        if ((size_t)(end_ - p_) < n) throw std::runtime_error("insufficient bytes");
    }
    uint64_t be(int n)
    {
        need((size_t)n);
        uint64_t v = 0;
        for (int k = 0; k < n; k++) v = (v << 8) | *p_++;
        return v;
    }
    value blob(type t, size_t n)
    {
        need(n);
        value v;
        v.t = t;
        v.data = p_;
        v.size = n;
        p_ += n;
        return v;
    }
    value seq(bool is_map, size_t n, int depth)
    {
        value v;
        v.t = is_map ? type::MAP : type::ARRAY;
        for (size_t k = 0; k < n; k++) {
            if (is_map) {
                value key = one(depth + 1);
                value val = one(depth + 1);
                v.map.emplace_back(std::move(key), std::move(val));
            } else
                v.arr.push_back(one(depth + 1));
        }
        return v;
    }
    value one(int depth)
    {
        if (depth > 32) throw std::runtime_error("depth limit");
        need(1);
        const uint8_t c = *p_++;
        value v;
        if (c <= 0x7f) { v.t = type::UINT; v.u = c; return v; }
        if (c >= 0xe0) { v.t = type::INT; v.i = (int8_t)c; return v; }
        if (c >= 0xa0 && c <= 0xbf) return blob(type::STR, c & 0x1f);
        if (c >= 0x90 && c <= 0x9f) return seq(false, c & 0x0f, depth);
        if (c >= 0x80 && c <= 0x8f) return seq(true, c & 0x0f, depth);
        switch (c) {
        case 0xc0: return v;
        case 0xc2: v.t = type::BOOL; v.b = false; return v;
        case 0xc3: v.t = type::BOOL; v.b = true; return v;
        case 0xc4: return blob(type::BIN, (size_t)be(1));
        case 0xc5: return blob(type::BIN, (size_t)be(2));
        case 0xc6: return blob(type::BIN, (size_t)be(4));
        case 0xca: { uint32_t r = (uint32_t)be(4); float f; memcpy(&f, &r, 4); v.t = type::FLOAT; v.f = f; return v; }
        case 0xcb: { uint64_t r = be(8); double d; memcpy(&d, &r, 8); v.t = type::FLOAT; v.f = d; return v; }
        case 0xcc: v.t = type::UINT; v.u = be(1); return v;
        case 0xcd: v.t = type::UINT; v.u = be(2); return v;
        case 0xce: v.t = type::UINT; v.u = be(4); return v;
        case 0xcf: v.t = type::UINT; v.u = be(8); return v;
        case 0xd0: v.t = type::INT; v.i = (int8_t)be(1); return v;
        case 0xd1: v.t = type::INT; v.i = (int16_t)be(2); return v;
        case 0xd2: v.t = type::INT; v.i = (int32_t)be(4); return v;
        case 0xd3: v.t = type::INT; v.i = (int64_t)be(8); return v;
        case 0xd9: return blob(type::STR, (size_t)be(1));
        case 0xda: return blob(type::STR, (size_t)be(2));
        case 0xdb: return blob(type::STR, (size_t)be(4));
        case 0xdc: return seq(false, (size_t)be(2), depth);
        case 0xdd: return seq(false, (size_t)be(4), depth);
        case 0xde: return seq(true, (size_t)be(2), depth);
        case 0xdf: return seq(true, (size_t)be(4), depth);
        default: throw std::runtime_error("unsupported MessagePack type byte");
        }
    }
};

class packer {
  public:
    std::vector<uint8_t> buf;
    void pack_map(uint32_t n)
    {
        if (n < 16) buf.push_back((uint8_t)(0x80 | n));
        else { buf.push_back(0xde); be(n, 2); }
    }
    void pack(const char *s)
    {
        const size_t n = strlen(s);
        if (n < 32) buf.push_back((uint8_t)(0xa0 | n));
        else if (n < 256) { buf.push_back(0xd9); buf.push_back((uint8_t)n); }
        else { buf.push_back(0xda); be(n, 2); }
        buf.insert(buf.end(), s, s + n);
    }
    void pack(double d)
    {
        uint64_t r;
        memcpy(&r, &d, 8);
        buf.push_back(0xcb);
        be(r, 8);
    }
    void pack(int v) /* msgpack-c pack_imp_int32: smallest encoding */
    {
        if (v < -(1 << 5)) {
            if (v < -(1 << 15)) { buf.push_back(0xd2); be((uint32_t)v, 4); }
            else if (v < -(1 << 7)) { buf.push_back(0xd1); be((uint16_t)v, 2); }
            else { buf.push_back(0xd0); buf.push_back((uint8_t)v); }
        } else if (v < (1 << 7)) {
            buf.push_back((uint8_t)v);
        } else if (v < (1 << 8)) { buf.push_back(0xcc); buf.push_back((uint8_t)v); }
        else if (v < (1 << 16)) { buf.push_back(0xcd); be((uint16_t)v, 2); }
        else { buf.push_back(0xce); be((uint32_t)v, 4); }
    }

  private:
    void be(uint64_t v, int n)
    {
        for (int k = n - 1; k >= 0; k--) buf.push_back((uint8_t)(v >> (8 * k)));
    }
};

} // namespace ssmp

#endif
