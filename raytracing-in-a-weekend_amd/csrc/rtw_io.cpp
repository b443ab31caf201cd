// rtw_io.cpp -- host I/O on either side of the render path:
//   * the reference's scene wire format, the `json` object
//       {"spheres":[{"origin":{x,y,z},"radius":r,"col_mod":{x,y,z},
//                    "material":{"metallicness":m,"opacity":o,"ir":i},
//                    "velocity":{x,y,z},"texture":{"row":w,"col":h,"img":[{x,y,z},...]}}]}
//     written/read by Rust/src/viewport.rs:174-205 (Scene), objects/sphere.rs:44-90 (Sphere),
//     objects/materials.rs:21-54 (Material: `emmited` is NOT serialised and reads back as 0),
//     vec3.rs:22-50 (Vec3), texture.rs:28-59,268-276 (ImageTexture).  The C++ dialect
//     (C++/headers/scene.h:17-67, sphere.h:28-91) has no velocity/texture: both are optional on input.
//   * the image writers: 8-bit PNG with write_img_f32's quantisation (Rust/src/write_img.rs:6-19) and the
//     C++ P3 PPM writer (C++/src/ppm_writer.cpp:12-27, truncating int(255 c), C++/src/RGB.cpp:16-20).
// Pure host code, no GPU, no third-party library (the PNG uses stored deflate blocks).
#include "rtw_host.h"

#include <cctype>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace {

// ---- a tiny JSON reader (objects, arrays, numbers, strings, literals) -----------------------------
struct JVal {
    enum Kind { Null, Num, Str, Arr, Obj, Bool } kind = Null;
    double num = 0;
    std::string str;
    std::vector<JVal> arr;
    std::vector<std::pair<std::string, JVal>> obj;
    const JVal *get(const char *key) const {
        if (kind != Obj) return nullptr;
        for (const auto &kv : obj) if (kv.first == key) return &kv.second;
        return nullptr;
    }
};

struct JParser {
    const char *p, *end;
    bool ok = true;
    void ws() { while (p < end && std::isspace((unsigned char)*p)) p++; }
    bool eat(char c) { ws(); if (p < end && *p == c) { p++; return true; } return false; }
    bool lit(const char *w) {                      // a literal, only when all of it lies inside [p, end)
        const size_t n = std::strlen(w);
        if ((size_t)(end - p) < n || std::memcmp(p, w, n) != 0) return false;
        p += n; return true;
    }
    std::string string() {
        std::string s;
        if (!eat('"')) { ok = false; return s; }
        while (p < end && *p != '"') {
            if (*p == '\\' && p + 1 < end) { p++; s.push_back(*p == 'n' ? '\n' : *p == 't' ? '\t' : *p); p++; }
            else s.push_back(*p++);
        }
        if (p >= end) ok = false; else p++;
        return s;
    }
    JVal value(int depth = 0) {
        JVal v;
        ws();
        if (p >= end || depth > 64) { ok = false; return v; }
        if (*p == '{') {
            p++; v.kind = JVal::Obj;
            if (eat('}')) return v;
            do {
                std::string k = string();
                if (!ok || !eat(':')) { ok = false; return v; }
                v.obj.emplace_back(k, value(depth + 1));
                if (!ok) return v;
            } while (eat(','));
            if (!eat('}')) ok = false;
        } else if (*p == '[') {
            p++; v.kind = JVal::Arr;
            if (eat(']')) return v;
            do { v.arr.push_back(value(depth + 1)); if (!ok) return v; } while (eat(','));
            if (!eat(']')) ok = false;
        } else if (*p == '"') {
            v.kind = JVal::Str; v.str = string();
        } else if (lit("true")) { v.kind = JVal::Bool; v.num = 1; }
        else if (lit("false")) { v.kind = JVal::Bool; }
        else if (lit("null")) { }
        else {
            // the number token is copied into a bounded, NUL-terminated buffer first: strtod on the caller's text would scan past
            // `end` when the buffer is not NUL-terminated and ends in digits
            char tok[64]; size_t n = 0;
            while (p + n < end && n + 1 < sizeof tok && (std::isdigit((unsigned char)p[n]) || (p[n] && std::strchr("+-.eE", p[n])))) n++;
            std::memcpy(tok, p, n); tok[n] = 0;
            char *e = nullptr;
            v.num = n ? std::strtod(tok, &e) : 0.0;
            if (!n || e == tok) { ok = false; return v; }
            v.kind = JVal::Num; p += e - tok;
        }
        return v;
    }
};

bool read_f32(const JVal *v, float &out) {              // JsonValue::as_f32
    if (!v || v->kind != JVal::Num) return false;
    out = (float)v->num;
    return true;
}
bool read_vec3(const JVal *v, float out[3]) {           // Vec3::try_from (vec3.rs:30-50)
    return v && read_f32(v->get("x"), out[0]) && read_f32(v->get("y"), out[1]) && read_f32(v->get("z"), out[2]);
}

void put_f32(std::string &s, float f) {
    char b[40];
    if (f == (float)(long long)f && std::fabs(f) < 1e15f) std::snprintf(b, sizeof b, "%lld", (long long)f);   // the json crate prints 1.0 as 1
    else std::snprintf(b, sizeof b, "%.9g", (double)f);                                                      // shortest exact f32 round trip
    s += b;
}
void put_vec3(std::string &s, const float v[3]) {
    s += "{\"x\":"; put_f32(s, v[0]); s += ",\"y\":"; put_f32(s, v[1]); s += ",\"z\":"; put_f32(s, v[2]); s += "}";
}

// ---- PNG: stored-deflate zlib stream ------------------------------------------------------------------
uint32_t crc_table[256];
void crc_init() {
    static bool done = false;
    if (done) return;
    for (uint32_t n = 0; n < 256; n++) { uint32_t c = n; for (int k = 0; k < 8; k++) c = c & 1 ? 0xEDB88320u ^ (c >> 1) : c >> 1; crc_table[n] = c; }
    done = true;
}
uint32_t crc32(const uint8_t *d, size_t n, uint32_t c = 0) {
    crc_init(); c = ~c;
    for (size_t i = 0; i < n; i++) c = crc_table[(c ^ d[i]) & 0xFF] ^ (c >> 8);
    return ~c;
}
void be32(std::vector<uint8_t> &v, uint32_t x) { v.push_back(x >> 24); v.push_back(x >> 16); v.push_back(x >> 8); v.push_back(x); }
void chunk(std::vector<uint8_t> &png, const char *type, const std::vector<uint8_t> &data) {
    be32(png, (uint32_t)data.size());
    std::vector<uint8_t> td(type, type + 4);
    td.insert(td.end(), data.begin(), data.end());
    png.insert(png.end(), td.begin(), td.end());
    be32(png, crc32(td.data(), td.size()));
}

} // namespace

extern "C" {

// Serialise a scene to the reference's JSON.  Returns the length needed (excluding the NUL); writes at most
// cap bytes (NUL-terminated when cap > 0).  Call with buf == NULL to size the buffer.
size_t rtw_scene_to_json(const RtwScene *sc, char *buf, size_t cap) {
    if (!sc) return 0;
    std::string s = "{\"spheres\":[";
    for (uint32_t i = 0; i < sc->n_spheres; i++) {
        const RtwSphere &sp = sc->spheres[i];
        if (i) s += ",";
        s += "{\"origin\":"; put_vec3(s, sp.center);
        s += ",\"radius\":"; put_f32(s, sp.radius);
        s += ",\"col_mod\":"; put_vec3(s, sp.col_mod);
        s += ",\"material\":{\"metallicness\":"; put_f32(s, sp.metallicness);
        s += ",\"opacity\":"; put_f32(s, sp.opacity); s += ",\"ir\":"; put_f32(s, sp.ir); s += "}";
        s += ",\"velocity\":"; put_vec3(s, sp.velocity);
        s += ",\"texture\":{";
        if (sp.tex >= 0 && (uint32_t)sp.tex < sc->n_textures) {
            const RtwTexture &t = sc->textures[sp.tex];
            s += "\"row\":" + std::to_string(t.row) + ",\"col\":" + std::to_string(t.col) + ",\"img\":[";
            for (uint32_t k = 0; k < t.row * t.col; k++) { if (k) s += ","; put_vec3(s, sc->texels + 3 * (size_t)(t.texel_offset + k)); }
        } else {
            s += "\"row\":1,\"col\":1,\"img\":["; put_vec3(s, sp.tex_color);       // ImageTexture::from_color
        }
        s += "]}}";
    }
    s += "]}";
    if (buf && cap) { size_t n = s.size() < cap - 1 ? s.size() : cap - 1; std::memcpy(buf, s.data(), n); buf[n] = 0; }
    return s.size();
}

// Parse the reference's JSON into caller arrays (count-query pattern like rtw_scene_generate: pass
// spheres == NULL to get the sizes).  Returns RTW_OK, or RTW_E_INVALID on a malformed document / missing
// mandatory member (the reference returns ParseError there) / insufficient capacity.
int rtw_scene_from_json(const char *text, size_t len,
                        RtwSphere *spheres, uint32_t sphere_cap, uint32_t *n_spheres,
                        RtwTexture *textures, uint32_t texture_cap, uint32_t *n_textures,
                        float *texels, uint32_t texel_cap, uint32_t *n_texels) {
    if (!text) return RTW_E_INVALID;
    JParser jp{ text, text + len };
    JVal root = jp.value();
    jp.ws();
    if (!jp.ok || jp.p != jp.end) return RTW_E_INVALID;
    const JVal *arr = root.get("spheres");
    if (!arr || arr->kind != JVal::Arr) return RTW_E_INVALID;           // viewport.rs:186-188
    std::vector<RtwSphere> sp; std::vector<RtwTexture> tx; std::vector<float> tl;
    for (const JVal &v : arr->arr) {
        RtwSphere s; std::memset(&s, 0, sizeof s);
        const JVal *mat = v.get("material");
        if (!read_vec3(v.get("origin"), s.center) || !read_f32(v.get("radius"), s.radius) || !read_vec3(v.get("col_mod"), s.col_mod) ||
            !mat || !read_f32(mat->get("metallicness"), s.metallicness) || !read_f32(mat->get("opacity"), s.opacity) || !read_f32(mat->get("ir"), s.ir))
            return RTW_E_INVALID;
        if (v.get("velocity") && !read_vec3(v.get("velocity"), s.velocity)) return RTW_E_INVALID;   // absent in the C++ dialect
        s.tex = -1;
        s.tex_color[0] = s.tex_color[1] = s.tex_color[2] = 1.0f;         // C++ dialect: no texture, albedo = col_mod
        if (const JVal *t = v.get("texture")) {
            float row = 0, col = 0;
            const JVal *img = t->get("img");
            if (!read_f32(t->get("row"), row) || !read_f32(t->get("col"), col) || !img || img->kind != JVal::Arr) return RTW_E_INVALID;
            // (validated as finite values in [1, 2^31) BEFORE the casts: a negative, NaN or huge f32 -> u32 is undefined behaviour)
            if (!(row >= 1.0f && row < 2147483648.0f) || !(col >= 1.0f && col < 2147483648.0f)) return RTW_E_INVALID;
            const uint32_t w = (uint32_t)row, h = (uint32_t)col;
            if (img->arr.size() < (size_t)w * h) return RTW_E_INVALID;
            if (w == 1 && h == 1) { if (!read_vec3(&img->arr[0], s.tex_color)) return RTW_E_INVALID; }
            else {
                RtwTexture d; d.row = w; d.col = h; d.texel_offset = (uint32_t)(tl.size() / 3); d.emit_tex = 0;
                for (size_t k = 0; k < (size_t)w * h; k++) { float c[3]; if (!read_vec3(&img->arr[k], c)) return RTW_E_INVALID; tl.insert(tl.end(), c, c + 3); }
                s.tex = (int32_t)tx.size(); tx.push_back(d);
            }
        }
        sp.push_back(s);
    }
    if (n_spheres) *n_spheres = (uint32_t)sp.size();
    if (n_textures) *n_textures = (uint32_t)tx.size();
    if (n_texels) *n_texels = (uint32_t)(tl.size() / 3);
    if (!spheres) return RTW_OK;
    if (sphere_cap < sp.size() || texture_cap < tx.size() || texel_cap < tl.size() / 3) return RTW_E_INVALID;
    if ((!tx.empty() && !textures) || (!tl.empty() && !texels)) return RTW_E_INVALID;
    std::copy(sp.begin(), sp.end(), spheres);
    if (!tx.empty()) std::copy(tx.begin(), tx.end(), textures);
    if (!tl.empty()) std::copy(tl.begin(), tl.end(), texels);
    return RTW_OK;
}

// write_img_f32 (Rust/src/write_img.rs:6-19): quantise [h][w][3] f32 and save an 8-bit RGB PNG.
int rtw_write_png_f32(const char *path, const float *rgb, uint32_t width, uint32_t height) {
    if (!path || !rgb || !width || !height) return RTW_E_INVALID;
    std::vector<uint8_t> q((size_t)width * height * 3);
    rtw_quantize_u8(rgb, q.size(), q.data());
    std::vector<uint8_t> raw;
    raw.reserve((size_t)height * (3 * width + 1));
    for (uint32_t y = 0; y < height; y++) { raw.push_back(0); raw.insert(raw.end(), q.begin() + (size_t)y * width * 3, q.begin() + (size_t)(y + 1) * width * 3); }
    std::vector<uint8_t> z = { 0x78, 0x01 };
    uint32_t a = 1, b = 0;
    for (size_t off = 0; off < raw.size() || off == 0; off += 65535) {
        size_t n = raw.size() - off < 65535 ? raw.size() - off : 65535;
        bool last = off + n >= raw.size();
        z.push_back(last ? 1 : 0); z.push_back(n & 0xFF); z.push_back(n >> 8); z.push_back(~n & 0xFF); z.push_back((~n >> 8) & 0xFF);
        z.insert(z.end(), raw.begin() + off, raw.begin() + off + n);
        for (size_t i = off; i < off + n; i++) { a = (a + raw[i]) % 65521u; b = (b + a) % 65521u; }
        if (last) break;
    }
    be32(z, (b << 16) | a);
    std::vector<uint8_t> png = { 0x89, 'P', 'N', 'G', '\r', '\n', 0x1A, '\n' };
    std::vector<uint8_t> ihdr; be32(ihdr, width); be32(ihdr, height); ihdr.insert(ihdr.end(), { 8, 2, 0, 0, 0 });
    chunk(png, "IHDR", ihdr); chunk(png, "IDAT", z); chunk(png, "IEND", {});
    FILE *f = std::fopen(path, "wb");
    if (!f) return RTW_E_INVALID;
    size_t w = std::fwrite(png.data(), 1, png.size(), f);
    std::fclose(f);
    return w == png.size() ? RTW_OK : RTW_E_INVALID;
}

// write_ppm (C++/src/ppm_writer.cpp:12-27): P3 text, "R G B  " per pixel, int(255 c) truncation.
int rtw_write_ppm_f32(const char *path, const float *rgb, uint32_t width, uint32_t height) {
    if (!path || !rgb || !width || !height) return RTW_E_INVALID;
    FILE *f = std::fopen(path, "wb");
    if (!f) return RTW_E_INVALID;
    std::fprintf(f, "P3\n%u %u\n255\n", width, height);
    for (uint32_t y = 0; y < height; y++) {
        for (uint32_t x = 0; x < width; x++) {
            const float *p = rgb + 3 * ((size_t)y * width + x);
            std::fprintf(f, "%d %d %d  ", (int)(255 * (double)p[0]), (int)(255 * (double)p[1]), (int)(255 * (double)p[2]));
        }
        std::fputc('\n', f);
    }
    std::fclose(f);
    return RTW_OK;
}

} // extern "C"
