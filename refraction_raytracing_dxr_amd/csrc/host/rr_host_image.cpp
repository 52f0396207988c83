// rr_host_image.cpp -- float image decode with the call shape and results of
// stbi_loadf(filename, &x, &y, &n, req_comp) as used by load_texture (RefractionDemo.cpp:111).
//
// stb_image (v2.28, vendored by the reference) is third-party source and is NOT copied here; this
// is an independent decoder for the two formats the demo's env map exists in:
//   * Radiance RGBE ".hdr" (flat and new-RLE scanlines), value = mantissa * 2^(e-136)
//   * PNG (1/2/4/8/16-bit gray, RGB, palette, +alpha; plain or Adam7-interlaced), expanded to linear float
//     the way stb does for LDR sources: pow(v/255, 2.2) for colour, v/255 for alpha
// stb_image's other formats (JPEG, BMP, TGA, PSD, GIF, PIC, PNM) are NOT decoded: rr_host_image_loadf returns NULL for them
// (tests/test_host.py::test_image_load_failures); the demo only ever loads an .hdr (RefractionDemo.cpp:527).
// tests/test_host.py checks the decoder bit for bit against the reference header compiled in place (oracle/_ref) on
// envmap.png, on generated PNGs of every colour type and on generated .hdr files.
#include "../../../include/rrdxr.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace {

typedef std::vector<unsigned char> Bytes;

bool read_file(const char* path, Bytes& out)
{
    FILE* f = fopen(path, "rb");
    if (!f) return false;
    unsigned char buf[65536];
    size_t n;
    while ((n = fread(buf, 1, sizeof buf, f)) > 0) out.insert(out.end(), buf, buf + n);
    fclose(f);
    return true;
}

// ------------------------------------------------------------------------------- inflate (RFC 1951)
struct BitReader {
    const unsigned char* p; const unsigned char* end;
    uint32_t acc = 0; int nbits = 0;
    bool need(int n)
    {
        while (nbits < n) {
            if (p >= end) return false;
            acc |= (uint32_t)(*p++) << nbits;
            nbits += 8;
        }
        return true;
    }
    bool bits(int n, uint32_t& v)
    {
        if (n == 0) { v = 0; return true; }
        if (!need(n)) return false;
        v = acc & ((1u << n) - 1u);
        acc >>= n; nbits -= n;
        return true;
    }
    void align_byte() { acc >>= (nbits & 7); nbits -= (nbits & 7); }
};

// canonical Huffman decoder: count[len] + symbols sorted by (len, value)
struct Huff {
    uint16_t count[16];
    uint16_t symbol[320];
    bool build(const unsigned char* lens, int n)
    {
        memset(count, 0, sizeof count);
        for (int i = 0; i < n; ++i) count[lens[i]]++;
        count[0] = 0;
        int left = 1;
        for (int l = 1; l < 16; ++l) { left = (left << 1) - count[l]; if (left < 0) return false; }
        uint16_t offs[16];
        offs[1] = 0;
        for (int l = 1; l < 15; ++l) offs[l + 1] = (uint16_t)(offs[l] + count[l]);
        for (int i = 0; i < n; ++i) if (lens[i]) symbol[offs[lens[i]]++] = (uint16_t)i;
        return true;
    }
    int decode(BitReader& br) const
    {
        int code = 0, first = 0, index = 0;
        for (int l = 1; l < 16; ++l) {
            uint32_t b;
            if (!br.bits(1, b)) return -1;
            code |= (int)b;
            int c = count[l];
            if (code - c < first) return symbol[index + (code - first)];
            index += c; first += c; first <<= 1; code <<= 1;
        }
        return -1;
    }
};

bool inflate_zlib(const Bytes& in, Bytes& out, size_t expected)
{
    if (in.size() < 2) return false;
    if ((in[0] & 0x0f) != 8 || ((in[0] << 8 | in[1]) % 31) != 0 || (in[1] & 0x20)) return false;
    BitReader br{ in.data() + 2, in.data() + in.size() };
    out.reserve(expected);
    static const uint16_t len_base[29] = { 3,4,5,6,7,8,9,10,11,13,15,17,19,23,27,31,35,43,51,59,67,83,99,115,131,163,195,227,258 };
    static const uint16_t len_extra[29] = { 0,0,0,0,0,0,0,0,1,1,1,1,2,2,2,2,3,3,3,3,4,4,4,4,5,5,5,5,0 };
    static const uint16_t dist_base[30] = { 1,2,3,4,5,7,9,13,17,25,33,49,65,97,129,193,257,385,513,769,1025,1537,2049,3073,4097,6145,8193,12289,16385,24577 };
    static const uint16_t dist_extra[30] = { 0,0,0,0,1,1,2,2,3,3,4,4,5,5,6,6,7,7,8,8,9,9,10,10,11,11,12,12,13,13 };
    for (;;) {
        uint32_t final_blk, type;
        if (!br.bits(1, final_blk) || !br.bits(2, type)) return false;
        if (type == 0) {
            br.align_byte();
            uint32_t len, nlen;
            if (!br.bits(16, len) || !br.bits(16, nlen) || (len ^ 0xffffu) != nlen) return false;
            for (uint32_t i = 0; i < len; ++i) { uint32_t b; if (!br.bits(8, b)) return false; out.push_back((unsigned char)b); }
        } else if (type == 1 || type == 2) {
            Huff lit, dist;
            unsigned char lens[320];
            if (type == 1) {
                for (int i = 0; i < 144; ++i) lens[i] = 8;
                for (int i = 144; i < 256; ++i) lens[i] = 9;
                for (int i = 256; i < 280; ++i) lens[i] = 7;
                for (int i = 280; i < 288; ++i) lens[i] = 8;
                lit.build(lens, 288);
                for (int i = 0; i < 30; ++i) lens[i] = 5;
                dist.build(lens, 30);
            } else {
                uint32_t hlit, hdist, hclen;
                if (!br.bits(5, hlit) || !br.bits(5, hdist) || !br.bits(4, hclen)) return false;
                hlit += 257; hdist += 1; hclen += 4;
                if (hlit > 286 || hdist > 30) return false;
                static const unsigned char order[19] = { 16,17,18,0,8,7,9,6,10,5,11,4,12,3,13,2,14,1,15 };
                unsigned char cl[19];
                memset(cl, 0, sizeof cl);
                for (uint32_t i = 0; i < hclen; ++i) { uint32_t v; if (!br.bits(3, v)) return false; cl[order[i]] = (unsigned char)v; }
                Huff clh;
                if (!clh.build(cl, 19)) return false;
                uint32_t idx = 0;
                while (idx < hlit + hdist) {
                    int sym = clh.decode(br);
                    if (sym < 0) return false;
                    if (sym < 16) lens[idx++] = (unsigned char)sym;
                    else {
                        uint32_t rep, prev = 0;
                        if (sym == 16) { if (idx == 0) return false; prev = lens[idx - 1]; if (!br.bits(2, rep)) return false; rep += 3; }
                        else if (sym == 17) { if (!br.bits(3, rep)) return false; rep += 3; }
                        else { if (!br.bits(7, rep)) return false; rep += 11; }
                        if (idx + rep > hlit + hdist) return false;
                        while (rep--) lens[idx++] = (unsigned char)prev;
                    }
                }
                if (lens[256] == 0) return false;
                if (!lit.build(lens, (int)hlit)) return false;
                dist.build(lens + hlit, (int)hdist);    // incomplete distance codes are legal
            }
            for (;;) {
                int sym = lit.decode(br);
                if (sym < 0) return false;
                if (sym < 256) out.push_back((unsigned char)sym);
                else if (sym == 256) break;
                else {
                    sym -= 257;
                    if (sym >= 29) return false;
                    uint32_t eb;
                    if (!br.bits(len_extra[sym], eb)) return false;
                    uint32_t len = len_base[sym] + eb;
                    int ds = dist.decode(br);
                    if (ds < 0 || ds >= 30) return false;
                    if (!br.bits(dist_extra[ds], eb)) return false;
                    size_t d = dist_base[ds] + eb;
                    if (d > out.size()) return false;
                    size_t from = out.size() - d;
                    for (uint32_t i = 0; i < len; ++i) out.push_back(out[from + i]);
                }
            }
        } else {
            return false;
        }
        if (final_blk) break;
    }
    return true;
}

// ------------------------------------------------------------------------------- PNG
uint32_t be32(const unsigned char* p) { return (uint32_t)p[0] << 24 | (uint32_t)p[1] << 16 | (uint32_t)p[2] << 8 | p[3]; }

int paeth(int a, int b, int c)
{
    int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
    if (pa <= pb && pa <= pc) return a;
    return pb <= pc ? b : c;
}

// decodes to 8 bits per channel; channels = 1..4 as stored in the file (palette expanded to 3 or 4)
bool decode_png(const Bytes& file, int& w, int& h, int& channels, Bytes& pix)
{
    static const unsigned char sig[8] = { 137, 80, 78, 71, 13, 10, 26, 10 };
    if (file.size() < 8 || memcmp(file.data(), sig, 8) != 0) return false;
    size_t pos = 8;
    int depth = 0, ctype = 0, interlace = 0;
    bool have_hdr = false;
    Bytes idat, plte, trns;
    while (pos + 12 <= file.size()) {
        uint32_t len = be32(&file[pos]);
        const unsigned char* type = &file[pos + 4];
        if (pos + 12 + (size_t)len > file.size()) return false;
        const unsigned char* data = &file[pos + 8];
        if (!memcmp(type, "IHDR", 4)) {
            if (len != 13) return false;
            w = (int)be32(data); h = (int)be32(data + 4);
            depth = data[8]; ctype = data[9]; interlace = data[12];
            if (data[10] != 0 || data[11] != 0) return false;
            have_hdr = true;
        } else if (!memcmp(type, "PLTE", 4)) plte.assign(data, data + len);
        else if (!memcmp(type, "tRNS", 4)) trns.assign(data, data + len);
        else if (!memcmp(type, "IDAT", 4)) idat.insert(idat.end(), data, data + len);
        else if (!memcmp(type, "IEND", 4)) break;
        pos += 12 + (size_t)len;
    }
    if (!have_hdr || w <= 0 || h <= 0 || w > (1 << 24) || h > (1 << 24) || interlace > 1) return false;
    int file_ch;
    switch (ctype) {
    case 0: file_ch = 1; break;
    case 2: file_ch = 3; break;
    case 3: file_ch = 1; break;
    case 4: file_ch = 2; break;
    case 6: file_ch = 4; break;
    default: return false;
    }
    if (!(depth == 1 || depth == 2 || depth == 4 || depth == 8 || depth == 16)) return false;
    if ((ctype == 2 || ctype == 4 || ctype == 6) && depth < 8) return false;
    if (ctype == 3 && (depth == 16 || plte.size() < 3)) return false;
    const size_t bits_pp = (size_t)file_ch * depth;
    const size_t stride = ((size_t)w * bits_pp + 7) / 8;
    const size_t bpp = bits_pp >= 8 ? bits_pp / 8 : 1;
    // unfilter `rows` scanlines of `line` bytes each (a filter-type byte in front of every one) from src into dst
    auto unfilter = [&](const unsigned char* src_all, unsigned char* dst_all, size_t line, int rows) -> bool {
        for (int y = 0; y < rows; ++y) {
            const unsigned char* src = src_all + (line + 1) * y;
            unsigned char* cur = dst_all + line * y;
            const unsigned char* up = y ? dst_all + line * (y - 1) : nullptr;
            const int ft = src[0];
            ++src;
            for (size_t i = 0; i < line; ++i) {
                int a = i >= bpp ? cur[i - bpp] : 0, b = up ? up[i] : 0, c = (up && i >= bpp) ? up[i - bpp] : 0;
                int v = src[i];
                switch (ft) {
                case 0: break;
                case 1: v += a; break;
                case 2: v += b; break;
                case 3: v += (a + b) >> 1; break;
                case 4: v += paeth(a, b, c); break;
                default: return false;
                }
                cur[i] = (unsigned char)v;
            }
        }
        return true;
    };
    Bytes raw;
    Bytes img(stride * h);
    if (interlace == 0) {
        if (!inflate_zlib(idat, raw, (stride + 1) * h) || raw.size() < (stride + 1) * (size_t)h) return false;
        if (!unfilter(raw.data(), img.data(), stride, h)) return false;
    } else {
        // Adam7: seven reduced images, each filtered on its own, one after the other in the stream; pass p holds the pixels
        // (x0 + i*dx, y0 + j*dy)
        static const int x0[7] = { 0, 4, 0, 2, 0, 1, 0 }, y0[7] = { 0, 0, 4, 0, 2, 0, 1 };
        static const int dx[7] = { 8, 8, 4, 4, 2, 2, 1 }, dy[7] = { 8, 8, 8, 4, 4, 2, 2 };
        size_t total = 0;
        int pw[7], ph[7];
        for (int p = 0; p < 7; ++p) {
            pw[p] = (w - x0[p] + dx[p] - 1) / dx[p]; ph[p] = (h - y0[p] + dy[p] - 1) / dy[p];
            if (pw[p] > 0 && ph[p] > 0) total += (((size_t)pw[p] * bits_pp + 7) / 8 + 1) * ph[p];
        }
        if (!inflate_zlib(idat, raw, total) || raw.size() < total) return false;
        size_t at = 0;
        Bytes sub;
        for (int p = 0; p < 7; ++p) {
            if (pw[p] <= 0 || ph[p] <= 0) continue;
            const size_t line = ((size_t)pw[p] * bits_pp + 7) / 8;
            sub.assign(line * ph[p], 0);
            if (!unfilter(&raw[at], sub.data(), line, ph[p])) return false;
            at += (line + 1) * ph[p];
            for (int j = 0; j < ph[p]; ++j) {
                const unsigned char* srow = &sub[line * j];
                unsigned char* drow = &img[stride * (size_t)(y0[p] + j * dy[p])];
                for (int i = 0; i < pw[p]; ++i) {
                    const size_t x = (size_t)(x0[p] + i * dx[p]);
                    if (bits_pp >= 8) memcpy(drow + x * bpp, srow + (size_t)i * bpp, bpp);
                    else {
                        const size_t sb = (size_t)i * bits_pp, db = x * bits_pp;
                        const unsigned q = (srow[sb >> 3] >> (8 - bits_pp - (sb & 7))) & ((1u << bits_pp) - 1u);
                        const unsigned sh = (unsigned)(8 - bits_pp - (db & 7));
                        drow[db >> 3] = (unsigned char)((drow[db >> 3] & ~(((1u << bits_pp) - 1u) << sh)) | (q << sh));
                    }
                }
            }
        }
    }
    // to 8-bit samples
    const bool pal = ctype == 3;
    const bool pal_alpha = pal && !trns.empty();
    channels = pal ? (pal_alpha ? 4 : 3) : file_ch;
    pix.resize((size_t)w * h * channels);
    for (int y = 0; y < h; ++y) {
        const unsigned char* row = &img[stride * y];
        for (int x = 0; x < w; ++x) {
            unsigned v[4] = { 0, 0, 0, 0 };
            for (int c = 0; c < file_ch; ++c) {
                const size_t s = (size_t)x * file_ch + c;
                if (depth == 8) v[c] = row[s];
                else if (depth == 16) v[c] = row[2 * s];                     // high byte, as an 8-bit load keeps
                else {
                    const size_t bit = s * depth;
                    unsigned q = (row[bit >> 3] >> (8 - depth - (bit & 7))) & ((1u << depth) - 1u);
                    static const unsigned scale[5] = { 0, 255, 85, 0, 17 };
                    v[c] = pal ? q : q * scale[depth];
                }
            }
            unsigned char* o = &pix[((size_t)y * w + x) * channels];
            if (pal) {
                const unsigned i = v[0];
                if ((size_t)i * 3 + 2 < plte.size()) { o[0] = plte[i * 3]; o[1] = plte[i * 3 + 1]; o[2] = plte[i * 3 + 2]; }
                else { o[0] = o[1] = o[2] = 0; }
                if (pal_alpha) o[3] = i < trns.size() ? trns[i] : 255;
            } else {
                for (int c = 0; c < file_ch; ++c) o[c] = (unsigned char)v[c];
            }
        }
    }
    return true;
}

unsigned char luma(int r, int g, int b) { return (unsigned char)(((r * 77) + (g * 150) + (29 * b)) >> 8); }

// channel-count conversion with the weights and fills stb uses
void convert_channels(const Bytes& in, int ic, int oc, size_t n, Bytes& out)
{
    out.resize(n * oc);
    for (size_t i = 0; i < n; ++i) {
        const unsigned char* s = &in[i * ic];
        unsigned char* d = &out[i * oc];
        unsigned char r, g, b, a = 255;
        if (ic <= 2) { r = g = b = s[0]; if (ic == 2) a = s[1]; }
        else { r = s[0]; g = s[1]; b = s[2]; if (ic == 4) a = s[3]; }
        switch (oc) {
        case 1: d[0] = ic <= 2 ? s[0] : luma(r, g, b); break;
        case 2: d[0] = ic <= 2 ? s[0] : luma(r, g, b); d[1] = a; break;
        case 3: d[0] = r; d[1] = g; d[2] = b; break;
        default: d[0] = r; d[1] = g; d[2] = b; d[3] = a; break;
        }
    }
}

// ------------------------------------------------------------------------------- Radiance HDR
struct ByteCursor {
    const Bytes& b; size_t pos = 0;
    bool eof() const { return pos >= b.size(); }
    int get() { return pos < b.size() ? b[pos++] : 0; }
    std::string token()
    {
        std::string t;
        int c = get();
        while (!eof() && c != '\n') {
            t.push_back((char)c);
            if (t.size() == 1023) { while (!eof() && get() != '\n') {} break; }
            c = get();
        }
        return t;
    }
};

void rgbe_to_float(const unsigned char* rgbe, float* out, int req)
{
    if (rgbe[3] != 0) {
        const float f = (float)ldexp(1.0f, rgbe[3] - (128 + 8));
        if (req <= 2) out[0] = (rgbe[0] + rgbe[1] + rgbe[2]) * f / 3;
        else { out[0] = rgbe[0] * f; out[1] = rgbe[1] * f; out[2] = rgbe[2] * f; }
        if (req == 2) out[1] = 1;
        if (req == 4) out[3] = 1;
    } else {
        for (int i = 0; i < req; ++i) out[i] = 0;
        if (req == 2) out[1] = 1;
        if (req == 4) out[3] = 1;
    }
}

float* decode_hdr(const Bytes& file, int* x, int* y, int* comp, int req)
{
    ByteCursor c{ file };
    std::string t = c.token();
    if (t != "#?RADIANCE" && t != "#?RGBE") return nullptr;
    bool ok_format = false;
    for (;;) {
        t = c.token();
        if (t.empty()) break;
        if (t == "FORMAT=32-bit_rle_rgbe") ok_format = true;
    }
    if (!ok_format) return nullptr;
    t = c.token();
    if (t.compare(0, 3, "-Y ") != 0) return nullptr;
    char* e = nullptr;
    const long height = strtol(t.c_str() + 3, &e, 10);
    while (*e == ' ') ++e;
    if (strncmp(e, "+X ", 3) != 0) return nullptr;
    const long width = strtol(e + 3, nullptr, 10);
    if (width <= 0 || height <= 0 || width > (1 << 24) || height > (1 << 24)) return nullptr;
    if (req == 0) req = 3;
    float* data = (float*)malloc((size_t)width * height * req * sizeof(float));
    if (!data) return nullptr;
    *x = (int)width; *y = (int)height;
    if (comp) *comp = 3;
    bool flat = width < 8 || width >= 32768;
    size_t flat_from = 0;           // pixel index the flat reader starts at
    unsigned char first[4];
    bool have_first = false;
    if (!flat) {
        std::vector<unsigned char> line((size_t)width * 4);
        for (long j = 0; j < height && !flat; ++j) {
            int c1 = c.get(), c2 = c.get(), len = c.get();
            if (c1 != 2 || c2 != 2 || (len & 0x80)) {
                // not run-length encoded after all: these bytes are the first pixel of a flat image
                first[0] = (unsigned char)c1; first[1] = (unsigned char)c2; first[2] = (unsigned char)len; first[3] = (unsigned char)c.get();
                have_first = true; flat = true; flat_from = 0;
                break;
            }
            len = (len << 8) | c.get();
            if (len != width) { free(data); return nullptr; }
            for (int k = 0; k < 4; ++k) {
                long i = 0;
                while (i < width) {
                    int count = c.get();
                    const long left = width - i;
                    if (count > 128) {
                        const int value = c.get();
                        count -= 128;
                        if (count == 0 || count > left) { free(data); return nullptr; }
                        for (int z = 0; z < count; ++z) line[(size_t)(i++) * 4 + k] = (unsigned char)value;
                    } else {
                        if (count == 0 || count > left) { free(data); return nullptr; }
                        for (int z = 0; z < count; ++z) line[(size_t)(i++) * 4 + k] = (unsigned char)c.get();
                    }
                }
            }
            for (long i = 0; i < width; ++i) rgbe_to_float(&line[(size_t)i * 4], data + ((size_t)j * width + i) * req, req);
        }
    }
    if (flat) {
        size_t n = (size_t)width * height, i = flat_from;
        if (have_first) { rgbe_to_float(first, data, req); i = 1; }
        for (; i < n; ++i) {
            unsigned char px[4];
            for (int k = 0; k < 4; ++k) px[k] = (unsigned char)c.get();
            rgbe_to_float(px, data + i * req, req);
        }
    }
    return data;
}

} // namespace

extern "C" float* rr_host_image_loadf(const char* filename, int* x, int* y, int* channels_in_file, int req_comp)
{
    if (!filename || !x || !y || req_comp < 0 || req_comp > 4) return nullptr;
    Bytes file;
    if (!read_file(filename, file)) return nullptr;
    int comp = 0;
    if (file.size() >= 7 && (!memcmp(file.data(), "#?RADIA", 7) || !memcmp(file.data(), "#?RGBE\n", 7))) {
        float* r = decode_hdr(file, x, y, &comp, req_comp);
        if (r && channels_in_file) *channels_in_file = comp;
        return r;
    }
    int w, h, ch;
    Bytes pix;
    if (!decode_png(file, w, h, ch, pix)) return nullptr;
    const int oc = req_comp ? req_comp : ch;
    Bytes conv;
    const Bytes* src = &pix;
    if (oc != ch) { convert_channels(pix, ch, oc, (size_t)w * h, conv); src = &conv; }
    float* out = (float*)malloc((size_t)w * h * oc * sizeof(float));
    if (!out) return nullptr;
    // LDR -> linear float: gamma 2.2 on colour channels, alpha (even channel counts) stays linear
    const int ncolor = (oc & 1) ? oc : oc - 1;
    float lut[256];
    for (int v = 0; v < 256; ++v) lut[v] = (float)(pow(v / 255.0f, 2.2f) * 1.0f);
    const size_t n = (size_t)w * h;
    for (size_t i = 0; i < n; ++i) {
        for (int k = 0; k < ncolor; ++k) out[i * oc + k] = lut[(*src)[i * oc + k]];
        if (ncolor < oc) out[i * oc + ncolor] = (*src)[i * oc + ncolor] / 255.0f;
    }
    *x = w; *y = h;
    if (channels_in_file) *channels_in_file = ch;
    return out;
}

extern "C" int rr_host_image_write_hdr(const char* filename, int w, int h, const float* rgb)
{
    if (!filename || !rgb || w <= 0 || h <= 0) return RR_ERR_INVALID_ARGUMENT;
    FILE* f = fopen(filename, "wb");
    if (!f) return RR_ERR_IO;
    fprintf(f, "#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y %d +X %d\n", h, w);
    std::vector<unsigned char> line((size_t)w * 4), outb;
    const bool rle = w >= 8 && w < 32768;
    for (int j = 0; j < h; ++j) {
        for (int i = 0; i < w; ++i) {
            const float* p = rgb + ((size_t)j * w + i) * 3;
            float m = p[0] > p[1] ? p[0] : p[1];
            if (p[2] > m) m = p[2];
            unsigned char* q = &line[(size_t)i * 4];
            if (!(m >= 1e-32f)) { q[0] = q[1] = q[2] = q[3] = 0; continue; }
            int e;
            const float scale = frexpf(m, &e) * 256.0f / m;
            for (int k = 0; k < 3; ++k) { float v = p[k] > 0.0f ? p[k] * scale : 0.0f; q[k] = (unsigned char)(v > 255.0f ? 255 : (int)v); }
            q[3] = (unsigned char)(e + 128);
        }
        outb.clear();
        if (!rle) outb.assign(line.begin(), line.end());
        else {
            outb.push_back(2); outb.push_back(2); outb.push_back((unsigned char)(w >> 8)); outb.push_back((unsigned char)(w & 255));
            for (int k = 0; k < 4; ++k) {
                int i = 0;
                while (i < w) {
                    // run of >= 3 equal bytes -> run packet, else a literal packet up to the next such run
                    int r = 1;
                    while (i + r < w && r < 127 && line[(size_t)(i + r) * 4 + k] == line[(size_t)i * 4 + k]) ++r;
                    if (r >= 3) { outb.push_back((unsigned char)(128 + r)); outb.push_back(line[(size_t)i * 4 + k]); i += r; continue; }
                    int s = i, n = 0;
                    while (i < w && n < 128) {
                        int rr2 = 1;
                        while (i + rr2 < w && rr2 < 3 && line[(size_t)(i + rr2) * 4 + k] == line[(size_t)i * 4 + k]) ++rr2;
                        if (rr2 >= 3) break;
                        ++i; ++n;
                    }
                    outb.push_back((unsigned char)n);
                    for (int z = 0; z < n; ++z) outb.push_back(line[(size_t)(s + z) * 4 + k]);
                }
            }
        }
        if (fwrite(outb.data(), 1, outb.size(), f) != outb.size()) { fclose(f); return RR_ERR_IO; }
    }
    fclose(f);
    return RR_OK;
}
