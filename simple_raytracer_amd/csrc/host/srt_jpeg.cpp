// simple_raytracer_amd/csrc/host/srt_jpeg.cpp -- JPEG decoder of the texture loader (host side, no GPU work).
//
// The reference loads diffuse maps with stbi_load(path, &w, &h, &ch, 3) (Object.cpp:57; stb_image v2.30 vendored
// under library/stb-master), and its tree / grass / horse / cat assets are JPEGs -- baseline 4:4:4 with restart
// intervals, baseline 4:2:0 and one progressive file.  Texel bytes feed the shading path (softShadow:350-361), so the
// decoder must give stb_image's bytes, not just "a correct JPEG decode": decoders differ in the inverse DCT, the
// chroma up-sampling filter and the YCbCr conversion, all of which are integer recipes here.
//
// Entropy decoding follows ITU-T T.81 (Annex F sequential Huffman, Annex G progressive: spectral selection and
// successive approximation), which fixes the coefficients for any conforming decoder.  The numeric steps restate
// stb_image's published algorithm (public domain, file:line of the vendored copy):
//   * dequantisation in 16-bit (coefficient x table entry truncated to int16)          stb_image.h:2227,3073-3078
//   * inverse DCT: jidctint-derived 12-bit fixed point, column pass >> 10 with +512, row pass >> 17 with
//     +65536 + (128 << 17), clamp to 0..255                                              stb_image.h:2430-2519
//   * up-sampling: h2v1 / h1v2 triangle filters, h2v2 3:1 in both directions (>> 4 with +8), nearest
//     otherwise, centred per the JFIF convention, driven row by row                       stb_image.h:3465-3527,3646-3656,3912-3950
//   * YCbCr -> RGB in 20-bit fixed point with the reduced-precision constants             stb_image.h:3658-3685
//   * RGB detection by component ids / Adobe APP14 transform, grey -> replicated, CMYK/YCCK blend        :3283-3286,3882,3951-3990
// Checked byte for byte against the reference's stbi_load on its own JPEG assets and on small synthetic files
// (tests/test_host_mirror.py, tests/golden/jpeg_*.npz).
#include "srt_host.h"

#include <cstdint>
#include <cstring>
#include <vector>

namespace srt_host {
namespace {

const uint8_t ZIGZAG[64] = { 0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
                             35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63 };

// T.81 Annex C: canonical code table; F.2.2.3: decode by code length with MINCODE / MAXCODE / VALPTR
struct HuffTable {
    bool present = false;
    uint8_t vals[256];
    int32_t mincode[17], maxcode[17], valptr[17];
    uint8_t look_len[512];      // 9-bit prefix -> code length (0 = longer than 9 bits)
    uint8_t look_val[512];
    HuffTable() {               // a table a scan names without a DHT having defined it decodes nothing
        std::memset(vals, 0, sizeof vals); std::memset(look_len, 0, sizeof look_len); std::memset(look_val, 0, sizeof look_val);
        for (int i = 0; i < 17; i++) { mincode[i] = 0; maxcode[i] = -1; valptr[i] = 0; }
    }
    bool build(const uint8_t counts[16], const uint8_t* symbols, int n) {
        std::memcpy(vals, symbols, (size_t)n);
        std::memset(look_len, 0, sizeof look_len);
        int32_t code = 0; int k = 0;
        for (int len = 1; len <= 16; len++) {
            valptr[len] = k; mincode[len] = code;
            for (int i = 0; i < counts[len - 1]; i++, k++, code++) {
                if (len <= 9) {
                    const int lo = code << (9 - len), hi = lo + (1 << (9 - len));
                    if (hi > 512) return false;
                    for (int c = lo; c < hi; c++) { look_len[c] = (uint8_t)len; look_val[c] = vals[k]; }
                }
            }
            maxcode[len] = counts[len - 1] ? code - 1 : -1;
            if (code > (1 << len)) return false;            // over-subscribed
            code <<= 1;
        }
        present = true;
        return true;
    }
};

struct Component {
    int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0;
    int x = 0, y = 0;             // samples the component really has
    int w2 = 0, h2 = 0;           // allocated plane: whole interleaved MCUs
    int dc_pred = 0;
    std::vector<uint8_t> plane;   // w2 * h2 samples after the inverse DCT
    std::vector<int16_t> coef;    // progressive: (w2/8) * (h2/8) blocks of 64, natural (row-major) order
};

class Jpeg {
public:
    Jpeg(const uint8_t* d, size_t n) : p_(d), end_(d + n) {}
    bool decode(Texture& out);

private:
    // ---- byte level -------------------------------------------------------------------------------
    const uint8_t* p_; const uint8_t* end_;
    bool eof() const { return p_ >= end_; }
    int u8() { return p_ < end_ ? *p_++ : 0; }
    int u16() { const int a = u8(); return (a << 8) | u8(); }
    // next marker code, 0xFF fill bytes skipped; 0 if the next byte is not a marker
    int next_marker() {
        if (pending_marker_) { const int m = pending_marker_; pending_marker_ = 0; return m; }
        int x = u8();
        if (x != 0xFF) return 0;
        while (x == 0xFF) x = u8();
        return x;
    }

    // ---- entropy-coded segment bit reader (T.81 F.2.2.5: 0xFF00 -> 0xFF, a marker ends the data: zeros follow) --
    uint32_t acc_ = 0; int nbits_ = 0; int pending_marker_ = 0; bool drained_ = false;
    void refill() {
        while (nbits_ <= 24) {
            int b = 0;
            if (!drained_) {
                if (p_ >= end_) drained_ = true;
                else {
                    b = *p_++;
                    if (b == 0xFF) {
                        int c = u8();
                        while (c == 0xFF) c = u8();                  // fill bytes
                        if (c != 0) { pending_marker_ = c; drained_ = true; b = 0; }
                    }
                }
            }
            acc_ |= (uint32_t)b << (24 - nbits_);
            nbits_ += 8;
        }
    }
    int bit() { if (nbits_ < 1) refill(); const int b = (int)(acc_ >> 31); acc_ <<= 1; nbits_--; return b; }
    int bits(int n) {
        if (n == 0) return 0;
        if (nbits_ < n) refill();
        const int v = (int)(acc_ >> (32 - n)); acc_ <<= n; nbits_ -= n; return v;
    }
    // RECEIVE + EXTEND (F.2.2.1, F.2.2.4)
    int receive_extend(int n) {
        if (n == 0) return 0;
        const int v = bits(n);
        return v < (1 << (n - 1)) ? v - (1 << n) + 1 : v;
    }
    int decode_symbol(const HuffTable& t) {
        if (nbits_ < 16) refill();
        const int pre = (int)(acc_ >> 23);
        if (t.look_len[pre]) { const int l = t.look_len[pre]; acc_ <<= l; nbits_ -= l; return t.look_val[pre]; }
        int32_t code = (int32_t)(acc_ >> 22);        // first 10 bits
        for (int len = 10; len <= 16; len++) {
            if (t.maxcode[len] >= 0 && code <= t.maxcode[len] && code >= t.mincode[len]) {
                acc_ <<= len; nbits_ -= len;
                return t.vals[t.valptr[len] + (code - t.mincode[len])];
            }
            code = (int32_t)(acc_ >> (32 - (len + 1)));
        }
        return -1;
    }
    void reset_entropy() {
        acc_ = 0; nbits_ = 0; drained_ = false; eob_run_ = 0;
        for (Component& c : comp_) c.dc_pred = 0;
    }

    // ---- tables and frame state ---------------------------------------------------------------------
    uint16_t quant_[4][64] = {};      // natural order
    HuffTable dc_[4], ac_[4];
    std::vector<Component> comp_;
    int W_ = 0, H_ = 0, hmax_ = 1, vmax_ = 1, mcus_x_ = 0, mcus_y_ = 0;
    bool progressive_ = false, have_frame_ = false, jfif_ = false;
    int adobe_transform_ = -1, rgb_ids_ = 0, restart_interval_ = 0;
    // scan state
    int scan_n_ = 0, order_[4] = {}, ss_ = 0, se_ = 63, ah_ = 0, al_ = 0, eob_run_ = 0;

    bool segment(int m);
    bool frame_header(int m);
    bool scan_header();
    bool scan_data();
    bool block_sequential(Component& c, int16_t blk[64]);
    bool block_dc_progressive(Component& c, int16_t* blk);
    bool block_ac_progressive(Component& c, int16_t* blk);
    bool block_ac_refine(int16_t* blk);
    bool restart_boundary(int& todo);
    static void idct(uint8_t* out, int stride, const int16_t d[64]);
    void finish_progressive();
    void to_rgb(Texture& out);
};

// ---- marker segments ----------------------------------------------------------------------------------
bool Jpeg::segment(int m) {
    if (m == 0xDD) {                                   // DRI
        if (u16() != 4) return false;
        restart_interval_ = u16();
        return true;
    }
    if (m == 0xDB) {                                   // DQT
        int L = u16() - 2;
        while (L > 0) {
            const int q = u8(), wide = q >> 4, t = q & 15;
            if (wide > 1 || t > 3) return false;
            for (int i = 0; i < 64; i++) quant_[t][ZIGZAG[i]] = (uint16_t)(wide ? u16() : u8());
            L -= wide ? 129 : 65;
        }
        return L == 0;
    }
    if (m == 0xC4) {                                   // DHT
        int L = u16() - 2;
        while (L > 0) {
            const int q = u8(), cls = q >> 4, id = q & 15;
            if (cls > 1 || id > 3) return false;
            uint8_t counts[16], symbols[256]; int n = 0;
            for (int i = 0; i < 16; i++) { counts[i] = (uint8_t)u8(); n += counts[i]; }
            if (n > 256) return false;
            for (int i = 0; i < n; i++) symbols[i] = (uint8_t)u8();
            if (!(cls ? ac_[id] : dc_[id]).build(counts, symbols, n)) return false;
            L -= 17 + n;
        }
        return L == 0;
    }
    if ((m >= 0xE0 && m <= 0xEF) || m == 0xFE) {       // APPn / COM
        int L = u16();
        if (L < 2) return false;
        L -= 2;
        if (m == 0xE0 && L >= 5) {
            static const uint8_t tag[5] = { 'J', 'F', 'I', 'F', 0 };
            bool ok = true;
            for (int i = 0; i < 5; i++) if (u8() != tag[i]) ok = false;
            L -= 5;
            if (ok) jfif_ = true;
        } else if (m == 0xEE && L >= 12) {
            static const uint8_t tag[6] = { 'A', 'd', 'o', 'b', 'e', 0 };
            bool ok = true;
            for (int i = 0; i < 6; i++) if (u8() != tag[i]) ok = false;
            L -= 6;
            if (ok) { u8(); u16(); u16(); adobe_transform_ = u8(); L -= 6; }
        }
        if (end_ - p_ < L) return false;
        p_ += L;
        return true;
    }
    return false;
}

bool Jpeg::frame_header(int m) {
    progressive_ = (m == 0xC2);
    const int Lf = u16();
    if (Lf < 11 || u8() != 8) return false;            // 8-bit samples only
    H_ = u16(); W_ = u16();
    if (H_ == 0 || W_ == 0 || H_ > (1 << 24) || W_ > (1 << 24)) return false;
    const int n = u8();
    if (n != 1 && n != 3 && n != 4) return false;
    if (Lf != 8 + 3 * n) return false;
    comp_.assign((size_t)n, Component());
    rgb_ids_ = 0;
    for (int i = 0; i < n; i++) {
        Component& c = comp_[i];
        c.id = u8();
        if (n == 3 && c.id == "RGB"[i]) rgb_ids_++;
        const int q = u8();
        c.h = q >> 4; c.v = q & 15; c.tq = u8();
        if (c.h < 1 || c.h > 4 || c.v < 1 || c.v > 4 || c.tq > 3) return false;
        if (c.h > hmax_) hmax_ = c.h;
        if (c.v > vmax_) vmax_ = c.v;
    }
    for (const Component& c : comp_) if (hmax_ % c.h || vmax_ % c.v) return false;
    if ((uint64_t)W_ * (uint64_t)H_ * (uint64_t)n > (1ull << 30)) return false;
    mcus_x_ = (W_ + hmax_ * 8 - 1) / (hmax_ * 8);
    mcus_y_ = (H_ + vmax_ * 8 - 1) / (vmax_ * 8);
    for (Component& c : comp_) {
        c.x = (W_ * c.h + hmax_ - 1) / hmax_;
        c.y = (H_ * c.v + vmax_ - 1) / vmax_;
        c.w2 = mcus_x_ * c.h * 8; c.h2 = mcus_y_ * c.v * 8;
        c.plane.assign((size_t)c.w2 * c.h2, 0);
        if (progressive_) c.coef.assign((size_t)c.w2 * c.h2, 0);
    }
    have_frame_ = true;
    return true;
}

bool Jpeg::scan_header() {
    const int Ls = u16();
    scan_n_ = u8();
    if (scan_n_ < 1 || scan_n_ > 4 || scan_n_ > (int)comp_.size() || Ls != 6 + 2 * scan_n_) return false;
    for (int i = 0; i < scan_n_; i++) {
        const int id = u8(), q = u8();
        int which = -1;
        for (size_t k = 0; k < comp_.size(); k++) if (comp_[k].id == id) { which = (int)k; break; }
        if (which < 0) return false;
        comp_[which].td = q >> 4; comp_[which].ta = q & 15;
        if (comp_[which].td > 3 || comp_[which].ta > 3) return false;
        order_[i] = which;
    }
    ss_ = u8(); se_ = u8();
    const int a = u8();
    ah_ = a >> 4; al_ = a & 15;
    if (progressive_) {
        if (ss_ > 63 || se_ > 63 || ss_ > se_ || ah_ > 13 || al_ > 13) return false;
    } else {
        if (ss_ != 0 || ah_ != 0 || al_ != 0) return false;
        se_ = 63;
    }
    return true;
}

// ---- block decoders -------------------------------------------------------------------------------------
bool Jpeg::block_sequential(Component& c, int16_t blk[64]) {          // F.2.2, dequantised on the fly
    const HuffTable& hd = dc_[c.td]; const HuffTable& ha = ac_[c.ta];
    const uint16_t* q = quant_[c.tq];
    std::memset(blk, 0, 64 * sizeof(int16_t));
    const int t = decode_symbol(hd);
    if (t < 0 || t > 15) return false;
    c.dc_pred += receive_extend(t);
    blk[0] = (int16_t)(c.dc_pred * q[0]);
    for (int k = 1; k < 64;) {
        const int rs = decode_symbol(ha);
        if (rs < 0) return false;
        const int r = rs >> 4, s = rs & 15;
        if (s == 0) {
            if (r != 15) break;                                         // EOB
            k += 16;
        } else {
            k += r;
            if (k > 63) return false;
            const int z = ZIGZAG[k++];
            blk[z] = (int16_t)(receive_extend(s) * q[z]);
        }
    }
    return true;
}

bool Jpeg::block_dc_progressive(Component& c, int16_t* blk) {           // G.1.2.1
    if (se_ != 0) return false;
    if (ah_ == 0) {
        std::memset(blk, 0, 64 * sizeof(int16_t));
        const int t = decode_symbol(dc_[c.td]);
        if (t < 0 || t > 15) return false;
        c.dc_pred += receive_extend(t);
        blk[0] = (int16_t)(c.dc_pred * (1 << al_));
    } else if (bit()) {
        blk[0] = (int16_t)(blk[0] + (1 << al_));
    }
    return true;
}

bool Jpeg::block_ac_progressive(Component& c, int16_t* blk) {           // G.1.2.2 first pass of a band
    if (ss_ == 0) return false;
    if (ah_ != 0) return block_ac_refine(blk) ;
    if (eob_run_) { eob_run_--; return true; }
    const HuffTable& ha = ac_[c.ta];
    for (int k = ss_; k <= se_;) {
        const int rs = decode_symbol(ha);
        if (rs < 0) return false;
        const int r = rs >> 4, s = rs & 15;
        if (s == 0) {
            if (r < 15) { eob_run_ = (1 << r) - 1 + (r ? bits(r) : 0); break; }
            k += 16;
        } else {
            k += r;
            if (k > 63) return false;
            blk[ZIGZAG[k++]] = (int16_t)(receive_extend(s) * (1 << al_));
        }
    }
    return true;
}

bool Jpeg::block_ac_refine(int16_t* blk) {                               // G.1.2.3 successive approximation
    const int16_t delta = (int16_t)(1 << al_);
    auto correct = [&](int16_t& v) {                                      // one correction bit for a nonzero history
        if (bit() && (v & delta) == 0) v = (int16_t)(v > 0 ? v + delta : v - delta);
    };
    int k = ss_;
    if (eob_run_ == 0) {
        const HuffTable& ha = ac_[comp_[order_[0]].ta];
        while (k <= se_) {
            const int rs = decode_symbol(ha);
            if (rs < 0) return false;
            int r = rs >> 4;
            const int s = rs & 15;
            int16_t fresh = 0;
            if (s == 0) {
                if (r < 15) { eob_run_ = (1 << r) + (r ? bits(r) : 0); break; }
            } else {
                if (s != 1) return false;
                fresh = bit() ? delta : (int16_t)-delta;
            }
            // skip r zero-history coefficients, correcting the nonzero ones passed on the way
            for (; k <= se_; k++) {
                int16_t& v = blk[ZIGZAG[k]];
                if (v != 0) correct(v);
                else if (r == 0) { v = fresh; k++; break; }      // s == 0 (ZRL): the 16th zero, left at 0
                else r--;
            }
        }
    }
    if (eob_run_ > 0) {                                                   // rest of the band: corrections only
        for (; k <= se_; k++) { int16_t& v = blk[ZIGZAG[k]]; if (v != 0) correct(v); }
        eob_run_--;
    }
    return true;
}

// restart interval bookkeeping after each MCU; false = stop the scan here (what was decoded is kept)
bool Jpeg::restart_boundary(int& todo) {
    if (--todo > 0) return true;
    if (nbits_ < 24) refill();
    if (!(pending_marker_ >= 0xD0 && pending_marker_ <= 0xD7)) return false;
    pending_marker_ = 0;
    reset_entropy();
    todo = restart_interval_ ? restart_interval_ : 0x7fffffff;
    return true;
}

bool Jpeg::scan_data() {
    reset_entropy();
    pending_marker_ = 0;
    int todo = restart_interval_ ? restart_interval_ : 0x7fffffff;
    int16_t blk[64];
    if (scan_n_ == 1) {                                // non-interleaved: the component's own blocks, row by row
        Component& c = comp_[order_[0]];
        const int bw = (c.x + 7) >> 3, bh = (c.y + 7) >> 3, cw = c.w2 / 8;
        for (int j = 0; j < bh; j++)
            for (int i = 0; i < bw; i++) {
                if (!progressive_) {
                    if (!block_sequential(c, blk)) return false;
                    idct(&c.plane[(size_t)c.w2 * j * 8 + i * 8], c.w2, blk);
                } else {
                    int16_t* b = &c.coef[64 * ((size_t)i + (size_t)j * cw)];
                    if (!(ss_ == 0 ? block_dc_progressive(c, b) : block_ac_progressive(c, b))) return false;
                }
                if (!restart_boundary(todo)) return true;
            }
        return true;
    }
    for (int j = 0; j < mcus_y_; j++)                  // interleaved MCUs
        for (int i = 0; i < mcus_x_; i++) {
            for (int k = 0; k < scan_n_; k++) {
                Component& c = comp_[order_[k]];
                for (int y = 0; y < c.v; y++)
                    for (int x = 0; x < c.h; x++) {
                        const int bx = i * c.h + x, by = j * c.v + y;
                        if (!progressive_) {
                            if (!block_sequential(c, blk)) return false;
                            idct(&c.plane[(size_t)c.w2 * by * 8 + bx * 8], c.w2, blk);
                        } else {
                            if (!block_dc_progressive(c, &c.coef[64 * ((size_t)bx + (size_t)by * (c.w2 / 8))])) return false;
                        }
                    }
            }
            if (!restart_boundary(todo)) return true;
        }
    return true;
}

// ---- inverse DCT (stb_image.h:2430-2519) -------------------------------------------------------------------
// All arithmetic in uint32_t: for sane coefficients it is the signed arithmetic of the original, for corrupt ones it wraps
// like two's complement hardware does instead of being undefined.  sra() is the arithmetic right shift of the signed value.
inline uint32_t fx(float x) { return (uint32_t)(int32_t)((double)x * 4096 + 0.5); }
inline int32_t sra(uint32_t v, int n) { return (int32_t)v >> n; }
struct Idct1D { uint32_t x0, x1, x2, x3, t0, t1, t2, t3; };
inline Idct1D idct_1d(int32_t s0_, int32_t s1_, int32_t s2_, int32_t s3_, int32_t s4_, int32_t s5_, int32_t s6_, int32_t s7_) {
    static const uint32_t C0 = fx(0.5411961f), C1 = fx(-1.847759065f), C2 = fx(0.765366865f), C3 = fx(1.175875602f),
                          C4 = fx(0.298631336f), C5 = fx(2.053119869f), C6 = fx(3.072711026f), C7 = fx(1.501321110f),
                          C8 = fx(-0.899976223f), C9 = fx(-2.562915447f), C10 = fx(-1.961570560f), C11 = fx(-0.390180644f);
    const uint32_t s0 = (uint32_t)s0_, s1 = (uint32_t)s1_, s2 = (uint32_t)s2_, s3 = (uint32_t)s3_,
                   s4 = (uint32_t)s4_, s5 = (uint32_t)s5_, s6 = (uint32_t)s6_, s7 = (uint32_t)s7_;
    Idct1D r;
    uint32_t p1 = (s2 + s6) * C0;
    uint32_t t2 = p1 + s6 * C1, t3 = p1 + s2 * C2;
    uint32_t t0 = (s0 + s4) * 4096u, t1 = (s0 - s4) * 4096u;
    r.x0 = t0 + t3; r.x3 = t0 - t3; r.x1 = t1 + t2; r.x2 = t1 - t2;
    t0 = s7; t1 = s5; t2 = s3; t3 = s1;
    uint32_t p3 = t0 + t2, p4 = t1 + t3;
    p1 = t0 + t3; uint32_t p2 = t1 + t2;
    const uint32_t p5 = (p3 + p4) * C3;
    t0 *= C4; t1 *= C5; t2 *= C6; t3 *= C7;
    p1 = p5 + p1 * C8; p2 = p5 + p2 * C9; p3 *= C10; p4 *= C11;
    r.t3 = t3 + p1 + p4; r.t2 = t2 + p2 + p3; r.t1 = t1 + p2 + p4; r.t0 = t0 + p1 + p3;
    return r;
}
inline uint8_t clamp8(int x) { return (uint8_t)(x < 0 ? 0 : (x > 255 ? 255 : x)); }

void Jpeg::idct(uint8_t* out, int stride, const int16_t d[64]) {
    int32_t v[64];
    for (int i = 0; i < 8; i++) {                      // columns, 2 extra bits kept
        Idct1D r = idct_1d(d[i], d[8 + i], d[16 + i], d[24 + i], d[32 + i], d[40 + i], d[48 + i], d[56 + i]);
        r.x0 += 512; r.x1 += 512; r.x2 += 512; r.x3 += 512;
        v[i] = sra(r.x0 + r.t3, 10);      v[56 + i] = sra(r.x0 - r.t3, 10);
        v[8 + i] = sra(r.x1 + r.t2, 10);  v[48 + i] = sra(r.x1 - r.t2, 10);
        v[16 + i] = sra(r.x2 + r.t1, 10); v[40 + i] = sra(r.x2 - r.t1, 10);
        v[24 + i] = sra(r.x3 + r.t0, 10); v[32 + i] = sra(r.x3 - r.t0, 10);
    }
    for (int i = 0; i < 8; i++, out += stride) {       // rows: >> 17 rounds, +128 level shift folded in
        const int32_t* w = v + 8 * i;
        Idct1D r = idct_1d(w[0], w[1], w[2], w[3], w[4], w[5], w[6], w[7]);
        const uint32_t bias = 65536u + (128u << 17);
        r.x0 += bias; r.x1 += bias; r.x2 += bias; r.x3 += bias;
        out[0] = clamp8(sra(r.x0 + r.t3, 17)); out[7] = clamp8(sra(r.x0 - r.t3, 17));
        out[1] = clamp8(sra(r.x1 + r.t2, 17)); out[6] = clamp8(sra(r.x1 - r.t2, 17));
        out[2] = clamp8(sra(r.x2 + r.t1, 17)); out[5] = clamp8(sra(r.x2 - r.t1, 17));
        out[3] = clamp8(sra(r.x3 + r.t0, 17)); out[4] = clamp8(sra(r.x3 - r.t0, 17));
    }
}

void Jpeg::finish_progressive() {                      // stb_image.h:3080-3097: dequantise in int16, then IDCT
    for (Component& c : comp_) {
        const int bw = (c.x + 7) >> 3, bh = (c.y + 7) >> 3, cw = c.w2 / 8;
        const uint16_t* q = quant_[c.tq];
        for (int j = 0; j < bh; j++)
            for (int i = 0; i < bw; i++) {
                int16_t* b = &c.coef[64 * ((size_t)i + (size_t)j * cw)];
                for (int k = 0; k < 64; k++) b[k] = (int16_t)(b[k] * q[k]);
                idct(&c.plane[(size_t)c.w2 * j * 8 + i * 8], c.w2, b);
            }
    }
}

// ---- up-sampling and colour conversion (stb_image.h:3465-3527, 3646-3685, 3899-3990) --------------------------
inline uint8_t blend8(uint8_t x, uint8_t y) { const unsigned t = x * y + 128; return (uint8_t)((t + (t >> 8)) >> 8); }

void Jpeg::to_rgb(Texture& out) {
    const int n = (int)comp_.size();
    out.dim.x = W_; out.dim.y = H_;
    out.rgb.assign((size_t)W_ * H_ * 3, 0);
    struct Up { int hs, vs, ystep, ypos, w_lo; size_t line0, line1; };
    std::vector<Up> up((size_t)n);
    std::vector<std::vector<uint8_t>> line((size_t)n, std::vector<uint8_t>((size_t)W_ + 8));
    std::vector<const uint8_t*> row((size_t)n);
    for (int k = 0; k < n; k++) {
        Up& u = up[k];
        u.hs = hmax_ / comp_[k].h; u.vs = vmax_ / comp_[k].v;
        u.ystep = u.vs >> 1; u.ypos = 0; u.w_lo = (W_ + u.hs - 1) / u.hs; u.line0 = u.line1 = 0;
    }
    const bool is_rgb = n == 3 && (rgb_ids_ == 3 || (adobe_transform_ == 0 && !jfif_));
    static const int CR_R = ((int)(1.40200f * 4096.0f + 0.5f)) << 8, CR_G = -(((int)(0.71414f * 4096.0f + 0.5f)) << 8),
                     CB_G = -(((int)(0.34414f * 4096.0f + 0.5f)) << 8), CB_B = ((int)(1.77200f * 4096.0f + 0.5f)) << 8;
    auto ycc = [&](uint8_t* o, const uint8_t* y, const uint8_t* cb, const uint8_t* cr) {
        for (int i = 0; i < W_; i++, o += 3) {
            const int yf = (y[i] << 20) + (1 << 19), r_ = cr[i] - 128, b_ = cb[i] - 128;
            int r = yf + r_ * CR_R;
            int g = yf + r_ * CR_G + (int)((unsigned)(b_ * CB_G) & 0xffff0000u);
            int b = yf + b_ * CB_B;
            r >>= 20; g >>= 20; b >>= 20;
            o[0] = clamp8(r); o[1] = clamp8(g); o[2] = clamp8(b);
        }
    };
    for (int j = 0; j < H_; j++) {
        for (int k = 0; k < n; k++) {
            Up& u = up[k];
            const Component& c = comp_[k];
            const bool bottom = u.ystep >= (u.vs >> 1);
            const uint8_t* near_ = &c.plane[bottom ? u.line1 : u.line0];
            const uint8_t* far_ = &c.plane[bottom ? u.line0 : u.line1];
            uint8_t* o = line[k].data();
            const int w = u.w_lo;
            if (u.hs == 1 && u.vs == 1) {
                row[k] = near_;
            } else if (u.hs == 1 && u.vs == 2) {
                for (int i = 0; i < w; i++) o[i] = (uint8_t)((3 * near_[i] + far_[i] + 2) >> 2);
                row[k] = o;
            } else if (u.hs == 2 && u.vs == 1) {
                if (w == 1) { o[0] = o[1] = near_[0]; }
                else {
                    o[0] = near_[0];
                    o[1] = (uint8_t)((near_[0] * 3 + near_[1] + 2) >> 2);
                    int i = 1;
                    for (; i < w - 1; i++) {
                        const int t = 3 * near_[i] + 2;
                        o[2 * i] = (uint8_t)((t + near_[i - 1]) >> 2);
                        o[2 * i + 1] = (uint8_t)((t + near_[i + 1]) >> 2);
                    }
                    o[2 * i] = (uint8_t)((near_[w - 2] * 3 + near_[w - 1] + 2) >> 2);
                    o[2 * i + 1] = near_[w - 1];
                }
                row[k] = o;
            } else if (u.hs == 2 && u.vs == 2) {
                if (w == 1) { o[0] = o[1] = (uint8_t)((3 * near_[0] + far_[0] + 2) >> 2); }
                else {
                    int t1 = 3 * near_[0] + far_[0];
                    o[0] = (uint8_t)((t1 + 2) >> 2);
                    for (int i = 1; i < w; i++) {
                        const int t0 = t1;
                        t1 = 3 * near_[i] + far_[i];
                        o[2 * i - 1] = (uint8_t)((3 * t0 + t1 + 8) >> 4);
                        o[2 * i] = (uint8_t)((3 * t1 + t0 + 8) >> 4);
                    }
                    o[2 * w - 1] = (uint8_t)((t1 + 2) >> 2);
                }
                row[k] = o;
            } else {
                for (int i = 0; i < w; i++) for (int s = 0; s < u.hs; s++) o[i * u.hs + s] = near_[i];
                row[k] = o;
            }
            if (++u.ystep >= u.vs) {
                u.ystep = 0;
                u.line0 = u.line1;
                if (++u.ypos < c.y) u.line1 += (size_t)c.w2;
            }
        }
        uint8_t* o = &out.rgb[(size_t)j * W_ * 3];
        if (n == 3) {
            if (is_rgb) for (int i = 0; i < W_; i++) { o[3 * i] = row[0][i]; o[3 * i + 1] = row[1][i]; o[3 * i + 2] = row[2][i]; }
            else ycc(o, row[0], row[1], row[2]);
        } else if (n == 4) {
            if (adobe_transform_ == 0) {                                   // CMYK
                for (int i = 0; i < W_; i++) { const uint8_t m = row[3][i]; o[3 * i] = blend8(row[0][i], m); o[3 * i + 1] = blend8(row[1][i], m); o[3 * i + 2] = blend8(row[2][i], m); }
            } else {
                ycc(o, row[0], row[1], row[2]);
                if (adobe_transform_ == 2)                                 // YCCK
                    for (int i = 0; i < W_; i++) { const uint8_t m = row[3][i]; for (int c = 0; c < 3; c++) o[3 * i + c] = blend8((uint8_t)(255 - o[3 * i + c]), m); }
            }
        } else {
            for (int i = 0; i < W_; i++) o[3 * i] = o[3 * i + 1] = o[3 * i + 2] = row[0][i];
        }
    }
}

bool Jpeg::decode(Texture& out) {
    if (next_marker() != 0xD8) return false;           // SOI
    int m = next_marker();
    while (!(m == 0xC0 || m == 0xC1 || m == 0xC2)) {   // tables / APPn up to the frame header
        if (m == 0) {                                  // padding between segments
            if (eof()) return false;
            m = next_marker();
            continue;
        }
        if (!segment(m)) return false;
        m = next_marker();
    }
    if (!frame_header(m)) return false;
    m = next_marker();
    bool truncated = false;
    while (m != 0xD9) {                                // until EOI
        if (m == 0xDA) {
            if (!scan_header() || !scan_data()) return false;
            if (!pending_marker_) {                    // skip anything up to the next real marker
                while (!eof()) {
                    int x = u8();
                    while (x == 0xFF) {
                        if (eof()) break;
                        x = u8();
                        if (x != 0x00 && x != 0xFF) { pending_marker_ = x; break; }
                    }
                    if (pending_marker_) break;
                }
            }
            m = next_marker();
            if (m >= 0xD0 && m <= 0xD7) m = next_marker();
        } else if (m == 0xDC) {                        // DNL
            if (u16() != 4 || u16() != H_) return false;
            m = next_marker();
        } else {
            if (!segment(m)) { truncated = true; break; }   // what was decoded so far is returned
            m = next_marker();
        }
    }
    (void)truncated;
    if (progressive_) finish_progressive();
    to_rgb(out);
    return true;
}

} // namespace

bool decode_jpeg(const std::vector<unsigned char>& d, Texture& t) {
    if (d.size() < 4 || d[0] != 0xFF || d[1] != 0xD8) return false;
    Jpeg j(d.data(), d.size());
    Texture tmp;
    if (!j.decode(tmp)) return false;
    t = std::move(tmp);
    return true;
}

} // namespace srt_host
