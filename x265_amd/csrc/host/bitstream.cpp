// bitstream.cpp - see bitstream.h.
#include "bitstream.h"

#include <cmath>
#include <cstring>

#include "analysis_core.h"

namespace xmi {
namespace bitstream {

namespace {
inline uint32_t shl(uint32_t v, int s) { return s >= 32 ? 0u : v << s; }
inline uint32_t shr(uint32_t v, int s) { return s >= 32 ? 0u : v >> s; }
}  // namespace

void BitWriter::drain(uint32_t &word, int bits)
{
    for (int i = 0; i < (bits >> 3); i++) {
        const uint32_t b = word >> 24;
        word <<= 8;
        if (end_ - p_ < 2) {      // room for the byte and its escape, or nothing
            overflow_ = true;
            continue;
        }
        if (p_[-1] == 0 && p_[-2] == 0 && b <= 3) {
            *p_++ = 0x03;
            escapes++;
        }
        *p_++ = (uint8_t)b;
    }
}

void BitWriter::put(uint32_t bits, int n)
{
    const int shift = 32 - cached_ - n;
    if (shift >= 0) {
        cache_ |= shl(bits, shift);
        cached_ += n;
    } else {
        uint32_t w = cache_ | shr(bits, -shift);
        drain(w, 32);
        cache_ = shl(bits, 32 + shift);
        cached_ = -shift;
    }
}

void BitWriter::uvlc(uint32_t v)
{
    v++;
    int len = 0;
    while ((v >> (len + 1)) != 0) len++;      // floor(log2 v)
    put(0, len);
    put(v, len + 1);
}

int BitWriter::flush()
{
    drain(cache_, cached_);
    cached_ &= 7;
    return (int)(p_ - p0_) + (cached_ + 7) / 8;
}

void BitWriter::raw_bytes(const uint8_t *src, size_t n)
{
    if (n > (size_t)(end_ - p_)) {
        overflow_ = true;
        return;
    }
    std::memcpy(p_, src, n);
    p_ += n;
}

// ---- arithmetic coder ----
namespace {
// rangeTabLPS of the standard (Table 9-48), indexed by probability state and bits 7..6 of the range
const uint8_t kRangeLps[64][4] = {
    { 128, 176, 208, 240 }, { 128, 167, 197, 227 }, { 128, 158, 187, 216 }, { 123, 150, 178, 205 },
    { 116, 142, 169, 195 }, { 111, 135, 160, 185 }, { 105, 128, 152, 175 }, { 100, 122, 144, 166 },
    { 95, 116, 137, 158 },  { 90, 110, 130, 150 },  { 85, 104, 123, 142 },  { 81, 99, 117, 135 },
    { 77, 94, 111, 128 },   { 73, 89, 105, 122 },   { 69, 85, 100, 116 },   { 66, 80, 95, 110 },
    { 62, 76, 90, 104 },    { 59, 72, 86, 99 },     { 56, 69, 81, 94 },     { 53, 65, 77, 89 },
    { 51, 62, 73, 85 },     { 48, 59, 69, 80 },     { 46, 56, 66, 76 },     { 43, 53, 63, 72 },
    { 41, 50, 59, 69 },     { 39, 48, 56, 65 },     { 37, 45, 54, 62 },     { 35, 43, 51, 59 },
    { 33, 41, 48, 56 },     { 32, 39, 46, 53 },     { 30, 37, 43, 50 },     { 29, 35, 41, 48 },
    { 27, 33, 39, 45 },     { 26, 31, 37, 43 },     { 24, 30, 35, 41 },     { 23, 28, 33, 39 },
    { 22, 27, 32, 37 },     { 21, 26, 30, 35 },     { 20, 24, 29, 33 },     { 19, 23, 27, 31 },
    { 18, 22, 26, 30 },     { 17, 21, 25, 28 },     { 16, 20, 23, 27 },     { 15, 19, 22, 25 },
    { 14, 18, 21, 24 },     { 14, 17, 20, 23 },     { 13, 16, 19, 22 },     { 12, 15, 18, 21 },
    { 12, 14, 17, 20 },     { 11, 14, 16, 19 },     { 11, 13, 15, 18 },     { 10, 12, 15, 17 },
    { 10, 12, 14, 16 },     { 9, 11, 13, 15 },      { 9, 11, 12, 14 },      { 8, 10, 12, 14 },
    { 8, 9, 11, 13 },       { 7, 9, 11, 12 },       { 7, 9, 10, 12 },       { 7, 8, 10, 11 },
    { 6, 8, 9, 11 },        { 6, 7, 9, 10 },        { 6, 7, 8, 9 },         { 2, 2, 2, 2 },
};
// shifts that bring an LPS range back to 256..510: 8 - bit length of the range (xg_aucRenormTable, indexed by range >> 3)
inline int renorm_shift(uint32_t lps)
{
    const uint32_t v = lps >> 3;
    int n = 6;
    for (uint32_t t = v; t; t >>= 1) n = n == 6 ? 5 : n - 1;
    return n;
}
}  // namespace

void Cabac::write_out()       // testAndWriteOut (src/bitstream.cpp:705)
{
    if (bits_left_ >= 12) return;
    const uint32_t lead = low_ >> (24 - bits_left_);
    bits_left_ += 8;
    low_ &= 0xffffffffu >> bits_left_;
    if (lead == 0xff) {
        num_bytes_++;
    } else if (num_bytes_ > 0) {
        const uint32_t carry = lead >> 8;
        uint32_t byte = cache_ + carry;
        cache_ = (uint8_t)(lead & 0xff);
        out->put(byte, 8);
        byte = (0xff + carry) & 0xff;
        while (num_bytes_ > 1) {
            out->put(byte, 8);
            num_bytes_--;
        }
    } else {
        num_bytes_ = 1;
        cache_ = (uint8_t)lead;
    }
}

void Cabac::bin(unsigned value, int c)       // xCabacEncodeBin (src/bitstream.cpp:738)
{
    const uint8_t *next = analysis::next_state_table();
    uint8_t st = ctx[c];
    const uint32_t lps = kRangeLps[st >> 1][(range_ >> 6) & 3];
    range_ -= lps;
    if (value != (unsigned)(st & 1)) {
        const int n = renorm_shift(lps);
        low_ = (low_ + range_) << n;
        range_ = lps << n;
        bits_left_ -= n;
    } else if (range_ < 256) {
        low_ <<= 1;
        range_ <<= 1;
        bits_left_--;
    }
    ctx[c] = next[st * 2 + (value ? 1 : 0)];
    write_out();
}

void Cabac::ep(unsigned value, int count)    // xCabacEncodeBinsEP (src/bitstream.cpp:780) on what a command word carries
{
    value &= 0xffffu;
    while (count > 8) {
        count -= 8;
        const uint32_t pattern = value >> count;
        low_ <<= 8;
        low_ += range_ * pattern;
        value -= pattern << count;
        bits_left_ -= 8;
        write_out();
    }
    low_ <<= count;
    low_ += range_ * value;
    bits_left_ -= count;
    write_out();
}

void Cabac::terminate(unsigned value)        // xCabacEncodeTerminatingBit (src/bitstream.cpp:801)
{
    range_ -= 2;
    if (value) {
        low_ += range_;
        low_ <<= 7;
        range_ = 2 << 7;
        bits_left_ -= 7;
    } else if (range_ < 256) {
        low_ <<= 1;
        range_ <<= 1;
        bits_left_--;
    }
    write_out();
}

void Cabac::flush()                          // xCabacFlush (src/bitstream.cpp:668)
{
    if (low_ >> (32 - bits_left_)) {
        out->put((uint32_t)cache_ + 1, 8);
        while (num_bytes_ > 1) {
            out->put(0x00, 8);
            num_bytes_--;
        }
        low_ -= 1u << (32 - bits_left_);
    } else {
        if (num_bytes_ > 0) out->put(cache_, 8);
        while (num_bytes_ > 1) {
            out->put(0xff, 8);
            num_bytes_--;
        }
    }
    out->put(low_ >> 8, 24 - bits_left_);
}

// ---- parameter sets and slice header ----
namespace {
void profile_tier_level(BitWriter &bs)       // xCodePTL / xCodeProfileTier (src/bitstream.cpp:53-86)
{
    bs.put(0, 2);                            // profile space
    bs.flag(0);                              // tier
    bs.put(1, 5);                            // profile idc
    for (int i = 0; i < 32; i++) bs.flag(i == 1 || i == 2);
    bs.flag(0), bs.flag(0), bs.flag(0), bs.flag(1);      // progressive, interlaced, non-packed, frame-only
    bs.put(0, 16), bs.put(0, 16), bs.put(0, 12);
    bs.put(180, 8);                          // level idc
}
}  // namespace

void write_vps(BitWriter &bs)                // xWriteVPS (src/bitstream.cpp:203)
{
    bs.put(0, 4), bs.put(3, 2), bs.put(0, 6), bs.put(0, 3);
    bs.flag(1);
    bs.put(0xffff, 16);
    profile_tier_level(bs);
    bs.flag(0);                              // sub_layer_ordering_info_present
    bs.uvlc(1), bs.uvlc(0), bs.uvlc(0);      // max_dec_pic_buffering_minus1, num_reorder_pics, max_latency_increase_plus1
    bs.put(0, 6);
    bs.uvlc(0);
    bs.flag(0);                              // timing info
    bs.flag(0);                              // extension
    bs.trailing_bits();
}

void write_sps(BitWriter &bs, const StreamParams &p)     // xWriteSPS (src/bitstream.cpp:88); parameters of src/params.cpp:28-61
{
    bs.put(0, 4), bs.put(0, 3);
    bs.flag(1);
    profile_tier_level(bs);
    bs.uvlc(0);
    bs.uvlc(1);                              // 4:2:0
    bs.uvlc((uint32_t)p.width), bs.uvlc((uint32_t)p.height);
    bs.flag(0);                              // conformance window
    bs.uvlc(0), bs.uvlc(0);                  // bit depths - 8
    bs.uvlc(8 - 4);                          // log2_max_pic_order_cnt_lsb_minus4
    bs.flag(0);
    bs.uvlc(1), bs.uvlc(0), bs.uvlc(0);
    // 64x64 CTUs, depth 4: minimum CU 8x8; transform blocks 4..32, hierarchy depth 2 (inter) / 1 (intra)
    bs.uvlc(3 - 3), bs.uvlc(4 - 1), bs.uvlc(2 - 2), bs.uvlc(5 - 2), bs.uvlc(2 - 1), bs.uvlc(1 - 1);
    bs.flag(0);                              // scaling lists
    bs.flag(0);                              // AMP
    bs.flag(1);                              // SAO
    bs.flag(0);                              // PCM
    bs.uvlc(1);                              // one short-term reference picture set: the previous picture
    bs.uvlc(1), bs.uvlc(0), bs.uvlc(0), bs.flag(1);
    bs.flag(0);                              // long-term references
    bs.flag(0);                              // temporal MVP
    bs.flag((unsigned)p.strong_intra_smoothing);
    bs.flag(0);                              // VUI
    bs.flag(0);                              // extension
    bs.trailing_bits();
}

void write_pps(BitWriter &bs)                // xWritePPS (src/bitstream.cpp:137)
{
    bs.uvlc(0), bs.uvlc(0);
    bs.flag(0);                              // dependent slices
    bs.flag(0);                              // output flag present
    bs.put(0, 3);
    bs.flag(0);                              // sign data hiding
    bs.flag(1);                              // cabac_init_present
    bs.uvlc(0), bs.uvlc(0);                  // default active references minus 1
    bs.svlc(0);                              // init_qp_minus26
    bs.flag(0), bs.flag(0), bs.flag(0);      // constrained intra, transform skip, cu_qp_delta
    bs.svlc(0), bs.svlc(0);
    bs.flag(0);
    bs.flag(0), bs.flag(0);                  // weighted prediction
    bs.flag(0);                              // transquant bypass
    bs.flag(0);                              // tiles
    bs.flag(1);                              // entropy_coding_sync_enabled
    bs.flag(0);                              // loop filter across slices
    bs.flag(0);                              // deblocking filter control present
    bs.flag(0);                              // scaling list data
    bs.flag(0);                              // lists modification
    bs.uvlc(0);                              // log2_parallel_merge_level_minus2
    bs.flag(0);
    bs.flag(0);
    bs.trailing_bits();
}

void write_slice_header(BitWriter &bs, int slice_type, int poc, int sao_luma, int sao_chroma, int qp)     // xWriteSliceHeader (:239)
{
    const bool intra = slice_type == 2;
    bs.flag(1);                              // first_slice_segment_in_pic
    if (intra) bs.flag(0);                   // no_output_of_prior_pics
    bs.uvlc(0);
    bs.uvlc((uint32_t)slice_type);
    if (!intra) {
        bs.put((uint32_t)(poc % 256), 8);
        bs.flag(1);                          // short_term_ref_pic_set_sps_flag
    }
    bs.flag((unsigned)sao_luma), bs.flag((unsigned)sao_chroma);
    if (!intra) {
        bs.flag(1);                          // num_ref_idx_active_override
        bs.uvlc(0);
        bs.flag(0);                          // cabac_init_flag
        bs.uvlc(0);                          // five_minus_max_num_merge_cand
    }
    bs.svlc(qp - 26);
}

void write_entry_points(BitWriter &bs, int count, int offset_len_minus1, const int32_t *offsets)         // xWriteTilesWPPEntryPoint_Yanan (:309)
{
    bs.uvlc((uint32_t)count);
    if (count > 0) bs.uvlc((uint32_t)offset_len_minus1);
    for (int i = 0; i < count; i++) bs.put((uint32_t)(offsets[i] - 1), offset_len_minus1 + 1);
    bs.trailing_bits();
}

// ---- MD5 (RFC 1321) ----
namespace {
struct Md5 {
    uint32_t a = 0x67452301u, b = 0xefcdab89u, c = 0x98badcfeu, d = 0x10325476u;
    uint64_t bytes = 0;
    uint8_t block[64];
    size_t fill = 0;

    static uint32_t rol(uint32_t v, int s) { return (v << s) | (v >> (32 - s)); }

    void transform(const uint8_t *p)
    {
        static const int shift[4][4] = { { 7, 12, 17, 22 }, { 5, 9, 14, 20 }, { 4, 11, 16, 23 }, { 6, 10, 15, 21 } };
        // (a function-local static object: its construction is thread safe, and two planes are hashed at the same time)
        struct Sines {
            uint32_t v[64];
            Sines()
            {
                for (int i = 0; i < 64; i++) v[i] = (uint32_t)(int64_t)std::floor(std::fabs(std::sin((double)(i + 1))) * 4294967296.0);
            }
        };
        static const Sines sines;
        const uint32_t *const k = sines.v;
        uint32_t m[16];
        for (int i = 0; i < 16; i++) m[i] = (uint32_t)p[4 * i] | ((uint32_t)p[4 * i + 1] << 8) | ((uint32_t)p[4 * i + 2] << 16) | ((uint32_t)p[4 * i + 3] << 24);
        uint32_t A = a, B = b, C = c, D = d;
        for (int i = 0; i < 64; i++) {
            uint32_t f;
            int g;
            if (i < 16) f = (B & C) | (~B & D), g = i;
            else if (i < 32) f = (D & B) | (~D & C), g = (5 * i + 1) & 15;
            else if (i < 48) f = B ^ C ^ D, g = (3 * i + 5) & 15;
            else f = C ^ (B | ~D), g = (7 * i) & 15;
            const uint32_t t = D;
            D = C;
            C = B;
            B = B + rol(A + f + k[i] + m[g], shift[i >> 4][i & 3]);
            A = t;
        }
        a += A, b += B, c += C, d += D;
    }

    void update(const uint8_t *p, size_t n)
    {
        bytes += n;
        while (n) {
            const size_t take = n < 64 - fill ? n : 64 - fill;
            std::memcpy(block + fill, p, take);
            fill += take, p += take, n -= take;
            if (fill == 64) {
                transform(block);
                fill = 0;
            }
        }
    }

    void finish(uint8_t out[16])
    {
        const uint64_t bits = bytes * 8;
        const uint8_t one = 0x80, zero = 0;
        update(&one, 1);
        while (fill != 56) update(&zero, 1);
        uint8_t len[8];
        for (int i = 0; i < 8; i++) len[i] = (uint8_t)(bits >> (8 * i));
        update(len, 8);
        const uint32_t v[4] = { a, b, c, d };
        for (int i = 0; i < 16; i++) out[i] = (uint8_t)(v[i >> 2] >> (8 * (i & 3)));
    }
};
}  // namespace

void md5_plane(const uint8_t *plane, size_t stride, size_t width, size_t height, uint8_t digest[16])
{
    Md5 m;
    for (size_t i = 0; i < height; i++) m.update(plane + i * stride, width);
    m.finish(digest);
}

}  // namespace bitstream
}  // namespace xmi
