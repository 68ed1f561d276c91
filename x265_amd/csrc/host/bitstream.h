// bitstream.h - byte-level output of the encoder: the bit writer with start-code emulation prevention (xBitStream,
// inc/bitstream.h:36-187), the CABAC arithmetic coder (xCabac*, src/bitstream.cpp:659-843) as a sink of syntax.h, the
// parameter sets and slice header (src/bitstream.cpp:88-332) and the MD5 digest of the picture-hash SEI
// (src/encode.cpp:6012-6053). Host code only.
#pragma once
#include <cstddef>
#include <cstdint>

namespace xmi {
namespace bitstream {

// Bits are collected msb first in a 32-bit cache and leave as bytes; a byte <= 3 that would follow two zero bytes of the
// output gets 0x03 in front (flushCache_WPP_Amendment_1_YANAN, inc/bitstream.h:48). The caller owns the buffer, and the
// two bytes in front of `begin` must be readable (the escape test looks back over them). Nothing is written at or beyond
// `end`: bytes that do not fit are dropped and overflowed() says so (the reference's writer has no such limit and no check).
class BitWriter {
public:
    void init(uint8_t *begin, uint8_t *end)
    {
        p_ = p0_ = begin, end_ = end;
        cache_ = 0, cached_ = 0, escapes = 0, overflow_ = false;
    }
    bool overflowed() const { return overflow_; }
    size_t room() const { return (size_t)(end_ - p_); }
    void put(uint32_t bits, int n);                 // xPutBits: n in 0..32, bits < 2^n
    void flag(unsigned v) { put(v ? 1u : 0u, 1); }
    void uvlc(uint32_t v);                          // xWriteUvlc
    void svlc(int32_t v) { uvlc(v > 0 ? (uint32_t)(v << 1) - 1 : (uint32_t)(-v) << 1); }   // xWriteSvlc
    void align_zero() { put(0, (8 - cached_) & 7); }
    void trailing_bits()                            // xWriteRBSPTrailingBits / xWriteByteAlignment
    {
        put(1, 1);
        align_zero();
    }
    int flush();                                    // xBitFlush: whole bytes leave the cache; bytes written so far (a partial one counted)
    void raw_byte(uint8_t b)                        // start codes bypass the cache (xPutBits32); only on a byte boundary
    {
        if (p_ < end_) *p_++ = b;
        else overflow_ = true;
    }
    void raw_bytes(const uint8_t *src, size_t n);
    uint8_t *pos() const { return p_; }
    int cached_bits() const { return cached_; }
    uint32_t escapes = 0;                           // emulation-prevention bytes inserted (numPreventByte)

private:
    void drain(uint32_t &word, int bits);
    uint8_t *p_ = nullptr, *p0_ = nullptr, *end_ = nullptr;
    bool overflow_ = false;
    uint32_t cache_ = 0;
    int cached_ = 0;
};

// The arithmetic coder. As a syntax sink: bin() and ep() each stand for one command of the reference's command lists and end
// with its testAndWriteOut.
class Cabac {
public:
    uint8_t ctx[256];
    BitWriter *out = nullptr;
    void reset()                                    // xCabacReset
    {
        low_ = 0, range_ = 510, bits_left_ = 23, cache_ = 0xff, num_bytes_ = 0;
    }
    void bin(unsigned value, int c);
    void ep(unsigned value, int count);
    void terminate(unsigned value);
    void flush();                                   // xCabacFlush (the state stays as the reference leaves it: it is flushed twice per slice)

private:
    void write_out();
    uint32_t low_ = 0, range_ = 510;
    int32_t bits_left_ = 23;
    uint8_t cache_ = 0xff;
    uint32_t num_bytes_ = 0;
};

struct StreamParams {
    int width, height;              // coded size
    int strong_intra_smoothing;
};
void write_vps(BitWriter &bs);
void write_sps(BitWriter &bs, const StreamParams &p);
void write_pps(BitWriter &bs);
// xWriteSliceHeader up to and including slice_qp_delta; the entry points follow once the substreams are known
void write_slice_header(BitWriter &bs, int slice_type, int poc, int sao_luma, int sao_chroma, int qp);
void write_entry_points(BitWriter &bs, int count, int offset_len_minus1, const int32_t *offsets);

void md5_plane(const uint8_t *plane, size_t stride, size_t width, size_t height, uint8_t digest[16]);    // digest over the rows in order

}  // namespace bitstream
}  // namespace xmi
