// skf_codec.cpp -- `.skf` reader/writer of the engine: snappy-frame( CBOR( MergeSkaArray ) ) as produced by
// MergeSkaArray::save / consumed by ::load (merge_ska_array.rs:191-204).  Field order and encodings follow
// what ciborium 0.2 + ndarray 0.15 serde emit for the struct at merge_ska_array.rs:108-126:
//   map(8){ k, rc, names[], split_kmers[] (u128 > 2^64-1 as tag-2 bignum), variants{v:1, dim:[U,S], data[]},
//           variant_count[], ska_version, k_bits }
// Chunks are written snappy-compressed (type 0x00) with masked CRC-32C so files are loadable by the real `ska`.
#include "skx_internal.h"
#include <cstdio>
#include <cstring>

namespace skx {
namespace {

// ---------------------------------------------------------------- CRC-32C (Castagnoli), slice-by-1
struct Crc32c {
    uint32_t t[256];
    Crc32c() { for (uint32_t i = 0; i < 256; i++) { uint32_t c = i; for (int k = 0; k < 8; k++) c = (c >> 1) ^ ((c & 1) ? 0x82F63B78u : 0); t[i] = c; } }
    uint32_t operator()(const uint8_t *p, size_t n) const { uint32_t c = ~0u; while (n--) c = t[(c ^ *p++) & 0xFF] ^ (c >> 8); return ~c; }
};
const Crc32c crc32c;
inline uint32_t mask_crc(uint32_t c) { return ((c >> 15) | (c << 17)) + 0xa282ead8u; }

// ---------------------------------------------------------------- snappy block (format_description.txt)
void put_varint(std::vector<uint8_t> &o, uint32_t v) { while (v >= 0x80) { o.push_back((uint8_t)(v | 0x80)); v >>= 7; } o.push_back((uint8_t)v); }
void put_literal(std::vector<uint8_t> &o, const uint8_t *p, size_t n)
{
    if (!n) return;
    size_t l = n - 1;
    if (l < 60) o.push_back((uint8_t)(l << 2));
    else if (l < 256) { o.push_back(60 << 2); o.push_back((uint8_t)l); }
    else { o.push_back(61 << 2); o.push_back((uint8_t)l); o.push_back((uint8_t)(l >> 8)); }      // blocks are <= 65536
    o.insert(o.end(), p, p + n);
}
void put_copy(std::vector<uint8_t> &o, size_t off, size_t len)
{
    while (len >= 68) { o.push_back((uint8_t)(((64 - 1) << 2) | 2)); o.push_back((uint8_t)off); o.push_back((uint8_t)(off >> 8)); len -= 64; }
    if (len > 64) { o.push_back((uint8_t)(((60 - 1) << 2) | 2)); o.push_back((uint8_t)off); o.push_back((uint8_t)(off >> 8)); len -= 60; }
    if (len >= 12 || off >= 2048) { o.push_back((uint8_t)(((len - 1) << 2) | 2)); o.push_back((uint8_t)off); o.push_back((uint8_t)(off >> 8)); }
    else { o.push_back((uint8_t)(((off >> 8) << 5) | ((len - 4) << 2) | 1)); o.push_back((uint8_t)off); }
}
void snappy_compress_block(const uint8_t *in, size_t n, std::vector<uint8_t> &o)
{
    put_varint(o, (uint32_t)n);
    if (n < 16) { put_literal(o, in, n); return; }
    static thread_local uint16_t table[1 << 14];
    memset(table, 0, sizeof table);
    auto load32 = [&](size_t i) { uint32_t v; memcpy(&v, in + i, 4); return v; };
    auto hash = [&](uint32_t v) { return (v * 0x1e35a7bdu) >> 18; };
    size_t lit = 0, i = 1;
    const size_t limit = n - 4;
    while (i <= limit) {
        uint32_t cur = load32(i), h = hash(cur);
        size_t cand = table[h];
        table[h] = (uint16_t)i;
        if (cand < i && load32(cand) == cur) {
            size_t len = 4;
            while (i + len < n && in[cand + len] == in[i + len]) len++;
            put_literal(o, in + lit, i - lit);
            put_copy(o, i - cand, len);
            i += len; lit = i;
        } else i++;
    }
    put_literal(o, in + lit, n - lit);
}
bool snappy_uncompress_block(const uint8_t *in, size_t n, std::vector<uint8_t> &out)
{
    size_t i = 0; uint32_t ulen = 0; int sh = 0;
    for (;;) { if (i >= n || sh > 28) return false; uint8_t b = in[i++]; ulen |= (uint32_t)(b & 0x7F) << sh; sh += 7; if (!(b & 0x80)) break; }
    const size_t base = out.size();
    out.reserve(base + ulen);
    while (i < n) {
        const uint8_t tag = in[i++];
        size_t len, off;
        if ((tag & 3) == 0) {
            len = tag >> 2;
            if (len >= 60) { size_t nb = len - 59; if (i + nb > n) return false; len = 0; for (size_t k = 0; k < nb; k++) len |= (size_t)in[i + k] << (8 * k); i += nb; }
            len++;
            if (i + len > n) return false;
            out.insert(out.end(), in + i, in + i + len); i += len;
            continue;
        }
        if ((tag & 3) == 1) { if (i >= n) return false; len = ((tag >> 2) & 7) + 4; off = ((size_t)(tag >> 5) << 8) | in[i++]; }
        else if ((tag & 3) == 2) { if (i + 2 > n) return false; len = (tag >> 2) + 1; off = in[i] | ((size_t)in[i + 1] << 8); i += 2; }
        else { if (i + 4 > n) return false; len = (tag >> 2) + 1; off = in[i] | ((size_t)in[i + 1] << 8) | ((size_t)in[i + 2] << 16) | ((size_t)in[i + 3] << 24); i += 4; }
        if (!off || off > out.size() - base) return false;
        for (size_t k = 0; k < len; k++) out.push_back(out[out.size() - off]);
    }
    return out.size() - base == ulen;
}

// ---------------------------------------------------------------- CBOR (RFC 8949, definite lengths only)
struct Writer {
    std::vector<uint8_t> b;
    void head(int major, uint64_t v)
    {
        uint8_t m = (uint8_t)(major << 5);
        if (v < 24) b.push_back(m | (uint8_t)v);
        else if (v <= 0xFF) { b.push_back(m | 24); b.push_back((uint8_t)v); }
        else if (v <= 0xFFFF) { b.push_back(m | 25); b.push_back((uint8_t)(v >> 8)); b.push_back((uint8_t)v); }
        else if (v <= 0xFFFFFFFFull) { b.push_back(m | 26); for (int s = 24; s >= 0; s -= 8) b.push_back((uint8_t)(v >> s)); }
        else { b.push_back(m | 27); for (int s = 56; s >= 0; s -= 8) b.push_back((uint8_t)(v >> s)); }
    }
    void text(const std::string &s) { head(3, s.size()); b.insert(b.end(), s.begin(), s.end()); }
};
struct Reader {
    const uint8_t *p; size_t n, i = 0; bool ok = true;
    bool head(int &major, uint64_t &v)
    {
        if (i >= n) return ok = false;
        uint8_t c = p[i++]; major = c >> 5; uint8_t ai = c & 31;
        if (ai < 24) { v = ai; return true; }
        int nb = ai == 24 ? 1 : ai == 25 ? 2 : ai == 26 ? 4 : ai == 27 ? 8 : 0;
        if (!nb || i + nb > n) return ok = false;
        v = 0; for (int k = 0; k < nb; k++) v = (v << 8) | p[i++];
        return true;
    }
    uint64_t uint() { int m; uint64_t v = 0; if (!head(m, v) || m != 0) ok = false; return v; }
    uint64_t array() { int m; uint64_t v = 0; if (!head(m, v) || m != 4) ok = false; return v; }
    std::string text() { int m; uint64_t v = 0; if (!head(m, v) || m != 3 || i + v > n) { ok = false; return {}; } std::string s((const char *)p + i, v); i += v; return s; }
    skx_key key()
    {
        skx_key k{0, 0}; int m; uint64_t v;
        if (!head(m, v)) return k;
        if (m == 0) { k.lo = v; return k; }
        if (m == 6 && v == 2 && head(m, v) && m == 2 && v <= 16 && i + v <= n) {
            unsigned __int128 x = 0; for (uint64_t t = 0; t < v; t++) x = (x << 8) | p[i++];
            k.lo = (uint64_t)x; k.hi = (uint64_t)(x >> 64); return k;
        }
        ok = false; return k;
    }
};

}  // namespace

int skf_read(const char *path, SkfData &d)
{
    FILE *f = fopen(path, "rb");
    if (!f) { set_error("cannot open %s", path); return SKX_EIO; }
    std::vector<uint8_t> raw;
    uint8_t tmp[1 << 16]; size_t r;
    while ((r = fread(tmp, 1, sizeof tmp, f)) > 0) raw.insert(raw.end(), tmp, tmp + r);
    fclose(f);
    // --- snappy frame
    std::vector<uint8_t> cbor;
    size_t i = 0; bool seen = false;
    while (i < raw.size()) {
        if (i + 4 > raw.size()) { set_error("skf: truncated frame"); return SKX_EFORMAT; }
        const uint8_t type = raw[i]; const size_t len = raw[i + 1] | ((size_t)raw[i + 2] << 8) | ((size_t)raw[i + 3] << 16);
        i += 4;
        if (i + len > raw.size()) { set_error("skf: truncated frame"); return SKX_EFORMAT; }
        if (type == 0xff) { if (len != 6 || memcmp(&raw[i], "sNaPpY", 6)) { set_error("skf: not a snappy stream"); return SKX_EFORMAT; } seen = true; }
        else if (type <= 0x01) {
            if (!seen || len < 4) { set_error("skf: bad chunk"); return SKX_EFORMAT; }
            uint32_t want; memcpy(&want, &raw[i], 4);
            const size_t before = cbor.size();
            if (type == 0x00) { if (!snappy_uncompress_block(&raw[i + 4], len - 4, cbor)) { set_error("skf: corrupt snappy block"); return SKX_EFORMAT; } }
            else cbor.insert(cbor.end(), raw.begin() + i + 4, raw.begin() + i + len);
            if (mask_crc(crc32c(cbor.data() + before, cbor.size() - before)) != want) { set_error("skf: checksum mismatch"); return SKX_EFORMAT; }
        } else if (type < 0x80) { set_error("skf: unsupported chunk type %u", type); return SKX_EFORMAT; }
        i += len;
    }
    if (!seen) { set_error("skf: not a snappy stream"); return SKX_EFORMAT; }
    // --- CBOR struct
    Reader rd{cbor.data(), cbor.size()};
    int m; uint64_t nf = 0;
    if (!rd.head(m, nf) || m != 5) { set_error("skf: CBOR decode failed"); return SKX_EFORMAT; }
    uint64_t dim0 = 0, dim1 = 0; bool have_var = false;
    for (uint64_t fidx = 0; fidx < nf && rd.ok; fidx++) {
        const std::string name = rd.text();
        if (name == "k") d.k = (int)rd.uint();
        else if (name == "rc") { if (rd.i < rd.n && (rd.p[rd.i] == 0xf4 || rd.p[rd.i] == 0xf5)) d.rc = rd.p[rd.i++] == 0xf5; else rd.ok = false; }
        else if (name == "names") { uint64_t n = rd.array(); for (uint64_t j = 0; j < n && rd.ok; j++) d.names.push_back(rd.text()); }
        else if (name == "split_kmers") { uint64_t n = rd.array(); d.keys.reserve(n); for (uint64_t j = 0; j < n && rd.ok; j++) d.keys.push_back(rd.key()); }
        else if (name == "variants") {
            uint64_t n3 = 0; if (!rd.head(m, n3) || m != 5) rd.ok = false;
            for (uint64_t g = 0; g < n3 && rd.ok; g++) {
                const std::string sub = rd.text();
                if (sub == "v") rd.uint();
                else if (sub == "dim") { if (rd.array() != 2) rd.ok = false; dim0 = rd.uint(); dim1 = rd.uint(); }
                else if (sub == "data") { uint64_t n = rd.array(); d.variants.resize(n); for (uint64_t j = 0; j < n && rd.ok; j++) d.variants[j] = (uint8_t)rd.uint(); have_var = true; }
                else rd.ok = false;
            }
        }
        else if (name == "variant_count") { uint64_t n = rd.array(); d.counts.resize(n); for (uint64_t j = 0; j < n && rd.ok; j++) d.counts[j] = rd.uint(); }
        else if (name == "ska_version") d.version = rd.text();
        else if (name == "k_bits") d.k_bits = (int)rd.uint();
        else rd.ok = false;
    }
    if (!rd.ok || !have_var || dim0 * dim1 != d.variants.size() || dim1 != d.names.size()) { set_error("skf: CBOR decode failed"); return SKX_EFORMAT; }
    d.n_rows = dim0;
    return SKX_OK;
}

int skf_write(const char *path, const SkfData &d)
{
    Writer w;
    const uint64_t S = d.names.size();
    w.b.reserve(d.variants.size() * 2 + d.keys.size() * 12 + 1024);
    w.head(5, 8);
    w.text("k"); w.head(0, (uint64_t)d.k);
    w.text("rc"); w.b.push_back(d.rc ? 0xf5 : 0xf4);
    w.text("names"); w.head(4, S); for (auto &s : d.names) w.text(s);
    w.text("split_kmers"); w.head(4, d.keys.size());
    for (auto &kk : d.keys) {
        if (!kk.hi) w.head(0, kk.lo);
        else {
            uint8_t be[16]; int nb = 0;
            for (int s = 56; s >= 0; s -= 8) { uint8_t b = (uint8_t)(kk.hi >> s); if (nb || b) be[nb++] = b; }
            for (int s = 56; s >= 0; s -= 8) be[nb++] = (uint8_t)(kk.lo >> s);
            w.head(6, 2); w.head(2, (uint64_t)nb); w.b.insert(w.b.end(), be, be + nb);
        }
    }
    w.text("variants"); w.head(5, 3);
    w.text("v"); w.head(0, 1);
    w.text("dim"); w.head(4, 2); w.head(0, d.n_rows); w.head(0, S);
    w.text("data"); w.head(4, d.variants.size());
    for (uint8_t v : d.variants) { if (v < 24) w.b.push_back(v); else { w.b.push_back(0x18); w.b.push_back(v); } }
    w.text("variant_count"); w.head(4, d.counts.size()); for (uint64_t c : d.counts) w.head(0, c);
    w.text("ska_version"); w.text(d.version);
    w.text("k_bits"); w.head(0, (uint64_t)d.k_bits);

    FILE *f = fopen(path, "wb");
    if (!f) { set_error("cannot create %s", path); return SKX_EIO; }
    bool ok = fwrite("\xff\x06\x00\x00sNaPpY", 1, 10, f) == 10;
    std::vector<uint8_t> blk;
    for (size_t off = 0; off < w.b.size() && ok; off += 65536) {
        const size_t n = std::min<size_t>(65536, w.b.size() - off);
        blk.clear();
        snappy_compress_block(w.b.data() + off, n, blk);
        const bool raw = blk.size() >= n - n / 8;           // snap's rule of thumb: store incompressible chunks raw
        const uint8_t *payload = raw ? w.b.data() + off : blk.data();
        const size_t plen = raw ? n : blk.size(), clen = plen + 4;
        const uint32_t crc = mask_crc(crc32c(w.b.data() + off, n));
        uint8_t hdr[8] = {(uint8_t)(raw ? 1 : 0), (uint8_t)clen, (uint8_t)(clen >> 8), (uint8_t)(clen >> 16),
                          (uint8_t)crc, (uint8_t)(crc >> 8), (uint8_t)(crc >> 16), (uint8_t)(crc >> 24)};
        ok = fwrite(hdr, 1, 8, f) == 8 && fwrite(payload, 1, plen, f) == plen;
    }
    ok = (fclose(f) == 0) && ok;
    if (!ok) { set_error("short write %s", path); return SKX_EIO; }
    return SKX_OK;
}

}  // namespace skx
