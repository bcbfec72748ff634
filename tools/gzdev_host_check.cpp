// tools/gzdev_host_check.cpp -- the device inflater's per-lane logic (ska.rust_amd/csrc/gz_device.h) run on the host, kernel by kernel, in the
// order and with the layout skx_gzdev.hip uses (block finder per chunk, symbolic decode per chunk, group maps, group windows, text, member
// lengths and CRCs), so that it can be checked against zlib without a GPU (tests/test_gz_device_logic.py feeds it files of every kind).
//   gzdev_host_check <file.gz> <out> [chunk_bytes=65536] [ratio=8] [group=32]
// prints: status total n_members n_chunks n_synced ; writes the text to <out> when status is 0.
// g++ -O2 -o gzdev_host_check tools/gzdev_host_check.cpp
#define GZD_HD inline
#include "../ska.rust_amd/csrc/gz_device.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
using namespace gzd;

int main(int argc, char **argv)
{
    if (argc < 3) { fprintf(stderr, "usage\n"); return 2; }
    const uint32_t chunk_bytes = argc > 3 ? (uint32_t)atol(argv[3]) : 65536, ratio = argc > 4 ? (uint32_t)atol(argv[4]) : 8, group = argc > 5 ? (uint32_t)atol(argv[5]) : 32;
    FILE *f = fopen(argv[1], "rb");
    if (!f) { perror("open"); return 2; }
    std::vector<uint8_t> raw;
    { uint8_t buf[1 << 16]; size_t r; while ((r = fread(buf, 1, sizeof buf, f)) > 0) raw.insert(raw.end(), buf, buf + r); }
    fclose(f);
    const uint64_t src_bytes = raw.size();
    std::vector<uint32_t> w((src_bytes + 3) / 4 + 8, 0);
    memcpy(w.data(), raw.data(), src_bytes);
    const uint64_t nwords = (src_bytes + 3) / 4;
    uint32_t n_chunks = (uint32_t)((src_bytes + chunk_bytes - 1) / chunk_bytes);
    if (!n_chunks) n_chunks = 1;
    const uint32_t n_groups = (n_chunks + group - 1) / group;
    Tables t;
    // K0
    std::vector<uint64_t> sync(n_chunks + 1, NONE);
    uint32_t n_synced = 0;
    for (uint32_t c = 1; c < n_chunks; c++) {
        const uint64_t lo = (uint64_t)c * chunk_bytes * 8, hi = std::min<uint64_t>(lo + (uint64_t)chunk_bytes * 8, src_bytes * 8);
        for (uint64_t pos = lo; pos < hi; pos++)
            if (sync_quick(w.data(), nwords, pos) && sync_verify(w.data(), src_bytes, pos, t)) { sync[c] = pos; n_synced++; break; }
    }
    // K1
    std::vector<uint16_t> sym((size_t)ratio * (src_bytes + 8));
    std::vector<ChunkInfo> info(n_chunks);
    std::vector<Member> members((size_t)n_chunks * MAX_MEMBERS);
    auto sym_of = [&](uint32_t c) { return sym.data() + (uint64_t)ratio * (c ? (sync[c] == NONE ? 0 : sync[c] >> 3) : 0); };
    for (uint32_t c = 0; c < n_chunks; c++) {
        if (c && sync[c] == NONE) { info[c] = ChunkInfo{0, 0, OK, 0}; continue; }
        uint32_t c2 = c + 1;
        while (c2 < n_chunks && sync[c2] == NONE) c2++;
        const uint64_t start = c ? sync[c] : NONE, stop = c2 < n_chunks ? sync[c2] : NONE;
        const uint64_t sb = c ? start >> 3 : 0, eb = stop == NONE ? src_bytes : stop >> 3;
        decode_chunk(w.data(), src_bytes, start, stop, t, sym.data() + (uint64_t)ratio * sb, (uint64_t)ratio * (eb - sb), &info[c], &members[(size_t)c * MAX_MEMBERS]);
    }
    // K2a
    std::vector<uint16_t> maps((size_t)n_chunks * WIN), gwin((size_t)n_groups * WIN);
    for (uint32_t c = 0; c < n_chunks; c++) {
        const uint16_t *prev = c % group ? &maps[(size_t)(c - 1) * WIN] : nullptr;
        for (uint32_t i = 0; i < WIN; i++) maps[(size_t)c * WIN + i] = map_entry(sym_of(c), info[c].n_out, prev, i);
    }
    // K2b
    for (uint32_t i = 0; i < WIN; i++) gwin[i] = INVALID;
    for (uint32_t g = 1; g < n_groups; g++)
        for (uint32_t i = 0; i < WIN; i++) gwin[(size_t)g * WIN + i] = through(&gwin[(size_t)(g - 1) * WIN], maps[(size_t)(g * group - 1) * WIN + i]);
    uint32_t status = OK, nm = 0;
    uint64_t total = 0;
    std::vector<uint64_t> base(n_chunks + 1), m_end;
    std::vector<uint32_t> m_crc, m_isize;
    for (uint32_t c = 0; c < n_chunks; c++) {
        base[c] = total;
        if (info[c].status && !status) status = info[c].status;
        for (uint32_t j = 0; j < info[c].n_members; j++) { const Member &mb = members[(size_t)c * MAX_MEMBERS + j]; m_end.push_back(total + mb.end); m_crc.push_back(mb.crc); m_isize.push_back(mb.isize); nm++; }
        total += info[c].n_out;
    }
    for (uint32_t i = 0; i < nm; i++) if ((uint32_t)(m_end[i] - (i ? m_end[i - 1] : 0)) != m_isize[i] && !status) status = E_CHECK;
    if (!status && (nm == 0 || m_end[nm - 1] != total)) status = E_DATA;
    // K3
    std::vector<uint8_t> text(total + 1);
    for (uint32_t c = 0; c < n_chunks && !status; c++) {
        const uint16_t *prev = c % group ? &maps[(size_t)(c - 1) * WIN] : nullptr, *gw = &gwin[(size_t)(c / group) * WIN], *s = sym_of(c);
        for (uint64_t j = 0; j < info[c].n_out; j++) {
            uint16_t v = s[j];
            if (v >= SYM0) { if (v != INVALID && prev) v = prev[v - SYM0]; v = through(gw, v); if (v >= SYM0) { status = E_DATA; v = '?'; } }
            text[base[c] + j] = (uint8_t)v;
        }
    }
    // the window behind the last chunk gives the last byte
    if (!status && total) {
        const uint32_t L = n_chunks - 1;
        const uint16_t last = through(&gwin[(size_t)(L / group) * WIN], maps[(size_t)L * WIN + WIN - 1]);
        if (last != text[total - 1]) { fprintf(stderr, "last byte from the window differs\n"); status = 99; }
    }
    // CRC pieces
    if (!status) {
        const uint32_t PIECE = 4096;
        std::vector<uint32_t> acc(nm, 0);
        for (uint64_t q = 0; q * PIECE < total; q++) {
            uint64_t pos = q * PIECE; const uint64_t rend = std::min<uint64_t>(pos + PIECE, total);
            uint32_t m = 0;
            while (m < nm && m_end[m] <= pos) m++;
            while (pos < rend && m < nm) {
                const uint64_t mend = m_end[m], e = std::min(rend, mend);
                const uint32_t piece = crc_bytes(0xFFFFFFFFu, &text[pos], e - pos) ^ 0xFFFFFFFFu;
                acc[m] ^= crc_mul(crc_xpow8(mend - e), piece);
                pos = e;
                if (pos == mend) m++;
            }
        }
        for (uint32_t i = 0; i < nm; i++) if (acc[i] != m_crc[i]) status = E_CHECK;
    }
    printf("%u %llu %u %u %u\n", status, (unsigned long long)total, nm, n_chunks, n_synced);
    if (!status) { FILE *o = fopen(argv[2], "wb"); if (!o) return 2; fwrite(text.data(), 1, total, o); fclose(o); }
    return 0;
}
