/*
 * ora_skf.c -- CPU ORACLE (test infrastructure).  `.skf` codec:
 * snappy-frame( CBOR( MergeSkaArray ) ), merge_ska_array.rs:108-126,191-204.
 * The byte layout comes from third-party crates that are not vendored in the
 * reference (ciborium 0.2, snap 1.1, ndarray 0.15 serde; Cargo.toml:36,38,51);
 * it is restated from their published formats (RFC 8949; the snappy framing
 * format description) and pinned by decoding the six .skf fixtures the
 * reference's tests hold (the .skf files under tests/golden/input).
 */
#include "ora_internal.h"

/* ------------------------------------------------------------- CRC-32C */
static uint32_t crc_tab[256];
static int crc_init_done;
static void crc_init(void)
{
    for (uint32_t i = 0; i < 256; i++) {
        uint32_t c = i;
        for (int k = 0; k < 8; k++) c = (c & 1) ? (c >> 1) ^ 0x82F63B78u : c >> 1;
        crc_tab[i] = c;
    }
    crc_init_done = 1;
}
uint32_t ora_crc32c(const uint8_t *p, size_t n)
{
    if (!crc_init_done) crc_init();
    uint32_t c = 0xFFFFFFFFu;
    for (size_t i = 0; i < n; i++) c = crc_tab[(c ^ p[i]) & 0xFF] ^ (c >> 8);
    return c ^ 0xFFFFFFFFu;
}
static uint32_t crc_mask(uint32_t c) { return ((c >> 15) | (c << 17)) + 0xa282ead8u; }

/* ---------------------------------------------------- snappy raw block */
static int snappy_uncompress(const uint8_t *in, size_t n, uint8_t *out, size_t out_cap, size_t *out_len)
{
    size_t i = 0, ulen = 0; int shift = 0;
    for (;;) {
        if (i >= n) return -1;
        uint8_t b = in[i++]; ulen |= (size_t)(b & 0x7F) << shift; shift += 7;
        if (!(b & 0x80)) break;
        if (shift > 35) return -1;
    }
    if (ulen > out_cap) return -1;
    size_t o = 0;
    while (i < n) {
        uint8_t tag = in[i++];
        size_t len, off;
        switch (tag & 3) {
        case 0:
            len = tag >> 2;
            if (len >= 60) {
                size_t nb = len - 59; if (i + nb > n) return -1;
                len = 0; for (size_t k = 0; k < nb; k++) len |= (size_t)in[i + k] << (8 * k);
                i += nb;
            }
            len += 1;
            if (i + len > n || o + len > ulen) return -1;
            memcpy(out + o, in + i, len); i += len; o += len;
            continue;
        case 1:
            if (i >= n) return -1;
            len = ((tag >> 2) & 7) + 4; off = ((size_t)(tag >> 5) << 8) | in[i++];
            break;
        case 2:
            if (i + 2 > n) return -1;
            len = (tag >> 2) + 1; off = (size_t)in[i] | ((size_t)in[i + 1] << 8); i += 2;
            break;
        default:
            if (i + 4 > n) return -1;
            len = (tag >> 2) + 1;
            off = (size_t)in[i] | ((size_t)in[i + 1] << 8) | ((size_t)in[i + 2] << 16) | ((size_t)in[i + 3] << 24); i += 4;
            break;
        }
        if (off == 0 || off > o || o + len > ulen) return -1;
        for (size_t k = 0; k < len; k++) out[o + k] = out[o + k - off];
        o += len;
    }
    if (o != ulen) return -1;
    *out_len = o;
    return 0;
}

uint8_t *ora_snappy_frame_decode(const uint8_t *in, size_t len, size_t *out_len)
{
    size_t cap = len * 4 + 65536, o = 0, i = 0;
    uint8_t *out = (uint8_t *)malloc(cap);
    int seen_id = 0;
    while (i < len) {
        if (i + 4 > len) goto bad;
        uint8_t type = in[i];
        size_t clen = (size_t)in[i + 1] | ((size_t)in[i + 2] << 8) | ((size_t)in[i + 3] << 16);
        i += 4;
        if (i + clen > len) goto bad;
        if (type == 0xff) {
            if (clen != 6 || memcmp(in + i, "sNaPpY", 6)) goto bad;
            seen_id = 1;
        } else if (type == 0x00 || type == 0x01) {
            if (!seen_id || clen < 4) goto bad;
            uint32_t want = (uint32_t)in[i] | ((uint32_t)in[i + 1] << 8) | ((uint32_t)in[i + 2] << 16) | ((uint32_t)in[i + 3] << 24);
            if (cap - o < 65536 + 16) { cap = cap * 2 + 65536; out = (uint8_t *)realloc(out, cap); }
            size_t got;
            if (type == 0x00) {
                if (snappy_uncompress(in + i + 4, clen - 4, out + o, 65536, &got)) goto bad;
            } else {
                got = clen - 4; if (got > 65536) goto bad;
                memcpy(out + o, in + i + 4, got);
            }
            if (crc_mask(ora_crc32c(out + o, got)) != want) { ora_set_error("skf: snappy CRC mismatch"); free(out); return NULL; }
            o += got;
        } else if (type >= 0x02 && type <= 0x7f) {
            goto bad;                       /* reserved unskippable */
        }                                   /* 0x80..0xfe: skippable / padding */
        i += clen;
    }
    if (!seen_id) goto bad;
    *out_len = o;
    return out;
bad:
    ora_set_error("skf: bad snappy frame");
    free(out);
    return NULL;
}

/* writer: uncompressed chunks only (any valid framing is accepted by the reference reader) */
uint8_t *ora_snappy_frame_encode(const uint8_t *in, size_t len, size_t *out_len)
{
    size_t nchunks = (len + 65535) / 65536;
    uint8_t *out = (uint8_t *)malloc(10 + len + nchunks * 8 + 8), *p = out;
    memcpy(p, "\xff\x06\x00\x00sNaPpY", 10); p += 10;
    for (size_t off = 0; off < len; off += 65536) {
        size_t n = len - off < 65536 ? len - off : 65536;
        uint32_t c = crc_mask(ora_crc32c(in + off, n));
        size_t cl = n + 4;
        *p++ = 0x01; *p++ = (uint8_t)cl; *p++ = (uint8_t)(cl >> 8); *p++ = (uint8_t)(cl >> 16);
        *p++ = (uint8_t)c; *p++ = (uint8_t)(c >> 8); *p++ = (uint8_t)(c >> 16); *p++ = (uint8_t)(c >> 24);
        memcpy(p, in + off, n); p += n;
    }
    *out_len = (size_t)(p - out);
    return out;
}

/* ---------------------------------------------------------------- CBOR */
typedef struct { const uint8_t *p; size_t n, i; int err; } cb_in;

static int cb_head(cb_in *c, int *major, uint64_t *val)
{
    if (c->i >= c->n) { c->err = 1; return -1; }
    uint8_t b = c->p[c->i++];
    *major = b >> 5;
    uint8_t ai = b & 31;
    if (ai < 24) { *val = ai; return 0; }
    int nb = ai == 24 ? 1 : ai == 25 ? 2 : ai == 26 ? 4 : ai == 27 ? 8 : -1;
    if (nb < 0 || c->i + (size_t)nb > c->n) { c->err = 1; return -1; }
    uint64_t v = 0;
    for (int k = 0; k < nb; k++) v = (v << 8) | c->p[c->i++];
    *val = v;
    return 0;
}
static uint64_t cb_uint(cb_in *c)
{
    int m; uint64_t v;
    if (cb_head(c, &m, &v) || m != 0) { c->err = 1; return 0; }
    return v;
}
static char *cb_text(cb_in *c)
{
    int m; uint64_t v;
    if (cb_head(c, &m, &v) || m != 3 || c->i + v > c->n) { c->err = 1; return NULL; }
    char *s = (char *)malloc(v + 1); memcpy(s, c->p + c->i, v); s[v] = 0; c->i += v;
    return s;
}
/* uint, or tag 2 (positive bignum) + bstr, big-endian */
static ora_key cb_key(cb_in *c)
{
    ora_key k = { 0, 0 };
    int m; uint64_t v;
    if (cb_head(c, &m, &v)) return k;
    if (m == 0) { k.lo = v; return k; }
    if (m == 6 && v == 2) {
        if (cb_head(c, &m, &v) || m != 2 || v > 16 || c->i + v > c->n) { c->err = 1; return k; }
        u128 x = 0;
        for (uint64_t t = 0; t < v; t++) x = (x << 8) | c->p[c->i++];
        k.lo = (uint64_t)x; k.hi = (uint64_t)(x >> 64);
        return k;
    }
    c->err = 1;
    return k;
}

struct ora_array *ora_skf_decode(const uint8_t *cbor, size_t len)
{
    cb_in c = { cbor, len, 0, 0 };
    struct ora_array *a = (struct ora_array *)calloc(1, sizeof *a);
    int m; uint64_t nf;
    size_t dim0 = 0, dim1 = 0;
    if (cb_head(&c, &m, &nf) || m != 5) goto bad;
    for (uint64_t f = 0; f < nf && !c.err; f++) {
        char *name = cb_text(&c);
        if (!name) goto bad;
        if (!strcmp(name, "k")) a->k = (int)cb_uint(&c);
        else if (!strcmp(name, "rc")) {
            if (c.i >= c.n) c.err = 1;
            else { uint8_t b = c.p[c.i++]; if (b == 0xf5) a->rc = 1; else if (b == 0xf4) a->rc = 0; else c.err = 1; }
        } else if (!strcmp(name, "names")) {
            uint64_t n; if (cb_head(&c, &m, &n) || m != 4) c.err = 1;
            else {
                a->ns = n; a->names = (char **)calloc(n ? n : 1, sizeof(char *));
                for (uint64_t i = 0; i < n && !c.err; i++) a->names[i] = cb_text(&c);
            }
        } else if (!strcmp(name, "split_kmers")) {
            uint64_t n; if (cb_head(&c, &m, &n) || m != 4) c.err = 1;
            else {
                a->nk = n; a->keys = (ora_key *)malloc((n ? n : 1) * sizeof(ora_key));
                for (uint64_t i = 0; i < n && !c.err; i++) a->keys[i] = cb_key(&c);
            }
        } else if (!strcmp(name, "variants")) {
            uint64_t n3 = 0; if (cb_head(&c, &m, &n3) || m != 5) c.err = 1;
            for (uint64_t g = 0; g < n3 && !c.err; g++) {
                char *sub = cb_text(&c);
                if (!sub) break;
                if (!strcmp(sub, "v")) (void)cb_uint(&c);
                else if (!strcmp(sub, "dim")) {
                    uint64_t n; if (cb_head(&c, &m, &n) || m != 4 || n != 2) c.err = 1;
                    else { dim0 = cb_uint(&c); dim1 = cb_uint(&c); }
                } else if (!strcmp(sub, "data")) {
                    uint64_t n; if (cb_head(&c, &m, &n) || m != 4) c.err = 1;
                    else {
                        a->var = (uint8_t *)malloc(n ? n : 1);
                        for (uint64_t i = 0; i < n && !c.err; i++) a->var[i] = (uint8_t)cb_uint(&c);
                        if (n != dim0 * dim1) c.err = 1;
                    }
                } else c.err = 1;
                free(sub);
            }
        } else if (!strcmp(name, "variant_count")) {
            uint64_t n; if (cb_head(&c, &m, &n) || m != 4) c.err = 1;
            else {
                a->counts = (uint64_t *)malloc((n ? n : 1) * 8);
                for (uint64_t i = 0; i < n && !c.err; i++) a->counts[i] = cb_uint(&c);
                if (!a->nrows) a->nrows = n;
            }
        } else if (!strcmp(name, "ska_version")) a->version = cb_text(&c);
        else if (!strcmp(name, "k_bits")) a->k_bits = (int)cb_uint(&c);
        else c.err = 1;
        free(name);
    }
    if (c.err || !a->keys || !a->var || !a->counts || !a->names || !a->version) goto bad;
    a->nrows = dim0;
    if (dim1 != a->ns) goto bad;
    return a;
bad:
    ora_set_error("skf: CBOR decode failed");
    ora_array_free(a);
    return NULL;
}

typedef struct { uint8_t *p; size_t n, cap; } cb_out;
static void co_need(cb_out *o, size_t k) { if (o->n + k > o->cap) { o->cap = (o->n + k) * 2 + 64; o->p = (uint8_t *)realloc(o->p, o->cap); } }
static void co_head(cb_out *o, int major, uint64_t v)
{
    co_need(o, 9);
    uint8_t mb = (uint8_t)(major << 5);
    if (v < 24) o->p[o->n++] = mb | (uint8_t)v;
    else if (v <= 0xFF) { o->p[o->n++] = mb | 24; o->p[o->n++] = (uint8_t)v; }
    else if (v <= 0xFFFF) { o->p[o->n++] = mb | 25; o->p[o->n++] = (uint8_t)(v >> 8); o->p[o->n++] = (uint8_t)v; }
    else if (v <= 0xFFFFFFFFu) { o->p[o->n++] = mb | 26; for (int s = 24; s >= 0; s -= 8) o->p[o->n++] = (uint8_t)(v >> s); }
    else { o->p[o->n++] = mb | 27; for (int s = 56; s >= 0; s -= 8) o->p[o->n++] = (uint8_t)(v >> s); }
}
static void co_text(cb_out *o, const char *s)
{
    size_t l = strlen(s); co_head(o, 3, l); co_need(o, l); memcpy(o->p + o->n, s, l); o->n += l;
}

uint8_t *ora_skf_encode(const struct ora_array *a, size_t *len)
{
    cb_out o = { 0 };
    co_head(&o, 5, 8);
    co_text(&o, "k"); co_head(&o, 0, (uint64_t)a->k);
    co_text(&o, "rc"); co_need(&o, 1); o.p[o.n++] = a->rc ? 0xf5 : 0xf4;
    co_text(&o, "names"); co_head(&o, 4, a->ns);
    for (size_t i = 0; i < a->ns; i++) co_text(&o, a->names[i]);
    co_text(&o, "split_kmers"); co_head(&o, 4, a->nk);
    for (size_t i = 0; i < a->nk; i++) {
        if (!a->keys[i].hi) co_head(&o, 0, a->keys[i].lo);
        else {                                   /* tag 2 + minimal big-endian bstr */
            uint8_t be[16]; int nb = 0;
            u128 x = ((u128)a->keys[i].hi << 64) | a->keys[i].lo;
            for (int s = 120; s >= 0; s -= 8) { uint8_t b = (uint8_t)(x >> s); if (nb || b) be[nb++] = b; }
            co_head(&o, 6, 2); co_head(&o, 2, (uint64_t)nb); co_need(&o, (size_t)nb); memcpy(o.p + o.n, be, (size_t)nb); o.n += (size_t)nb;
        }
    }
    co_text(&o, "variants"); co_head(&o, 5, 3);
    co_text(&o, "v"); co_head(&o, 0, 1);
    co_text(&o, "dim"); co_head(&o, 4, 2); co_head(&o, 0, a->nrows); co_head(&o, 0, a->ns);
    co_text(&o, "data"); co_head(&o, 4, a->nrows * a->ns);
    for (size_t i = 0; i < a->nrows * a->ns; i++) co_head(&o, 0, a->var[i]);
    co_text(&o, "variant_count"); co_head(&o, 4, a->nrows);
    for (size_t i = 0; i < a->nrows; i++) co_head(&o, 0, a->counts[i]);
    co_text(&o, "ska_version"); co_text(&o, a->version);
    co_text(&o, "k_bits"); co_head(&o, 0, (uint64_t)a->k_bits);
    *len = o.n;
    return o.p;
}
