/*
 * ora_fastx.c -- CPU ORACLE (test infrastructure).  Minimal FASTA/FASTQ reader
 * reproducing what needletail 0.5 (Cargo.toml:34; third-party, not vendored)
 * hands to ska_dict.rs:131-153: records in file order, `seq()` with line
 * endings removed (multi-line FASTA joined), `qual()` for FASTQ, format from
 * the first byte ('>' FASTA, '@' FASTQ), gzip transparently (zlib).
 */
#include "ora_internal.h"
#include <zlib.h>

static uint8_t *slurp(const char *path, size_t *len)
{
    gzFile g = gzopen(path, "rb");
    if (!g) { ora_set_error("Invalid path/file: %s", path); return NULL; }
    size_t cap = 1 << 20, n = 0;
    uint8_t *buf = (uint8_t *)malloc(cap);
    for (;;) {
        if (cap - n < (1 << 19)) { cap *= 2; buf = (uint8_t *)realloc(buf, cap); }
        int r = gzread(g, buf + n, (unsigned)(cap - n > (1u << 30) ? (1u << 30) : cap - n));
        if (r < 0) { ora_set_error("Invalid path/file: %s", path); gzclose(g); free(buf); return NULL; }
        if (r == 0) break;
        n += (size_t)r;
    }
    gzclose(g);
    *len = n;
    return buf;
}

void ora_fastx_free(ora_fastx *f) { free(f->recs); free(f->arena); memset(f, 0, sizeof *f); }

static void push(ora_fastx *f, size_t *cap, const uint8_t *seq, size_t len, const uint8_t *qual)
{
    if (f->n == *cap) { *cap = *cap ? *cap * 2 : 64; f->recs = (ora_rec *)realloc(f->recs, *cap * sizeof(ora_rec)); }
    f->recs[f->n].seq = seq; f->recs[f->n].len = len; f->recs[f->n].qual = qual; f->recs[f->n].id = NULL; f->recs[f->n].id_len = 0; f->n++;
}

int ora_fastx_read(const char *path, ora_fastx *out)
{
    memset(out, 0, sizeof *out);
    size_t n; uint8_t *b = slurp(path, &n);
    if (!b) return -1;
    if (n == 0) { free(b); ora_set_error("Invalid path/file: %s", path); return -1; }   /* needletail EmptyFile */
    out->arena = b;
    size_t cap = 0, i = 0;
    if (b[0] == '>') {
        out->is_fastq = 0;
        while (i < n) {
            if (b[i] != '>') { ora_set_error("Invalid FASTA/Q record"); ora_fastx_free(out); return -1; }
            const size_t id0 = i + 1;
            while (i < n && b[i] != '\n') i++;          /* header line */
            size_t id1 = i;
            if (id1 > id0 && b[id1 - 1] == '\r') id1--;
            if (i < n) i++;
            /* sequence: up to the next line that starts with '>' ; compact in place */
            size_t w = i, s0 = i;
            while (i < n) {
                if (b[i] == '>' && (i == s0 || b[i - 1] == '\n')) break;
                uint8_t c = b[i++];
                if (c != '\n' && c != '\r') b[w++] = c;
            }
            push(out, &cap, b + s0, w - s0, NULL);
            out->recs[out->n - 1].id = b + id0; out->recs[out->n - 1].id_len = id1 - id0;
        }
    } else if (b[0] == '@') {
        out->is_fastq = 1;
        while (i < n) {
            if (b[i] == '\n' || b[i] == '\r') { i++; continue; }   /* trailing blank lines */
            if (b[i] != '@') { ora_set_error("Invalid FASTA/Q record"); ora_fastx_free(out); return -1; }
            while (i < n && b[i] != '\n') i++;
            if (i < n) i++;
            size_t s0 = i;
            while (i < n && b[i] != '\n') i++;
            size_t s1 = i; if (s1 > s0 && b[s1 - 1] == '\r') s1--;
            if (i < n) i++;
            if (i >= n || b[i] != '+') { ora_set_error("Invalid FASTA/Q record"); ora_fastx_free(out); return -1; }
            while (i < n && b[i] != '\n') i++;
            if (i < n) i++;
            size_t q0 = i;
            while (i < n && b[i] != '\n') i++;
            size_t q1 = i; if (q1 > q0 && b[q1 - 1] == '\r') q1--;
            if (i < n) i++;
            if (q1 - q0 != s1 - s0) { ora_set_error("Invalid FASTA/Q record"); ora_fastx_free(out); return -1; }
            push(out, &cap, b + s0, s1 - s0, b + q0);
        }
    } else {
        ora_set_error("Invalid FASTA/Q record");
        ora_fastx_free(out);
        return -1;
    }
    return 0;
}
