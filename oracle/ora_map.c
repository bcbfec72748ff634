/*
 * ora_map.c -- CPU ORACLE (test infrastructure): `ska map` restated from the reference
 *   RefSka::new / map / pseudoalignment / write_aln / write_vcf   src/ska_ref.rs:189-311,508-533,551-765
 *   AlnWriter                                                      src/ska_ref/aln_writer.rs
 *   IdxCheck                                                       src/ska_ref/idx_check.rs
 *   generic_modes::map                                             src/generic_modes.rs:56-84
 * noodles-vcf (Cargo.toml) is not in /root/reference: the VCF text is restated from its output as the reference's own
 * goldens pin it (tests/test_results_correct/map_vcf*.stdout): "##fileformat=VCFv4.x", one "##contig=<ID=..>" per
 * chromosome, the #CHROM header, tab-separated records with '.' for missing ID/ALT/QUAL/FILTER/INFO.
 * Nothing here is part of the product path.
 */
#include "ora_internal.h"

/* RC_IUPAC (bit_encoding.rs:475-510): complement of an IUPAC code, anything else -> '-' */
static uint8_t rc_iupac(uint8_t b)
{
    switch (b | 0x20) {
    case 'a': return 'T'; case 'b': return 'V'; case 'c': return 'G'; case 'd': return 'H'; case 'g': return 'C';
    case 'h': return 'D'; case 'k': return 'M'; case 'm': return 'K'; case 'n': return 'N'; case 'r': return 'Y';
    case 's': return 'S'; case 't': return 'A'; case 'v': return 'B'; case 'w': return 'W'; case 'y': return 'R';
    default: return '-';
    }
}
static int is_ambiguous(uint8_t b)      /* bit_encoding.rs:58-61 */
{
    b |= 0x20;
    return !(b == 'a' || b == 'c' || b == 'g' || b == 't' || b == 'u' || b == ('-' | 0x20));
}
static int cmp_key2(const ora_key *x, const ora_key *y) { return x->hi != y->hi ? (x->hi < y->hi ? -1 : 1) : (x->lo != y->lo ? (x->lo < y->lo ? -1 : 1) : 0); }

struct ora_ref {
    int k, rc, ambig_mask;
    size_t n_chrom; uint8_t **seq; size_t *seq_len; char **chrom_names;
    size_t n_kmers; ora_key *kmer; uint8_t *is_rc; size_t *pos; uint32_t *chrom;       /* split_kmer_pos */
    size_t n_repeat; size_t *repeat_coors;
    size_t n_mapped, S; uint32_t *m_chrom; size_t *m_pos; uint8_t *m_var; char **names;  /* mapped_pos / mapped_variants / mapped_names */
};

void ora_ref_free(ora_ref *r)
{
    if (!r) return;
    for (size_t c = 0; c < r->n_chrom; c++) { free(r->seq[c]); free(r->chrom_names[c]); }
    free(r->seq); free(r->seq_len); free(r->chrom_names);
    free(r->kmer); free(r->is_rc); free(r->pos); free(r->chrom); free(r->repeat_coors);
    free(r->m_chrom); free(r->m_pos); free(r->m_var);
    if (r->names) for (size_t s = 0; s < r->S; s++) free(r->names[s]);
    free(r->names); free(r);
}

typedef struct { ora_key k; uint32_t n; } kcount;
static int cmp_kcount(const void *a, const void *b) { return cmp_key2(&((const kcount *)a)->k, &((const kcount *)b)->k); }

/* RefSka::new (ska_ref.rs:189-311) */
ora_ref *ora_ref_new(int k, const char *fasta, int rc, int ambig_mask, int repeat_mask)
{
    if (k < 5 || k > 63 || !(k & 1)) { ora_set_error("Invalid k-mer length"); return NULL; }                 /* :190-192 */
    ora_fastx fx;
    if (ora_fastx_read(fasta, &fx)) { ora_set_error("Invalid path/file: %s", fasta); return NULL; }           /* :200-201 */
    if (fx.is_fastq) { ora_fastx_free(&fx); ora_set_error("Cannot create reference from FASTQ files"); return NULL; }   /* :206-208 */
    ora_ref *r = (ora_ref *)calloc(1, sizeof *r);
    r->k = k; r->rc = rc; r->ambig_mask = ambig_mask;
    r->n_chrom = fx.n;
    r->seq = (uint8_t **)calloc(fx.n ? fx.n : 1, sizeof(uint8_t *)); r->seq_len = (size_t *)calloc(fx.n ? fx.n : 1, sizeof(size_t));
    r->chrom_names = (char **)calloc(fx.n ? fx.n : 1, sizeof(char *));
    size_t total = 0;
    for (size_t c = 0; c < fx.n; c++) total += fx.recs[c].len;
    r->kmer = (ora_key *)malloc((total ? total : 1) * sizeof(ora_key)); r->is_rc = (uint8_t *)malloc(total ? total : 1);
    r->pos = (size_t *)malloc((total ? total : 1) * sizeof(size_t)); r->chrom = (uint32_t *)malloc((total ? total : 1) * sizeof(uint32_t));
    for (size_t c = 0; c < fx.n; c++) {
        const ora_rec *rec = &fx.recs[c];
        size_t e = 0;                                                                                          /* id up to the first white space, :209-214 */
        while (e < rec->id_len && rec->id[e] != ' ' && rec->id[e] != '\t') e++;
        r->chrom_names[c] = (char *)malloc(e + 1); memcpy(r->chrom_names[c], rec->id, e); r->chrom_names[c][e] = 0;
        r->seq[c] = (uint8_t *)malloc(rec->len ? rec->len : 1); memcpy(r->seq[c], rec->seq, rec->len); r->seq_len[c] = rec->len;
        uint8_t *flags = (uint8_t *)malloc(rec->len ? rec->len : 1);
        const size_t n = extract_record_pos(rec->seq, rec->len, k, rc, r->kmer + r->n_kmers, flags, r->pos + r->n_kmers, total - r->n_kmers);
        for (size_t i = 0; i < n; i++) { r->is_rc[r->n_kmers + i] = (flags[i] & ORA_F_IS_RC) != 0; r->chrom[r->n_kmers + i] = (uint32_t)c; }
        r->n_kmers += n;
        free(flags);
    }
    ora_fastx_free(&fx);
    if (r->n_kmers == 0) { ora_set_error("%s has no valid sequence", fasta); ora_ref_free(r); return NULL; }  /* :255-257 */
    if (repeat_mask) {                                                                                         /* :259-293 */
        kcount *kc = (kcount *)malloc(r->n_kmers * sizeof *kc);
        for (size_t i = 0; i < r->n_kmers; i++) { kc[i].k = r->kmer[i]; kc[i].n = 1; }
        qsort(kc, r->n_kmers, sizeof *kc, cmp_kcount);
        size_t u = 0;
        for (size_t i = 0; i < r->n_kmers; i++) { if (u && cmp_key2(&kc[u - 1].k, &kc[i].k) == 0) kc[u - 1].n++; else kc[u++] = kc[i]; }
        const size_t half = (size_t)(k - 1) / 2;
        size_t cap = 1024; r->repeat_coors = (size_t *)malloc(cap * sizeof(size_t));
        size_t last_chrom = 0, last_end = 0, chrom_offset = 0;
        for (size_t i = 0; i < r->n_kmers; i++) {
            if (r->chrom[i] > last_chrom) { chrom_offset += r->seq_len[last_chrom]; last_chrom = r->chrom[i]; }
            size_t lo = 0, hi = u;
            while (lo < hi) { size_t mid = (lo + hi) / 2; if (cmp_key2(&kc[mid].k, &r->kmer[i]) < 0) lo = mid + 1; else hi = mid; }
            if (kc[lo].n < 2) continue;                                                                        /* not in `repeats` */
            const size_t start = r->pos[i] - half + chrom_offset, end = r->pos[i] + half + chrom_offset;
            const size_t from = (start > last_end || start == 0) ? start : last_end + 1;
            for (size_t p = from; p < end + 1; p++) {
                if (r->n_repeat == cap) { cap *= 2; r->repeat_coors = (size_t *)realloc(r->repeat_coors, cap * sizeof(size_t)); }
                r->repeat_coors[r->n_repeat++] = p;
            }
            last_chrom = r->chrom[i]; last_end = end;
        }
        free(kc);
    }
    return r;
}

typedef struct { ora_key k; size_t row; } krow;
static int cmp_krow(const void *a, const void *b) { return cmp_key2(&((const krow *)a)->k, &((const krow *)b)->k); }

/* generic_modes::map's to_dict + RefSka::map (ska_ref.rs:508-533) */
int ora_ref_map(ora_ref *r, const ora_array *a)
{
    if (r->k != a->k) { ora_set_error("K-mer sizes do not match ref:%d skf:%zu", r->k, a->nk); return -1; }  /* :509-515 (prints ksize() of the skf) */
    if (a->nk != a->nrows) { ora_set_error("array keys and rows are out of step"); return -1; }
    const size_t S = a->ns;
    krow *kr = (krow *)malloc((a->nk ? a->nk : 1) * sizeof *kr);
    for (size_t i = 0; i < a->nk; i++) { kr[i].k = a->keys[i]; kr[i].row = i; }
    qsort(kr, a->nk, sizeof *kr, cmp_krow);
    r->S = S; r->names = (char **)malloc(S * sizeof(char *));
    for (size_t s = 0; s < S; s++) r->names[s] = strdup(a->names[s]);
    r->m_chrom = (uint32_t *)malloc(r->n_kmers * sizeof(uint32_t)); r->m_pos = (size_t *)malloc(r->n_kmers * sizeof(size_t));
    r->m_var = (uint8_t *)malloc(r->n_kmers * S + 1); r->n_mapped = 0;
    for (size_t i = 0; i < r->n_kmers; i++) {
        size_t lo = 0, hi = a->nk;
        while (lo < hi) { size_t mid = (lo + hi) / 2; if (cmp_key2(&kr[mid].k, &r->kmer[i]) < 0) lo = mid + 1; else hi = mid; }
        if (lo >= a->nk || cmp_key2(&kr[lo].k, &r->kmer[i]) != 0) continue;
        const uint8_t *row = a->var + kr[lo].row * S;
        uint8_t *dst = r->m_var + r->n_mapped * S;
        for (size_t s = 0; s < S; s++) dst[s] = r->is_rc[i] ? rc_iupac(row[s]) : row[s];                       /* :522-528 */
        r->m_chrom[r->n_mapped] = r->chrom[i]; r->m_pos[r->n_mapped] = r->pos[i]; r->n_mapped++;
    }
    free(kr);
    return 0;
}

/* ---- AlnWriter (aln_writer.rs) ---- */
typedef struct {
    size_t next_pos, curr_chrom, last_mapped, last_written, chrom_offset, half;
    const ora_ref *r; uint8_t *out; size_t total;
    uint8_t *mid_base; size_t *mid_pos; size_t n_mid;
} alnw;
static void fill_fwd_bases(alnw *w, size_t maximum)                                                            /* :78-92 */
{
    if (w->last_written > 0) {
        const size_t lm = w->last_mapped + w->half;
        const size_t overhang = lm > w->last_written ? lm - w->last_written : 0;
        const size_t start = w->last_written + 1;
        size_t end = start + overhang; if (end > maximum) end = maximum;
        if (end > start) {
            memcpy(w->out + start + w->chrom_offset, w->r->seq[w->curr_chrom] + start, end - start);
            w->last_written = end;
        }
    }
}
static void fill_contig(alnw *w)                                                                               /* :95-101 */
{
    const size_t len = w->r->seq_len[w->curr_chrom];
    fill_fwd_bases(w, len);
    w->chrom_offset += len; w->curr_chrom += 1; w->next_pos = w->half;
}
static void write_split_kmer(alnw *w, size_t mapped_pos, size_t mapped_chrom, uint8_t base)                    /* :105-137 */
{
    while (mapped_chrom > w->curr_chrom) fill_contig(w);
    w->mid_base[w->n_mid] = (is_ambiguous(base) && w->r->ambig_mask) ? 'N' : base;
    w->mid_pos[w->n_mid++] = mapped_pos + w->chrom_offset;
    if (mapped_pos < w->next_pos) w->last_mapped = mapped_pos;
    else {
        if (mapped_pos > w->next_pos) fill_fwd_bases(w, mapped_pos - w->half);
        const size_t start = mapped_pos - w->half, end = mapped_pos;
        memcpy(w->out + start + w->chrom_offset, w->r->seq[w->curr_chrom] + start, end - start);
        w->next_pos = mapped_pos + w->half + 1; w->last_mapped = mapped_pos; w->last_written = mapped_pos;
    }
}
/* pseudoalignment of one sample (ska_ref.rs:551-583 + AlnWriter::finalise :140-158); returns a malloc'd total-size buffer */
static uint8_t *pseudo_one(const ora_ref *r, size_t sample, size_t *total_out)
{
    size_t total = 0;
    for (size_t c = 0; c < r->n_chrom; c++) total += r->seq_len[c];
    alnw w; memset(&w, 0, sizeof w);
    w.half = (size_t)(r->k - 1) / 2; w.next_pos = w.half; w.r = r; w.total = total;
    w.out = (uint8_t *)malloc(total + 1); memset(w.out, '-', total + 1);
    w.mid_base = (uint8_t *)malloc(r->n_mapped ? r->n_mapped : 1); w.mid_pos = (size_t *)malloc((r->n_mapped ? r->n_mapped : 1) * sizeof(size_t));
    for (size_t i = 0; i < r->n_mapped; i++) {
        const uint8_t base = r->m_var[i * r->S + sample];
        if (base != '-') write_split_kmer(&w, r->m_pos[i], r->m_chrom[i], base);
    }
    while (w.curr_chrom < r->n_chrom) fill_contig(&w);
    for (size_t i = 0; i < w.n_mid; i++) w.out[w.mid_pos[i]] = w.mid_base[i];
    for (size_t i = 0; i < r->n_repeat; i++) if (w.out[r->repeat_coors[i]] != '-') w.out[r->repeat_coors[i]] = 'N';
    free(w.mid_base); free(w.mid_pos);
    *total_out = total;
    return w.out;
}

typedef struct { char *p; size_t n, cap; } sbuf;
static void sb_put(sbuf *b, const void *src, size_t n)
{
    if (b->n + n + 1 > b->cap) { while (b->n + n + 1 > b->cap) b->cap = b->cap ? b->cap * 2 : 4096; b->p = (char *)realloc(b->p, b->cap); }
    memcpy(b->p + b->n, src, n); b->n += n; b->p[b->n] = 0;
}
static void sb_str(sbuf *b, const char *s) { sb_put(b, s, strlen(s)); }

/* write_aln (ska_ref.rs:622-645): ">name\nSEQ\n" per sample, contigs concatenated */
char *ora_ref_write_aln(ora_ref *r, size_t *len)
{
    if (r->n_mapped == 0) { ora_set_error("No split k-mers mapped to reference"); return NULL; }              /* :553-555 */
    sbuf b = {0, 0, 0};
    for (size_t s = 0; s < r->S; s++) {
        size_t total; uint8_t *seq = pseudo_one(r, s, &total);
        sb_str(&b, ">"); sb_str(&b, r->names[s]); sb_str(&b, "\n"); sb_put(&b, seq, total); sb_str(&b, "\n");
        free(seq);
    }
    if (len) *len = b.n;
    return b.p;
}

static char vcf_base(uint8_t b) { return (b == 'A' || b == 'C' || b == 'G' || b == 'T') ? (char)b : 'N'; }   /* u8_to_base :137-146 */

/* write_vcf (ska_ref.rs:648-765) */
char *ora_ref_write_vcf(ora_ref *r, size_t *len)
{
    if (r->n_mapped == 0) { ora_set_error("No split k-mers mapped to reference"); return NULL; }
    size_t total = 0;
    uint8_t **aln = (uint8_t **)malloc(r->S * sizeof(uint8_t *));
    for (size_t s = 0; s < r->S; s++) aln[s] = pseudo_one(r, s, &total);
    sbuf b = {0, 0, 0};
    sb_str(&b, "##fileformat=VCFv4.4\n");
    for (size_t c = 0; c < r->n_chrom; c++) { sb_str(&b, "##contig=<ID="); sb_str(&b, r->chrom_names[c]); sb_str(&b, ">\n"); }
    sb_str(&b, "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT");
    for (size_t s = 0; s < r->S; s++) { sb_str(&b, "\t"); sb_str(&b, r->names[s]); }
    sb_str(&b, "\n");
    size_t chrom = 0, base0 = 0;                                                                               /* IdxCheck (idx_check.rs) */
    int *gt = (int *)malloc(r->S * sizeof(int));
    for (size_t idx = 0; idx < total; idx++) {
        while (chrom < r->n_chrom && idx >= base0 + r->seq_len[chrom]) { base0 += r->seq_len[chrom]; chrom++; }
        const size_t pos = idx - base0;
        const uint8_t ref_base = r->seq[chrom][pos];
        char alts[5]; int n_alt = 0, variant = 0;
        for (size_t s = 0; s < r->S; s++) {
            const uint8_t m = aln[s][idx];
            if (m == ref_base) gt[s] = 0;
            else if (m == '-') { variant = 1; gt[s] = -1; }
            else {
                variant = 1;
                const char ab = vcf_base(m);
                int at = -1;
                for (int q = 0; q < n_alt; q++) if (alts[q] == ab) at = q;
                if (at < 0) { alts[n_alt] = ab; at = n_alt++; }                                                /* Base has 5 values; one could equal the REF letter too */
                gt[s] = at + 1;
            }
        }
        if (!variant) continue;
        char tmp[64];
        sb_str(&b, r->chrom_names[chrom]);
        snprintf(tmp, sizeof tmp, "\t%zu\t.\t%c\t", pos + 1, vcf_base(ref_base)); sb_str(&b, tmp);
        if (n_alt == 0) sb_str(&b, ".");
        for (int q = 0; q < n_alt; q++) { if (q) sb_str(&b, ","); sb_put(&b, &alts[q], 1); }
        sb_str(&b, "\t.\t.\t.\tGT");
        for (size_t s = 0; s < r->S; s++) { if (gt[s] < 0) sb_str(&b, "\t."); else { snprintf(tmp, sizeof tmp, "\t%d", gt[s]); sb_str(&b, tmp); } }
        sb_str(&b, "\n");
    }
    free(gt);
    for (size_t s = 0; s < r->S; s++) free(aln[s]);
    free(aln);
    if (len) *len = b.n;
    return b.p;
}
