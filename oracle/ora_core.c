/*
 * ora_core.c -- CPU ORACLE (test infrastructure, NOT product code).
 * Restatement of ska.rust's ska_dict / merge_ska_dict / merge_ska_array /
 * generic_modes for the build -> merge -> align/distance path.  See
 * ska_oracle.h for scope and the parity pin.
 */
#define _GNU_SOURCE
#include "ora_internal.h"
#include <math.h>
#include <pthread.h>
#include <stdarg.h>
#include <time.h>

/* ------------------------------------------------------------------ misc */
static __thread char g_err[512];
const char *ora_last_error(void) { return g_err; }
void ora_set_error(const char *fmt, ...)
{
    va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof g_err, fmt, ap); va_end(ap);
}
void ora_free(void *p) { free(p); }
double ora_now(void)
{
    struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

static pthread_mutex_t g_tmu = PTHREAD_MUTEX_INITIALIZER;
static ora_timers g_timers;
static void timer_add(double *slot, double dt)
{
    pthread_mutex_lock(&g_tmu); *slot += dt; pthread_mutex_unlock(&g_tmu);
}
void ora_timers_get(ora_timers *t, int reset)
{
    pthread_mutex_lock(&g_tmu);
    if (t) *t = g_timers;
    if (reset) memset(&g_timers, 0, sizeof g_timers);
    pthread_mutex_unlock(&g_tmu);
}

/* ------------------------------------------------- bit_encoding.rs:30-61 */
static const char ORA_LETTER_CODE[4] = { 'A', 'C', 'T', 'G' };          /* :30 */
static inline uint8_t ora_encode_base(uint8_t b) { return (b >> 1) & 0x3; } /* :34-36 */
static inline int ora_valid_base(uint8_t b) { return (b & 0xF) != 14; }     /* :52-54 */
static inline int ora_is_ambiguous(uint8_t b)                               /* :58-61 */
{
    b |= 0x20;
    return !(b == 'a' || b == 'c' || b == 'g' || b == 't' || b == 'u' || b == '-');
}
/* split_kmer.rs:66-71 valid_qual: (q - 33) > min_qual in u8 arithmetic */
static inline int ora_valid_qual(size_t idx, const uint8_t *qual, uint8_t min_qual)
{
    if (!qual) return 1;
    return (uint8_t)(qual[idx] - 33) > min_qual;
}

/* IUPAC[new_base*256 + existing] (bit_encoding.rs:388-453).  The 1024-entry table is exactly
 * "union of base sets" on the IUPAC codes; restated as such.
 * set bits: A=1 C=2 T=4 G=8 (bit index == 2-bit encoding). */
static const char ORA_MASK2IUPAC[16] = { 0, 'A', 'C', 'M', 'T', 'W', 'Y', 'H', 'G', 'R', 'S', 'V', 'K', 'D', 'B', 'N' };
static inline uint8_t ora_iupac_mask(uint8_t c)
{
    switch (c & 0xDF) {
    case 'A': return 1;  case 'C': return 2;  case 'T': return 4;  case 'G': return 8;
    case 'M': return 3;  case 'W': return 5;  case 'Y': return 6;  case 'H': return 7;
    case 'R': return 9;  case 'S': return 10; case 'V': return 11; case 'K': return 12;
    case 'D': return 13; case 'B': return 14; case 'N': return 15;
    default: return 0;
    }
}
static inline uint8_t ora_iupac_update(uint8_t new_base, uint8_t existing)
{
    uint8_t m = ora_iupac_mask(existing);
    if (!m) return 0;                       /* table holds 0 for non-IUPAC existing bytes */
    return (uint8_t)ORA_MASK2IUPAC[m | (1u << new_base)];
}

/* bit_encoding.rs:65-85 base_to_prob, order [A, C, T, G] */
static void ora_base_to_prob(uint8_t base, double p[4])
{
    static const double third = 1.0 / 3.0;
    p[0] = p[1] = p[2] = p[3] = 0.0;
    switch (base) {
    case 'A': p[0] = 1.0; break;
    case 'C': p[1] = 1.0; break;
    case 'G': p[3] = 1.0; break;
    case 'T': case 'U': p[2] = 1.0; break;
    case 'R': p[0] = 0.5; p[3] = 0.5; break;
    case 'Y': p[1] = 0.5; p[2] = 0.5; break;
    case 'S': p[1] = 0.5; p[3] = 0.5; break;
    case 'W': p[0] = 0.5; p[2] = 0.5; break;
    case 'K': p[2] = 0.5; p[3] = 0.5; break;
    case 'M': p[0] = 0.5; p[1] = 0.5; break;
    case 'B': p[1] = third; p[2] = third; p[3] = third; break;
    case 'D': p[0] = third; p[2] = third; p[3] = third; break;
    case 'H': p[0] = third; p[1] = third; p[2] = third; break;
    case 'V': p[0] = third; p[1] = third; p[3] = third; break;
    default: break;  /* N and everything else: zeros */
    }
}

/* ------------------------------------------------------ nthash.rs:12-77 */
typedef struct { int k; uint64_t fh, rh; int has_rh; } ora_nthash;
static const uint64_t ORA_HASH_LOOKUP[4] = { 0x3c8bfbb395c60474ULL, 0x3193c18562a02b4cULL, 0x295549f54be24456ULL, 0x20323ed082572324ULL };
static const uint64_t ORA_RC_HASH_LOOKUP[4] = { 0x295549f54be24456ULL, 0x20323ed082572324ULL, 0x3c8bfbb395c60474ULL, 0x3193c18562a02b4cULL };
static inline uint64_t rotl64(uint64_t x, unsigned r) { r &= 63; return r ? (x << r) | (x >> (64 - r)) : x; }
static inline uint64_t rotr64(uint64_t x, unsigned r) { r &= 63; return r ? (x >> r) | (x << (64 - r)) : x; }
static void ora_nthash_new(ora_nthash *h, const uint8_t *seq, int k, int rc)     /* :35-52 */
{
    uint64_t fh = 0;
    for (int i = 0; i < k; i++) fh ^= rotl64(ORA_HASH_LOOKUP[ora_encode_base(seq[i])], (unsigned)(k - i - 1));
    h->k = k; h->fh = fh; h->has_rh = rc; h->rh = 0;
    if (rc) {
        uint64_t r = 0;
        for (int i = 0; i < k; i++) r ^= rotl64(ORA_RC_HASH_LOOKUP[ora_encode_base(seq[k - 1 - i])], (unsigned)(k - i - 1));
        h->rh = r;
    }
}
static inline void ora_nthash_roll(ora_nthash *h, uint8_t old_base, uint8_t new_base)   /* :55-67 */
{
    h->fh = rotl64(h->fh, 1) ^ rotl64(ORA_HASH_LOOKUP[old_base], (unsigned)h->k) ^ ORA_HASH_LOOKUP[new_base];
    if (h->has_rh)
        h->rh = rotr64(h->rh, 1) ^ rotr64(ORA_RC_HASH_LOOKUP[old_base], 1) ^ rotl64(ORA_RC_HASH_LOOKUP[new_base], (unsigned)h->k - 1);
}
static inline uint64_t ora_nthash_curr(const ora_nthash *h)                            /* :70-76 */
{
    return h->has_rh ? (h->fh < h->rh ? h->fh : h->rh) : h->fh;
}

/* ------------------------------------------- bloom_filter.rs:35-148 KmerFilter */
typedef struct {
    uint64_t buf_size; uint64_t *buffer;   /* blocked bloom */
    uint64_t *ckeys; uint16_t *cvals; size_t ccap, cn;   /* HashMap<u64,u16>; cval 0 == empty */
    uint16_t min_count;
} ora_kmer_filter;

static void ora_filter_new(ora_kmer_filter *f, uint16_t min_count)          /* :93-104 */
{
    memset(f, 0, sizeof *f);
    /* BLOOM_WIDTH = 1<<27, BITS_PER_ENTRY = 12 (bloom_filter.rs:19-24) */
    f->buf_size = (uint64_t)round((double)(1u << 27) * (12.0 / 8.0) / 64.0);
    f->min_count = min_count;
}
static void ora_filter_init(ora_kmer_filter *f)                              /* :109-113 */
{
    if (!f->buffer) f->buffer = (uint64_t *)calloc(f->buf_size, 8);
}
static void ora_filter_free(ora_kmer_filter *f) { free(f->buffer); free(f->ckeys); free(f->cvals); memset(f, 0, sizeof *f); }
static inline uint64_t ora_cheap_mix(uint64_t key) { return (key ^ (key >> 31)) * 0x85D059AA333121CFULL; }       /* :56-58 */
static inline uint64_t ora_reduce(uint64_t key, uint64_t range) { return (uint64_t)(((u128)key * (u128)range) >> 64); } /* :50-52 */
static inline uint64_t ora_fingerprint(uint64_t key)                                                                /* :62-68 */
{
    return (1ULL << (key & 63)) | (1ULL << ((key >> 6) & 63)) | (1ULL << ((key >> 12) & 63)) |
           (1ULL << ((key >> 18) & 63)) | (1ULL << ((key >> 24) & 63));
}
static inline int ora_bloom_add_and_check(ora_kmer_filter *f, uint64_t key)                                         /* :77-86 */
{
    uint64_t fp = ora_fingerprint(key);
    uint64_t *v = &f->buffer[ora_reduce(ora_cheap_mix(key), f->buf_size)];
    if ((*v & fp) == fp) return 1;
    *v |= fp; return 0;
}
static uint16_t *ora_counts_slot(ora_kmer_filter *f, uint64_t key, int *found)
{
    if (f->ccap == 0 || (f->cn + 1) * 10 > f->ccap * 7) {
        size_t ocap = f->ccap; uint64_t *ok = f->ckeys; uint16_t *ov = f->cvals;
        f->ccap = ocap ? ocap * 2 : 1024;
        f->ckeys = (uint64_t *)malloc(f->ccap * 8); f->cvals = (uint16_t *)calloc(f->ccap, 2);
        for (size_t i = 0; i < ocap; i++) if (ov[i]) {
            size_t m = f->ccap - 1, j = (size_t)((ok[i] * 0x9E3779B97F4A7C15ULL) >> 20) & m;
            while (f->cvals[j]) j = (j + 1) & m;
            f->ckeys[j] = ok[i]; f->cvals[j] = ov[i];
        }
        free(ok); free(ov);
    }
    size_t m = f->ccap - 1, j = (size_t)((key * 0x9E3779B97F4A7C15ULL) >> 20) & m;
    for (;;) {
        if (!f->cvals[j]) { f->ckeys[j] = key; *found = 0; return &f->cvals[j]; }
        if (f->ckeys[j] == key) { *found = 1; return &f->cvals[j]; }
        j = (j + 1) & m;
    }
}
/* KmerFilter::filter (:116-148): returns 1 for Ordering::Equal */
static int ora_filter_pass(ora_kmer_filter *f, uint64_t hash)
{
    switch (f->min_count) {
    case 0: case 1: return 1;
    case 2: return ora_bloom_add_and_check(f, hash);
    default:
        if (ora_bloom_add_and_check(f, hash)) {
            uint16_t count = 2; int found;
            uint16_t *c = ora_counts_slot(f, hash, &found);
            if (found) { count = (*c == 0xFFFF) ? 0xFFFF : (uint16_t)(*c + 1); *c = count; }
            else { *c = count; f->cn++; }
            return f->min_count == count;
        }
        return 0;
    }
}

/* ------------------------------------------- generic instantiation u64/u128 */
#define KT uint64_t
#define KT_BITS 64
#define SFX(n) n##_64
#include "ora_kmer_impl.h"
#undef KT
#undef KT_BITS
#undef SFX
#define KT u128
#define KT_BITS 128
#define SFX(n) n##_128
#include "ora_kmer_impl.h"
#undef KT
#undef KT_BITS
#undef SFX

size_t ora_extract_record(const uint8_t *seq, size_t len, const uint8_t *qual, int k, int rc, int min_qual,
                          int qual_filter, int is_reads, ora_key *keys, uint8_t *mid, uint8_t *flags,
                          uint64_t *hashes, size_t cap)
{
    if (k <= 31) return extract_record_64(seq, len, qual, k, rc, min_qual, qual_filter, is_reads, keys, mid, flags, hashes, NULL, cap);
    return extract_record_128(seq, len, qual, k, rc, min_qual, qual_filter, is_reads, keys, mid, flags, hashes, NULL, cap);
}
/* the same enumeration with get_middle_pos() of every window (RefSka::new, ska_ref.rs:215-246) */
size_t extract_record_pos(const uint8_t *seq, size_t len, int k, int rc, ora_key *keys, uint8_t *flags, size_t *pos, size_t cap)
{
    if (k <= 31) return extract_record_64(seq, len, NULL, k, rc, 0, ORA_QUAL_NOFILTER, 0, keys, NULL, flags, NULL, pos, cap);
    return extract_record_128(seq, len, NULL, k, rc, 0, ORA_QUAL_NOFILTER, 0, keys, NULL, flags, NULL, pos, cap);
}

/* ------------------------------------------------------------- SkaDict */
struct ora_dict {
    int k, rc; ora_qual q;
    int bits;
    kmap_64 m64; kmap_128 m128;
    ora_kmer_filter filt;
};

static int valid_k(int k) { return k >= 5 && k <= 63 && (k & 1); }     /* ska_dict.rs:342-344 */

ora_dict *ora_dict_new(int k, int rc, const ora_qual *q)
{
    if (!valid_k(k)) { ora_set_error("Invalid k-mer length"); return NULL; }
    ora_dict *d = (ora_dict *)calloc(1, sizeof *d);
    d->k = k; d->rc = rc; d->q = *q; d->bits = k <= 31 ? 64 : 128;     /* lib.rs:592 */
    if (d->bits == 64) kmap_init_64(&d->m64, 1024); else kmap_init_128(&d->m128, 1024);
    ora_filter_new(&d->filt, q->min_count);
    return d;
}
void ora_dict_free(ora_dict *d)
{
    if (!d) return;
    if (d->bits == 64) kmap_free_64(&d->m64); else kmap_free_128(&d->m128);
    ora_filter_free(&d->filt);
    free(d);
}
void ora_dict_add_record(ora_dict *d, const uint8_t *seq, size_t len, const uint8_t *qual, int is_reads)
{
    if (is_reads) ora_filter_init(&d->filt);
    if (d->bits == 64) dict_add_record_64(&d->m64, &d->filt, seq, len, qual, d->k, d->rc, &d->q, is_reads);
    else dict_add_record_128(&d->m128, &d->filt, seq, len, qual, d->k, d->rc, &d->q, is_reads);
}
size_t ora_dict_size(const ora_dict *d) { return d->bits == 64 ? d->m64.n : d->m128.n; }
int ora_dict_key_bits(const ora_dict *d) { return d->bits; }

/* add_file_kmers (ska_dict.rs:118-180) */
static int dict_add_file(ora_dict *d, const char *path, int is_reads, double proportion_reads)
{
    size_t step = 1;
    if (proportion_reads > 0.0) step = (size_t)round(1.0 / proportion_reads);   /* :125-127 */
    if (step == 0) step = 1;
    double t0 = ora_now();
    ora_fastx fx;
    if (ora_fastx_read(path, &fx)) return -1;
    double t1 = ora_now();
    for (size_t r = 0; r < fx.n; r++) {
        if (r % step != 0) continue;                                             /* :134-141 */
        ora_dict_add_record(d, fx.recs[r].seq, fx.recs[r].len, is_reads ? fx.recs[r].qual : NULL, is_reads);
    }
    double t2 = ora_now();
    timer_add(&g_timers.read_parse, t1 - t0); timer_add(&g_timers.dict, t2 - t1);
    ora_fastx_free(&fx);
    return 0;
}

/* SkaDict::new (ska_dict.rs:333-378) */
ora_dict *ora_dict_from_files(int k, int rc, const char *file1, const char *file2, const ora_qual *q, double proportion_reads)
{
    ora_dict *d = ora_dict_new(k, rc, q);
    if (!d) return NULL;
    /* peek first record for the format (:356-366) */
    ora_fastx peek;
    if (ora_fastx_read(file1, &peek)) { ora_dict_free(d); return NULL; }
    int is_reads = peek.is_fastq;
    ora_fastx_free(&peek);
    if (is_reads) ora_filter_init(&d->filt);
    if (dict_add_file(d, file1, is_reads, proportion_reads)) { ora_dict_free(d); return NULL; }
    if (file2 && dict_add_file(d, file2, is_reads, proportion_reads)) { ora_dict_free(d); return NULL; }
    if (ora_dict_size(d) == 0) {                                                  /* :374-376 */
        ora_set_error("%s has no valid sequence", file1);
        ora_dict_free(d); return NULL;
    }
    return d;
}

typedef struct { ora_key k; uint8_t b; } kb_pair;
static int cmp_kb(const void *a, const void *b)
{
    const kb_pair *x = (const kb_pair *)a, *y = (const kb_pair *)b;
    if (x->k.hi != y->k.hi) return x->k.hi < y->k.hi ? -1 : 1;
    if (x->k.lo != y->k.lo) return x->k.lo < y->k.lo ? -1 : 1;
    return 0;
}
void ora_dict_export_sorted(const ora_dict *d, ora_key *keys, uint8_t *bases)
{
    size_t n = ora_dict_size(d), j = 0;
    kb_pair *p = (kb_pair *)malloc((n ? n : 1) * sizeof *p);
    if (d->bits == 64) {
        for (size_t i = 0; i < d->m64.cap; i++) if (d->m64.vals[i]) { p[j].k.lo = d->m64.keys[i]; p[j].k.hi = 0; p[j].b = d->m64.vals[i]; j++; }
    } else {
        for (size_t i = 0; i < d->m128.cap; i++) if (d->m128.vals[i]) {
            p[j].k.lo = (uint64_t)d->m128.keys[i]; p[j].k.hi = (uint64_t)(d->m128.keys[i] >> 64); p[j].b = d->m128.vals[i]; j++;
        }
    }
    qsort(p, n, sizeof *p, cmp_kb);
    for (size_t i = 0; i < n; i++) { if (keys) keys[i] = p[i].k; if (bases) bases[i] = p[i].b; }
    free(p);
}

/* ------------------------------------------- MergeSkaDict build drivers */
typedef struct {
    int bits; mdict_64 m64; mdict_128 m128;
    char **names; size_t n_samples;
    int failed; char err[512];
} mdict_any;

static void mdict_any_init(mdict_any *m, int k, size_t n, int rc)
{
    memset(m, 0, sizeof *m);
    m->bits = k <= 31 ? 64 : 128; m->n_samples = n;
    m->names = (char **)calloc(n ? n : 1, sizeof(char *));
    if (m->bits == 64) mdict_init_64(&m->m64, k, n, rc); else mdict_init_128(&m->m128, k, n, rc);
}
static void mdict_any_free(mdict_any *m)
{
    if (m->bits == 64) mdict_free_64(&m->m64); else mdict_free_128(&m->m128);
    if (m->names) { for (size_t i = 0; i < m->n_samples; i++) free(m->names[i]); free(m->names); }
    m->names = NULL;
}
static size_t mdict_any_size(const mdict_any *m) { return m->bits == 64 ? m->m64.n : m->m128.n; }

/* MergeSkaDict::append (merge_ska_dict.rs:77-109) */
static void mdict_any_append(mdict_any *m, const ora_dict *d, size_t idx, const char *name)
{
    double t0 = ora_now();
    free(m->names[idx]); m->names[idx] = strdup(name);
    if (m->bits == 64) mdict_append_64(&m->m64, &d->m64, idx); else mdict_append_128(&m->m128, &d->m128, idx);
    timer_add(&g_timers.append, ora_now() - t0);
}
/* MergeSkaDict::merge (merge_ska_dict.rs:119-151) */
static void mdict_any_merge(mdict_any *self, mdict_any *other)
{
    double t0 = ora_now();
    if (mdict_any_size(other) > 0) {
        if (mdict_any_size(self) == 0) {
            char **t = self->names; self->names = other->names; other->names = t;
        } else {
            for (size_t i = 0; i < self->n_samples; i++)
                if (!self->names[i] || !self->names[i][0]) { char *t = self->names[i]; self->names[i] = other->names[i]; other->names[i] = t; }
        }
        if (self->bits == 64) mdict_merge_64(&self->m64, &other->m64); else mdict_merge_128(&self->m128, &other->m128);
    }
    if (other->failed && !self->failed) { self->failed = 1; memcpy(self->err, other->err, sizeof self->err); }
    timer_add(&g_timers.merge, ora_now() - t0);
}

typedef struct {
    const char *const *names, *const *file1, *const *file2;
    int k, rc; const ora_qual *q; double prop;
} build_args;

/* multi_append (merge_ska_dict.rs:230-256) */
static void multi_append(mdict_any *out, const build_args *a, size_t lo, size_t hi, size_t total)
{
    mdict_any_init(out, a->k, total, a->rc);
    for (size_t i = lo; i < hi; i++) {
        ora_dict *d = ora_dict_from_files(a->k, a->rc, a->file1[i], a->file2 ? a->file2[i] : NULL, a->q, a->prop);
        if (!d) { out->failed = 1; snprintf(out->err, sizeof out->err, "%s", ora_last_error()); return; }
        mdict_any_append(out, d, i, a->names[i]);
        ora_dict_free(d);
    }
}

/* parallel_append (merge_ska_dict.rs:264-326): rayon::join over halves */
typedef struct { mdict_any *out; const build_args *a; size_t lo, hi, total; int depth; } par_task;
static void parallel_append(mdict_any *out, const build_args *a, size_t lo, size_t hi, size_t total, int depth);
static void *par_thread(void *p)
{
    par_task *t = (par_task *)p;
    parallel_append(t->out, t->a, t->lo, t->hi, t->total, t->depth);
    return NULL;
}
static void parallel_append(mdict_any *out, const build_args *a, size_t lo, size_t hi, size_t total, int depth)
{
    if (depth == 0) { multi_append(out, a, lo, hi, total); return; }
    size_t mid = lo + (hi - lo) / 2;                  /* split_at(len/2), :272-275 */
    mdict_any right;
    par_task lt = { out, a, lo, mid, total, depth - 1 };
    pthread_t th; pthread_create(&th, NULL, par_thread, &lt);
    parallel_append(&right, a, mid, hi, total, depth - 1);
    pthread_join(th, NULL);
    mdict_any_merge(out, &right);
    mdict_any_free(&right);
}

static int cmp_key(const ora_key *x, const ora_key *y)
{
    if (x->hi != y->hi) return x->hi < y->hi ? -1 : 1;
    if (x->lo != y->lo) return x->lo < y->lo ? -1 : 1;
    return 0;
}
typedef struct { ora_key k; uint32_t rid; } kr_pair;
static int cmp_kr(const void *a, const void *b) { return cmp_key(&((const kr_pair *)a)->k, &((const kr_pair *)b)->k); }

/* MergeSkaArray::new (merge_ska_array.rs:166-186); rows emitted sorted by key */
static ora_array *array_from_mdict(mdict_any *m, int k, int rc)
{
    double t0 = ora_now();
    size_t U = mdict_any_size(m), S = m->n_samples;
    ora_array *a = (ora_array *)calloc(1, sizeof *a);
    a->k = k; a->rc = rc; a->k_bits = m->bits; a->nk = a->nrows = U; a->ns = S;
    a->keys = (ora_key *)malloc((U ? U : 1) * sizeof(ora_key));
    a->var = (uint8_t *)malloc(U * S + 1);
    a->counts = (uint64_t *)malloc((U ? U : 1) * 8);
    a->names = (char **)calloc(S ? S : 1, sizeof(char *));
    for (size_t i = 0; i < S; i++) a->names[i] = strdup(m->names[i] ? m->names[i] : "");
    a->version = strdup("0.5.2");                      /* CARGO_PKG_VERSION, Cargo.toml:3 */
    kr_pair *p = (kr_pair *)malloc((U ? U : 1) * sizeof *p);
    size_t j = 0;
    if (m->bits == 64) {
        for (size_t i = 0; i < m->m64.cap; i++) if (m->m64.rowid[i]) { p[j].k.lo = m->m64.keys[i]; p[j].k.hi = 0; p[j].rid = m->m64.rowid[i]; j++; }
    } else {
        for (size_t i = 0; i < m->m128.cap; i++) if (m->m128.rowid[i]) {
            p[j].k.lo = (uint64_t)m->m128.keys[i]; p[j].k.hi = (uint64_t)(m->m128.keys[i] >> 64); p[j].rid = m->m128.rowid[i]; j++;
        }
    }
    qsort(p, U, sizeof *p, cmp_kr);
    for (size_t r = 0; r < U; r++) {
        const uint8_t *row = m->bits == 64 ? mdict_row_64(&m->m64, p[r].rid) : mdict_row_128(&m->m128, p[r].rid);
        a->keys[r] = p[r].k;
        uint64_t c = 0;
        for (size_t s = 0; s < S; s++) {
            uint8_t b = row[s];
            if (b != 0 && b != '-') c++;                /* :172 */
            a->var[r * S + s] = b > '-' ? b : '-';      /* mapv_inplace(max(b,'-')) :175 */
        }
        a->counts[r] = c;
    }
    free(p);
    timer_add(&g_timers.to_array, ora_now() - t0);
    return a;
}

/* build_and_merge (merge_ska_dict.rs:354-417) */
ora_array *ora_build_and_merge(const char *const *names, const char *const *file1, const char *const *file2,
                               int n, int k, int rc, const ora_qual *q, int threads, double proportion_reads)
{
    if (!valid_k(k)) { ora_set_error("Invalid k-mer length"); return NULL; }
    build_args a = { names, file1, file2, k, rc, q, proportion_reads };
    size_t total = (size_t)n;
    size_t max_threads = (size_t)threads < 1 + total / 10 ? (size_t)threads : 1 + total / 10;   /* :384 */
    if (max_threads < 1) max_threads = 1;
    int max_depth = (int)floor(log2((double)max_threads));                                         /* :385 */
    mdict_any m;
    if (max_depth > 0) parallel_append(&m, &a, 0, total, total, max_depth);
    else multi_append(&m, &a, 0, total, total);
    if (m.failed) { ora_set_error("%s", m.err); mdict_any_free(&m); return NULL; }
    ora_array *arr = array_from_mdict(&m, k, rc);
    mdict_any_free(&m);
    return arr;
}

ora_array *ora_array_from_dicts(ora_dict *const *dicts, const char *const *names, int n)
{
    if (n <= 0) { ora_set_error("no dicts"); return NULL; }
    mdict_any m; mdict_any_init(&m, dicts[0]->k, (size_t)n, dicts[0]->rc);
    for (int i = 0; i < n; i++) {
        if (dicts[i]->k != dicts[0]->k) { ora_set_error("K-mer lengths do not match: %d %d", dicts[i]->k, dicts[0]->k); mdict_any_free(&m); return NULL; }
        if (dicts[i]->rc != dicts[0]->rc) { ora_set_error("Strand use inconsistent"); mdict_any_free(&m); return NULL; }
        mdict_any_append(&m, dicts[i], (size_t)i, names[i]);
    }
    ora_array *arr = array_from_mdict(&m, dicts[0]->k, dicts[0]->rc);
    mdict_any_free(&m);
    return arr;
}

/* ------------------------------------------------------- MergeSkaArray */
void ora_array_free(ora_array *a)
{
    if (!a) return;
    free(a->keys); free(a->var); free(a->counts);
    if (a->names) { for (size_t i = 0; i < a->ns; i++) free(a->names[i]); free(a->names); }
    free(a->version); free(a);
}
int ora_array_k(const ora_array *a) { return a->k; }
int ora_array_rc(const ora_array *a) { return a->rc; }
int ora_array_k_bits(const ora_array *a) { return a->k_bits; }
size_t ora_array_nrows(const ora_array *a) { return a->nrows; }
size_t ora_array_nkmers(const ora_array *a) { return a->nk; }
size_t ora_array_nsamples(const ora_array *a) { return a->ns; }
const char *ora_array_name(const ora_array *a, size_t i) { return a->names[i]; }
const char *ora_array_version(const ora_array *a) { return a->version; }
void ora_array_export(const ora_array *a, ora_key *keys, uint8_t *variants, uint64_t *counts)
{
    if (keys) memcpy(keys, a->keys, a->nk * sizeof(ora_key));
    if (variants) memcpy(variants, a->var, a->nrows * a->ns);
    if (counts) memcpy(counts, a->counts, a->nrows * 8);
}
void ora_array_sort_rows(ora_array *a)
{
    if (a->nk != a->nrows) return;
    size_t U = a->nrows, S = a->ns;
    kr_pair *p = (kr_pair *)malloc((U ? U : 1) * sizeof *p);
    for (size_t i = 0; i < U; i++) { p[i].k = a->keys[i]; p[i].rid = (uint32_t)i; }
    qsort(p, U, sizeof *p, cmp_kr);
    uint8_t *nv = (uint8_t *)malloc(U * S + 1); uint64_t *nc = (uint64_t *)malloc((U ? U : 1) * 8);
    for (size_t i = 0; i < U; i++) { a->keys[i] = p[i].k; memcpy(nv + i * S, a->var + (size_t)p[i].rid * S, S); nc[i] = a->counts[p[i].rid]; }
    free(a->var); free(a->counts); a->var = nv; a->counts = nc; free(p);
}

/* update_counts (merge_ska_array.rs:139-163) */
static void update_counts(ora_array *a, int filter_ambig_as_missing)
{
    size_t S = a->ns, w = 0;
    for (size_t r = 0; r < a->nrows; r++) {
        const uint8_t *row = a->var + r * S;
        uint64_t c = 0;
        for (size_t s = 0; s < S; s++)
            if (row[s] != '-' && (!filter_ambig_as_missing || !ora_is_ambiguous(row[s]))) c++;
        if (c > 0) {
            if (w != r) memmove(a->var + w * S, row, S);
            a->counts[w] = c;
            if (r < a->nk) a->keys[w] = a->keys[r];
            w++;
        }
    }
    a->nrows = w; a->nk = w;
}

/* ---- skf life-cycle: `ska merge` / `ska delete` / `ska weed` ------------------------------------ */

/* generic_modes::merge (generic_modes.rs:90-106): to_dict (merge_ska_array.rs:208-222) + MergeSkaDict::extend
 * (merge_ska_dict.rs:160-193) folded over the inputs + MergeSkaArray::new (merge_ska_array.rs:166-186).  Rows of the
 * result = union of the inputs' split k-mers (sorted by key here; the reference's order is the hash map's), columns = the
 * inputs' samples in order, absent = '-', variant_count = #cells that are neither 0 nor '-' (:172). */
ora_array *ora_array_merge(const ora_array *const *in, int n)
{
    if (n <= 0) { ora_set_error("no input arrays"); return NULL; }
    size_t tot_rows = 0, S = 0;
    for (int i = 0; i < n; i++) {
        if (in[i]->k != in[0]->k) { ora_set_error("K-mer lengths do not match: %d %d", in[i]->k, in[0]->k); return NULL; }   /* :169-171 */
        if (in[i]->rc != in[0]->rc) { ora_set_error("Strand use inconsistent"); return NULL; }                                /* :172-174 */
        if (in[i]->nk != in[i]->nrows) { ora_set_error("array keys and rows are out of step"); return NULL; }
        tot_rows += in[i]->nrows; S += in[i]->ns;
    }
    /* all (key, source, row) triples sorted by key: one output row per distinct key */
    typedef struct { ora_key k; uint32_t src, rid; } ksr;
    ksr *p = (ksr *)malloc((tot_rows ? tot_rows : 1) * sizeof *p);
    size_t w = 0;
    for (int i = 0; i < n; i++) for (size_t r = 0; r < in[i]->nrows; r++) { p[w].k = in[i]->keys[r]; p[w].src = (uint32_t)i; p[w].rid = (uint32_t)r; w++; }
    /* stable insertion of the source index into the comparison keeps equal keys grouped; order inside a group is free */
    for (size_t gap = tot_rows / 2; gap > 0; gap /= 2)          /* shell sort: no dependence on qsort_r */
        for (size_t i = gap; i < tot_rows; i++) {
            ksr t = p[i]; size_t j = i;
            while (j >= gap && cmp_key(&p[j - gap].k, &t.k) > 0) { p[j] = p[j - gap]; j -= gap; }
            p[j] = t;
        }
    size_t U = 0;
    for (size_t i = 0; i < tot_rows; i++) if (i == 0 || cmp_key(&p[i].k, &p[i - 1].k) != 0) U++;
    ora_array *a = (ora_array *)calloc(1, sizeof *a);
    a->k = in[0]->k; a->rc = in[0]->rc; a->k_bits = in[0]->k_bits; a->nk = a->nrows = U; a->ns = S;
    a->keys = (ora_key *)malloc((U ? U : 1) * sizeof(ora_key));
    a->var = (uint8_t *)malloc(U * S + 1); memset(a->var, '-', U * S + 1);
    a->counts = (uint64_t *)calloc(U ? U : 1, 8);
    a->names = (char **)malloc(S * sizeof(char *));
    size_t *col0 = (size_t *)malloc(n * sizeof(size_t)), c = 0;
    for (int i = 0; i < n; i++) { col0[i] = c; for (size_t s = 0; s < in[i]->ns; s++) a->names[c++] = strdup(in[i]->names[s]); }
    a->version = strdup(in[0]->version ? in[0]->version : "");
    size_t row = (size_t)-1;
    for (size_t i = 0; i < tot_rows; i++) {
        if (i == 0 || cmp_key(&p[i].k, &p[i - 1].k) != 0) { row++; a->keys[row] = p[i].k; }
        const ora_array *src = in[p[i].src];
        memcpy(a->var + row * S + col0[p[i].src], src->var + (size_t)p[i].rid * src->ns, src->ns);
    }
    for (size_t r = 0; r < U; r++) { uint64_t cnt = 0; for (size_t s = 0; s < S; s++) { uint8_t b = a->var[r * S + s]; cnt += b != 0 && b != '-'; } a->counts[r] = cnt; }
    free(p); free(col0);
    return a;
}

/* MergeSkaArray::delete_samples (merge_ska_array.rs:231-271): 0 on success, -1 + error text where the reference panics */
int ora_array_delete_samples(ora_array *a, const char *const *del_names, int n_del)
{
    if (n_del <= 0 || (size_t)n_del == a->ns) { ora_set_error("Invalid number of samples to remove"); return -1; }      /* :232-234 */
    /* a name set: duplicates in the request collapse, every name must match one column (first match, :243-249) */
    uint8_t *drop = (uint8_t *)calloc(a->ns, 1), *found = (uint8_t *)calloc(n_del, 1);
    for (int d = 0; d < n_del; d++) {
        int dup = 0;
        for (int e = 0; e < d; e++) if (!strcmp(del_names[e], del_names[d])) { dup = 1; found[d] = 1; }
        if (dup) continue;
        for (size_t s = 0; s < a->ns; s++) if (!drop[s] && !strcmp(a->names[s], del_names[d])) { drop[s] = 1; found[d] = 1; break; }
    }
    for (int d = 0; d < n_del; d++) if (!found[d]) { ora_set_error("Could not find sample(s): {\"%s\"}", del_names[d]); free(drop); free(found); return -1; }   /* :252-254 */
    size_t S = a->ns, S2 = 0;
    for (size_t s = 0; s < S; s++) S2 += !drop[s];
    uint8_t *nv = (uint8_t *)malloc(a->nrows * S2 + 1);
    for (size_t r = 0; r < a->nrows; r++) { size_t c = 0; for (size_t s = 0; s < S; s++) if (!drop[s]) nv[r * S2 + c++] = a->var[r * S + s]; }
    size_t c = 0;
    for (size_t s = 0; s < S; s++) { if (drop[s]) free(a->names[s]); else a->names[c++] = a->names[s]; }
    free(a->var); a->var = nv; a->ns = S2;
    free(drop); free(found);
    update_counts(a, 0);                                                                                                  /* :270 */
    return 0;
}

/* MergeSkaArray::weed (merge_ska_array.rs:452-487) with the split k-mers of a RefSka (ska_ref.rs:189-262, kmer_iter :541):
 * every canonical split k-mer of the weed FASTA's records, whatever its middle base = the keys of the file's SkaDict */
int ora_array_weed(ora_array *a, const ora_key *weed_keys, size_t n_weed, int reverse)
{
    if (a->nk != a->nrows) { ora_set_error("array keys and rows are out of step"); return -1; }
    ora_key *wk = (ora_key *)malloc((n_weed ? n_weed : 1) * sizeof *wk);
    memcpy(wk, weed_keys, n_weed * sizeof *wk);
    for (size_t gap = n_weed / 2; gap > 0; gap /= 2)
        for (size_t i = gap; i < n_weed; i++) { ora_key t = wk[i]; size_t j = i; while (j >= gap && cmp_key(&wk[j - gap], &t) > 0) { wk[j] = wk[j - gap]; j -= gap; } wk[j] = t; }
    size_t S = a->ns, w = 0;
    for (size_t r = 0; r < a->nrows; r++) {
        size_t lo = 0, hi = n_weed;
        while (lo < hi) { size_t mid = (lo + hi) / 2; if (cmp_key(&wk[mid], &a->keys[r]) < 0) lo = mid + 1; else hi = mid; }
        const int found = lo < n_weed && cmp_key(&wk[lo], &a->keys[r]) == 0;
        if ((!reverse && !found) || (reverse && found)) {                                                                 /* :468 */
            if (w != r) { memmove(a->var + w * S, a->var + r * S, S); a->keys[w] = a->keys[r]; a->counts[w] = a->counts[r]; }
            w++;
        }
    }
    a->nrows = a->nk = w;
    free(wk);
    return 0;
}

/* generic_modes::weed (generic_modes.rs:207-262): weed file (FASTA only, RefSka::new) then the optional filter with a
 * FLOOR threshold and update_kmers = true */
int ora_weed(ora_array *a, const char *weed_fasta, int reverse, double min_freq, int filter_ambig_as_missing, int filter_type,
             int ambig_mask, int ignore_const_gaps)
{
    if (weed_fasta) {
        ora_fastx peek;                                                                                                   /* ska_ref.rs:206-208 */
        if (ora_fastx_read(weed_fasta, &peek)) return -1;
        const int fq = peek.is_fastq;
        ora_fastx_free(&peek);
        if (fq) { ora_set_error("Cannot create reference from FASTQ files"); return -1; }
        ora_qual q = {1, 0, 0};
        ora_dict *d = ora_dict_from_files(a->k, a->rc, weed_fasta, NULL, &q, 0.0);
        if (!d) return -1;
        size_t n = ora_dict_size(d);
        ora_key *keys = (ora_key *)malloc((n ? n : 1) * sizeof *keys); uint8_t *bases = (uint8_t *)malloc(n ? n : 1);
        ora_dict_export_sorted(d, keys, bases);
        int r = ora_array_weed(a, keys, n, reverse);
        free(keys); free(bases); ora_dict_free(d);
        if (r) return r;
    }
    const size_t thr = (size_t)floor((double)a->ns * min_freq);                                                           /* :249 */
    if (thr > 0 || filter_type != 0 || ambig_mask || ignore_const_gaps)
        ora_array_filter(a, thr, filter_ambig_as_missing, filter_type, ambig_mask, ignore_const_gaps, 1);
    return 0;
}

/* filter (merge_ska_array.rs:289-402) */
int32_t ora_array_filter(ora_array *a, size_t min_count, int filter_ambig_as_missing, int filter_type,
                         int mask_ambig, int ignore_const_gaps, int update_kmers)
{
    double t0 = ora_now();
    size_t S = a->ns, w = 0;
    int32_t removed = 0;
    if (filter_ambig_as_missing) update_counts(a, 1);                    /* :308-310 */
    for (size_t r = 0; r < a->nrows; r++) {
        const uint8_t *row = a->var + r * S;
        int keep = 0;
        if (a->counts[r] >= min_count) {
            switch (filter_type) {
            case ORA_FILTER_NONE: keep = 1; break;
            case ORA_FILTER_NO_CONST: {                                  /* :322-334 */
                uint8_t seen[256] = { 0 }; int n = 0;
                for (size_t s = 0; s < S && n <= 1; s++)
                    if (!ignore_const_gaps || row[s] != '-') { if (!seen[row[s]]) { seen[row[s]] = 1; n++; } }
                keep = n > 1; break;
            }
            case ORA_FILTER_NO_AMBIG: {                                  /* :335-344 */
                keep = 1;
                for (size_t s = 0; s < S; s++) if (ora_is_ambiguous(row[s])) { keep = 0; break; }
                break;
            }
            case ORA_FILTER_NO_AMBIG_OR_CONST: {                         /* :345-367 */
                uint8_t seen[256] = { 0 }; int n = 0;
                for (size_t s = 0; s < S; s++) seen[row[s]] = 1;
                for (int b = 0; b < 256; b++) if (seen[b]) {
                    uint8_t lb = (uint8_t)(b | 0x20);
                    if (lb == 'a' || lb == 'c' || lb == 'g' || lb == 't' || lb == 'u') n++;
                    else if (lb == '-') n += ignore_const_gaps ? 0 : 1;
                }
                keep = n > 1; break;
            }
            default: keep = 1;
            }
        }
        if (keep) {
            if (w != r) memmove(a->var + w * S, row, S);
            a->counts[w] = a->counts[r];
            if (update_kmers && r < a->nk) a->keys[w] = a->keys[r];
            w++;
        } else removed++;
    }
    a->nrows = w;
    if (update_kmers) a->nk = w;
    if (mask_ambig)                                                      /* :388-399 */
        for (size_t i = 0; i < a->nrows * S; i++) if (ora_is_ambiguous(a->var[i])) a->var[i] = 'N';
    timer_add(&g_timers.filter, ora_now() - t0);
    return removed;
}

/* apply_filters (generic_modes.rs:112-131) */
int32_t ora_apply_filters(ora_array *a, double min_freq, int filter_ambig_as_missing, int filter_type,
                          int ambig_mask, int ignore_const_gaps)
{
    size_t thr = (size_t)ceil((double)a->ns * min_freq);
    return ora_array_filter(a, thr, filter_ambig_as_missing, filter_type, ambig_mask, ignore_const_gaps, 0);
}

/* write_fasta (merge_ska_array.rs:499-517): ">name\nSEQ\n", unwrapped, Unix endings */
char *ora_array_fasta(const ora_array *a, size_t *len)
{
    double t0 = ora_now();
    size_t S = a->ns, U = a->nrows, tot = 0;
    for (size_t s = 0; s < S; s++) tot += strlen(a->names[s]) + U + 3;
    char *out = (char *)malloc(tot + 1), *p = out;
    for (size_t s = 0; s < S; s++) {
        *p++ = '>'; size_t nl = strlen(a->names[s]); memcpy(p, a->names[s], nl); p += nl; *p++ = '\n';
        for (size_t r = 0; r < U; r++) *p++ = (char)a->var[r * S + s];
        *p++ = '\n';
    }
    *p = 0; *len = (size_t)(p - out);
    timer_add(&g_timers.fasta, ora_now() - t0);
    return out;
}

/* variant_dist (merge_ska_array.rs:587-632) */
static ora_dist variant_dist(const ora_array *a, size_t i, size_t j, double constant, int filt_ambig)
{
    double distance = 0.0, mismatches = 0.0, matches = constant;
    size_t S = a->ns;
    for (size_t r = 0; r < a->nrows; r++) {
        uint8_t v1 = a->var[r * S + i], v2 = a->var[r * S + j];
        if (v1 == '-' || v2 == '-') {
            if (!(v1 == '-' && v2 == '-')) mismatches += 1.0;
        } else if (filt_ambig) {
            if (!ora_is_ambiguous(v1) && !ora_is_ambiguous(v2)) { matches += 1.0; if (v1 != v2) distance += 1.0; }
        } else {
            double p1[4], p2[4], ov = 0.0;
            ora_base_to_prob(v1, p1); ora_base_to_prob(v2, p2);
            for (int b = 0; b < 4; b++) ov += p1[b] * p2[b];
            if (ov > 0.0) matches += 1.0;
            distance += 1.0 - ov;
        }
    }
    ora_dist d;
    d.distance = distance;
    d.mismatch_prop = (matches + mismatches) == 0.0 ? 0.0 : mismatches / (matches + mismatches);
    d.match_count = (uint64_t)matches; d.mismatch_count = (uint64_t)mismatches;
    return d;
}
void ora_array_distance(const ora_array *a, double constant, int filt_ambig, ora_dist *out)
{
    size_t n = 0;
    for (size_t i = 0; i < a->ns; i++)
        for (size_t j = i + 1; j < a->ns; j++) out[n++] = variant_dist(a, i, j, constant, filt_ambig);
}

typedef struct { char *p; size_t n, cap; } sbuf;
static void sb_put(sbuf *b, const char *s, size_t n)
{
    if (b->n + n + 1 > b->cap) { b->cap = (b->n + n + 1) * 2; b->p = (char *)realloc(b->p, b->cap); }
    memcpy(b->p + b->n, s, n); b->n += n; b->p[b->n] = 0;
}
static void sb_printf(sbuf *b, const char *fmt, ...)
{
    char tmp[512]; va_list ap; va_start(ap, fmt); int n = vsnprintf(tmp, sizeof tmp, fmt, ap); va_end(ap);
    if (n < (int)sizeof tmp) { sb_put(b, tmp, (size_t)n); return; }
    char *big = (char *)malloc((size_t)n + 1); va_start(ap, fmt); vsnprintf(big, (size_t)n + 1, fmt, ap); va_end(ap);
    sb_put(b, big, (size_t)n); free(big);
}

/* generic_modes::distance (generic_modes.rs:136-189) + VariantDist Display (merge_ska_array.rs:57-65) */
char *ora_distance_tsv(ora_array *a, double min_freq, int filt_ambig, size_t *len)
{
    if (min_freq * (double)a->ns >= 1.0) ora_apply_filters(a, min_freq, 0, ORA_FILTER_NONE, 0, 0);
    int32_t constant = ora_apply_filters(a, 0.0, 0, ORA_FILTER_NO_CONST, 0, 0);
    size_t S = a->ns, np = S * (S - 1) / 2;
    ora_dist *d = (ora_dist *)malloc((np ? np : 1) * sizeof *d);
    ora_array_distance(a, (double)constant, filt_ambig, d);
    sbuf b = { 0 };
    sb_printf(&b, "Sample1\tSample2\tDistance\tMismatches (proportion)\tMatch count\tMismatch count\n");
    size_t n = 0;
    for (size_t i = 0; i < S; i++)
        for (size_t j = i + 1; j < S; j++, n++)
            sb_printf(&b, "%s\t%s\t%.2f\t%.5f\t%llu\t%llu\n", a->names[i], a->names[j], d[n].distance, d[n].mismatch_prop,
                      (unsigned long long)d[n].match_count, (unsigned long long)d[n].mismatch_count);
    free(d);
    *len = b.n;
    return b.p;
}

/* generic_modes::align (generic_modes.rs:22-50) */
char *ora_align_fasta(ora_array *a, int filter_type, int mask_ambig, int ignore_const_gaps, double min_freq,
                      int filter_ambig_as_missing, size_t *len)
{
    ora_apply_filters(a, min_freq, filter_ambig_as_missing, filter_type, mask_ambig, ignore_const_gaps);
    return ora_array_fasta(a, len);
}

/* Rust `{:?}` of a String */
static void sb_rust_debug_str(sbuf *b, const char *s)
{
    sb_put(b, "\"", 1);
    for (; *s; s++) {
        unsigned char c = (unsigned char)*s;
        if (c == '"') sb_put(b, "\\\"", 2);
        else if (c == '\\') sb_put(b, "\\\\", 2);
        else if (c == '\n') sb_put(b, "\\n", 2);
        else if (c == '\r') sb_put(b, "\\r", 2);
        else if (c == '\t') sb_put(b, "\\t", 2);
        else if (c < 0x20 || c == 0x7f) sb_printf(b, "\\u{%x}", c);
        else sb_put(b, (const char *)&c, 1);
    }
    sb_put(b, "\"", 1);
}

/* decode_kmer (bit_encoding.rs:307-335) */
static void decode_arm(u128 bits, int half, char *out)
{
    for (int i = 0; i < half; i++) { out[half - 1 - i] = ORA_LETTER_CODE[(int)(bits & 3)]; bits >>= 2; }
    out[half] = 0;
}

/* Display + Debug as printed by `ska nk` (merge_ska_array.rs:649-698; lib.rs:808-827) */
char *ora_array_nk(const ora_array *a, int full_info, size_t *len)
{
    sbuf b = { 0 };
    sb_printf(&b, "ska_version=%s\nk=%d\nk_bits=%d\nrc=%s\nk-mers=%zu\nsamples=%zu\n", a->version, a->k, a->k_bits,
              a->rc ? "true" : "false", a->nk, a->ns);
    sb_printf(&b, "sample_names=[");
    for (size_t s = 0; s < a->ns; s++) { if (s) sb_put(&b, ", ", 2); sb_rust_debug_str(&b, a->names[s]); }
    sb_printf(&b, "]\nsample_kmers=[");
    for (size_t s = 0; s < a->ns; s++) {           /* n_sample_kmers :554-559 */
        long c = 0;
        for (size_t r = 0; r < a->nrows; r++) if (a->var[r * a->ns + s] != '-') c++;
        sb_printf(&b, "%s%ld", s ? ", " : "", c);
    }
    sb_printf(&b, "]\n");
    sb_put(&b, "\n", 1);                            /* println! */
    if (full_info) {
        int half = (a->k - 1) / 2;
        char up[40], lo[40];
        for (size_t r = 0; r < a->nk && r < a->nrows; r++) {
            u128 key = ((u128)a->keys[r].hi << 64) | a->keys[r].lo;
            u128 lower_mask = (((u128)1) << (half * 2)) - 1;
            decode_arm((key >> (half * 2)) & lower_mask, half, up);
            decode_arm(key & lower_mask, half, lo);
            sb_printf(&b, "%s\t%s\t", up, lo);
            for (size_t s = 0; s < a->ns; s++) {
                char c = a->var[r * a->ns + s] == 0 ? '-' : (char)a->var[r * a->ns + s];
                if (s) sb_put(&b, ",", 1);
                sb_put(&b, &c, 1);
            }
            sb_put(&b, "\n", 1);
        }
        sb_put(&b, "\n", 1);                        /* println! */
    }
    *len = b.n;
    return b.p;
}

/* save / load (merge_ska_array.rs:191-204) */
int ora_array_save(const ora_array *a, const char *path)
{
    size_t clen, flen;
    uint8_t *cbor = ora_skf_encode(a, &clen);
    uint8_t *frame = ora_snappy_frame_encode(cbor, clen, &flen);
    free(cbor);
    FILE *f = fopen(path, "wb");
    if (!f) { ora_set_error("cannot create %s", path); free(frame); return -1; }
    size_t w = fwrite(frame, 1, flen, f);
    fclose(f); free(frame);
    if (w != flen) { ora_set_error("short write %s", path); return -1; }
    return 0;
}
ora_array *ora_array_load(const char *path, int want_bits)
{
    FILE *f = fopen(path, "rb");
    if (!f) { ora_set_error("cannot open %s", path); return NULL; }
    fseek(f, 0, SEEK_END); long sz = ftell(f); fseek(f, 0, SEEK_SET);
    uint8_t *buf = (uint8_t *)malloc((size_t)sz + 1);
    if (fread(buf, 1, (size_t)sz, f) != (size_t)sz) { fclose(f); free(buf); ora_set_error("read error %s", path); return NULL; }
    fclose(f);
    size_t clen; uint8_t *cbor = ora_snappy_frame_decode(buf, (size_t)sz, &clen);
    free(buf);
    if (!cbor) return NULL;
    ora_array *a = ora_skf_decode(cbor, clen);
    free(cbor);
    if (!a) return NULL;
    /* serde into Vec<u64> fails when any key needs more than 64 bits (lib.rs:635-661 falls through to u128) */
    if (want_bits == 64) {
        for (size_t i = 0; i < a->nk; i++) if (a->keys[i].hi) { ora_set_error("key does not fit u64"); ora_array_free(a); return NULL; }
    }
    return a;
}

/* io_utils.rs:31-46: ^.+/(.+)\.(?i:fa|fasta|fastq|fastq\.gz)$ or ^(.+)\.(?i:...)$ else whole string */
static int ends_with_ci(const char *s, size_t n, const char *suf)
{
    size_t m = strlen(suf);
    if (n < m) return 0;
    for (size_t i = 0; i < m; i++) { char c = s[n - m + i]; if (c >= 'A' && c <= 'Z') c = (char)(c + 32); if (c != suf[i]) return 0; }
    return 1;
}
char *ora_sample_name(const char *path)
{
    static const char *exts[] = { ".fa", ".fasta", ".fastq", ".fastq.gz" };   /* shortest first: (.+) is greedy */
    size_t n = strlen(path);
    /* re_path: ^.+/(.+)\.ext$ -- last '/' with >=1 char before it and >=1 stem char after it */
    for (int e = 0; e < 4; e++) {
        size_t m = strlen(exts[e]);
        if (!ends_with_ci(path, n, exts[e])) continue;
        size_t stem_end = n - m;
        for (size_t i = stem_end; i-- > 1;)
            if (path[i] == '/' && i + 1 < stem_end) {
                size_t l = stem_end - (i + 1);
                char *r = (char *)malloc(l + 1); memcpy(r, path + i + 1, l); r[l] = 0; return r;
            }
    }
    /* re_name: ^(.+)\.ext$ */
    for (int e = 0; e < 4; e++) {
        size_t m = strlen(exts[e]);
        if (!ends_with_ci(path, n, exts[e]) || n - m < 1) continue;
        char *r = (char *)malloc(n - m + 1); memcpy(r, path, n - m); r[n - m] = 0; return r;
    }
    return strdup(path);
}
