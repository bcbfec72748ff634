/*
 * ora_kmer_impl.h -- ORACLE (test infrastructure).  "Generic" part of the
 * restatement, instantiated twice (KT = uint64_t for k<=31, unsigned __int128
 * for k<=63) exactly as the reference instantiates its UInt trait
 * (bit_encoding.rs:88-302, lib.rs:592-622).
 *
 * Included by ora_core.c with:  #define KT <type>   #define SFX(name) name##_64|_128
 */

/* ---- UInt::rev_comp (bit_encoding.rs:182-195 u64, :241-261 u128) ------- */
static inline KT SFX(rev_comp)(KT x, int k_size)
{
#if KT_BITS == 64
    x = ((x >> 2) & 0x3333333333333333ULL) | ((x & 0x3333333333333333ULL) << 2);
    x = ((x >> 4) & 0x0F0F0F0F0F0F0F0FULL) | ((x & 0x0F0F0F0F0F0F0F0FULL) << 4);
    x = ((x >> 8) & 0x00FF00FF00FF00FFULL) | ((x & 0x00FF00FF00FF00FFULL) << 8);
    x = ((x >> 16) & 0x0000FFFF0000FFFFULL) | ((x & 0x0000FFFF0000FFFFULL) << 16);
    x = ((x >> 32) & 0x00000000FFFFFFFFULL) | ((x & 0x00000000FFFFFFFFULL) << 32);
    x ^= 0xAAAAAAAAAAAAAAAAULL;
    return x >> (2 * (32 - k_size));
#else
    const KT m2 = ((KT)0x3333333333333333ULL << 64) | 0x3333333333333333ULL;
    const KT m4 = ((KT)0x0F0F0F0F0F0F0F0FULL << 64) | 0x0F0F0F0F0F0F0F0FULL;
    const KT m8 = ((KT)0x00FF00FF00FF00FFULL << 64) | 0x00FF00FF00FF00FFULL;
    const KT m16 = ((KT)0x0000FFFF0000FFFFULL << 64) | 0x0000FFFF0000FFFFULL;
    const KT m32 = ((KT)0x00000000FFFFFFFFULL << 64) | 0x00000000FFFFFFFFULL;
    const KT m64 = (KT)0xFFFFFFFFFFFFFFFFULL;
    const KT ma = ((KT)0xAAAAAAAAAAAAAAAAULL << 64) | 0xAAAAAAAAAAAAAAAAULL;
    x = ((x >> 2) & m2) | ((x & m2) << 2);
    x = ((x >> 4) & m4) | ((x & m4) << 4);
    x = ((x >> 8) & m8) | ((x & m8) << 8);
    x = ((x >> 16) & m16) | ((x & m16) << 16);
    x = ((x >> 32) & m32) | ((x & m32) << 32);
    x = ((x >> 64) & m64) | ((x & m64) << 64);
    x ^= ma;
    return x >> (2 * (64 - k_size));
#endif
}

/* ---- SplitKmer (split_kmer.rs:26-61) ----------------------------------- */
typedef struct {
    int k;
    KT upper_mask, lower_mask;
    const uint8_t *seq;
    size_t seq_len;
    const uint8_t *qual;      /* NULL == None */
    int qual_filter;
    uint8_t min_qual;
    size_t index;
    KT upper, lower;
    uint8_t middle_base;
    int rc;
    KT rc_upper, rc_lower;
    uint8_t rc_middle_base;
    int has_hash;             /* hash_gen.is_some() */
    ora_nthash hash_gen;
} SFX(splitkmer);

/* split_kmer.rs:78-140 `build`: returns 0 for None */
static int SFX(sk_build)(const uint8_t *seq, size_t seq_len, const uint8_t *qual, int k, size_t *idx,
                         int qual_filter, uint8_t min_qual, int is_reads, int rc,
                         KT *upper_o, KT *lower_o, uint8_t *mid_o, ora_nthash *hash_o)
{
    if (*idx + (size_t)k >= seq_len) return 0;                       /* :89 */
    KT upper = 0, lower = 0;
    uint8_t middle_base = 0;
    const int middle_idx = (k + 1) / 2 - 1;                          /* k.div_ceil(2)-1 :95 */
    int i = 0;
    while (i < k) {
        size_t p = (size_t)i + *idx;
        if (ora_valid_base(seq[p]) &&
            (qual_filter != ORA_QUAL_STRICT || ora_valid_qual(p, qual, min_qual))) {   /* :98-101 */
            uint8_t next_base = ora_encode_base(seq[p]);
            if (i > middle_idx) {
                lower <<= 2;
                lower |= (KT)next_base;
            } else if (i < middle_idx) {
                upper <<= 2;
                upper |= (KT)next_base << (middle_idx * 2);
            } else {
                middle_base = next_base;
            }
            i++;
        } else {
            *idx += (size_t)i + 1;                                   /* :120 */
            if (*idx + (size_t)k >= seq_len) return 0;
            upper = 0; lower = 0; middle_base = 0; i = 0;
        }
    }
    if (is_reads) ora_nthash_new(hash_o, seq + *idx, k, rc);         /* :132-136 */
    *idx += (size_t)k - 1;                                           /* :138 */
    *upper_o = upper; *lower_o = lower; *mid_o = middle_base;
    return 1;
}

/* split_kmer.rs:149-153 */
static inline void SFX(sk_update_rc)(SFX(splitkmer) *s)
{
    s->rc_upper = SFX(rev_comp)(s->lower, s->k - 1) & s->upper_mask;
    s->rc_middle_base = s->middle_base ^ 2;
    s->rc_lower = SFX(rev_comp)(s->upper, s->k - 1) & s->lower_mask;
}

/* split_kmer.rs:226-275 `new`: returns 0 for None */
static int SFX(sk_new)(SFX(splitkmer) *s, const uint8_t *seq, size_t seq_len, const uint8_t *qual, int k,
                       int rc, uint8_t min_qual, int qual_filter, int is_reads)
{
    size_t index = 0;
    KT upper, lower; uint8_t mid;
    memset(s, 0, sizeof *s);
    if (!SFX(sk_build)(seq, seq_len, qual, k, &index, qual_filter, min_qual, is_reads, rc,
                       &upper, &lower, &mid, &s->hash_gen))
        return 0;
    const int half = (k - 1) / 2;                                    /* generate_masks, bit_encoding.rs:208-213 */
    s->lower_mask = (((KT)1) << (half * 2)) - 1;
    s->upper_mask = s->lower_mask << (half * 2);
    s->k = k; s->seq = seq; s->seq_len = seq_len; s->qual = qual; s->qual_filter = qual_filter;
    s->min_qual = min_qual; s->upper = upper; s->lower = lower; s->middle_base = mid; s->rc = rc;
    s->index = index; s->has_hash = is_reads;
    if (rc) SFX(sk_update_rc)(s);
    return 1;
}

/* split_kmer.rs:159-217 `roll_fwd` */
static int SFX(sk_roll_fwd)(SFX(splitkmer) *s)
{
    s->index += 1;
    if (s->index >= s->seq_len) return 0;
    uint8_t base = s->seq[s->index];
    if (!ora_valid_base(base) ||
        (s->qual_filter == ORA_QUAL_STRICT && !ora_valid_qual(s->index, s->qual, s->min_qual))) {
        KT upper, lower; uint8_t mid;
        if (SFX(sk_build)(s->seq, s->seq_len, s->qual, s->k, &s->index, s->qual_filter, s->min_qual,
                          s->has_hash, s->rc, &upper, &lower, &mid, &s->hash_gen)) {
            s->upper = upper; s->lower = lower; s->middle_base = mid;
            if (s->rc) SFX(sk_update_rc)(s);
            return 1;
        }
        return 0;
    }
    const int half_k = (s->k - 1) / 2;
    uint8_t new_base = ora_encode_base(base);
    if (s->has_hash) {
        uint8_t old_base = (uint8_t)(s->upper >> ((s->k - 2) * 2));  /* :190 */
        ora_nthash_roll(&s->hash_gen, old_base, new_base);
    }
    s->upper = ((s->upper << 2) | ((KT)s->middle_base << (half_k * 2))) & s->upper_mask;
    s->middle_base = (uint8_t)(s->lower >> (2 * (half_k - 1)));
    s->lower = ((s->lower << 2) | (KT)new_base) & s->lower_mask;
    if (s->rc) {
        s->rc_lower = ((s->rc_lower >> 2) | ((KT)s->rc_middle_base << (2 * (half_k - 1)))) & s->lower_mask;
        s->rc_middle_base = s->middle_base ^ 2;
        s->rc_upper = ((s->rc_upper >> 2) | ((KT)(new_base ^ 2) << (2 * ((half_k * 2) - 1)))) & s->upper_mask;
    }
    return 1;
}

/* split_kmer.rs:144-146 */
static inline int SFX(sk_self_palindrome)(const SFX(splitkmer) *s)
{
    return s->rc && s->upper == s->rc_upper && s->lower == s->rc_lower;
}

/* split_kmer.rs:281-295 */
static inline KT SFX(sk_curr)(const SFX(splitkmer) *s, uint8_t *base, int *is_rc)
{
    KT split_kmer = s->upper | s->lower;
    if (s->rc) {
        KT rc_split = s->rc_upper | s->rc_lower;
        if (split_kmer > rc_split) { *base = s->rc_middle_base; *is_rc = 1; return rc_split; }
    }
    *base = s->middle_base; *is_rc = 0;
    return split_kmer;
}

/* split_kmer.rs:322-339 */
static inline int SFX(sk_middle_base_qual)(const SFX(splitkmer) *s)
{
    if (!s->qual) return 1;
    if (s->qual_filter == ORA_QUAL_MIDDLE || s->qual_filter == ORA_QUAL_STRICT) {
        const int middle_idx = (s->k + 1) / 2 - 1;
        return ora_valid_qual(s->index - (size_t)middle_idx, s->qual, s->min_qual);
    }
    return 1;
}

/* ---- SkaDict: HashMap<IntT,u8> (ska_dict.rs:56-113) --------------------
 * open addressing, value 0 == empty slot (a stored base is never 0). */
typedef struct {
    KT *keys; uint8_t *vals; size_t cap, n;
} SFX(kmap);

static inline uint64_t SFX(khash)(KT k)
{
    uint64_t x = (uint64_t)k;
#if KT_BITS == 128
    x ^= (uint64_t)(k >> 64) * 0x9E3779B97F4A7C15ULL;
#endif
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
    return x;
}

static void SFX(kmap_init)(SFX(kmap) *m, size_t cap)
{
    m->cap = cap; m->n = 0;
    m->keys = (KT *)malloc(cap * sizeof(KT));
    m->vals = (uint8_t *)calloc(cap, 1);
}
static void SFX(kmap_free)(SFX(kmap) *m) { free(m->keys); free(m->vals); m->keys = NULL; m->vals = NULL; m->cap = m->n = 0; }

static uint8_t *SFX(kmap_slot)(SFX(kmap) *m, KT key, int *found);
static void SFX(kmap_grow)(SFX(kmap) *m)
{
    SFX(kmap) nm; SFX(kmap_init)(&nm, m->cap * 2);
    for (size_t i = 0; i < m->cap; i++) if (m->vals[i]) {
        int f; uint8_t *v = SFX(kmap_slot)(&nm, m->keys[i], &f); *v = m->vals[i]; nm.n++;
    }
    free(m->keys); free(m->vals); *m = nm;
}
/* entry(): returns pointer to the value slot; *found says whether it was occupied.  The caller must
 * store a non-zero value into a fresh slot and bump n. */
static uint8_t *SFX(kmap_slot)(SFX(kmap) *m, KT key, int *found)
{
    size_t mask = m->cap - 1, i = (size_t)SFX(khash)(key) & mask;
    for (;;) {
        if (!m->vals[i]) { m->keys[i] = key; *found = 0; return &m->vals[i]; }
        if (m->keys[i] == key) { *found = 1; return &m->vals[i]; }
        i = (i + 1) & mask;
    }
}
static inline uint8_t *SFX(kmap_entry)(SFX(kmap) *m, KT key, int *found)
{
    if ((m->n + 1) * 10 > m->cap * 7) SFX(kmap_grow)(m);
    return SFX(kmap_slot)(m, key, found);
}

/* ska_dict.rs:76-81 add_to_dict */
static inline void SFX(add_to_dict)(SFX(kmap) *m, KT kmer, uint8_t base)
{
    int found; uint8_t *b = SFX(kmap_entry)(m, kmer, &found);
    if (found) *b = ora_iupac_update(base, *b);          /* IUPAC[base*256 + *b], bit_encoding.rs:388 */
    else { *b = (uint8_t)ORA_LETTER_CODE[base]; m->n++; } /* decode_base */
}

/* ska_dict.rs:85-113 add_palindrome_to_dict */
static inline void SFX(add_palindrome_to_dict)(SFX(kmap) *m, KT kmer, uint8_t base)
{
    int found; uint8_t *b = SFX(kmap_entry)(m, kmer, &found);
    if (found) {
        switch (*b) {
        case 'W': *b = (base == 0 || base == 2) ? 'W' : 'N'; break;
        case 'S': *b = (base == 0 || base == 2) ? 'N' : 'S'; break;
        case 'N': break;
        default:
            fprintf(stderr, "Palindrome middle base not W/S: %c\n", *b); abort();
        }
    } else {
        *b = (base == 0 || base == 2) ? 'W' : 'S';
        m->n++;
    }
}

/* one record: body of the `while let Some(record)` loop, ska_dict.rs:143-177 */
static void SFX(dict_add_record)(SFX(kmap) *m, ora_kmer_filter *filt, const uint8_t *seq, size_t len,
                                 const uint8_t *qual, int k, int rc, const ora_qual *q, int is_reads)
{
    SFX(splitkmer) it;
    if (!SFX(sk_new)(&it, seq, len, qual, k, rc, q->min_qual, q->qual_filter, is_reads)) return;
    int more = 1;
    while (more) {
        /* `&&` short-circuit: the count filter is only touched when the middle base passes (:155-157) */
        if (!is_reads || (SFX(sk_middle_base_qual)(&it) && ora_filter_pass(filt, ora_nthash_curr(&it.hash_gen)))) {
            uint8_t base; int is_rc;
            KT kmer = SFX(sk_curr)(&it, &base, &is_rc);
            if (SFX(sk_self_palindrome)(&it)) SFX(add_palindrome_to_dict)(m, kmer, base);
            else SFX(add_to_dict)(m, kmer, base);
        }
        more = SFX(sk_roll_fwd)(&it);
    }
}

/* known-answer enumeration for ora_extract_record */
static size_t SFX(extract_record)(const uint8_t *seq, size_t len, const uint8_t *qual, int k, int rc,
                                  int min_qual, int qual_filter, int is_reads, ora_key *keys, uint8_t *mid,
                                  uint8_t *flags, uint64_t *hashes, size_t *pos, size_t cap)
{
    SFX(splitkmer) it; size_t n = 0;
    if (!SFX(sk_new)(&it, seq, len, qual, k, rc, (uint8_t)min_qual, qual_filter, is_reads)) return 0;
    int more = 1;
    while (more) {
        uint8_t base; int is_rc;
        KT kmer = SFX(sk_curr)(&it, &base, &is_rc);
        if (n < cap) {
            if (keys) {
                keys[n].lo = (uint64_t)kmer;
#if KT_BITS == 128
                keys[n].hi = (uint64_t)(kmer >> 64);
#else
                keys[n].hi = 0;
#endif
            }
            if (mid) mid[n] = base;
            if (flags) flags[n] = (uint8_t)((is_rc ? ORA_F_IS_RC : 0) | (SFX(sk_self_palindrome)(&it) ? ORA_F_PALIN : 0) |
                                            (SFX(sk_middle_base_qual)(&it) ? ORA_F_MIDQ_OK : 0));
            if (hashes) hashes[n] = is_reads ? ora_nthash_curr(&it.hash_gen) : 0;
            if (pos) pos[n] = it.index - (size_t)((k + 1) / 2 - 1);         /* get_middle_pos, split_kmer.rs:322-325 */
        }
        n++;
        more = SFX(sk_roll_fwd)(&it);
    }
    return n;
}

/* ---- MergeSkaDict: HashMap<IntT, Vec<u8>> (merge_ska_dict.rs:28-151) ---
 * key -> row id into an arena of n_samples-byte rows (one "Vec<u8>" per key). */
typedef struct {
    int k, rc; size_t n_samples;
    KT *keys; uint32_t *rowid;  /* rowid 0 == empty; stored id = row+1 */
    size_t cap, n;
    uint8_t **chunks; size_t n_chunks, rows_in_last;  /* arena: chunks of MROWS rows */
} SFX(mdict);
#define ORA_MROWS 65536u

static void SFX(mdict_init)(SFX(mdict) *m, int k, size_t n_samples, int rc)
{
    memset(m, 0, sizeof *m);
    m->k = k; m->rc = rc; m->n_samples = n_samples;
    m->cap = 1024; m->keys = (KT *)malloc(m->cap * sizeof(KT)); m->rowid = (uint32_t *)calloc(m->cap, 4);
}
static void SFX(mdict_free)(SFX(mdict) *m)
{
    for (size_t i = 0; i < m->n_chunks; i++) free(m->chunks[i]);
    free(m->chunks); free(m->keys); free(m->rowid); memset(m, 0, sizeof *m);
}
static inline uint8_t *SFX(mdict_row)(const SFX(mdict) *m, uint32_t id)
{
    uint32_t r = id - 1;
    return m->chunks[r / ORA_MROWS] + (size_t)(r % ORA_MROWS) * m->n_samples;
}
/* vec![0; n_samples] */
static uint32_t SFX(mdict_new_row)(SFX(mdict) *m)
{
    if (m->n_chunks == 0 || m->rows_in_last == ORA_MROWS) {
        m->chunks = (uint8_t **)realloc(m->chunks, (m->n_chunks + 1) * sizeof(uint8_t *));
        m->chunks[m->n_chunks++] = (uint8_t *)calloc((size_t)ORA_MROWS, m->n_samples ? m->n_samples : 1);
        m->rows_in_last = 0;
    }
    uint32_t r = (uint32_t)((m->n_chunks - 1) * ORA_MROWS + m->rows_in_last++);
    return r + 1;
}
/* the merged map hashes with another seed than the per-sample maps (hashbrown gives every map its own RandomState):
 * feeding a linear-probing table in the slot order of a table with the SAME hash degenerates into long probe runs */
static uint32_t *SFX(mdict_slot)(SFX(mdict) *m, KT key, int *found)
{
    size_t mask = m->cap - 1, i = (size_t)((SFX(khash)(key ^ (KT)0x5851F42D4C957F2DULL) * 0x9E3779B97F4A7C15ULL) >> 20) & mask;
    for (;;) {
        if (!m->rowid[i]) { m->keys[i] = key; *found = 0; return &m->rowid[i]; }
        if (m->keys[i] == key) { *found = 1; return &m->rowid[i]; }
        i = (i + 1) & mask;
    }
}
static uint32_t *SFX(mdict_entry)(SFX(mdict) *m, KT key, int *found)
{
    if ((m->n + 1) * 10 > m->cap * 7) {
        size_t ocap = m->cap; KT *ok = m->keys; uint32_t *orid = m->rowid;
        m->cap = ocap * 2; m->keys = (KT *)malloc(m->cap * sizeof(KT)); m->rowid = (uint32_t *)calloc(m->cap, 4);
        for (size_t i = 0; i < ocap; i++) if (orid[i]) { int f; *SFX(mdict_slot)(m, ok[i], &f) = orid[i]; }
        free(ok); free(orid);
    }
    return SFX(mdict_slot)(m, key, found);
}

/* merge_ska_dict.rs:77-109 append (names handled by the caller) */
static void SFX(mdict_append)(SFX(mdict) *m, const SFX(kmap) *other, size_t idx)
{
    for (size_t i = 0; i < other->cap; i++) {
        if (!other->vals[i]) continue;
        int found; uint32_t *rid = SFX(mdict_entry)(m, other->keys[i], &found);
        if (!found) { *rid = SFX(mdict_new_row)(m); m->n++; }
        SFX(mdict_row)(m, *rid)[idx] = other->vals[i];
    }
}

/* merge_ska_dict.rs:119-151 merge (other is consumed) */
static void SFX(mdict_merge)(SFX(mdict) *self, SFX(mdict) *other)
{
    if (other->n == 0) return;
    if (self->n == 0) { SFX(mdict) t = *self; *self = *other; *other = t; return; }
    for (size_t i = 0; i < other->cap; i++) {
        if (!other->rowid[i]) continue;
        const uint8_t *ov = SFX(mdict_row)(other, other->rowid[i]);
        int found; uint32_t *rid = SFX(mdict_entry)(self, other->keys[i], &found);
        if (found) {
            uint8_t *sv = SFX(mdict_row)(self, *rid);
            for (size_t s = 0; s < self->n_samples; s++) sv[s] |= ov[s];
        } else {
            *rid = SFX(mdict_new_row)(self); self->n++;
            memcpy(SFX(mdict_row)(self, *rid), ov, self->n_samples);   /* mem::take(other_vec) */
        }
    }
}
