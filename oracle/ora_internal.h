/* ora_internal.h -- ORACLE (test infrastructure): shared private declarations. */
#ifndef ORA_INTERNAL_H
#define ORA_INTERNAL_H
#include "ska_oracle.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef unsigned __int128 u128;

void ora_set_error(const char *fmt, ...);

/* ---- FASTX (needletail 0.5 behaviour used at ska_dict.rs:131-153,356-366) ---- */
typedef struct { const uint8_t *seq; size_t len; const uint8_t *qual; const uint8_t *id; size_t id_len; } ora_rec;   /* id: header line without '>' */
typedef struct { ora_rec *recs; size_t n; int is_fastq; uint8_t *arena; } ora_fastx;
int  ora_fastx_read(const char *path, ora_fastx *out);   /* 0 ok */
void ora_fastx_free(ora_fastx *f);

/* ---- .skf codec (merge_ska_array.rs:191-204) ---- */
struct ora_array {
    int k, rc, k_bits;
    size_t nk;        /* split_kmers.len() */
    size_t nrows;     /* variants.nrows()  */
    size_t ns;        /* variants.ncols() == names.len() */
    ora_key *keys;
    uint8_t *var;     /* row-major [nrows, ns] */
    uint64_t *counts; /* variant_count, len nrows */
    char **names;
    char *version;
};
struct ora_array *ora_skf_decode(const uint8_t *cbor, size_t len);
uint8_t *ora_skf_encode(const struct ora_array *a, size_t *len);
uint8_t *ora_snappy_frame_decode(const uint8_t *in, size_t len, size_t *out_len);
uint8_t *ora_snappy_frame_encode(const uint8_t *in, size_t len, size_t *out_len);
uint32_t ora_crc32c(const uint8_t *p, size_t n);

double ora_now(void);
/* SplitKmer enumeration of one record with get_middle_pos() of every window (ora_core.c) */
size_t extract_record_pos(const uint8_t *seq, size_t len, int k, int rc, ora_key *keys, uint8_t *flags, size_t *pos, size_t cap);
#endif
