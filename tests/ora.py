"""ctypes binding of the CPU oracle (oracle/libska_oracle.so).

Test infrastructure only: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg -- never by the product package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
_LIB = os.path.join(ORACLE_DIR, "libska_oracle.so")

QUAL_NOFILTER, QUAL_MIDDLE, QUAL_STRICT = 0, 1, 2
FILTER_NONE, FILTER_NO_CONST, FILTER_NO_AMBIG, FILTER_NO_AMBIG_OR_CONST = 0, 1, 2, 3
F_IS_RC, F_PALIN, F_MIDQ_OK = 1, 2, 4

KEY_DT = np.dtype([("lo", "<u8"), ("hi", "<u8")])
DIST_DT = np.dtype([("distance", "<f8"), ("mismatch_prop", "<f8"), ("match_count", "<u8"), ("mismatch_count", "<u8")])


class Qual(C.Structure):
    _fields_ = [("min_count", C.c_uint16), ("min_qual", C.c_uint8), ("qual_filter", C.c_int)]


class Timers(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("read_parse", "dict", "append", "merge", "to_array", "filter", "fasta")]


def build_lib():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR])


def _load():
    try:
        build_lib()                      # no-op when up to date; rebuilds a stale checker after a source change
    except Exception:
        if not os.path.exists(_LIB):
            raise
    lib = C.CDLL(_LIB)
    vp, sz, i, d, cp = C.c_void_p, C.c_size_t, C.c_int, C.c_double, C.c_char_p
    lib.ora_last_error.restype = cp
    lib.ora_free.argtypes = [vp]
    lib.ora_extract_record.restype = sz
    lib.ora_extract_record.argtypes = [vp, sz, vp, i, i, i, i, i, vp, vp, vp, vp, sz]
    lib.ora_dict_new.restype = vp
    lib.ora_dict_new.argtypes = [i, i, C.POINTER(Qual)]
    lib.ora_dict_add_record.argtypes = [vp, vp, sz, vp, i]
    lib.ora_dict_from_files.restype = vp
    lib.ora_dict_from_files.argtypes = [i, i, cp, cp, C.POINTER(Qual), d]
    lib.ora_dict_size.restype = sz
    lib.ora_dict_size.argtypes = [vp]
    lib.ora_dict_key_bits.argtypes = [vp]
    lib.ora_dict_export_sorted.argtypes = [vp, vp, vp]
    lib.ora_dict_free.argtypes = [vp]
    lib.ora_build_and_merge.restype = vp
    lib.ora_build_and_merge.argtypes = [C.POINTER(cp), C.POINTER(cp), C.POINTER(cp), i, i, i, C.POINTER(Qual), i, d]
    lib.ora_array_from_dicts.restype = vp
    lib.ora_array_from_dicts.argtypes = [C.POINTER(vp), C.POINTER(cp), i]
    lib.ora_array_load.restype = vp
    lib.ora_array_load.argtypes = [cp, i]
    lib.ora_array_save.argtypes = [vp, cp]
    lib.ora_array_free.argtypes = [vp]
    for f in ("k", "rc", "k_bits"):
        getattr(lib, "ora_array_" + f).argtypes = [vp]
    for f in ("nrows", "nkmers", "nsamples"):
        getattr(lib, "ora_array_" + f).argtypes = [vp]
        getattr(lib, "ora_array_" + f).restype = sz
    lib.ora_array_name.restype = cp
    lib.ora_array_name.argtypes = [vp, sz]
    lib.ora_array_version.restype = cp
    lib.ora_array_version.argtypes = [vp]
    lib.ora_array_export.argtypes = [vp, vp, vp, vp]
    lib.ora_array_sort_rows.argtypes = [vp]
    lib.ora_array_filter.restype = C.c_int32
    lib.ora_array_filter.argtypes = [vp, sz, i, i, i, i, i]
    lib.ora_apply_filters.restype = C.c_int32
    lib.ora_apply_filters.argtypes = [vp, d, i, i, i, i]
    lib.ora_array_fasta.restype = vp
    lib.ora_array_fasta.argtypes = [vp, C.POINTER(sz)]
    lib.ora_array_nk.restype = vp
    lib.ora_array_nk.argtypes = [vp, i, C.POINTER(sz)]
    lib.ora_array_distance.argtypes = [vp, d, i, vp]
    lib.ora_distance_tsv.restype = vp
    lib.ora_distance_tsv.argtypes = [vp, d, i, C.POINTER(sz)]
    lib.ora_align_fasta.restype = vp
    lib.ora_align_fasta.argtypes = [vp, i, i, i, d, i, C.POINTER(sz)]
    lib.ora_array_merge.restype = vp
    lib.ora_array_merge.argtypes = [C.POINTER(vp), i]
    lib.ora_array_delete_samples.argtypes = [vp, C.POINTER(cp), i]
    lib.ora_array_weed.argtypes = [vp, vp, sz, i]
    lib.ora_weed.argtypes = [vp, cp, i, d, i, i, i, i]
    lib.ora_ref_new.restype = vp
    lib.ora_ref_new.argtypes = [i, cp, i, i, i]
    lib.ora_ref_map.argtypes = [vp, vp]
    lib.ora_ref_write_aln.restype = vp
    lib.ora_ref_write_aln.argtypes = [vp, C.POINTER(sz)]
    lib.ora_ref_write_vcf.restype = vp
    lib.ora_ref_write_vcf.argtypes = [vp, C.POINTER(sz)]
    lib.ora_ref_free.argtypes = [vp]
    lib.ora_cov_histogram.argtypes = [cp, cp, i, i, vp]
    lib.ora_cov_fit.argtypes = [vp, sz, C.POINTER(d), C.POINTER(d), C.POINTER(sz)]
    lib.ora_cov.restype = vp
    lib.ora_cov.argtypes = [cp, cp, i, i, C.POINTER(sz), C.POINTER(sz)]
    lib.ora_timers_get.argtypes = [C.POINTER(Timers), i]
    lib.ora_sample_name.restype = vp
    lib.ora_sample_name.argtypes = [cp]
    lib.ora_snappy_frame_decode.restype = vp
    lib.ora_snappy_frame_decode.argtypes = [cp, sz, C.POINTER(sz)]
    return lib


lib = _load()


class OracleError(RuntimeError):
    pass


def skf_cbor(path):
    """the CBOR document inside a .skf: its snappy frame taken off (chunk CRCs checked) -- what ciborium reads (merge_ska_array.rs:199-203)"""
    raw = open(path, "rb").read()
    n = C.c_size_t()
    p = lib.ora_snappy_frame_decode(raw, len(raw), C.byref(n))
    if not p:
        raise OracleError(lib.ora_last_error().decode())
    return _take(p, n.value)


def _err():
    return OracleError(lib.ora_last_error().decode())


def _take(ptr, n):
    s = C.string_at(ptr, n)
    lib.ora_free(ptr)
    return s


def _np_ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def qual(min_count=5, min_qual=20, qual_filter=QUAL_STRICT):
    return Qual(min_count, min_qual, qual_filter)


def extract_record(seq, k, rc=True, qual_bytes=None, min_qual=20, qual_filter=QUAL_STRICT, is_reads=False):
    """(keys[KEY_DT], mid[u8], flags[u8], hashes[u64]) for every window the reference iterator yields."""
    seq = np.frombuffer(bytes(seq), dtype=np.uint8)
    q = np.frombuffer(bytes(qual_bytes), dtype=np.uint8) if qual_bytes is not None else None
    cap = max(len(seq), 1)
    keys = np.zeros(cap, KEY_DT)
    mid = np.zeros(cap, np.uint8)
    flags = np.zeros(cap, np.uint8)
    hashes = np.zeros(cap, np.uint64)
    n = lib.ora_extract_record(_np_ptr(seq), len(seq), _np_ptr(q), k, int(rc), min_qual, qual_filter, int(is_reads),
                               _np_ptr(keys), _np_ptr(mid), _np_ptr(flags), _np_ptr(hashes), cap)
    return keys[:n], mid[:n], flags[:n], hashes[:n]


class Dict:
    def __init__(self, handle):
        if not handle:
            raise _err()
        self.h = handle

    @classmethod
    def new(cls, k, rc=True, q=None):
        q = q or qual()
        return cls(lib.ora_dict_new(k, int(rc), C.byref(q)))

    @classmethod
    def from_files(cls, k, file1, file2=None, rc=True, q=None, proportion_reads=0.0):
        q = q or qual()
        return cls(lib.ora_dict_from_files(k, int(rc), file1.encode(), file2.encode() if file2 else None, C.byref(q),
                                           proportion_reads))

    def add_record(self, seq, qual_bytes=None, is_reads=False):
        seq = np.frombuffer(bytes(seq), dtype=np.uint8)
        q = np.frombuffer(bytes(qual_bytes), dtype=np.uint8) if qual_bytes is not None else None
        lib.ora_dict_add_record(self.h, _np_ptr(seq), len(seq), _np_ptr(q), int(is_reads))

    def __len__(self):
        return lib.ora_dict_size(self.h)

    @property
    def key_bits(self):
        return lib.ora_dict_key_bits(self.h)

    def export(self):
        n = len(self)
        keys = np.zeros(n, KEY_DT)
        bases = np.zeros(n, np.uint8)
        lib.ora_dict_export_sorted(self.h, _np_ptr(keys), _np_ptr(bases))
        return keys, bases

    def __del__(self):
        if getattr(self, "h", None):
            lib.ora_dict_free(self.h)
            self.h = None


class Array:
    def __init__(self, handle):
        if not handle:
            raise _err()
        self.h = handle

    @classmethod
    def build(cls, inputs, k=31, rc=True, q=None, threads=1, proportion_reads=0.0):
        """inputs: list of (name, file1, file2|None) -- build_and_merge + MergeSkaArray::new"""
        q = q or qual()
        n = len(inputs)
        names = (C.c_char_p * n)(*[x[0].encode() for x in inputs])
        f1 = (C.c_char_p * n)(*[x[1].encode() for x in inputs])
        f2 = (C.c_char_p * n)(*[(x[2].encode() if x[2] else None) for x in inputs])
        return cls(lib.ora_build_and_merge(names, f1, f2, n, k, int(rc), C.byref(q), threads, proportion_reads))

    @classmethod
    def from_dicts(cls, dicts, names):
        n = len(dicts)
        hs = (C.c_void_p * n)(*[d.h for d in dicts])
        nm = (C.c_char_p * n)(*[x.encode() for x in names])
        return cls(lib.ora_array_from_dicts(hs, nm, n))

    @classmethod
    def load(cls, path, want_bits=0):
        return cls(lib.ora_array_load(path.encode(), want_bits))

    def save(self, path):
        if lib.ora_array_save(self.h, path.encode()):
            raise _err()

    k = property(lambda s: lib.ora_array_k(s.h))
    rc = property(lambda s: bool(lib.ora_array_rc(s.h)))
    k_bits = property(lambda s: lib.ora_array_k_bits(s.h))
    nrows = property(lambda s: lib.ora_array_nrows(s.h))
    nkmers = property(lambda s: lib.ora_array_nkmers(s.h))
    nsamples = property(lambda s: lib.ora_array_nsamples(s.h))
    version = property(lambda s: lib.ora_array_version(s.h).decode())

    @property
    def names(self):
        return [lib.ora_array_name(self.h, i).decode() for i in range(self.nsamples)]

    def export(self):
        keys = np.zeros(self.nkmers, KEY_DT)
        var = np.zeros((self.nrows, self.nsamples), np.uint8)
        counts = np.zeros(self.nrows, np.uint64)
        lib.ora_array_export(self.h, _np_ptr(keys), _np_ptr(var), _np_ptr(counts))
        return keys, var, counts

    def sort_rows(self):
        lib.ora_array_sort_rows(self.h)

    def filter(self, min_count, filter_ambig_as_missing=False, filter_type=FILTER_NO_CONST, mask_ambig=False,
               ignore_const_gaps=False, update_kmers=True):
        return lib.ora_array_filter(self.h, min_count, int(filter_ambig_as_missing), filter_type, int(mask_ambig),
                                    int(ignore_const_gaps), int(update_kmers))

    def apply_filters(self, min_freq, filter_ambig_as_missing=False, filter_type=FILTER_NO_CONST, ambig_mask=False,
                      ignore_const_gaps=False):
        return lib.ora_apply_filters(self.h, min_freq, int(filter_ambig_as_missing), filter_type, int(ambig_mask),
                                     int(ignore_const_gaps))

    def fasta(self):
        n = C.c_size_t()
        return _take(lib.ora_array_fasta(self.h, C.byref(n)), n.value)

    def nk(self, full_info=False):
        n = C.c_size_t()
        return _take(lib.ora_array_nk(self.h, int(full_info), C.byref(n)), n.value)

    def distance(self, constant=0.0, filt_ambig=True):
        s = self.nsamples
        out = np.zeros(s * (s - 1) // 2, DIST_DT)
        lib.ora_array_distance(self.h, constant, int(filt_ambig), _np_ptr(out))
        return out

    def distance_tsv(self, min_freq=0.0, filt_ambig=True):
        n = C.c_size_t()
        return _take(lib.ora_distance_tsv(self.h, min_freq, int(filt_ambig), C.byref(n)), n.value)

    def align(self, filter_type=FILTER_NO_CONST, mask_ambig=False, ignore_const_gaps=False, min_freq=0.9,
              filter_ambig_as_missing=False):
        n = C.c_size_t()
        return _take(lib.ora_align_fasta(self.h, filter_type, int(mask_ambig), int(ignore_const_gaps), min_freq,
                                         int(filter_ambig_as_missing), C.byref(n)), n.value)

    # ---- skf life-cycle (ska merge / delete / weed) ----
    @classmethod
    def merge(cls, arrays):
        hs = (C.c_void_p * len(arrays))(*[a.h for a in arrays])
        return cls(lib.ora_array_merge(hs, len(arrays)))

    def delete_samples(self, names):
        nm = (C.c_char_p * len(names))(*[x.encode() for x in names])
        if lib.ora_array_delete_samples(self.h, nm, len(names)):
            raise _err()

    def weed_keys(self, keys, reverse=False):
        keys = np.ascontiguousarray(keys, KEY_DT)
        if lib.ora_array_weed(self.h, _np_ptr(keys), len(keys), int(reverse)):
            raise _err()

    def weed(self, weed_fasta=None, reverse=False, min_freq=0.9, filter_ambig_as_missing=False, filter_type=FILTER_NONE,
             ambig_mask=False, ignore_const_gaps=False):
        if lib.ora_weed(self.h, weed_fasta.encode() if weed_fasta else None, int(reverse), min_freq,
                        int(filter_ambig_as_missing), filter_type, int(ambig_mask), int(ignore_const_gaps)):
            raise _err()

    # ---- ska map (generic_modes.rs:56-84) ----
    def map(self, reference, fmt="aln", ambig_mask=False, repeat_mask=False):
        """RefSka::new(k, reference, rc, ambig_mask, repeat_mask) + map + write_aln | write_vcf -> text"""
        r = lib.ora_ref_new(self.k, reference.encode(), int(self.rc), int(ambig_mask), int(repeat_mask))
        if not r:
            raise _err()
        try:
            if lib.ora_ref_map(r, self.h):
                raise _err()
            n = C.c_size_t()
            p = (lib.ora_ref_write_vcf if fmt == "vcf" else lib.ora_ref_write_aln)(r, C.byref(n))
            if not p:
                raise _err()
            return _take(p, n.value)
        finally:
            lib.ora_ref_free(r)

    def __del__(self):
        if getattr(self, "h", None):
            lib.ora_array_free(self.h)
            self.h = None


def cov_histogram(fq1, fq2, k=31, rc=True):
    h = np.zeros(1000, np.uint32)
    if lib.ora_cov_histogram(fq1.encode(), fq2.encode(), k, int(rc), _np_ptr(h)):
        raise _err()
    return h


def cov_fit(counts):
    """coverage.rs fit_histogram on an already truncated histogram -> (w0, c, cutoff)"""
    c = np.ascontiguousarray(counts, np.float64)
    w0, cc, cut = C.c_double(), C.c_double(), C.c_size_t()
    if lib.ora_cov_fit(_np_ptr(c), len(c), C.byref(w0), C.byref(cc), C.byref(cut)):
        raise _err()
    return w0.value, cc.value, cut.value


def cov(fq1, fq2, k=31, rc=True):
    """`ska cov`: (plot_hist text, cutoff)"""
    n, cut = C.c_size_t(), C.c_size_t()
    p = lib.ora_cov(fq1.encode(), fq2.encode(), k, int(rc), C.byref(n), C.byref(cut))
    if not p:
        raise _err()
    return _take(p, n.value), cut.value


def sample_name(path):
    p = lib.ora_sample_name(path.encode())
    s = C.string_at(p)
    lib.ora_free(p)
    return s.decode()


def timers(reset=False):
    t = Timers()
    lib.ora_timers_get(C.byref(t), int(reset))
    return {n: getattr(t, n) for n, _ in Timers._fields_}


# --- helpers shared by the golden tests (tests/common/mod.rs:165-212) ---
def var_hash(aln: bytes):
    lines = aln.decode().splitlines()
    seqs = [lines[i] for i in range(1, len(lines), 2)]
    return {tuple(s[c] for s in seqs) for c in range(len(seqs[0]))} if seqs and seqs[0] else set()


def aln_length(aln: bytes):
    lines = aln.decode().splitlines()
    return [len(lines[i]) for i in range(1, len(lines), 2)]
