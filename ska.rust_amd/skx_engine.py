"""ctypes binding of the MI355X split-k-mer engine (libskx.so: include/skx.h + include/skx_host.h).

The product path.  It never imports anything from oracle/ and has no CPU
fallback: if libskx.so is missing or no gfx950 device is usable, every call
raises.  The Python surface mirrors the reference's Rust API for the hot path
(SkaDict / build_and_merge / MergeSkaArray / generic_modes), so tests read like
the reference's own.
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libskx.so")

QUAL_NOFILTER, QUAL_MIDDLE, QUAL_STRICT = 0, 1, 2
FILTER_NONE, FILTER_NO_CONST, FILTER_NO_AMBIG, FILTER_NO_AMBIG_OR_CONST = 0, 1, 2, 3
OK, EINVAL, EIO, ENODEV, ENOMEM, EEMPTY, EUNSUP, EFORMAT = range(8)

KEY_DT = np.dtype([("lo", "<u8"), ("hi", "<u8")])
DIST_DT = np.dtype([("distance", "<f8"), ("mismatch_prop", "<f8"), ("match_count", "<u8"), ("mismatch_count", "<u8")])


class Qual(C.Structure):
    _fields_ = [("min_count", C.c_uint16), ("min_qual", C.c_uint8), ("qual_filter", C.c_int32)]


class Stream(C.Structure):
    _fields_ = [("seq", C.c_void_p), ("qual", C.c_void_p), ("len", C.c_uint64)]


class FilterSpec(C.Structure):
    _fields_ = [("min_freq", C.c_double), ("filter_ambig_as_missing", C.c_int32), ("filter_type", C.c_int32), ("mask_ambig", C.c_int32),
                ("ignore_const_gaps", C.c_int32), ("two_stage", C.c_int32)]


class ArrayInfo(C.Structure):
    _fields_ = [("k", C.c_int32), ("rc", C.c_int32), ("k_bits", C.c_int32), ("n_kmers", C.c_uint64),
                ("n_rows", C.c_uint64), ("n_samples", C.c_uint64), ("total_samples", C.c_uint64)]


class Timings(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("hist", "scatter", "dedupe", "key_union", "assemble", "filter", "compact", "distance",
                                                 "append_probe", "append", "pieces_stats", "pieces_rows")]


class Job(C.Structure):
    """skh_job (include/skx_host.h): one build | align | distance job over the GPUs of a node"""
    _fields_ = [("names", C.POINTER(C.c_char_p)), ("file1", C.POINTER(C.c_char_p)), ("file2", C.POINTER(C.c_char_p)), ("n_samples", C.c_int),
                ("k", C.c_int), ("rc", C.c_int), ("qual", Qual), ("threads", C.c_int), ("proportion_reads", C.c_double),
                ("output", C.c_char_p), ("merge_parts", C.c_int), ("min_freq", C.c_double),
                ("filter_type", C.c_int), ("mask_ambig", C.c_int), ("ignore_const_gaps", C.c_int), ("filter_ambig_as_missing", C.c_int),
                ("filt_ambig", C.c_int)]


class EngineError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"[skx {code}] {msg}")
        self.code = code


# every symbol include/skx.h and include/skx_host.h declare
SYMBOLS = """skx_last_error skx_version skx_ctx_create skx_ctx_destroy skx_ctx_sync skx_ctx_stream skx_dictset_build
skx_dictset_build_files skx_read_records skx_dictset_free skx_dictset_nsamples skx_dictset_key_bits skx_dictset_size skx_dictset_export
skx_keyset_union skx_keyset_union_notes skx_keyset_size skx_keyset_device skx_keyset_from_device skx_keyset_merge skx_keyset_free
skx_array_assemble skx_array_assemble_lazy skx_merge skx_build_and_merge skx_array_free skx_array_save skx_array_load skx_array_from_host
skx_array_info skx_array_name skx_array_version skx_array_export skx_array_sample_kmers skx_array_pieces_info skx_array_filter
skx_array_write_fasta skx_array_fasta skx_array_device_matrix skx_array_device_stats skx_array_set_total_samples skx_array_distance skx_free skx_ctx_timings skx_ctx_merge_path
skx_array_merge skx_array_delete_samples skx_array_weed skx_keyset_from_fasta skx_array_ctx skx_set_last_error skx_array_map skx_cov_histogram skx_phases_json skx_phase_add skx_skf_peek_k skx_array_load_filtered skx_ctx_expect_output skx_array_distance_planes skx_planes_distance skx_array_distance_filtered
skx_comm_unique_id skx_comm_create skx_comm_create_local skx_comm_destroy skx_comm_rank skx_comm_world skx_comm_bytes_received skx_comm_transport skx_comm_barrier
skx_comm_allgather skx_comm_allreduce_u32 skx_comm_gather_root skx_shard_range skx_pair_bands skx_keyset_allgather skx_array_reduce_stats skx_array_distance_sharded
skh_build_sharded skh_align_sharded skh_distance_sharded
skh_apply_filters skh_align skh_align_fd skh_distance_tsv skh_nk skh_save_skf skh_load_array skh_sample_name skh_main skh_merge skh_delete skh_weed skh_cov skh_cov_fit skh_align_inputs_fd skh_distance_skf_tsv skh_help skh_log""".split()

_lib = None


def load_library():
    """Load libskx.so (no GPU needed for loading); fails loudly when the extension is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          f"(make -C ska.rust_amd). There is no fallback path.")
    lib = C.CDLL(LIB_PATH)
    vp, cp, i, d, u64 = C.c_void_p, C.c_char_p, C.c_int, C.c_double, C.c_uint64
    pp = C.POINTER(vp)
    lib.skx_last_error.restype = cp
    lib.skx_version.restype = cp
    lib.skx_ctx_create.argtypes = [i, pp]
    lib.skx_ctx_destroy.argtypes = [vp]
    lib.skx_ctx_sync.argtypes = [vp]
    lib.skx_ctx_stream.argtypes = [vp]
    lib.skx_ctx_stream.restype = vp
    lib.skx_ctx_timings.argtypes = [vp, C.POINTER(Timings), i]
    lib.skx_ctx_merge_path.argtypes = [vp]
    lib.skx_ctx_merge_path.restype = C.c_char_p
    lib.skx_dictset_build.argtypes = [vp, C.POINTER(Stream), i, i, i, i, C.POINTER(Qual), pp]
    lib.skx_dictset_build_files.argtypes = [vp, C.POINTER(cp), C.POINTER(cp), i, i, i, C.POINTER(Qual), i, d, pp]
    lib.skx_dictset_free.argtypes = [vp]
    lib.skx_dictset_nsamples.argtypes = [vp]
    lib.skx_dictset_key_bits.argtypes = [vp]
    lib.skx_dictset_size.argtypes = [vp, i, C.POINTER(u64)]
    lib.skx_dictset_export.argtypes = [vp, i, vp, vp, u64]
    lib.skx_keyset_union.argtypes = [vp, vp, pp]
    lib.skx_keyset_union_notes.argtypes = [vp, vp, pp]
    lib.skx_keyset_size.argtypes = [vp, C.POINTER(u64)]
    lib.skx_keyset_device.argtypes = [vp, pp, C.POINTER(u64), C.POINTER(i)]
    lib.skx_keyset_from_device.argtypes = [vp, vp, u64, i, i, pp]
    lib.skx_keyset_merge.argtypes = [vp, pp, i, pp]
    lib.skx_keyset_free.argtypes = [vp]
    lib.skx_array_assemble.argtypes = [vp, vp, vp, C.POINTER(cp), pp]
    lib.skx_array_assemble_lazy.argtypes = [vp, vp, vp, C.POINTER(cp), pp]
    lib.skx_merge.argtypes = [vp, vp, C.POINTER(cp), pp]
    lib.skx_build_and_merge.argtypes = [vp, C.POINTER(cp), C.POINTER(cp), C.POINTER(cp), i, i, i, C.POINTER(Qual), i, d, pp]
    lib.skx_array_free.argtypes = [vp]
    lib.skx_array_save.argtypes = [vp, cp]
    lib.skx_array_load.argtypes = [vp, cp, i, pp]
    lib.skx_array_from_host.argtypes = [vp, i, i, C.POINTER(cp), i, vp, vp, vp, u64, cp, pp]
    lib.skx_array_info.argtypes = [vp, C.POINTER(ArrayInfo)]
    lib.skx_array_name.argtypes = [vp, u64]
    lib.skx_array_name.restype = cp
    lib.skx_array_version.argtypes = [vp]
    lib.skx_array_version.restype = cp
    lib.skx_array_export.argtypes = [vp, vp, vp, vp]
    lib.skx_array_sample_kmers.argtypes = [vp, vp]
    lib.skx_array_pieces_info.argtypes = [vp, C.POINTER(u64), C.POINTER(u64), C.POINTER(C.c_uint32)]
    lib.skx_array_filter.argtypes = [vp, u64, i, i, i, i, i, C.POINTER(C.c_int32)]
    lib.skx_array_write_fasta.argtypes = [vp, i]
    lib.skx_array_fasta.argtypes = [vp, pp, C.POINTER(u64)]
    lib.skx_array_device_matrix.argtypes = [vp, pp, C.POINTER(u64), C.POINTER(u64)]
    lib.skx_array_device_stats.argtypes = [vp, pp, pp, pp, pp]
    lib.skx_array_set_total_samples.argtypes = [vp, u64]
    lib.skx_array_distance.argtypes = [vp, d, i, vp]
    lib.skx_free.argtypes = [vp]
    lib.skh_apply_filters.argtypes = [vp, d, i, i, i, i, C.POINTER(C.c_int32)]
    lib.skh_align.argtypes = [vp, i, i, i, d, i, pp, C.POINTER(u64)]
    lib.skh_distance_tsv.argtypes = [vp, d, i, pp, C.POINTER(u64)]
    lib.skh_nk.argtypes = [vp, i, pp, C.POINTER(u64)]
    lib.skh_save_skf.argtypes = [vp, cp]
    lib.skh_load_array.argtypes = [vp, C.POINTER(cp), i, i, pp]
    lib.skx_array_merge.argtypes = [vp, pp, i, pp]
    lib.skx_array_delete_samples.argtypes = [vp, C.POINTER(cp), i]
    lib.skx_array_weed.argtypes = [vp, vp, i, C.POINTER(u64)]
    lib.skx_keyset_from_fasta.argtypes = [vp, cp, i, i, pp]
    lib.skx_array_map.argtypes = [vp, cp, i, i, i, i, pp, C.POINTER(u64)]
    lib.skx_cov_histogram.argtypes = [vp, cp, cp, i, i, vp]
    lib.skh_cov.argtypes = [vp, cp, cp, i, i, pp, C.POINTER(u64), C.POINTER(u64)]
    lib.skh_cov_fit.argtypes = [vp, u64, C.POINTER(d), C.POINTER(d), C.POINTER(u64)]
    lib.skx_array_ctx.argtypes = [vp]
    lib.skx_array_ctx.restype = vp
    lib.skx_set_last_error.argtypes = [cp]
    lib.skh_merge.argtypes = [vp, C.POINTER(cp), i, cp]
    lib.skh_delete.argtypes = [vp, C.POINTER(cp), i, cp]
    lib.skh_weed.argtypes = [vp, cp, i, d, i, i, i, i, cp]
    lib.skx_read_records.argtypes = [cp, cp, d, i, pp, pp, C.POINTER(u64)]
    lib.skh_sample_name.argtypes = [cp]
    lib.skh_sample_name.restype = vp
    lib.skx_ctx_expect_output.argtypes = [vp, i]
    lib.skx_array_distance_planes.argtypes = [vp, i, pp, C.POINTER(u64), C.POINTER(i)]
    lib.skx_planes_distance.argtypes = [vp, vp, i, u64, i, d, i, i, vp]
    lib.skx_array_distance_filtered.argtypes = [vp, d, i, vp, C.POINTER(C.c_int64), C.POINTER(u64)]
    lib.skx_array_load_filtered.argtypes = [vp, cp, C.POINTER(FilterSpec), pp, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    lib.skh_align_inputs_fd.argtypes = [vp, C.POINTER(cp), i, i, i, i, i, d, i, i]
    lib.skh_distance_skf_tsv.argtypes = [vp, cp, d, i, pp, C.POINTER(u64)]
    lib.skx_phases_json.argtypes = [pp, C.POINTER(u64), i]
    lib.skx_phase_add.argtypes = [cp, d]
    lib.skx_phase_add.restype = None
    lib.skx_comm_unique_id.argtypes = [vp]
    lib.skx_comm_create.argtypes = [vp, i, i, vp, pp]
    lib.skx_comm_create_local.argtypes = [vp, i, i, cp, pp]
    lib.skx_comm_destroy.argtypes = [vp]
    lib.skx_comm_destroy.restype = None
    lib.skx_comm_rank.argtypes = [vp]
    lib.skx_comm_world.argtypes = [vp]
    lib.skx_comm_bytes_received.argtypes = [vp]
    lib.skx_comm_bytes_received.restype = u64
    lib.skx_comm_transport.argtypes = [vp, C.POINTER(i), C.POINTER(i), C.POINTER(i)]
    lib.skx_comm_barrier.argtypes = [vp]
    lib.skx_comm_allgather.argtypes = [vp, vp, vp, u64, i]
    lib.skx_comm_allreduce_u32.argtypes = [vp, vp, u64, i]
    lib.skx_comm_gather_root.argtypes = [vp, vp, vp, vp]
    lib.skx_shard_range.argtypes = [u64, i, i, C.POINTER(u64), C.POINTER(u64)]
    lib.skx_pair_bands.argtypes = [i, i, i, vp]
    lib.skx_keyset_allgather.argtypes = [vp, vp, pp]
    lib.skx_array_reduce_stats.argtypes = [vp, vp, u64]
    lib.skx_array_distance_sharded.argtypes = [vp, vp, i, d, vp, u64]
    lib.skh_build_sharded.argtypes = [vp, vp, C.POINTER(Job)]
    lib.skh_align_sharded.argtypes = [vp, vp, C.POINTER(Job)]
    lib.skh_distance_sharded.argtypes = [vp, vp, C.POINTER(Job)]
    _lib = lib
    return lib


def _check(rc):
    if rc != OK:
        raise EngineError(rc, _lib.skx_last_error().decode())


def _np_ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _take(ptr, n):
    s = C.string_at(ptr, n.value)
    _lib.skx_free(ptr)
    return s


def qual(min_count=5, min_qual=20, qual_filter=QUAL_STRICT):
    return Qual(min_count, min_qual, qual_filter)


def cov_histogram(fq1, fq2, k=31, rc=True, ctx=None):
    """CoverageHistogram::new + histogram (coverage.rs:70-148,158-163) on the device."""
    ctx = ctx or default_context()
    h = np.zeros(1000, np.uint32)
    _check(_lib.skx_cov_histogram(ctx.h, fq1.encode(), fq2.encode(), k, int(rc), _np_ptr(h)))
    return h


def cov_fit(counts):
    c = np.ascontiguousarray(counts, np.float64)
    w0, cc, cut = C.c_double(), C.c_double(), C.c_uint64()
    load_library()
    _check(_lib.skh_cov_fit(_np_ptr(c), len(c), C.byref(w0), C.byref(cc), C.byref(cut)))
    return w0.value, cc.value, cut.value


def cov(fq1, fq2, k=31, rc=True, ctx=None):
    """`ska cov`: (plot_hist text, cutoff)"""
    ctx = ctx or default_context()
    p, n, cut = C.c_void_p(), C.c_uint64(), C.c_uint64()
    _check(_lib.skh_cov(ctx.h, fq1.encode(), fq2.encode(), k, int(rc), C.byref(p), C.byref(n), C.byref(cut)))
    return _take(p, n), cut.value


def phases(reset=False):
    """Wall-clock phases of the host path recorded since the last reset: {name: seconds} in first-use order."""
    import json
    buf, n = C.c_void_p(), C.c_uint64()
    _check(_lib.skx_phases_json(C.byref(buf), C.byref(n), int(reset)))
    return json.loads(_take(buf, n))


def planes_distance(planes_ptr, n_samples, words_per_row, filt_ambig, constant, i_lo, i_hi, ctx=None):
    """VariantDist of the pairs (i in [i_lo, i_hi), j > i) from gathered bit planes (skx_planes_distance)"""
    ctx = ctx or default_context()
    n = sum(n_samples - 1 - i for i in range(i_lo, i_hi))
    out = np.zeros(n, DIST_DT)
    if n == 0:
        return out
    _check(_lib.skx_planes_distance(ctx.h, planes_ptr, n_samples, words_per_row, int(filt_ambig), float(constant), i_lo, i_hi, _np_ptr(out)))
    return out


def read_records(file1, file2=None, proportion_reads=0.0, streaming=False):
    """One sample's files as the record stream the device takes (host only): (seq bytes, qual bytes or None)."""
    lib = load_library()
    ps, pq, n = C.c_void_p(), C.c_void_p(), C.c_uint64()
    _check(lib.skx_read_records(file1.encode(), file2.encode() if file2 else None, float(proportion_reads), int(bool(streaming)), C.byref(ps), C.byref(pq), C.byref(n)))
    seq = C.string_at(ps, n.value)
    q = C.string_at(pq, n.value) if pq.value else None
    lib.skx_free(ps)
    if pq.value:
        lib.skx_free(pq)
    return seq, q


def sample_name(path):
    lib = load_library()
    p = lib.skh_sample_name(path.encode())
    s = C.string_at(p).decode()
    lib.skx_free(p)
    return s


class Context:
    def __init__(self, device=0):
        lib = load_library()
        h = C.c_void_p()
        _check(lib.skx_ctx_create(device, C.byref(h)))
        self.h = h

    def sync(self):
        _check(_lib.skx_ctx_sync(self.h))

    @property
    def stream(self):
        return _lib.skx_ctx_stream(self.h)

    def timings(self, reset=False):
        t = Timings()
        _lib.skx_ctx_timings(self.h, C.byref(t), int(reset))
        return {n: getattr(t, n) for n, _ in Timings._fields_}

    def merge_path(self):
        """'append64' / 'append128' / 'sorted: <why>': the kernels the last merge on this context went through"""
        return _lib.skx_ctx_merge_path(self.h).decode()

    def close(self):
        if getattr(self, "h", None) and _lib is not None:
            _lib.skx_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()


_default_ctx = None


def default_context():
    global _default_ctx
    if _default_ctx is None:
        _default_ctx = Context(int(os.environ.get("LOCAL_RANK", "0")) if os.environ.get("SKX_USE_LOCAL_RANK") else 0)
    return _default_ctx


def record_stream(records):
    """records: iterable of bytes -> the device record stream (each record + b'\\n')."""
    return b"".join(bytes(r) + b"\n" for r in records)


COMM_ID_BYTES = 128


def comm_unique_id():
    """ncclGetUniqueId through the ABI: rank 0 makes it, the host carries the 128 bytes to the other ranks"""
    load_library()
    buf = (C.c_uint8 * COMM_ID_BYTES)()
    _check(_lib.skx_comm_unique_id(buf))
    return bytes(buf)


def shard_range(n_items, rank, world):
    """Contiguous shard [lo, hi) of rank (skx_shard_range; input order kept, cf. merge_ska_dict.rs:243-253,277-291)"""
    load_library()
    lo, hi = C.c_uint64(), C.c_uint64()
    _check(_lib.skx_shard_range(n_items, rank, world, C.byref(lo), C.byref(hi)))
    return lo.value, hi.value


def pair_bands(n_samples, world, align=8):
    """Bands [i_lo, i_hi) of first samples of the pair matrix, one per rank (skx_pair_bands)"""
    load_library()
    out = np.zeros(2 * world, np.int32)
    _check(_lib.skx_pair_bands(n_samples, world, align, _np_ptr(out)))
    return [(int(out[2 * r]), int(out[2 * r + 1])) for r in range(world)]


class Comm:
    """The exchanges of a multi-GPU job behind the C ABI (include/skx.h "Collectives"): RCCL (Comm.rccl) or, for ranks that share
    one device, host-staged through a directory (Comm.local).  Every method is collective."""

    def __init__(self, handle, ctx):
        self.h, self.ctx = handle, ctx

    @classmethod
    def rccl(cls, rank, world, unique_id, ctx=None):
        ctx = ctx or default_context()
        h = C.c_void_p()
        idb = (C.c_uint8 * COMM_ID_BYTES).from_buffer_copy(unique_id)
        _check(_lib.skx_comm_create(ctx.h, rank, world, idb, C.byref(h)))
        return cls(h, ctx)

    @classmethod
    def local(cls, rank, world, directory, ctx=None):
        """ctx may be None: host buffers only (no GPU needed)"""
        load_library()
        h = C.c_void_p()
        _check(_lib.skx_comm_create_local(ctx.h if ctx is not None else None, rank, world, directory.encode(), C.byref(h)))
        return cls(h, ctx)

    rank = property(lambda self: _lib.skx_comm_rank(self.h))
    world = property(lambda self: _lib.skx_comm_world(self.h))
    bytes_received = property(lambda self: int(_lib.skx_comm_bytes_received(self.h)))

    def transport(self):
        """(ranks of the RCCL communicator -- 0 for the host-staged transport --, the device RCCL bound it to, the context's device)"""
        a, b, c = C.c_int(), C.c_int(), C.c_int()
        _check(_lib.skx_comm_transport(self.h, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def barrier(self):
        _check(_lib.skx_comm_barrier(self.h))

    def allgather_host(self, arr):
        """numpy array (same shape on every rank) -> [world, ...]"""
        a = np.ascontiguousarray(arr)
        out = np.empty((self.world,) + a.shape, a.dtype)
        _check(_lib.skx_comm_allgather(self.h, _np_ptr(a), _np_ptr(out), a.nbytes, 0))
        return out

    def allgather_device(self, send_ptr, recv_ptr, nbytes):
        _check(_lib.skx_comm_allgather(self.h, send_ptr, recv_ptr, nbytes, 1))

    def allreduce_u32_host(self, arr):
        a = np.ascontiguousarray(arr, np.uint32).copy()
        _check(_lib.skx_comm_allreduce_u32(self.h, _np_ptr(a), a.size, 0))
        return a

    def gather_root_host(self, arr, sizes):
        """byte blocks of different sizes (sizes known everywhere) -> rank 0: one uint8 array there, None elsewhere"""
        a = np.ascontiguousarray(arr).view(np.uint8).reshape(-1)
        sz = np.ascontiguousarray(sizes, np.uint64)
        out = np.empty(int(sz.sum()) if self.rank == 0 else 0, np.uint8)
        _check(_lib.skx_comm_gather_root(self.h, _np_ptr(a), _np_ptr(sz), _np_ptr(out)))
        return out if self.rank == 0 else None

    def allreduce_u32_device(self, ptr, n):
        _check(_lib.skx_comm_allreduce_u32(self.h, ptr, n, 1))

    def keyset_allgather(self, local):
        """exchange 1: the global row set from every rank's key table (skx_keyset_allgather)"""
        h = C.c_void_p()
        _check(_lib.skx_keyset_allgather(self.h, local.h, C.byref(h)))
        return KeySet(h, self.ctx)

    def reduce_stats(self, array, total_samples):
        """exchange 2: per-row filter statistics over all ranks (skx_array_reduce_stats)"""
        _check(_lib.skx_array_reduce_stats(self.h, array.h, total_samples))

    def distance_sharded(self, array, total_samples, filt_ambig=True, constant=0.0):
        """exchange 3 + the pair sweep by bands (skx_array_distance_sharded): the whole table on rank 0, None elsewhere"""
        n = total_samples * (total_samples - 1) // 2
        out = np.zeros(n + 1 if self.rank == 0 else 1, DIST_DT)
        _check(_lib.skx_array_distance_sharded(self.h, array.h, int(filt_ambig), float(constant), _np_ptr(out), out.size))
        return out[:n] if self.rank == 0 else None

    def _job(self, names, inputs, k, rc, q, threads, proportion_reads, output):
        n = len(names)
        keep = [(C.c_char_p * n)(*[x.encode() for x in names]), (C.c_char_p * n)(*[a.encode() for a, _ in inputs]),
                (C.c_char_p * n)(*[(b.encode() if b else None) for _, b in inputs]), output.encode() if output else None]
        j = Job()
        j.names, j.file1, j.file2, j.n_samples = keep[0], keep[1], keep[2], n
        j.k, j.rc, j.qual, j.threads, j.proportion_reads, j.output = k, int(rc), q or qual(), threads, proportion_reads, keep[3]
        return j, keep

    def build(self, names, inputs, output, k=31, rc=True, q=None, threads=1, proportion_reads=0.0, merge_parts=False):
        """one rank of `ska build` over several GPUs (skh_build_sharded); inputs = [(file1, file2 | None)] of the WHOLE job"""
        j, keep = self._job(names, inputs, k, rc, q, threads, proportion_reads, output)
        j.merge_parts = int(merge_parts)
        _check(_lib.skh_build_sharded(self.ctx.h, self.h, C.byref(j)))

    def align(self, names, inputs, output, k=31, rc=True, q=None, threads=1, proportion_reads=0.0, min_freq=0.9, filter_type=FILTER_NO_CONST,
              mask_ambig=False, ignore_const_gaps=False, filter_ambig_as_missing=False):
        j, keep = self._job(names, inputs, k, rc, q, threads, proportion_reads, output)
        j.min_freq, j.filter_type, j.mask_ambig, j.ignore_const_gaps, j.filter_ambig_as_missing = min_freq, filter_type, int(mask_ambig), int(ignore_const_gaps), int(filter_ambig_as_missing)
        _check(_lib.skh_align_sharded(self.ctx.h, self.h, C.byref(j)))

    def distance(self, names, inputs, output, k=31, rc=True, q=None, threads=1, proportion_reads=0.0, min_freq=0.0, filt_ambig=True):
        j, keep = self._job(names, inputs, k, rc, q, threads, proportion_reads, output)
        j.min_freq, j.filt_ambig = min_freq, int(filt_ambig)
        _check(_lib.skh_distance_sharded(self.ctx.h, self.h, C.byref(j)))

    def free(self):
        if getattr(self, "h", None) and _lib is not None:
            _lib.skx_comm_destroy(self.h)
            self.h = None

    def __del__(self):
        self.free()


class DictSet:
    """A batch of per-sample SkaDicts (ska_dict.rs:56-69), device resident."""

    def __init__(self, handle, ctx):
        self.h, self.ctx = handle, ctx

    @classmethod
    def build(cls, streams, k, rc=True, q=None, ctx=None, quals=None):
        """streams: list of record-stream bytes (host); quals: optional list of quality streams."""
        ctx = ctx or default_context()
        q = q or qual()
        n = len(streams)
        bufs = [np.frombuffer(s, dtype=np.uint8) for s in streams]
        qbufs = [np.frombuffer(s, dtype=np.uint8) if s is not None else None for s in (quals or [None] * n)]
        arr = (Stream * n)()
        for i in range(n):
            arr[i].seq = bufs[i].ctypes.data if len(bufs[i]) else None
            arr[i].qual = qbufs[i].ctypes.data if qbufs[i] is not None else None
            arr[i].len = len(bufs[i])
        h = C.c_void_p()
        _check(_lib.skx_dictset_build(ctx.h, arr, n, 0, k, int(rc), C.byref(q), C.byref(h)))
        return cls(h, ctx)

    @classmethod
    def build_device(cls, ptrs, lens, k, rc=True, q=None, ctx=None):
        """ptrs: device pointers (ints) of 16-B aligned record streams already in HBM."""
        ctx = ctx or default_context()
        q = q or qual()
        n = len(ptrs)
        arr = (Stream * n)()
        for i in range(n):
            arr[i].seq, arr[i].qual, arr[i].len = ptrs[i], None, lens[i]
        h = C.c_void_p()
        _check(_lib.skx_dictset_build(ctx.h, arr, n, 1, k, int(rc), C.byref(q), C.byref(h)))
        return cls(h, ctx)

    @classmethod
    def from_files(cls, inputs, k, rc=True, q=None, threads=1, proportion_reads=0.0, ctx=None):
        ctx = ctx or default_context()
        q = q or qual()
        n = len(inputs)
        f1 = (C.c_char_p * n)(*[x[0].encode() for x in inputs])
        f2 = (C.c_char_p * n)(*[(x[1].encode() if x[1] else None) for x in inputs])
        h = C.c_void_p()
        _check(_lib.skx_dictset_build_files(ctx.h, f1, f2, n, k, int(rc), C.byref(q), threads, proportion_reads, C.byref(h)))
        return cls(h, ctx)

    @property
    def nsamples(self):
        return _lib.skx_dictset_nsamples(self.h)

    @property
    def key_bits(self):
        return _lib.skx_dictset_key_bits(self.h)

    def size(self, sample):
        n = C.c_uint64()
        _check(_lib.skx_dictset_size(self.h, sample, C.byref(n)))
        return n.value

    def export(self, sample):
        n = self.size(sample)
        keys = np.zeros(n, KEY_DT)
        bases = np.zeros(n, np.uint8)
        _check(_lib.skx_dictset_export(self.h, sample, _np_ptr(keys), _np_ptr(bases), n))
        return keys, bases

    def union_keys(self, notes=False):
        """notes: keep the pass's notes for the eager assemble that follows (directly or after KeySet all-gather): skx_keyset_union_notes"""
        h = C.c_void_p()
        _check((_lib.skx_keyset_union_notes if notes else _lib.skx_keyset_union)(self.ctx.h, self.h, C.byref(h)))
        return KeySet(h, self.ctx)

    def merge(self, names):
        n = len(names)
        nm = (C.c_char_p * n)(*[x.encode() for x in names])
        h = C.c_void_p()
        _check(_lib.skx_merge(self.ctx.h, self.h, nm, C.byref(h)))
        return Array(h, self.ctx)

    def assemble(self, rows, names):
        n = len(names)
        nm = (C.c_char_p * n)(*[x.encode() for x in names])
        h = C.c_void_p()
        _check(_lib.skx_array_assemble(self.ctx.h, self.h, rows.h, nm, C.byref(h)))
        return Array(h, self.ctx)

    def assemble_lazy(self, rows, names):
        """rows + these dictionaries as an array without a matrix; the dictset and the row set pass into the array (both handles end here)"""
        n = len(names)
        nm = (C.c_char_p * n)(*[x.encode() for x in names])
        h = C.c_void_p()
        dh, rh = self.h, rows.h
        self.h = None; rows.h = None
        _check(_lib.skx_array_assemble_lazy(self.ctx.h, dh, rh, nm, C.byref(h)))
        return Array(h, self.ctx)

    def free(self):
        if getattr(self, "h", None) and _lib is not None:
            _lib.skx_dictset_free(self.h)
            self.h = None

    def __del__(self):
        self.free()


class KeySet:
    def __init__(self, handle, ctx):
        self.h, self.ctx = handle, ctx

    def __len__(self):
        n = C.c_uint64()
        _check(_lib.skx_keyset_size(self.h, C.byref(n)))
        return n.value

    def device(self):
        p, n, w = C.c_void_p(), C.c_uint64(), C.c_int()
        _check(_lib.skx_keyset_device(self.h, C.byref(p), C.byref(n), C.byref(w)))
        return p.value, n.value, w.value

    @classmethod
    def from_device(cls, ptr, n_keys, k, rc=True, ctx=None):
        ctx = ctx or default_context()
        h = C.c_void_p()
        _check(_lib.skx_keyset_from_device(ctx.h, ptr, n_keys, k, int(rc), C.byref(h)))
        return cls(h, ctx)

    @classmethod
    def from_fasta(cls, path, k, rc=True, ctx=None):
        """RefSka::new + kmer_iter (ska_ref.rs:189-262,541): the split k-mers of a FASTA file (what `ska weed` removes)."""
        ctx = ctx or default_context()
        h = C.c_void_p()
        _check(_lib.skx_keyset_from_fasta(ctx.h, path.encode(), k, int(rc), C.byref(h)))
        return cls(h, ctx)

    @classmethod
    def merge(cls, sets, ctx=None):
        ctx = ctx or sets[0].ctx
        hs = (C.c_void_p * len(sets))(*[s.h for s in sets])
        h = C.c_void_p()
        _check(_lib.skx_keyset_merge(ctx.h, hs, len(sets), C.byref(h)))
        return cls(h, ctx)

    def free(self):
        if getattr(self, "h", None) and _lib is not None:
            _lib.skx_keyset_free(self.h)
            self.h = None

    def __del__(self):
        self.free()


class Array:
    """MergeSkaArray (merge_ska_array.rs:109-126) on the device."""

    def __init__(self, handle, ctx):
        self.h, self.ctx = handle, ctx

    @classmethod
    def build(cls, inputs, k=31, rc=True, q=None, threads=1, proportion_reads=0.0, ctx=None):
        """inputs: list of (name, file1, file2|None): build_and_merge + MergeSkaArray::new."""
        ctx = ctx or default_context()
        q = q or qual()
        n = len(inputs)
        names = (C.c_char_p * n)(*[x[0].encode() for x in inputs])
        f1 = (C.c_char_p * n)(*[x[1].encode() for x in inputs])
        f2 = (C.c_char_p * n)(*[(x[2].encode() if x[2] else None) for x in inputs])
        h = C.c_void_p()
        _check(_lib.skx_build_and_merge(ctx.h, names, f1, f2, n, k, int(rc), C.byref(q), threads, proportion_reads, C.byref(h)))
        return cls(h, ctx)

    @classmethod
    def load(cls, path, want_bits=0, ctx=None):
        ctx = ctx or default_context()
        h = C.c_void_p()
        if want_bits:
            _check(_lib.skx_array_load(ctx.h, path.encode(), want_bits, C.byref(h)))
        else:
            p = (C.c_char_p * 1)(path.encode())
            _check(_lib.skh_load_array(ctx.h, p, 1, 1, C.byref(h)))
        return cls(h, ctx)

    # ---- .skf life-cycle: ska merge / delete / weed (generic_modes.rs:90-106,192-267) ----
    @classmethod
    def load_filtered(cls, path, min_freq=0.9, filter_ambig_as_missing=False, filter_type=FILTER_NO_CONST, mask_ambig=False, ignore_const_gaps=False,
                      two_stage=False, ctx=None):
        """skx_array_load_filtered: load + apply_filters (or distance's two filters) in one pass -> (array, removed, constant)"""
        ctx = ctx or default_context()
        fs = FilterSpec(min_freq, int(filter_ambig_as_missing), filter_type, int(mask_ambig), int(ignore_const_gaps), int(two_stage))
        h, rem, cst = C.c_void_p(), C.c_int64(), C.c_int64()
        _check(_lib.skx_array_load_filtered(ctx.h, path.encode(), C.byref(fs), C.byref(h), C.byref(rem), C.byref(cst)))
        return cls(h, ctx), rem.value, cst.value

    @classmethod
    def merge(cls, arrays, ctx=None):
        ctx = ctx or arrays[0].ctx
        hs = (C.c_void_p * len(arrays))(*[a.h for a in arrays])
        h = C.c_void_p()
        _check(_lib.skx_array_merge(ctx.h, hs, len(arrays), C.byref(h)))
        return cls(h, ctx)

    def delete_samples(self, names):
        nm = (C.c_char_p * max(len(names), 1))(*[x.encode() for x in names])
        _check(_lib.skx_array_delete_samples(self.h, nm, len(names)))

    def map(self, reference, fmt="aln", ambig_mask=False, repeat_mask=False, threads=0):
        """generic_modes::map: RefSka::new + map + write_aln | write_vcf -> text (generic_modes.rs:56-84)."""
        p, n = C.c_void_p(), C.c_uint64()
        _check(_lib.skx_array_map(self.h, reference.encode(), int(ambig_mask), int(repeat_mask), 1 if fmt == "vcf" else 0, threads, C.byref(p), C.byref(n)))
        return _take(p, n)

    def weed_keys(self, keyset, reverse=False):
        r = C.c_uint64()
        _check(_lib.skx_array_weed(self.h, keyset.h, int(reverse), C.byref(r)))
        return r.value

    def weed(self, weed_fasta=None, reverse=False, min_freq=0.9, filter_ambig_as_missing=False, filter_type=FILTER_NONE,
             ambig_mask=False, ignore_const_gaps=False):
        _check(_lib.skh_weed(self.h, weed_fasta.encode() if weed_fasta else None, int(reverse), min_freq, int(filter_ambig_as_missing),
                             filter_type, int(ambig_mask), int(ignore_const_gaps), None))

    @classmethod
    def from_host(cls, k, rc, names, keys, variants, counts=None, version=None, ctx=None):
        ctx = ctx or default_context()
        keys = np.ascontiguousarray(keys, dtype=KEY_DT)
        variants = np.ascontiguousarray(variants, dtype=np.uint8)
        n = len(names)
        nm = (C.c_char_p * n)(*[x.encode() for x in names])
        h = C.c_void_p()
        counts = np.ascontiguousarray(counts, dtype=np.uint64) if counts is not None else None
        _check(_lib.skx_array_from_host(ctx.h, k, int(rc), nm, n, _np_ptr(keys), _np_ptr(variants), _np_ptr(counts), len(keys),
                                        version.encode() if version else None, C.byref(h)))
        return cls(h, ctx)

    def save(self, path):
        _check(_lib.skx_array_save(self.h, path.encode()))

    def save_skf(self, prefix):
        _check(_lib.skh_save_skf(self.h, prefix.encode()))

    def _info(self):
        info = ArrayInfo()
        _check(_lib.skx_array_info(self.h, C.byref(info)))
        return info

    k = property(lambda s: s._info().k)
    rc = property(lambda s: bool(s._info().rc))
    k_bits = property(lambda s: s._info().k_bits)
    nrows = property(lambda s: s._info().n_rows)
    nkmers = property(lambda s: s._info().n_kmers)
    nsamples = property(lambda s: s._info().n_samples)
    version = property(lambda s: _lib.skx_array_version(s.h).decode())

    @property
    def names(self):
        return [_lib.skx_array_name(self.h, i).decode() for i in range(self.nsamples)]

    def export(self):
        info = self._info()
        keys = np.zeros(info.n_kmers, KEY_DT)
        var = np.zeros((info.n_rows, info.n_samples), np.uint8)
        counts = np.zeros(info.n_rows, np.uint64)
        _check(_lib.skx_array_export(self.h, _np_ptr(keys), _np_ptr(var), _np_ptr(counts)))
        return keys, var, counts

    def export_keys(self):
        """split k-mers (sorted) and their counts without moving the matrix"""
        info = self._info()
        keys = np.zeros(info.n_kmers, KEY_DT)
        counts = np.zeros(info.n_rows, np.uint64)
        _check(_lib.skx_array_export(self.h, _np_ptr(keys), None, _np_ptr(counts)))
        return keys, counts

    def sample_kmers(self):
        out = np.zeros(self.nsamples, np.int64)
        _check(_lib.skx_array_sample_kmers(self.h, _np_ptr(out)))
        return out

    def pieces_info(self):
        """(bytes of pieces, row blocks, ranks per block) of an array still held as the append pass left it; (0, 0, 0) otherwise"""
        b, j, c = C.c_uint64(), C.c_uint64(), C.c_uint32()
        _check(_lib.skx_array_pieces_info(self.h, C.byref(b), C.byref(j), C.byref(c)))
        return b.value, j.value, c.value

    def filter(self, min_count, filter_ambig_as_missing=False, filter_type=FILTER_NO_CONST, mask_ambig=False,
               ignore_const_gaps=False, update_kmers=True):
        r = C.c_int32()
        _check(_lib.skx_array_filter(self.h, min_count, int(filter_ambig_as_missing), filter_type, int(mask_ambig),
                                     int(ignore_const_gaps), int(update_kmers), C.byref(r)))
        return r.value

    def apply_filters(self, min_freq, filter_ambig_as_missing=False, filter_type=FILTER_NO_CONST, ambig_mask=False,
                      ignore_const_gaps=False):
        r = C.c_int32()
        _check(_lib.skh_apply_filters(self.h, min_freq, int(filter_ambig_as_missing), filter_type, int(ambig_mask),
                                      int(ignore_const_gaps), C.byref(r)))
        return r.value

    def fasta(self):
        p, n = C.c_void_p(), C.c_uint64()
        _check(_lib.skx_array_fasta(self.h, C.byref(p), C.byref(n)))
        return _take(p, n)

    def write_fasta(self, fd):
        """write_fasta streamed to a file descriptor (merge_ska_array.rs:507-520)."""
        _check(_lib.skx_array_write_fasta(self.h, int(fd)))

    def device_matrix(self):
        p, pitch, rows = C.c_void_p(), C.c_uint64(), C.c_uint64()
        _check(_lib.skx_array_device_matrix(self.h, C.byref(p), C.byref(pitch), C.byref(rows)))
        return p.value, pitch.value, rows.value

    def device_stats(self):
        p = [C.c_void_p() for _ in range(4)]
        _check(_lib.skx_array_device_stats(self.h, *[C.byref(x) for x in p]))
        return [x.value for x in p]

    def set_total_samples(self, total):
        _check(_lib.skx_array_set_total_samples(self.h, total))

    def nk(self, full_info=False):
        p, n = C.c_void_p(), C.c_uint64()
        _check(_lib.skh_nk(self.h, int(full_info), C.byref(p), C.byref(n)))
        return _take(p, n)

    def distance(self, constant=0.0, filt_ambig=True):
        s = self.nsamples
        out = np.zeros(s * (s - 1) // 2, DIST_DT)
        _check(_lib.skx_array_distance(self.h, constant, int(filt_ambig), _np_ptr(out)))
        return out

    def distance_filtered(self, min_freq=0.0, filt_ambig=True):
        """generic_modes::distance without filtering the array itself -> (pairs, constant sites, rows used)"""
        s = self.nsamples
        out = np.zeros(max(s * (s - 1) // 2, 1), DIST_DT)
        cst, rows = C.c_int64(), C.c_uint64()
        _check(_lib.skx_array_distance_filtered(self.h, float(min_freq), int(filt_ambig), _np_ptr(out), C.byref(cst), C.byref(rows)))
        return out[: s * (s - 1) // 2], cst.value, rows.value

    def distance_planes(self, filt_ambig=True):
        """device pointer to this array's bit planes [n_planes][n_samples][words_per_row] -> (ptr, words_per_row, n_planes)"""
        p, w, n = C.c_void_p(), C.c_uint64(), C.c_int()
        _check(_lib.skx_array_distance_planes(self.h, int(filt_ambig), C.byref(p), C.byref(w), C.byref(n)))
        return p.value, w.value, n.value

    def distance_tsv(self, min_freq=0.0, filt_ambig=True):
        p, n = C.c_void_p(), C.c_uint64()
        _check(_lib.skh_distance_tsv(self.h, min_freq, int(filt_ambig), C.byref(p), C.byref(n)))
        return _take(p, n)

    def align(self, filter_type=FILTER_NO_CONST, mask_ambig=False, ignore_const_gaps=False, min_freq=0.9,
              filter_ambig_as_missing=False):
        p, n = C.c_void_p(), C.c_uint64()
        _check(_lib.skh_align(self.h, filter_type, int(mask_ambig), int(ignore_const_gaps), min_freq,
                              int(filter_ambig_as_missing), C.byref(p), C.byref(n)))
        return _take(p, n)

    def free(self):
        if getattr(self, "h", None) and _lib is not None:
            _lib.skx_array_free(self.h)
            self.h = None

    def __del__(self):
        self.free()
