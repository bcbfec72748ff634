"""The reference's own integration tests (tests/*.rs of bacpop/ska.rust), restated engine-agnostically.

Each case takes an engine `E` exposing the mirrored reference API
(E.Array.build / load / save / align / distance_tsv / nk, E.qual, E.FILTER_*,
E.QUAL_*, E.sample_name).  tests/test_oracle_golden.py runs them on the CPU
oracle (pins the oracle); tests/test_gpu_golden.py runs the same cases on the
HIP path through the C ABI.
"""
import os
import re

HERE = os.path.dirname(os.path.abspath(__file__))
IN = os.path.join(HERE, "golden", "input")
OK = os.path.join(HERE, "golden", "correct")


def fin(name):
    return os.path.join(IN, name)


def correct(name):
    with open(os.path.join(OK, name), "rb") as f:
        return f.read()


def var_hash(aln: bytes):
    """tests/common/mod.rs:166-189"""
    lines = aln.decode().splitlines()
    seqs = [lines[i] for i in range(1, len(lines), 2)]
    if not seqs:
        return set()
    return {tuple(s[c] for s in seqs) for c in range(len(seqs[0]))}


def aln_length(aln: bytes):
    lines = aln.decode().splitlines()
    return [len(lines[i]) for i in range(1, len(lines), 2)]


def matches_path(out: bytes, golden: bytes):
    """snapbox stdout_matches_path: `[..]` is a within-line wildcard."""
    o, g = out.decode().split("\n"), golden.decode().split("\n")
    assert len(o) == len(g), (out, golden)
    for a, b in zip(o, g):
        pat = ".*".join(re.escape(x) for x in b.split("[..]"))
        assert re.fullmatch(pat, a), (a, b)


def fasta_inputs(E, files):
    return [(E.sample_name(f), f, None) for f in files]


def rfile(prefix, fastq):
    if fastq:
        return [(f"{prefix}_{i}", fin(f"{prefix}_{i}_fwd.fastq.gz"), fin(f"{prefix}_{i}_rev.fastq.gz")) for i in (1, 2)]
    return [(f"{prefix}_{i}", fin(f"{prefix}_{i}.fa"), None) for i in (1, 2)]


def roundtrip(E, arr, tmp_path, name):
    """`ska build -o x` then a second command loading x.skf: exercise save+load like the CLI tests do."""
    p = os.path.join(str(tmp_path), name + ".skf")
    arr.save(p)
    return E.Array.load(p)


# ---------------------------------------------------------------- tests/align.rs
def case_build_proportion_reads(E, tmp_path):       # align.rs:33-60
    a = E.Array.build(fasta_inputs(E, [fin("proportion_reads.fa")]), k=17, rc=False, proportion_reads=0.5)
    a = roundtrip(E, a, tmp_path, "build_proportion_reads")
    matches_path(a.nk(full_info=True), correct("proportion_reads.stdout"))


def case_build_and_align(E, tmp_path):              # align.rs:86-114, basic_align :169-184
    a = E.Array.build(fasta_inputs(E, [fin("test_1.fa"), fin("test_2.fa")]), k=15)
    a = roundtrip(E, a, tmp_path, "basic_build")
    assert var_hash(a.align()) == {("A", "T"), ("C", "T")}
    b = E.Array.build(fasta_inputs(E, [fin("test_1.fa"), fin("test_2.fa")]))   # `ska align a.fa b.fa` defaults
    assert var_hash(b.align()) == {("A", "T"), ("C", "T")}


def case_long_kmers(E, tmp_path):                   # align.rs:116-167
    a = E.Array.build(fasta_inputs(E, [fin("test_1.fa"), fin("test_2.fa")]), k=33)
    a = roundtrip(E, a, tmp_path, "build_k33")
    assert var_hash(a.align()) == {("C", "T"), ("T", "A")}
    a = E.Array.load(os.path.join(str(tmp_path), "build_k33.skf"))
    matches_path(a.nk(), correct("k33.stdout"))
    try:
        E.Array.build(fasta_inputs(E, [fin("test_1.fa"), fin("test_2.fa")]), k=65)
    except Exception:
        pass
    else:
        raise AssertionError("k=65 must fail")


def case_filters(E, tmp_path):                      # align.rs:186-347
    def load():
        return E.Array.load(fin("merge_k9.skf"))
    assert set(aln_length(load().align(filter_type=E.FILTER_NONE, ignore_const_gaps=True))) == {38}
    assert set(aln_length(load().align(filter_type=E.FILTER_NO_AMBIG, filter_ambig_as_missing=True))) == {37}
    assert var_hash(load().align(filter_type=E.FILTER_NO_CONST)) == {("T", "A"), ("C", "T"), ("S", "G")}
    assert var_hash(load().align(filter_type=E.FILTER_NO_AMBIG_OR_CONST)) == {("T", "A"), ("C", "T")}
    assert var_hash(load().align(filter_type=E.FILTER_NO_CONST, mask_ambig=True)) == {("T", "A"), ("C", "T"), ("N", "G")}
    assert set(aln_length(load().align(filter_type=E.FILTER_NO_CONST, min_freq=0.0))) == {33}
    assert set(aln_length(load().align(filter_type=E.FILTER_NO_CONST, min_freq=0.0, ignore_const_gaps=True))) == {3}
    assert set(aln_length(load().align(filter_type=E.FILTER_NO_AMBIG_OR_CONST, min_freq=0.0))) == {32}
    assert set(aln_length(load().align(filter_type=E.FILTER_NO_AMBIG_OR_CONST, min_freq=0.0, ignore_const_gaps=True))) == {2}


def case_parallel_align(E, tmp_path):               # align.rs:349-397
    d = os.path.join(IN, "par_test")
    inputs = [(os.path.join(d, f), os.path.join(d, f), None) for f in sorted(os.listdir(d))]
    s = roundtrip(E, E.Array.build(inputs, k=15, threads=1), tmp_path, "serial_build")
    p = roundtrip(E, E.Array.build(inputs, k=15, threads=4), tmp_path, "parallel_build")
    assert var_hash(s.align()) == var_hash(p.align())
    assert s.nsamples == 45


# ---------------------------------------------------------------- tests/fasta_input.rs
def case_align_n(E, tmp_path):                      # fasta_input.rs:11-32
    a = E.Array.build(fasta_inputs(E, [fin("N_test_1.fa"), fin("N_test_2.fa")]))
    a = roundtrip(E, a, tmp_path, "N_test")
    assert a.align() == correct("align_N.stdout")


def case_rev_comp(E, tmp_path):                     # fasta_input.rs:60-154
    no_rc = E.Array.build(fasta_inputs(E, [fin("test_1.fa"), fin("test_2.fa")]), k=15).align()
    rc = E.Array.build(fasta_inputs(E, [fin("test_1.fa"), fin("test_2_rc.fa")]), k=15).align()
    assert var_hash(no_rc) == var_hash(rc)
    ss = E.Array.build(fasta_inputs(E, [fin("test_1.fa"), fin("test_2_rc.fa")]), k=15, rc=False).align()
    assert var_hash(ss) == set()
    k33 = E.Array.build(fasta_inputs(E, [fin("test_1.fa"), fin("test_2.fa")]), k=33, rc=False).align()
    assert var_hash(k33) == {("T", "A"), ("G", "A")}


def case_repeats(E, tmp_path):                      # fasta_input.rs:156-220 (the weed step is next-tier N1)
    a = E.Array.build(fasta_inputs(E, [fin("dup_test_1.fa"), fin("dup_test_2.fa")]), k=9, rc=False)
    a = roundtrip(E, a, tmp_path, "dup_ss")
    assert a.align() == correct("dup_ss.stdout")
    b = E.Array.build(fasta_inputs(E, [fin("dup_test_1.fa"), fin("dup_test_2.fa")]), k=9)
    assert roundtrip(E, b, tmp_path, "dup_rc").align() == correct("dup_rc.stdout")


def case_palindromes(E, tmp_path):                  # fasta_input.rs:222-291
    a = E.Array.build(fasta_inputs(E, [fin("palindrome_1.fa"), fin("palindrome_2.fa")]), k=15)
    assert roundtrip(E, a, tmp_path, "otto").align(filter_type=E.FILTER_NONE) == correct("palindrome.stdout")
    b = E.Array.build(fasta_inputs(E, [fin("palindrome_1.fa"), fin("palindrome_2.fa")]), k=15, rc=False)
    assert roundtrip(E, b, tmp_path, "otan").align() == correct("palindrome_norc.stdout")
    c = E.Array.build(fasta_inputs(E, [fin("palindrome_reps_1.fa"), fin("palindrome_reps_2.fa")]), k=15)
    assert roundtrip(E, c, tmp_path, "ottootto").align(filter_type=E.FILTER_NONE) == correct("palindrome_reps.stdout")


# ---------------------------------------------------------------- tests/fastq_input.rs
def case_align_fastq(E, tmp_path):                  # fastq_input.rs:11-55
    r = E.Array.build(rfile("test", True), k=9, q=E.qual(min_count=2, min_qual=2))
    f = E.Array.build(fasta_inputs(E, [fin("test_1.fa"), fin("test_2.fa")]), k=9)
    assert var_hash(roundtrip(E, r, tmp_path, "reads").align()) == var_hash(roundtrip(E, f, tmp_path, "fasta_k9").align())


def case_count_check(E, tmp_path):                  # fastq_input.rs:57-109
    c1 = E.Array.build(rfile("test_count", True), k=7, q=E.qual(min_count=1))
    assert var_hash(c1.align()) == {("C", "W")}
    c3 = E.Array.build(rfile("test_count", True), k=7, q=E.qual(min_count=3))
    assert var_hash(c3.align()) == {("C", "T")}


def case_count_check_long(E, tmp_path):             # fastq_input.rs:111-192
    c1 = E.Array.build(rfile("test_long", True), k=63, q=E.qual(min_count=1))
    assert var_hash(c1.align()) == {("G", "M")}
    c3 = E.Array.build(rfile("test_long", True), k=63, q=E.qual(min_count=3))
    h3 = var_hash(roundtrip(E, c3, tmp_path, "reads_k63_c3").align())
    assert h3 == {("G", "A")}
    ss = E.Array.build(rfile("test_long", True), k=63, rc=False, q=E.qual(min_count=2))
    assert var_hash(ss.align()) == h3


def case_map_fastq(E, tmp_path):                    # fastq_input.rs:196-276: the one reference test that joins the reads path to `ska map`
    r = roundtrip(E, E.Array.build(rfile("test", True), k=9, q=E.qual(min_count=1, min_qual=2)), tmp_path, "reads")
    f = roundtrip(E, E.Array.build(fasta_inputs(E, [fin("test_1.fa"), fin("test_2.fa")]), k=9), tmp_path, "assemblies")
    aln = r.map(fin("test_ref.fa"))
    assert aln == f.map(fin("test_ref.fa")) and aln.startswith(b">test_1\n")
    r = E.Array.load(os.path.join(str(tmp_path), "reads.skf"))                      # (a second `ska map` invocation: the file again)
    f = E.Array.load(os.path.join(str(tmp_path), "assemblies.skf"))
    vcf = r.map(fin("test_ref.fa"), fmt="vcf")
    assert vcf == f.map(fin("test_ref.fa"), fmt="vcf") and vcf.startswith(b"##fileformat=VCF")


def case_error_fastq(E, tmp_path):                  # fastq_input.rs:276-470
    allh = var_hash(E.Array.build(rfile("test", True), k=9, q=E.qual(min_count=3, min_qual=2)).align())
    assert allh
    nofilter = E.Array.build(rfile("test_quality", True), k=9, q=E.qual(min_count=5, qual_filter=E.QUAL_NOFILTER))
    assert var_hash(nofilter.align()) == allh
    mid = E.Array.build(rfile("test_quality_base", True), k=9,
                        q=E.qual(min_count=5, qual_filter=E.QUAL_MIDDLE, min_qual=5))
    assert var_hash(mid.align()) == allh
    allh = allh - {("C", "T")}                       # "With errors"
    err = E.Array.build(rfile("test_error", True), k=9, q=E.qual(min_count=5, min_qual=2))
    assert var_hash(err.align()) == allh
    qual30 = E.Array.build(rfile("test_quality", True), k=9, q=E.qual(min_count=5, min_qual=30))
    assert var_hash(qual30.align()) == allh
    strict = E.Array.build(rfile("test_quality_base", True), k=9,
                           q=E.qual(min_count=5, min_qual=5, qual_filter=E.QUAL_STRICT))
    assert var_hash(strict.align()) == allh
    dflt = E.Array.build(rfile("test_quality_base", True), k=9, q=E.qual(min_count=5))
    assert var_hash(dflt.align()) == allh


# ---------------------------------------------------------------- tests/distance.rs
def case_basic_dists(E, tmp_path):                  # distance.rs:9-41
    assert E.Array.load(fin("merge.skf")).distance_tsv() == correct("merge.dist.stdout")
    assert E.Array.load(fin("merge_k41.skf")).distance_tsv() == correct("merge_k41.dist.stdout")


def case_dist_filter(E, tmp_path):                  # distance.rs:44-77
    assert E.Array.load(fin("merge_k9.skf")).distance_tsv(filt_ambig=False) == correct("merge_k9.dist.stdout")
    assert E.Array.load(fin("merge_k9.skf")).distance_tsv() == correct("merge_k9_no_ambig.dist.stdout")
    assert E.Array.load(fin("merge_k9.skf")).distance_tsv(min_freq=1.0) == correct("merge_k9_min_freq.dist.stdout")


def case_multisample_dists(E, tmp_path):            # distance.rs:79-129
    files = ["N_test_1.fa", "N_test_2.fa", "ambig_test_1.fa", "ambig_test_2.fa", "test_1.fa", "test_2.fa"]
    a = E.Array.build(fasta_inputs(E, [fin(f) for f in files]), k=9)
    p = os.path.join(str(tmp_path), "multidist.skf")
    a.save(p)
    assert E.Array.load(p).distance_tsv() == correct("multidist.stdout")
    assert E.Array.load(p).distance_tsv(min_freq=0.9) == correct("multidist.minfreq.stdout")
    assert E.Array.load(p).distance_tsv(filt_ambig=False) == correct("multidist.ambig.stdout")
    # the fixture written by the Rust binary gives the same tables
    assert E.Array.load(fin("multidist.skf")).distance_tsv() == correct("multidist.stdout")


# ---------------------------------------------------------------- tests/skf_ops.rs (nk part)
def case_merge_nk(E, tmp_path):                     # skf_ops.rs merge_nk golden: build of test_1/test_2 at k=31
    a = E.Array.build(fasta_inputs(E, [fin("test_1.fa"), fin("test_2.fa")]), k=31)
    matches_path(roundtrip(E, a, tmp_path, "merge").nk(), correct("merge_nk.stdout"))


def case_dup_ss_nk_shape(E, tmp_path):
    a = E.Array.build(fasta_inputs(E, [fin("dup_test_1.fa"), fin("dup_test_2.fa")]), k=9, rc=False)
    out = a.nk(full_info=True).decode()
    assert "GATT\tAAGG\tK,D\n" in out


def _must_fail(fn):
    try:
        fn()
    except Exception:
        return
    raise AssertionError("the reference panics here")


def _merge_delete(E, tmp_path, k, del_list):        # shared body of skf_ops.rs:11-83 and :85-161
    t1 = roundtrip(E, E.Array.build(fasta_inputs(E, [fin("test_1.fa")]), k=k), tmp_path, "test_1")
    t2 = roundtrip(E, E.Array.build(fasta_inputs(E, [fin("test_2.fa")]), k=k), tmp_path, "test_2")
    m = roundtrip(E, E.Array.merge([t1, t2]), tmp_path, "merge")
    assert m.names == ["test_1", "test_2"] and m.k == k
    # removing a sample that is not there panics (merge_ska_array.rs:252-254)
    _must_fail(lambda: E.Array.load(os.path.join(str(tmp_path), "merge.skf")).delete_samples(del_list))
    # delete test_2: nk must equal the single-sample build's
    m.delete_samples(["test_2"])
    d = roundtrip(E, m, tmp_path, "merge_delete")
    assert d.nk() == t1.nk()
    assert d.nk(full_info=True).decode().split("\n")[:9] == t1.nk(full_info=True).decode().split("\n")[:9]
    return t1, t2


def case_merge_delete(E, tmp_path):                 # skf_ops.rs:11-83
    t1, t2 = _merge_delete(E, tmp_path, 31, ["test_3"])
    matches_path(E.Array.load(os.path.join(str(tmp_path), "merge.skf")).nk(), correct("merge_nk.stdout"))
    # merging is the same as building both together
    both = E.Array.build(fasta_inputs(E, [fin("test_1.fa"), fin("test_2.fa")]), k=31)
    assert sorted(E.Array.merge([t1, t2]).nk(full_info=True).decode().split("\n")) == sorted(both.nk(full_info=True).decode().split("\n"))
    # delete preconditions (merge_ska_array.rs:232-234)
    _must_fail(lambda: E.Array.merge([t1, t2]).delete_samples([]))
    _must_fail(lambda: E.Array.merge([t1, t2]).delete_samples(["test_1", "test_2"]))
    # merge preconditions (merge_ska_dict.rs:169-174)
    _must_fail(lambda: E.Array.merge([t1, E.Array.build(fasta_inputs(E, [fin("test_2.fa")]), k=17)]))
    _must_fail(lambda: E.Array.merge([t1, E.Array.build(fasta_inputs(E, [fin("test_2.fa")]), k=31, rc=False)]))


def case_merge_delete_u128(E, tmp_path):            # skf_ops.rs:85-161
    with open(fin("missing_delete.txt")) as f:
        names = [ln.split()[0] for ln in f if ln.strip()]
    _merge_delete(E, tmp_path, 41, names)


def case_weed(E, tmp_path):                         # skf_ops.rs:163-290
    def load(name):                                 # every CLI step of the reference test starts from the file
        return E.Array.load(os.path.join(str(tmp_path), name + ".skf"))
    # the weed FASTA's split k-mers go; default min_freq 0.9 -> floor(2 * 0.9) = 1 (generic_modes.rs:249)
    a = E.Array.load(fin("merge.skf"))
    a.weed(fin("weed.fa"))
    roundtrip(E, a, tmp_path, "weeded")
    assert load("weeded").align() == correct("weed_align.stdout")
    # a second pass without a weed file: filter no-const, min_freq 1
    a = load("weeded")
    a.weed(None, min_freq=1.0, filter_type=E.FILTER_NO_CONST)
    matches_path(roundtrip(E, a, tmp_path, "weeded2").nk(full_info=True), correct("weed_nk.stdout"))
    # masking ambiguous sites
    b = E.Array.load(fin("merge_k9.skf"))
    b.weed(None, ambig_mask=True)
    matches_path(roundtrip(E, b, tmp_path, "weed_k9").nk(), correct("weed_nk_k9.stdout"))
    # keep rather than weed
    c = E.Array.load(fin("merge.skf"))
    c.weed(fin("weed.fa"), reverse=True)
    assert roundtrip(E, c, tmp_path, "weed_rev").align() == correct("weed_align_reverse.stdout")
    # longer k-mers (u128)
    d = roundtrip(E, E.Array.build(fasta_inputs(E, [fin("test_1.fa"), fin("test_2.fa")]), k=41), tmp_path, "build_k41")
    d.weed(None, min_freq=1.0, filter_type=E.FILTER_NO_AMBIG_OR_CONST)
    matches_path(roundtrip(E, d, tmp_path, "weed_k41").nk(full_info=True), correct("weed_nk_k41.stdout"))
    # a FASTQ weed file is refused (ska_ref.rs:206-208)
    _must_fail(lambda: E.Array.load(fin("merge.skf")).weed(fin("test_1_fwd.fastq.gz")))


def case_repeats_weed(E, tmp_path):                 # fasta_input.rs:156-220, the weed step
    a = E.Array.build(fasta_inputs(E, [fin("dup_test_1.fa"), fin("dup_test_2.fa")]), k=9, rc=False)
    a = roundtrip(E, a, tmp_path, "dup_ss")
    a.weed(None, filter_type=E.FILTER_NO_CONST, min_freq=1.0)
    matches_path(roundtrip(E, a, tmp_path, "dup_ss_w").nk(full_info=True), correct("dup_ss_nk.stdout"))


# ---------------------------------------------------------------- tests/map.rs, fasta_input.rs map_n
def case_map_aln(E, tmp_path):                      # map.rs:11-93
    a = E.Array.build(fasta_inputs(E, [fin("test_1.fa"), fin("test_2.fa")]), k=31)          # `ska map ref a.fa b.fa`: build defaults (io_utils.rs:76-92)
    assert a.map(fin("test_ref.fa")).startswith(b">test_1\n")
    assert E.Array.load(fin("merge.skf")).map(fin("test_ref.fa")) == correct("map_aln.stdout")
    assert E.Array.load(fin("merge_k9.skf")).map(fin("test_ref.fa")) == correct("map_aln_k9.stdout")
    assert E.Array.load(fin("merge_k9.skf")).map(fin("test_ref.fa"), ambig_mask=True) == correct("map_aln_k9_filter.stdout")
    assert E.Array.load(fin("merge.skf")).map(fin("test_ref_two_chrom.fa")) == correct("map_aln_two_chrom.stdout")
    b = E.Array.build(fasta_inputs(E, [fin("test_1.fa"), fin("indel_test.fa")]), k=31)
    assert b.map(fin("test_ref.fa")) == correct("map_aln_indels.stdout")
    c = roundtrip(E, E.Array.build(fasta_inputs(E, [fin("ambig_test_1.fa"), fin("ambig_test_2.fa")]), k=17, rc=False), tmp_path, "ambig_map")
    assert c.map(fin("ambig_test_ref.fa")) == correct("map_aln_ambig.stdout")


def case_map_u128(E, tmp_path):                     # map.rs:94-117
    assert E.Array.load(fin("merge_k41.skf")).map(fin("test_ref.fa")) == correct("map_aln_k41.stdout")
    matches_path(E.Array.load(fin("merge_k41.skf")).map(fin("test_ref.fa"), fmt="vcf"), correct("map_vcf_k41.stdout"))


def case_map_vcf(E, tmp_path):                      # map.rs:119-168
    matches_path(E.Array.load(fin("merge.skf")).map(fin("test_ref.fa"), fmt="vcf"), correct("map_vcf.stdout"))
    matches_path(E.Array.load(fin("merge.skf")).map(fin("test_ref_two_chrom.fa"), fmt="vcf"), correct("map_vcf_two_chrom.stdout"))
    b = E.Array.build(fasta_inputs(E, [fin("test_1.fa"), fin("indel_test.fa")]), k=31)
    matches_path(b.map(fin("test_ref.fa"), fmt="vcf"), correct("map_vcf_indels.stdout"))


def case_map_rev_comp(E, tmp_path):                 # map.rs:170-237
    rcb = roundtrip(E, E.Array.build(fasta_inputs(E, [fin("test_1.fa"), fin("test_2_rc.fa")]), k=9), tmp_path, "rc_build")
    fwd = E.Array.load(fin("merge_k9.skf"))
    # (the reference test maps a file that does not exist and so compares nothing; the two maps agree except at the
    #  sequence ends, where the idx + k >= len rule bites on the other strand)
    strip = lambda t: [ln[12:70] for ln in t.decode().splitlines() if not ln.startswith(">")]
    assert strip(rcb.map(fin("test_ref.fa"))) == strip(fwd.map(fin("test_ref.fa")))
    ss = roundtrip(E, E.Array.build(fasta_inputs(E, [fin("test_1.fa"), fin("test_2_rc.fa")]), k=9, rc=False), tmp_path, "ss_map")
    assert ss.map(fin("test_ref.fa")) == correct("map_ss.stdout")
    matches_path(ss.map(fin("test_ref.fa"), fmt="vcf"), correct("map_vcf_ss.stdout"))


def case_map_repeat_mask(E, tmp_path):              # map.rs:239-320
    k9 = lambda: E.Array.load(fin("merge_k9.skf"))
    assert k9().map(fin("test_ref.fa"), repeat_mask=True) == correct("map_aln_k9.masked.stdout")
    matches_path(k9().map(fin("test_ref.fa"), fmt="vcf", repeat_mask=True), correct("map_vcf_k9.masked.stdout"))
    assert k9().map(fin("test_ref_two_chrom.fa"), repeat_mask=True) == correct("map_all_repeats.masked.stdout")
    assert k9().map(fin("test_ref_two_chrom_repeats.fa"), repeat_mask=True) == correct("map_aln_two_chrom.masked.stdout")
    matches_path(k9().map(fin("test_ref_two_chrom_repeats.fa"), fmt="vcf", repeat_mask=True), correct("map_vcf_two_chrom.masked.stdout"))


def case_map_n(E, tmp_path):                        # fasta_input.rs:34-58
    a = roundtrip(E, E.Array.build(fasta_inputs(E, [fin("N_test_1.fa"), fin("N_test_2.fa")]), k=11), tmp_path, "N_test")
    assert a.map(fin("test_ref.fa")) == correct("map_N.stdout")


# ---------------------------------------------------------------- coverage.rs unit test + fastq_input.rs cov_check
COV_EXAMPLE = [44633459, 950672, 104410, 44137, 24170, 21232, 21699, 24145, 30696, 39210, 49878, 63683, 77690, 95147, 112416, 130307,
               146531, 160932, 175130, 185113, 193149, 197468, 199189, 198235, 192150, 185565, 176362, 165455, 152487, 139495, 127036,
               112803, 103080, 90425, 80637, 70960, 62698, 54949, 46744, 41240, 35591, 30025, 25856, 22105, 19405, 16668, 14780, 12620,
               11074, 9807, 8517, 7731, 7112, 6846, 6126, 5696, 5233, 4779, 4288, 3873, 3519, 3406, 2994, 2859, 2650, 2394, 2376, 2260,
               2233, 2050, 1859, 1863, 1792, 1777, 1773, 1738, 1648]


def case_cov(E, tmp_path):
    w0, c, cutoff = E.cov_fit(COV_EXAMPLE)                                      # coverage.rs:369-385: the reference's known answer
    assert cutoff == 9 and 0.9 < w0 < 0.93 and 25.5 < c < 26.5
    text, cut = E.cov(fin("test_1_fwd.fastq.gz"), fin("test_1_rev.fastq.gz"), k=9)       # fastq_input.rs:474-510: runs
    assert text.startswith(b"Count\tK_mers\tMixture_density\tComponent\n") and cut >= 1
    text, cut = E.cov(fin("test_long_1_fwd.fastq.gz"), fin("test_long_1_rev.fastq.gz"), k=33)
    assert text.startswith(b"Count\t")
    _must_fail(lambda: E.cov(fin("test_1.fa"), fin("test_2.fa"), k=9))              # FASTA is refused


ALL_CASES = [v for k, v in sorted(globals().items()) if k.startswith("case_")]
