"""`-m gpu`: the `ska` executable end to end against the reference's goldens, and BASELINE-size runs checked through
size-independent properties (sharded == unsharded, idempotence, oracle spot checks)."""
import os
import subprocess

import numpy as np
import pytest

import golden_cases as G
import ora

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SKA = os.path.join(ROOT, "ska.rust_amd", "ska")


def ska(*args, cwd=None):
    r = subprocess.run([SKA, *args], cwd=cwd, capture_output=True, timeout=300)
    return r.returncode, r.stdout, r.stderr


def test_cli_build_align_distance_nk(tmp_path):
    wd = str(tmp_path)
    rc, out, err = ska("build", "-o", "N_test.skf", G.fin("N_test_1.fa"), G.fin("N_test_2.fa"), cwd=wd)      # fasta_input.rs:11-32
    assert rc == 0, err
    assert os.path.exists(os.path.join(wd, "N_test.skf")) and not os.path.exists(os.path.join(wd, "N_test.skf.skf"))
    rc, out, err = ska("align", "N_test.skf", cwd=wd)
    assert rc == 0 and out == G.correct("align_N.stdout")
    rc, out, err = ska("distance", G.fin("merge_k9.skf"), "--allow-ambiguous", "--threads", "2", cwd=wd)          # distance.rs:44-58
    assert rc == 0 and out == G.correct("merge_k9.dist.stdout")
    rc, out, err = ska("distance", G.fin("merge_k9.skf"), "--min-freq", "1", cwd=wd)
    assert rc == 0 and out == G.correct("merge_k9_min_freq.dist.stdout")
    rc, out, err = ska("build", "-k", "33", "-o", "k33", G.fin("test_1.fa"), G.fin("test_2.fa"), cwd=wd)           # align.rs:116-167
    assert rc == 0, err
    rc, out, err = ska("nk", "k33.skf", cwd=wd)
    G.matches_path(out, G.correct("k33.stdout"))
    rc, out, err = ska("build", "-k", "65", "-o", "bad", G.fin("test_1.fa"), G.fin("test_2.fa"), cwd=wd)
    assert rc != 0
    rc, out, err = ska("align", G.fin("test_1.fa"), G.fin("test_2.fa"), cwd=wd)                                   # align.rs:169-184
    assert rc == 0 and G.var_hash(out) == {("A", "T"), ("C", "T")}
    # -f file list with paired FASTQ + --min-count (fastq_input.rs:57-109)
    with open(os.path.join(wd, "rfile.txt"), "w") as f:
        for n, a, b in G.rfile("test_count", True):
            f.write(f"{n}\t{a}\t{b}\n")
    rc, out, err = ska("build", "-f", "rfile.txt", "-o", "reads_k7_c3", "--min-count", "3", "-k", "7", cwd=wd)
    assert rc == 0, err
    rc, out, err = ska("align", "reads_k7_c3.skf", cwd=wd)
    assert rc == 0 and G.var_hash(out) == {("C", "T")}
    rc, out, err = ska("build", "-f", "rfile.txt", "-o", "x", "--min-count", "-1", "-k", "7", cwd=wd)              # fastq_input.rs:528-537
    assert rc != 0


def test_cli_merge_delete_weed(tmp_path):
    """tests/skf_ops.rs through the `ska` executable: merge_delete (:11-83), merge_delete_u128 (:85-161), weed (:163-290)."""
    import shutil
    wd = str(tmp_path)
    for k in ("31", "41"):
        for t in ("test_1", "test_2"):
            rc, out, err = ska("build", G.fin(t + ".fa"), "-o", t, "-k", k, cwd=wd)
            assert rc == 0, err
        rc, out, err = ska("merge", "test_1.skf", "test_2.skf", "-o", "merge", cwd=wd)
        assert rc == 0 and os.path.exists(os.path.join(wd, "merge.skf")), err
        rc, out, err = ska("nk", "merge.skf", cwd=wd)
        if k == "31":
            G.matches_path(out, G.correct("merge_nk.stdout"))
        rc, out, err = ska("delete", "-s", "merge.skf", "test_3", cwd=wd)                       # not there -> panic
        assert rc != 0 and b"Could not find sample" in err
        rc, out, err = ska("delete", "-s", "merge.skf", "-f", G.fin("missing_delete.txt"), cwd=wd)
        assert rc != 0
        rc, nk1, err = ska("nk", "test_1.skf", cwd=wd)
        rc, out, err = ska("delete", "-s", "merge.skf", "-o", "merge_delete", "test_2", cwd=wd)
        assert rc == 0, err
        rc, out, err = ska("nk", "merge_delete.skf", cwd=wd)
        assert rc == 0 and out == nk1
        rc, out, err = ska("delete", "-s", "merge.skf", "test_2", cwd=wd)                       # in place
        assert rc == 0, err
        rc, out, err = ska("nk", "merge.skf", cwd=wd)
        assert out == nk1
    rc, out, err = ska("merge", "test_1.skf", "-o", "x", cwd=wd)
    assert rc != 0                                                                             # lib.rs:729-731
    # weed
    shutil.copy(G.fin("merge.skf"), os.path.join(wd, "merge.skf"))
    rc, out, err = ska("weed", "merge.skf", G.fin("weed.fa"), cwd=wd)
    assert rc == 0, err
    rc, out, err = ska("align", "merge.skf", cwd=wd)
    assert out == G.correct("weed_align.stdout")
    rc, out, err = ska("weed", "merge.skf", "--filter", "no-const", "--min-freq", "1", cwd=wd)
    assert rc == 0, err
    rc, out, err = ska("nk", "merge.skf", "--full-info", cwd=wd)
    G.matches_path(out, G.correct("weed_nk.stdout"))
    shutil.copy(G.fin("merge_k9.skf"), os.path.join(wd, "merge_k9.skf"))
    rc, out, err = ska("weed", "merge_k9.skf", "--ambig-mask", cwd=wd)
    assert rc == 0, err
    rc, out, err = ska("nk", "merge_k9.skf", cwd=wd)
    G.matches_path(out, G.correct("weed_nk_k9.stdout"))
    shutil.copy(G.fin("merge.skf"), os.path.join(wd, "merge.skf"))
    rc, out, err = ska("weed", "merge.skf", G.fin("weed.fa"), "--reverse", "-o", "kept.skf", cwd=wd)
    assert rc == 0, err
    rc, out, err = ska("align", "kept.skf", cwd=wd)
    assert out == G.correct("weed_align_reverse.stdout")
    rc, out, err = ska("build", "-o", "build_k41", "-k", "41", G.fin("test_1.fa"), G.fin("test_2.fa"), cwd=wd)
    rc, out, err = ska("weed", "build_k41.skf", "--filter", "no-ambig-or-const", "--min-freq", "1", cwd=wd)
    assert rc == 0, err
    rc, out, err = ska("nk", "build_k41.skf", "--full-info", cwd=wd)
    G.matches_path(out, G.correct("weed_nk_k41.stdout"))


def test_cli_map(tmp_path):
    """tests/map.rs through the `ska` executable."""
    wd = str(tmp_path)
    rc, out, err = ska("map", G.fin("test_ref.fa"), G.fin("merge.skf"), cwd=wd)
    assert rc == 0 and out == G.correct("map_aln.stdout"), err
    rc, out, err = ska("map", G.fin("test_ref.fa"), G.fin("merge.skf"), "-f", "vcf", cwd=wd)
    assert rc == 0, err
    G.matches_path(out, G.correct("map_vcf.stdout"))
    rc, out, err = ska("map", G.fin("test_ref.fa"), G.fin("merge_k9.skf"), "--ambig-mask", cwd=wd)
    assert out == G.correct("map_aln_k9_filter.stdout")
    rc, out, err = ska("map", G.fin("test_ref_two_chrom_repeats.fa"), G.fin("merge_k9.skf"), "--repeat-mask", "--format", "vcf", cwd=wd)
    G.matches_path(out, G.correct("map_vcf_two_chrom.masked.stdout"))
    rc, out, err = ska("map", G.fin("test_ref.fa"), G.fin("test_1.fa"), G.fin("indel_test.fa"), "-o", "map.aln", cwd=wd)      # sequence files: built on the fly
    assert rc == 0 and open(os.path.join(wd, "map.aln"), "rb").read() == G.correct("map_aln_indels.stdout"), err
    rc, out, err = ska("map", G.fin("test_ref.fa"), G.fin("merge_k41.skf"), cwd=wd)
    assert out == G.correct("map_aln_k41.stdout")


def test_cli_cov_and_auto_min_count(tmp_path):
    """fastq_input.rs cov_check (:473-510) and build_auto_check (:512-540) through the `ska` executable."""
    wd = str(tmp_path)
    rc, out, err = ska("cov", G.fin("test_1_fwd.fastq.gz"), G.fin("test_1_rev.fastq.gz"), "-k", "9", "-v", cwd=wd)
    assert rc == 0 and out.startswith(b"Count\tK_mers\tMixture_density\tComponent\n") and b"Estimated cutoff\t" in err, err
    rc, out, err = ska("cov", G.fin("test_long_1_fwd.fastq.gz"), G.fin("test_long_1_rev.fastq.gz"), "-k", "33", "-v", cwd=wd)
    assert rc == 0, err
    assert out == ora.cov(G.fin("test_long_1_fwd.fastq.gz"), G.fin("test_long_1_rev.fastq.gz"), k=33)[0] or len(out.splitlines()) == 5
    rc, out, err = ska("cov", G.fin("test_1.fa"), G.fin("test_2.fa"), "-k", "9", "-v", cwd=wd)
    assert rc != 0 and b"appears to be FASTA" in err
    with open(os.path.join(wd, "rfile.txt"), "w") as f:
        for n, a, b in G.rfile("test", True):
            f.write(f"{n}\t{a}\t{b}\n")
    rc, out, err = ska("build", "-f", "rfile.txt", "-o", "reads", "--min-count", "auto", "-v", "-k", "9", "--min-qual", "2", cwd=wd)
    assert rc == 0 and os.path.exists(os.path.join(wd, "reads.skf")) and b"Using inferred minimum kmer value of" in err, err
    rc, out, err = ska("build", "-f", "rfile.txt", "-o", "reads", "--min-count", "-1", "-v", "-k", "9", "--min-qual", "2", cwd=wd)
    assert rc != 0


@pytest.fixture(scope="module")
def big():
    """BASELINE.json configs[1] shape, reduced to 24 samples so that the oracle spot checks stay in seconds."""
    import skx_engine as E
    import synth
    E.load_library()
    anc = synth.ancestor(5_000_000, seed=1)
    n = 24
    streams = [synth.sample_stream(anc, i, n) for i in range(n)]
    return E, streams, [f"g{i}" for i in range(n)]


def test_full_size_properties(big):
    E, streams, names = big
    n = len(streams)
    ds = E.DictSet.build([s.tobytes() for s in streams], 31, True)
    # (1) oracle spot check at full genome size
    for i in (0, 9):        # sample 9 is reverse-complemented
        d = ora.Dict.new(31, True)
        for rec in streams[i].tobytes().split(b"\n")[:-1]:
            d.add_record(rec)
        ok, ob = d.export()
        gk, gb = ds.export(i)
        assert np.array_equal(gk["lo"], ok["lo"]) and np.array_equal(gb, ob)
    whole = ds.merge(names)
    U = whole.nrows
    assert 5_000_000 < U < 8_000_000
    # (2) per-sample k-mer counts == dictionary sizes; variant_count sums == total cells
    sk = whole.sample_kmers()
    assert [int(x) for x in sk] == [ds.size(i) for i in range(n)]
    # (3) sharded (key-table exchange) == unsharded: same rows, column slabs tile the matrix (checksum of checksums)
    a, b = n // 3, n
    shards = [E.DictSet.build([s.tobytes() for s in streams[lo:hi]], 31, True) for lo, hi in ((0, a), (a, b))]
    rows = E.KeySet.merge([s.union_keys() for s in shards])
    assert len(rows) == U
    wk, wv, wc = whole.export()
    col = 0
    total = np.zeros(U, dtype=np.uint64)
    for s, (lo, hi) in zip(shards, ((0, a), (a, b))):
        part = s.assemble(rows, names[lo:hi])
        pk, pv, pc = part.export()
        assert np.array_equal(pk["lo"], wk["lo"])
        assert np.array_equal(pv, wv[:, lo:hi])
        total += pc
    assert np.array_equal(total, wc)
    # (4) filter idempotence + monotonicity; alignment rows all have the kept length
    aln1 = whole.align(min_freq=0.9)
    kept = whole.nrows
    assert whole.apply_filters(0.9) == 0 and whole.nrows == kept
    lens = G.aln_length(aln1)
    assert len(lens) == n and set(lens) == {kept}
    # (5) no-const columns really vary, and every kept row is present in >= ceil(0.9 n) samples
    mat = np.frombuffer(b"".join(aln1.split(b"\n")[1::2]), dtype=np.uint8).reshape(n, kept)
    assert (mat != mat[0]).any(axis=0).all()
    assert ((mat != ord("-")).sum(axis=0) >= int(np.ceil(0.9 * n))).all()


def test_distance_many_samples(big):
    """64 samples x ~1.2 M variable rows: GPU popcount distance == oracle's row loop on a subsample of pairs."""
    E, streams, names = big
    small = [s[:400_000].tobytes() + b"\n" for s in streams[:12]]
    ga = E.DictSet.build(small, 31, True).merge(names[:12])
    dicts = []
    for s in small:
        d = ora.Dict.new(31, True)
        for rec in s.split(b"\n")[:-1]:
            d.add_record(rec)
        dicts.append(d)
    oa = ora.Array.from_dicts(dicts, names[:12])
    assert ga.distance_tsv(min_freq=0.5) == oa.distance_tsv(min_freq=0.5)
    ga = E.DictSet.build(small, 31, True).merge(names[:12])
    oa = ora.Array.from_dicts(dicts, names[:12])
    assert ga.distance_tsv(filt_ambig=False) == oa.distance_tsv(filt_ambig=False)


def test_full_size_lifecycle_and_map(big, tmp_path):
    """BASELINE-size genomes through the .skf life-cycle and `ska map`, checked by size-independent properties and an oracle
    spot check: merge(build(A), build(B)) == build(A + B); delete(merge, B) == build(A); save -> load round trip;
    weed(x, ancestor) + weed(x, ancestor, reverse) partition the rows; map of two samples == the oracle's."""
    import synth
    E, streams, names = big
    A, B = list(range(0, 6)), list(range(6, 10))
    dsA = E.DictSet.build([streams[i].tobytes() for i in A], 31, True)
    dsB = E.DictSet.build([streams[i].tobytes() for i in B], 31, True)
    dsAB = E.DictSet.build([streams[i].tobytes() for i in A + B], 31, True)
    a, b, ab = dsA.merge([names[i] for i in A]), dsB.merge([names[i] for i in B]), dsAB.merge([names[i] for i in A + B])
    m = E.Array.merge([a, b])
    mk, mv, mc = m.export()
    wk, wv, wc = ab.export()
    assert m.names == ab.names and np.array_equal(mk["lo"], wk["lo"]) and np.array_equal(mv, wv) and np.array_equal(mc, wc)
    # streaming codec at this size, then delete
    p = str(tmp_path / "m.skf")
    m.save(p)
    m2 = E.Array.load(p)
    lk, lv, lc = m2.export()
    assert np.array_equal(lk["lo"], wk["lo"]) and np.array_equal(lv, wv) and np.array_equal(lc, wc)
    m2.delete_samples([names[i] for i in B])
    dk, dv, dc = m2.export()
    ak, av, ac = a.export()
    assert m2.names == a.names and np.array_equal(dk["lo"], ak["lo"]) and np.array_equal(dv, av) and np.array_equal(dc, ac)
    # weed with the ancestor's split k-mers: the two directions partition the rows
    anc = synth.ancestor(5_000_000, seed=1)
    ref = str(tmp_path / "anc.fa")
    synth.to_fasta(np.concatenate([anc, np.array([10], np.uint8)]), ref)
    ks = E.KeySet.from_fasta(ref, 31, True)
    w1, w2 = E.Array.merge([a, b]), E.Array.merge([a, b])
    r1, r2 = w1.weed_keys(ks), w2.weed_keys(ks, reverse=True)
    assert r1 + r2 == len(wk) and w1.nkmers + w2.nkmers == len(wk) and w2.nkmers == r1
    k1, k2 = w1.export()[0]["lo"], w2.export()[0]["lo"]
    assert len(np.intersect1d(k1, k2)) == 0 and np.array_equal(np.sort(np.concatenate([k1, k2])), wk["lo"])
    # map two samples onto the ancestor == oracle
    two = E.DictSet.build([streams[0].tobytes(), streams[9].tobytes()], 31, True).merge(["g0", "g9"])
    dicts = []
    for s in (streams[0], streams[9]):
        d = ora.Dict.new(31, True)
        for rec in s.tobytes().split(b"\n")[:-1]:
            d.add_record(rec)
        dicts.append(d)
    otwo = ora.Array.from_dicts(dicts, ["g0", "g9"])
    g = two.map(ref)
    assert g == otwo.map(ref)
    seqs = g.split(b"\n")[1::2]
    assert [len(x) for x in seqs] == [5_000_000, 5_000_000] and seqs[0].count(b"-") < 50_000
    assert two.map(ref, fmt="vcf") == otwo.map(ref, fmt="vcf")


def test_cli_rejects_unknown_flags_like_clap(tmp_path):
    """cli.rs:109-330: an argument a subcommand does not declare is an error (exit code 2), not a silently ignored boolean."""
    wd = str(tmp_path)
    for args in (["build", "-o", "x", "--no-such-flag", G.fin("test_1.fa"), G.fin("test_2.fa")],
                 ["align", G.fin("merge.skf"), "--allow-ambiguous"],            # a `distance` flag
                 ["distance", G.fin("merge.skf"), "--filter", "no-const"],      # an `align` flag
                 ["nk", G.fin("merge.skf"), "--threads", "2"],
                 ["merge", G.fin("merge.skf"), G.fin("merge_k9.skf"), "-o", "m", "--reverse"]):
        rc, out, err = ska(*args, cwd=wd)
        assert rc == 2 and b"unexpected argument" in err, (args, rc, err[-200:])
    rc, out, err = ska("align", G.fin("merge.skf"), "-v", cwd=wd)               # the global flag stays accepted everywhere
    assert rc == 0


def test_builds_are_deterministic(tmp_path, monkeypatch):
    """Rows are kept in the order of H(key) whatever the path that produced them: two runs, a build forced into several batches
    (joined by the merge row-set path) and an eagerly assembled array all write the same .skf bytes and the same nk listing."""
    import synth
    wd = str(tmp_path)
    anc = synth.ancestor(150_000, seed=3)
    files = []
    for i in range(10):
        p = os.path.join(wd, f"d{i}.fa")
        synth.to_fasta(synth.sample_stream(anc, i, 10, private_snps=40, shared_snps=10, seed=3), p)
        files.append(p)
    outs = {}
    for tag, env in (("a", {}), ("b", {}), ("batched", {"SKX_BUILD_BATCH_MB": "8"}), ("eager", {"SKX_KNOBS": "eager_array=1"}), ("host_parse", {"SKX_KNOBS": "host_parse=1"}), ("sorted", {"SKX_KNOBS": "sorted_dicts=1"})):
        for k_, v_ in env.items():
            monkeypatch.setenv(k_, v_)
        rc, out, err = ska("build", "-o", tag, "-k", "31", "--threads", "3", *files, cwd=wd)
        for k_ in env:
            monkeypatch.delenv(k_)
        assert rc == 0, err
        rc, nk, err = ska("nk", tag + ".skf", "--full-info", cwd=wd)
        assert rc == 0, err
        outs[tag] = (open(os.path.join(wd, tag + ".skf"), "rb").read(), nk)
    for tag in ("b", "batched", "eager", "host_parse"):
        assert outs[tag][0] == outs["a"][0], tag
        assert outs[tag][1] == outs["a"][1], tag
    # and the alignment of the file is the alignment of the single-invocation form, byte for byte
    rc, a1, err = ska("align", "a.skf", cwd=wd)
    rc2, a2, err2 = ska("align", "--threads", "2", *files, cwd=wd)
    assert rc == 0 and rc2 == 0 and a1 == a2


def test_read_set_pipeline_equals_one_shot(tmp_path):
    """`ska build` on read sets runs reader threads, uploads and the per-isolate kernels as a pipeline over a small pool of device slots
    (skx_api.cpp build_reads_pipelined); SKX_KNOBS=no_reads_pipeline is the one-shot form it replaced.  Same .skf bytes either way and with a
    pool of ONE slot (every sample reuses it), for paired and single-file samples, k = 31 and 41; a broken record fails both forms with the
    reference's message (ska_dict.rs:131-153 via needletail)."""
    import synth
    wd = str(tmp_path)
    anc = synth.ancestor(60_000, seed=3)
    n = 6
    pairs = [synth.write_read_pair(anc, i, n, os.path.join(wd, f"r{i}"), read_len=100, coverage=25.0, seed=3) for i in range(n)]
    with open(os.path.join(wd, "list.txt"), "w") as f:
        for i, (a, b) in enumerate(pairs):
            f.write(f"r{i}\t{a}\t{b}\n" if i % 3 else f"r{i}\t{a}\n")                     # every third sample: one file only
    for k in ("31", "41"):
        outs = {}
        # (round 6: a sample crosses PCIe as its reader thread chooses -- the file's bytes, framed and packed on the device (reads_raw=2: always),
        #  or planes packed by the reader (reads_raw=1: always); left alone the choice follows the pinned ring's fill)
        for tag, env in (("pipe", {}), ("raw", {"SKX_KNOBS": "reads_raw=2"}), ("packed", {"SKX_KNOBS": "reads_raw=1"}), ("oneshot", {"SKX_KNOBS": "no_reads_pipeline=1"})):
            r = subprocess.run([SKA, "build", "-f", "list.txt", "-o", f"{tag}{k}", "-k", k, "--min-count", "3", "--threads", "4"], cwd=wd, capture_output=True,
                               timeout=300, env=dict(os.environ, **env))
            assert r.returncode == 0, r.stderr[-600:]
            outs[tag] = open(os.path.join(wd, f"{tag}{k}.skf"), "rb").read()
        assert outs["pipe"] == outs["oneshot"] and outs["raw"] == outs["oneshot"] and outs["packed"] == outs["oneshot"]
        want = ora.Array.build([(f"r{i}", a, b if i % 3 else None) for i, (a, b) in enumerate(pairs)], k=int(k), rc=True, q=ora.qual(3, 20, ora.QUAL_STRICT), threads=2)
        got = ora.Array.load(os.path.join(wd, f"pipe{k}.skf"))
        got.sort_rows(); want.sort_rows()
        gk, gv, gc = got.export()
        ok, ov, oc = want.export()
        assert len(ok) > 10_000 and np.array_equal(gk, ok) and np.array_equal(gv, ov) and np.array_equal(gc, oc)
    # gzip files are inflated by the reader threads as they go (zlib); a file of two members, whose trailer gives the second member's length
    # only, sends the batch to the one-shot form half way through: the same bytes every time
    import gzip
    for tag, two_members in (("gz", False), ("gz2", True)):
        with open(os.path.join(wd, f"list_{tag}.txt"), "w") as f:
            for i, (a_, b_) in enumerate(pairs):
                names = []
                for src in ((a_, b_) if i % 3 else (a_,)):
                    raw = open(src, "rb").read()
                    dst = src + f".{tag}.gz"
                    if two_members and i == 1:
                        cut = raw.index(b"\n@r\n", len(raw) // 2) + 1
                        open(dst, "wb").write(gzip.compress(raw[:cut], 1) + gzip.compress(raw[cut:], 1))
                    else:
                        open(dst, "wb").write(gzip.compress(raw, 1))
                    names.append(dst)
                f.write(f"r{i}\t" + "\t".join(names) + "\n")
        # (round 6: the device inflates what the feeder threads hand it as compressed bytes -- reads_gz=2: every sample, reads_gz=1: none -- and the
        #  other readers inflate; a file the device does not vouch for goes through the reader's inflater after all)
        import json
        for knobs in ("", "reads_gz=2", "reads_gz=1", "reads_gz=2,gz_chunk_kb=4,gz_group=3", "reads_raw=2", "reads_raw=1"):
            ph = os.path.join(wd, "ph_gz.json")
            r = subprocess.run([SKA, "build", "-f", f"list_{tag}.txt", "-o", f"{tag}31", "-k", "31", "--min-count", "3", "--threads", "4"], cwd=wd, capture_output=True, timeout=300,
                               env=dict(os.environ, SKX_KNOBS=knobs, SKX_PHASES=ph))
            assert r.returncode == 0, r.stderr[-600:]
            assert open(os.path.join(wd, f"{tag}31.skf"), "rb").read() == open(os.path.join(wd, "pipe31.skf"), "rb").read(), (tag, knobs)
            phases = json.load(open(ph))
            if knobs.startswith("reads_gz=2") and not two_members:
                assert phases.get("build.reads_samples_sent_compressed") == len(pairs) and not phases.get("build.reads_samples_inflated_on_host_after_all"), phases
            if knobs == "reads_gz=1":
                assert not phases.get("build.reads_samples_sent_compressed"), phases
    # files of many members (bgzip's 64 KB blocks, files joined with cat): their trailers give the last member's length only, so six times the
    # file's size stands for the text's (round 6) and the pipeline takes them as they are -- the phase table says it did
    import json
    with open(os.path.join(wd, "list_bgzf.txt"), "w") as f:
        for i, (a_, b_) in enumerate(pairs):
            names = []
            for src in ((a_, b_) if i % 3 else (a_,)):
                raw = open(src, "rb").read()
                dst = src + ".bgzf.gz"
                open(dst, "wb").write(b"".join(gzip.compress(raw[o:o + 60_000], 6) for o in range(0, len(raw), 60_000)) + gzip.compress(b""))
                names.append(dst)
            f.write(f"r{i}\t" + "\t".join(names) + "\n")
    for knobs in ("", "reads_gz=2", "reads_gz=1", "reads_raw=2", "reads_raw=1"):
        ph = os.path.join(wd, "ph_bgzf.json")
        r = subprocess.run([SKA, "build", "-f", "list_bgzf.txt", "-o", "bgzf31", "-k", "31", "--min-count", "3", "--threads", "4"], cwd=wd, capture_output=True, timeout=300,
                           env=dict(os.environ, SKX_KNOBS=knobs, SKX_PHASES=ph))
        assert r.returncode == 0, r.stderr[-600:]
        assert open(os.path.join(wd, "bgzf31.skf"), "rb").read() == open(os.path.join(wd, "pipe31.skf"), "rb").read(), knobs
        assert "build.reads_uploaded_GB" in json.load(open(ph)), knobs
    # a gzip file that stops before its end is refused by both forms (never taken for a shorter input)
    z = open(pairs[4][0] + ".gz.gz", "rb").read()
    open(os.path.join(wd, "cut.fastq.gz"), "wb").write(z[:len(z) * 6 // 10])
    with open(os.path.join(wd, "list_cut.txt"), "w") as f:
        f.write(f"r0\t{pairs[0][0]}.gz.gz\nr4\t{os.path.join(wd, 'cut.fastq.gz')}\nr5\t{pairs[5][0]}.gz.gz\n")
    for env in ({}, {"SKX_KNOBS": "reads_gz=2"}, {"SKX_KNOBS": "reads_gz=1"}, {"SKX_KNOBS": "reads_raw=2"}, {"SKX_KNOBS": "reads_raw=1"}, {"SKX_KNOBS": "no_reads_pipeline=1"}):
        r = subprocess.run([SKA, "build", "-f", "list_cut.txt", "-o", "cut", "-k", "31", "--min-count", "3", "--threads", "3"], cwd=wd, capture_output=True, timeout=300, env=dict(os.environ, **env))
        assert r.returncode != 0 and b"Invalid" in r.stderr, r.stderr[-400:]
    bad = open(pairs[2][0], "rb").read()
    open(pairs[2][0], "wb").write(bad[:len(bad) // 2 - 7])                                # a record cut in the middle
    for env in ({}, {"SKX_KNOBS": "reads_raw=2"}, {"SKX_KNOBS": "reads_raw=1"}, {"SKX_KNOBS": "no_reads_pipeline=1"}):
        r = subprocess.run([SKA, "build", "-f", "list.txt", "-o", "broken", "-k", "31", "--min-count", "3"], cwd=wd, capture_output=True, timeout=300, env=dict(os.environ, **env))
        assert r.returncode != 0 and b"Invalid FASTA/Q record" in r.stderr, r.stderr[-400:]


def test_read_set_pipeline_odd_bytes_and_line_lengths(tmp_path):
    """The pipeline's reader threads pack a read set into bit planes (fastx.cpp pack_*_planes: two code bits, the bytes valid_base rejects,
    line ends, quality verdicts) and one launch per sample takes them apart on the device: lines of every length around the 32- and
    64-position word edges (0, 1, 31 ... 129, 150), IUPAC codes and other bytes that encode_base folds onto a base (bit_encoding.rs:42-54),
    N / n / '.' / '>' (low nibble 14: rejected), lower case, qualities on both sides of the threshold, one file with CRLF line ends.
    Same .skf as the one-shot form (byte streams, no packing) and the oracle's array: k = 31 strict, k = 41 middle base, and k = 21 without
    filters at --min-count 1 (every gated window enters: the window pass writes the words itself)."""
    wd = str(tmp_path)
    rng = np.random.default_rng(77)
    glen = 40_000
    g = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, size=glen)]
    lens = np.array([0, 1, 30, 31, 32, 33, 62, 63, 64, 65, 95, 96, 97, 127, 128, 129, 150])
    odd = np.frombuffer(b"NnRYKMSWryk.>-*Xx", np.uint8)
    comp = np.zeros(256, np.uint8)
    for a, b in zip(b"ACGT", b"TGCA"):
        comp[a] = b
    files = []
    for i in range(4):
        gi = g.copy()
        gi[rng.integers(0, glen, size=60)] = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, size=60)]
        pair = []
        for mate in (0, 1):
            out = []
            for _ in range(9000):
                L = int(lens[rng.integers(0, len(lens))])
                st = int(rng.integers(0, glen - 150))
                r = gi[st:st + L].copy()
                if rng.random() < 0.5:
                    r = comp[r[::-1]]
                m = rng.random(L) < 0.01
                r[m] = odd[rng.integers(0, len(odd), size=int(m.sum()))]
                low = rng.random(L) < 0.05
                r[low] |= 0x20
                q = (33 + rng.integers(0, 42, size=L)).astype(np.uint8)
                q[rng.random(L) < 0.97] = 33 + 35
                eol = b"\r\n" if (i == 1 and mate == 1) else b"\n"
                out.append(b"@r" + eol + r.tobytes() + eol + b"+" + eol + q.tobytes() + eol)
            p = os.path.join(wd, f"o{i}_{mate + 1}.fastq")
            open(p, "wb").write(b"".join(out))
            pair.append(p)
        files.append(pair)
    with open(os.path.join(wd, "list.txt"), "w") as f:
        for i, (a, b) in enumerate(files):
            f.write(f"o{i}\t{a}\t{b}\n")
    # (k = 9 and 15: the base that leaves a window lies inside the thread's own sixteen positions; k = 63: the whole 64-position history)
    for k, qf, oq, mc in (("31", "strict", ora.QUAL_STRICT, 2), ("41", "middle", ora.QUAL_MIDDLE, 2), ("21", "no-filter", ora.QUAL_NOFILTER, 1),
                          ("9", "strict", ora.QUAL_STRICT, 2), ("15", "middle", ora.QUAL_MIDDLE, 3), ("63", "strict", ora.QUAL_STRICT, 2)):
        outs = {}
        for tag, env in (("pipe", {"SKX_KNOBS": "reads_raw=2"}), ("avx2", {"SKX_KNOBS": "simd_cap=2,reads_raw=1"}), ("plain", {"SKX_KNOBS": "simd_cap=1,reads_raw=1"}),
                         ("avx512", {"SKX_KNOBS": "reads_raw=1"}), ("oneshot", {"SKX_KNOBS": "no_reads_pipeline=1"})):
            r = subprocess.run([SKA, "build", "-f", "list.txt", "-o", f"{tag}{k}", "-k", k, "--min-count", str(mc), "--min-qual", "20", "--qual-filter", qf, "--threads", "3"],
                               cwd=wd, capture_output=True, timeout=300, env=dict(os.environ, **env))
            assert r.returncode == 0, r.stderr[-600:]
            outs[tag] = open(os.path.join(wd, f"{tag}{k}.skf"), "rb").read()
        assert outs["pipe"] == outs["oneshot"] and outs["avx2"] == outs["oneshot"] and outs["plain"] == outs["oneshot"] and outs["avx512"] == outs["oneshot"], k
        want = ora.Array.build([(f"o{i}", a, b) for i, (a, b) in enumerate(files)], k=int(k), rc=True, q=ora.qual(mc, 20, oq), threads=2)
        got = ora.Array.load(os.path.join(wd, f"pipe{k}.skf"))
        got.sort_rows(); want.sort_rows()
        gk, gv, gc = got.export()
        ok, ov, oc = want.export()
        assert len(ok) > (5_000 if int(k) > 9 else 500) and np.array_equal(gk, ok) and np.array_equal(gv, ov) and np.array_equal(gc, oc), k


def test_read_set_pipeline_gives_way_to_the_sort_based_form(tmp_path):
    """A sample whose reads are one read six thousand times over puts every one of its k-mers into a partition of the count filter that
    cannot hold it: reads_sample_words leaves the sample to the sort-based form (SKF_NOT_TAKEN) while other readers are still in their
    files.  The pipeline must stop, report nothing of its own, and the one-shot form must take the batch: same .skf as
    SKX_KNOBS=no_reads_pipeline, which is the oracle's array (the kernels' verdict before any reader's)."""
    import synth
    wd = str(tmp_path)
    anc = synth.ancestor(60_000, seed=5)
    n = 6
    pairs = [synth.write_read_pair(anc, i, n, os.path.join(wd, f"r{i}"), read_len=100, coverage=25.0, seed=5) for i in range(n)]
    one = open(pairs[2][0], "rb").read().split(b"\n")[:4]
    open(pairs[2][0], "wb").write((b"\n".join(one) + b"\n") * 6000)
    open(pairs[2][1], "wb").write((b"\n".join(one) + b"\n") * 10)
    with open(os.path.join(wd, "list.txt"), "w") as f:
        for i, (a, b) in enumerate(pairs):
            f.write(f"r{i}\t{a}\t{b}\n")
    outs = {}
    for tag, env in (("pipe", {}), ("packed", {"SKX_KNOBS": "reads_raw=1"}), ("oneshot", {"SKX_KNOBS": "no_reads_pipeline=1"})):
        r = subprocess.run([SKA, "build", "-f", "list.txt", "-o", tag, "-k", "31", "--min-count", "3", "--threads", "4"], cwd=wd, capture_output=True, timeout=300,
                           env=dict(os.environ, SKX_DEBUG="1", **env))
        assert r.returncode == 0, r.stderr[-600:]
        outs[tag] = (open(os.path.join(wd, tag + ".skf"), "rb").read(), r.stderr)
    assert b"left to the sort-based form" in outs["pipe"][1]                    # the case this test is about did arise
    assert b"left to the sort-based form" in outs["packed"][1]
    assert outs["pipe"][0] == outs["oneshot"][0] and outs["packed"][0] == outs["oneshot"][0]
    want = ora.Array.build([(f"r{i}", a, b) for i, (a, b) in enumerate(pairs)], k=31, rc=True, q=ora.qual(3, 20, ora.QUAL_STRICT), threads=2)
    got = ora.Array.load(os.path.join(wd, "pipe.skf"))
    got.sort_rows(); want.sort_rows()
    gk, gv, gc = got.export()
    ok, ov, oc = want.export()
    assert len(ok) > 10_000 and np.array_equal(gk, ok) and np.array_equal(gv, ov) and np.array_equal(gc, oc)


def test_read_set_pipeline_gzip_samples_of_every_shape(tmp_path):
    """Round 6: `.fastq.gz` samples reach the device as compressed bytes (skx_gzdev.hip inflates) or as what a reader thread inflated.  A batch
    whose samples differ -- both files gzip, one gzip and one plain, one file only, a sample four times the others' size (the inflater's buffers
    are sized for the batch's largest file before the first decode), files deflated at levels 1 / 6 / 9 and as bgzip-like members -- gives the
    one-shot form's .skf bytes with the device inflating everything it may (reads_gz=2), nothing (reads_gz=1), or sharing (default), on 1, 2 and
    5 reader threads (with 3 or fewer every thread feeds the device)."""
    import gzip
    import json
    import synth
    wd = str(tmp_path)
    anc = synth.ancestor(50_000, seed=11)
    n = 7
    with open(os.path.join(wd, "list.txt"), "w") as f:
        for i in range(n):
            a, b = synth.write_read_pair(anc, i, n, os.path.join(wd, f"r{i}"), read_len=100, coverage=120.0 if i == 3 else 30.0, seed=11)
            names = []
            for j, src in enumerate((a, b)):
                raw = open(src, "rb").read()
                if i == 1 and j == 1:
                    names.append(src)                                   # a plain second file beside a gzip first one
                    continue
                if i == 5 and j == 1:
                    continue                                            # one file only
                dst = src + ".gz"
                if i == 4:
                    open(dst, "wb").write(b"".join(gzip.compress(raw[o:o + 60_000], 6) for o in range(0, len(raw), 60_000)) + gzip.compress(b""))
                else:
                    open(dst, "wb").write(gzip.compress(raw, (1, 6, 9)[i % 3]))
                names.append(dst)
            f.write(f"r{i}\t" + "\t".join(names) + "\n")
    want = None
    for knobs, threads in (("no_reads_pipeline=1", 4), ("reads_gz=2", 5), ("reads_gz=2", 1), ("", 5), ("", 2), ("reads_gz=1", 5), ("reads_gz=2,gz_chunk_kb=4", 3)):
        ph = os.path.join(wd, "ph.json")
        r = subprocess.run([SKA, "build", "-f", "list.txt", "-o", "out", "-k", "31", "--min-count", "4", "--threads", str(threads)], cwd=wd, capture_output=True, timeout=300,
                           env=dict(os.environ, SKX_KNOBS=knobs, SKX_PHASES=ph))
        assert r.returncode == 0, (knobs, r.stderr[-600:])
        got = open(os.path.join(wd, "out.skf"), "rb").read()
        want = want or got
        assert got == want, (knobs, threads)
        phases = json.load(open(ph))
        if knobs.startswith("reads_gz=2"):
            assert phases.get("build.reads_samples_sent_compressed") == n - 1 and not phases.get("build.reads_samples_inflated_on_host_after_all"), (knobs, phases)
