"""Deterministic synthetic bacterial assemblies (BASELINE.json configs 2-4).

One random ancestor, samples related through a balanced binary tree: every clade at every level carries
`shared_snps` substitutions, every sample `private_snps` more.  Each sample is cut into 20-100 contigs,
sprinkled with a few N runs and lower-case stretches, and every 10th sample is reverse-complemented, so
the inputs exercise contig ends, N restarts, case folding and the canonical (rc) path.

`write_read_pair` simulates BASELINE config 5's read sets from the same genomes (2 x 150 bp at 50x, 0.5 % substitution errors, a
per-cycle Phred profile).

Output is the engine's *record stream* (contig bases + '\\n' per contig) as a numpy uint8 array; `to_fasta`
writes the same sample as a 60-column FASTA file for the CPU baseline.
"""
import numpy as np

_ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)
_COMP = np.zeros(256, dtype=np.uint8)
for _a, _b in zip(b"ACGTNacgtn", b"TGCANtgcan"):
    _COMP[_a] = _b


def ancestor(length=5_000_000, seed=1):
    rng = np.random.default_rng(seed)
    return _ACGT[rng.integers(0, 4, size=length, dtype=np.uint8)]


def _apply_snps(seq, rng, n):
    pos = rng.integers(0, len(seq), size=n)
    shift = rng.integers(1, 4, size=n, dtype=np.uint8)
    lut = np.zeros(256, dtype=np.uint8)
    lut[_ACGT] = np.arange(4, dtype=np.uint8)
    seq[pos] = _ACGT[(lut[seq[pos]] + shift) & 3]


def sample_bases(anc, index, n_samples, private_snps=500, shared_snps=50, seed=1):
    """Mutated genome of sample `index` (uppercase ACGT, same orientation as the ancestor)."""
    seq = anc.copy()
    levels = max(1, int(np.ceil(np.log2(max(n_samples, 2)))))
    for lvl in range(1, levels + 1):
        clade = index >> (levels - lvl)
        _apply_snps(seq, np.random.default_rng([seed, 7, lvl, clade]), shared_snps)
    _apply_snps(seq, np.random.default_rng([seed, 11, index]), private_snps)
    return seq


def sample_stream(anc, index, n_samples, private_snps=500, shared_snps=50, seed=1, decorate=True):
    """Record stream (np.uint8) of sample `index`."""
    seq = sample_bases(anc, index, n_samples, private_snps, shared_snps, seed)
    rng = np.random.default_rng([seed, 13, index])
    n = len(seq)
    if decorate:
        if index % 10 == 9:
            seq = _COMP[seq[::-1]]
        for _ in range(max(1, n // 500_000)):                 # ~0.01 % N
            p = int(rng.integers(0, max(1, n - 50)))
            seq[p:p + 50] = ord("N")
        for _ in range(5):                                    # lower-case stretches
            p = int(rng.integers(0, max(1, n - 1000)))
            seq[p:p + 1000] |= 0x20
    n_contigs = int(rng.integers(20, 101)) if (decorate and n > 10_000) else 1
    # (choice(n - 1) + 1: the same draws as choice(np.arange(1, n)) without the 40 MB array per sample)
    cuts = np.sort(rng.choice(n - 1, size=n_contigs - 1, replace=False) + 1) if n_contigs > 1 else np.array([], dtype=np.int64)
    out = np.empty(n + n_contigs, dtype=np.uint8)
    bounds = np.concatenate(([0], cuts, [n]))
    w = 0
    for a, b in zip(bounds[:-1], bounds[1:]):
        out[w:w + (b - a)] = seq[a:b]
        w += b - a
        out[w] = 10
        w += 1
    return out


def to_fasta(stream, path, width=60):
    """Write a record stream as a wrapped multi-FASTA file (one reshape per contig, no per-line Python work)."""
    ends = np.flatnonzero(stream == 10)
    parts = []
    a = 0
    for i, b in enumerate(ends):
        rec = stream[a:b]
        a = b + 1
        parts.append(np.frombuffer(b">contig_%d\n" % i, dtype=np.uint8))
        full = len(rec) // width
        if full:
            lines = np.empty((full, width + 1), dtype=np.uint8)
            lines[:, :width] = rec[:full * width].reshape(full, width)
            lines[:, width] = 10
            parts.append(lines.reshape(-1))
        if len(rec) > full * width:
            parts.append(rec[full * width:])
            parts.append(np.frombuffer(b"\n", dtype=np.uint8))
    with open(path, "wb") as f:
        f.write(np.concatenate(parts).tobytes() if parts else b"")


def write_read_pair(anc, index, n_samples, prefix, read_len=150, coverage=50.0, error_rate=0.005, seed=1):
    """BASELINE.json configs[4] / SURVEY.md 8d: paired reads of sample `index` (2 x read_len at `coverage` x, substitution errors at
    `error_rate` with low qualities, Phred from a fixed per-cycle profile with jitter, half of the reads reverse-complemented) as
    plain FASTQ files <prefix>_1.fastq / <prefix>_2.fastq.  Deterministic in (seed, index).  Returns the two paths."""
    g = sample_bases(anc, index, n_samples, seed=seed)
    glen = len(g)
    npairs = int(coverage * glen / read_len / 2)
    rng = np.random.default_rng([seed, 99, index])
    paths = []
    for mate in (0, 1):
        start = rng.integers(0, glen - read_len, size=npairs)
        reads = g[start[:, None] + np.arange(read_len)[None, :]]
        rev = rng.random(npairs) < 0.5
        reads[rev] = _COMP[reads[rev][:, ::-1]]
        err = rng.random(reads.shape) < error_rate
        reads[err] = _ACGT[rng.integers(0, 4, size=int(err.sum()))]
        prof = np.clip(38 - (np.arange(read_len) // 10), 2, 40)
        q = np.clip(prof[None, :] + rng.integers(-6, 3, size=reads.shape), 2, 41).astype(np.uint8) + 33
        q[err] = 33 + 8
        rec = np.empty((npairs, 3 + read_len + 3 + read_len + 1), np.uint8)        # "@r\n" seq "\n+\n" qual "\n"
        rec[:, 0:3] = np.frombuffer(b"@r\n", np.uint8)
        rec[:, 3:3 + read_len] = reads
        rec[:, 3 + read_len:6 + read_len] = np.frombuffer(b"\n+\n", np.uint8)
        rec[:, 6 + read_len:6 + 2 * read_len] = q
        rec[:, -1] = 10
        p = f"{prefix}_{mate + 1}.fastq"
        rec.tofile(p)
        paths.append(p)
    return paths


def write_read_pair_of(index, n_samples, prefix, genome_len=5_000_000, seed=1):
    """write_read_pair with the ancestor made on the spot: the form worker processes call (nothing large is sent to them)"""
    return write_read_pair(ancestor(genome_len, seed=seed), index, n_samples, prefix, seed=seed)
