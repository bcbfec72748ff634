"""Deterministic synthetic bacterial assemblies (BASELINE.json configs 2-4).

One random ancestor, samples related through a balanced binary tree: every clade at every level carries
`shared_snps` substitutions, every sample `private_snps` more.  Each sample is cut into 20-100 contigs,
sprinkled with a few N runs and lower-case stretches, and every 10th sample is reverse-complemented, so
the inputs exercise contig ends, N restarts, case folding and the canonical (rc) path.

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
    cuts = np.sort(rng.choice(np.arange(1, n), size=n_contigs - 1, replace=False)) if n_contigs > 1 else np.array([], dtype=np.int64)
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
