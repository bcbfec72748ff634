#!/usr/bin/env python
"""tools/reads_1000.py [isolates=1000] [batches=4] [spots_per_batch=2]: BASELINE config 5 at its stated size on ONE GPU, through the executable:
paired FASTQ isolates (2 x 150 bp at 50x of a 5 Mbp genome, 0.5 % errors, per-cycle Phred profile; RSC_GZ=1: as .fastq.gz) in
`batches` batches -- a batch's files are simulated on tmpfs (a 50x isolate is 0.55 GB of text), built (`ska build -k 41 --min-count 5
--min-qual 20 --qual-filter strict`) and deleted --, then `ska merge` of the batch files and `ska distance` of all pairs.
Checked against the oracle (CPU, one thread per isolate, beside the next batch's simulation): `spots_per_batch` isolates of every batch --
the isolate's column of its batch .skf against the oracle's dictionary of its two files (ska_dict.rs:118-180, bloom_filter.rs:116-148),
and the rows of the final distance table for every pair of the spot isolates against the oracle's table of an array built from those
dictionaries (generic_modes.rs:136-189: a pair's counts do not depend on the other samples once constant rows are counted back in)."""
import json, os, shutil, subprocess, sys, tempfile, time
from concurrent.futures import ProcessPoolExecutor, ThreadPoolExecutor
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ska.rust_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import synth
import ora

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
NB = int(sys.argv[2]) if len(sys.argv) > 2 else 4
SPOTS = int(sys.argv[3]) if len(sys.argv) > 3 else 2
GZ = bool(os.environ.get("RSC_GZ"))
THREADS = os.environ.get("RSC_THREADS", "16")
td = tempfile.mkdtemp(dir="/dev/shm")
SKA = os.path.join(ROOT, "ska.rust_amd", "ska")
opts = ["-k", "41", "--min-count", "5", "--min-qual", "20", "--qual-filter", "strict"]
mem = {l.split(":")[0]: int(l.split()[1]) // 1024 for l in open("/proc/meminfo") if l.split(":")[0] in ("MemTotal", "MemAvailable")}
print(f"{N} isolates in {NB} batches, {SPOTS} oracle spot checks per batch, gz={GZ}; host memory {mem} MB, cpus {os.cpu_count()}", flush=True)


def _write(i, prefix):
    pr = synth.write_read_pair_of(i, N, prefix)
    if GZ:
        import zlib
        out = []
        for f in pr:
            c = zlib.compressobj(1, zlib.DEFLATED, 31)
            with open(f, "rb") as src, open(f + ".gz", "wb") as dst:
                while True:
                    b = src.read(8 << 20)
                    if not b:
                        break
                    dst.write(c.compress(b))
                dst.write(c.flush())
            os.unlink(f)
            out.append(f + ".gz")
        pr = out
    return pr


def run(tag, cmd, phases=None):
    env = dict(os.environ)
    if phases:
        env["SKX_PHASES"] = phases
    t = time.perf_counter(); r = subprocess.run(cmd, cwd=td, capture_output=True, env=env); dt = time.perf_counter() - t
    assert r.returncode == 0, (tag, r.stderr[-1500:])
    print(f"{tag}: {dt:.2f} s", flush=True)
    return dt


def oracle_dict(i, f1, f2):
    t = time.perf_counter()
    d = ora.Dict.from_files(41, f1, f2, True, ora.qual(5, 20, ora.QUAL_STRICT))
    return i, d, time.perf_counter() - t


workers = int(os.environ.get("RSC_WORKERS", str(min(64, os.cpu_count() or 1))))
bounds = [N * b // NB for b in range(NB + 1)]
wall = {"simulate": 0.0, "build": 0.0}
spot_pool = ThreadPoolExecutor(max_workers=SPOTS * 2)
pending = []            # (batch, future, files to delete afterwards)
spot_dicts = {}         # isolate -> oracle Dict
spot_batch = {}         # isolate -> its batch
t_all = time.perf_counter()
for b in range(NB):
    lo, hi = bounds[b], bounds[b + 1]
    t0 = time.perf_counter()
    with ProcessPoolExecutor(max_workers=workers) as ex:
        pairs = list(ex.map(_write, range(lo, hi), [os.path.join(td, f"iso{i}") for i in range(lo, hi)], chunksize=1))
    wall["simulate"] += time.perf_counter() - t0
    print(f"batch {b}: isolates {lo}..{hi - 1} written in {time.perf_counter() - t0:.1f} s", flush=True)
    with open(os.path.join(td, f"list{b}.txt"), "w") as lst:
        for i, (f1, f2) in zip(range(lo, hi), pairs):
            lst.write(f"iso{i}\t{f1}\t{f2}\n")
    ph = os.path.join(td, f"ph{b}.json")
    dt = run(f"batch {b}: ska build ({hi - lo} isolates)", [SKA, "build", "-f", f"list{b}.txt", "-o", f"batch{b}", "--threads", THREADS, *opts], ph)
    wall["build"] += dt
    p = json.load(open(ph))
    print("  phases", {k: round(v, 3) for k, v in p.items() if v >= 0.03}, f"= {(hi - lo) / dt:.1f} isolates/s", flush=True)
    # the oracle's dictionaries of this batch's spot isolates run beside the next batch's simulation; every other file goes now
    idx = sorted({lo + (hi - lo - 1) * j // max(1, SPOTS - 1) for j in range(SPOTS)}) if SPOTS > 0 else []
    for i, (f1, f2) in zip(range(lo, hi), pairs):
        if i in idx:
            spot_batch[i] = b
            pending.append((b, spot_pool.submit(oracle_dict, i, f1, f2), (f1, f2)))
        else:
            os.unlink(f1); os.unlink(f2)

# the batch columns of the spot isolates against the oracle's dictionaries
for b, fut, files in pending:
    i, d, secs = fut.result()
    for f in files:
        os.unlink(f)
    spot_dicts[i] = d
t0 = time.perf_counter()
for b in range(NB):
    mine = sorted(i for i, bb in spot_batch.items() if bb == b)
    if not mine:
        continue
    arr = ora.Array.load(os.path.join(td, f"batch{b}.skf"))
    keys, var, _ = arr.export()
    order = np.lexsort((keys["lo"], keys["hi"]))
    for i in mine:
        ok, ob = spot_dicts[i].export()
        col = var[order, i - bounds[b]]
        have = col != ord("-")
        same = int(have.sum()) == len(ok) and np.array_equal(keys["lo"][order][have], ok["lo"]) and np.array_equal(keys["hi"][order][have], ok["hi"]) and np.array_equal(col[have], ob)
        print(f"  isolate {i} (batch {b}): {len(ok)} split k-mers in the oracle's dictionary; its column of batch{b}.skf {'IDENTICAL' if same else 'DIFFERENT'}", flush=True)
        assert same
    del arr, keys, var
print(f"  ({len(spot_dicts)} isolates spot-checked against the oracle's dictionaries; reading the batch files took {time.perf_counter() - t0:.1f} s)", flush=True)

t_merge = run(f"ska merge ({NB} batch files)", [SKA, "merge", *[f"batch{b}.skf" for b in range(NB)], "-o", "all"], os.path.join(td, "phm.json")) if NB > 1 else 0.0
if NB == 1:
    os.rename(os.path.join(td, "batch0.skf"), os.path.join(td, "all.skf"))
print("  phases", {k: round(v, 3) for k, v in json.load(open(os.path.join(td, "phm.json"))).items() if v >= 0.05} if NB > 1 else "", flush=True)
t_dist = run(f"ska distance all.skf ({N * (N - 1) // 2} pairs)", [SKA, "distance", "all.skf", "-o", "all.tsv"], os.path.join(td, "phd.json"))
print("  phases", {k: round(v, 3) for k, v in json.load(open(os.path.join(td, "phd.json"))).items() if v >= 0.05}, flush=True)
print(f"  all.skf {os.path.getsize(os.path.join(td, 'all.skf')) / 1e9:.2f} GB, all.tsv {os.path.getsize(os.path.join(td, 'all.tsv')) / 1e6:.1f} MB", flush=True)

# the distance rows of every pair of spot isolates against the oracle's own table
if len(spot_dicts) >= 2:
    t0 = time.perf_counter()
    order = sorted(spot_dicts)
    oarr = ora.Array.from_dicts([spot_dicts[i] for i in order], [f"iso{i}" for i in order])
    want = {}
    for line in oarr.distance_tsv(0.0, True).decode().splitlines()[1:]:
        f = line.split("\t")
        want[(f[0], f[1])] = line
    got = {}
    with open(os.path.join(td, "all.tsv")) as tsv:
        next(tsv)
        for line in tsv:
            a, bn, _ = line.split("\t", 2)
            if (a, bn) in want:
                got[(a, bn)] = line.rstrip("\n")
    bad = [k for k in want if got.get(k) != want[k]]
    for k in sorted(want)[:4]:
        print("   ", got.get(k))
    print(f"  {len(want)} pairs of the spot isolates: rows of all.tsv {'IDENTICAL to' if not bad else 'DIFFERENT from'} the oracle's table ({time.perf_counter() - t0:.1f} s)", flush=True)
    assert not bad, bad[:3]
if os.environ.get("RSC_PROFILE"):                  # where `ska merge` / `ska distance` spend their device time on these very files (rocprofv3 kernel trace)
    import glob
    for tag, cmd in (("ska distance all.skf", [SKA, "distance", "all.skf", "-o", "prof.tsv"]), ("ska merge", [SKA, "merge", *[f"batch{b}.skf" for b in range(NB)], "-o", "prof_all"])):
        pd = os.path.join(td, "prof")
        shutil.rmtree(pd, ignore_errors=True)
        env = dict(os.environ, TMPDIR="/tmp", SKX_KEEP_TEARDOWN="1", SKX_PHASES=os.path.join(td, "php.json"))
        r = subprocess.run(["rocprofv3", "--kernel-trace", "--stats", "-d", pd, "--", *cmd], cwd=td, capture_output=True, env=env)
        print(f"== {tag} under rocprofv3 (rc {r.returncode})", flush=True)
        try:
            print("  phases", {k: round(v, 3) for k, v in json.load(open(os.path.join(td, "php.json"))).items() if v >= 0.03}, flush=True)
        except Exception:
            pass
        dbs = glob.glob(os.path.join(pd, "**", "*.db"), recursive=True)
        if dbs:
            out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "rocprof_stats.py"), dbs[0]], capture_output=True, text=True).stdout
            print("\n".join(l[:150] for l in out.splitlines()[:14]), flush=True)
        else:
            print(r.stderr.decode()[-800:], flush=True)
total = time.perf_counter() - t_all
print(f"wall clock: simulate {wall['simulate']:.1f} s, ska build {wall['build']:.2f} s ({N / wall['build']:.1f} isolates/s), ska merge {t_merge:.2f} s, "
      f"ska distance {t_dist:.2f} s; everything incl. the oracle {total:.1f} s", flush=True)
shutil.rmtree(td)
