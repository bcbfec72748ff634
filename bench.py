#!/usr/bin/env python
"""bench.py -- `ska build` + `ska align` throughput of the MI355X engine (BASELINE.json metric).

One *step* = one pass of the hot path over one batch of synthetic assemblies that are already resident in HBM:
  split-k-mer extraction + per-sample dictionaries  ->  key union + samples x k-mers matrix (MergeSkaArray)
  ->  variant-site filter (min_freq 0.9, no-const) + alignment compaction.
Workload at N=1: BASELINE.json configs[2] (1 000 related 5 Mbp assemblies, k=31).  With --gpus N each rank owns
the same number of samples (weak scaling); the only collective is one all-gather of the per-rank key tables plus
the reduction of the per-row filter statistics (ska.rust_amd/dist.py).

Prints ONE JSON line (rank 0).  `value` is that device-resident rate (inputs in HBM when the clock starts, as the bench
contract asks; the line says so in `value_is`) -- NOT what a user of the executable times: that is `end_to_end.genomes_per_s`,
~70 x lower (files in and out, not kernels).  `roofline` is for the split-k-mer extraction+scatter kernel (HBM bound): algorithmic bytes = 10 B per
input base (1 B ASCII read + 9 B (key, middle base) written) over the kernel's launch duration measured with
HIP events on the engine's stream.

At N=1 three more objects ride on the line:
  `end_to_end`   the metric as a user of the reference would time it: the same assemblies as FASTA files on tmpfs through the
                 `ska` executable -- `ska build` -> .skf -> `ska align` (two processes) and the single `ska align *.fa` form
                 (io_utils.rs:60-93) -- wall clock around each process plus the engine's own phase table (file reading +
                 upload, kernels, .skf encode / write, load, filter, FASTA out);
  `cpu_baseline` the CPU oracle (a restatement of ska.rust's algorithm, NOT the Rust binary) through the same phases (build with
                 the reference's thread rule, .skf save, .skf load, filter + write_fasta) on a bounded sample of the same files,
                 host core count and threads stated, plus its extrapolation to the thread count the reference would use at S=1000;
  `check`        engine == oracle on that sample (exact {split k-mer -> row} map, alignment columns) and on per-sample
                 dictionaries spot-checked across the full set (incl. a reverse-complemented sample).
`vs_baseline` = end_to_end genomes/s over the (conservatively extrapolated) CPU genomes/s of the same box: BASELINE.md holds no
published number, so this is a same-run, like-for-like ratio and says so in `vs_baseline_basis`.
"""
import argparse
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "ska.rust_amd"))

import numpy as np  # noqa: E402

# torch is imported inside main(), after the end-to-end leg: a parent process that has loaded the HIP runtime is a second client
# of the GPU driver, and with one alive the first large device allocation of every ska process it starts takes 2.2 s longer
# (measured: tools/seq_probe.py, profiles/r02_seq_probe.log) -- a user's shell is not such a process
torch = None
dist = None

HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: 8 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--genomes", type=int, default=1000, help="samples per GPU")
    ap.add_argument("--genome-len", type=int, default=5_000_000)
    ap.add_argument("-k", type=int, default=31)
    ap.add_argument("--cpu-genomes", type=int, default=160, help="size of the CPU-baseline sample (0 = skip)")
    ap.add_argument("--no-e2e", action="store_true", help="skip the files -> .skf -> FASTA leg through the ska executable")
    ap.add_argument("--no-check", action="store_true", help="skip the oracle comparison")
    ap.add_argument("--no-distance", action="store_true", help="skip the all-vs-all distance stage")
    ap.add_argument("--check", action="store_true", help="(kept for compatibility: the check always runs unless --no-check)")
    ap.add_argument("--cli-threads", type=int, default=0, help="--threads given to the ska executable (0 = min(64, cores))")
    ap.add_argument("--settle-s", type=float, default=12.0, help="idle seconds before each independent end-to-end measurement (the chain build -> align has none in between)")
    ap.add_argument("--no-selftest", action="store_true", help="--gpus N > 1: skip the `ska selftest --gpus N` pre-flight (id hand-off, all-reduce, gather, all-gather of unequal key tables over RCCL)")
    ap.add_argument("--no-pmc", action="store_true", help="skip the two rocprofv3 --pmc passes that measure roofline.traffic in this run")
    ap.add_argument("--pmc-genomes", type=int, default=0, help="samples of the workload the --pmc passes extract (0 = all of --genomes)")
    ap.add_argument("--pmc-child", type=int, default=0, help=argparse.SUPPRESS)      # the process rocprofv3 runs: extraction of that many samples, nothing else
    return ap.parse_args()


def private_snps(n_total):
    """500 private SNPs per sample up to 1 000 samples (configs 2-3), 100 beyond (config 4), so that the
    rows x samples matrix of the sharded runs stays within one GPU's HBM."""
    return 500 if n_total <= 1000 else 100


def classify_kernel(name):
    """which stage of the step a kernel of the rocprofv3 trace belongs to"""
    if "extract_kernel<true" in name or "extract_wide_kernel<true" in name:
        return "extract"
    if "dedupe" in name:
        return "dedup"
    for t in ("append_kernel", "append_wide_kernel", "pieces_stats_kernel", "union_kernel", "union_wide_kernel", "assemble", "gather_keys", "region_totals", "compose_perm"):
        if t in name:
            return "merge"
    for t in ("filter_flags", "scan_u8", "pieces_rows_kernel", "compact", "count_u8", "mask_ambig"):
        if t in name:
            return "filter"
    return "other"


def other_kernels(tm, steps, n_bases, n_distinct, rows, rows_kept, n_samples, key_bytes=8, piece_bytes=None, traffic=None):
    """The stages behind the extraction kernel.  `achieved` / `frac` are HBM rates on BYTES MOVED: the FETCH_SIZE x 2 + WRITE_SIZE of the
    stage's kernels measured in this run (`traffic`), or -- without the --pmc passes -- the bytes the stage's kernels must move by
    design (`needed_bytes`), and `achieved_basis` says which.  SURVEY.md 8d prices the REFERENCE's data flow (W + 1 = 9 | 17 B per
    dictionary entry, a byte per cell read and written); the fused kernels here do not move those bytes, so that figure is given as
    `reference_flow_bytes` with `speedup_vs_reference_flow` = the time that flow would take at the HBM peak / this stage's time -- a
    ratio of times that may pass 1, never a bandwidth."""
    out = []
    w1 = key_bytes + 1.0
    dict_ms, merge_ms = tm["dedupe"] / steps, (tm["key_union"] + tm["assemble"]) / steps
    ref_dict, ref_merge = w1 * (n_bases + n_distinct), w1 * n_distinct + rows * (float(key_bytes) + n_samples)
    ref_filter = rows * (float(key_bytes) + n_samples) + rows_kept * float(n_samples)
    pieces = float(piece_bytes) if piece_bytes else rows * n_samples / 2.0 * 0.6
    if dict_ms > 0:     # sorted path: per-sample dictionaries sorted and folded, then union + assemble
        stages = [("dedup", "per-sample dedup (dedupe kernel)", dict_ms, ref_dict, float(key_bytes) * (n_bases + n_distinct)),
                  ("merge", "merge (union + assemble kernels)", merge_ms, ref_merge, float(key_bytes) * n_distinct * 2 + rows * (float(key_bytes) + n_samples)),
                  ("filter", "filter + compaction", (tm["filter"] + tm["compact"]) / steps, ref_filter, ref_filter)]
    else:               # append pass: dictionaries + merge in one kernel over the unsorted regions, cells kept as 4-bit pieces
        stages = [("merge", "dictionaries + merge as one pass (append probe + append kernel + pieces_stats_kernel: regions read unsorted, cells kept as 4-bit pieces)",
                   merge_ms, ref_dict + ref_merge, float(key_bytes) * n_bases + 2.0 * pieces + rows * (key_bytes + 2.0 + 16.0)),
                  ("filter", "filter + compaction (row verdicts + pieces_rows_kernel: kept rows only)", (tm["filter"] + tm["compact"]) / steps, ref_filter,
                   pieces + rows_kept * float(n_samples) + 16.0 * rows + 9.0 * rows)]
    for key, name, ms, ref_b, need_b in stages:
        moved = None
        if traffic:
            moved = sum(v["fetch_bytes"] + v["write_bytes"] for kname, v in traffic.items() if classify_kernel(kname) == key) or None
        basis_b = moved if moved else need_b
        gbs = basis_b / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
        out.append({"stage": name, "ms": ms, "needed_bytes": need_b, "traffic": moved,
                    "achieved": gbs, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                    "achieved_basis": "bytes moved (FETCH_SIZE x 2 + WRITE_SIZE of the stage's kernels, measured in this run)" if moved else "needed_bytes (what the stage's kernels must move by design; no --pmc passes in this run)",
                    "frac_of_hbm_on_needed_bytes": need_b / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS if ms > 0 else 0.0,
                    "reference_flow_bytes": ref_b,
                    "speedup_vs_reference_flow": (ref_b / (HBM_PEAK_GBS * 1e9)) / (ms * 1e-3) if ms > 0 else 0.0})
    return out


def fill_dominant(res, tm, steps, n_bases, shape, n_samples, key_bytes, pieces_info, traffic, n_pmc):
    """roofline.dominant: the append kernel (MergeSkaDict::append for all samples, csrc/skx_append.hip -- since round 4 the largest kernel of the
    k <= 31 step).  needed_bytes = what it must move by design: every packed word of the regions once (W B per window) + the pieces, the row
    keys and perm it writes; traffic = FETCH_SIZE x 2 + WRITE_SIZE of its launches measured in this run.  Both fractions of the 8 TB/s peak."""
    ms = tm["append"] / steps
    if ms <= 0:
        res["roofline"]["dominant"] = None          # the sorted path ran (see merge_path): no append kernel in this step
        return
    words = float(key_bytes) * n_bases               # P ~ N windows, W bytes each
    written = float(pieces_info[0]) + shape[0] * (key_bytes + 2.0) if pieces_info[0] else None
    d = {"kernel": ("append_kernel<false, true>" if key_bytes == 8 else "append_wide_kernel<false, ..>") + " (rows, first-seen ranks and every sample's cells in one pass over the unsorted regions)",
         "bound": "hbm", "launch_ms": ms, "largest_kernel_of_step": bool(ms > tm["scatter"] / steps),
         "needed_bytes": None if written is None else words + written, "needed_read_bytes": words, "needed_write_bytes": written,
         "peak": HBM_PEAK_GBS, "unit": "GB/s", "traffic": None}
    if written is not None:
        d["achieved_on_needed_bytes"] = (words + written) / (ms * 1e-3) / 1e9
        d["frac_on_needed_bytes"] = d["achieved_on_needed_bytes"] / HBM_PEAK_GBS
    if traffic:
        ap = [v for k, v in traffic.items() if ("append_kernel<false" in k or "append_wide_kernel<false" in k)]
        if ap:
            fetch, wr = sum(v["fetch_bytes"] for v in ap), sum(v["write_bytes"] for v in ap)
            d.update({"traffic": fetch + wr, "fetch_bytes": fetch, "write_bytes": wr, "refetch_factor": fetch / words,
                      "achieved_on_traffic": (fetch + wr) / (ms * 1e-3) / 1e9, "frac_on_traffic": (fetch + wr) / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                      "traffic_source": f"measured in this run: the same --pmc passes as roofline.traffic ({n_pmc} samples" + (")" if n_pmc == n_samples else f", scaled per base: the re-fetch of a region by its readers depends on the set's size)")})
    res["roofline"]["dominant"] = d


def reference_threads(n_samples, cores):
    """merge_ska_dict.rs:381-385: largest power of two <= min(threads, 1 + n / 10) (the CLI's --threads defaults to the cores here)"""
    want = max(1, min(cores, 1 + n_samples // 10))
    return 1 << int(np.floor(np.log2(want)))


def cpu_baseline(args, files, n_total, td):
    """The oracle through the phases of `ska build` + `ska align` on the first --cpu-genomes files, timed on the host cores."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import ora
    n = min(args.cpu_genomes, len(files))
    cores = os.cpu_count() or 1
    threads = reference_threads(n, cores)
    threads_full = reference_threads(n_total, cores)
    inputs = [(f"g{i}", files[i], None) for i in range(n)]
    ora.timers(reset=True)
    t0 = time.perf_counter()
    arr = ora.Array.build(inputs, k=args.k, rc=True, threads=threads)
    t1 = time.perf_counter()
    skf = os.path.join(td, "cpu_sample.skf")
    arr.save(skf)
    t2 = time.perf_counter()
    arr2 = ora.Array.load(skf)
    t3 = time.perf_counter()
    aln = arr2.align(min_freq=0.9)
    with open(os.path.join(td, "cpu_sample.aln"), "wb") as f:
        f.write(aln)
    t4 = time.perf_counter()
    tm = ora.timers()
    build_s, save_s, load_s, align_s = t1 - t0, t2 - t1, t3 - t2, t4 - t3
    total = t4 - t0
    # What the reference would do at S = n_total on this box: its thread rule gives `threads_full` threads for the sample-parallel
    # part (dictionaries + appends); everything else (tree merge tail, .skf codec, filter, write_fasta) is serial in the reference.
    # Perfect scaling of the parallel part and per-genome serial costs no higher than in the sample are both assumptions in the
    # CPU's favour (the serial costs grow with rows x samples).
    par = tm["read_parse"] + tm["dict"] + tm["append"]           # summed over threads
    build_serial = max(0.0, build_s - par / threads)
    per_genome_scaled = (par / n) / threads_full + (build_serial + save_s + load_s + align_s) / n
    quota = None                                       # CPUs the job may use at a time (cgroup v2 cpu.max), where the box says
    try:
        q_, p_ = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        quota = None if q_ == "max" else float(q_) / float(p_)
    except Exception:
        pass
    res = {"value": n / total, "unit": "genomes/s", "cores": threads, "threads": threads, "host_cores": cores, "cpu_quota": quota, "kind": "port",
           "sample": f"{n} of the {n_total} synthetic {args.genome_len} bp assemblies as FASTA files on tmpfs: ska build ({threads} threads = "
                     f"the reference's rule for {n} samples) -> .skf -> load -> filter + write_fasta",
           "phases_s": {"build": build_s, "skf_save": save_s, "skf_load": load_s, "filter_write_fasta": align_s},
           "oracle_timers_thread_seconds": tm,
           "rows": int(arr.nrows), "alignment_bytes": len(aln), "skf_bytes": os.path.getsize(skf),
           "threads_reference_rule_full_set": threads_full,
           "scaled_to_full_set_threads": {"value": 1.0 / per_genome_scaled, "unit": "genomes/s",
                                          "how": f"sample-parallel thread-seconds / {threads_full} threads (perfect scaling) + the serial phases per genome as measured on the sample"}}
    return res, arr, aln


def check_against_oracle(args, E, ctx, files, oarr, oaln, n_total, anc, synth):
    """engine == oracle: the CPU sample's merged array and alignment, and dictionaries spot-checked across the whole set"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import ora
    out = {}
    t0 = time.perf_counter()
    if oarr is not None:
        n = oarr.nsamples
        garr = E.Array.build([(f"g{i}", files[i], None) for i in range(n)], k=args.k, rc=True, threads=min(32, os.cpu_count() or 1), ctx=ctx)
        gk, gv, gc = garr.export()
        ok_, ov, oc = oarr.export()
        out["sample_array_equal_oracle"] = bool(np.array_equal(gk["lo"], ok_["lo"]) and np.array_equal(gv, ov) and np.array_equal(gc, oc))
        out["sample_array_shape"] = [int(garr.nrows), int(n)]
        galn = garr.align(min_freq=0.9)
        # the reference leaves the column order to its hash map (tests/common/mod.rs:166-189 compares column sets): compare the
        # alignments as sorted column lists
        def cols(aln):
            rows = aln.split(b"\n")[1::2]
            m = np.frombuffer(b"".join(rows), dtype=np.uint8).reshape(len(rows), -1) if rows and rows[0] else np.zeros((0, 0), np.uint8)
            order = np.lexsort(m[::-1]) if m.size else np.zeros(0, np.int64)
            return m[:, order]
        out["sample_alignment_columns_equal_oracle"] = bool(np.array_equal(cols(galn), cols(oaln)))
        garr.free()
    # dictionaries of samples spread over the set; index 9 mod 10 is reverse-complemented by the generator
    idx = sorted({0, 9, len(files) // 3, len(files) // 2 + 9 - (len(files) // 2) % 10 if len(files) > 20 else 1, len(files) - 1} & set(range(len(files))))
    ds = E.DictSet.from_files([(files[i], None) for i in idx], args.k, True, threads=len(idx), ctx=ctx)
    ok = True
    for j, i in enumerate(idx):
        od = ora.Dict.from_files(args.k, files[i])
        gk, gb = ds.export(j)
        okk, ob = od.export()
        ok &= bool(np.array_equal(gk["lo"], okk["lo"]) and np.array_equal(gb, ob))
    ds.free()
    out["dicts_equal_oracle"] = ok
    out["dict_samples_checked"] = idx
    out["seconds"] = time.perf_counter() - t0
    return out


def run_cli(ska, argv, cwd, phases_path, extra_env=None):
    env = dict(os.environ, SKX_PHASES=phases_path, **(extra_env or {}))
    t0 = time.perf_counter()
    r = subprocess.run([ska, *argv], cwd=cwd, capture_output=True, env=env)
    dt = time.perf_counter() - t0
    if r.returncode != 0:
        raise RuntimeError(f"ska {argv[0]} failed ({r.returncode}): {r.stderr[-400:].decode(errors='replace')}")
    try:
        ph = json.load(open(phases_path))
    except Exception:
        ph = None
    return dt, ph


def files_equal(a, b, block=64 << 20):
    if os.path.getsize(a) != os.path.getsize(b):
        return False
    with open(a, "rb") as fa, open(b, "rb") as fb:
        while True:
            x, y = fa.read(block), fb.read(block)
            if x != y:
                return False
            if not x:
                return True


def warm_device(device=0, frac=0.9, chunk_gb=16):
    """Device memory costs a process time twice over, and neither is the executable's: a box that has just come up clears VRAM the first time
    it is handed out (tools/vram_probe.sh, profiles/r04zzm_vram_probe.log: hipMalloc of 56 GB 0.97 s as the box's first GPU process, and
    `ska build` 0.98 s in build.dictionaries behind a warm-up that had only taken 56 GB -- the allocator need not hand out the same
    pages), and memory a process released a moment ago is wiped before it is handed out again (2.4-3.8 s right behind one, 0.000 s two
    seconds later).  The CONDITIONED end-to-end chain is measured behind this: a process takes `frac` of what hipMemGetInfo calls free on
    the bench's device through the HIP runtime (no torch: the executable does not use it either), chunk by chunk, and releases it; the idle
    seconds that follow cover the wipe.  The cold chain (end_to_end.first_process) runs before it, as a user's first job on a fresh node."""
    code = ("import ctypes\n"
            "h = ctypes.CDLL('libamdhip64.so')\n"
            f"assert h.hipSetDevice({int(device)}) == 0\n"
            "fr, tot = ctypes.c_size_t(), ctypes.c_size_t()\n"
            "assert h.hipMemGetInfo(ctypes.byref(fr), ctypes.byref(tot)) == 0\n"
            f"want, ps = int(fr.value * {frac}), []\n"
            f"while want >= ({chunk_gb} << 30):\n"
            "    p = ctypes.c_void_p()\n"
            f"    if h.hipMalloc(ctypes.byref(p), ctypes.c_size_t({chunk_gb} << 30)) != 0: break\n"
            f"    h.hipMemset(p, 0, ctypes.c_size_t(1)); ps.append(p); want -= ({chunk_gb} << 30)\n"
            "h.hipDeviceSynchronize()\n"
            "for p in ps: h.hipFree(p)\n")
    try:
        subprocess.run([sys.executable, "-c", code], timeout=180, env=dict(os.environ, LD_LIBRARY_PATH=os.environ.get("LD_LIBRARY_PATH", "") + ":/opt/rocm/lib"))
    except Exception as e:            # a warm-up that fails changes a number, not a result
        sys.stderr.write(f"bench.py: device warm-up skipped: {e}\n")


def end_to_end(args, files, td, device=0):
    """files -> `ska build` -> .skf -> `ska align` and the single `ska align *.fa`, through the executable, on tmpfs -- twice: as the first GPU
    processes this bench starts (on a fresh node: the box's first, device memory cleared as it is first handed out), then conditioned"""
    ska = os.path.join(ROOT, "ska.rust_amd", "ska")
    n = len(files)
    threads = args.cli_threads or min(64, os.cpu_count() or 1)
    with open(os.path.join(td, "list.txt"), "w") as f:
        for i, p in enumerate(files):
            f.write(f"g{i}\t{p}\n")
    env_dev = {"HIP_VISIBLE_DEVICES": str(device)} if device else {}
    build_argv = ["build", "-f", "list.txt", "-o", "all", "-k", str(args.k), "--threads", str(threads)]
    align_argv = ["align", "all.skf", "-o", "aln.fa", "--threads", str(threads)]
    # 1. the cold chain: no GPU process of this bench has run yet (bench.py opens the device only after this leg)
    tb0, pb0 = run_cli(ska, build_argv, td, os.path.join(td, "ph_build0.json"), env_dev)
    ta0, pa0 = run_cli(ska, align_argv, td, os.path.join(td, "ph_align0.json"), env_dev)
    os.replace(os.path.join(td, "aln.fa"), os.path.join(td, "aln0.fa"))
    # 2. the conditioned chain.  Each independent measurement starts on a box that has been idle for a few seconds: VRAM another process has
    # just released is wiped by the driver before it is handed out again, and an allocation of tens of GB made right after waits for it
    # (DESIGN.md section 8).  The chain `ska build` -> `ska align x.skf` is one measurement and runs back to back, as a user's script would.
    # (the cold chain's .skf goes first: overwriting it would charge `ska build` for giving back the 2.8 GB of the old file -- 0.3 s of tmpfs
    # page frees inside its fopen -- which is no part of a build that writes a new file)
    os.unlink(os.path.join(td, "all.skf"))
    warm_device(device)
    time.sleep(max(args.settle_s, 0.0))
    tb, pb = run_cli(ska, build_argv, td, os.path.join(td, "ph_build.json"), env_dev)
    skf_bytes = os.path.getsize(os.path.join(td, "all.skf"))
    ta, pa = run_cli(ska, align_argv, td, os.path.join(td, "ph_align.json"), env_dev)
    aln_bytes = os.path.getsize(os.path.join(td, "aln.fa"))
    same_cold = files_equal(os.path.join(td, "aln.fa"), os.path.join(td, "aln0.fa"))
    os.unlink(os.path.join(td, "aln0.fa"))
    # the single-invocation form builds with the CLI defaults (k = 31): only comparable when the bench runs at k = 31
    ts, ps, same = None, None, None
    if args.k == 31:
        time.sleep(max(args.settle_s, 0.0))
        ts, ps = run_cli(ska, ["align", "--threads", str(threads), "-o", "aln2.fa", *files], td, os.path.join(td, "ph_single.json"), env_dev)
        same = files_equal(os.path.join(td, "aln.fa"), os.path.join(td, "aln2.fa"))
    res = {"genomes_per_s": n / (tb + ta), "unit": "genomes/s", "samples": n, "cli_threads": threads,
           "what": "wall clock around the ska executable (process start to exit), FASTA files / .skf / alignment on tmpfs; build -> align back to back.  "
                   f"genomes_per_s is the CONDITIONED chain (a warm-up process took and released 90 % of the device's free memory, then {args.settle_s:g} s of idle; "
                   "the same idle before the single-invocation form); first_process is the same chain run BEFORE any other GPU process of this bench -- "
                   "on a fresh node the box's first: what a user's first job gets",
           "ska_build_s": tb, "ska_align_skf_s": ta, "skf_bytes": skf_bytes, "alignment_bytes": aln_bytes,
           "first_process": {"genomes_per_s": n / (tb0 + ta0), "ska_build_s_first_process": tb0, "ska_align_skf_s_first_process": ta0,
                             "alignment_identical_to_conditioned_run": same_cold, "phases_ska_build": pb0, "phases_ska_align_skf": pa0},
           "phases_ska_build": pb, "phases_ska_align_skf": pa}
    if ts is not None:
        res.update({"ska_align_fasta_single_s": ts, "genomes_per_s_single_invocation": n / ts, "phases_ska_align_fasta_single": ps,
                    "single_invocation_alignment_identical": same})
    return res


def preflight(args, world):
    """The first contact of a sharded job with a node should not be the timed run: rank 0 runs `ska selftest --gpus N` (one process per GPU,
    each exchange of the sharded job once over RCCL and once over the host-staged transport, then BASELINE config 4's key-table exchange at
    its size; 60 s + 180 s limits with the stage it stopped in) before
    any rank of the bench opens its device, and the bench stops with that message when it fails."""
    ska = os.path.join(ROOT, "ska.rust_amd", "ska")
    t0 = time.perf_counter()
    # A failed pre-flight is reported -- on stderr at once, and in the line's `selftest` -- and the timed run goes ahead: the self-test
    # covers more than the bench uses (the host-staged transport, the gather to rank 0), and a bench that can run should leave its line.
    # What the bench itself needs and does not get fails it loudly further down (SKX_BENCH_STRICT_SELFTEST=1: stop here instead).
    def failed(why):
        sys.stderr.write(f"bench.py: `ska selftest --gpus {world}` {why}\n")
        if os.environ.get("SKX_BENCH_STRICT_SELFTEST") == "1":
            raise SystemExit(f"bench.py: `ska selftest --gpus {world}` {why}")
        return {"seconds": time.perf_counter() - t0, "failed": why}
    try:
        r = subprocess.run([ska, "selftest", "--gpus", str(world)], capture_output=True, timeout=300)
    except subprocess.TimeoutExpired:
        return failed("did not finish in 300 s")
    except OSError as e:
        return failed(f"could not be started: {e}")
    msg = r.stderr.decode(errors="replace").strip().splitlines()
    if r.returncode != 0:
        return failed(f"failed (rc {r.returncode}): " + " | ".join(msg[-4:]))
    return {"seconds": time.perf_counter() - t0, "report": msg[-1] if msg else ""}


def launch_ranks(args):
    """`python bench.py --gpus N` without a launcher around it: start the N ranks (torch.distributed.run, one per GPU, 127.0.0.1) and hand on
    their exit code -- a plain call must never run one rank and print n_gpus: 1."""
    port = int(os.environ.get("MASTER_PORT", "29533"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd)


PMC_CHILD_STEPS = 2


def pmc_child(args):
    """What the --pmc passes profile: the whole step (extraction, merge, filter + kept rows) on the first --pmc-child samples of the same
    workload (same ancestor, same SNP pattern), PMC_CHILD_STEPS times, nothing else on the device -- measure_traffic reads the counters of
    every kernel launched."""
    import synth
    import torch
    import skx_engine as E
    n = args.pmc_child
    anc = synth.ancestor(args.genome_len, seed=1)
    streams = [synth.sample_stream(anc, i, args.genomes, private_snps=private_snps(args.genomes)) for i in range(n)]
    offs, tot = [], 0
    for s in streams:
        offs.append(tot)
        tot += (len(s) + 255) // 256 * 256
    E.load_library()
    ctx = E.Context(int(os.environ.get("SKX_BENCH_DEVICE", "0")))
    dev = torch.device("cuda", int(os.environ.get("SKX_BENCH_DEVICE", "0")))
    pool = torch.empty(tot + 256, dtype=torch.uint8, device=dev)
    for o, s in zip(offs, streams):
        pool[o:o + len(s)] = torch.from_numpy(s).to(dev)
    torch.cuda.synchronize()
    ptrs, lens = [pool.data_ptr() + o for o in offs], [len(s) for s in streams]
    names = [f"g{i}" for i in range(n)]
    for _ in range(PMC_CHILD_STEPS):
        ds = E.DictSet.build_device(ptrs, lens, args.k, True, ctx=ctx)
        arr = ds.merge(names)
        arr.apply_filters(0.9, False, E.FILTER_NO_CONST, False, False)
        ctx.sync()
        arr.free()
        ds.free()


def measure_traffic(args, lens):
    """HBM traffic measured in this run: two rocprofv3 passes (FETCH_SIZE, WRITE_SIZE: they do not fit one pass, and --pmc goes with
    --kernel-trace only) over a child process that runs the step on the first n samples of this workload.  Per kernel name and step: HBM
    bytes read = FETCH_SIZE x 2 (gfx950 tallies 128-B read requests at 64 B: MI355X_MICROARCH.md) and written = WRITE_SIZE, both counted in KB.
    Returns ({kernel name: {fetch_bytes, write_bytes, launches_per_step}}, n, bases the child processed per step) or (None, why not, 0)."""
    import csv
    import glob
    import subprocess
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "rocprofv3 not found", 0
    if "ROCP_TOOL_LIBRARIES" in os.environ or "rocprofiler" in os.environ.get("LD_PRELOAD", ""):
        return None, "this process is itself running under a profiler", 0
    n = min(args.pmc_genomes or args.genomes, args.genomes)
    bases = float(sum(lens[:n]))
    out = tempfile.mkdtemp(prefix="skx_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    for v in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(v, None)
    per = {}
    try:
        for c in ("FETCH_SIZE", "WRITE_SIZE"):
            cmd = [exe, "--pmc", c, "--kernel-trace", "--output-format", "csv", "-d", os.path.join(out, c), "--", sys.executable,
                   os.path.abspath(__file__), "--pmc-child", str(n), "--genomes", str(args.genomes), "--genome-len", str(args.genome_len), "-k", str(args.k)]
            try:
                p = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=420)
            except subprocess.TimeoutExpired:
                return None, f"the {c} pass did not finish in 420 s", 0
            if p.returncode != 0:
                return None, f"the {c} pass failed (rc {p.returncode}): " + p.stdout.decode(errors="replace")[-200:].replace("\n", " | "), 0
            rows = 0
            for f in glob.glob(os.path.join(out, c, "**", "*counter_collection.csv"), recursive=True):
                for r in csv.DictReader(open(f)):
                    if r.get("Counter_Name") != c:
                        continue
                    e = per.setdefault(r.get("Kernel_Name", "?"), {"FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0, "n": 0})
                    e[c] += float(r["Counter_Value"]) * 1024.0
                    if c == "FETCH_SIZE":
                        e["n"] += 1
                    rows += 1
            if not rows:
                return None, f"no {c} rows in the pass's output", 0
    finally:
        shutil.rmtree(out, ignore_errors=True)
    res = {k: {"fetch_bytes": v["FETCH_SIZE"] * 2.0 / PMC_CHILD_STEPS, "write_bytes": v["WRITE_SIZE"] / PMC_CHILD_STEPS, "launches_per_step": v["n"] / PMC_CHILD_STEPS}
           for k, v in per.items()}
    return res, n, bases


def main():
    args = parse()
    if args.pmc_child > 0:
        return pmc_child(args)
    if args.gpus < 1:
        raise SystemExit("--gpus must be at least 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world != args.gpus and os.environ.get("SKX_BENCH_FORCE_SHARDED") != "1":
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s) (WORLD_SIZE): refusing to report a line for the wrong number of GPUs")
    local_rank = int(os.environ.get("SKX_BENCH_DEVICE", os.environ.get("LOCAL_RANK", "0")))   # override: ranks sharing one GPU in tests
    if not os.path.exists("/dev/kfd"):
        raise SystemExit("bench.py needs a gfx950 GPU: the engine has no CPU path")
    sharded = world > 1 or os.environ.get("SKX_BENCH_FORCE_SHARDED") == "1"     # the override runs the exchange path at world size 1
    selftest = None
    if world > 1 and rank == 0 and not args.no_selftest and os.environ.get("SKX_BENCH_BACKEND", "nccl") == "nccl":
        selftest = preflight(args, world)      # (the other ranks wait in init_process_group below)
    import synth

    G = args.genomes
    n_total = G * world
    lo = rank * G
    anc = synth.ancestor(args.genome_len, seed=1)
    # ---- inputs -> HBM (not timed): one 16-B aligned record stream per sample
    lens, offs, tot = [], [], 0
    streams = []
    # N = 1: the same assemblies also go to tmpfs as 60-column FASTA files for the end-to-end, CPU-baseline and check legs
    want_files = rank == 0 and world == 1 and not (args.no_e2e and args.no_check and args.cpu_genomes <= 0)
    td = tempfile.mkdtemp(prefix="skx_bench_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None) if want_files else None
    files = []
    for i in range(G):
        s = synth.sample_stream(anc, lo + i, n_total, private_snps=private_snps(n_total))
        if want_files:
            files.append(os.path.join(td, f"g{lo + i}.fa"))
            synth.to_fasta(s, files[-1])
        streams.append(s)
        offs.append(tot)
        lens.append(len(s))
        tot += (len(s) + 255) // 256 * 256
    # The end-to-end leg goes first, before this process opens the device: the ska executable then finds the GPU as a fresh login
    # would (device memory that another process has just released is wiped by the driver before it is handed out again, at
    # ~20 GB/s on this box -- DESIGN.md section 8 -- so a leg that follows 80 GB of this process's own buffers would be charged
    # for them)
    e2e = None
    if want_files and not args.no_e2e:
        e2e = end_to_end(args, files, td, local_rank)
    global torch, dist
    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a gfx950 GPU: the engine has no CPU path")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    backend = None
    if sharded:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        backend = os.environ.get("SKX_BENCH_BACKEND", "nccl")      # "gloo" lets two ranks share one GPU in tests
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    import dist as skdist
    import skx_engine as E
    E.load_library()
    ctx = E.Context(local_rank)
    comm = skdist.make_comm(ctx, rank, world, transport="rccl" if backend == "nccl" else "local") if sharded else None
    rank_report = None
    if sharded:
        # what the communicator says it runs on, per rank: ncclCommCount, the device RCCL bound, the context's device, the device's bus id
        nr, rdev, cdev = comm.transport()
        mine = {"rank": rank, "rccl_ranks": nr, "rccl_device": rdev, "ctx_device": cdev, "pci_bus_id": getattr(torch.cuda.get_device_properties(dev), "pci_bus_id", None),
                "hip_device": torch.cuda.current_device()}
        allr = [None] * world
        if world > 1:
            dist.all_gather_object(allr, mine)
        else:
            allr = [mine]
        rank_report = {"rccl_ranks": nr, "devices": allr}
        if backend == "nccl" and nr != world:
            raise SystemExit(f"bench.py: rank {rank}: the RCCL communicator has {nr} rank(s), the job {world}")
    pool = torch.empty(tot + 256, dtype=torch.uint8, device=dev)
    for i, s in enumerate(streams):
        pool[offs[i]:offs[i] + lens[i]] = torch.from_numpy(s).to(dev, non_blocking=False)
    del streams
    base = pool.data_ptr()
    assert base % 256 == 0
    ptrs = [base + o for o in offs]
    names = [f"g{lo + i}" for i in range(G)]
    total_bases = int(sum(lens))
    torch.cuda.synchronize()

    host_ms = {"build": 0.0, "merge": 0.0, "align": 0.0, "free": 0.0}
    xch = {"key_table_exchange_s": 0.0, "row_stats_s": 0.0, "row_stats_bytes_per_rank": 0}     # sharded runs: the exchanges

    def step():
        t_a = time.perf_counter()
        ds = E.DictSet.build_device(ptrs, lens, args.k, True, ctx=ctx)
        ctx.sync()
        t_b = time.perf_counter()
        host_ms["build"] += (t_b - t_a) * 1e3
        if not sharded:
            arr = ds.merge(names)
        else:
            # the exchanges are the engine's (include/skx.h "Collectives": RCCL on the engine's stream)
            ks = ds.union_keys(notes=True)           # the notes travel with the key set to the global rows: the assemble below reads no dictionary twice
            ctx.sync()
            t_x = time.perf_counter()
            rows = comm.keyset_allgather(ks)         # one all-gather of the per-rank key tables + their union: the global rows
            ctx.sync()
            xch["key_table_exchange_s"] += time.perf_counter() - t_x
            ks.free()
            arr = ds.assemble(rows, names)
            ctx.sync()
            t_x = time.perf_counter()
            comm.reduce_stats(arr, n_total)          # counts in one all-reduce, code sets all-gathered and OR-ed
            ctx.sync()
            xch["row_stats_s"] += time.perf_counter() - t_x
            xch["row_stats_bytes_per_rank"] = int(arr.nrows * (4 + 2 * world))
        ctx.sync()
        t_c = time.perf_counter()
        host_ms["merge"] += (t_c - t_b) * 1e3
        info = (arr.nrows, arr.nsamples)
        removed = arr.apply_filters(0.9, False, E.FILTER_NO_CONST, False, False)      # ska align defaults
        ctx.sync()
        t_d = time.perf_counter()
        host_ms["align"] += (t_d - t_c) * 1e3
        out = (info[0], arr.nrows, removed)
        ds.free()
        host_ms["free"] += (time.perf_counter() - t_d) * 1e3
        return arr, out

    for _ in range(args.warmup):
        arr, _ = step()
        arr.free()
    ctx.timings(reset=True)
    for kk in host_ms:
        host_ms[kk] = 0.0
    xch["key_table_exchange_s"] = xch["row_stats_s"] = 0.0
    E.phases(reset=True)
    if sharded:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    last = None
    for _ in range(args.steps):
        if last is not None:
            last.free()
        last, shape = step()
    torch.cuda.synchronize()
    if sharded:
        dist.barrier()
    dt = time.perf_counter() - t0
    tm = ctx.timings()
    if sharded:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    n_distinct = None
    distance_stage = None
    unfiltered = None
    pieces_info, merge_path = (0, 0, 0), ctx.merge_path()
    if rank == 0 and world == 1:       # the unfiltered rows x samples matrix (what a .skf holds), produced from the merge's pieces: outside `value`
        ds = E.DictSet.build_device(ptrs, lens, args.k, True, ctx=ctx)
        arr_u = ds.merge(names)
        ctx.sync()
        pieces_info = arr_u.pieces_info()          # (bytes of pieces, row blocks, ranks per block) while the array is still held as the append pass left it
        merge_path = ctx.merge_path()
        ctx.timings(reset=True)
        t_u0 = time.perf_counter()
        arr_u.device_matrix()
        ctx.sync()
        unfiltered = {"rows": int(arr_u.nrows), "samples": G, "matrix_bytes": float(arr_u.nrows) * G, "wall_ms": (time.perf_counter() - t_u0) * 1e3,
                      "kernel_ms": ctx.timings()["assemble"],
                      "what": "MergeSkaArray's full matrix (1 byte per cell, rows in the order of H) written from the 4-bit pieces the merge leaves "
                              "(pieces_rows_kernel<all rows>); `ska build` streams it window by window into the .skf, `ska align` never writes it"}
        arr_u.free()
        ds.free()
    if rank == 0:                      # sum of the per-sample dictionary sizes (untimed): the D of SURVEY.md 8d's per-stage bytes
        # (counted from the merged array's cells -- a sample's split k-mers are its cells: no dictionary is sorted for it)
        ds = E.DictSet.build_device(ptrs, lens, args.k, True, ctx=ctx)
        arr_d = ds.merge(names)
        n_distinct = int(arr_d.sample_kmers().sum())
        if world != 1 or args.no_distance:
            arr_d.free()
        else:
            # `ska distance` on the same array (outside `value`): the constant-site filter, bit planes, all pairs
            ctx.sync()
            ctx.timings(reset=True)
            t_d0 = time.perf_counter()
            if args.warmup > 0:                                                 # like the steps: one untimed pass first (the first heavy popcount
                arr_d.distance_filtered(0.0, True)                              # kernel after a pause runs at half speed until the clocks are up)
                ctx.sync()
                ctx.timings(reset=True)
                t_d0 = time.perf_counter()
            dd, constant, rows_d = arr_d.distance_filtered(0.0, True)          # generic_modes.rs:136-189: constant rows skipped while the planes are built
            ctx.sync()
            t_d = time.perf_counter() - t_d0
            pairs = G * (G - 1) // 2
            k_ms = ctx.timings()["distance"]
            distance_stage = {"samples": G, "rows_after_no_const": int(rows_d), "pairs": pairs, "wall_s": t_d, "kernels_ms": k_ms,
                              "pairs_per_s": pairs / t_d, "naive_bytes": 2.0 * rows_d * pairs, "naive_GBps": 2.0 * rows_d * pairs / (k_ms * 1e-3) / 1e9,
                              "tiled_bound_bytes": float(G) * rows_d, "tiled_bound_ms_at_peak": float(G) * rows_d / (HBM_PEAK_GBS * 1e9) * 1e3,
                              "constant_rows": int(constant),
                              "what": "in process, array resident: row verdicts + bit planes of the kept rows + 64x64-pair popcount tiles (merge_ska_array.rs:416-438,587-632); "
                                      "naive = every pair reading both rows (the reference's loop), tiled bound = the matrix read once",
                              "first_pair": [float(dd["distance"][0]), int(dd["match_count"][0]), int(dd["mismatch_count"][0])]}
            arr_d.free()
        ds.free()
    res = None
    if rank == 0:
        steps = max(args.steps, 1)
        scatter_ms = tm["scatter"] / steps
        key_bytes = 8 if args.k <= 31 else 16                      # 1 B read + (W + 1) B written per base, W = 8 | 16
        algo_bytes = (2.0 + key_bytes) * total_bases
        achieved = algo_bytes / (scatter_ms * 1e-3) / 1e9 if scatter_ms > 0 else 0.0
        traffic, traffic_src = None, None       # HBM bytes per launch from the newest PMC passes recorded under profiles/ (FETCH_SIZE x2 + WRITE_SIZE), scaled per base
        try:
            import glob
            cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_extract*.json")), key=lambda f: (json.load(open(f)).get("date", ""), os.path.basename(f)))
            pmc = json.load(open(cands[-1]))
            if args.k <= 31:
                traffic = (pmc["read_bytes_per_base"] + pmc["write_bytes_per_base"]) * total_bases
                traffic_src = f"{pmc.get('source', os.path.basename(cands[-1]))}, recorded {pmc.get('date', 'in round 2')}: a separate --pmc run of this kernel, scaled per base (not measured in this run)"
        except Exception:
            pass
        res = {
            "metric": "genomes/sec ska build+align, 1 000x5 Mbp k=31; bit-exact vs CPU" if args.k == 31 else f"genomes/sec ska build+align, k={args.k}; bit-exact vs CPU",
            "value": n_total * steps / dt, "unit": "genomes/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u64" if args.k <= 31 else "u128", "data": "synthetic",
            "value_is": "the device-resident step (inputs in HBM when the clock starts, no file in or out: the bench contract's `value`); the metric as a user times it -- "
                        "FASTA files in, .skf and alignment out, through the ska executable -- is end_to_end.genomes_per_s (N = 1), ~70 x lower: files, not kernels",
            "config": {"workload": f"ska build + ska align, {G} synthetic {args.genome_len} bp assemblies per GPU, k={args.k}, "
                                   f"inputs resident in HBM (BASELINE.json configs[2])",
                       "samples_per_gpu": G, "private_snps": private_snps(n_total), "genome_len": args.genome_len, "k": args.k,
                       "rows_U": shape[0], "rows_kept": shape[1], "parallelism": f"samples sharded x{world}" + (" (key-table all-gather path)" if sharded else "")},
            "roofline": {"bound": "hbm", "kernel": ("extract_kernel<true>" if args.k <= 31 else "extract_wide_kernel<true>") + " (split k-mer extraction + bucket scatter)",
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "algorithmic_bytes_per_launch": algo_bytes, "launch_ms": scatter_ms,
                         "largest_kernel_of_step": "extraction" if scatter_ms >= tm["append"] / steps else "append (see dominant)"},
            "kernel_pipeline": {"genomes_per_s": n_total * steps / dt, "what": "= value: extraction -> dictionaries -> merge -> filter with the record streams resident in HBM"},
            "stage_ms_per_step": {k: v / steps for k, v in tm.items()},
            "merge_path": merge_path,
            "host_wall_ms_per_step": {k: v / steps for k, v in host_ms.items()},
        }
        if distance_stage:
            res["distance"] = distance_stage
        if unfiltered:
            res["unfiltered_form"] = unfiltered
        if sharded:
            res["selftest"] = selftest
            res["rccl_ranks"] = rank_report["rccl_ranks"]
            res["rank_devices"] = rank_report["devices"]
            ph = E.phases()
            res["exchange_per_step_rank0"] = {"transport": "rccl" if backend == "nccl" else "local (host-staged: ranks sharing a device)",
                                              "key_table_exchange_ms": xch["key_table_exchange_s"] / steps * 1e3,
                                              "key_table_allgather_ms": ph.get("comm.key_table_allgather", 0.0) / steps * 1e3,
                                              "row_stats_ms": xch["row_stats_s"] / steps * 1e3, "row_stats_bytes": xch["row_stats_bytes_per_rank"],
                                              "bytes_received_per_step": comm.bytes_received / (steps + args.warmup),
                                              "what": "skx_keyset_allgather (one ncclAllGather of the per-rank key tables, then their union) + skx_array_reduce_stats "
                                                      "(one ncclAllReduce + one ncclAllGather of the per-row filter statistics), issued by the engine; nothing else crosses xGMI"}
        res["roofline"]["traffic_source"] = traffic_src
        if world > 1:
            res["other_kernels"] = other_kernels(tm, steps, total_bases, n_distinct, shape[0], shape[1], G, key_bytes)
        try:                                     # measurement builds (-DSKX_PHASE_PROF=n, tools/mkvariant.sh): cycle shares per phase of the instrumented kernel
            import ctypes
            buf = (ctypes.c_ulonglong * 16)()
            E._lib.skx_debug_phase_prof(buf, 1)
            tot = sum(buf) or 1
            sys.stderr.write("phases (share of cycles): %s cycles in all: %d raw: %s\n" % ([round(x / tot, 3) for x in buf[:10]], tot, list(buf)))
        except AttributeError:
            pass
        if world == 1:
            # the legs below run outside the timed region; the bench's own device buffers go first (the ska executable gets the GPU)
            if last is not None:
                last.free()
            del pool
            torch.cuda.empty_cache()
            kernel_traffic = None
            if not args.no_pmc:
                kt, n_pmc, bases_pmc = measure_traffic(args, lens)
                if kt is not None:
                    scale = total_bases / bases_pmc            # (1 when the child ran the whole workload: the default)
                    kernel_traffic = {k: {"fetch_bytes": v["fetch_bytes"] * scale, "write_bytes": v["write_bytes"] * scale, "launches_per_step": v["launches_per_step"]} for k, v in kt.items()}
                    ex = [v for k, v in kernel_traffic.items() if classify_kernel(k) == "extract"]
                    if ex:
                        res["roofline"]["traffic"] = sum(v["fetch_bytes"] + v["write_bytes"] for v in ex)
                        res["roofline"]["traffic_source"] = (f"measured in this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (one pass each, --kernel-trace only) over a child process running the step on "
                                                             f"{n_pmc} samples of this workload {PMC_CHILD_STEPS} times; per launch of the extraction kernel: FETCH_SIZE x2 = "
                                                             f"{sum(v['fetch_bytes'] for v in ex) / total_bases:.3f} B + WRITE_SIZE = {sum(v['write_bytes'] for v in ex) / total_bases:.3f} B per base"
                                                             + ("" if n_pmc == G else ", scaled per base to this step"))
                    else:
                        kernel_traffic = None
                        res["roofline"]["traffic_source"] = (res["roofline"]["traffic_source"] or "none recorded") + " [the --pmc passes of this run held no row of the extraction kernel]"
                else:
                    res["roofline"]["traffic_source"] = (res["roofline"]["traffic_source"] or "none recorded") + f" [the --pmc passes of this run gave nothing: {n_pmc}]"
            res["traffic_by_kernel"] = None if kernel_traffic is None else {k: v for k, v in sorted(kernel_traffic.items(), key=lambda kv: -(kv[1]["fetch_bytes"] + kv[1]["write_bytes"]))[:12]}
            fill_dominant(res, tm, steps, total_bases, shape, G, key_bytes, pieces_info, kernel_traffic, n_pmc if kernel_traffic else 0)
            res["other_kernels"] = other_kernels(tm, steps, total_bases, n_distinct, shape[0], shape[1], G, key_bytes, pieces_info[0], kernel_traffic)
            try:
                if e2e is not None:
                    res["end_to_end"] = e2e
                oarr = oaln = None
                if args.cpu_genomes > 0:
                    res["cpu_baseline"], oarr, oaln = cpu_baseline(args, files, n_total, td)
                if not args.no_check:
                    res["check"] = check_against_oracle(args, E, ctx, files, oarr, oaln, n_total, anc, synth)
                if "end_to_end" in res and "cpu_baseline" in res:
                    cpu = res["cpu_baseline"]["scaled_to_full_set_threads"]["value"]
                    res["vs_baseline"] = res["end_to_end"]["genomes_per_s"] / cpu
                    res["vs_baseline_basis"] = ("end_to_end.genomes_per_s (ska build + ska align through files, this run) / cpu_baseline.scaled_to_full_set_threads.value "
                                                "(CPU restatement of ska.rust on this box's host cores, same phases, extrapolated in the CPU's favour); "
                                                "BASELINE.md holds no published number for the metric")
                    res["vs_cpu_baseline_measured_sample"] = res["end_to_end"]["genomes_per_s"] / res["cpu_baseline"]["value"]
            finally:
                if td:
                    shutil.rmtree(td, ignore_errors=True)
    if sharded:
        dist.destroy_process_group()
    if rank == 0:
        # RCCL writes its version banner through C stdio, which is flushed at exit: push it out first so that the
        # JSON line is the last line of stdout
        import ctypes
        try:
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        sys.stdout.flush()
        print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
