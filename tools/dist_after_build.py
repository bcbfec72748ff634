"""tools/dist_after_build.py: rounds of `ska build` + `ska distance x.skf` + the same distance again (1 000 x 5 Mbp, k = 41 then 31), with the phases that say
where a slow process lost its seconds: hipMalloc in all threads (memory a process has just released), the loaders' stager (pinning, reads).  NOTEBOOK round 6."""
import os, subprocess, sys, time, json, tempfile, shutil
ROOT = "/root/repo"
sys.path.insert(0, os.path.join(ROOT, "ska.rust_amd"))
import synth
n = 1000
td = tempfile.mkdtemp(dir="/dev/shm")
anc = synth.ancestor(5_000_000, seed=1)
files = []
for i in range(n):
    p = os.path.join(td, f"g{i}.fa"); synth.to_fasta(synth.sample_stream(anc, i, n), p); files.append(p)
open(os.path.join(td, "list.txt"), "w").write("".join(f"g{i}\t{p}\n" for i, p in enumerate(files)))
SKA = os.path.join(ROOT, "ska.rust_amd", "ska")
def run(args, knobs="", extra=None):
    env = dict(os.environ, SKX_PHASES=os.path.join(td, "ph.json"), SKX_KNOBS=knobs, **(extra or {}))
    t = time.perf_counter(); r = subprocess.run([SKA, *args], cwd=td, capture_output=True, env=env); dt = time.perf_counter() - t
    assert r.returncode == 0, r.stderr[-300:]
    return dt, json.load(open(os.path.join(td, "ph.json")))
for k in ("41", "31"):
    for tag, knobs, pause in (("default", "", 0), ("default", "", 0), ("default", "", 0)):
        if pause == -3:
            for f in ("all.skf", "d.tsv"):
                if os.path.exists(os.path.join(td, f)): os.unlink(os.path.join(td, f))
        dt, ph = run(["build", "-f", "list.txt", "-o", "all", "-k", k, "--threads", "32"], "", {"SKX_KEEP_TEARDOWN": "1"} if pause == -2 else None)
        if pause > 0: time.sleep(pause)
        if pause == -1:
            t = time.perf_counter(); subprocess.run(["cat", os.path.join(td, "all.skf")], stdout=subprocess.DEVNULL); print("  cat all.skf: %.2f s (%.1f GB)" % (time.perf_counter() - t, os.path.getsize(os.path.join(td, "all.skf")) / 1e9))
        d1, p1 = run(["distance", "all.skf", "-o", "d.tsv"], knobs)
        d2, p2 = run(["distance", "all.skf", "-o", "d.tsv"], knobs)
        print(f"k={k} build {dt:.2f} s (hipMalloc {ph.get('alloc.hipMalloc_all_threads', 0):.2f}); distance [{tag}] first {d1:.2f} s (decode_filter {p1.get('load.stream_decode_filter', 0):.2f}, ctx {p1.get('main.device_context', 0):.2f}, hipMalloc {p1.get('alloc.hipMalloc_all_threads', 0):.2f}, reads {p1.get('load.stager_reads_wall', 0):.2f}), again {d2:.2f} s (decode_filter {p2.get('load.stream_decode_filter', 0):.2f})", flush=True)
shutil.rmtree(td)
