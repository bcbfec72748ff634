"""`ska --help | --version | <cmd> --help | help <cmd>` (cli.rs:154 `#[command(author, version, about)]`, propagate_version; flags and
their descriptions cli.rs:168-459): answered on stdout with exit code 0 and without the banner, before any device is touched -- so
this runs on the CPU.  Every flag the reference declares for a subcommand must appear in that subcommand's help."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SKA = os.path.join(ROOT, "ska.rust_amd", "ska")

# cli.rs: the flags of each subcommand (global -v / --verbose, clap's -h / -V on all of them)
FLAGS = {
    "build": ["-f <FILE_LIST>", "-o <OUTPUT>", "-k <K>", "--proportion-reads", "--single-strand", "--min-count", "--min-qual", "--qual-filter", "--threads",
              "[default: 31]", "[default: 20]", "[default: strict]", "no-filter, middle, strict"],
    "align": ["-o <OUTPUT>", "-m, --min-freq <MIN_FREQ>", "--filter-ambig-as-missing", "--filter <FILTER>", "--ambig-mask", "--no-gap-only-sites", "--threads",
              "[default: 0.9]", "[default: no-const]", "no-filter, no-const, no-ambig, no-ambig-or-const"],
    "map": ["<REFERENCE>", "-o <OUTPUT>", "-f, --format <FORMAT>", "--ambig-mask", "--repeat-mask", "--threads", "[default: aln]", "vcf, aln"],
    "distance": ["<SKF_FILE>", "-o <OUTPUT>", "-m, --min-freq <MIN_FREQ>", "--allow-ambiguous", "--threads", "[default: 0]"],
    "merge": ["[SKF_FILES]...", "-o <OUTPUT>"],
    "delete": ["-s, --skf-file <SKF_FILE>", "-o <OUTPUT>", "-f <FILE_LIST>", "[NAMES]..."],
    "weed": ["<SKF_FILE>", "[WEED_FILE]", "-o <OUTPUT>", "--reverse", "-m, --min-freq <MIN_FREQ>", "--filter-ambig-as-missing", "--filter <FILTER>", "--ambig-mask",
             "--no-gap-only-sites", "[default: no-filter]"],
    "nk": ["<SKF_FILE>", "--full-info"],
    "cov": ["<FASTQ_FWD>", "<FASTQ_REV>", "-k <K>", "--single-strand"],
}


def _run(*args):
    return subprocess.run([SKA, *args], capture_output=True, text=True, timeout=60)


@pytest.mark.skipif(not os.path.exists(SKA), reason="ska executable not built")
def test_top_level_help_and_version():
    for flag in ("--help", "-h", "help"):
        r = _run(flag)
        assert r.returncode == 0 and r.stderr == "", (flag, r.stderr)
        assert r.stdout.startswith("Split k-mer analysis\n\nUsage: ska [OPTIONS] <COMMAND>\n")
        for cmd in FLAGS:
            assert f"\n  {cmd} " in r.stdout, cmd
        assert "-v, --verbose" in r.stdout and "-V, --version" in r.stdout
    for flag in ("--version", "-V"):
        r = _run(flag)
        assert (r.returncode, r.stdout, r.stderr) == (0, "ska 0.5.2\n", "")
    r = _run()                                          # clap: no subcommand = usage on stderr, exit code 2
    assert r.returncode == 2 and r.stdout == "" and "Usage: ska [OPTIONS] <COMMAND>" in r.stderr
    r = _run("help", "nonsense")
    assert r.returncode == 2 and "unrecognized subcommand 'nonsense'" in r.stderr


@pytest.mark.skipif(not os.path.exists(SKA), reason="ska executable not built")
@pytest.mark.parametrize("cmd", sorted(FLAGS))
def test_subcommand_help_lists_the_reference_flags(cmd):
    outs = set()
    for args in ((cmd, "--help"), (cmd, "-h"), ("help", cmd), (cmd, "x.skf", "--help")):
        r = _run(*args)
        assert r.returncode == 0 and r.stderr == "", (args, r.stderr)
        outs.add(r.stdout)
    assert len(outs) == 1
    out = outs.pop()
    assert out.split("\n")[2].startswith(f"Usage: ska {cmd} ")
    for f in FLAGS[cmd] + ["-v, --verbose", "-h, --help", "-V, --version"]:
        assert f in out, (cmd, f)
    r = _run(cmd, "--version")                          # propagate_version
    assert (r.returncode, r.stdout) == (0, f"ska-{cmd} 0.5.2\n")
