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


@pytest.mark.skipif(not os.path.exists(SKA), reason="ska executable not built")
def test_refusals_in_claps_wording_before_any_device():
    """What clap refuses for the reference (cli.rs: required arguments, the argument groups of build / delete, value_parser and value_enum
    of the options) is refused here in the same words, with exit code 2, without the banner and before a device is opened -- so on the CPU."""
    miss = "error: the following required arguments were not provided:\n  {}\n\nUsage: {}\n\nFor more information, try '--help'.\n"
    cases = [
        (["build", "a.fa"], miss.format("-o <OUTPUT>", "ska build [OPTIONS] -o <OUTPUT> <SEQ_FILES|-f <FILE_LIST>>")),
        (["build", "-o", "x"], miss.format("<SEQ_FILES|-f <FILE_LIST>>", "ska build [OPTIONS] -o <OUTPUT> <SEQ_FILES|-f <FILE_LIST>>")),
        (["align"], miss.format("<INPUT>...", "ska align [OPTIONS] <INPUT>...")),
        (["distance"], miss.format("<SKF_FILE>", "ska distance [OPTIONS] <SKF_FILE>")),
        (["nk"], miss.format("<SKF_FILE>", "ska nk [OPTIONS] <SKF_FILE>")),
        (["merge", "a.skf", "b.skf"], miss.format("-o <OUTPUT>", "ska merge -o <OUTPUT> [SKF_FILES]...")),
        (["delete", "name"], miss.format("--skf-file <SKF_FILE>", "ska delete [OPTIONS] --skf-file <SKF_FILE> <-f <FILE_LIST>|NAMES>")),
        (["weed"], miss.format("<SKF_FILE>", "ska weed [OPTIONS] <SKF_FILE> [WEED_FILE]")),
        (["map"], miss.format("<REFERENCE>", "ska map [OPTIONS] <REFERENCE> [INPUT]...")),
        (["cov", "a.fq"], miss.format("<FASTQ_REV>", "ska cov [OPTIONS] <FASTQ_FWD> <FASTQ_REV>")),
        (["build", "-o", "x", "-k", "4", "a.fa"], "error: invalid value '4' for '-k <K>': K-mer must be an odd number between 5 and 63 (inclusive)\n\nFor more information, try '--help'.\n"),
        (["build", "-o", "x", "-k", "65", "a.fa"], "error: invalid value '65' for '-k <K>': K-mer must be an odd number between 5 and 63 (inclusive)\n\nFor more information, try '--help'.\n"),
        (["build", "-o", "x", "-k", "abc", "a.fa"], "error: invalid value 'abc' for '-k <K>': `abc` isn't a valid k-mer\n\nFor more information, try '--help'.\n"),
        (["build", "-o", "x", "--min-count", "0", "a.fa"], "error: invalid value '0' for '--min-count <MIN_COUNT>': Minimum kmer count must be >= 1\n\nFor more information, try '--help'.\n"),
        (["build", "-o", "x", "--qual-filter", "soft", "a.fa"], "error: invalid value 'soft' for '--qual-filter <QUAL_FILTER>'\n  [possible values: no-filter, middle, strict]\n\nFor more information, try '--help'.\n"),
        (["align", "x.skf", "--filter", "none"], "error: invalid value 'none' for '--filter <FILTER>'\n  [possible values: no-filter, no-const, no-ambig, no-ambig-or-const]\n\nFor more information, try '--help'.\n"),
        (["align", "x.skf", "-m", "1.5"], "error: invalid value '1.5' for '--min-freq <MIN_FREQ>': Frequency must be between 0 and 1 (inclusive)\n\nFor more information, try '--help'.\n"),
        (["weed", "x.skf", "--min-freq", "-0.1"], "error: invalid value '-0.1' for '--min-freq <MIN_FREQ>': Frequency must be between 0 and 1 (inclusive)\n\nFor more information, try '--help'.\n"),
        (["map", "ref.fa", "x.skf", "-f", "txt"], "error: invalid value 'txt' for '--format <FORMAT>'\n  [possible values: vcf, aln]\n\nFor more information, try '--help'.\n"),
        (["align", "x.skf", "--threads", "0"], "error: invalid value '0' for '--threads <THREADS>': Threads must be one or higher\n\nFor more information, try '--help'.\n"),
        (["frobnicate"], "error: unrecognized subcommand 'frobnicate'\n\nUsage: ska [OPTIONS] <COMMAND>\n\nFor more information, try '--help'.\n"),
        (["build", "-o", "x", "-f", "list.tsv", "a.fa"], "error: the argument '[SEQ_FILES]...' cannot be used with '-f <FILE_LIST>'\n\nUsage: ska build [OPTIONS] -o <OUTPUT> <SEQ_FILES|-f <FILE_LIST>>\n\nFor more information, try '--help'.\n"),
    ]
    for args, want in cases:
        r = _run(*args)
        assert (r.returncode, r.stdout, r.stderr) == (2, "", want), (args, r.stderr)
    r = _run("align", "x.skf", "--bogus")
    assert r.returncode == 2 and r.stderr.startswith("error: unexpected argument '--bogus' found\n") and "SKA:" not in r.stderr


@pytest.mark.skipif(not os.path.exists(SKA), reason="ska executable not built")
def test_the_references_warnings_in_simple_loggers_form():
    """Warnings the reference logs whatever -v says (simple_logger at Warn: lib.rs:559-563): cli.rs:86-91 (more threads than cores),
    io_utils.rs:66-73 (--threads with a single .skf), merge_ska_array.rs:298-300 (--no-gap-only-sites without a constant-site filter); -v adds
    the Info lines.  They come behind the banner and before the device is opened, so they can be read here (the command itself then fails on
    a box without a GPU)."""
    import re
    stamp = r"\d{4}-\d\d-\d\dT\d\d:\d\d:\d\d\.\d{3}Z "
    r = _run("align", "x.skf", "--threads", "4", "--no-gap-only-sites", "--filter", "no-ambig")
    lines = r.stderr.splitlines()
    assert lines[0] == "SKA: Split K-mer Analysis (the alignment-free aligner)"
    assert re.fullmatch(stamp + r"WARN  \[ska::io_utils\] --threads only used if building skf, setting to 1", lines[1]), lines[:4]
    assert re.fullmatch(stamp + r"WARN  \[ska::merge_ska_array\] --no-gap-only-sites can only be applied when filtering constant bases", lines[2]), lines[:4]
    r = _run("build", "-o", "x", "-k", "41", "-v", "--threads", "100000", "a.fa")
    lines = r.stderr.splitlines()
    assert re.fullmatch(stamp + r"WARN  \[ska::cli\] 100000 threads is greater than available cores \d+", lines[1]), lines[:4]
    assert re.fullmatch(stamp + r"INFO  \[ska\] k=41: using 128-bit representation", lines[2]), lines[:4]
    r = _run("build", "-o", "x", "a.fa")                               # without -v: no Info line
    assert not any(" INFO " in l for l in r.stderr.splitlines())
