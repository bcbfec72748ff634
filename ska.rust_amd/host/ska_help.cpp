// ska_help.cpp -- `ska --help | -h | help [cmd]`, `ska <cmd> --help`, `ska --version | -V`: what clap derives from the reference's
// argument structs (cli.rs:154 `#[command(author, version, about)]`, `propagate_version`; per-subcommand arguments cli.rs:168-459),
// laid out the way clap 4 lays its help out (about, usage, arguments, options with defaults and possible values; printed on stdout,
// exit code 0, no banner: the reference parses its arguments before it prints anything, lib.rs:558-565).  The version is the
// reference's (Cargo.toml:3) -- this executable stands in for that release of `ska`; the engine's own extensions are marked.
#include "../../include/skx_host.h"
#include "../../include/skx.h"

#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

namespace {

struct Opt { const char *flag; const char *value; const char *help; };
struct Cmd {
    const char *name, *about, *usage;
    std::vector<Opt> args, opts;
};

const char *const THREADS = "Number of CPU threads [default: 1]";

const std::vector<Cmd> &commands()
{
    static const std::vector<Cmd> C = {
        {"build", "Create a split-kmer file from input sequences", "ska build [OPTIONS] -o <OUTPUT> <SEQ_FILES|-f <FILE_LIST>>",
         {{"[SEQ_FILES]...", "", "List of input FASTA files"}},
         {{"-f", "<FILE_LIST>", "File listing input files (tab separated name, sequences)"},
          {"-o", "<OUTPUT>", "Output prefix"},
          {"-k", "<K>", "K-mer size [default: 31]"},
          {"--proportion-reads", "<PROPORTION_READS>", "Number of reads before stopping"},
          {"--single-strand", "", "Ignore reverse complement (all contigs are oriented along same strand)"},
          {"--min-count", "<MIN_COUNT>", "Minimum k-mer count (with reads)"},
          {"--min-qual", "<MIN_QUAL>", "Minimum k-mer quality (with reads) [default: 20]"},
          {"--qual-filter", "<QUAL_FILTER>", "Quality filtering criteria (with reads) [default: strict] [possible values: no-filter, middle, strict]"},
          {"--threads", "<THREADS>", THREADS},
          {"--gpus", "<GPUS>", "(MI355X engine) Number of GPUs: one rank each, samples sharded by rank"},
          {"--merge", "", "(MI355X engine) With --gpus, join the per-rank files into one"}}},
        {"align", "Write an unordered alignment", "ska align [OPTIONS] <INPUT>...",
         {{"<INPUT>...", "", "A .skf file, or list of .fasta files"}},
         {{"-o", "<OUTPUT>", "Output filename (omit to output to stdout)"},
          {"-m, --min-freq", "<MIN_FREQ>", "Minimum fraction of samples a k-mer has to appear in [default: 0.9]"},
          {"--filter-ambig-as-missing", "", "With min_freq, only count non-ambiguous sites"},
          {"--filter", "<FILTER>", "Filter for constant middle base sites [default: no-const] [possible values: no-filter, no-const, no-ambig, no-ambig-or-const]"},
          {"--ambig-mask", "", "Mask any ambiguous bases in the alignment with 'N'"},
          {"--no-gap-only-sites", "", "Ignore gaps '-' in constant sites (for low coverage samples)"},
          {"--threads", "<THREADS>", THREADS},
          {"--gpus", "<GPUS>", "(MI355X engine) Number of GPUs (with sequence files or -f and the build options)"}}},
        {"map", "Write an ordered alignment using a reference sequence", "ska map [OPTIONS] <REFERENCE> [INPUT]...",
         {{"<REFERENCE>", "", "Reference FASTA file to map to"}, {"[INPUT]...", "", "A .skf file, or list of .fasta files"}},
         {{"-o", "<OUTPUT>", "Output filename (omit to output to stdout)"},
          {"-f, --format", "<FORMAT>", "Format of output file [default: aln] [possible values: vcf, aln]"},
          {"--ambig-mask", "", "Mask any ambiguous bases in the alignment with 'N'"},
          {"--repeat-mask", "", "Mask any repeats in the alignment with 'N'"},
          {"--threads", "<THREADS>", THREADS}}},
        {"distance", "Calculate SNP distances and k-mer mismatches", "ska distance [OPTIONS] <SKF_FILE>",
         {{"<SKF_FILE>", "", "Split-kmer (.skf) file to operate on"}},
         {{"-o", "<OUTPUT>", "Output filename (omit to output to stdout)"},
          {"-m, --min-freq", "<MIN_FREQ>", "Minimum fraction of samples a k-mer has to appear in across the entire alignment [default: 0]"},
          {"--allow-ambiguous", "", "Don't filter out ambiguous bases and compute fractional distances"},
          {"--threads", "<THREADS>", THREADS},
          {"--gpus", "<GPUS>", "(MI355X engine) Number of GPUs (with sequence files or -f and the build options)"}}},
        {"merge", "Combine multiple split k-mer files", "ska merge -o <OUTPUT> [SKF_FILES]...",
         {{"[SKF_FILES]...", "", "List of input split-kmer (.skf) files"}},
         {{"-o", "<OUTPUT>", "Output prefix"}}},
        {"delete", "Remove samples from a split k-mer file", "ska delete [OPTIONS] --skf-file <SKF_FILE> <-f <FILE_LIST>|NAMES>",
         {{"[NAMES]...", "", "List of sample names to remove"}},
         {{"-s, --skf-file", "<SKF_FILE>", "Split-kmer (.skf) file to operate on"},
          {"-o", "<OUTPUT>", "Output name. If not provided, will overwrite the input file"},
          {"-f", "<FILE_LIST>", "File listing sample names to remove"}}},
        {"weed", "Remove k-mers from a split k-mer file", "ska weed [OPTIONS] <SKF_FILE> [WEED_FILE]",
         {{"<SKF_FILE>", "", "Split-kmer (.skf) file to operate on"}, {"[WEED_FILE]", "", "A FASTA file containing sequences to remove"}},
         {{"-o", "<OUTPUT>", "Output filename (omit to overwrite input file)"},
          {"--reverse", "", "Remove k-mers not in the weed_file"},
          {"-m, --min-freq", "<MIN_FREQ>", "Minimum fraction of samples a k-mer has to appear in [default: 0.9]"},
          {"--filter-ambig-as-missing", "", "With min_freq, only count non-ambiguous sites"},
          {"--filter", "<FILTER>", "Filter for constant middle base sites [default: no-filter] [possible values: no-filter, no-const, no-ambig, no-ambig-or-const]"},
          {"--ambig-mask", "", "Mask any ambiguous bases in the alignment with 'N'"},
          {"--no-gap-only-sites", "", "Ignore gaps '-' in constant sites"}}},
        {"nk", "Get the number of k-mers in a split k-mer file, and other information", "ska nk [OPTIONS] <SKF_FILE>",
         {{"<SKF_FILE>", "", "Split-kmer (.skf) file to operate on"}},
         {{"--full-info", "", "Also write out split-kmers, and middle base matrix"}}},
        {"cov", "Estimate a coverage cutoff using a k-mer count profile (FASTQ only)", "ska cov [OPTIONS] <FASTQ_FWD> <FASTQ_REV>",
         {{"<FASTQ_FWD>", "", "FASTQ file (or .fastq.gz) with forward reads"}, {"<FASTQ_REV>", "", "FASTQ file (or .fastq.gz) with reverse reads"}},
         {{"-k", "<K>", "K-mer size [default: 31]"},
          {"--single-strand", "", "Ignore reverse complement (all reads are oriented along same strand)"}}},
    };
    return C;
}

// clap's two-column form: flags with a short name start at column 2, long-only flags at column 6
std::string left_of(const Opt &o)
{
    std::string f = o.flag;
    std::string s = (f.size() > 1 && f[0] == '-' && f[1] == '-') ? "      " + f : "  " + f;
    if (o.value && *o.value) s += std::string(" ") + o.value;
    return s;
}
void table(FILE *out, const char *title, const std::vector<Opt> &rows)
{
    if (rows.empty()) return;
    size_t w = 0;
    std::vector<std::string> left;
    for (auto &r : rows) { left.push_back(left_of(r)); if (left.back().size() > w) w = left.back().size(); }
    fprintf(out, "\n%s:\n", title);
    for (size_t i = 0; i < rows.size(); i++) fprintf(out, "%-*s  %s\n", (int)w, left[i].c_str(), rows[i].help);
}
void command_help(FILE *out, const Cmd &c)
{
    fprintf(out, "%s\n\nUsage: %s\n", c.about, c.usage);
    table(out, "Arguments", c.args);
    std::vector<Opt> o = c.opts;
    o.push_back({"-v, --verbose", "", "Show progress messages"});
    o.push_back({"-h, --help", "", "Print help"});
    o.push_back({"-V, --version", "", "Print version"});
    table(out, "Options", o);
}
void top_help(FILE *out)
{
    fprintf(out, "Split k-mer analysis\n\nUsage: ska [OPTIONS] <COMMAND>\n\nCommands:\n");
    size_t w = 4;
    for (auto &c : commands()) if (strlen(c.name) > w) w = strlen(c.name);
    for (auto &c : commands()) fprintf(out, "  %-*s  %s\n", (int)w, c.name, c.about);
    fprintf(out, "  %-*s  %s\n", (int)w, "help", "Print this message or the help of the given subcommand(s)");
    std::vector<Opt> o = {{"-v, --verbose", "", "Show progress messages"}, {"-h, --help", "", "Print help"}, {"-V, --version", "", "Print version"}};
    table(out, "Options", o);
}

}  // namespace

// the usage line clap prints under an error of subcommand `cmd` ("ska <cmd> ..."; the top-level one for an unknown name)
const char *skh_usage_line(const char *cmd)
{
    for (auto &c : commands()) if (!strcmp(cmd, c.name)) return c.usage;
    return "ska [OPTIONS] <COMMAND>";
}

// 1: the invocation was a help / version request and has been answered (exit code 0); 0: not one; 2: `ska help <unknown>`
extern "C" int skh_help(int argc, char **argv)
{
    if (argc < 2) return 0;
    const std::string first = argv[1];
    auto is_help = [](const char *s) { return !strcmp(s, "--help") || !strcmp(s, "-h"); };
    auto is_version = [](const char *s) { return !strcmp(s, "--version") || !strcmp(s, "-V"); };
    auto find = [](const std::string &n) -> const Cmd * { for (auto &c : commands()) if (n == c.name) return &c; return nullptr; };
    if (is_help(argv[1])) { top_help(stdout); fflush(stdout); return 1; }
    if (is_version(argv[1])) { printf("ska %s\n", skx_version()); fflush(stdout); return 1; }
    if (first == "help") {
        if (argc < 3) { top_help(stdout); fflush(stdout); return 1; }
        if (const Cmd *c = find(argv[2])) { command_help(stdout, *c); fflush(stdout); return 1; }
        fprintf(stderr, "error: unrecognized subcommand '%s'\n\nUsage: ska [OPTIONS] <COMMAND>\n\nFor more information, try '--help'.\n", argv[2]);
        return 2;
    }
    const Cmd *c = find(first);
    for (int i = 2; i < argc; i++) {
        if (!strcmp(argv[i], "--")) break;
        if (is_help(argv[i])) { if (c) command_help(stdout, *c); else top_help(stdout); fflush(stdout); return 1; }
        if (is_version(argv[i])) { printf("ska-%s %s\n", c ? c->name : first.c_str(), skx_version()); fflush(stdout); return 1; }       // propagate_version
    }
    return 0;
}
