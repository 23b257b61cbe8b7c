// main.cpp -- the `nimpress` command line (reference: src/nimpress.nim:652-757, docopt-driven).
// Same usage text, options, defaults, version string, exit codes and output format; the work is
// done by computePolygenicScores (nimpress_host.cpp) on top of libnps (HIP, MI355X).
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

#include "nimpress_host.hpp"

namespace nimpress {

// The usage text is the command-line contract of the reference (nim:653-706) and is reproduced as
// is: scripts and users see the same help, flags and defaults.
static const char *kDoc =
    "  Compute polygenic scores from a VCF/BCF.\n"
    "\n"
    "  Usage:\n"
    "    nimpress [options] <scoredef> <genotypes.vcf>\n"
    "    nimpress (-h | --help)\n"
    "    nimpress --version\n"
    "\n"
    "  Options:\n"
    "    -h --help          Show this screen.\n"
    "    --version          Show version.\n"
    "    --cov=<path>       Path to a BED file supplying genome regions that have been\n"
    "                       genotyped in the genotypes.vcf file.\n"
    "    --imp-locus=<m>    Imputation to apply for whole loci which are either not\n"
    "                       in the sequenced BED regions, or fail (too many samples \n"
    "                       with missing genotype, as set by --maxmis, or failing VCF\n"
    "                       QUAL field if --ignorefilt is not set). Valid values are \n"
    "                       ps, homref, fail, ignore [default: ps].\n"
    "    --imp-missing=<m>  Imputation to apply for loci which are in the sequenced BED\n"
    "                       regions (and thus should have been genotyped), but are \n"
    "                       completely missing from the VCF. Valid values are homref,\n"
    "                       ignore [default: homref].\n"
    "    --imp-sample=<m>   Imputation to apply for an individual sample with missing \n"
    "                       genotype. Valid values are ps, homref, fail, int_fail, \n"
    "                       int_ps [default: int_ps].\n"
    "    --maxmis=<f>       Maximum fraction of samples with missing genotypes allowed\n"
    "                       at a locus. Loci containing more than this fraction of \n"
    "                       samples missing will be considered bad, and have all \n"
    "                       genotypes (even non-missing ones) imputed [default: 0.05].\n"
    "    --mincs=<n>        Minimum number of genotypes.vcf samples without missing \n"
    "                       genotype at a locus for this locus to be eligible for \n"
    "                       internal imputation [default: 100].\n"
    "    --afmisp=<f>       p-value threshold for warning about allele frequency \n"
    "                       mismatch between the polygenic score and the supplied \n"
    "                       cohort [default: 0.001].\n"
    "    --ignorefilt       Ignore the VCF FILTER field. If set, all variants in the \n"
    "                       VCF will be used regardless of FILTER field contents. If\n"
    "                       not set, variants with a FILTER field other than \".\" or\n"
    "                       \"PASS\" will always be imputed.\n"
    "\n"
    "  Imputation methods:\n"
    "  ps        Impute with dosage based on the polygenic score effect allele \n"
    "            frequency.\n"
    "  homref    Impute to homozygous reference genotype.\n"
    "  fail      Do not impute, but fail. Failed samples will have a score of \"nan\"\n"
    "  ignore    Completely ignore missing loci, as if they were never in the score \n"
    "            definition.\n"
    "  int_ps    Impute with dosage calculated from non-missing samples in the \n"
    "            cohort. At least --mincs non-missing samples must be available for \n"
    "            this method to be used, else it will fall back to ps.\n"
    "  int_fail  Impute with dosage calculated from non-missing samples in the \n"
    "            cohort. At least --mincs non-missing samples must be available for \n"
    "            this method to be used, else it will fall back to fail.\n"
    "  ";

static void usageError() {
    fputs("Usage:\n  nimpress [options] <scoredef> <genotypes.vcf>\n  nimpress (-h | --help)\n"
          "  nimpress --version\n", stdout);
    exit(1);
}

int cliMain(int argc, char **argv) {
    std::map<std::string, std::string> opt = {{"--cov", ""},          {"--imp-locus", "ps"},
                                              {"--imp-missing", "homref"}, {"--imp-sample", "int_ps"},
                                              {"--maxmis", "0.05"},   {"--mincs", "100"},
                                              {"--afmisp", "0.001"}};
    bool have_cov = false, ignorefilt = false;
    std::vector<std::string> positional;
    for (int i = 1; i < argc; ++i) {
        std::string a = argv[i];
        if (a == "-h" || a == "--help") {
            puts(kDoc);
            return 0;
        }
        if (a == "--version") {
            puts("nimpress 1.0.0");
            return 0;
        }
        if (a == "--ignorefilt") {
            ignorefilt = true;
            continue;
        }
        if (a.size() > 2 && a[0] == '-' && a[1] == '-') {
            std::string key = a, val;
            const size_t eq = a.find('=');
            bool has_val = false;
            if (eq != std::string::npos) {
                key = a.substr(0, eq);
                val = a.substr(eq + 1);
                has_val = true;
            }
            if (!opt.count(key)) usageError();
            if (!has_val) {
                if (i + 1 >= argc) usageError();
                val = argv[++i];
            }
            opt[key] = val;
            if (key == "--cov") have_cov = true;
            continue;
        }
        if (a.size() > 1 && a[0] == '-') usageError();
        positional.push_back(a);
    }
    if (positional.size() != 2) usageError();

    Log log;
    double maxMissingRate, afMismatchPthresh;
    long long mincs;
    ImputeMethodLocus iml;
    ImputeMethodMissing imm;
    ImputeMethodSample ims;
    try {  // parseFloat / parseInt / parseEnum of nim:714-719 (errors are uncaught there: exit 1)
        char *end = nullptr;
        maxMissingRate = strtod(opt["--maxmis"].c_str(), &end);
        if (end == opt["--maxmis"].c_str() || *end) throw std::runtime_error("invalid float: " + opt["--maxmis"]);
        afMismatchPthresh = strtod(opt["--afmisp"].c_str(), &end);
        if (end == opt["--afmisp"].c_str() || *end) throw std::runtime_error("invalid float: " + opt["--afmisp"]);
        mincs = strtoll(opt["--mincs"].c_str(), &end, 10);
        if (end == opt["--mincs"].c_str() || *end) throw std::runtime_error("invalid integer: " + opt["--mincs"]);
        if (!parseEnum(opt["--imp-locus"], iml)) throw std::runtime_error("invalid enum value: " + opt["--imp-locus"]);
        if (!parseEnum(opt["--imp-missing"], imm)) throw std::runtime_error("invalid enum value: " + opt["--imp-missing"]);
        if (!parseEnum(opt["--imp-sample"], ims)) throw std::runtime_error("invalid enum value: " + opt["--imp-sample"]);
    } catch (const std::exception &ex) {
        fprintf(stderr, "Error: unhandled exception: %s [ValueError]\n", ex.what());
        return 1;
    }

    int device = 0;
    if (const char *d = getenv("NIMPRESS_DEVICE")) device = atoi(d);
    // the HIP context and libnps's code object come up on a thread of their own while the files are opened, inflated
    // and parsed (joined inside computePolygenicScores, before its first libnps call, and here on every other way out)
    warmupStart(device);
    struct Join {
        ~Join() { warmupJoin(); }
    } join_on_exit;
    const auto t_start = std::chrono::steady_clock::now();
    try {
        ScoreFile scoreFile;
        const bool score_ok = scoreFile.open(positional[0]);
        VCF vcf;
        // order of the reference: VCF first (nim:728), then the score file (nim:732)
        bool vcf_ok = false;
        {
            std::string err;
            try {
                // indexed files (vcf.gz + .tbi, BCF + .csi) are streamed window by window of score
                // rows; anything else is read whole, keeping the records of the score's loci
                vcf_ok = vcf.openStreaming(positional[1]) ||
                         vcf.open(positional[1], score_ok ? &scoreFile.entries : nullptr);
            } catch (const std::exception &ex) {
                err = ex.what();
            }
            if (!vcf_ok) {
                log.fatal("Could not open input VCF file " + positional[1] + (err.empty() ? "" : " (" + err + ")"));
                return 255;  // quit(-1)
            }
        }
        if (!score_ok) {
            log.fatal("Could not open polygenic score file " + positional[0]);
            return 255;
        }
        GenomeIntervals cov;
        bool restrict = false;
        if (have_cov) {
            restrict = true;
            if (!loadBedIntervals(cov, opt["--cov"]))  // logged, NOT fatal in the reference (nim:739-740)
                log.fatal("Could not open coverage BED file " + opt["--cov"]);
        }
        std::vector<double> scores;
        const double t_open = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();
        const double parse_during_open = timings().inflate_parse;  // (a file read whole is parsed by open(); a streamed one later)
        computePolygenicScores(scores, scoreFile, vcf, restrict, cov, iml, imm, ims, maxMissingRate,
                               afMismatchPthresh, mincs, ignorefilt, log, device);
        const auto t_write0 = std::chrono::steady_clock::now();
        for (size_t i = 0; i < scores.size(); ++i)  // nim:752-753
            printf("%s\t%s\n", vcf.samples[i].c_str(), formatFloat(scores[i]).c_str());
        if (getenv("NIMPRESS_TIMINGS")) {  // where the run's time went, one JSON line on stderr (bench.py reads it)
            fflush(stdout);
            const Timings &t = timings();
            fprintf(stderr,
                    "{\"nimpress_timings\": {\"hip_init_s\": %.4f, \"hip_init_wait_s\": %.4f, \"open_s\": %.4f, "
                    "\"inflate_parse_s\": %.4f, \"push_s\": %.4f, \"kernel_s\": %.4f, \"warnings_s\": %.4f, "
                    "\"write_s\": %.4f}}\n",
                    t.hip_init, t.hip_init_wait, t_open - parse_during_open, t.inflate_parse, t.push, t.kernels, t.warnings,
                    std::chrono::duration<double>(std::chrono::steady_clock::now() - t_write0).count());
        }
    } catch (const std::exception &ex) {
        fprintf(stderr, "Error: unhandled exception: %s\n", ex.what());
        return 1;
    }
    return 0;
}

}  // namespace nimpress

int main(int argc, char **argv) { return nimpress::cliMain(argc, argv); }
