// nimpress_host.hpp -- host side above the libnps C-ABI, mirroring the reference module's exported
// interface (src/nimpress.nim; cited as nim:LINE): ScoreFile/open/items (195-254), GenomeIntervals/
// loadBedIntervals/isVariantCovered (262-345), findVariant (353-364), the three enums (412-414) and
// computePolygenicScores (592-649).  The reference gets VCF access from hts-nim/htslib; neither is
// available here, so this file carries a small reader of its own (text VCF, plain or BGZF/gzip).
//
// The per-row arithmetic is NOT here: computePolygenicScores hands every located record's
// bcf_get_genotypes-layout buffer to libnps (HIP) and only keeps the reference's control flow,
// warnings and output formatting.
#pragma once
#include <cstdint>
#include <map>
#include <memory>
#include <string>
#include <utility>
#include <vector>

namespace nimpress {

// nim:412-414 (declaration order = the integer values the C-ABI takes)
enum class ImputeMethodLocus { ps = 0, homref = 1, fail = 2, ignore = 3 };
enum class ImputeMethodMissing { homref = 0, ignore = 1 };
enum class ImputeMethodSample { ps = 0, homref = 1, fail = 2, int_ps = 3, int_fail = 4 };
bool parseEnum(const std::string &s, ImputeMethodLocus &out);
bool parseEnum(const std::string &s, ImputeMethodMissing &out);
bool parseEnum(const std::string &s, ImputeMethodSample &out);

// nim:221-231
struct ScoreEntry {
    std::string contig;
    int64_t pos = 0;
    std::string refseq, easeq;
    double beta = 0.0, eaf = 0.0;
    int64_t stop() const { return pos + (int64_t)refseq.size() - 1; }
};

// nim:195-254: 5 header lines (name, description, citation, genome version, offset) then 6-column
// TSV rows.  open() returns false when the file cannot be opened; malformed content throws
// std::runtime_error (the reference asserts / raises).
struct ScoreFile {
    std::string name, desc, cite, genomever;
    double offset = 0.0;
    std::vector<ScoreEntry> entries;  // file order (the reference's iterator is single-pass)
    bool open(const std::string &path);
};

// nim:262-345.  BED: 0-based half-open; a locus is covered iff some interval has
// start < pos && stop >= pos + len(ref) - 1  (nim:310-311).
struct GenomeIntervals {
    bool init = false;
    std::map<std::string, std::vector<std::pair<int64_t, int64_t>>> contigIntervals;
};
bool loadBedIntervals(GenomeIntervals &ivals, const std::string &path);
bool isVariantCovered(const ScoreEntry &e, const GenomeIntervals &ivals, std::string *warning);

// What the reference reads through hts-nim: one record with its FORMAT/GT in the
// bcf_get_genotypes layout (n_samples*ploidy int32, (allele+1)<<1|phased, 0 = missing allele,
// 0x80000001 = vector-end pad).
struct Variant {
    std::string contig;
    int64_t pos = 0;
    std::string id, ref;
    std::vector<std::string> alt;
    std::string filter;  // as written in the file ("." / "PASS" / "a;b")
    int ploidy = 2;
    // GT as read: text VCF -> int32 in `gts` (gt_bytes = 4); BCF -> the record's typed vector as it
    // stands in the file (int8 / int16 / int32, gt_bytes = 1 / 2 / 4) in `gt_raw`, never widened on
    // the host (libnps decodes it on the device: nps_push_gt_raw)
    int gt_bytes = 4;
    std::vector<int32_t> gts;
    std::vector<uint8_t> gt_raw;
    bool has_gt = true;
    // FORMAT/DS (build-defined extension: the reference decodes GT only, nim:381-391): float32 ALT dosages,
    // ds_per_sample values per sample (1, or one per ALT allele for Number=A), NaN = missing.  A record's DS is used
    // INSTEAD of its GT when it has no GT, or when NIMPRESS_FORMAT=DS is set and it has DS (then GT is not kept).
    bool has_ds = false;
    int ds_per_sample = 0;
    std::vector<float> ds;
    // the dosage row the loop scores for effect allele index eaidx (0 = REF: the sum of the ALT dosages, which
    // nps_push_ds turns into 2 - sum; k = ALT[k-1]): `tmp` is used unless the record's own vector can be handed over
    const float *dsRow(int eaidx, size_t n_samples, std::vector<float> &tmp) const;
    // PLINK 1 .bed source (build-defined extension): gt_raw holds the variant's ceil(N/4) .bed bytes,
    // ref = A2 and alt = {A1} of the .bim line.  A .bim has no notion of REF, so findVariant also
    // accepts a score row whose ref is A1.
    bool is_bed = false;
    // PLINK 2 .pgen source, storage mode 0x02 only (fixed-width hard calls; build-defined extension, PARITY UNPINNED:
    // pinned by the build's own writer alone): gt_raw holds the variant's ceil(N/4) record bytes (2-bit code = number
    // of ALT alleles, 3 = missing), ref / alt = REF / ALT of the .pvar line; findVariant applies the VCF rule.
    bool is_pgen = false;
    const void *gtData() const { return gt_raw.empty() ? (const void *)gts.data() : (const void *)gt_raw.data(); }
    int32_t gtValue(size_t i) const;  // element i widened like bcf_get_genotypes does
};

struct IndexedSource;  // header + tabix / CSI index of an indexed file (nimpress_host.cpp)

struct VCF {
    std::vector<std::string> samples;
    std::vector<Variant> records;  // file order
    bool indexed = false;          // records were fetched through the tabix / CSI index
    // non-null for vcf.gz + .tbi and BCF + .csi: what hts-nim's vcf.query() needs (nim:358)
    std::shared_ptr<IndexedSource> source;
    // streaming: `records` stays empty and computePolygenicScores fetches the records of a window of
    // score rows at a time (memory = window x samples instead of loci x samples)
    bool streaming = false;
    // open(): text VCF (plain or gzip/BGZF; CRLF tolerant), BCF2 (BGZF) or a PLINK 1 fileset.  If
    // `keep` is non-null only records overlapping one of its loci are retained (memory = loci x
    // samples); with an index next to the file (.tbi for vcf.gz, .csi for BCF) only the chunks of
    // those loci are inflated, by several threads (NIMPRESS_THREADS, default: the host's cores, at
    // most 16).
    bool open(const std::string &path, const std::vector<ScoreEntry> *keep = nullptr);
    // openStreaming(): indexed files only (false otherwise): header + index, no records.
    bool openStreaming(const std::string &path);
    // the records overlapping entries [first, first+count), in file order (indexed files only)
    std::vector<Variant> fetch(const ScoreEntry *first, size_t count) const;
    int64_t n_samples() const { return (int64_t)samples.size(); }
};

// nim:353-364: first record overlapping contig:pos-stop with REF == ref and (ea == ref or ea in ALT).
const Variant *findVariant(const std::string &contig, int64_t pos, const std::string &refseq,
                           const std::string &easeq, const VCF &vcf);
// the same rule over a set of records with a per-contig position index (built once per set): the
// score driver's lookups are O(log records) instead of a scan
struct RecordIndex {
    struct Contig {
        std::vector<uint32_t> ids;  // records of the contig in file order
        bool sorted = true;         // ... which is position order (always, for indexed files)
        int64_t maxlen = 1;         // longest REF (or .bim allele): bounds the overlap window
    };
    std::map<std::string, Contig> contigs;
    const std::vector<Variant> *records = nullptr;
    void build(const std::vector<Variant> &recs);
    const Variant *find(const std::string &contig, int64_t pos, const std::string &refseq,
                        const std::string &easeq) const;
};

// nim:50-188 (only used for the AF-mismatch warnings)
double dbinom(int64_t x, int64_t n, double p);
double betai(double a, double b, double x);
double pbinom(int64_t x, int64_t n, double p);
double binomTest(int64_t x, int64_t n, double p);      // literal enumeration, O(n)
double binomTestFast(int64_t x, int64_t n, double p);  // same value by bisection, O(log n)

// Nim's `$float` as the reference prints it: "%.16g" plus ".0" when no '.', 'e', 'n', 'i' appears.
std::string formatFloat(double x);

// Where a run's wall time went (seconds; what is not listed is process start and output, which the caller sees).
// Filled by computePolygenicScores[Multi] and by the C hooks around them; reset by timingsReset().
struct Timings {
    double hip_init = 0;       // nps_warmup on its own thread: HIP context + code object
    double hip_init_wait = 0;  // ... of which the run had to wait for (the rest overlapped open + inflate + parse)
    double open = 0;           // score files, genotype file header, index
    double inflate_parse = 0;  // BGZF inflate + record parsing (all fetch threads, wall)
    double push = 0;           // nps_push_* / nps_cohort_push_* calls: pinned copy, PCIe, decode launches
    double kernels = 0;        // convert + score definitions + scoring + finish, until the scores are there
    double warnings = 0;       // the reference's per-row warnings (binomial tests) and log text
};
Timings &timings();
void timingsReset();
// the HIP context and libnps's code object on a thread of their own, while the caller opens and parses its files:
// start it before the first file is touched, join it (warmupJoin) before the first libnps call
void warmupStart(int device);
void warmupJoin();

struct Log {
    std::vector<std::string> lines;  // "WARN ..." / "FATAL ..." in emission order
    bool echo = true;                // also print to stdout like Nim's ConsoleLogger
    void warn(const std::string &m);
    void fatal(const std::string &m);
};

// nim:592-649.  Same parameters and meaning; `device` selects the GPU.  Throws std::runtime_error
// when libnps reports an error (there is no CPU fallback).  nloci_out (optional) = rows used.
void computePolygenicScores(std::vector<double> &scores, const ScoreFile &scoreFile,
                            const VCF &genotypeVcf, bool restrictToCoveredRgns,
                            const GenomeIntervals &coveredIvals, ImputeMethodLocus imputeMethodLocus,
                            ImputeMethodMissing imputeMethodMissing,
                            ImputeMethodSample imputeMethodSample, double maxMissingRate,
                            double afMismatchPthresh, int64_t minGtForInternalImput,
                            bool ignoreFilterField, Log &log, int device = 0,
                            uint64_t *nloci_out = nullptr, double *d_scores_out = nullptr);
// d_scores_out (optional, device memory, n_samples doubles): the scores are left on the device (nps_finish_device)
// and `scores` comes back empty -- what a multi-GPU gather wants (no host bounce).

// S score files against ONE genotype file in one pass over the genotypes (the reference runs
// its loop nim:634-641 once per file): the union of the files' loci is located once (findVariant), every located
// record is decoded on the device straight into a resident 2-bit cohort (nps_cohort_push_gt_raw / _push_bed), and
// the S definitions are applied together on the matrix cores (nps_score_cohort_multi), at most NPS_MULTI_MAX_SCORES
// at a time.  Results, nloci and warnings per file are those of computePolygenicScores run file by file (scores up
// to the 2^-49 quantisation of the weights).  Throws std::runtime_error (also for what this path does not cover:
// FORMAT/DS records, ploidy above 2, non-finite beta -- callers fall back to the per-file path).
void computePolygenicScoresMulti(std::vector<std::vector<double>> &scores, const std::vector<const ScoreFile *> &scoreFiles,
                                 const VCF &genotypeVcf, bool restrictToCoveredRgns, const GenomeIntervals &coveredIvals,
                                 ImputeMethodLocus imputeMethodLocus, ImputeMethodMissing imputeMethodMissing,
                                 ImputeMethodSample imputeMethodSample, double maxMissingRate, double afMismatchPthresh,
                                 int64_t minGtForInternalImput, bool ignoreFilterField, std::vector<Log> &logs,
                                 int device = 0, std::vector<uint64_t> *nloci_out = nullptr, int shard = 0,
                                 int n_shards = 1, bool partial = false, double *d_out = nullptr);
// d_out (optional, device memory, [files][n_samples] doubles): results stay on the device, `scores` stays empty.
// shard / n_shards / partial: the rows-sharded x all-scores layout over several GPUs.  The
// union's rows are cut into n_shards contiguous blocks; this call locates, decodes and scores block `shard` only
// (1 / n_shards of the ingest and of the cohort) for ALL files, and with partial = true returns the state of the
// reference's loop BEFORE its normalisation -- scores[s] = the un-normalised sums, nloci_out[s] = the block's count
// (nps_multi_partial) -- for the caller's sum all-reduce over the shards followed by sums / (2 nloci) + offset
// (nim:643-649).  Warnings are those of the block's rows.

// nim:652-757
int cliMain(int argc, char **argv);

}  // namespace nimpress
