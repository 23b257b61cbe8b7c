// Test driver for the host ingest (no GPU, no libnps call): opens a genotype file the three ways the
// host can -- whole file, index with all loci at once, streaming windows -- and prints one line per
// score row: the record findVariant / RecordIndex::find picks (position, FILTER, ploidy, a checksum of
// the GT values as bcf_get_genotypes would hand them out).  Built with sanitizers by
// tests/test_host_logic.py::test_ingest_under_sanitizers.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>

#include "../../nimpress_amd/csrc/host/nimpress_host.hpp"

using namespace nimpress;

static unsigned long long checksum(const Variant &v, size_t n_samples) {
    unsigned long long h = 1469598103934665603ull;
    if (v.has_ds) {  // a record scored from FORMAT/DS: its dosage values as they were parsed
        for (const float f : v.ds) {
            unsigned int u;
            memcpy(&u, &f, 4);
            h = (h ^ (unsigned long long)u) * 1099511628211ull;
        }
        return h;
    }
    const size_t nval = v.is_bed || v.is_pgen ? 2 * n_samples
                        : v.gt_raw.empty()    ? v.gts.size()
                                              : v.gt_raw.size() / (size_t)v.gt_bytes;
    for (size_t i = 0; i < nval; ++i) h = (h ^ (unsigned long long)(unsigned int)v.gtValue(i)) * 1099511628211ull;
    return h;
}

static void dump(const char *tag, const ScoreFile &sf, const std::vector<Variant> &recs, size_t n_samples) {
    RecordIndex idx;
    idx.build(recs);
    for (const ScoreEntry &e : sf.entries) {
        const Variant *v = idx.find(e.contig, e.pos, e.refseq, e.easeq);
        if (!v)
            printf("%s %s:%lld absent\n", tag, e.contig.c_str(), (long long)e.pos);
        else
            printf("%s %s:%lld pos=%lld filter=%s ploidy=%d gt=%016llx\n", tag, e.contig.c_str(), (long long)e.pos,
                   (long long)v->pos, v->filter.c_str(), v->ploidy, checksum(*v, n_samples));
    }
}

static int run(int argc, char **argv);
int main(int argc, char **argv) {
    try {
        return run(argc, argv);
    } catch (const std::exception &e) {  // a file the readers refuse: a message and a status, never a crash
        printf("refused: %s\n", e.what());
        return 5;
    }
}
static int run(int argc, char **argv) {
    if (argc < 3) return 2;
    ScoreFile sf;
    if (!sf.open(argv[1])) return 3;
    const size_t window = argc > 3 ? (size_t)atoi(argv[3]) : 3;
    {
        VCF vcf;
        if (!vcf.open(argv[2], &sf.entries)) return 4;
        dump("A", sf, vcf.records, vcf.samples.size());
    }
    {
        VCF vcf;
        if (vcf.openStreaming(argv[2])) {
            for (size_t a = 0; a < sf.entries.size(); a += window) {
                ScoreFile part = sf;
                part.entries.assign(sf.entries.begin() + (long)a,
                                    sf.entries.begin() + (long)std::min(sf.entries.size(), a + window));
                const std::vector<Variant> recs = vcf.fetch(part.entries.data(), part.entries.size());
                dump("A", part, recs, vcf.samples.size());
            }
        } else {
            printf("no index\n");
        }
    }
    return 0;
}
