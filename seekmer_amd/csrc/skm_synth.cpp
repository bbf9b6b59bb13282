// Seeded synthetic transcriptome and read generator (libseekmer_host.so).
// The configs of BASELINE.json name the ENSEMBL GRCh38 cDNA index, which is not
// available offline; this is the stand-in SURVEY.md 8(d) specifies: genes x
// shared exons x isoforms, reads drawn from a log-normal abundance with
// N(200,25) fragments, 1 % substitutions and random mate swaps.  All draws are
// integer arithmetic on a counter-based generator keyed by (seed, unit), so
// any slice of the read set can be produced independently on any number of
// threads and is reproducible bit for bit.
#include "../../include/seekmer_hip.h"

#include <fcntl.h>
#include <unistd.h>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

namespace {

inline uint64_t mix64(uint64_t z)
{
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

struct Rng {
    uint64_t s;
    explicit Rng(uint64_t seed, uint64_t stream) : s(mix64(seed ^ mix64(stream + 0x9E3779B97F4A7C15ULL))) {}
    inline uint64_t next() { s += 0x9E3779B97F4A7C15ULL; return mix64(s); }
    inline uint32_t below(uint32_t n) { return (uint32_t)(((next() >> 32) * (uint64_t)n) >> 32); }
};

const char ALPHABET[4] = {'A', 'C', 'G', 'T'};

inline char complement(char c)
{
    switch (c) { case 'A': return 'T'; case 'C': return 'G'; case 'G': return 'C'; default: return 'A'; }
}

}  // namespace

// genes x 12 exons of U[80,420) bp; Binomial(190, 0.05) (mean 9.5) isoforms per
// gene, at least one; each isoform keeps each exon with probability 0.7 (at
// least one exon).
extern "C" int skm_synth_transcriptome(uint64_t seed, int64_t n_genes, int64_t *n_tx, char **pool,
                                       int64_t **offsets)
{
    if (n_genes <= 0 || !n_tx || !pool || !offsets) return SKM_ERR_ARG;
    std::vector<char> bases;
    std::vector<int64_t> off(1, 0);
    std::vector<char> exon[12];
    for (int64_t g = 0; g < n_genes; ++g) {
        Rng rng(seed, (uint64_t)g);
        for (auto &e : exon) {
            const int len = 80 + (int)rng.below(340);
            e.resize(len);
            for (int i = 0; i < len; ++i) e[i] = ALPHABET[rng.below(4)];
        }
        int isoforms = 0;
        for (int i = 0; i < 190; ++i) isoforms += rng.below(100) < 5;
        if (isoforms < 1) isoforms = 1;
        for (int t = 0; t < isoforms; ++t) {
            int kept = 0;
            for (int e = 0; e < 12; ++e) {
                if (rng.below(10) < 7) { bases.insert(bases.end(), exon[e].begin(), exon[e].end()); ++kept; }
            }
            if (!kept) {
                const int e = (int)rng.below(12);
                bases.insert(bases.end(), exon[e].begin(), exon[e].end());
            }
            off.push_back((int64_t)bases.size());
        }
    }
    *n_tx = (int64_t)off.size() - 1;
    *pool = (char *)malloc(bases.size() + 1);
    *offsets = (int64_t *)malloc(off.size() * sizeof(int64_t));
    if (!*pool || !*offsets) return SKM_ERR_STATE;
    memcpy(*pool, bases.data(), bases.size());
    (*pool)[bases.size()] = 0;
    memcpy(*offsets, off.data(), off.size() * sizeof(int64_t));
    return SKM_OK;
}

extern "C" int skm_synth_free(void *p)
{
    free(p);
    return SKM_OK;
}

// Units [first_unit, first_unit + n_units) of the read set `seed`: fixed read
// length; paired -> 2 reads per unit (mate 1, mate 2), else mate 1 only.
// `bases` receives n_units * (paired ? 2 : 1) * read_len bytes.
extern "C" int skm_synth_reads(uint64_t seed, const char *pool, const int64_t *tx_offsets,
                               int64_t n_tx, int64_t first_unit, int64_t n_units, int read_len,
                               int paired, int n_threads, char *bases)
{
    if (!pool || !tx_offsets || n_tx <= 0 || n_units < 0 || read_len < 25 || read_len > 299 || !bases)
        return SKM_ERR_ARG;
    // abundance ~ lognormal(0, 2) over transcripts of >= 300 bp, as an integer CDF
    std::vector<int64_t> eligible;
    std::vector<double> weight;
    for (int64_t t = 0; t < n_tx; ++t) {
        if (tx_offsets[t + 1] - tx_offsets[t] < 300) continue;
        Rng rng(seed ^ 0xABCDEF12345ULL, (uint64_t)t);
        const double u1 = ((double)(rng.next() >> 11) + 1.0) / 9007199254740993.0;
        const double u2 = (double)(rng.next() >> 11) / 9007199254740992.0;
        const double z = std::sqrt(-2.0 * std::log(u1)) * std::cos(6.283185307179586 * u2);
        eligible.push_back(t);
        weight.push_back(std::exp(2.0 * z));
    }
    if (eligible.empty()) return SKM_ERR_ARG;
    double total = 0;
    for (double w : weight) total += w;
    std::vector<uint64_t> cdf(weight.size());
    double run = 0;
    for (size_t i = 0; i < weight.size(); ++i) {
        run += weight[i];
        const double frac = run / total;
        cdf[i] = frac >= 1.0 ? ~0ULL : (uint64_t)(frac * 18446744073709551616.0);
    }
    cdf.back() = ~0ULL;

    const int mates = paired ? 2 : 1;
    auto work = [&](int64_t lo, int64_t hi) {
        std::vector<char> frag(300);
        for (int64_t u = lo; u < hi; ++u) {
            Rng rng(seed, (uint64_t)(first_unit + u) + (1ULL << 40));
            const uint64_t r = rng.next();
            const size_t pick = (size_t)(std::upper_bound(cdf.begin(), cdf.end(), r) - cdf.begin());
            const int64_t t = eligible[std::min(pick, eligible.size() - 1)];
            const int64_t tlen = tx_offsets[t + 1] - tx_offsets[t];
            // fragment ~ N(200, 25): Irwin-Hall sum of twelve 16-bit uniforms
            int64_t sum = 0;
            for (int i = 0; i < 12; ++i) sum += (int64_t)(rng.next() >> 48);
            int64_t flen = 200 + (25 * (sum - 6 * 65536)) / 65536;
            flen = std::max<int64_t>(read_len, std::min<int64_t>(299, flen));
            const int64_t start = (int64_t)(((rng.next() >> 32) * (uint64_t)(tlen - flen + 1)) >> 32);
            memcpy(frag.data(), pool + tx_offsets[t] + start, (size_t)flen);
            char *m1 = bases + (u * mates) * read_len;
            char *m2 = paired ? m1 + read_len : nullptr;
            const bool swap = rng.below(2) != 0;
            char *fwd = swap && paired ? m2 : m1;                 // forward-strand mate
            char *rev = swap && paired ? m1 : m2;                 // reverse-strand mate
            if (!paired && swap) {                                // single-ended: mate 1 of the swapped pair
                for (int i = 0; i < read_len; ++i) m1[i] = complement(frag[flen - 1 - i]);
            } else {
                memcpy(fwd, frag.data(), (size_t)read_len);
            }
            if (paired)
                for (int i = 0; i < read_len; ++i) rev[i] = complement(frag[flen - 1 - i]);
            // 1 % uniform substitutions
            for (int m = 0; m < mates; ++m) {
                char *p = m ? m2 : m1;
                for (int i = 0; i < read_len; ++i) {
                    if (rng.below(100) == 0) {
                        int cur = p[i] == 'A' ? 0 : p[i] == 'C' ? 1 : p[i] == 'G' ? 2 : 3;
                        p[i] = ALPHABET[(cur + 1 + (int)rng.below(3)) & 3];
                    }
                }
            }
        }
    };
    if (n_threads < 1) n_threads = 1;
    n_threads = (int)std::min<int64_t>(n_threads, std::max<int64_t>(1, n_units / 4096));
    if (n_threads == 1) {
        work(0, n_units);
    } else {
        std::vector<std::thread> pool_threads;
        for (int i = 0; i < n_threads; ++i)
            pool_threads.emplace_back(work, n_units * i / n_threads, n_units * (i + 1) / n_threads);
        for (auto &th : pool_threads) th.join();
    }
    return SKM_OK;
}

// Reads as FASTQ text, for measuring the path from files: unit u of the fixed-length batch
// `bases` ([n_units][mates][read_len], the layout skm_synth_reads fills) becomes the record
// "@r<first_unit + u, ten digits>/<mate>\n<bases>\n+\n<read_len x 'I'>\n" of path1 (mate 1) and
// path2 (mate 2, paired only).  Every record has the same size, so threads write disjoint ranges.
extern "C" int skm_synth_fastq_write(const char *bases, int64_t n_units, int read_len, int paired,
                                     int64_t first_unit, const char *path1, const char *path2,
                                     int n_threads)
{
    if (!bases || n_units < 0 || read_len < 1 || !path1 || (paired && !path2)) return SKM_ERR_ARG;
    const int mates = paired ? 2 : 1;
    const size_t record = (size_t)(2 * read_len + 19);
    int fds[2] = {-1, -1};
    for (int m = 0; m < mates; ++m) {
        fds[m] = open(m ? path2 : path1, O_CREAT | O_TRUNC | O_WRONLY, 0644);
        if (fds[m] < 0 || ftruncate(fds[m], (off_t)(record * (size_t)n_units)) != 0) {
            for (int k = 0; k <= m; ++k) if (fds[k] >= 0) close(fds[k]);
            return SKM_ERR_IO;
        }
    }
    bool failed = false;
    auto work = [&](int64_t lo, int64_t hi) {
        constexpr int64_t CHUNK = 8192;
        std::vector<char> buf((size_t)CHUNK * record);
        for (int m = 0; m < mates; ++m) {
            for (int64_t first = lo; first < hi; first += CHUNK) {
                const int64_t last = std::min(hi, first + CHUNK);
                char *p = buf.data();
                for (int64_t u = first; u < last; ++u) {
                    int64_t id = first_unit + u;
                    p[0] = '@'; p[1] = 'r';
                    for (int d = 11; d >= 2; --d) { p[d] = (char)('0' + id % 10); id /= 10; }
                    p[12] = '/'; p[13] = (char)('1' + m); p[14] = '\n';
                    memcpy(p + 15, bases + ((size_t)u * mates + m) * read_len, (size_t)read_len);
                    p[15 + read_len] = '\n'; p[16 + read_len] = '+'; p[17 + read_len] = '\n';
                    memset(p + 18 + read_len, 'I', (size_t)read_len);
                    p[18 + 2 * read_len] = '\n';
                    p += record;
                }
                const size_t bytes = (size_t)(last - first) * record;
                if (pwrite(fds[m], buf.data(), bytes, (off_t)((size_t)first * record)) != (ssize_t)bytes)
                    failed = true;
            }
        }
    };
    if (n_threads < 1) n_threads = 1;
    n_threads = (int)std::min<int64_t>(n_threads, std::max<int64_t>(1, n_units / 8192));
    std::vector<std::thread> pool_threads;
    for (int i = 0; i < n_threads; ++i)
        pool_threads.emplace_back(work, n_units * i / n_threads, n_units * (i + 1) / n_threads);
    for (auto &th : pool_threads) th.join();
    for (int m = 0; m < mates; ++m) if (close(fds[m]) != 0) failed = true;
    return failed ? SKM_ERR_IO : SKM_OK;
}
