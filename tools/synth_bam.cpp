// Synthetic coordinate-sorted BAM (+ .bai) and its reference FASTA (+ .fai) of any size, written at memory speed:
// bench / profiling data for image generation (bench.py secondary.make_images, tools/bench_variant_images.py).
// Not product code and not an oracle: the package's own BAM reader and the tests' Python writer define the format checks.
//
//   synth_bam <out_dir> <genome_bases> [coverage=60] [seed=2027] [threads=0 (all)] [contigs=1] [level=1] [tags=0] [quals=0]
//
// level: 1 = libdeflate level 1 (the fast default: what the data of rounds 3-4 was written with); 2..9 = zlib's deflate at that
// level -- 6 is what samtools / htslib write (longer matches, longer codes: the blocks a real BAM holds).  tags = 1: every record
// carries NM:i, an MD:Z string spelling out its mismatches and deletions, and RG:Z, as aligners write them (the aux data is a
// fifth of a real record's bytes and compresses unlike the bases).  quals = 1: quality strings with run-length structure (binned
// values in plateaus, an ONT-like Q7 - Q30 histogram) instead of ~5 bits of entropy per base: members then compress 3-4 x, as the
// BAMs of a binning basecaller do ("realistic" in bench.py's legs); quals = 0 is the harder case for the inflater per output byte.
//
// Reads as pepper_amd.synthetic.encoder_region models them (E-syn): 4-12 kb, an insert or a deletion of 1-5 bases every ~50
// positions, 4 % substitutions, a heterozygous SNP site per ~1 kb and a systematic indel site per ~700 b carried by one of
// two haplotypes; qualities with ~5 bits of entropy per base (so that the BGZF blocks inflate at the rate real ones do;
// constant qualities would flatter the reader); a few mapq-0 / duplicate / secondary / supplementary records.
// The genome is cut into pieces of 1 Mb; a thread generates, serialises and deflates the records that START in its piece
// (libdeflate level 1 where the header is there, zlib otherwise); pieces are concatenated in order and the index is computed
// from the block sizes.
#include <zlib.h>
#ifdef PA_HAVE_LIBDEFLATE
#include <libdeflate.h>
#endif

#include <algorithm>
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <thread>
#include <vector>

namespace {

struct Rng {
    uint64_t s;
    explicit Rng(uint64_t seed) : s(seed * 0x9e3779b97f4a7c15ull + 0x632be59bd9b4e019ull) {}
    uint64_t next() {
        uint64_t z = (s += 0x9e3779b97f4a7c15ull);
        z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
        z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
        return z ^ (z >> 31);
    }
    uint32_t below(uint32_t n) { return (uint32_t)((next() >> 32) * (uint64_t)n >> 32); }
    double unit() { return (next() >> 11) * (1.0 / 9007199254740992.0); }
};

inline uint64_t site_hash(uint64_t contig, uint64_t pos, uint64_t seed) {
    uint64_t z = (pos + 0x9e3779b97f4a7c15ull * (contig + 1)) ^ (seed * 0xd6e8feb86659fd93ull);
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}

int reg2bin(int64_t beg, int64_t end) {
    --end;
    if (beg >> 14 == end >> 14) return (int)(((1 << 15) - 1) / 7 + (beg >> 14));
    if (beg >> 17 == end >> 17) return (int)(((1 << 12) - 1) / 7 + (beg >> 17));
    if (beg >> 20 == end >> 20) return (int)(((1 << 9) - 1) / 7 + (beg >> 20));
    if (beg >> 23 == end >> 23) return (int)(((1 << 6) - 1) / 7 + (beg >> 23));
    if (beg >> 26 == end >> 26) return (int)(((1 << 3) - 1) / 7 + (beg >> 26));
    return 0;
}

struct RecIndex { int32_t pos, end; uint32_t block, within, bytes; };   // where a record starts: block of the piece, offset in it
struct Piece {
    std::vector<uint8_t> comp;            // BGZF blocks
    std::vector<uint32_t> block_size;     // compressed size of each
    std::vector<RecIndex> recs;
    int64_t n_bases = 0;
};

int g_level = 1, g_tags = 0, g_quals = 0;

struct Deflater {
#ifdef PA_HAVE_LIBDEFLATE
    libdeflate_compressor* c = libdeflate_alloc_compressor(1);
    ~Deflater() { libdeflate_free_compressor(c); }
#endif
    // one BGZF block out of `n` <= 0xff00 bytes
    void block(const uint8_t* data, size_t n, Piece& out) {
        const size_t at = out.comp.size();
        out.comp.resize(at + 18 + 0x10400 + 8);
        uint8_t* p = out.comp.data() + at;
        static const uint8_t head[16] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0};
        std::memcpy(p, head, 16);
        size_t clen = 0;
        bool done = false;
#ifdef PA_HAVE_LIBDEFLATE
        if (g_level == 1) {
            clen = libdeflate_deflate_compress(c, data, n, p + 18, 0x10400);
            done = true;
        }
#endif
        if (!done) {                                  // zlib's own deflate at the level asked for (6: what htslib writes)
            z_stream zs{};
            deflateInit2(&zs, g_level, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY);
            zs.next_in = const_cast<uint8_t*>(data);
            zs.avail_in = (uInt)n;
            zs.next_out = p + 18;
            zs.avail_out = 0x10400;
            deflate(&zs, Z_FINISH);
            clen = zs.total_out;
            deflateEnd(&zs);
        }
        const uint32_t bsize = (uint32_t)(clen + 18 + 8 - 1);
        p[16] = bsize & 0xff;
        p[17] = bsize >> 8;
        const uint32_t crc = (uint32_t)crc32(crc32(0L, Z_NULL, 0), data, (uInt)n), isize = (uint32_t)n;
        std::memcpy(p + 18 + clen, &crc, 4);
        std::memcpy(p + 18 + clen + 4, &isize, 4);
        out.comp.resize(at + bsize + 1);
        out.block_size.push_back(bsize + 1);
    }
};

const char kLetters[4] = {'A', 'C', 'G', 'T'};
const uint8_t kCode[4] = {1, 2, 4, 8};   // 4-bit codes of A C G T

struct Genome {
    std::vector<std::vector<uint8_t>> contig;     // 0..3 per base
};

// the records that start in [lo, hi) of contig `tid`
void make_piece(const Genome& g, int tid, int64_t lo, int64_t hi, double coverage, uint64_t seed, Piece& out) {
    const std::vector<uint8_t>& ref = g.contig[(size_t)tid];
    const int64_t L = (int64_t)ref.size();
    Rng rng(seed ^ site_hash((uint64_t)tid, (uint64_t)lo, 77));
    const double mean_len = 8000.0;
    const int64_t n_reads = (int64_t)((hi - lo) * coverage / mean_len + rng.unit());
    std::vector<int64_t> starts((size_t)n_reads);
    for (auto& s : starts) s = lo + (int64_t)(rng.unit() * (hi - lo));
    std::sort(starts.begin(), starts.end());
    Deflater defl;
    std::vector<uint8_t> raw;                      // uncompressed bytes waiting for a block
    raw.reserve(0x20000);
    uint32_t n_blocks = 0;
    std::vector<uint32_t> cigar;
    std::vector<uint8_t> bases, quals, rec, md;
    auto flush = [&](bool all) {
        size_t at = 0;
        while (raw.size() - at >= 0xff00 || (all && at < raw.size())) {
            const size_t n = std::min<size_t>(0xff00, raw.size() - at);
            defl.block(raw.data() + at, n, out);
            at += n;
            ++n_blocks;
        }
        raw.erase(raw.begin(), raw.begin() + (long)at);
    };
    char name[32];
    for (int64_t k = 0; k < n_reads; ++k) {
        const int64_t pos = starts[(size_t)k];
        int64_t want = 4000 + (int64_t)rng.below(8001);
        if (pos + want > L) want = L - pos;
        if (want < 50) continue;
        const int hap = (int)(rng.next() & 1);
        cigar.clear();
        bases.clear();
        quals.clear();
        auto push = [&](uint32_t op, uint32_t n) {
            if (!cigar.empty() && (cigar.back() & 15u) == op) cigar.back() += n << 4;
            else cigar.push_back((n << 4) | op);
        };
        int64_t rp = pos;
        const int64_t stop = pos + want;
        md.clear();
        uint32_t md_run = 0, nm = 0;
        auto md_flush = [&]() {
            char num[16];
            const int len = snprintf(num, sizeof num, "%u", md_run);
            md.insert(md.end(), num, num + len);
            md_run = 0;
        };
        while (rp < stop) {
            uint8_t b = ref[(size_t)rp];
            const uint64_t h = site_hash((uint64_t)tid, (uint64_t)rp, seed);
            const bool snp_site = (h % 1000) == 0, indel_site = ((h >> 20) % 700) == 0;
            if (snp_site && hap == 1) b = (uint8_t)((b + 1 + ((h >> 40) % 3)) & 3);
            else if (rng.below(100) < 4) b = (uint8_t)((b + 1 + rng.below(3)) & 3);
            if (g_tags) {
                if (b == ref[(size_t)rp]) ++md_run;
                else { md_flush(); md.push_back((uint8_t)kLetters[ref[(size_t)rp]]); ++nm; }
            }
            bases.push_back(b);
            push(0, 1);
            ++rp;
            if (rp >= stop - 2) continue;           // end on aligned bases
            const uint32_t u = rng.below(1000);
            bool ins = false, del = false;
            uint32_t n = 1 + rng.below(5);
            if (indel_site && hap == 1 && rng.below(10) < 9) {
                ins = (h >> 50) & 1;
                del = !ins;
                n = 1 + (uint32_t)((h >> 52) % 5);
            } else if (u < 10) ins = true;
            else if (u < 20) del = true;
            if (ins) {
                for (uint32_t i = 0; i < n; ++i) bases.push_back((uint8_t)(indel_site && hap == 1 ? ((h >> (2 * i)) & 3) : rng.below(4)));
                push(1, n);
                nm += n;
            } else if (del && rp + n < stop - 2) {
                push(2, n);
                if (g_tags) {
                    md_flush();
                    md.push_back('^');
                    for (uint32_t i = 0; i < n; ++i) md.push_back((uint8_t)kLetters[ref[(size_t)(rp + i)]]);
                    nm += n;
                }
                rp += n;
            }
        }
        if (g_tags) md_flush();
        const uint32_t l_seq = (uint32_t)bases.size();
        quals.resize(l_seq);
        if (g_quals == 1) {
            // qualities with run-length structure, as a basecaller that bins them writes (plateaus of one value a dozen bases long,
            // the next plateau a step or two away, an occasional dip to a very low value inside a homopolymer-like stretch): the
            // histogram is ONT-like (Q7 .. Q30, most mass around Q12 - Q22); deflate finds the runs, so a member of such records
            // inflates to 3-4 x its size -- more output per consumed bit than the uniform qualities above give
            static const uint8_t kBins[8] = {7, 10, 13, 16, 19, 22, 26, 30};
            int bin = 2 + (int)rng.below(4);
            for (uint32_t i = 0; i < l_seq;) {
                uint64_t r = rng.next();
                uint32_t run = 1 + (uint32_t)(r & 7) + (uint32_t)((r >> 3) & 7) + (uint32_t)((r >> 6) & 7);       // 1 .. 22, mean 11.5
                const uint32_t kind = (uint32_t)((r >> 9) & 63);
                uint8_t q = kBins[bin];
                if (kind == 0) { q = (uint8_t)(2 + ((r >> 15) & 3)); run = 1 + (uint32_t)((r >> 17) & 3); }     // a dip
                for (uint32_t j = 0; j < run && i < l_seq; ++j, ++i) quals[i] = q;
                const int step = (int)((r >> 20) % 5) - 2;                                                       // -2 .. +2
                bin = std::min(7, std::max(0, bin + step));
            }
        } else
        for (uint32_t i = 0; i < l_seq; i += 8) {
            uint64_t r = rng.next();
            for (uint32_t j = i; j < std::min(l_seq, i + 8); ++j, r >>= 8) {
                const uint32_t v = (uint32_t)(r & 0xff);
                quals[j] = (uint8_t)(v < 8 ? 2 + (v & 3) : 6 + ((v * 34u) >> 8));       // mostly 6..39, a tail of very low ones
            }
        }
        const int64_t end = rp;
        uint32_t flag = (rng.next() & 1) ? 16u : 0u;
        const uint32_t f = rng.below(1000);
        if (f < 10) flag |= 0x800;
        else if (f < 15) flag |= 0x100;
        else if (f < 20) flag |= 0x400;
        const uint32_t mapq = rng.below(100) < 2 ? 0 : 60;
        const int l_name = snprintf(name, sizeof name, "r%d_%lld_%lld", tid, (long long)lo, (long long)k) + 1;
        const uint32_t n_cig = (uint32_t)cigar.size();
        // aux data as aligners write it: NM:i (uint16), MD:Z, RG:Z
        const uint32_t aux_bytes = g_tags ? (3 + 2) + (3 + (uint32_t)md.size() + 1) + (3 + 4) : 0;
        const uint32_t body = 32 + (uint32_t)l_name + 4 * std::min<uint32_t>(n_cig, 65535u) + (l_seq + 1) / 2 + l_seq + aux_bytes;
        if (n_cig > 65535) continue;                 // (never with these lengths)
        rec.resize(4 + body);
        uint8_t* p = rec.data();
        auto w32 = [&](uint32_t v) { std::memcpy(p, &v, 4); p += 4; };
        w32(body);
        w32((uint32_t)tid);
        w32((uint32_t)pos);
        w32((uint32_t)l_name | (mapq << 8) | ((uint32_t)reg2bin(pos, end) << 16));
        w32(n_cig | (flag << 16));
        w32(l_seq);
        w32(0xffffffffu);
        w32(0xffffffffu);
        w32(0);
        std::memcpy(p, name, (size_t)l_name);
        p += l_name;
        std::memcpy(p, cigar.data(), 4ull * n_cig);
        p += 4ull * n_cig;
        for (uint32_t i = 0; i + 1 < l_seq; i += 2) *p++ = (uint8_t)((kCode[bases[i]] << 4) | kCode[bases[i + 1]]);
        if (l_seq & 1) *p++ = (uint8_t)(kCode[bases[l_seq - 1]] << 4);
        std::memcpy(p, quals.data(), l_seq);
        p += l_seq;
        if (g_tags) {
            const uint16_t nm16 = (uint16_t)std::min<uint32_t>(nm, 65535u);
            std::memcpy(p, "NMS", 3); p += 3;
            std::memcpy(p, &nm16, 2); p += 2;
            std::memcpy(p, "MDZ", 3); p += 3;
            std::memcpy(p, md.data(), md.size()); p += md.size();
            *p++ = 0;
            std::memcpy(p, "RGZ", 3); p += 3;
            std::memcpy(p, "rg1", 4); p += 4;
        }
        // the record starts in block n_blocks + (bytes waiting) / 0xff00 of this piece
        const uint64_t waiting = raw.size();
        out.recs.push_back(RecIndex{(int32_t)pos, (int32_t)end, n_blocks + (uint32_t)(waiting / 0xff00), (uint32_t)(waiting % 0xff00), 4 + body});
        raw.insert(raw.end(), rec.begin(), rec.end());
        out.n_bases += l_seq;
        flush(false);
    }
    flush(true);
}

}  // namespace

int main(int argc, char** argv) {
    if (argc < 3) {
        fprintf(stderr, "usage: synth_bam <out_dir> <genome_bases> [coverage=60] [seed=2027] [threads=0] [contigs=1]\n");
        return 2;
    }
    const std::string dir = argv[1];
    const int64_t total = atoll(argv[2]);
    const double coverage = argc > 3 ? atof(argv[3]) : 60.0;
    const uint64_t seed = argc > 4 ? (uint64_t)atoll(argv[4]) : 2027;
    int threads = argc > 5 ? atoi(argv[5]) : 0;
    const int n_contigs = std::max(1, argc > 6 ? atoi(argv[6]) : 1);
    g_level = argc > 7 ? std::max(1, std::min(9, atoi(argv[7]))) : 1;
    g_tags = argc > 8 ? atoi(argv[8]) != 0 : 0;
    g_quals = argc > 9 ? atoi(argv[9]) : 0;
    if (threads <= 0) threads = std::max(1u, std::thread::hardware_concurrency());
    if (FILE* fh = fopen("/sys/fs/cgroup/cpu.max", "r")) {       // a cgroup quota below the hardware's thread count
        char quota[32] = {0};
        long long period = 0;
        if (fscanf(fh, "%31s %lld", quota, &period) == 2 && strcmp(quota, "max") != 0 && period > 0 && argc <= 5)
            threads = std::min(threads, std::max(1, (int)(atoll(quota) / period)));
        fclose(fh);
    }
    Genome g;
    g.contig.resize((size_t)n_contigs);
    std::vector<std::string> names;
    for (int c = 0; c < n_contigs; ++c) {
        names.push_back("ctg" + std::to_string(c + 1));
        g.contig[(size_t)c].resize((size_t)(total / n_contigs));
    }
    {   // the genome, in parallel slabs
        std::vector<std::thread> pool;
        std::atomic<int64_t> next{0};
        const int64_t slab = 1 << 20;
        for (int t = 0; t < threads; ++t)
            pool.emplace_back([&] {
                for (;;) {
                    const int64_t s = next.fetch_add(1);
                    const int64_t per = (int64_t)g.contig[0].size(), slabs_per = (per + slab - 1) / slab;
                    if (s >= slabs_per * n_contigs) return;
                    const int c = (int)(s / slabs_per);
                    const int64_t lo = (s % slabs_per) * slab, hi = std::min(per, lo + slab);
                    Rng rng(seed * 31 + (uint64_t)s);
                    for (int64_t i = lo; i < hi; i += 32) {
                        uint64_t r = rng.next();
                        for (int64_t j = i; j < std::min(hi, i + 32); ++j, r >>= 2) g.contig[(size_t)c][(size_t)j] = (uint8_t)(r & 3);
                    }
                }
            });
        for (auto& t : pool) t.join();
    }
    // FASTA + .fai
    {
        FILE* fa = fopen((dir + "/draft.fa").c_str(), "wb");
        FILE* fai = fopen((dir + "/draft.fa.fai").c_str(), "w");
        if (!fa || !fai) { perror("fasta"); return 1; }
        int64_t off = 0;
        std::vector<char> line;
        for (int c = 0; c < n_contigs; ++c) {
            const auto& s = g.contig[(size_t)c];
            off += fprintf(fa, ">%s\n", names[(size_t)c].c_str());
            fprintf(fai, "%s\t%zu\t%lld\t60\t61\n", names[(size_t)c].c_str(), s.size(), (long long)off);
            line.resize(s.size() + s.size() / 60 + 2);
            size_t w = 0;
            for (size_t i = 0; i < s.size(); ++i) {
                line[w++] = kLetters[s[i]];
                if (i % 60 == 59 || i + 1 == s.size()) line[w++] = '\n';
            }
            fwrite(line.data(), 1, w, fa);
            off += (int64_t)w;
        }
        fclose(fa);
        fclose(fai);
    }
    // pieces
    struct Job { int tid; int64_t lo, hi; };
    std::vector<Job> jobs;
    const int64_t piece = 1 << 20;
    for (int c = 0; c < n_contigs; ++c)
        for (int64_t lo = 0; lo < (int64_t)g.contig[(size_t)c].size(); lo += piece)
            jobs.push_back(Job{c, lo, std::min<int64_t>((int64_t)g.contig[(size_t)c].size(), lo + piece)});
    std::vector<Piece> pieces(jobs.size());
    {
        std::vector<std::thread> pool;
        std::atomic<size_t> next{0};
        for (int t = 0; t < threads; ++t)
            pool.emplace_back([&] {
                for (;;) {
                    const size_t j = next.fetch_add(1);
                    if (j >= jobs.size()) return;
                    make_piece(g, jobs[j].tid, jobs[j].lo, jobs[j].hi, coverage, seed, pieces[j]);
                }
            });
        for (auto& t : pool) t.join();
    }
    // header block, pieces in order, EOF marker; the index from the block sizes
    FILE* bam = fopen((dir + "/reads.bam").c_str(), "wb");
    if (!bam) { perror("bam"); return 1; }
    int64_t coff = 0;
    {
        std::string text = "@HD\tVN:1.6\tSO:coordinate\n";
        for (int c = 0; c < n_contigs; ++c) text += "@SQ\tSN:" + names[(size_t)c] + "\tLN:" + std::to_string(g.contig[(size_t)c].size()) + "\n";
        text += "@RG\tID:syn\tSM:SYN\n";
        std::vector<uint8_t> h;
        auto a32 = [&](uint32_t v) { h.insert(h.end(), (uint8_t*)&v, (uint8_t*)&v + 4); };
        h.insert(h.end(), {'B', 'A', 'M', 1});
        a32((uint32_t)text.size());
        h.insert(h.end(), text.begin(), text.end());
        a32((uint32_t)n_contigs);
        for (int c = 0; c < n_contigs; ++c) {
            a32((uint32_t)names[(size_t)c].size() + 1);
            h.insert(h.end(), names[(size_t)c].begin(), names[(size_t)c].end());
            h.push_back(0);
            a32((uint32_t)g.contig[(size_t)c].size());
        }
        Piece hp;
        Deflater d;
        d.block(h.data(), h.size(), hp);
        fwrite(hp.comp.data(), 1, hp.comp.size(), bam);
        coff += (int64_t)hp.comp.size();
    }
    struct BinChunks { std::vector<std::pair<uint64_t, uint64_t>> chunks; };
    std::vector<std::map<uint32_t, BinChunks>> bins((size_t)n_contigs);
    std::vector<std::vector<uint64_t>> lin((size_t)n_contigs);
    int64_t n_records = 0, n_bases = 0;
    for (size_t j = 0; j < jobs.size(); ++j) {
        const Piece& p = pieces[j];
        std::vector<int64_t> block_off(p.block_size.size() + 1, coff);
        for (size_t b = 0; b < p.block_size.size(); ++b) block_off[b + 1] = block_off[b] + p.block_size[b];
        for (const RecIndex& r : p.recs) {
            const uint64_t vbeg = ((uint64_t)block_off[r.block] << 16) | r.within;
            // the record ends `bytes` further in the uncompressed stream of the piece (blocks of 0xff00 bytes but the last)
            const uint64_t stream_end = (uint64_t)r.block * 0xff00 + r.within + r.bytes;
            uint64_t eb = stream_end / 0xff00, ew = stream_end % 0xff00;
            if (eb >= p.block_size.size()) { eb = p.block_size.size() - 1; ew = stream_end - eb * 0xff00; }
            const uint64_t vend = ((uint64_t)block_off[eb] << 16) | ew;
            auto& chunks = bins[(size_t)jobs[j].tid][(uint32_t)reg2bin(r.pos, r.end)].chunks;
            if (!chunks.empty() && chunks.back().second == vbeg) chunks.back().second = vend;
            else chunks.emplace_back(vbeg, vend);
            auto& l = lin[(size_t)jobs[j].tid];
            const size_t w1 = (size_t)((r.end - 1) >> 14);
            if (l.size() <= w1) l.resize(w1 + 1, 0);
            for (size_t w = (size_t)(r.pos >> 14); w <= w1; ++w)
                if (l[w] == 0) l[w] = vbeg;
        }
        fwrite(p.comp.data(), 1, p.comp.size(), bam);
        coff = block_off.back();
        n_records += (int64_t)p.recs.size();
        n_bases += p.n_bases;
    }
    static const uint8_t eof_block[28] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0, 27, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    fwrite(eof_block, 1, 28, bam);
    fclose(bam);
    FILE* bai = fopen((dir + "/reads.bam.bai").c_str(), "wb");
    if (!bai) { perror("bai"); return 1; }
    auto o32 = [&](uint32_t v) { fwrite(&v, 4, 1, bai); };
    auto o64 = [&](uint64_t v) { fwrite(&v, 8, 1, bai); };
    fwrite("BAI\1", 1, 4, bai);
    o32((uint32_t)n_contigs);
    for (int c = 0; c < n_contigs; ++c) {
        o32((uint32_t)bins[(size_t)c].size());
        for (const auto& kv : bins[(size_t)c]) {
            o32(kv.first);
            o32((uint32_t)kv.second.chunks.size());
            for (const auto& ch : kv.second.chunks) { o64(ch.first); o64(ch.second); }
        }
        o32((uint32_t)lin[(size_t)c].size());
        for (uint64_t v : lin[(size_t)c]) o64(v);
    }
    fclose(bai);
    printf("{\"records\": %lld, \"read_bases\": %lld, \"genome_bases\": %lld, \"coverage\": %.1f, \"bam_bytes\": %lld, \"threads\": %d, "
           "\"deflate\": \"%s\", \"aux_tags\": %s, \"quals\": \"%s\"}\n",
           (long long)n_records, (long long)n_bases, (long long)total, (double)n_bases / (double)total, (long long)coff + 28, threads,
           g_level == 1 ? "libdeflate level 1 (zlib level 1 where libdeflate is absent)" : (std::string("zlib level ") + std::to_string(g_level)).c_str(),
           g_tags ? "true" : "false", g_quals == 1 ? "run-length (binned plateaus, Q7-Q30)" : "~5 bits of entropy per base");
    return 0;
}
