// freddie_host.cpp -- native host side of the segmentation stage (include/freddie_host.h).
//
// Multi-threaded replacement of the reference's per-partition Python I/O: the split / reads TSV parser with
// the read_reps grouping (py/freddie_segment.py:121-185), the per-read soft-clip / poly-A / unaligned-gap
// annotation (:289-472) and the segment TSV writer (:703-732).  What each function must produce is defined by
// those reference lines; the implementation is this project's own and is built around not allocating per read:
//   * the two TSVs of a partition are mapped, not copied; read names, contigs and sequences are views into the
//     mappings for as long as the batch lives (a partition loaded from a side-car: views into the mapped side-car, sequences packed);
//   * one pass over the bytes of a line, integers parsed in place; the rep grouping is an open-addressing table over
//     the exon tuples; sequences are matched to reads by position when the two files list the reads in the same
//     order (the split stage writes them that way), by a hash of the read id otherwise;
//   * the writer assembles a partition's TSV in one reused buffer with a streaming poly-A scan (no score array) and
//     fixed-size gap tokens, and hands it to the kernel in one write().
#include "freddie_host.h"

#include <algorithm>
#include <atomic>
#include <cerrno>
#include <cstdio>
#include <cstring>
#include <exception>
#include <fcntl.h>
#include <mutex>
#include <string>
#include <sys/mman.h>
#include <sys/stat.h>
#include <thread>
#include <unistd.h>
#include <unordered_map>
#include <vector>
#include <dirent.h>

namespace {

typedef long long i64;

struct Str {                   // a view into a mapped file (a TSV, or the partition's side-car)
    const char *p = nullptr;
    uint32_t n = 0;
};

struct Read {
    i64 id = 0, tint = 0;
    Str name, chr, seq;          // seq.p == nullptr: the sequence lives two bits a base in the partition's side-car (seq_g0 = its first base there)
    uint64_t seq_g0 = 0;
    int ex0 = 0, ex1 = 0;     // exon range of this read inside the partition's exon arrays
    int rep = 0;
    char strand = '+';
};

struct Mapping {              // a whole file, read-only; empty files map to (nullptr, 0)
    const char *p = nullptr;
    size_t n = 0;
    bool open(const char *path, uint64_t *size_out, int64_t *mtime_ns_out) {
        close();
        int fd = ::open(path, O_RDONLY | O_CLOEXEC);
        if (fd < 0) return false;
        struct stat st;
        if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode)) { ::close(fd); return false; }
        if (size_out) *size_out = (uint64_t)st.st_size;
        if (mtime_ns_out) *mtime_ns_out = (int64_t)st.st_mtim.tv_sec * 1000000000ll + (int64_t)st.st_mtim.tv_nsec;
        n = (size_t)st.st_size;
        if (n) {
            // no MAP_POPULATE: pages are faulted in by the threads that parse them (populating under the address-space lock
            // serialised the loader threads: 0.12 s vs 0.02 s per 100 k reads at 8 threads), and the bulk of the sequences --
            // everything but the soft-clipped ends the poly-A search looks at -- is never touched at all
            void *m = mmap(nullptr, n, PROT_READ, MAP_PRIVATE, fd, 0);
            if (m == MAP_FAILED) { ::close(fd); n = 0; return false; }
            p = static_cast<const char *>(m);
        }
        ::close(fd);
        return true;
    }
    void close() {
        if (p) munmap(const_cast<char *>(p), n);
        p = nullptr; n = 0;
    }
};

struct Partition {
    std::string chr;
    i64 id = 0, read_count = 0;
    std::vector<int> iv_s, iv_e;
    std::vector<Read> reads;
    std::vector<int> ts, te, qs, qe;        // per exon of every read
    std::vector<int> cig_off, cig_len;      // CIGAR ops per exon (CSR)
    std::vector<char> cig_op;
    // read reps in first-occurrence order (:165-170)
    std::vector<int> rep_first_read, rep_weight;
    Mapping map_split, map_reads;           // the TSVs stay mapped while the partition lives
    // side-car path: the .fsc stays mapped instead; names and contigs are views into it and the sequences stay PACKED (two bits a
    // base + an exception list) -- the writer unpacks the soft-clipped ends it looks at (seq_window), a few dozen bases of a read's
    // ~1 500.  (Until round 6 every sequence was unpacked into an arena when the side-car was loaded: 1.5 KB of stores and page
    // faults per read, which made a side-car load SLOWER than parsing the TSVs: 0.15 against 0.08 s per 250 k reads.)
    Mapping map_fsc;
    const uint8_t *packed = nullptr;
    const uint64_t *exc_pos = nullptr;
    const uint8_t *exc_ch = nullptr;
    size_t n_exc = 0;
    bool from_sidecar = false;
    // the TSVs as they were when this partition was read (what a side-car written from it is bound to)
    uint64_t split_size = 0, reads_size = 0;
    int64_t split_mtime_ns = 0, reads_mtime_ns = 0;
    bool have_stat = false;
    std::string err;

    Partition() = default;
    Partition(const Partition &) = delete;
    Partition &operator=(const Partition &) = delete;
    ~Partition() { map_split.close(); map_reads.close(); map_fsc.close(); }
    void reset() {
        chr.clear(); id = read_count = 0;
        iv_s.clear(); iv_e.clear(); reads.clear(); ts.clear(); te.clear(); qs.clear(); qe.clear();
        cig_off.clear(); cig_len.clear(); cig_op.clear(); rep_first_read.clear(); rep_weight.clear();
        map_split.close(); map_reads.close(); map_fsc.close(); packed = nullptr; exc_pos = nullptr; exc_ch = nullptr; n_exc = 0; from_sidecar = false;
        have_stat = false; err.clear();
    }
};

// strict unsigned decimal ([0-9]+), as the reference's regexes require
inline bool parse_uint(const char *b, const char *e, i64 &v) {
    if (b >= e || e - b > 18) return false;
    i64 x = 0;
    for (const char *p = b; p < e; ++p) {
        const unsigned d = (unsigned)(*p - '0');
        if (d > 9) return false;
        x = x * 10 + d;
    }
    v = x;
    return true;
}
// digits from p up to (not including) the first non-digit before e; false if there is none (or too many)
inline bool scan_uint(const char *&p, const char *e, i64 &v) {
    const char *b = p;
    i64 x = 0;
    while (p < e) {
        const unsigned d = (unsigned)(*p - '0');
        if (d > 9) break;
        x = x * 10 + d;
        ++p;
    }
    if (p == b || p - b > 18) return false;
    v = x;
    return true;
}
bool chr_ok(const char *b, const char *e) {      // chr_re of the reference (:23)
    static const char first[] = "!#$%&+./:;?@^_|~-", rest[] = "!#$%&*+./:;=?@^_|~-";
    if (b >= e) return false;
    for (const char *p = b; p < e; ++p) {
        unsigned char c = (unsigned char)*p;
        bool alnum = (c >= '0' && c <= '9') || (c >= 'A' && c <= 'Z') || (c >= 'a' && c <= 'z');
        if (alnum) continue;
        if (!strchr(p == b ? first : rest, c) || c == 0) return false;
    }
    return true;
}
inline bool name_ok(const char *b, const char *e) {     // [!-?A-~]{1,254} (:30)
    if (e - b < 1 || e - b > 254) return false;
    for (const char *p = b; p < e; ++p) {
        unsigned char c = (unsigned char)*p;
        if (!((c >= '!' && c <= '?') || (c >= 'A' && c <= '~'))) return false;
    }
    return true;
}
inline const char *find_tab(const char *p, const char *e) {
    const char *t = (const char *)memchr(p, '\t', (size_t)(e - p));
    return t ? t : e;
}
inline bool is_cigar_op(char c) {
    switch (c) { case 'M': case 'I': case 'D': case 'N': case 'S': case 'H': case 'P': case 'X': case '=': return true; default: return false; }
}

// read rep = reads with the same tuple of target intervals, first-occurrence order (:165-170): open addressing over a
// hash of the tuple, equality checked against the exons of the rep's first read
struct RepTable {
    std::vector<int> slot;       // rep + 1, 0 = empty
    std::vector<uint64_t> hash;  // per rep
    size_t mask = 0;
    void init(size_t expect) {
        size_t n = 1024;
        while (n < expect * 2 + 16) n <<= 1;
        slot.assign(n, 0); mask = n - 1; hash.clear();
    }
    static uint64_t mix(uint64_t h, uint64_t v) { h ^= v; h *= 0x9E3779B97F4A7C15ull; h ^= h >> 29; return h; }
    void grow() {
        std::vector<int> ns((mask + 1) * 2, 0);
        const size_t nm = ns.size() - 1;
        for (size_t r = 0; r < hash.size(); ++r) { size_t i = hash[r] & nm; while (ns[i]) i = (i + 1) & nm; ns[i] = (int)r + 1; }
        slot.swap(ns); mask = nm;
    }
};

void parse_partition(const char *split_path, const char *reads_path, Partition &P) {
    if (!P.map_split.open(split_path, &P.split_size, &P.split_mtime_ns)) { P.err = std::string("cannot read ") + split_path; return; }
    const char *p = P.map_split.p, *end = p + P.map_split.n;
    bool have_header = false;
    RepTable reps;
    {   // exons are ~30 bytes of text each, CIGAR ops ~4: reserving from the file size avoids regrowth
        const size_t guess = P.map_split.n / 24 + 16;
        P.ts.reserve(guess); P.te.reserve(guess); P.qs.reserve(guess); P.qe.reserve(guess); P.cig_off.reserve(guess + 1);
        P.cig_len.reserve(guess * 2); P.cig_op.reserve(guess * 2);
    }
    const std::string sp(split_path);
    while (p < end) {
        const char *nl = (const char *)memchr(p, '\n', (size_t)(end - p));
        if (!nl) { P.err = sp + ": line without newline"; return; }
        if (*p == '#') {
            if (have_header) { P.err = sp + ": more than one tint in the file (py/freddie_segment.py:699)"; return; }
            // #chr \t id \t intervals \t read_count
            const char *c0e = find_tab(p, nl), *c1 = c0e + 1;
            const char *c1e = c0e < nl ? find_tab(c1, nl) : nl, *c2 = c1e + 1;
            const char *c2e = c1e < nl ? find_tab(c2, nl) : nl, *c3 = c2e + 1;
            const char *c3e = c2e < nl ? find_tab(c3, nl) : nl;
            if (c2e >= nl || c3e != nl || !chr_ok(p + 1, c0e) || !parse_uint(c1, c1e, P.id) || !parse_uint(c3, c3e, P.read_count)) {
                P.err = sp + ": malformed header line"; return;
            }
            P.chr.assign(p + 1, c0e);
            const char *q = c2;
            while (q <= c2e) {
                const char *c = (const char *)memchr(q, ',', (size_t)(c2e - q));
                if (!c) c = c2e;
                i64 s, e;
                const char *t = q;
                if (!scan_uint(t, c, s) || t >= c || *t != '-' || !parse_uint(t + 1, c, e)) { P.err = sp + ": malformed tint intervals"; return; }
                P.iv_s.push_back((int)s); P.iv_e.push_back((int)e);
                q = c + 1;
            }
            for (size_t k = 0; k < P.iv_s.size(); ++k) {
                if (!(P.iv_s[k] < P.iv_e[k])) { P.err = sp + ": interval with start >= end (py/freddie_segment.py:140)"; return; }
                if (k && !(P.iv_e[k - 1] < P.iv_s[k])) { P.err = sp + ": intervals overlap or are unordered (py/freddie_segment.py:138)"; return; }
            }
            have_header = true;
            const size_t expect = (size_t)std::min<i64>(P.read_count, (i64)(P.map_split.n / 16) + 1);
            P.reads.reserve(expect);
            reps.init(expect);
        } else {
            if (!have_header) { P.err = sp + ": read line before the tint header"; return; }
            // rid \t name \t chr \t strand \t tint \t interval(\t interval)*
            Read r;
            const char *c0e = find_tab(p, nl);
            const char *c1 = c0e + 1, *c1e = c0e < nl ? find_tab(c1, nl) : nl;
            const char *c2 = c1e + 1, *c2e = c1e < nl ? find_tab(c2, nl) : nl;
            const char *c3 = c2e + 1, *c3e = c2e < nl ? find_tab(c3, nl) : nl;
            const char *c4 = c3e + 1, *c4e = c3e < nl ? find_tab(c4, nl) : nl;
            if (c4e >= nl || !parse_uint(p, c0e, r.id) || !name_ok(c1, c1e) || !chr_ok(c2, c2e) || c3e - c3 != 1 ||
                (*c3 != '+' && *c3 != '-') || !parse_uint(c4, c4e, r.tint)) {
                P.err = sp + ": malformed read line"; return;
            }
            if (r.tint != P.id) { P.err = sp + ": read refers to another tint"; return; }
            r.name.p = c1; r.name.n = (uint32_t)(c1e - c1);
            r.chr.p = c2; r.chr.n = (uint32_t)(c2e - c2);
            r.strand = *c3;
            r.ex0 = (int)P.ts.size();
            uint64_t h = 0x243F6A8885A308D3ull;
            const char *q = c4e + 1;                   // first interval field (c4e < nl)
            for (;;) {
                // ts-te:qs-qe:CIGAR up to the next tab / end of line
                i64 ts, te, qs, qe;
                if (!scan_uint(q, nl, ts) || q >= nl || *q != '-' || (++q, !scan_uint(q, nl, te)) || q >= nl || *q != ':' ||
                    (++q, !scan_uint(q, nl, qs)) || q >= nl || *q != '-' || (++q, !scan_uint(q, nl, qe)) || q >= nl || *q != ':') {
                    P.err = sp + ": malformed read interval"; return;
                }
                ++q;
                if (q >= nl || *q == '\t') { P.err = sp + ": malformed read interval"; return; }
                P.cig_off.push_back((int)P.cig_len.size());
                while (q < nl && *q != '\t') {
                    i64 len;
                    if (!scan_uint(q, nl, len) || q >= nl || !is_cigar_op(*q)) { P.err = sp + ": malformed CIGAR"; return; }
                    P.cig_len.push_back((int)len); P.cig_op.push_back(*q);
                    ++q;
                }
                P.ts.push_back((int)ts); P.te.push_back((int)te); P.qs.push_back((int)qs); P.qe.push_back((int)qe);
                h = RepTable::mix(h, ((uint64_t)(uint32_t)(int)ts << 32) | (uint32_t)(int)te);
                if (q >= nl) break;
                ++q;                                   // the tab: another interval must follow
                if (q >= nl) { P.err = sp + ": malformed read interval"; return; }
            }
            r.ex1 = (int)P.ts.size();
            for (int x = r.ex0; x < r.ex1; ++x) {       // :158-161
                if (!(P.ts[x] < P.te[x] && P.qs[x] < P.qe[x])) { P.err = sp + ": exon with start >= end (py/freddie_segment.py:160)"; return; }
                if (x > r.ex0 && !(P.te[x - 1] <= P.ts[x] && P.qe[x - 1] <= P.qs[x])) { P.err = sp + ": exons out of order (py/freddie_segment.py:158)"; return; }
            }
            {
                const int m = r.ex1 - r.ex0;
                size_t i = h & reps.mask;
                int found = -1;
                while (reps.slot[i]) {
                    const int cand = reps.slot[i] - 1;
                    if (reps.hash[(size_t)cand] == h) {
                        const Read &f = P.reads[(size_t)P.rep_first_read[(size_t)cand]];
                        if (f.ex1 - f.ex0 == m && memcmp(&P.ts[(size_t)f.ex0], &P.ts[(size_t)r.ex0], (size_t)m * 4) == 0 &&
                            memcmp(&P.te[(size_t)f.ex0], &P.te[(size_t)r.ex0], (size_t)m * 4) == 0) { found = cand; break; }
                    }
                    i = (i + 1) & reps.mask;
                }
                if (found < 0) {
                    r.rep = (int)P.rep_first_read.size();
                    reps.slot[i] = r.rep + 1;
                    reps.hash.push_back(h);
                    P.rep_first_read.push_back((int)P.reads.size());
                    P.rep_weight.push_back(1);
                    if (reps.hash.size() * 2 > reps.mask) reps.grow();
                } else {
                    r.rep = found;
                    P.rep_weight[(size_t)found] += 1;
                }
            }
            P.reads.push_back(r);
        }
        p = nl + 1;
    }
    P.cig_off.push_back((int)P.cig_len.size());
    if (!have_header) { P.err = sp + ": no tint header"; return; }
    if ((i64)P.reads.size() != P.read_count) { P.err = sp + ": read_count does not match the number of read lines (py/freddie_segment.py:164)"; return; }
    // sequences (:174-185): rid \t contig \t tint \t seq ; a dict keyed by rid in the reference (a later line replaces an
    // earlier one with the same rid), whose size must equal the number of reads
    if (!P.map_reads.open(reads_path, &P.reads_size, &P.reads_mtime_ns)) { P.err = std::string("cannot read ") + reads_path; return; }
    P.have_stat = true;
    const std::string rpath(reads_path);
    p = P.map_reads.p; end = p + P.map_reads.n;
    const size_t n_reads = P.reads.size();
    // fast path: line i carries the sequence of read i and the rids are strictly increasing (so the dict has one entry
    // per line); anything else goes through the dict
    bool positional = true;
    size_t line = 0;
    i64 prev_rid = -1;
    std::vector<std::pair<i64, Str>> later;      // lines seen after the order broke
    while (p < end) {
        const char *nl = (const char *)memchr(p, '\n', (size_t)(end - p));
        const char *le = nl ? nl : end;
        const char *re = le;
        while (re > p && (re[-1] == '\r' || re[-1] == ' ' || re[-1] == '\t' || re[-1] == '\n' || re[-1] == '\v' || re[-1] == '\f')) --re;   // line.rstrip()
        const char *c0e = find_tab(p, re);
        const char *c1e = c0e < re ? find_tab(c0e + 1, re) : re;
        const char *c2e = c1e < re ? find_tab(c1e + 1, re) : re;
        i64 rid;
        if (c2e >= re || !parse_uint(p, c0e, rid)) { P.err = rpath + ": malformed line"; return; }
        const char *c3 = c2e + 1, *c3e = find_tab(c3, re);
        Str seq; seq.p = c3; seq.n = (uint32_t)(c3e - c3);
        if (positional && line < n_reads && rid == P.reads[line].id && rid > prev_rid) P.reads[line].seq = seq;
        else { positional = false; later.emplace_back(rid, seq); }
        prev_rid = rid;
        ++line;
        p = nl ? nl + 1 : end;
    }
    if (positional) {
        if (line != n_reads) { P.err = rpath + ": number of sequences differs from the number of reads (py/freddie_segment.py:181)"; return; }
        return;
    }
    std::unordered_map<i64, Str> seq_of;
    seq_of.reserve(line * 2);
    const size_t n_pos = line - later.size();          // the lines that matched positionally before the order broke
    for (size_t i = 0; i < n_pos; ++i) seq_of[P.reads[i].id] = P.reads[i].seq;
    for (const auto &kv : later) seq_of[kv.first] = kv.second;
    if (seq_of.size() != n_reads) { P.err = rpath + ": number of sequences differs from the number of reads (py/freddie_segment.py:181)"; return; }
    for (Read &r : P.reads) {
        auto it = seq_of.find(r.id);
        if (it == seq_of.end()) { P.err = rpath + ": a read has no sequence"; return; }
        r.seq = it->second;
    }
}

// ---- per-read annotation (get_unaligned_gaps_and_polyA :370-472 and helpers) ------------------------------
struct Fail { const char *what; };

i64 thread_cigar(const Partition &P, int ex, i64 t_goal, i64 t_pos, i64 q_pos) {     // forward_thread_cigar :289-304
    if (t_pos > t_goal) throw Fail{"forward_thread_cigar: t_pos > t_goal (:290)"};
    int idx = P.cig_off[(size_t)ex], end = P.cig_off[(size_t)ex + 1];
    while (t_pos < t_goal) {
        if (idx >= end) throw Fail{"forward_thread_cigar: CIGAR exhausted before the goal (:293)"};
        i64 step = std::min<i64>(P.cig_len[(size_t)idx], t_goal - t_pos);       // every op is clipped, insertions too
        char op = P.cig_op[(size_t)idx];
        if (op == 'M' || op == 'X' || op == '=') { t_pos += step; q_pos += step; }
        else if (op == 'D') t_pos += step;
        else if (op == 'I') q_pos += step;
        ++idx;
    }
    return q_pos;
}
// (`from`: where the scan starts -- the queries of one read come with ascending positions, so each starts where the last one ended
// instead of at the read's first / last exon; the exon found is the same)
void query_at_or_after(const Partition &P, const Read &r, i64 start, i64 &q, i64 &slack, int *from = nullptr) {    // get_interval_start :307-326
    for (int x = from ? *from : r.ex0; x < r.ex1; ++x) {
        if (P.te[x] < start) continue;
        if (from) *from = x;
        if (start < P.ts[x]) { q = P.qs[x]; slack = start - P.ts[x]; }
        else { q = thread_cigar(P, x, start, P.ts[x], P.qs[x]); slack = 0; }
        if (!(slack <= 0 && P.qs[x] <= q && q <= P.qe[x])) throw Fail{"get_interval_start: slack / query position out of range (:323-324)"};
        return;
    }
    throw Fail{"get_interval_start: no exon at or after the position (:326)"};
}
void query_at_or_before(const Partition &P, const Read &r, i64 end, i64 &q, i64 &slack, int *from = nullptr) {     // get_interval_end :329-349
    // the LAST exon with ts <= end: scanned backwards from the read's end (the reference's order), or forwards from `from` -- the
    // last exon known to start at or before an earlier, smaller position
    int x = r.ex1 - 1;
    if (from) { x = *from; while (x + 1 < r.ex1 && P.ts[x + 1] <= end) ++x; if (x < r.ex0 || P.ts[x] > end) x = r.ex0 - 1; else *from = x; }
    else while (x >= r.ex0 && P.ts[x] > end) --x;
    if (x < r.ex0) throw Fail{"get_interval_end: no exon at or before the position (:349)"};
    if (P.te[x] < end) { q = P.qe[x]; slack = P.te[x] - end; }
    else { q = thread_cigar(P, x, end, P.ts[x], P.qs[x]); slack = 0; }
    if (!(slack <= 0 && 0 <= q && q <= P.qe[x])) throw Fail{"get_interval_end: slack / query position out of range (:346-347)"};
}

struct PolyRun { i64 first, len; double purity; char ch; };

// the stored letters [lo, hi) of a read: where they lie (TSV path), or unpacked into `scratch` (side-car path)
const char *seq_window(const Partition &P, const Read &r, i64 lo, i64 hi, std::vector<char> &scratch) {
    if (r.seq.p) return r.seq.p + lo;
    static const char LUT[4] = {'A', 'C', 'G', 'T'};
    const size_t m = (size_t)(hi - lo);
    if (scratch.size() < m + 8) scratch.resize(m + 64);
    char *dst = scratch.data();
    const uint64_t g0 = r.seq_g0 + (uint64_t)lo;
    for (size_t t = 0; t < m; ++t) { const uint64_t g = g0 + t; dst[t] = LUT[(P.packed[(size_t)(g >> 2)] >> ((g & 3) * 2)) & 3]; }
    if (P.n_exc) {                                            // bytes other than upper-case ACGT (rare): the exception list, ascending
        const uint64_t *e = std::lower_bound(P.exc_pos, P.exc_pos + P.n_exc, g0);
        for (; e < P.exc_pos + P.n_exc && *e < g0 + m; ++e) dst[*e - g0] = (char)P.exc_ch[e - P.exc_pos];
    }
    return dst;
}

// find_longest_poly (:352-367) over the window [s0, e0) of the read in alignment orientation (for '-' reads the window
// is taken from the end of the stored sequence backwards and the complement letter is searched), fused with the
// caller's choice (:397-408, :427-439): runs of positive local score (+1 match, -2 mismatch, floored at 0), each cut at
// its LAST maximum; of those with length >= 20 and purity >= 0.85 the purest wins, the first one on ties, 'A' runs
// before 'T' runs.  Streaming: no score array.
bool best_poly(const Partition &P, const Read &r, bool minus, i64 s0, i64 e0, PolyRun &best, std::vector<char> &scratch) {
    const i64 n = (i64)r.seq.n;
    if (e0 - s0 == 0) return false;
    i64 count = e0 - s0;
    if (count < 0) count = 0;
    {   // the first element of the window must exist (:355)
        const i64 idx0 = minus ? n - 1 - s0 : s0;
        if (idx0 < 0 || idx0 >= n) throw Fail{"find_longest_poly: sequence index out of range (:355)"};
    }
    if (!minus && s0 + count > n) count = n - s0;
    if (minus && n - 1 - s0 - (count - 1) < 0) count = n - s0;
    // a run shorter than 20 is never reported (:399, :429), so a window of fewer than 20 letters has none
    if (count < 20) return false;
    // both letters in ONE pass over the window; "A runs before T runs, the first of the purest wins" = the best A run unless a T run
    // is strictly purer
    struct Scan { int prev = 0, best_s = 0; bool in_run = false, have = false; i64 run_i = 0, best_i = 0, hits_run = 0, hits_best = 0; PolyRun best{}; };
    Scan sc[2];
    const char tgt[2] = {minus ? 'T' : 'A', minus ? 'A' : 'T'};          // (a '-' read is scanned backwards for the complement letter)
    auto emit = [](Scan &z, char ch) {
        const i64 len = z.best_i + 1 - z.run_i;
        if (len < 20) return;
        const double purity = (double)z.hits_best / (double)len;
        if (purity < 0.85) return;
        if (!z.have || purity > z.best.purity) { z.best = PolyRun{z.run_i, len, purity, ch}; z.have = true; }
    };
    // the window in stored coordinates: [s0, s0 + count) of a '+' read, [n - s0 - count, n - s0) of a '-' read (walked backwards)
    const i64 lo = minus ? n - s0 - count : s0;
    const char *win = seq_window(P, r, lo, lo + count, scratch);
    const char *q = minus ? win + (count - 1) : win;
    const i64 step = minus ? -1 : 1;
    for (i64 t = 0; t < count; ++t, q += step) {
        const char c = *q;
#define FHOST_POLY_STEP(Z, CH, TARGET)                                                                          \
        {                                                                                                       \
            const int m = c == (TARGET);                                                                        \
            const int s = std::max(0, (Z).prev + (m ? 1 : -2));                                                 \
            if (s > 0) {                                                                                        \
                if (!(Z).in_run) { (Z).in_run = true; (Z).run_i = t; (Z).best_s = 0; (Z).hits_run = 0; }        \
                (Z).hits_run += m;                                                                              \
                if (s >= (Z).best_s) { (Z).best_s = s; (Z).best_i = t; (Z).hits_best = (Z).hits_run; }   /* max over (score, index): the last maximum */ \
            } else if ((Z).in_run) { emit((Z), (CH)); (Z).in_run = false; }                                     \
            (Z).prev = s;                                                                                       \
        }
        FHOST_POLY_STEP(sc[0], 'A', tgt[0])
        FHOST_POLY_STEP(sc[1], 'T', tgt[1])
#undef FHOST_POLY_STEP
    }
    if (sc[0].in_run) emit(sc[0], 'A');
    if (sc[1].in_run) emit(sc[1], 'T');
    if (sc[0].have && (!sc[1].have || !(sc[1].best.purity > sc[0].best.purity))) { best = sc[0].best; return true; }
    if (sc[1].have) { best = sc[1].best; return true; }
    return false;
}

struct Tok { char s[48]; };

inline char *put_u(char *p, unsigned long long v) {
    char tmp[24];
    int n = 0;
    do { tmp[n++] = (char)('0' + v % 10); v /= 10; } while (v);
    while (n) *p++ = tmp[--n];
    return p;
}
inline char *put_i(char *p, i64 v) {
    if (v < 0) { *p++ = '-'; return put_u(p, (unsigned long long)(-(v + 1)) + 1ull); }
    return put_u(p, (unsigned long long)v);
}
inline char *put_s(char *p, const char *s) { while (*s) *p++ = *s++; return p; }

// gaps: the tokens of read['gaps'] = sorted(set(...)) (:472), in `toks` (sorted, unique); empty when the read has no '1'
void annotate_read(const Partition &P, const Read &r, const unsigned char *data, i64 S, const int *fp,
                   std::vector<Tok> &toks, std::vector<std::pair<i64, i64>> &runs, std::vector<char> &scratch) {
    toks.clear();
    runs.clear();
    for (i64 i = 0; i < S;) {
        if (data[i] != '1') {                                   // (rows are mostly '0': skip to the next '1' a word at a time)
            const void *nx = memchr(data + i, '1', (size_t)(S - i));
            if (!nx) break;
            i = (i64)(static_cast<const unsigned char *>(nx) - data);
        }
        i64 j = i;
        while (j + 1 < S && data[j + 1] == '1') ++j;
        runs.emplace_back(i, j);
        i = j + 1;
    }
    if (runs.empty()) return;
    const i64 length = (i64)r.seq.n;
    const bool minus = r.strand == '-';
    i64 q_ssc, q_esc, slack;
    query_at_or_after(P, r, fp[runs.front().first], q_ssc, slack);          // segs[f][0]
    query_at_or_before(P, r, fp[runs.back().second + 1], q_esc, slack);     // segs[l][1]
    if (!(0 <= q_ssc && q_ssc <= q_esc && q_esc <= length)) throw Fail{"soft-clip positions out of order (:389)"};
    auto tok = [&]() -> char * { toks.emplace_back(); return toks.back().s; };
    PolyRun b;
    if (best_poly(P, r, minus, 0, q_ssc, b, scratch)) {
        i64 gap = q_ssc - b.first - b.len;
        if (!(0 <= b.first && b.first < q_ssc && 0 <= gap && gap < q_ssc)) throw Fail{"start poly tail out of range (:405,:410)"};
        char *t = tok(); *t++ = 'S'; *t++ = b.ch; *t++ = '_'; t = put_i(t, b.len); *t++ = ':'; t = put_i(t, gap); *t = 0;
        t = tok(); t = put_s(t, "SSC:"); t = put_i(t, b.first); *t = 0;
    } else { char *t = tok(); t = put_s(t, "SSC:"); t = put_i(t, q_ssc); *t = 0; }
    if (best_poly(P, r, minus, q_esc, length, b, scratch)) {
        if (!(0 <= b.first && b.first < length - q_esc && length - q_esc - b.first > 0)) throw Fail{"end poly tail out of range (:435,:441,:450)"};
        char *t = tok(); *t++ = 'E'; *t++ = b.ch; *t++ = '_'; t = put_i(t, b.len); *t++ = ':'; t = put_i(t, b.first); *t = 0;
        t = tok(); t = put_s(t, "ESC:"); t = put_i(t, length - q_esc - b.first); *t = 0;
    } else { char *t = tok(); t = put_s(t, "ESC:"); t = put_i(t, length - q_esc); *t = 0; }
    const size_t n_ends = toks.size();                      // the soft-clip / poly tokens: "E.." and "S.."
    int xa = r.ex0, xb = r.ex0;
    for (size_t k = 0; k + 1 < runs.size(); ++k) {
        i64 last1 = runs[k].second, first2 = runs[k + 1].first, q_a, slack_a, q_b, slack_b;
        query_at_or_before(P, r, fp[last1 + 1], q_a, slack_a, &xb);
        query_at_or_after(P, r, fp[first2], q_b, slack_b, &xa);
        if (!(0 < q_a && q_a <= q_b && q_b < length)) throw Fail{"unaligned gap positions out of order (:462)"};
        i64 size = std::max<i64>(0, q_b - q_a + slack_a + slack_b);
        if (!(0 <= size && size < length && last1 < first2)) throw Fail{"unaligned gap size out of range (:466-468)"};
        char *t = tok(); t = put_i(t, last1); *t++ = '-'; t = put_i(t, first2); *t++ = ':'; t = put_i(t, size); *t = 0;
    }
    // read['gaps'] = sorted(set(...)) on strings (:472).  No two tokens are equal (the gap tokens differ in their first number, the
    // others in their first letters), and a digit sorts before 'E' before 'S': the gap tokens, sorted among themselves, go first,
    // then the (at most three each) E and S tokens -- a rotation and two short sorts instead of one sort of 48-byte records.
    std::rotate(toks.begin(), toks.begin() + (std::ptrdiff_t)n_ends, toks.end());
    const auto less = [](const Tok &a, const Tok &b2) { return strcmp(a.s, b2.s) < 0; };
    const size_t n_gaps = toks.size() - n_ends;
    if (n_gaps > 1) {
        // "last1-first2:size" with last1 ascending: already sorted unless the first numbers differ in their digit counts
        bool sorted = true;
        for (size_t k = 1; k < n_gaps && sorted; ++k) sorted = less(toks[k - 1], toks[k]);
        if (!sorted) std::sort(toks.begin(), toks.begin() + (std::ptrdiff_t)n_gaps, less);
    }
    std::sort(toks.begin() + (std::ptrdiff_t)n_gaps, toks.end(), less);
}

// ---- binary side-car of a partition ------------------------------------------
// split_<contig>_<tint>.fsc holds what parse_partition() produces from the two TSVs: the flat exon / CIGAR arrays,
// the read_reps grouping and the sequences packed two bits per base (bytes other than upper-case ACGT are kept in
// an exception list, so the round trip is exact).  It is bound to its TSVs by their sizes and mtimes and is only an
// accelerator: the TSVs stay the stage's contract (py/freddie_split.py:445-481 writes them, :121-185 reads them).
const char FSC_MAGIC[8] = {'F', 'S', 'C', '3', 0, 0, 0, 0};     // ('2': the checksum was one FNV lane)

struct FscHeader {
    char magic[8];
    uint64_t split_size; int64_t split_mtime_ns; uint64_t reads_size; int64_t reads_mtime_ns;
    int64_t id, read_count;
    uint64_t n_iv, n_reads, n_exons, n_cigar, n_reps, name_bytes, chr_bytes, read_chr_bytes, seq_bases, n_exc;
    uint64_t payload_bytes, checksum;
};

bool stat_file(const char *path, uint64_t &size, int64_t &mtime_ns) {
    struct stat st;
    if (stat(path, &st) != 0) return false;
    size = (uint64_t)st.st_size;
    mtime_ns = (int64_t)st.st_mtim.tv_sec * 1000000000ll + (int64_t)st.st_mtim.tv_nsec;
    return true;
}

// FNV-1a over 8-byte words (sections are 8-aligned), in FOUR interleaved lanes (word i goes to lane i mod 4) that are folded at the
// end: one lane is a chain of dependent multiplies -- 1.4 GB/s, as much as the rest of a side-car load -- four run side by side
// (format "FSC3"; "FSC2" had one lane)
uint64_t fsc_checksum(const unsigned char *p, size_t n, uint64_t h = 1469598103934665603ull) {
    const uint64_t K = 1099511628211ull;
    uint64_t a = h, b = h ^ 0x9e3779b97f4a7c15ull, c = h ^ 0xc2b2ae3d27d4eb4full, d = h ^ 0x165667b19e3779f9ull;
    const size_t w = n / 8;
    size_t i = 0;
    for (; i + 4 <= w; i += 4) {
        uint64_t v[4];
        memcpy(v, p + i * 8, 32);
        a = (a ^ v[0]) * K; b = (b ^ v[1]) * K; c = (c ^ v[2]) * K; d = (d ^ v[3]) * K;
    }
    for (; i < w; ++i) { uint64_t v; memcpy(&v, p + i * 8, 8); a = (a ^ v) * K; }
    for (size_t j = w * 8; j < n; ++j) a = (a ^ p[j]) * K;
    h = a;
    h = (h ^ b) * K; h = (h ^ c) * K; h = (h ^ d) * K;
    return h;
}

// the header's counts steer the loader, so the checksum covers the header (with its checksum field zeroed) and the payload
uint64_t fsc_total_checksum(FscHeader h, const unsigned char *payload, size_t n) {
    h.checksum = 0;
    return fsc_checksum(payload, n, fsc_checksum(reinterpret_cast<const unsigned char *>(&h), sizeof h));
}

struct Sink {
    std::string buf;
    template <typename T> void put(const T *p, size_t n) {
        buf.append(reinterpret_cast<const char *>(p), n * sizeof(T));
        buf.append((8 - buf.size() % 8) % 8, '\0');
    }
};
struct Source {
    const unsigned char *p, *end;
    bool ok = true;
    // n comes from the (untrusted) header: a count whose byte size does not fit what is left of the payload --
    // including one that would wrap around -- fails the load instead of reaching a resize() or a pointer bump
    template <typename T> bool fits(size_t n, size_t &padded) const {
        const size_t left = (size_t)(end - p);
        if (n > left / sizeof(T)) return false;
        const size_t bytes = n * sizeof(T);
        padded = bytes + (8 - bytes % 8) % 8;
        return padded <= left;
    }
    // a section that is only read while loading: a pointer into the file image instead of a copy (sections are 8-aligned)
    template <typename T> const T *view(size_t n) {
        size_t padded = 0;
        if (!ok || !fits<T>(n, padded)) { ok = false; return nullptr; }
        const T *r = reinterpret_cast<const T *>(p);
        p += padded;
        return r;
    }
    template <typename T> void get(std::vector<T> &v, size_t n) {
        size_t padded = 0;
        if (!ok || !fits<T>(n, padded)) { ok = false; return; }
        const size_t bytes = n * sizeof(T);
        v.resize(n);
        if (bytes) memcpy(v.data(), p, bytes);
        p += padded;
    }
};

inline int base_code(unsigned char c) { return c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : c == 'T' ? 3 : -1; }

bool write_sidecar(const Partition &P, const char *split_path, const char *reads_path, const char *out_path, std::string &err) {
    FscHeader h;
    memset(&h, 0, sizeof h);
    memcpy(h.magic, FSC_MAGIC, 8);
    // bound to the TSVs as they were when the partition was read, not as they are now: a TSV replaced in between must
    // make this side-car stale
    if (P.have_stat) { h.split_size = P.split_size; h.split_mtime_ns = P.split_mtime_ns; h.reads_size = P.reads_size; h.reads_mtime_ns = P.reads_mtime_ns; }
    else if (!stat_file(split_path, h.split_size, h.split_mtime_ns) || !stat_file(reads_path, h.reads_size, h.reads_mtime_ns)) {
        err = std::string("cannot stat ") + split_path + " / " + reads_path; return false;
    }
    const size_t n = P.reads.size();
    std::vector<int64_t> read_id(n);
    std::vector<int32_t> read_ex_off(n + 1), read_rep(n);
    std::vector<uint8_t> strand(n);
    std::vector<uint32_t> name_off(n + 1), chr_off;
    std::vector<uint64_t> seq_off(n + 1), exc_pos;
    std::vector<uint8_t> exc_ch;
    std::string names, read_chrs;
    bool chr_differs = false;
    for (const Read &r : P.reads) chr_differs |= !(r.chr.n == P.chr.size() && memcmp(r.chr.p, P.chr.data(), r.chr.n) == 0);
    uint64_t bases = 0;
    for (size_t i = 0; i < n; ++i) {
        const Read &r = P.reads[i];
        read_id[i] = r.id; read_ex_off[i] = r.ex0; read_rep[i] = r.rep; strand[i] = (uint8_t)r.strand;
        name_off[i] = (uint32_t)names.size(); names.append(r.name.p, r.name.n);
        if (chr_differs) { chr_off.push_back((uint32_t)read_chrs.size()); read_chrs.append(r.chr.p, r.chr.n); }
        seq_off[i] = bases; bases += r.seq.n;
    }
    read_ex_off[n] = (int32_t)P.ts.size(); name_off[n] = (uint32_t)names.size(); seq_off[n] = bases;
    if (chr_differs) chr_off.push_back((uint32_t)read_chrs.size());
    std::vector<uint8_t> packed((size_t)((bases + 3) / 4), 0);
    std::vector<char> unpacked;
    for (size_t i = 0; i < n; ++i) {
        const Str &q = P.reads[i].seq;
        const char *qp = seq_window(P, P.reads[i], 0, (i64)q.n, unpacked);     // (a partition that itself came from a side-car)
        uint64_t g = seq_off[i];
        for (size_t t = 0; t < q.n; ++t, ++g) {
            int c = base_code((unsigned char)qp[t]);
            if (c < 0) { exc_pos.push_back(g); exc_ch.push_back((uint8_t)qp[t]); c = 0; }
            packed[(size_t)(g >> 2)] |= (uint8_t)(c << ((g & 3) * 2));
        }
    }
    h.id = P.id; h.read_count = P.read_count;
    h.n_iv = P.iv_s.size(); h.n_reads = n; h.n_exons = P.ts.size(); h.n_cigar = P.cig_len.size();
    h.n_reps = P.rep_first_read.size(); h.name_bytes = names.size(); h.chr_bytes = P.chr.size();
    h.read_chr_bytes = chr_differs ? read_chrs.size() + 1 : 0;       // 0 = every read carries the tint's chr
    h.seq_bases = bases; h.n_exc = exc_pos.size();
    Sink s;
    s.put(P.chr.data(), P.chr.size());
    s.put(P.iv_s.data(), P.iv_s.size()); s.put(P.iv_e.data(), P.iv_e.size());
    s.put(read_id.data(), n); s.put(read_ex_off.data(), n + 1); s.put(read_rep.data(), n); s.put(strand.data(), n);
    s.put(name_off.data(), n + 1); s.put(names.data(), names.size());
    if (chr_differs) { s.put(chr_off.data(), n + 1); s.put(read_chrs.data(), read_chrs.size()); }
    s.put(P.ts.data(), P.ts.size()); s.put(P.te.data(), P.te.size()); s.put(P.qs.data(), P.qs.size()); s.put(P.qe.data(), P.qe.size());
    s.put(P.cig_off.data(), P.cig_off.size()); s.put(P.cig_len.data(), P.cig_len.size()); s.put(P.cig_op.data(), P.cig_op.size());
    s.put(P.rep_first_read.data(), P.rep_first_read.size()); s.put(P.rep_weight.data(), P.rep_weight.size());
    s.put(seq_off.data(), n + 1); s.put(packed.data(), packed.size());
    s.put(exc_pos.data(), exc_pos.size()); s.put(exc_ch.data(), exc_ch.size());
    h.payload_bytes = s.buf.size();
    h.checksum = fsc_total_checksum(h, reinterpret_cast<const unsigned char *>(s.buf.data()), s.buf.size());
    std::string tmp = std::string(out_path) + ".tmp";
    FILE *f = fopen(tmp.c_str(), "wb");
    bool ok = f && fwrite(&h, sizeof h, 1, f) == 1 && fwrite(s.buf.data(), 1, s.buf.size(), f) == s.buf.size();
    if (f) ok = fclose(f) == 0 && ok;
    if (ok) ok = rename(tmp.c_str(), out_path) == 0;                 // readers never see a half-written side-car
    if (!ok) { remove(tmp.c_str()); err = std::string("cannot write ") + out_path; }
    return ok;
}

// Returns true when the side-car exists, belongs to exactly these TSVs and is intact; P is then what
// parse_partition() would have produced.  Any mismatch returns false (the caller parses the TSVs instead).
bool load_sidecar(const char *sidecar_path, const char *split_path, const char *reads_path, Partition &P, bool verify) {
    Mapping &blob = P.map_fsc;                             // stays mapped on success: names, contigs and the packed sequences are views into it
    struct Closer { Mapping &m; bool keep = false; ~Closer() { if (!keep) m.close(); } } closer{blob};
    if (!blob.open(sidecar_path, nullptr, nullptr) || blob.n < sizeof(FscHeader)) return false;
    FscHeader h;
    memcpy(&h, blob.p, sizeof h);
    if (memcmp(h.magic, FSC_MAGIC, 8) != 0 || blob.n != sizeof h + h.payload_bytes) return false;
    uint64_t sz; int64_t mt;
    if (!stat_file(split_path, sz, mt) || sz != h.split_size || mt != h.split_mtime_ns) return false;
    if (!stat_file(reads_path, sz, mt) || sz != h.reads_size || mt != h.reads_mtime_ns) return false;
    const unsigned char *pay = reinterpret_cast<const unsigned char *>(blob.p) + sizeof h;
    if (verify && fsc_total_checksum(h, pay, (size_t)h.payload_bytes) != h.checksum) return false;
    const size_t n = (size_t)h.n_reads;
    if (h.n_reads > (1ull << 31) || h.n_exons > (1ull << 31) || h.n_cigar > (1ull << 31) || h.n_reps > h.n_reads ||
        (int64_t)h.n_reads != h.read_count || h.seq_bases > (1ull << 40) || h.name_bytes > (1ull << 32) ||
        h.read_chr_bytes > (1ull << 32) || h.chr_bytes > (1ull << 20)) return false;
    Source s{pay, pay + h.payload_bytes};
    const char *chr = s.view<char>((size_t)h.chr_bytes);
    s.get(P.iv_s, (size_t)h.n_iv); s.get(P.iv_e, (size_t)h.n_iv);
    const int64_t *read_id = s.view<int64_t>(n);
    const int32_t *read_ex_off = s.view<int32_t>(n + 1), *read_rep = s.view<int32_t>(n);
    const uint8_t *strand = s.view<uint8_t>(n);
    const uint32_t *name_off = s.view<uint32_t>(n + 1);
    const char *names = s.view<char>((size_t)h.name_bytes);
    const uint32_t *chr_off = nullptr;
    const char *read_chrs = nullptr;
    const size_t read_chr_len = h.read_chr_bytes ? (size_t)h.read_chr_bytes - 1 : 0;
    if (h.read_chr_bytes) { chr_off = s.view<uint32_t>(n + 1); read_chrs = s.view<char>(read_chr_len); }
    s.get(P.ts, (size_t)h.n_exons); s.get(P.te, (size_t)h.n_exons); s.get(P.qs, (size_t)h.n_exons); s.get(P.qe, (size_t)h.n_exons);
    s.get(P.cig_off, (size_t)h.n_exons + 1); s.get(P.cig_len, (size_t)h.n_cigar); s.get(P.cig_op, (size_t)h.n_cigar);
    s.get(P.rep_first_read, (size_t)h.n_reps); s.get(P.rep_weight, (size_t)h.n_reps);
    const uint64_t *seq_off = s.view<uint64_t>(n + 1);
    const uint8_t *packed = s.view<uint8_t>((size_t)((h.seq_bases + 3) / 4));
    const uint64_t *exc_pos = s.view<uint64_t>((size_t)h.n_exc);
    const uint8_t *exc_ch = s.view<uint8_t>((size_t)h.n_exc);
    if (!s.ok || s.p != s.end) return false;
    // structural checks: every offset table must be monotone and end at its array's size
    if (read_ex_off[0] != 0 || read_ex_off[n] != (int32_t)h.n_exons || name_off[n] != h.name_bytes || seq_off[n] != h.seq_bases ||
        P.cig_off[(size_t)h.n_exons] != (int32_t)h.n_cigar) return false;
    for (size_t i = 0; i < n; ++i) {
        if (read_ex_off[i] > read_ex_off[i + 1] || name_off[i] > name_off[i + 1] || seq_off[i] > seq_off[i + 1]) return false;
        if (seq_off[i + 1] - seq_off[i] > 0xffffffffull) return false;
        if (read_rep[i] < 0 || (uint64_t)read_rep[i] >= h.n_reps) return false;
        if (h.read_chr_bytes && (chr_off[i] > chr_off[i + 1] || chr_off[i + 1] > read_chr_len)) return false;
    }
    for (size_t x = 0; x < (size_t)h.n_exons; ++x) if (P.cig_off[x] < 0 || P.cig_off[x] > P.cig_off[x + 1]) return false;
    for (size_t r = 0; r < (size_t)h.n_reps; ++r) if (P.rep_first_read[r] < 0 || (size_t)P.rep_first_read[r] >= n) return false;
    for (size_t k = 0; k < (size_t)h.n_exc; ++k) if (exc_pos[k] >= h.seq_bases || (k && exc_pos[k] <= exc_pos[k - 1])) return false;
    P.chr.assign(chr, chr + h.chr_bytes);
    P.id = h.id; P.read_count = h.read_count;
    P.split_size = h.split_size; P.split_mtime_ns = h.split_mtime_ns; P.reads_size = h.reads_size; P.reads_mtime_ns = h.reads_mtime_ns;
    P.have_stat = true;
    P.packed = packed; P.exc_pos = exc_pos; P.exc_ch = exc_ch; P.n_exc = (size_t)h.n_exc;
    P.reads.resize(n);
    for (size_t i = 0; i < n; ++i) {
        Read &r = P.reads[i];
        r.id = read_id[i]; r.tint = h.id; r.strand = (char)strand[i]; r.ex0 = read_ex_off[i]; r.ex1 = read_ex_off[i + 1]; r.rep = read_rep[i];
        r.name.p = names + name_off[i]; r.name.n = name_off[i + 1] - name_off[i];
        if (h.read_chr_bytes) { r.chr.p = read_chrs + chr_off[i]; r.chr.n = chr_off[i + 1] - chr_off[i]; }
        else { r.chr.p = P.chr.data(); r.chr.n = (uint32_t)P.chr.size(); }
        r.seq.p = nullptr; r.seq.n = (uint32_t)(seq_off[i + 1] - seq_off[i]); r.seq_g0 = seq_off[i];
    }
    P.from_sidecar = true;
    closer.keep = true;
    return true;
}

// nothing may leave an extern "C" entry point (or a worker thread) as a C++ exception
void parse_guarded(const char *split_path, const char *reads_path, Partition &P) {
    try { parse_partition(split_path, reads_path, P); }
    catch (const std::exception &e) { P.err = std::string(split_path) + ": " + e.what(); }
    catch (...) { P.err = std::string(split_path) + ": internal error while parsing"; }
}

template <typename F>
void parallel_for(int n, int n_threads, F fn) {
    if (n_threads < 1) n_threads = 1;
    if (n_threads > n) n_threads = n;
    if (n_threads <= 1) { for (int i = 0; i < n; ++i) fn(i); return; }
    std::atomic<int> next(0);
    std::vector<std::thread> pool;
    for (int t = 0; t < n_threads; ++t)
        pool.emplace_back([&]() { for (int i; (i = next.fetch_add(1)) < n;) fn(i); });
    for (std::thread &t : pool) t.join();
}

}  // namespace

struct fhost_batch {
    std::vector<Partition> parts;
    std::string err;
    std::vector<int64_t> part_iv_off, part_rep_off, rep_exon_off;
    std::vector<int32_t> iv_start, iv_end, rep_weight, ex_ts, ex_te;
    int64_t n_reads = 0;
    int32_t n_from_sidecar = 0;
};

namespace {
// the flat arrays of fseg_batch: offsets first (serial, cheap), then every partition copies its own slice (parallel)
void flatten(fhost_batch *b, int n_threads) {
    const size_t n = b->parts.size();
    b->part_iv_off.assign(n + 1, 0); b->part_rep_off.assign(n + 1, 0);
    std::vector<int64_t> part_ex_off(n + 1, 0);
    for (size_t p = 0; p < n; ++p) {
        const Partition &P = b->parts[p];
        int64_t ex = 0;
        for (int f : P.rep_first_read) { const Read &rd = P.reads[(size_t)f]; ex += rd.ex1 - rd.ex0; }
        b->part_iv_off[p + 1] = b->part_iv_off[p] + (int64_t)P.iv_s.size();
        b->part_rep_off[p + 1] = b->part_rep_off[p] + (int64_t)P.rep_first_read.size();
        part_ex_off[p + 1] = part_ex_off[p] + ex;
        b->n_reads += (int64_t)P.reads.size();
    }
    b->iv_start.resize((size_t)b->part_iv_off[n]); b->iv_end.resize((size_t)b->part_iv_off[n]);
    b->rep_weight.resize((size_t)b->part_rep_off[n]); b->rep_exon_off.resize((size_t)b->part_rep_off[n] + 1);
    b->ex_ts.resize((size_t)part_ex_off[n]); b->ex_te.resize((size_t)part_ex_off[n]);
    b->rep_exon_off[0] = 0;
    parallel_for((int)n, n_threads, [&](int pi) {
        const Partition &P = b->parts[(size_t)pi];
        const size_t k0 = (size_t)b->part_iv_off[(size_t)pi], r0 = (size_t)b->part_rep_off[(size_t)pi];
        if (!P.iv_s.empty()) { memcpy(&b->iv_start[k0], P.iv_s.data(), P.iv_s.size() * 4); memcpy(&b->iv_end[k0], P.iv_e.data(), P.iv_e.size() * 4); }
        int64_t e = part_ex_off[(size_t)pi];
        for (size_t r = 0; r < P.rep_first_read.size(); ++r) {
            const Read &rd = P.reads[(size_t)P.rep_first_read[r]];
            const int m = rd.ex1 - rd.ex0;
            if (m > 0) { memcpy(&b->ex_ts[(size_t)e], &P.ts[(size_t)rd.ex0], (size_t)m * 4); memcpy(&b->ex_te[(size_t)e], &P.te[(size_t)rd.ex0], (size_t)m * 4); }
            e += m;
            b->rep_exon_off[r0 + r + 1] = e;
            b->rep_weight[r0 + r] = P.rep_weight[r];
        }
    });
}
}  // namespace

extern "C" {

fhost_batch *fhost_load(const char *const *split_paths, const char *const *reads_paths, int32_t n, int32_t n_threads) {
    return fhost_load_sidecar(split_paths, reads_paths, nullptr, n, n_threads, 0);
}

fhost_batch *fhost_load_sidecar(const char *const *split_paths, const char *const *reads_paths, const char *const *sidecar_paths,
                                int32_t n, int32_t n_threads, int32_t verify_checksum) {
    fhost_batch *b = new (std::nothrow) fhost_batch();
    if (!b) return nullptr;
    if (n <= 0) { b->err = "fhost_load: empty batch"; return b; }
    try {
        b->parts = std::vector<Partition>((size_t)n);
        std::atomic<int> hits(0);
        parallel_for(n, n_threads, [&](int i) {
            Partition &P = b->parts[(size_t)i];
            bool hit = false;
            try {        // a damaged side-car must never be worse than a missing one: whatever it throws, the TSVs are parsed instead
                hit = sidecar_paths && sidecar_paths[i] && load_sidecar(sidecar_paths[i], split_paths[i], reads_paths[i], P, verify_checksum != 0);
            } catch (...) { hit = false; }
            if (hit) { hits.fetch_add(1); return; }
            P.reset();
            parse_guarded(split_paths[i], reads_paths[i], P);
        });
        b->n_from_sidecar = hits.load();
        for (const Partition &P : b->parts) if (!P.err.empty()) { b->err = P.err; return b; }
        flatten(b, n_threads);
    } catch (const std::exception &e) {
        b->err = std::string("fhost_load: ") + e.what();
    } catch (...) {
        b->err = "fhost_load: internal error";
    }
    return b;
}

int32_t fhost_n_from_sidecar(const fhost_batch *b) { return b->n_from_sidecar; }

int32_t fhost_sidecar_write(fhost_batch *b, const char *const *split_paths, const char *const *reads_paths,
                            const char *const *sidecar_paths, int32_t n_threads) {
    if (!b || !b->err.empty()) return 1;
    std::mutex err_mutex;
    parallel_for((int)b->parts.size(), n_threads, [&](int p) {
        std::string err;
        bool ok = false;
        if (b->parts[(size_t)p].from_sidecar) return;        // loaded from this very side-car, which was found fresh: nothing to write
        try { ok = write_sidecar(b->parts[(size_t)p], split_paths[p], reads_paths[p], sidecar_paths[p], err); }
        catch (const std::exception &e) { err = std::string(sidecar_paths[p]) + ": " + e.what(); }
        catch (...) { err = std::string(sidecar_paths[p]) + ": internal error"; }
        if (!ok) {
            std::lock_guard<std::mutex> lock(err_mutex);
            if (b->err.empty()) b->err = err;
        }
    });
    return b->err.empty() ? 0 : 2;
}

void fhost_free(fhost_batch *b) { delete b; }

// ---- the split directory's listing and the empty .log files: two loops of system calls the Python driver used to make one by one ----
struct fhost_listing {
    std::string err;
    std::vector<std::string> contigs;
    std::vector<int32_t> contig_of;
    std::vector<int64_t> tint, size;
};

fhost_listing *fhost_discover(const char *split_dir, int32_t n_threads) {
    fhost_listing *l = new (std::nothrow) fhost_listing();
    if (!l) return nullptr;
    try {
        std::vector<std::string> paths;
        DIR *top = opendir(split_dir);
        if (!top) { l->err = std::string("cannot list ") + split_dir; return l; }
        std::vector<std::string> dirs;
        while (struct dirent *e = readdir(top)) {
            if (e->d_name[0] == '.' && (e->d_name[1] == 0 || (e->d_name[1] == '.' && e->d_name[2] == 0))) continue;
            bool is_dir = e->d_type == DT_DIR;
            if (e->d_type == DT_UNKNOWN || e->d_type == DT_LNK) {           // file systems that do not say: ask (os.DirEntry.is_dir follows links too)
                struct stat st;
                is_dir = stat((std::string(split_dir) + "/" + e->d_name).c_str(), &st) == 0 && S_ISDIR(st.st_mode);
            }
            if (is_dir) dirs.push_back(e->d_name);
        }
        closedir(top);
        for (const std::string &c : dirs) {
            const std::string dpath = std::string(split_dir) + "/" + c;
            DIR *d = opendir(dpath.c_str());
            if (!d) { l->err = "cannot list " + dpath; return l; }
            const int32_t ci = (int32_t)l->contigs.size();
            l->contigs.push_back(c);
            while (struct dirent *e = readdir(d)) {
                const char *nm = e->d_name;
                const size_t n = strlen(nm);
                if (n < 11 || strncmp(nm, "split_", 6) != 0 || strcmp(nm + n - 4, ".tsv") != 0) continue;      // split_*.tsv (:853)
                const char *us = nm + n - 4;
                while (us > nm && us[-1] != '_') --us;                     // int(name[:-4].split("_")[-1])
                i64 id = 0;
                if (!parse_uint(us, nm + n - 4, id)) {          // (the message first: the entry's name lives in the directory stream)
                    l->err = dpath + "/" + nm + ": the tint id in the file name is not a number";
                    closedir(d);
                    return l;
                }
                l->contig_of.push_back(ci); l->tint.push_back(id);
                paths.push_back(dpath + "/" + nm);
            }
            closedir(d);
        }
        l->size.assign(paths.size(), 0);
        std::atomic<bool> bad(false);
        parallel_for((int)paths.size(), n_threads, [&](int i) {
            struct stat st;
            if (stat(paths[(size_t)i].c_str(), &st) != 0) bad.store(true); else l->size[(size_t)i] = (int64_t)st.st_size;
        });
        if (bad.load()) l->err = std::string("cannot stat a split file under ") + split_dir;
    } catch (const std::exception &e) { l->err = std::string("fhost_discover: ") + e.what(); }
    catch (...) { l->err = "fhost_discover: internal error"; }
    return l;
}
void fhost_listing_free(fhost_listing *l) { delete l; }
const char *fhost_listing_error(const fhost_listing *l) { return l ? l->err.c_str() : "null listing"; }
int64_t fhost_listing_n(const fhost_listing *l) { return (int64_t)l->tint.size(); }
int32_t fhost_listing_n_contigs(const fhost_listing *l) { return (int32_t)l->contigs.size(); }
const char *fhost_listing_contig(const fhost_listing *l, int32_t k) { return (k >= 0 && (size_t)k < l->contigs.size()) ? l->contigs[(size_t)k].c_str() : ""; }
const int32_t *fhost_listing_contig_of(const fhost_listing *l) { return l->contig_of.data(); }
const int64_t *fhost_listing_tint(const fhost_listing *l) { return l->tint.data(); }
const int64_t *fhost_listing_size(const fhost_listing *l) { return l->size.data(); }

int32_t fhost_touch(const char *const *paths, int32_t n, int32_t n_threads) {
    if (n <= 0) return 0;
    std::atomic<int> failed(0);
    try {
        parallel_for(n, n_threads, [&](int i) {
            const int fd = ::open(paths[i], O_WRONLY | O_CREAT | O_TRUNC | O_CLOEXEC, 0666);
            if (fd < 0 || ::close(fd) != 0) failed.fetch_add(1);
        });
    } catch (...) { return 1; }
    return failed.load() ? 2 : 0;
}
const char *fhost_error(const fhost_batch *b) { return b ? b->err.c_str() : "null batch"; }
int32_t fhost_n_part(const fhost_batch *b) { return (int32_t)b->parts.size(); }
int64_t fhost_n_reads(const fhost_batch *b) { return b->n_reads; }
const int64_t *fhost_part_iv_off(const fhost_batch *b) { return b->part_iv_off.data(); }
const int32_t *fhost_iv_start(const fhost_batch *b) { return b->iv_start.data(); }
const int32_t *fhost_iv_end(const fhost_batch *b) { return b->iv_end.data(); }
const int64_t *fhost_part_rep_off(const fhost_batch *b) { return b->part_rep_off.data(); }
const int32_t *fhost_rep_weight(const fhost_batch *b) { return b->rep_weight.data(); }
const int64_t *fhost_rep_exon_off(const fhost_batch *b) { return b->rep_exon_off.data(); }
const int32_t *fhost_ex_ts(const fhost_batch *b) { return b->ex_ts.data(); }
const int32_t *fhost_ex_te(const fhost_batch *b) { return b->ex_te.data(); }

static int32_t write_impl(fhost_batch *b, const int64_t *part_final_off, const int32_t *final_pos, const int64_t *label_off,
                          const uint8_t *labels, const char *const *out_paths, int32_t n_threads, bool packed) {
    if (!b || !b->err.empty()) return 1;
    // packed labels: two bits each, label byte g of the arena at bits 2(g & 3).. of packed byte g >> 2 (fseg_results_packed);
    // a packed byte becomes four ASCII digits through a table
    static char DIGITS4[256][4];
    static std::once_flag digits_once;
    std::call_once(digits_once, []() { for (int v = 0; v < 256; ++v) for (int k = 0; k < 4; ++k) DIGITS4[v][k] = (char)('0' + ((v >> (2 * k)) & 3)); });
    std::mutex err_mutex;
    const int n = (int)b->parts.size();
    auto set_err = [&](const std::string &m) { std::lock_guard<std::mutex> lock(err_mutex); if (b->err.empty()) b->err = m; };
    parallel_for(n, n_threads, [&](int p) {
        // the output of a partition is assembled in one buffer that the worker thread keeps from partition to partition
        static thread_local std::vector<char> out;
        static thread_local std::vector<Tok> toks;
        static thread_local std::vector<std::pair<i64, i64>> runs;
        static thread_local std::vector<char> scratch;
        try {
            const Partition &P = b->parts[(size_t)p];
            const int *fp = final_pos + part_final_off[p];
            const i64 F = part_final_off[p + 1] - part_final_off[p], S = F - 1;
            const i64 Sc = S > 0 ? S : 0;
            size_t need = P.chr.size() + 64 + (size_t)F * 12;
            for (const Read &r : P.reads) need += (size_t)Sc + r.name.n + r.chr.n + 64 + 48 * 6;
            if (out.size() < need) out.resize(need + need / 4);
            char *w = out.data();
            *w++ = '#'; memcpy(w, P.chr.data(), P.chr.size()); w += P.chr.size(); *w++ = '\t';
            w = put_i(w, P.id); *w++ = '\t';
            for (i64 i = 0; i < F; ++i) { if (i) *w++ = ','; w = put_i(w, fp[i]); }
            *w++ = '\n';
            for (const Read &r : P.reads) {
                {   // room for the line up to and including its label field (the gap tokens are accounted for below)
                    const size_t used = (size_t)(w - out.data()), line_max = (size_t)Sc + r.name.n + r.chr.n + 64;
                    if (used + line_max > out.size()) { out.resize((used + line_max) * 2); w = out.data() + used; }
                }
                w = put_i(w, r.id); *w++ = '\t';
                memcpy(w, r.name.p, r.name.n); w += r.name.n; *w++ = '\t';
                memcpy(w, r.chr.p, r.chr.n); w += r.chr.n; *w++ = '\t';
                *w++ = r.strand; *w++ = '\t';
                w = put_i(w, r.tint); *w++ = '\t';
                // the rep's label row goes straight into the line; the annotation reads it from there
                const i64 g0 = label_off[p] + (i64)r.rep * S;
                char *row = w;
                if (Sc && !packed) memcpy(row, labels + g0, (size_t)Sc);
                if (Sc && packed) {
                    i64 g = g0, i = 0;
                    for (; i < Sc && (g & 3); ++i, ++g) row[i] = (char)('0' + ((labels[g >> 2] >> ((g & 3) * 2)) & 3));
                    for (; i + 4 <= Sc; i += 4, g += 4) memcpy(row + i, DIGITS4[labels[g >> 2]], 4);
                    for (; i < Sc; ++i, ++g) row[i] = (char)('0' + ((labels[g >> 2] >> ((g & 3) * 2)) & 3));
                }
                w += Sc;
                annotate_read(P, r, reinterpret_cast<const unsigned char *>(row), S, fp, toks, runs, scratch);
                {   // a read with many label runs has many gap tokens: make room before writing them
                    const size_t used = (size_t)(w - out.data()), rest_max = 8 + toks.size() * 48;
                    if (used + rest_max > out.size()) { out.resize((used + rest_max) * 2); w = out.data() + used; }
                }
                *w++ = '\t';
                for (const Tok &g : toks) { w = put_s(w, g.s); *w++ = ','; }
                *w++ = '\n';
            }
            const size_t total = (size_t)(w - out.data());
            int fd = ::open(out_paths[p], O_WRONLY | O_CREAT | O_TRUNC | O_CLOEXEC, 0666);
            bool ok = fd >= 0;
            size_t done = 0;
            while (ok && done < total) {
                ssize_t k = ::write(fd, out.data() + done, total - done);
                if (k < 0) { if (errno == EINTR) continue; ok = false; }
                else done += (size_t)k;
            }
            if (fd >= 0 && ::close(fd) != 0) ok = false;
            if (!ok) set_err(std::string("cannot write ") + out_paths[p]);
        } catch (const Fail &f) {
            set_err(std::string(out_paths[p]) + ": " + f.what + " (reference: py/freddie_segment.py)");
        } catch (const std::exception &e) {
            set_err(std::string(out_paths[p]) + ": " + e.what());
        } catch (...) {
            set_err(std::string(out_paths[p]) + ": internal error");
        }
    });
    return b->err.empty() ? 0 : 2;
}

int32_t fhost_write(fhost_batch *b, const int64_t *part_final_off, const int32_t *final_pos, const int64_t *label_off,
                    const uint8_t *labels, const char *const *out_paths, int32_t n_threads) {
    return write_impl(b, part_final_off, final_pos, label_off, labels, out_paths, n_threads, false);
}

int32_t fhost_write_packed(fhost_batch *b, const int64_t *part_final_off, const int32_t *final_pos, const int64_t *label_off,
                           const uint8_t *labels2, const char *const *out_paths, int32_t n_threads) {
    return write_impl(b, part_final_off, final_pos, label_off, labels2, out_paths, n_threads, true);
}

}  // extern "C"

#ifdef FREDDIE_SOURCE_HASH
/* what this binary was built from (freddie_amd/build.py looks for the marker in the file) */
static const char freddie_source_stamp[] __attribute__((used)) = "FREDDIE_SRC_HASH=" FREDDIE_SOURCE_HASH;
#endif
