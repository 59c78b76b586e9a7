// Host-side helpers of the C ABI that need no GPU (include/spliser.h).
#include <cstdint>
#include <cstring>

#include "../../include/spliser.h"
#include "spl_error.h"
#include "spl_fmt.h"

namespace {
inline int64_t floor_div2(int64_t v) { return v >= 0 ? v / 2 : -((-v + 1) / 2); } // Python's v // 2
} // namespace

// binary_gene_search (SpliSER_v0_1_8.py:118-173) for a batch of positions, probe for probe: overlapping genes make the
// list only partially ordered, so the outcome depends on the exact probe sequence -- the lo/hi trackers, the idx == 1
// special case, the stop when an index repeats and the last-ditch sweep over offsets -3..2 that re-bases itself on every
// hit and never looks at the final list element are all kept.  Strand bytes: '+', '-' or 0 (anything else).
extern "C" int spl_gene_search(const int64_t *left, const int64_t *right, const uint8_t *gene_strand, int64_t n_genes,
                               const int64_t *q_pos, const uint8_t *q_strand, int64_t n_queries, int is_stranded, int32_t *out)
{
    if (n_queries < 0 || n_genes < 0) return spl_set_error(SPL_ERR_ARG, "spl_gene_search: negative count");
    if (n_queries && (!q_pos || !q_strand || !out)) return spl_set_error(SPL_ERR_ARG, "spl_gene_search: null query arrays");
    if (n_genes && (!left || !right || !gene_strand)) return spl_set_error(SPL_ERR_ARG, "spl_gene_search: null gene arrays");
    if (n_genes > 0x7fffffff) return spl_set_error(SPL_ERR_ARG, "spl_gene_search: too many genes");
    const int64_t n = n_genes;
    for (int64_t q = 0; q < n_queries; ++q) {
        if (n == 0) { out[q] = -1; continue; }
        const int64_t pos = q_pos[q];
        const uint8_t qs = q_strand[q];
        const bool free = !is_stranded || (qs != '+' && qs != '-');
        auto hit = [&](int64_t g) { return left[g] <= pos && pos <= right[g] && (free || qs == gene_strand[g]); };
        int64_t idx = n / 2, hi = n, lo = 0, last = -1, nxt = n / 2;
        int64_t found_at = -1;
        for (;;) {
            if (hit(idx)) { found_at = idx; break; }
            if (pos >= right[idx]) { nxt = idx + floor_div2(hi - idx); lo = idx; }
            else if (pos <= left[idx]) { nxt = idx - floor_div2(idx - lo); hi = idx; if (idx == 1) nxt = 0; }
            if (idx == last) break;
            last = idx;
            idx = nxt;
        }
        if (found_at >= 0) { out[q] = (int32_t)found_at; continue; }
        bool found = false;
        for (int off = -3; off < 3; ++off) {
            const int64_t j = idx + off;
            if (j >= 0 && j < n - 1 && left[j] <= pos && pos <= right[j] && (free || qs == gene_strand[j])) { found = true; idx = j; }
        }
        out[q] = found ? (int32_t)idx : -1;
    }
    return SPL_OK;
}

// ---- .SpliSER.tsv rows (outputBedFile, SpliSER_v0_1_8.py:641-664) -----------------------------------------------
// One chromosome's rows appended to an open file: the same bytes spliser_amd/tsv.py formats -- "{0:.3f}" / "{0:.5f}" are
// correctly rounded decimal conversions in CPython and in glibc's printf alike, str(dict) / str(list) of ints are
// "{a: b, c: d}" / "[a, b]".  Strings come as one blob + offsets.
#include <algorithm>
#include <atomic>
#include <cinttypes>
#include <cmath>
#include <cstdio>
#include <chrono>
#include <string>
#include <thread>
#include <vector>

using splfmt::fmt_int;
using splfmt::fmt_fixed;

// (test hook: the formatter above against printf / Python over many values -- tests/test_tsv_native.py)
extern "C" int spl_fmt_fixed(double x, int digits, char *out64)
{
    if (!out64) return spl_set_error(SPL_ERR_ARG, "spl_fmt_fixed: null argument");
    out64[fmt_fixed(out64, x, digits)] = 0;
    return SPL_OK;
}

// (test hook: str(float) as Python prints it -- tests/test_combine_native.py)
extern "C" int spl_fmt_repr(double x, char *out64)
{
    if (!out64) return spl_set_error(SPL_ERR_ARG, "spl_fmt_repr: null argument");
    out64[splfmt::fmt_repr(out64, x)] = 0;
    return SPL_OK;
}

// The rows of several chromosomes appended to one file, in the order given: one pool of threads over the slices of all of
// them (a chromosome of 12 000 rows is a dozen slices; one after the other that was a millisecond each of mostly waiting).
extern "C" int spl_tsv_append_many(const char *path, int32_t n_chrom, const spl_tsv_rows *rows, int cryptic)
{
    if (!path || n_chrom < 0 || (n_chrom && !rows)) return spl_set_error(SPL_ERR_ARG, "spl_tsv_append_many: bad argument");
    for (int32_t c = 0; c < n_chrom; ++c) {
        const spl_tsv_rows &r = rows[c];
        if (!r.chrom || r.n_sites < 0) return spl_set_error(SPL_ERR_ARG, "spl_tsv_append_many: bad argument");
        if (r.n_sites && (!r.pos || !r.strand_blob || !r.strand_off || !r.gene_blob || !r.gene_off || !r.sse || !r.alpha || !r.beta1 || !r.beta2_simple ||
                          !r.part_off || !r.comp_off || (cryptic && (!r.beta2_cryptic || !r.beta2_weighted))))
            return spl_set_error(SPL_ERR_ARG, "spl_tsv_append_many: null array");
    }
    FILE *f = fopen(path, "ab");
    if (!f) return spl_set_error(SPL_ERR_IO, "cannot open %s for appending", path);
    const bool tm = getenv("SPL_TSV_TIMING") != nullptr;
    auto now = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t0 = now();
    // rows [a, b) of one chromosome as text.  Numbers are written by hand (fmt_int, fmt_fixed above: the digits printf would give,
    // at a tenth of printf's cost -- a row has seven to ten numbers and a table a few hundred thousand rows).
    auto format = [&](const spl_tsv_rows &r, int64_t a, int64_t b, std::string &out) {
        char num[96];
        const size_t chrom_len = strlen(r.chrom);
        out.clear();
        for (int64_t i = a; i < b; ++i) {
            out.append(r.chrom, chrom_len);
            out.push_back('\t');
            out.append(num, fmt_int(num, r.pos[i]));
            out.push_back('\t');
            out.append(r.strand_blob + r.strand_off[i], r.strand_off[i + 1] - r.strand_off[i]);
            out.push_back('\t');
            out.append(r.gene_blob + r.gene_off[i], r.gene_off[i + 1] - r.gene_off[i]);
            out.push_back('\t');
            out.append(num, fmt_fixed(num, r.sse[i], 3));
            out.push_back('\t');
            out.append(num, fmt_int(num, r.alpha[i]));
            out.push_back('\t');
            out.append(num, fmt_int(num, (int64_t)r.beta1[i]));
            out.push_back('\t');
            out.append(num, fmt_int(num, r.beta2_simple[i]));
            out.push_back('\t');
            if (cryptic) {
                out.append(num, fmt_int(num, r.beta2_cryptic[i]));
                out.push_back('\t');
                out.append(num, fmt_fixed(num, r.beta2_weighted[i], 5));
            } else {
                out.append("NA\tNA");
            }
            out.append("\t{");
            bool first_pair = true;
            for (uint32_t e = r.part_off[i]; e < r.part_off[i + 1]; ++e) {
                // (the column is str(dict) of PartnerCounts, :652: a row with two partner SITES at one position -- two edges -- has
                //  the position once)
                bool listed = false;
                for (uint32_t e2 = r.part_off[i]; e2 < e && !listed; ++e2) listed = r.part_pos[e2] == r.part_pos[e];
                if (listed) continue;
                if (!first_pair) out.append(", ");
                first_pair = false;
                out.append(num, fmt_int(num, r.part_pos[e]));
                out.append(": ");
                out.append(num, fmt_int(num, r.edge_cnt[e]));
            }
            out.append("}\t[");
            for (uint32_t e = r.comp_off[i]; e < r.comp_off[i + 1]; ++e) {
                if (e != r.comp_off[i]) out.append(", ");
                out.append(num, fmt_int(num, r.comp_pos[e]));
            }
            out.append("]\n");
        }
    };
    // slices of rows formatted on a few threads (snprintf of a quarter of a million rows is 80 ms on one), written in order
    const int64_t SLICE = 1024;
    struct Slice { int32_t chrom; int64_t a, b; };
    std::vector<Slice> slices;
    for (int32_t c = 0; c < n_chrom; ++c)
        for (int64_t a = 0; a < rows[c].n_sites; a += SLICE) slices.push_back(Slice{c, a, std::min<int64_t>(rows[c].n_sites, a + SLICE)});
    const size_t n_slices = slices.size();
    std::vector<std::string> text(n_slices);
    {
        std::atomic<size_t> next(0);
        auto work = [&]() {
            // (a thread formats into one buffer of its own that keeps its size from slice to slice, and leaves an exact-size copy:
            //  strings that grow while eight threads do the same went through mmap / munmap for every doubling, and eight threads
            //  were 1.8 times one)
            std::string scratch;
            scratch.reserve((size_t)SLICE * 160);
            for (;;) {
                const size_t k = next.fetch_add(1);
                if (k >= n_slices) break;
                format(rows[slices[k].chrom], slices[k].a, slices[k].b, scratch);
                text[k].assign(scratch.data(), scratch.size());
            }
        };
        int nt = (int)std::thread::hardware_concurrency();
        nt = nt > 16 ? 16 : (nt < 1 ? 1 : nt);
        nt = (int)std::min<size_t>((size_t)nt, n_slices);
        std::vector<std::thread> pool;
        for (int t = 1; t < nt; ++t) pool.emplace_back(work);
        work();
        for (auto &th : pool) th.join();
    }
    const double t1 = now();
    bool ok = true;
    for (size_t k = 0; k < n_slices && ok; ++k) ok = fwrite(text[k].data(), 1, text[k].size(), f) == text[k].size();
    if (fclose(f) != 0) ok = false;
    if (tm) {
        int64_t n_all = 0;
        for (int32_t c = 0; c < n_chrom; ++c) n_all += rows[c].n_sites;
        fprintf(stderr, "[spl_tsv_append] %d chromosome(s), %lld rows: formatted in %.4f s, written in %.4f s\n", (int)n_chrom, (long long)n_all, t1 - t0, now() - t1);
    }
    return ok ? SPL_OK : spl_set_error(SPL_ERR_IO, "write error on %s", path);
}

extern "C" int spl_tsv_append(const char *path, const char *chrom, int64_t n_sites, const int64_t *pos, const char *strand_blob,
                              const uint32_t *strand_off, const char *gene_blob, const uint32_t *gene_off, const double *sse,
                              const int64_t *alpha, const uint32_t *beta1, const int64_t *beta2_simple, int cryptic,
                              const int64_t *beta2_cryptic, const double *beta2_weighted, const uint32_t *part_off,
                              const int64_t *part_pos, const int64_t *edge_cnt, const uint32_t *comp_off, const int64_t *comp_pos)
{
    if (!path || !chrom || n_sites < 0) return spl_set_error(SPL_ERR_ARG, "spl_tsv_append: bad argument");
    spl_tsv_rows r;
    r.chrom = chrom; r.n_sites = n_sites; r.pos = pos; r.strand_blob = strand_blob; r.strand_off = strand_off; r.gene_blob = gene_blob;
    r.gene_off = gene_off; r.sse = sse; r.alpha = alpha; r.beta1 = beta1; r.beta2_simple = beta2_simple; r.beta2_cryptic = beta2_cryptic;
    r.beta2_weighted = beta2_weighted; r.part_off = part_off; r.part_pos = part_pos; r.edge_cnt = edge_cnt; r.comp_off = comp_off; r.comp_pos = comp_pos;
    return spl_tsv_append_many(path, 1, &r, cryptic);
}

// ---- Steps 0-1 text input: the BED12 junction file and the gene lines of the annotation, as columns ---------------------
// What findAlphaCounts reads from a line (SpliSER_v0_1_8.py:257-277) and what createGenes keeps of a gene line (:81-87, with
// HTSeq's conventions: start = column 4 - 1, end = column 5, name = value of the first attribute).  Files of this kind have
// 10^5 lines; line by line in Python that is a fifth of a second, here it is a few milliseconds.  Anything these parsers are
// not sure to read the way Python's str.split / int() would -- a number that is not a plain decimal integer, a strand
// column longer than one byte, a lone carriage return, bytes outside ASCII where they would matter -- makes the call fail with
// SPL_ERR_FORMAT, and the caller takes its line-by-line path, which then says what Python says.
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <map>

struct spl_textfile {
    std::vector<int32_t> chrom;
    std::vector<int64_t> a, b, c;      // BED: left, right, alpha;  GFF: left, right, -
    std::vector<uint8_t> strand;
    std::string name_blob;             // GFF: gene names
    std::vector<uint32_t> name_off;
    std::vector<std::string> chrom_names;
};

namespace {

struct Mapped {
    const char *p = nullptr;
    size_t n = 0;
    bool ok = false;
    explicit Mapped(const char *path)
    {
        const int fd = open(path, O_RDONLY);
        if (fd < 0) return;
        struct stat st;
        if (fstat(fd, &st) == 0) {
            n = (size_t)st.st_size;
            if (n == 0) ok = true;
            else {
                void *m = mmap(nullptr, n, PROT_READ, MAP_PRIVATE, fd, 0);
                if (m != MAP_FAILED) { p = (const char *)m; ok = true; }
            }
        }
        close(fd);
    }
    ~Mapped() { if (p) munmap((void *)p, n); }
};

inline bool is_space(char ch) { return ch == ' ' || ch == '\t' || ch == '\n' || ch == '\r' || ch == '\v' || ch == '\f'; }

// Python's int(text) for the texts that occur in these files: optional blanks, optional sign, decimal digits, optional blanks.
bool py_int(const char *s, const char *e, int64_t *out)
{
    while (s < e && is_space(*s)) ++s;
    while (e > s && is_space(e[-1])) --e;
    bool neg = false;
    if (s < e && (*s == '+' || *s == '-')) { neg = *s == '-'; ++s; }
    if (s >= e || e - s > 18) return false;
    int64_t v = 0;
    for (; s < e; ++s) {
        if (*s < '0' || *s > '9') return false;
        v = v * 10 + (*s - '0');
    }
    *out = neg ? -v : v;
    return true;
}

int32_t chrom_id(spl_textfile *t, std::map<std::string, int32_t> &ids, const char *s, const char *e)
{
    std::string key(s, e);
    auto it = ids.find(key);
    if (it != ids.end()) return it->second;
    const int32_t id = (int32_t)t->chrom_names.size();
    ids.emplace(key, id);
    t->chrom_names.push_back(key);
    return id;
}

} // namespace

extern "C" int spl_bed_open(const char *path, spl_textfile **out)
{
    if (!path || !out) return spl_set_error(SPL_ERR_ARG, "spl_bed_open: null argument");
    *out = nullptr;
    Mapped f(path);
    if (!f.ok) return spl_set_error(SPL_ERR_IO, "cannot read %s", path);
    spl_textfile *t = new spl_textfile();
    std::map<std::string, int32_t> ids;
    const char *p = f.p, *end = f.p + f.n;
    int64_t line_no = 0;
    while (p < end) {
        ++line_no;
        const char *nl = (const char *)memchr(p, '\n', (size_t)(end - p));
        const char *le = nl ? nl : end;             // the line without its newline
        if (le > p && le[-1] == '\r') --le;         // (text mode turns "\r\n" into "\n")
        if (memchr(p, '\r', (size_t)(le - p))) { delete t; return spl_set_error(SPL_ERR_FORMAT, "%s: carriage return inside line %lld", path, (long long)line_no); }
        const char *field[13];
        int n_field = 0;
        field[0] = p;
        for (const char *q = p; q < le && n_field < 12; ++q)
            if (*q == '\t') field[++n_field] = q + 1;
        // (n_field = tabs seen, capped at 12: twelve columns = eleven tabs)
        if (n_field == 11) {
            field[12] = le + 1;
            auto fe = [&](int k) { return field[k + 1] - 1; }; // end of column k
            int64_t start, stop, score, b0, b1;
            const char *blocks = field[10], *be = fe(10);
            const char *c1 = (const char *)memchr(blocks, ',', (size_t)(be - blocks));
            if (!c1) { delete t; return spl_set_error(SPL_ERR_FORMAT, "%s: line %lld has one block size", path, (long long)line_no); }
            const char *c2 = (const char *)memchr(c1 + 1, ',', (size_t)(be - c1 - 1));
            if (!py_int(field[1], fe(1), &start) || !py_int(field[2], fe(2), &stop) || !py_int(field[4], fe(4), &score) ||
                !py_int(blocks, c1, &b0) || !py_int(c1 + 1, c2 ? c2 : be, &b1)) {
                delete t;
                return spl_set_error(SPL_ERR_FORMAT, "%s: line %lld holds a number that is not a plain integer", path, (long long)line_no);
            }
            const size_t sl = (size_t)(fe(5) - field[5]);
            if (sl > 1 || (sl == 1 && (unsigned char)field[5][0] >= 128)) { delete t; return spl_set_error(SPL_ERR_FORMAT, "%s: line %lld has a strand column of more than one character", path, (long long)line_no); }
            t->chrom.push_back(chrom_id(t, ids, field[0], fe(0)));
            t->a.push_back(start + b0);   // :275
            t->b.push_back(stop - b1);    // :276
            t->c.push_back(score);        // :277
            t->strand.push_back(sl ? (uint8_t)field[5][0] : 0);
        }
        if (!nl) break;
        p = nl + 1;
    }
    *out = t;
    return SPL_OK;
}

extern "C" int spl_gff_open(const char *path, spl_textfile **out)
{
    if (!path || !out) return spl_set_error(SPL_ERR_ARG, "spl_gff_open: null argument");
    *out = nullptr;
    Mapped f(path);
    if (!f.ok) return spl_set_error(SPL_ERR_IO, "cannot read %s", path);
    spl_textfile *t = new spl_textfile();
    t->name_off.push_back(0);
    std::map<std::string, int32_t> ids;
    const char *p = f.p, *end = f.p + f.n;
    int64_t line_no = 0;
    auto fail = [&](const char *what) { delete t; return spl_set_error(SPL_ERR_FORMAT, "%s: line %lld %s", path, (long long)line_no, what); };
    while (p < end) {
        ++line_no;
        const char *nl = (const char *)memchr(p, '\n', (size_t)(end - p));
        const char *le = nl ? nl : end;
        if (le > p && le[-1] == '\r') --le;
        if (memchr(p, '\r', (size_t)(le - p))) return fail("holds a carriage return");
        bool blank = true;
        for (const char *q = p; q < le && blank; ++q) blank = is_space(*q);
        if (!(le == p || *p == '#' || blank)) {
            std::vector<const char *> field;
            field.push_back(p);
            for (const char *q = p; q < le; ++q) if (*q == '\t') field.push_back(q + 1);
            field.push_back(le + 1);
            const size_t n_col = field.size() - 1;
            auto fe = [&](size_t k) { return field[k + 1] - 1; };
            if (n_col >= 9 && fe(2) - field[2] == 4 && memcmp(field[2], "gene", 4) == 0) {
                int64_t c4, c5;
                if (!py_int(field[3], fe(3), &c4) || !py_int(field[4], fe(4), &c5)) return fail("holds a coordinate that is not a plain integer");
                const size_t sl = (size_t)(fe(6) - field[6]);
                if (sl != 1 || (unsigned char)field[6][0] >= 128) return fail("has a strand column that is not one character");
                // name = value of the first attribute (HTSeq), quotes stripped: sites.py _first_attribute
                const char *s = field[8], *e = fe(8);
                for (const char *q = s; q < e; ++q) if ((unsigned char)*q >= 128) return fail("has non-ASCII attributes");
                while (s < e && is_space(*s)) ++s;
                while (e > s && is_space(e[-1])) --e;
                const char *semi = (const char *)memchr(s, ';', (size_t)(e - s));
                if (semi) e = semi;
                while (s < e && is_space(*s)) ++s;
                while (e > s && is_space(e[-1])) --e;
                const char *eq = (const char *)memchr(s, '=', (size_t)(e - s));
                const char *sp = (const char *)memchr(s, ' ', (size_t)(e - s));
                if (eq) s = eq + 1;
                else if (sp) s = sp + 1;
                while (s < e && is_space(*s)) ++s;
                while (e > s && is_space(e[-1])) --e;
                while (s < e && *s == '"') ++s;
                while (e > s && e[-1] == '"') --e;
                t->chrom.push_back(chrom_id(t, ids, field[0], fe(0)));
                t->a.push_back(c4 - 1);
                t->b.push_back(c5);
                t->strand.push_back((uint8_t)field[6][0]);
                t->name_blob.append(s, e);
                t->name_off.push_back((uint32_t)t->name_blob.size());
            }
        }
        if (!nl) break;
        p = nl + 1;
    }
    *out = t;
    return SPL_OK;
}

extern "C" void spl_text_close(spl_textfile *t) { delete t; }
extern "C" int64_t spl_text_rows(const spl_textfile *t) { return t ? (int64_t)t->chrom.size() : 0; }
extern "C" int32_t spl_text_n_chrom(const spl_textfile *t) { return t ? (int32_t)t->chrom_names.size() : 0; }
extern "C" const char *spl_text_chrom_name(const spl_textfile *t, int32_t k) { return (t && k >= 0 && (size_t)k < t->chrom_names.size()) ? t->chrom_names[(size_t)k].c_str() : nullptr; }
extern "C" const int32_t *spl_text_chrom(const spl_textfile *t) { return t ? t->chrom.data() : nullptr; }
// which: 0 = left, 1 = right, 2 = alpha (BED only)
extern "C" const int64_t *spl_text_i64(const spl_textfile *t, int which) { return !t ? nullptr : (which == 0 ? t->a.data() : (which == 1 ? t->b.data() : t->c.data())); }
extern "C" const uint8_t *spl_text_strand(const spl_textfile *t) { return t ? t->strand.data() : nullptr; }
extern "C" const char *spl_text_names(const spl_textfile *t, const uint32_t **off_out) { if (!t) return nullptr; if (off_out) *off_out = t->name_off.data(); return t->name_blob.data(); }
