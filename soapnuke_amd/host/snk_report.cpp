// snk_report.cpp -- see snk_report.h.  Byte-exact restatement of the reference's report
// semantics, including its quirks: fp32 percentages, truncated integer
// means, int32 quantile positions, per-thread running-max merges, `tlq` printed twice in the
// zero-total branches, fq1's read_length driving both mates' row counts.
#include "snk_report.h"

#include <stdio.h>
#include <string.h>

#include <fstream>
#include <iomanip>
#include <sstream>
#include <string>
#include <vector>

namespace {

const int ROWS = SNK_READ_MAX_LEN;   // READ_MAX_LEN, src/global_variable.h:9

struct FileStat {                    // C_fastq_file_stat, src/global_variable.h:88-134
    uint64_t read_max_length = 0, read_length = 0, reads = 0, bases = 0;
    uint64_t acgtn[5] = {0, 0, 0, 0, 0}, q20 = 0, q30 = 0;
    std::vector<uint64_t> bs, qs, ts;
    int nq = 0;
    void init(int nq_) {
        nq = nq_;
        bs.assign((size_t)ROWS * 5, 0);
        qs.assign((size_t)ROWS * nq, 0);
        ts.assign(SNK_TS_N, 0);
    }
    uint64_t &B(uint64_t i, int j) { return bs[i * 5 + j]; }
    uint64_t &Q(uint64_t i, int j) { return qs[i * nq + j]; }
};

// one virtual thread's file block
struct View {
    const uint64_t *f = nullptr;
    int lcap = 0, nq = 0;
    uint64_t read_length = 0;
    uint64_t gs(int k) const { return f[k]; }
    uint64_t B(uint64_t i, int j) const { return i < (uint64_t)lcap ? f[SNK_GS_N + i * 5 + j] : 0; }
    uint64_t Q(uint64_t i, int j) const { return (i < (uint64_t)lcap && j < nq) ? f[SNK_GS_N + (uint64_t)lcap * 5 + i * nq + j] : 0; }
    uint64_t T(int k) const { return f[SNK_GS_N + (uint64_t)lcap * 5 + (uint64_t)lcap * nq + k]; }
};

void add_gs(FileStat &g, const View &t) {
    g.reads += t.gs(SNK_GS_READS);
    g.bases += t.gs(SNK_GS_BASES);
    for (int j = 0; j < 5; ++j) g.acgtn[j] += t.gs(SNK_GS_A + j);
    g.q20 += t.gs(SNK_GS_Q20);
    g.q30 += t.gs(SNK_GS_Q30);
}

int thread_max_qual(const View &t, uint64_t rows, int max_base_quality) {
    int mq = 0;
    for (uint64_t i = 0; i != rows; ++i)
        for (int j = 1; j <= max_base_quality; ++j)
            if (t.Q(i, j) > 0 && j > mq) mq = j;
    return mq;
}

void add_ts(FileStat &g, const View &t, uint64_t lo, uint64_t hi_excl) {
    static const int base[5] = {SNK_TS_HT, SNK_TS_HLQ, SNK_TS_TT, SNK_TS_TLQ, SNK_TS_TA};
    for (uint64_t i = lo; i < hi_excl && i < 1000; ++i)
        for (int a = 0; a < 5; ++a) g.ts[base[a] + i] += t.T(base[a] + (int)i);
}

struct Quart { float mean, median, lower, upper, first10, last10; };

// cal_quar_from_array, src/gc.cpp:68-119 (int32 counters and positions on purpose)
// *set (optional): bit k = field k was assigned.  A wrapped, negative position matches no bin: the reference then
// returns that field of its constructor-less quartile_result uninitialised (whatever its stack held); 0 here.
Quart quartiles_of(const uint64_t *data, int nq, int len, int *set = nullptr) {
    int mask = 1;
    Quart r = {0, 0, 0, 0, 0, 0};
    unsigned long long total = 0;
    int32_t data_num = 0;
    auto val = [&](int i) -> uint64_t { return i < nq ? data[i] : 0; };      // (one slot past the row reads 0 there)
    for (int i = 0; i <= len; ++i) {
        total += (unsigned long long)i * val(i);
        data_num = (int32_t)((uint32_t)data_num + (uint32_t)val(i));
    }
    r.mean = data_num == 0 ? 0 : (float)(total / (unsigned long long)(long long)data_num);
    auto mul = [](int32_t a, int32_t b) { return (int32_t)((uint32_t)a * (uint32_t)b); };
    const int32_t lower_pos = data_num / 4, upper_pos = mul(data_num, 3) / 4, first10_pos = data_num / 10,
                  last10_pos = mul(data_num, 9) / 10, median_pos = data_num / 2;
    int32_t last = 0, cur = 0;
    for (int i = 0; i <= len; ++i) {
        cur = (int32_t)((uint32_t)cur + (uint32_t)val(i));
        if (lower_pos >= last && lower_pos <= cur) { r.lower = (float)i; mask |= 4; }
        if (upper_pos >= last && upper_pos <= cur) { r.upper = (float)i; mask |= 8; }
        if (first10_pos >= last && first10_pos <= cur) { r.first10 = (float)i; mask |= 16; }
        if (last10_pos >= last && last10_pos <= cur) { r.last10 = (float)i; mask |= 32; }
        if (median_pos >= last && median_pos <= cur) { r.median = (float)i; mask |= 2; }
        last = cur;
    }
    if (set) *set = mask;
    return r;
}

Quart quartiles(FileStat &g, uint64_t row, int len) { return quartiles_of(&g.Q(row, 0), g.nq, len); }

std::string pct2(float v) {            // sprintf("%.2f", float)
    char b[64];
    snprintf(b, sizeof b, "%.2f", (double)v);
    return b;
}

struct Gv {
    FileStat f[4];                     // raw1 raw2 clean1 clean2
    uint64_t fs[SNK_FS_N];
};

void quar_cols(std::ostream &o, const Quart &q) {
    o << std::setiosflags(std::ios::fixed) << std::setprecision(2) << q.mean << "\t";
    o << std::setprecision(0) << q.median << "\t" << q.lower << "\t" << q.upper << "\t" << q.first10 << "\t" << q.last10
      << std::endl;
}

void trim_cols(std::ostream &o, FileStat &g, uint64_t i, uint64_t head_total, uint64_t tail_total, bool last_block) {
    auto hlq = g.ts[SNK_TS_HLQ + i], ht = g.ts[SNK_TS_HT + i], ta = g.ts[SNK_TS_TA + i], tlq = g.ts[SNK_TS_TLQ + i],
         tt = g.ts[SNK_TS_TT + i];
    auto pc = [&](uint64_t x, uint64_t tot) { o << x << "\t" << std::setiosflags(std::ios::fixed) << std::setprecision(2) << 100 * (float)x / tot; };
    if (head_total > 0) { pc(hlq, head_total); o << "%\t"; pc(ht, head_total); o << "%\t"; }
    else { o << hlq << "\t0.00%\t" << ht << "\t0.00%\t"; }
    if (tail_total > 0) {
        pc(ta, tail_total); o << "%\t"; pc(tlq, tail_total); o << "%\t"; pc(tt, tail_total);
        if (last_block) o << "%" << std::endl; else o << "%\t";
    } else {                           // prints tlq twice, never tt (src/peprocess.cpp:641-643)
        o << ta << "\t0.00%\t" << tlq << "\t0.00%\t" << tlq;
        if (last_block) o << "\t0.00%" << std::endl; else o << "\t0.00%\t";
    }
}

const char *ITEMS_PE[] = {"Reads are duplicate", "Reads limited to output number", "Reads with filtered tile",
                          "Reads with filtered fov", "Reads too short", "Reads too long",
                          "Reads with global contam sequence", "Reads with contam sequence", "Reads with n rate exceed",
                          "Reads with highA", "Reads with polyX", "Reads with low quality", "Reads with low mean quality",
                          "Reads with small insert size", "Reads with adapter"};
const char *ITEMS_SE[] = {"Reads are duplicate", "Reads limited to output number", "Reads with filtered tile",
                          "Reads with filtered fov", "Reads too short", "Reads too long", "Reads with contam sequence",
                          "Reads with n rate exceed", "Reads with highA", "Reads with polyX", "Reads with low quality",
                          "Reads with low mean quality", "Reads with adapter", "Reads with global contam sequence"};

// item -> base index into fs (family of 4) or single counter; -1 = never counted
int item_index(const std::string &s, bool &single) {
    single = false;
    if (s == "Reads are duplicate") { single = true; return SNK_FS_DUP; }
    if (s == "Reads with filtered tile") { single = true; return SNK_FS_TILE; }
    if (s == "Reads with filtered fov") { single = true; return SNK_FS_FOV; }
    if (s == "Reads with small insert size") { single = true; return SNK_FS_OVERLAP; }
    if (s == "Reads too short") return SNK_FS_SHORT;
    if (s == "Reads too long") return SNK_FS_LONG;
    if (s == "Reads with global contam sequence") return SNK_FS_GCONTAM;
    if (s == "Reads with contam sequence") return SNK_FS_CONTAM;
    if (s == "Reads with n rate exceed") return SNK_FS_NRATE;
    if (s == "Reads with highA") return SNK_FS_HIGHA;
    if (s == "Reads with polyX") return SNK_FS_POLYX;
    if (s == "Reads with low quality") return SNK_FS_LOWQUAL;
    if (s == "Reads with low mean quality") return SNK_FS_MEANQ;
    if (s == "Reads with adapter") return SNK_FS_ADAPTER;
    return -1;
}

void general_line(std::ostream &o, const char *name, uint64_t v[4], std::string p[4], int nfiles, bool se_tab) {
    o << name << "\t" << std::setprecision(15);
    for (int k = 0; k < nfiles; ++k) {
        o << v[k] << " (" << p[k] << "%)";
        if (k + 1 < nfiles) o << "\t";
    }
    if (se_tab) o << "\t";
    o << std::endl;
}

int write_all(const snk_params *P, Gv &gv, const std::string &dir, std::string &err) {
    const bool pe = P->paired != 0;
    const int mbq = P->max_base_quality;
    FileStat &raw1 = gv.f[0], &raw2 = gv.f[1], &clean1 = gv.f[2], &clean2 = gv.f[3];
    auto open = [&](const std::string &name, std::ofstream &f) {
        f.open((dir + "/" + name).c_str());
        if (!f) { err = "Error:cannot open such file," + dir + "/" + name; return false; }
        return true;
    };
    // ---- Statistics_of_Filtered_Reads.txt (src/peprocess.cpp:225-322, src/seprocess.cpp:135-181)
    uint64_t total_filter = 0;
    {
        std::ofstream o;
        if (!open("Statistics_of_Filtered_Reads.txt", o)) return -1;
        // the map sums every reason, duplicates included; overlap only exists for PE
        const int fams[] = {SNK_FS_SHORT, SNK_FS_LONG, SNK_FS_GCONTAM, SNK_FS_CONTAM, SNK_FS_NRATE, SNK_FS_HIGHA,
                            SNK_FS_POLYX, SNK_FS_LOWQUAL, SNK_FS_MEANQ, SNK_FS_ADAPTER};
        total_filter = gv.fs[SNK_FS_DUP] + gv.fs[SNK_FS_TILE] + gv.fs[SNK_FS_FOV] + (pe ? gv.fs[SNK_FS_OVERLAP] : 0);
        for (int f : fams) total_filter += gv.fs[f];
        if (pe) o << "Item\t\t\t\tTotal\tPercentage\tfastq1\tfastq2\toverlap" << std::endl;
        else o << "Item\tTotal\tPercentage" << std::endl;
        o << std::setiosflags(std::ios::fixed);
        if (pe) o << "Total filtered read pair number\t" << total_filter << "\t100.00%\t\t" << total_filter << "\t" << total_filter << "\t" << total_filter << std::endl;
        else o << "Total filtered read pair number\t" << total_filter << "\t100.00%" << std::endl;
        const char **items = pe ? ITEMS_PE : ITEMS_SE;
        const int nitems = pe ? 15 : 14;
        for (int k = 0; k < nitems; ++k) {
            bool single;
            const int ix = item_index(items[k], single);
            if (ix < 0 || (!pe && ix == SNK_FS_OVERLAP)) continue;
            const uint64_t n = gv.fs[ix];
            if (n == 0) continue;
            o << items[k] << "\t" << n << "\t" << std::setprecision(2) << 100 * (float)n / total_filter;
            if (pe) {
                const uint64_t a = single ? n : gv.fs[ix + 1], b = single ? n : gv.fs[ix + 2], c = single ? n : gv.fs[ix + 3];
                o << "%\t" << a << "\t" << b << "\t" << c << std::endl;
            } else {
                o << "%" << std::endl;
            }
        }
    }
    // ---- Basic_Statistics_of_Sequencing_Quality.txt (src/peprocess.cpp:324-413, src/seprocess.cpp:182-235)
    {
        std::ofstream o;
        if (!open("Basic_Statistics_of_Sequencing_Quality.txt", o)) return -1;
        const int order[4] = {0, 2, 1, 3};            // raw1 clean1 raw2 clean2
        const int nfiles = pe ? 4 : 2;
        float rl[4] = {0, 0, 0, 0};
        std::string ratio[4][7], fr[2];
        for (int k = 0; k < 4; ++k) {
            FileStat &g = gv.f[k];
            if (g.reads == 0) continue;
            // PE: 1.0*base/reads (double) narrowed to float; SE: (float)base/reads
            rl[k] = pe ? (float)(1.0 * g.bases / g.reads) : (float)g.bases / g.reads;
            if (k < 2) fr[k] = pct2(100 * (float)total_filter / g.reads);
            for (int j = 0; j < 5; ++j) ratio[k][j] = pct2(100 * (float)g.acgtn[j] / g.bases);
            ratio[k][5] = pct2(100 * (float)g.q20 / g.bases);
            ratio[k][6] = pct2(100 * (float)g.q30 / g.bases);
        }
        if (pe) o << "Item\traw reads(fq1)\tclean reads(fq1)\traw reads(fq2)\tclean reads(fq2)" << std::endl;
        else o << "Item\traw reads(fq1)\tclean reads(fq1)" << std::endl;
        o << std::setiosflags(std::ios::fixed) << std::setprecision(1) << "Read length";
        for (int k = 0; k < nfiles; ++k) o << "\t" << rl[order[k]];
        o << std::endl;
        o << "Total number of reads\t" << std::setprecision(15);
        for (int k = 0; k < nfiles; ++k) o << gv.f[order[k]].reads << " (100.00%)" << (k + 1 < nfiles ? "\t" : "");
        o << std::endl;
        const uint64_t fbases = total_filter * raw1.read_length;   // fq1's length for both mates (Q9)
        if (pe) o << "Number of filtered reads\t" << total_filter << " (" << fr[0] << "%)\t-\t" << total_filter << " (" << fr[1] << "%)\t-" << std::endl;
        else o << "Number of filtered reads\t" << total_filter << " (" << fr[0] << "%)\t-" << std::endl;
        o << "Total number of bases\t" << std::setprecision(15);
        for (int k = 0; k < nfiles; ++k) o << gv.f[order[k]].bases << " (100.00%)" << (k + 1 < nfiles ? "\t" : "");
        o << std::endl;
        if (pe) o << "Number of filtered bases\t" << std::setprecision(15) << fbases << " (" << fr[0] << "%)\t-\t" << fbases << " (" << fr[1] << "%)\t-" << std::endl;
        else o << "Number of filtered bases\t" << std::setprecision(15) << fbases << " (" << fr[0] << "%)\t-" << std::endl;
        const char *bn[5] = {"Number of base A", "Number of base C", "Number of base G", "Number of base T", "Number of base N"};
        for (int j = 0; j < 7; ++j) {
            uint64_t v[4];
            std::string p[4];
            for (int k = 0; k < nfiles; ++k) {
                FileStat &g = gv.f[order[k]];
                v[k] = j < 5 ? g.acgtn[j] : (j == 5 ? g.q20 : g.q30);
                p[k] = ratio[order[k]][j];
            }
            general_line(o, j < 5 ? bn[j] : (j == 5 ? "Q20 number" : "Q30 number"), v, p, nfiles, !pe && j < 5);
        }
    }
    // ---- Base_distributions_by_read_position_{1,2}.txt (src/peprocess.cpp:414-466)
    for (int m = 0; m < (pe ? 2 : 1); ++m) {
        std::ofstream o;
        if (!open(std::string("Base_distributions_by_read_position_") + (m ? "2" : "1") + ".txt", o)) return -1;
        FileStat &r = gv.f[m], &c = gv.f[2 + m];
        o << "Pos\tA\tC\tG\tT\tN\tclean A\tclean C\tclean G\tclean T\tclean N" << std::endl;
        for (uint64_t i = 0; i < raw1.read_length && i < (uint64_t)ROWS; ++i) {   // fq1's length for both (Q1)
            o << i + 1 << "\t";
            float rt = 0, ct = 0;
            for (int j = 0; j < 5; ++j) { rt += r.B(i, j); ct += c.B(i, j); }
            for (int j = 0; j < 5; ++j)
                o << std::setiosflags(std::ios::fixed) << std::setprecision(2) << 100 * (float)r.B(i, j) / rt << "%\t";
            for (int j = 0; j < 5; ++j) {
                o << std::setiosflags(std::ios::fixed) << std::setprecision(2) << 100 * (float)c.B(i, j) / ct << "%";
                if (j != 4) o << "\t"; else o << std::endl;
            }
        }
    }
    // ---- quality distribution + Q20/Q30 files (src/peprocess.cpp:468-602, src/seprocess.cpp:270-361)
    int max_qual = 0;
    for (uint64_t i = 0; i < raw1.read_length && i < (uint64_t)ROWS; ++i)
        for (int j = 1; j <= mbq; ++j)
            if (raw1.Q(i, j) > 0 && j > max_qual) max_qual = j;
    if (pe) {
        const uint64_t rml = raw1.read_max_length > raw2.read_max_length ? raw1.read_max_length : raw2.read_max_length;
        std::ofstream q[2], d[2];
        for (int m = 0; m < 2; ++m) {
            if (!open(std::string("Base_quality_value_distribution_by_read_position_") + (m ? "2" : "1") + ".txt", q[m])) return -1;
            if (!open(std::string("Distribution_of_Q20_Q30_bases_by_read_position_") + (m ? "2" : "1") + ".txt", d[m])) return -1;
        }
        std::vector<float> q20[4], q30[4];
        for (int k = 0; k < 4; ++k) { q20[k].assign(rml + 1, 0.f); q30[k].assign(rml + 1, 0.f); }
        auto header = [&](std::ostream &o) {
            o << "Pos\t";
            for (int i = 0; i <= max_qual; ++i) o << "Q" << i << "\t";
            o << "Mean\tMedian\tLower quartile\tUpper quartile\t10th percentile\t90th percentile" << std::endl;
        };
        auto block = [&](int kbase) {   // kbase 0: raw, 2: clean
            for (uint64_t i = 0; i != rml && i < (uint64_t)ROWS; ++i) {
                for (int m = 0; m < 2; ++m) {
                    FileStat &g = gv.f[kbase + m];
                    std::ostream &o = q[m];
                    o << i + 1 << "\t";
                    uint64_t n20 = 0, n30 = 0, tot = 0;
                    for (int j = 0; j <= max_qual; ++j) {
                        const uint64_t x = g.Q(i, j);
                        if (j >= 20) n20 += x;
                        if (j >= 30) n30 += x;
                        tot += x;
                        o << std::setiosflags(std::ios::fixed) << std::setprecision(0) << x << "\t";
                    }
                    q20[kbase + m][i] = (float)n20 / tot;
                    q30[kbase + m][i] = (float)n30 / tot;
                    quar_cols(o, quartiles(g, i, max_qual));       // PE passes len = max_qual (Q2)
                }
                if (kbase == 2)
                    for (int m = 0; m < 2; ++m)
                        d[m] << i + 1 << std::setiosflags(std::ios::fixed) << std::setprecision(2) << "\t" << 100 * q20[m][i]
                             << "%\t" << 100 * q30[m][i] << "%\t" << 100 * q20[2 + m][i] << "%\t" << 100 * q30[2 + m][i] << "%" << std::endl;
            }
        };
        for (int m = 0; m < 2; ++m) { q[m] << "#raw fastq" << (m + 1) << " quality distribution" << std::endl; header(q[m]); }
        block(0);
        for (int m = 0; m < 2; ++m) {
            q[m] << "#clean fastq" << (m + 1) << " quality distribution" << std::endl;
            header(q[m]);
            d[m] << "Position in reads\tPercentage of Q20+ bases\tPercentage of Q30+ bases\tPercentage of Clean Q20+\tPercentage of Clean Q30+" << std::endl;
        }
        block(2);
    } else {
        std::ofstream q, d;
        if (!open("Base_quality_value_distribution_by_read_position_1.txt", q)) return -1;
        if (!open("Distribution_of_Q20_Q30_bases_by_read_position_1.txt", d)) return -1;
        const uint64_t cap = raw1.read_max_length > clean1.read_max_length ? raw1.read_max_length : clean1.read_max_length;
        std::vector<float> rq20(cap + 1, 0.f), rq30(cap + 1, 0.f);
        auto header = [&](std::ostream &o) {
            o << "Pos\t";
            for (int i = 0; i <= max_qual; ++i) o << "Q" << i << "\t";
            o << "Mean\tMedian\tLower quartile\tUpper quartile\t10th percentile\t90th percentile" << std::endl;
        };
        q << "#raw fastq1 quality distribution" << std::endl;
        header(q);
        for (uint64_t i = 0; i != raw1.read_length && i < (uint64_t)ROWS; ++i) {
            q << i + 1 << "\t";
            uint64_t n20 = 0, n30 = 0, tot = 0;
            for (int j = 0; j <= max_qual; ++j) {
                const uint64_t x = raw1.Q(i, j);
                if (j >= 20) n20 += x;
                if (j >= 30) n30 += x;
                tot += x;
                q << std::setiosflags(std::ios::fixed) << std::setprecision(0) << x << "\t";
            }
            rq20[i] = (float)n20 / tot;
            rq30[i] = (float)n30 / tot;
            quar_cols(q, quartiles(raw1, i, max_qual + 1));        // SE passes max_qual+1 (Q2)
        }
        q << "#clean fastq1 quality distribution" << std::endl;
        header(q);
        d << "Position in reads\tPercentage of Q20+ bases\tPercentage of Q30+ bases\tPercentage of Clean Q20+\tPercentage of Clean Q30+" << std::endl;
        for (uint64_t i = 0; i != clean1.read_max_length && i < (uint64_t)ROWS; ++i) {
            q << i + 1 << "\t";
            uint64_t n20 = 0, n30 = 0, tot = 0;
            for (int j = 0; j <= max_qual; ++j) {
                const uint64_t x = clean1.Q(i, j);
                if (j >= 20) n20 += x;
                if (j >= 30) n30 += x;
                tot += x;
                q << std::setiosflags(std::ios::fixed) << std::setprecision(0) << x << "\t";
            }
            const float c20 = (float)n20 / tot, c30 = (float)n30 / tot;
            quar_cols(q, quartiles(clean1, i, max_qual + 1));
            d << i + 1 << std::setiosflags(std::ios::fixed) << std::setprecision(4) << "\t" << rq20[i] << "\t" << rq30[i] << "\t"
              << c20 << "\t" << c30 << std::endl;
        }
    }
    // ---- Statistics_of_Trimming_Position_of_Reads_{1,2}.txt (src/peprocess.cpp:603-715)
    {
        std::ofstream o[2];
        uint64_t head[4] = {0, 0, 0, 0}, tail[4] = {0, 0, 0, 0};
        for (int m = 0; m < (pe ? 2 : 1); ++m) {
            if (!open(std::string("Statistics_of_Trimming_Position_of_Reads_") + (m ? "2" : "1") + ".txt", o[m])) return -1;
            o[m] << "Pos\tHeadLowQual\tHeadFixLen\tTailAdapter\tTailLowQual\tTailFixLen\tCleanHeadLowQual\tCleanHeadFixLen\tCleanTailAdapter\tCleanTailLowQual\tCleanTailFixLen" << std::endl;
        }
        for (uint64_t i = 0; i < raw1.read_length && i < 1000; ++i)
            for (int k = 0; k < 4; ++k) {
                head[k] += gv.f[k].ts[SNK_TS_HT + i] + gv.f[k].ts[SNK_TS_HLQ + i];
                tail[k] += gv.f[k].ts[SNK_TS_TA + i] + gv.f[k].ts[SNK_TS_TLQ + i] + gv.f[k].ts[SNK_TS_TT + i];
            }
        for (uint64_t i = 1; i <= raw1.read_length && i < 1000; ++i)
            for (int m = 0; m < (pe ? 2 : 1); ++m) {
                o[m] << i << "\t";
                trim_cols(o[m], gv.f[m], i, head[m], tail[m], false);
                trim_cols(o[m], gv.f[2 + m], i, head[2 + m], tail[2 + m], true);
            }
    }
    return 0;
}

}  // namespace

// test hook: the quartile columns of one quality histogram row (mean, median, lower, upper, first10, last10)
extern "C" int snk_report_quartiles(const uint64_t *data, int nq, int len, float out[6]) {
    int set = 0;
    const Quart q = quartiles_of(data, nq, len, &set);
    out[0] = q.mean; out[1] = q.median; out[2] = q.lower; out[3] = q.upper; out[4] = q.first10; out[5] = q.last10;
    return set;
}

extern "C" int64_t snk_vthread_block(int threads, int patch_size) {
    if (threads < 1) threads = 1;
    const int patch = 160 / threads;                          // src/peprocess.cpp:81
    const int64_t ps = patch_size > 0 ? patch_size : (int64_t)threads * 20000 / 8;   // src/process_argv.cpp:541-544
    const int64_t b = ps * (patch > 0 ? patch : 1);
    return b > 0 ? b : 1;
}

extern "C" int snk_write_reports(const snk_params *P, int T, const uint64_t *const *sums, const uint64_t *const *maxs,
                                 const char *out_dir, char *errbuf, int cap) {
    const int lcap = P->max_read_len, nq = P->max_base_quality + 1, mbq = P->max_base_quality;
    const bool pe = P->paired != 0;
    Gv gv;
    memset(gv.fs, 0, sizeof gv.fs);
    for (auto &f : gv.f) f.init(nq);
    FileStat &raw1 = gv.f[0], &raw2 = gv.f[1], &clean1 = gv.f[2], &clean2 = gv.f[3];
    // merge_stat(): threads folded in order, src/peprocess.cpp:1994-2005 / seprocess.cpp
    for (int t = 0; t < T; ++t) {
        View v[4];
        for (int k = 0; k < 4; ++k) {
            v[k].f = sums[t] + snk_file_off(lcap, nq, k);
            v[k].lcap = lcap;
            v[k].nq = nq;
            v[k].read_length = maxs[t][k] & 0xFFFF;           // length of the last read that thread saw
        }
        // ---------- "raw" (src/peprocess.cpp:734-875, src/seprocess.cpp:438-506)
        for (int m = 0; m < (pe ? 2 : 1); ++m) {
            FileStat &g = gv.f[m];
            if (g.read_length == 0) g.read_length = v[m].read_length;
            if (g.read_max_length < v[m].read_length) g.read_max_length = v[m].read_length;
            add_gs(g, v[m]);
        }
        {
            const uint64_t rows = raw1.read_max_length;       // fq1's running max drives both mates
            for (uint64_t i = 0; i != rows && i < (uint64_t)ROWS; ++i)
                for (int j = 0; j < 5; ++j) {
                    raw1.B(i, j) += v[0].B(i, j);
                    if (pe) raw2.B(i, j) += v[1].B(i, j);
                }
            if (pe) { add_ts(raw1, v[0], 0, rows); add_ts(raw2, v[1], 0, rows); }
            else add_ts(raw1, v[0], 1, rows + 1);             // SE: 1..read_max_length inclusive
            const int mq = thread_max_qual(v[0], rows, mbq);
            for (uint64_t i = 0; i != rows && i < (uint64_t)ROWS; ++i)
                for (int j = 0; j <= mq; ++j) {
                    raw1.Q(i, j) += v[0].Q(i, j);
                    if (pe) raw2.Q(i, j) += v[1].Q(i, j);
                }
            for (int k = 0; k < SNK_FS_N; ++k) gv.fs[k] += sums[t][k];
        }
        // ---------- "clean" (src/peprocess.cpp:952-1069, src/seprocess.cpp:565-624)
        for (int m = 0; m < (pe ? 2 : 1); ++m) {
            FileStat &g = gv.f[2 + m];
            const View &w = v[2 + m];
            g.reads += w.gs(SNK_GS_READS);
            g.bases += w.gs(SNK_GS_BASES);
            if (g.reads == 0) g.read_length = w.read_length;
            else g.read_length = g.bases / g.reads;
            if (m == 0) { if (g.read_max_length < w.read_length) g.read_max_length = w.read_length; }
            else { if (g.read_max_length < g.read_length) g.read_max_length = g.read_length; }   // fq2: running MEAN (Q3)
            for (int j = 0; j < 5; ++j) g.acgtn[j] += w.gs(SNK_GS_A + j);
            g.q20 += w.gs(SNK_GS_Q20);
            g.q30 += w.gs(SNK_GS_Q30);
        }
        for (int m = 0; m < (pe ? 2 : 1); ++m) {
            FileStat &g = gv.f[2 + m];
            const View &w = v[2 + m];
            const uint64_t rows = g.read_max_length;
            for (uint64_t i = 0; i != rows && i < (uint64_t)ROWS; ++i)
                for (int j = 0; j < 5; ++j) g.B(i, j) += w.B(i, j);
            add_ts(g, w, 0, rows);
            const int mq = thread_max_qual(w, rows, mbq);
            for (uint64_t i = 0; i != rows && i < (uint64_t)ROWS; ++i)
                for (int j = 0; j <= mq; ++j) g.Q(i, j) += w.Q(i, j);
        }
    }
    (void)clean1; (void)clean2;
    std::string err;
    const int rc = write_all(P, gv, out_dir, err);
    if (rc && errbuf && cap > 0) snprintf(errbuf, cap, "%s", err.c_str());
    return rc;
}

// peStreaming_stat / seStreaming_stat.  As printed by the reference: the rows of the "raw" quality block of fq1 show
// the CLEAN counters, fq1's quality rows have 40 columns and fq2's 41, both closed by a literal 0.
extern "C" void snk_streaming_stat_text(const snk_params *P, const uint64_t *sum, const uint64_t *mx, void *out_string) {
    std::string &o = *static_cast<std::string *>(out_string);
    const int lcap = P->max_read_len, nq = P->max_base_quality + 1;
    const bool pe = P->paired != 0;
    View v[4];
    for (int k = 0; k < 4; ++k) {
        v[k].f = sum + snk_file_off(lcap, nq, k);
        v[k].lcap = lcap;
        v[k].nq = nq;
        v[k].read_length = mx[k] & 0xFFFF;
    }
    auto num = [&](uint64_t x) { o += std::to_string(x); };
    const uint64_t *fs = sum;
    o += "#Total_statistical_information\n";
    const int total = (int)(fs[SNK_FS_ADAPTER] + fs[SNK_FS_CONTAM] + fs[SNK_FS_LOWQUAL] + fs[SNK_FS_MEANQ] + fs[SNK_FS_NRATE] + fs[SNK_FS_OVERLAP] +
                            fs[SNK_FS_HIGHA] + fs[SNK_FS_POLYX]);
    o += std::to_string(total);
    for (int k : {SNK_FS_ADAPTER, SNK_FS_CONTAM, SNK_FS_LOWQUAL, SNK_FS_MEANQ, SNK_FS_NRATE, SNK_FS_OVERLAP, SNK_FS_HIGHA, SNK_FS_POLYX}) {
        o += ' ';
        o += std::to_string((int)fs[k]);
    }
    o += '\n';
    for (int m = 0; m < (pe ? 2 : 1); ++m) {
        const View &raw = v[m], &clean = v[2 + m];
        o += m == 0 ? "#Fq1_statistical_information\n" : "#Fq2_statistical_information\n";
        num(raw.read_length); o += ' '; num(clean.read_length);
        for (int k : {SNK_GS_READS, SNK_GS_BASES, SNK_GS_A, SNK_GS_C, SNK_GS_G, SNK_GS_T, SNK_GS_N_, SNK_GS_Q20, SNK_GS_Q30}) {
            o += ' '; num(raw.gs(k)); o += ' '; num(clean.gs(k));
        }
        o += '\n';
        o += "#Base_distributions_by_read_position\n";
        for (const View *w : {&raw, &clean})
            for (uint64_t i = 0; i != w->read_length; ++i) {
                for (int j = 0; j < 4; ++j) { num(w->B(i, j)); o += ' '; }
                num(w->B(i, 4));
                o += '\n';
            }
        o += "#Raw_Base_quality_value_distribution_by_read_position\n";
        const int cols = m == 0 ? 40 : 41;
        const View &first = m == 0 ? clean : raw;              // fq1: the reference reads clean1 here, over raw1's length
        for (uint64_t i = 0; i != raw.read_length; ++i) {
            for (int j = 0; j < cols; ++j) { num(first.Q(i, j)); o += ' '; }
            o += "0\n";
        }
        for (uint64_t i = 0; i != clean.read_length; ++i) {
            for (int j = 0; j < cols; ++j) { num(clean.Q(i, j)); o += ' '; }
            o += "0\n";
        }
    }
}
