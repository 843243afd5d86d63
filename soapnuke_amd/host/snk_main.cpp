// snk_main.cpp -- `SOAPnuke filter` command line on top of the C ABI (include/snk_filter.h).
//
// Host counterpart of main() + peProcess::process()/seProcess::process() of the reference
// (src/main.cpp:17-68, src/peprocess.cpp:3051-3201, src/seprocess.cpp:2087): same options,
// config-file keys, output files and report bytes for the `filter` module -- but one pass over
// the input, FASTQ parsed once into pinned structure-of-arrays batches, the per-read hot path on
// the GPU, clean reads written in input order by the host.  No temp files, no `cat`.
//
// Round-1 scope: correctness of the drop-in surface; the host pipeline is
// single-threaded and synchronous (its overlap/parallel inflate is row N2).
#include <getopt.h>
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <time.h>
#include <zlib.h>

#include <fstream>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>

#include "../../include/snk_filter.h"
#include "snk_report.h"
#include "../../include/snk_rmdup.h"

using std::cerr;
using std::cout;
using std::endl;
using std::string;

namespace {

struct Options {
    string fq1, fq2, clean1, clean2, out_dir, log = "log";
    std::vector<string> ada1, ada2;
    snk_params p;
    string trim, trim_bad_head, trim_bad_tail, out_file_type = "fastq";
    int threads = 6, patch_size = 0, batch_pairs = 1 << 18, device = 0;
    bool in_gz = true, out_gz = true;
};

[[noreturn]] void die(const string &msg) {          // the reference's convention: message, exit(1)
    cerr << "Error:" << msg << endl;
    exit(1);
}

bool ends_with_gz(const string &s) { return s.size() >= 3 && s.rfind(".gz") == s.size() - 3; }

std::vector<string> split(const string &s, char sep) {
    std::vector<string> out;
    string cur;
    for (char c : s) {
        if (c == sep) { out.push_back(cur); cur.clear(); }
        else cur.push_back(c);
    }
    out.push_back(cur);
    return out;
}

string ltrim(const string &s) {                      // chomp_space(...,"all") strips leading blanks only
    size_t i = 0;
    while (i < s.size() && (s[i] == ' ' || s[i] == '\t')) ++i;
    return s.substr(i);
}

void load_adapters(const char *arg, std::vector<string> &dst, int which) {   // src/process_argv.cpp:242-304
    std::ifstream f(arg);
    if (!f) {
        const string valid = "ACGTacgtNn", seq(arg);
        for (char c : seq)
            if (valid.find(c) == string::npos) die(string("invalid character found in adapter:") + c + ". Only ACGTacgtNn are supported");
        dst.push_back(seq);
        return;
    }
    cout << "input adapter" << which << " list file:" << arg << endl;
    string line;
    while (std::getline(f, line)) dst.push_back(line);
}

void pair_int(const string &v, int32_t &a, int32_t &b) {   // "x[,y]"
    auto e = split(v, ',');
    a = atoi(e[0].c_str());
    b = e.size() > 1 ? atoi(e[1].c_str()) : a;
}
void pair_float(const string &v, float &a, float &b) {
    auto e = split(v, ',');
    a = (float)atof(e[0].c_str());
    b = e.size() > 1 ? (float)atof(e[1].c_str()) : a;
}

void read_config(Options &o, const char *path) {             // src/process_argv.cpp:1158-1638
    static const char *legal[] = {"trimFq1", "trimFq2", "seqType", "outFileType", "contam_trim", "contam1", "contam2",
        "ctMatchR", "global_contams", "glob_cotm_mR", "glob_cotm_mM", "tile", "fov", "index", "qualSys", "outQualSys",
        "baseConvert", "maxBaseQuality", "overlap", "mis", "pe_info", "patch", "maxReadLen", "adaMis", "adaMR", "adaEdge",
        "adaRCtg", "adaRAr", "adaRMa", "adaREr", "adaRMm", "log", "totalReadsNum", "cleanOutSplit", "trim", "trimBadHead",
        "trimBadTail", "barcodeListPath", "barcodeRegionStr", "notCutNoLFR", "inputAsList", "tenX", "rmdup"};
    std::ifstream f(path);
    if (!f) die(string("cannot open such file,") + path);
    string line;
    while (std::getline(f, line)) {
        if (line.find("#") == 0 || line.empty()) continue;
        string key = line, val;
        if (line.find("=") != string::npos) {
            auto e = split(line, '=');
            if (e.size() != 2) die("unrecgonized format parameter," + line);
            key = ltrim(e[0]);
            val = ltrim(e[1]);
        } else {
            key = ltrim(line);
        }
        bool ok = false;
        for (const char *l : legal) ok = ok || key == l;
        if (!ok) die("no such parameter," + key);
        snk_params &p = o.p;
        if (key == "qualSys") { int v = atoi(val.c_str()); p.quality_phred = v == 1 ? 64 : (v == 2 ? 33 : v); }
        else if (key == "outQualSys") { int v = atoi(val.c_str()); p.output_quality_phred = v == 1 ? 64 : (v == 2 ? 33 : v); }
        else if (key == "maxBaseQuality") p.max_base_quality = atoi(val.c_str());
        else if (key == "maxReadLen") p.max_read_length = atoi(val.c_str());
        else if (key == "adaMis") pair_int(val, p.ada_mis[0], p.ada_mis[1]);
        else if (key == "adaMR") pair_float(val, p.ada_mr[0], p.ada_mr[1]);
        else if (key == "adaEdge") pair_int(val, p.ada_edge[0], p.ada_edge[1]);
        else if (key == "patch") o.patch_size = atoi(val.c_str());
        else if (key == "log") o.log = val;
        else if (key == "trim") o.trim = val;
        else if (key == "trimBadHead") o.trim_bad_head = val;
        else if (key == "trimBadTail") o.trim_bad_tail = val;
        else if (key == "outFileType") o.out_file_type = val;
        else if (key == "seqType") { /* only affects tile/index parsing, not on this path */ }
        else if (key == "rmdup") p.rmdup = 1;
        else die("parameter " + key + " is not supported by the GPU filter path yet");
    }
}

void usage() {
    cout << "Usage: SOAPnuke filter [OPTION]... \n"
            "  -1, --fq1 FILE  -2, --fq2 FILE  -C, --cleanFq1 FILE  -D, --cleanFq2 FILE  -o, --outDir DIR\n"
            "  -c, --configFile FILE  -f, --adapter1 SEQ|FILE  -r, --adapter2 SEQ|FILE  -J, --ada_trim\n"
            "  -l, --lowQual INT [5]  -q, --qualRate FLOAT [0.5]  -n, --nRate FLOAT [0.05]  -m, --mean INT\n"
            "  -p, --highA FLOAT  -g, --polyG_tail FLOAT  -X, --polyX INT  -4, --minReadLen INT [30]\n"
            "  -x, --trimBadHead Q,LEN  -y, --trimBadTail Q,LEN  -t, --trim H1,T1,H2,T2  -T, --thread INT [6]\n"
            "  -h, --help  -v, --version\n";
}

void parse_args(int argc, char **argv, Options &o) {         // src/process_argv.cpp:72-552
    static const char *shortopts = "j1:2:C:D:o:c:E:Jf:r:l:q:m:x:y:n:p:g:X:t:T:3:4:L:w:hv";
    static const struct option longopts[] = {
        {"fq1", 1, NULL, '1'}, {"fq2", 1, NULL, '2'}, {"cleanFq1", 1, NULL, 'C'}, {"cleanFq2", 1, NULL, 'D'},
        {"outDir", 1, NULL, 'o'}, {"configFile", 1, NULL, 'c'}, {"adapter1", 1, NULL, 'f'}, {"adapter2", 1, NULL, 'r'},
        {"ada_trim", 0, NULL, 'J'}, {"lowQual", 1, NULL, 'l'}, {"qualRate", 1, NULL, 'q'}, {"nRate", 1, NULL, 'n'},
        {"mean", 1, NULL, 'm'}, {"highA", 1, NULL, 'p'}, {"polyG_tail", 1, NULL, 'g'}, {"polyX", 1, NULL, 'X'},
        {"minReadLen", 1, NULL, '4'}, {"trimBadHead", 1, NULL, 'x'}, {"trimBadTail", 1, NULL, 'y'}, {"trim", 1, NULL, 't'},
        {"thread", 1, NULL, 'T'}, {"help", 0, NULL, 'h'}, {"version", 0, NULL, 'v'}, {NULL, 0, NULL, 0}};
    snk_params_default(&o.p);
    int c;
    while ((c = getopt_long(argc, argv, shortopts, longopts, NULL)) != -1) {
        switch (c) {
        case '1': o.fq1 = optarg; o.in_gz = ends_with_gz(o.fq1); break;
        case '2': o.fq2 = optarg; break;
        case 'C': o.clean1 = optarg; o.out_gz = ends_with_gz(o.clean1); break;
        case 'D': o.clean2 = optarg; break;
        case 'o': o.out_dir = optarg; break;
        case 'c': read_config(o, optarg); break;
        case 'f': load_adapters(optarg, o.ada1, 1); break;
        case 'r': load_adapters(optarg, o.ada2, 2); break;
        case 'J': o.p.ada_trim = 1; break;
        case 'l': o.p.low_qual = atoi(optarg); break;
        case 'q': o.p.low_qual_ratio = (float)atof(optarg); break;
        case 'm': o.p.mean_quality = atoi(optarg); break;
        case 'x': o.trim_bad_head = optarg; break;
        case 'y': o.trim_bad_tail = optarg; break;
        case 'n': o.p.n_ratio = (float)atof(optarg); break;
        case 'p': o.p.highA_ratio = (float)atof(optarg); break;
        case 'g': o.p.polyG_tail = (float)atof(optarg); break;
        case 'X': o.p.polyX_num = (int)atof(optarg); break;
        case 't': o.trim = optarg; break;
        case 'T': o.threads = atoi(optarg); break;
        case '4': o.p.min_read_length = atoi(optarg); break;
        case 'v': cerr << "SOAPnuke filter tools version 2.1.9 (MI355X hot path)" << endl; exit(1);
        case 'h': usage(); exit(1);
        case 'j': case 'E': case 'w': die("option not supported by the GPU filter path yet");
        default: exit(1);
        }
    }
    if (argc != optind + 1) die("please check the options");
    if (string(argv[optind]) != "filter") die("only the filter module is built on this path");
    // check_parameter(), src/process_argv.cpp:554-917 (the checks that matter on this path)
    if (o.fq1.empty()) die("input fastq1 file is required");
    if (o.out_dir.empty()) die("output directory is required");
    if (o.clean1.empty()) die("output clean fastq1 file is required");
    if (!o.fq2.empty() && o.clean2.empty()) die("output clean fastq2 file is required");
    if (!o.fq2.empty() && ends_with_gz(o.fq2) != o.in_gz) die("input fastq files should be both gz format or not");
    if (!o.clean2.empty() && ends_with_gz(o.clean2) != o.out_gz) die("output clean fastq files should be both gz format or not");
    if (o.threads < 1) die("thread number should be positive");
    o.p.paired = o.fq2.empty() ? 0 : 1;
    if (!o.trim.empty()) {
        auto e = split(o.trim, ',');
        if ((int)e.size() != (o.p.paired ? 4 : 2)) die("trim value format error");
        o.p.has_hard_trim = 1;
        for (size_t i = 0; i < e.size(); ++i) o.p.hard_trim[i] = atoi(e[i].c_str());
    }
    if (!o.trim_bad_head.empty() || !o.trim_bad_tail.empty()) {
        auto h = split(o.trim_bad_head, ','), t = split(o.trim_bad_tail, ',');
        if (h.size() != 2 && t.size() != 2) die("low quality base at end format error," + o.trim_bad_head + " " + o.trim_bad_head);
        o.p.has_lq_trim = 1;
        if (h.size() == 2) { o.p.lq_head_qual = atoi(h[0].c_str()); o.p.lq_head_len = atoi(h[1].c_str()); }
        if (t.size() == 2) { o.p.lq_tail_qual = atoi(t[0].c_str()); o.p.lq_tail_len = atoi(t[1].c_str()); }
    }
    if (o.ada1.size() > SNK_MAX_ADAPTERS || o.ada2.size() > SNK_MAX_ADAPTERS) die("too many adapters");
    o.p.n_adapters[0] = (int)o.ada1.size();
    o.p.n_adapters[1] = (int)o.ada2.size();
    for (size_t i = 0; i < o.ada1.size(); ++i) o.p.adapters[0][i] = o.ada1[i].c_str();
    for (size_t i = 0; i < o.ada2.size(); ++i) o.p.adapters[1][i] = o.ada2[i].c_str();
    if (o.log.find("/") == string::npos) o.log = o.out_dir + "/" + o.log;
    if (o.out_file_type != "fastq" && o.out_file_type != "fasta") die("output_file_type value error");
}

string local_time() {                                   // get_local_time(), src/gc.cpp:186-199 (unpadded)
    time_t t = time(NULL);
    struct tm *l = localtime(&t);
    std::ostringstream s;
    s << l->tm_year + 1900 << "-" << l->tm_mon + 1 << "-" << l->tm_mday << "  " << l->tm_hour << ":" << l->tm_min << ":" << l->tm_sec;
    return s.str();
}

// ------------------------------------------------------------------ FASTQ in
struct Reader {
    gzFile f = nullptr;
    std::vector<char> buf;
    size_t pos = 0, end = 0;
    bool eof = false;
    void open(const string &path) {
        struct stat st;
        if (stat(path.c_str(), &st) != 0 || st.st_size == 0) die("cannot open file or empty file," + path);
        f = gzopen(path.c_str(), "rb");                // reads plain files transparently
        if (!f) die("cannot open file," + path);
        gzbuffer(f, 1 << 22);
        buf.resize(1 << 24);
    }
    bool line(const char *&s, int &n) {                // next line without its '\n' / '\r'
        for (;;) {
            char *nl = pos < end ? (char *)memchr(&buf[pos], '\n', end - pos) : nullptr;
            if (nl) {
                s = &buf[pos];
                n = (int)(nl - s);
                pos = (size_t)(nl - buf.data()) + 1;
                if (n > 0 && s[n - 1] == '\r') --n;
                return true;
            }
            if (eof) {
                if (pos >= end) return false;
                s = &buf[pos];
                n = (int)(end - pos);
                pos = end;
                return true;
            }
            memmove(buf.data(), &buf[pos], end - pos);
            end -= pos;
            pos = 0;
            if (end == buf.size()) buf.resize(buf.size() * 2);
            int got = gzread(f, &buf[end], (unsigned)(buf.size() - end));
            if (got < 0) die("read error in input fastq");
            if (got == 0) eof = true;
            end += (size_t)got;
        }
    }
};

struct Writer {
    gzFile gz = nullptr;
    FILE *fp = nullptr;
    string acc;
    void open(const string &path, bool gzip) {
        if (gzip) {
            gz = gzopen(path.c_str(), "wb");
            if (!gz) die("cannot write to the file," + path);
            gzsetparams(gz, 2, Z_DEFAULT_STRATEGY);    // level 2 as src/peprocess.cpp:1809
            gzbuffer(gz, 1 << 23);
        } else {
            fp = fopen(path.c_str(), "w");
            if (!fp) die("cannot write to the file," + path);
        }
    }
    void flush() {
        if (acc.empty()) return;
        if (gz) gzwrite(gz, acc.data(), (unsigned)acc.size());
        else fwrite(acc.data(), 1, acc.size(), fp);
        acc.clear();
    }
    void close() { flush(); if (gz) gzclose(gz); if (fp) fclose(fp); }
};

#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) die(string(#x) + ": " + hipGetErrorString(e_)); } while (0)

struct HostBatch {
    int mates, pitch, cap;
    uint8_t *seq[2] = {nullptr, nullptr}, *qual[2] = {nullptr, nullptr};
    uint16_t *len[2] = {nullptr, nullptr};
    std::vector<string> ids[2];
    int n = 0;
};

}  // namespace

int main(int argc, char **argv) {
    if (argc < 2) { usage(); return 1; }
    Options o;
    parse_args(argc, argv, o);
    const int mates = o.p.paired ? 2 : 1;
    mkdir(o.out_dir.c_str(), 0755);
    std::ofstream log(o.log.c_str());
    if (!log) die("cannot open such file," + o.log);
    log << local_time() << "\tAnalysis start!" << endl;

    Reader rd[2];
    rd[0].open(o.fq1);
    if (mates == 2) rd[1].open(o.fq2);
    Writer wr[2];
    wr[0].open(o.out_dir + "/" + o.clean1, o.out_gz);
    if (mates == 2) wr[1].open(o.out_dir + "/" + o.clean2, o.out_gz);

    // ---- first batch decides the capacity (longest read) and the pitch
    const int B = o.batch_pairs;
    struct Rec { string id, seq, qual; };
    std::vector<Rec> first[2];
    auto read_record = [&](int m, Rec &r) -> bool {
        const char *s; int n;
        if (!rd[m].line(s, n)) return false;
        r.id.assign(s, n);
        if (!rd[m].line(s, n)) die("truncated fastq record");
        r.seq.assign(s, n);
        if (!rd[m].line(s, n)) die("truncated fastq record");
        if (!rd[m].line(s, n)) die("truncated fastq record");
        r.qual.assign(s, n);
        if (r.qual.size() != r.seq.size()) die("sequence and quality lengths differ," + r.id);
        return true;
    };
    int maxlen = 1;
    for (int i = 0; i < B; ++i) {
        Rec a, b;
        const bool ok1 = read_record(0, a);
        const bool ok2 = mates == 2 ? read_record(1, b) : ok1;
        if (ok1 != ok2) die("reads number in fq1 and fq2 are different");
        if (!ok1) break;
        maxlen = std::max<int>(maxlen, (int)a.seq.size());
        first[0].push_back(std::move(a));
        if (mates == 2) { maxlen = std::max<int>(maxlen, (int)b.seq.size()); first[1].push_back(std::move(b)); }
    }
    if (first[0].empty()) die("no data");
    if (maxlen > SNK_READ_MAX_LEN) die("read longer than 1000 bases");
    o.p.max_read_len = maxlen;
    const int pitch = (maxlen + 15) / 16 * 16;

    HIPCHK(hipSetDevice(o.device));
    snk_ctx *ctx = snk_create(&o.p, o.device);
    if (!ctx) die(snk_last_error());
    int32_t lcap, nq; int64_t nsum;
    snk_stats_geometry(ctx, &lcap, &nq, &nsum);
    // one accumulator per virtual reference thread
    const int T = o.threads;
    const int64_t vblock = snk_vthread_block(T, o.patch_size);
    std::vector<uint64_t *> d_sum(T), d_max(T);
    for (int t = 0; t < T; ++t) {
        HIPCHK(hipMalloc(&d_sum[t], nsum * sizeof(uint64_t)));
        HIPCHK(hipMalloc(&d_max[t], SNK_MAX_N * sizeof(uint64_t)));
        HIPCHK(hipMemset(d_sum[t], 0, nsum * sizeof(uint64_t)));
        HIPCHK(hipMemset(d_max[t], 0, SNK_MAX_N * sizeof(uint64_t)));
    }
    // pinned SoA planes + device mirrors
    uint8_t *h_seq[2], *h_qual[2], *d_seq[2], *d_qual[2];
    uint16_t *h_len[2], *d_len[2];
    snk_read_result *h_rec[2], *d_rec[2];
    const size_t plane = (size_t)B * pitch;
    for (int m = 0; m < mates; ++m) {
        HIPCHK(hipHostMalloc(&h_seq[m], plane)); HIPCHK(hipHostMalloc(&h_qual[m], plane));
        HIPCHK(hipHostMalloc(&h_len[m], (size_t)B * 2)); HIPCHK(hipHostMalloc(&h_rec[m], (size_t)B * sizeof(snk_read_result)));
        HIPCHK(hipMalloc(&d_seq[m], plane)); HIPCHK(hipMalloc(&d_qual[m], plane));
        HIPCHK(hipMalloc(&d_len[m], (size_t)B * 2)); HIPCHK(hipMalloc(&d_rec[m], (size_t)B * sizeof(snk_read_result)));
        memset(h_seq[m], 0, plane); memset(h_qual[m], 0, plane);
    }
    // ---- rmdup pre-pass (src/peprocess.cpp:3071-3152): hash every raw pair on the GPU, keep the hashes
    // resident, mark every later occurrence; the flags enter the cascade as snk_batch.dup below.
    uint8_t *d_dup_all = nullptr;
    Writer dupw[2][64];
    if (o.p.rmdup) {
        if (T > 64) die("rmdup: more than 64 threads");
        Reader pr[2];
        pr[0].open(o.fq1);
        if (mates == 2) pr[1].open(o.fq2);
        std::vector<uint64_t *> chunks;
        std::vector<int> chunk_n;
        uint64_t nall = 0;
        for (;;) {
            int n = 0;
            for (; n < B; ++n) {
                bool ok[2] = {true, true};
                for (int m = 0; m < mates; ++m) {
                    const char *sq; int ln;
                    if (!pr[m].line(sq, ln)) { ok[m] = false; continue; }          // id
                    if (!pr[m].line(sq, ln)) die("truncated fastq record");
                    if (ln > lcap) die("read longer than the first batch's longest read (" + std::to_string(lcap) + ")");
                    memcpy(h_seq[m] + (size_t)n * pitch, sq, ln);
                    h_len[m][n] = (uint16_t)ln;
                    const char *t; int tn;
                    if (!pr[m].line(t, tn) || !pr[m].line(t, tn)) die("truncated fastq record");
                }
                if (mates == 2 && ok[0] != ok[1]) die("reads number in fq1 and fq2 are different");
                if (!ok[0]) break;
            }
            if (n == 0) break;
            uint64_t *dh;
            HIPCHK(hipMalloc(&dh, (size_t)n * sizeof(uint64_t)));
            snk_batch b;
            memset(&b, 0, sizeof b);
            b.n = n;
            b.pitch = pitch;
            for (int m = 0; m < mates; ++m) {
                HIPCHK(hipMemcpyAsync(d_seq[m], h_seq[m], (size_t)n * pitch, hipMemcpyHostToDevice, 0));
                HIPCHK(hipMemcpyAsync(d_len[m], h_len[m], (size_t)n * 2, hipMemcpyHostToDevice, 0));
                b.seq[m] = d_seq[m];
                b.qual[m] = d_qual[m];
                b.len[m] = d_len[m];
            }
            if (snk_rmdup_hash_device(ctx, &b, dh, nullptr) != SNK_OK) die(snk_last_error());
            HIPCHK(hipStreamSynchronize(0));                 // the pinned planes are refilled next
            chunks.push_back(dh);
            chunk_n.push_back(n);
            nall += (uint64_t)n;
            if (n < B) break;
        }
        if (nall > 4294967295ull) die("reads number is too large to do remove duplication," + std::to_string(nall));
        uint64_t *d_hash_all;
        HIPCHK(hipMalloc(&d_hash_all, (size_t)nall * sizeof(uint64_t)));
        HIPCHK(hipMalloc(&d_dup_all, (size_t)nall));
        uint64_t off = 0;
        for (size_t k = 0; k < chunks.size(); ++k) {
            HIPCHK(hipMemcpyAsync(d_hash_all + off, chunks[k], (size_t)chunk_n[k] * sizeof(uint64_t), hipMemcpyDeviceToDevice, 0));
            off += (uint64_t)chunk_n[k];
        }
        HIPCHK(hipStreamSynchronize(0));
        for (uint64_t *c : chunks) HIPCHK(hipFree(c));
        cout << "totalReadsNum:\t" << nall << endl;
        if (snk_rmdup_mark_device(ctx, d_hash_all, nullptr, (int64_t)nall, nall, -1, d_dup_all, nullptr) != SNK_OK) die(snk_last_error());
        HIPCHK(hipStreamSynchronize(0));
        HIPCHK(hipFree(d_hash_all));
        std::vector<uint8_t> flags((size_t)nall);
        HIPCHK(hipMemcpy(flags.data(), d_dup_all, (size_t)nall, hipMemcpyDeviceToHost));
        uint64_t ndup = 0;
        for (uint8_t f : flags) ndup += f;
        log << "duplicate reads number:\t" << ndup << endl;
        if (mates == 1) {
            // Reference quirk (SE only): seProcess records "reads so far" BEFORE counting the quality line of
            // the patch's last read (src/seprocess.cpp:1086,1159 vs src/peprocess.cpp:2147), so inside every
            // full patch read i is filtered with the flag of read i-1 (read 0: the byte in front of the
            // array, 0 in practice); only the partial patch at the end of the file (:1112) is aligned.
            // Reproduced here, on the host, so that outputs stay identical; the C ABI flags are the true ones.
            const uint64_t ps = o.patch_size > 0 ? (uint64_t)o.patch_size : (uint64_t)T * 20000 / 8;
            const uint64_t full_end = nall / ps * ps;
            std::vector<uint8_t> eff(flags);
            for (uint64_t i = 0; i < full_end; ++i) eff[i] = i ? flags[i - 1] : 0;
            HIPCHK(hipMemcpy(d_dup_all, eff.data(), (size_t)nall, hipMemcpyHostToDevice));
        }
        for (int t = 0; t < T; ++t)                            // dupReads.<thread>.<mate>.gz, src/peprocess.cpp:167-174
            for (int m = 0; m < mates; ++m)                    // SE: only .1.gz (src/seprocess.cpp:89)
                dupw[m][t].open(o.out_dir + "/dupReads." + std::to_string(t) + "." + std::to_string(m + 1) + ".gz", true);
        for (int m = 0; m < mates; ++m) { memset(h_seq[m], 0, plane); }
    }
    uint64_t ndup_written = 0;
    std::vector<string> ids[2], raw_seq[2], raw_qual[2];
    uint64_t total = 0;
    const int dq = o.p.output_quality_phred - o.p.quality_phred;
    bool more = true;
    std::vector<Rec> cur[2];
    cur[0].swap(first[0]);
    cur[1].swap(first[1]);
    while (!cur[0].empty()) {
        const int n = (int)cur[0].size();
        for (int m = 0; m < mates; ++m)
            for (int i = 0; i < n; ++i) {
                const Rec &r = cur[m][i];
                if ((int)r.seq.size() > lcap) die("read longer than the first batch's longest read (" + std::to_string(lcap) + "): " + r.id);
                memcpy(h_seq[m] + (size_t)i * pitch, r.seq.data(), r.seq.size());
                memcpy(h_qual[m] + (size_t)i * pitch, r.qual.data(), r.qual.size());
                h_len[m][i] = (uint16_t)r.seq.size();
            }
        for (int m = 0; m < mates; ++m) {
            HIPCHK(hipMemcpyAsync(d_seq[m], h_seq[m], (size_t)n * pitch, hipMemcpyHostToDevice, 0));
            HIPCHK(hipMemcpyAsync(d_qual[m], h_qual[m], (size_t)n * pitch, hipMemcpyHostToDevice, 0));
            HIPCHK(hipMemcpyAsync(d_len[m], h_len[m], (size_t)n * 2, hipMemcpyHostToDevice, 0));
        }
        // split at virtual-thread block boundaries so every segment lands in its thread's accumulator
        for (int lo = 0; lo < n;) {
            const uint64_t g = total + (uint64_t)lo;
            const int vt = (int)((g / (uint64_t)vblock) % (uint64_t)T);
            const uint64_t next = (g / (uint64_t)vblock + 1) * (uint64_t)vblock;
            const int hi = (int)std::min<uint64_t>((uint64_t)n, next - total);
            snk_batch b;
            memset(&b, 0, sizeof b);
            b.n = hi - lo;
            b.pitch = pitch;
            for (int m = 0; m < mates; ++m) {
                b.seq[m] = d_seq[m] + (size_t)lo * pitch;
                b.qual[m] = d_qual[m] + (size_t)lo * pitch;
                b.len[m] = d_len[m] + lo;
            }
            b.first_index = g;
            if (d_dup_all) b.dup = d_dup_all + g;
            if (snk_bind_stats(ctx, d_sum[vt], d_max[vt]) != SNK_OK) die(snk_last_error());
            if (snk_filter_batch_device(ctx, &b, d_rec[0] + lo, mates == 2 ? d_rec[1] + lo : nullptr, nullptr, 0) != SNK_OK)
                die(snk_last_error());
            lo = hi;
        }
        for (int m = 0; m < mates; ++m)
            HIPCHK(hipMemcpyAsync(h_rec[m], d_rec[m], (size_t)n * sizeof(snk_read_result), hipMemcpyDeviceToHost, 0));
        HIPCHK(hipStreamSynchronize(0));
        // clean output, input order (src/peprocess.cpp:3383-3484)
        for (int i = 0; i < n; ++i) {
            if (d_dup_all && h_rec[0][i].reason == SNK_R_DUP) {          // C_fastq::toString of the raw records, :1541
                const int vt = (int)(((total + (uint64_t)i) / (uint64_t)vblock) % (uint64_t)T);
                for (int m = 0; m < mates; ++m) {
                    const Rec &r = cur[m][i];
                    string &out = dupw[m][vt].acc;
                    out += r.id; out += '\n'; out += r.seq; out += "\n+\n"; out += r.qual; out += '\n';
                    if (out.size() > (1u << 22)) dupw[m][vt].flush();
                }
                ++ndup_written;
            }
            if (h_rec[0][i].reason != SNK_KEEP) continue;
            for (int m = 0; m < mates; ++m) {
                const Rec &r = cur[m][i];
                const snk_read_result &x = h_rec[m][i];
                string &out = wr[m].acc;
                if (o.out_file_type == "fasta") {
                    string id = r.id;
                    const size_t at = id.find("@");
                    if (at != string::npos) id.replace(at, 1, ">");
                    out += id; out += '\n';
                    out.append(r.seq, x.clean_start, x.clean_len); out += '\n';
                } else {
                    out += r.id; out += '\n';
                    out.append(r.seq, x.clean_start, x.clean_len);
                    out += "\n+\n";
                    if (dq == 0) out.append(r.qual, x.clean_start, x.clean_len);
                    else for (int k = 0; k < x.clean_len; ++k) out += (char)(r.qual[x.clean_start + k] + dq);
                    out += '\n';
                }
                if (out.size() > (1u << 24)) wr[m].flush();
            }
        }
        total += (uint64_t)n;
        log << local_time() << " processed_reads:\t" << total << endl;
        // next batch
        for (int m = 0; m < 2; ++m) cur[m].clear();
        if (more) {
            for (int i = 0; i < B; ++i) {
                Rec a, b;
                const bool ok1 = read_record(0, a);
                const bool ok2 = mates == 2 ? read_record(1, b) : ok1;
                if (ok1 != ok2) die("reads number in fq1 and fq2 are different");
                if (!ok1) { more = false; break; }
                cur[0].push_back(std::move(a));
                if (mates == 2) cur[1].push_back(std::move(b));
            }
        }
    }
    for (int m = 0; m < mates; ++m) wr[m].close();
    if (d_dup_all) {
        for (int t = 0; t < T; ++t) for (int m = 0; m < mates; ++m) dupw[m][t].close();
        log << "dup number:\t" << ndup_written << endl;
    }

    // ---- stats: finalize each virtual thread's block, fetch, check errors, write the reports
    std::vector<std::vector<uint64_t>> sums(T, std::vector<uint64_t>(nsum)), maxs(T, std::vector<uint64_t>(SNK_MAX_N));
    std::vector<const uint64_t *> sp(T), mp(T);
    for (int t = 0; t < T; ++t) {
        snk_error err;
        if (snk_bind_stats(ctx, d_sum[t], d_max[t]) != SNK_OK) die(snk_last_error());
        if (snk_stats_fetch(ctx, sums[t].data(), maxs[t].data(), &err, nullptr) != SNK_OK) die(snk_last_error());
        if (err.code == SNK_E_BAD_BASE) die("unrecognized sequence, read " + std::to_string(err.index) + " of fq" + std::to_string(err.mate + 1));
        if (err.code == SNK_E_EMPTY_SEQ) die("empty sequence");
        if (err.code == SNK_E_QUAL_RANGE) die("quality is too high or too low,please check the quality system parameter or fastq file");
        if (err.code) die("device reported error " + std::to_string(err.code));
        sp[t] = sums[t].data();
        mp[t] = maxs[t].data();
    }
    char ebuf[512];
    if (snk_write_reports(&o.p, T, sp.data(), mp.data(), o.out_dir.c_str(), ebuf, sizeof ebuf) != 0) { cerr << ebuf << endl; return 1; }
    log << local_time() << "\tAnalysis accomplished!" << endl;
    snk_destroy(ctx);
    return 0;
}
