// snk_main.cpp -- `SOAPnuke filter` command line on top of the C ABI (include/snk_filter.h).
//
// Host counterpart of main() + peProcess::process()/seProcess::process() of the reference
// (src/main.cpp:17-68, src/peprocess.cpp:3051-3201, src/seprocess.cpp:2087): same options,
// config-file keys, output files and report bytes for the `filter` module -- but one pass over
// the input, FASTQ parsed once into pinned structure-of-arrays batches, the per-read hot path on
// the GPU, clean reads written in input order by the host.  No temp files, no `cat`.
//
// Host pipeline: one reader thread per input file (inflate / read + line index), a
// pack stage (records -> pinned SoA planes, parallel over records), the GPU stage (async copies and
// kernels on a stream per slot), and a write stage (parallel formatting; for .gz output every worker
// deflates its slice into its own gzip member, level 2, members concatenated in input order -- the
// same legal-gzip trick the reference uses for its per-thread part files, src/peprocess.cpp:2386).
// Three batch slots are in flight, so reading, packing, the GPU and writing overlap.
#include <dlfcn.h>
#include <getopt.h>
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sched.h>
#include <sys/stat.h>
#include <unistd.h>
#include <spawn.h>
#include <sys/wait.h>
#include <time.h>
#include <zlib.h>

#include <atomic>
#include <condition_variable>
#include <deque>
#include <fstream>
#include <functional>
#include <iostream>
#include <memory>
#include <mutex>
#include <sstream>
#include <string>
#include <thread>
#include <map>
#include <vector>

#include "../../include/snk_filter.h"
#include "snk_inflate.h"
#include "snk_pgunzip.h"
#include "snk_dgunzip.h"
#include "snk_wire.h"
#include "snk_deflate.h"
#include "snk_report.h"
#include "../../include/snk_rmdup.h"
#include "../../include/snk_fastq.h"
#include <immintrin.h>

using std::cerr;
using std::cout;
using std::endl;
using std::string;

namespace {

struct Options {
    string fq1, fq2, clean1, clean2, out_dir, log = "log";
    std::vector<string> ada1, ada2;
    std::vector<const char *> ada_ptr[2];                             // snk_params.adapter_list (lists of any length)
    snk_params p;
    string trim, trim_bad_head, trim_bad_tail, out_file_type = "fastq";
    int threads = 6, patch_size = 0, batch_pairs = 1 << 18;
    std::vector<int> devices;                                         // --devices 0,1,...: HIP devices the batches go round (default: 0)
    bool in_gz = true, out_gz = true, pe_info = false, index_remove = false;
    string seq_type = "0";
    string contam[2], ct_match_r, global_contams, g_mrs, g_mms;     // kept here: snk_params points into them
    string trim_fq[2];                                               // trimFq1 / trimFq2 (gz): trimmed, not filtered
    int trim_fq_gz[2] = {-1, -1};                                    // (a shard: its part's name no longer ends like the file's)
    string tile, fov;                                                // reads of these tiles / fovs are dropped (by name)
    string base_convert;
    // limits on the clean output (src/process_argv.cpp:426-443,1476-1552)
    uint64_t clean_out_split = 0;        // -w / cleanOutSplit: split.<k>.<cleanFq> files of that many reads
    float total_reads = 0;               // totalReadsNum as atof() saw it (> 0: set)
    float total_ratio = 0;               //   < 1: a ratio of the clean reads
    uint64_t total_num = 0;              //   >= 1: a number of reads
    bool total_head = false;             //   "<N>head": the first N clean reads; otherwise every k-th read
    std::vector<string> wrong_paras;     // sRNA adapter keys: an error in this module
    bool streaming = false;              // -j: clean reads and cumulative statistics of every patch on stdout
};

[[noreturn]] void die(const string &msg) {          // the reference's convention: message, exit(1)
    cerr << "Error:" << msg << endl;
    cout.flush();
    fflush(stdout);
    _exit(1);                                        // (no exit handlers: other threads may be inside the HIP runtime)
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
        else if (key == "seqType") { o.seq_type = val; if (val != "0" && val != "1") die("seq_type value should be 0 or 1"); }
        else if (key == "index") o.index_remove = true;
        else if (key == "tile") o.tile = val;
        else if (key == "fov") o.fov = val;
        else if (key == "trimFq1") o.trim_fq[0] = val;
        else if (key == "trimFq2") o.trim_fq[1] = val;
        else if (key == "contam_trim") p.contam_trim = 1;
        else if (key == "contam1") o.contam[0] = val;
        else if (key == "contam2") o.contam[1] = val;
        else if (key == "ctMatchR") o.ct_match_r = val;
        else if (key == "global_contams") o.global_contams = val;
        else if (key == "glob_cotm_mR") o.g_mrs = val;
        else if (key == "glob_cotm_mM") o.g_mms = val;
        else if (key == "rmdup") p.rmdup = 1;
        else if (key == "cleanOutSplit") {                   // src/process_argv.cpp:1537-1554
            for (char ch : val) if (!isdigit((unsigned char)ch)) die("-w value should be a positive integer");
            o.clean_out_split = (uint64_t)atoi(val.c_str());
            if (o.clean_out_split == 0) die("-w value should be a positive integer");
        }
        else if (key == "totalReadsNum") {                   // src/process_argv.cpp:1476-1536
            string t = val;
            if (t.find("head") == string::npos) {
                o.total_head = false;
                for (char ch : t) if (!isdigit((unsigned char)ch) && ch != '.') die("-L value should be a positive integer or float");
            } else {
                o.total_head = true;
                t.erase(t.find("head"), 4);
                if (t.find(".") != string::npos) die("-L value should be a integer when with head suffix");
                for (char ch : t) if (!isdigit((unsigned char)ch)) die("-L value should be an integer when with head suffix");
            }
            const float tv = (float)atof(val.c_str());
            if (tv == 0) die("-L value should be a positive integer or float");
            o.total_reads = tv;
            if (tv < 1) o.total_ratio = tv;
            else { std::istringstream is(t); is >> o.total_num; }
            if (o.total_ratio > 0 && o.total_num > 0) die("reads number and ratio should not be both assigned at the same time");
        }
        else if (key == "pe_info") o.pe_info = true;
        else if (key == "baseConvert") o.base_convert = val;
        // accepted without effect on `filter`, as in the reference: `overlap` / `mis` feed whether_over_overlapped(), which nothing
        // calls (reads_result.over_lapped stays false, src/sequence.cpp:195,364); the stLFR keys and inputAsList are read by other modules
        else if (key == "overlap" || key == "mis" || key == "barcodeListPath" || key == "barcodeRegionStr" || key == "notCutNoLFR" ||
                 key == "inputAsList" || key == "tenX") {}
        else if (key == "adaRCtg") o.wrong_paras.push_back("-S|--adaRCtg");        // src/process_argv.cpp:1446-1470,763-771
        else if (key == "adaRAr") o.wrong_paras.push_back("-s|--adaRAr");
        else if (key == "adaRMa") o.wrong_paras.push_back("-U|--adaRMa");
        else if (key == "adaREr") o.wrong_paras.push_back("-u|--adaREr");
        else if (key == "adaRMm") o.wrong_paras.push_back("-b|--adaRMm");
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
            "  -w, --output_clean INT (reads per split.<k>.<cleanFq> file)  -j, --streaming (reads + statistics on stdout)\n"
            "      --devices LIST  HIP devices to use, e.g. 0,1,2,3 [0]\n"
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
        {"thread", 1, NULL, 'T'}, {"output_clean", 1, NULL, 'w'}, {"ref", 1, NULL, 'E'}, {"streaming", 0, NULL, 'j'}, {"devices", 1, NULL, 1000}, {"help", 0, NULL, 'h'}, {"version", 0, NULL, 'v'}, {NULL, 0, NULL, 0}};
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
        case 'w': {                                          // src/process_argv.cpp:426-443
            for (const char *q = optarg; *q; ++q) if (!isdigit((unsigned char)*q)) die("-w value should be a positive integer");
            o.clean_out_split = (uint64_t)atoi(optarg);
            if (o.clean_out_split == 0) die("-w value should be a positive integer");
            break;
        }
        case 'E': break;                                     // --ref: CRAM reference of the Hts module, unused by `filter`
        case 'j': o.streaming = true; break;
        case 1000:                                           // not a reference option: the GPUs of this node to use
            for (const string &e : split(optarg, ',')) {
                if (e.empty() || e.find_first_not_of("0123456789") != string::npos) die("--devices takes a comma separated list of HIP device numbers");
                o.devices.push_back(atoi(e.c_str()));
            }
            break;
        default: exit(1);
        }
    }
    if (const char *e = getenv("SNK_BATCH_PAIRS")) { const int v = atoi(e); if (v >= 64 && v <= (1 << 24)) o.batch_pairs = v; }   // pairs per pipeline slot (tuning / tests)
    if (o.devices.empty())                                   // no --devices: SNK_DEVICES (the same list; a wrapper's or a test's default), else device 0
        if (const char *e = getenv("SNK_DEVICES"))
            for (const string &x : split(e, ',')) {
                if (x.empty() || x.find_first_not_of("0123456789") != string::npos) die("SNK_DEVICES takes a comma separated list of HIP device numbers");
                o.devices.push_back(atoi(x.c_str()));
            }
    if (o.devices.empty()) o.devices.push_back(0);
    if (o.devices.size() > 16) die("--devices: at most 16 devices");
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
    // adapter lists of any length (a list file, src/process_argv.cpp:242-304): snk_params.adapter_list
    o.p.n_adapters[0] = (int)o.ada1.size();
    o.p.n_adapters[1] = (int)o.ada2.size();
    for (const string &a : o.ada1) o.ada_ptr[0].push_back(a.c_str());
    for (const string &a : o.ada2) o.ada_ptr[1].push_back(a.c_str());
    for (int m = 0; m < 2; ++m) o.p.adapter_list[m] = o.ada_ptr[m].empty() ? nullptr : o.ada_ptr[m].data();
    if (!o.fov.empty() && o.seq_type != "0") { cerr << "Warning:Zebra-500 data(--fov), --seqType is 0" << endl; exit(1); }   // src/read_filter.cpp:137-140
    o.p.contam[0] = o.contam[0].empty() ? nullptr : o.contam[0].c_str();
    o.p.contam[1] = o.contam[1].empty() ? nullptr : o.contam[1].c_str();
    o.p.ct_match_r = o.ct_match_r.empty() ? nullptr : o.ct_match_r.c_str();
    o.p.global_contams = o.global_contams.empty() ? nullptr : o.global_contams.c_str();
    o.p.g_mrs = o.g_mrs.empty() ? nullptr : o.g_mrs.c_str();
    o.p.g_mms = o.g_mms.empty() ? nullptr : o.g_mms.c_str();
    if (o.log.find("/") == string::npos) o.log = o.out_dir + "/" + o.log;
    if (o.out_file_type != "fastq" && o.out_file_type != "fasta") die("output_file_type value error");
    // limited clean output (src/process_argv.cpp:614-622,785-789,892-896)
    if (o.p.paired && (o.clean_out_split > 0 || o.total_reads > 0) && !ends_with_gz(o.clean1) && !ends_with_gz(o.clean2))
        die("the clean out fastq should be non-gz format when clean output reads are limited");
    {
        const uint64_t ps = o.patch_size > 0 ? (uint64_t)o.patch_size : (uint64_t)o.threads * 20000 / 8;
        if (o.clean_out_split != 0 && o.clean_out_split < ps) die(" output reads in each clean fastq file(-w) should be more than patch size(-e)");
    }
    if (o.clean_out_split > 0 && o.total_reads > 0) die("-w and -L cannot be both assigned");
    if (o.streaming)                                          // one batch = one patch of the reference (src/peprocess.cpp:2137-2148)
        o.batch_pairs = (int)(o.patch_size > 0 ? o.patch_size : o.threads * 20000 / 8);
    else if (o.p.rmdup && !o.p.paired) {
        // single-end rmdup in one pass: batches end on patch borders, so that only the file's last batch can hold a partial patch --
        // the one place where the reference's duplicate flags are not shifted by a read (see the rmdup pre-pass below)
        const int64_t ps = o.patch_size > 0 ? o.patch_size : (int64_t)o.threads * 20000 / 8;
        if (ps > 0 && ps <= (1 << 24)) o.batch_pairs = (int)std::max<int64_t>(ps, (int64_t)o.batch_pairs / ps * ps);
    }
    if (!o.wrong_paras.empty()) {
        string l = o.wrong_paras[0];
        for (size_t i = 1; i < o.wrong_paras.size(); ++i) l += "," + o.wrong_paras[i];
        die("these parameters should not appear in the module," + l);
    }
    if (!o.base_convert.empty()) {                               // src/process_argv.cpp:865-890
        const string acgt = "ACGTacgt", &b = o.base_convert;
        if (b.find("TO") == string::npos && b.find("2") == string::npos) die("base_convert value format error");
        if (acgt.find(b[0]) == string::npos || acgt.find(b[b.size() - 1]) == string::npos) die("base_convert value format error");
        // seProcess::preOutput calls string::replace(npos, ...) for whichever of "TO" / "2" is absent and dies of the
        // exception (src/seprocess.cpp:923-924): there is no single-end behaviour to reproduce
        if (!o.p.paired) die("baseConvert aborts the reference in single-end mode (src/seprocess.cpp:923); not supported");
    }
}

string local_time() {                                   // get_local_time(), src/gc.cpp:186-199 (unpadded)
    time_t t = time(NULL);
    struct tm *l = localtime(&t);
    std::ostringstream s;
    s << l->tm_year + 1900 << "-" << l->tm_mon + 1 << "-" << l->tm_mday << "  " << l->tm_hour << ":" << l->tm_min << ":" << l->tm_sec;
    return s.str();
}

// ------------------------------------------------------------------ tile / fov verdicts from the read name
// check_tile_or_fov(), src/read_filter.cpp:14-79: only whole list elements ever match; a range element
// "a-b" is parsed (and its format checked) but compared as a string, so it never matches a 4-digit tile.
bool check_tile_or_fov(const string &tile, const string &param) {
    auto check_range = [](const string &e) {
        if (split(e, '-').size() != 2) die("input tile parameter format error," + e);
    };
    if (param.find(",") == string::npos) {
        if (param.find("C") == string::npos && param.find("-") != string::npos) {
            check_range(param);
            const auto e = split(param, '-');
            return atoi(e[0].c_str()) <= atoi(e[1].c_str()) && tile == param;
        }
        return tile == param;
    }
    const bool fov_style = param.find("C") != string::npos;
    for (const string &e : split(param, ',')) {
        if (!fov_style && e.find("-") != string::npos) {
            check_range(e);
            const auto r = split(e, '-');
            if (atoi(r[0].c_str()) <= atoi(r[1].c_str()) && tile == e) return true;
        } else if (e == tile) return true;
    }
    return false;
}
// read_tile / read_fov of stat_read(), src/read_filter.cpp:86-150 (its stderr warnings are not reproduced)
string read_tile(const char *id, int n, const string &seq_type) {
    const int want = seq_type == "0" ? 2 : 4;
    int i = 0, num = 0;
    for (; i < n; ++i) {
        if (id[i] == ':') ++num;
        if (num >= want) break;
    }
    string t;
    for (int j = 0; j != 4; ++j) {
        const int k = i + j + 1;
        if (k < n && id[k] >= '0' && id[k] <= '9') t += id[k];
    }
    return t;
}
string read_fov(const char *id, int n) {
    int i = 0;
    for (; i < n; ++i)
        if (id[i] == 'C' && i + 8 < n && id[i + 4] == 'R') break;
    return i < n ? string(id + i, (size_t)std::min(8, n - i)) : string();
}

// ------------------------------------------------------------------ host pipeline pieces
template <class T>
class Channel {                                        // bounded FIFO between pipeline stages
    std::mutex m_;
    std::condition_variable cv_;
    std::deque<T> q_;
    size_t cap_;
    bool closed_ = false;
public:
    explicit Channel(size_t cap) : cap_(cap) {}
    void push(T v) {
        std::unique_lock<std::mutex> l(m_);
        cv_.wait(l, [&] { return q_.size() < cap_; });
        q_.push_back(std::move(v));
        cv_.notify_all();
    }
    bool pop(T &v) {                                   // false: closed and drained
        std::unique_lock<std::mutex> l(m_);
        cv_.wait(l, [&] { return !q_.empty() || closed_; });
        if (q_.empty()) return false;
        v = std::move(q_.front());
        q_.pop_front();
        cv_.notify_all();
        return true;
    }
    void close() {
        std::unique_lock<std::mutex> l(m_);
        closed_ = true;
        cv_.notify_all();
    }
};

// Host worker pool: parallel_for() cuts [0, n) into `workers` slices and runs them on the pool's threads (several
// pipeline stages call it concurrently; the caller takes slices too, so a busy pool never blocks it).
class Pool {
    std::mutex m_;
    std::condition_variable cv_;
    std::deque<std::function<void()>> q_;
    std::vector<std::thread> th_;
    bool quit_ = false;
    void loop() {
        for (;;) {
            std::function<void()> job;
            {
                std::unique_lock<std::mutex> l(m_);
                cv_.wait(l, [&] { return quit_ || !q_.empty(); });
                if (q_.empty()) return;
                job = std::move(q_.front());
                q_.pop_front();
            }
            job();
        }
    }
public:
    void start(int n) { for (int i = 0; i < n; ++i) th_.emplace_back([this] { loop(); }); }
    ~Pool() {
        { std::lock_guard<std::mutex> l(m_); quit_ = true; }
        cv_.notify_all();
        for (auto &t : th_) t.join();
    }
    bool try_run_one() {
        std::function<void()> job;
        {
            std::lock_guard<std::mutex> l(m_);
            if (q_.empty()) return false;
            job = std::move(q_.front());
            q_.pop_front();
        }
        job();
        return true;
    }
    void push(std::function<void()> j) {
        { std::lock_guard<std::mutex> l(m_); q_.push_back(std::move(j)); }
        cv_.notify_one();
    }
};
Pool g_pool;
int g_host_threads = 1;
// a child of a sharded run (run_sharded in main): its record range per input file, the global index of its first pair
struct ShardEnv {
    bool child = false;
    int g = 0, G = 1;
    uint64_t first = 0;
    std::map<string, std::pair<size_t, size_t>> range;     // input path -> [lo, hi) bytes
    string stats_path;
    // .gz input: where this shard's decoder starts (a block header + the 32 KiB of text in front of it, from the parent's scout
    // pass; bit == ~0: the stream's own header), what to drop behind that and how many records to hand on
    struct Gz { uint64_t bit = ~0ull, skip_bytes = 0, skip_records = 0, nrec = ~0ull; std::vector<uint8_t> win; };
    std::map<string, Gz> gz;
    uint64_t total = 0;                                     // records of the whole input (rmdup: rmdup::getPrime and the SE patch rule want it)
    string wire_kind, wire_addr;                            // "rccl" (addr = the communicator id in hex) | "host" (addr = socket name) | "file"
    std::unique_ptr<snk::ShardWire> wire;
    bool wire_tried = false;
} g_shard;

// the collectives between the shards (host/snk_wire.h), made on first use; null: this run merges through files only.  The host
// wire always comes up first (a socket); for the RCCL wire it is the bootstrap -- rank 0 makes the communicator id, everybody joins,
// everybody learns whether everybody could -- and the fallback: one shard that cannot join (no RCCL, a device RCCL refuses) puts ALL of
// them on the host wire, with a warning from rank 0, instead of leaving the others inside ncclCommInitRank.
snk::ShardWire *shard_wire() {
    if (!g_shard.child || g_shard.wire_tried) return g_shard.wire.get();
    g_shard.wire_tried = true;
    if (g_shard.wire_kind != "rccl" && g_shard.wire_kind != "host") return nullptr;
    string why;
    std::unique_ptr<snk::HostWire> hw(snk::HostWire::connect(g_shard.g, g_shard.G, g_shard.wire_addr, why));
    if (!hw) { cerr << "Error:shard " << g_shard.g << " cannot join the host wire (" << why << ")" << endl; _exit(1); }
    if (g_shard.wire_kind == "rccl") {
        char id[257];
        memset(id, 0, sizeof id);
        string why0;
        if (g_shard.g == 0) { const string s = snk::RcclWire::make_id(why0); if (s.size() == 256) memcpy(id, s.data(), 256); }
        if (!hw->bcast_bytes(id, 256)) { cerr << "Error:shard " << g_shard.g << ": " << hw->err << endl; _exit(1); }
        std::unique_ptr<snk::RcclWire> rw;
        if (id[0]) rw.reset(snk::RcclWire::connect(g_shard.g, g_shard.G, string(id, 256), why));
        else why = why0.empty() ? "rank 0 could not make a communicator id" : why0;
        int64_t ok = rw ? 1 : 0;
        if (!hw->min_of_all(ok)) { cerr << "Error:shard " << g_shard.g << ": " << hw->err << endl; _exit(1); }
        if (ok) { g_shard.wire = std::move(rw); return g_shard.wire.get(); }
        if (!rw || g_shard.g == 0) cerr << "Warning:shard " << g_shard.g << ": no RCCL communicator over the shards (" << (rw ? "another shard could not join" : why) << "): the host wire carries the collectives" << endl;
    }
    g_shard.wire = std::move(hw);
    return g_shard.wire.get();
}

// CPUs this process may really use: the affinity mask, cut down to the cgroup's CPU quota (a container with 256
// visible hardware threads and a quota of 16 CPUs only gets slower with more than 16 busy threads)
int usable_cpus() {
    int n = (int)std::thread::hardware_concurrency();
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof set, &set) == 0) n = std::min(n > 0 ? n : 1 << 20, CPU_COUNT(&set));
    auto quota = [](const char *path, const char *path_period) -> double {
        FILE *f = fopen(path, "r");
        if (!f) return 0;
        char a[64] = "", b[64] = "";
        const int k = fscanf(f, "%63s %63s", a, b);
        fclose(f);
        if (k < 1 || !strcmp(a, "max") || atof(a) <= 0) return 0;
        double period = k == 2 ? atof(b) : 0;
        if (path_period) {
            FILE *g = fopen(path_period, "r");
            if (g) { if (fscanf(g, "%63s", b) == 1) period = atof(b); fclose(g); }
        }
        return period > 0 ? atof(a) / period : 0;
    };
    double q = quota("/sys/fs/cgroup/cpu.max", nullptr);                                             // cgroup v2
    if (q <= 0) q = quota("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us");   // v1
    if (q > 0) n = std::min(n, std::max(1, (int)(q + 0.5)));
    return n > 0 ? n : 1;
}

// SNK_TIMING=1: where the wall clock of the host pipeline goes (seconds per stage, printed at the end)
struct StageClock {
    std::atomic<long long> ns[12];
    const char *name[12] = {"reader: inflate/copy", "reader: newline index", "reader: push wait", "main: input wait", "main: slot wait",
                            "main: pack", "main: submit", "writer: slot wait", "writer: gpu wait", "writer: format+deflate", "writer: write", "other"};
    bool on = false;
    StageClock() { for (auto &x : ns) x = 0; }
    static long long now() { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return (long long)t.tv_sec * 1000000000ll + t.tv_nsec; }
    void report() const {
        if (!on) return;
        for (int i = 0; i < 11; ++i) fprintf(stderr, "[timing] %-24s %8.3f s\n", name[i], (double)ns[i].load() * 1e-9);
    }
};
StageClock g_clk;
struct Tick {
    int k; long long t0;
    explicit Tick(int k_) : k(k_), t0(g_clk.on ? StageClock::now() : 0) {}
    ~Tick() { if (g_clk.on) g_clk.ns[k] += StageClock::now() - t0; }
};

void parallel_for(int workers, int n, const std::function<void(int, int, int)> &f) {   // f(worker, lo, hi)
    workers = std::max(1, std::min(workers, n));
    if (workers == 1) { f(0, 0, n); return; }
    std::atomic<int> left(workers - 1);
    for (int w = 1; w < workers; ++w) {
        const int lo = (int)((long)n * w / workers), hi = (int)((long)n * (w + 1) / workers);
        g_pool.push([&f, &left, w, lo, hi] { f(w, lo, hi); left.fetch_sub(1, std::memory_order_release); });
    }
    f(0, 0, (int)((long)n / workers));
    while (left.load(std::memory_order_acquire) > 0)
        if (!g_pool.try_run_one()) std::this_thread::yield();
}

// n whole FASTQ records of one file as they were read, plus the index of their 4n lines
struct RawChunk {
    const char *base = nullptr;                        // into `own` (inflated data) or into the mmap of a plain file
    char *own = nullptr;                               // uninitialised storage (no zero fill of 100+ MB per chunk)
    size_t own_cap = 0;
    ~RawChunk() { free(own); }
    void reserve(size_t n) {
        if (n <= own_cap) return;
        own = (char *)realloc(own, n);
        if (!own) { fprintf(stderr, "Error:out of memory\n"); exit(1); }
        own_cap = n;
    }
    std::vector<uint32_t> ls, le;                      // line start / end (end excludes the line terminator); empty in count mode
    int n = 0;
    size_t nbytes = 0;                                 // text of the n records: base[0, nbytes)
    const char *line(int k, int &len) const { len = (int)(le[k] - ls[k]); return base + ls[k]; }
    // chunks are recycled: a fresh 100 MB buffer per batch means 100 MB of page faults per batch (and one mmap lock
    // for all threads of the process)
    static std::mutex &pool_mutex() { static std::mutex m; return m; }
    static std::vector<RawChunk *> &pool() { static std::vector<RawChunk *> p; return p; }
    static RawChunk *get() {
        std::lock_guard<std::mutex> l(pool_mutex());
        if (pool().empty()) return new RawChunk;
        RawChunk *c = pool().back();
        pool().pop_back();
        return c;
    }
    static void put(RawChunk *c) {
        if (!c) return;
        c->ls.clear(); c->le.clear(); c->n = 0; c->base = nullptr; c->nbytes = 0;
        std::lock_guard<std::mutex> l(pool_mutex());
        if (pool().size() < 16) pool().push_back(c); else delete c;
    }
};

// gz input (multi-member ok): inflate (snk_inflate.h, from the mapped file) + line index on one thread per
// file.  Every line loses its last `space_num` characters, the number of trailing white-space characters of
// the FIRST line of fq1 (src/peprocess.cpp:2066-2077: 1 for "\n", 2 for "\r\n", more with trailing blanks).
// Chunk buffers carry the last 32 KiB of the stream in front of their data (the inflate window).
// offsets (relative to p) of every '\n' in p[0, n), found by `workers` threads
void parallel_newlines(const char *p, size_t n, int workers, std::vector<uint32_t> &pos) {
    pos.clear();
    const int k = (int)std::max<size_t>(1, std::min<size_t>((size_t)workers, n / (1 << 20)));
    std::vector<std::vector<uint32_t>> part((size_t)k);
    parallel_for(k, k, [&](int, int lo, int hi) {
        for (int w = lo; w < hi; ++w) {
            const char *a = p + n * (size_t)w / (size_t)k, *e = p + n * (size_t)(w + 1) / (size_t)k;
            std::vector<uint32_t> &v = part[(size_t)w];
            v.reserve((size_t)(e - a) / 64 + 16);
            while (a < e && (a = (const char *)memchr(a, '\n', (size_t)(e - a)))) { v.push_back((uint32_t)(a - p)); ++a; }
        }
    });
    size_t tot = 0;
    for (auto &v : part) tot += v.size();
    pos.reserve(tot);
    for (auto &v : part) pos.insert(pos.end(), v.begin(), v.end());
}

// ---- count mode (the device parses the text, SURVEY 8f N2 / include/snk_fastq.h): the readers only have to cut the stream
// behind every `batch` records, i.e. behind 4 * batch line ends -- newlines are counted (SIMD compare + popcount), not indexed
__attribute__((target("avx2"))) size_t count_nl_avx2(const char *p, size_t n) {
    const __m256i nl = _mm256_set1_epi8('\n');
    size_t c = 0, i = 0;
    for (; i + 128 <= n; i += 128) {
        const unsigned a = (unsigned)_mm256_movemask_epi8(_mm256_cmpeq_epi8(_mm256_loadu_si256((const __m256i *)(p + i)), nl));
        const unsigned b = (unsigned)_mm256_movemask_epi8(_mm256_cmpeq_epi8(_mm256_loadu_si256((const __m256i *)(p + i + 32)), nl));
        const unsigned d = (unsigned)_mm256_movemask_epi8(_mm256_cmpeq_epi8(_mm256_loadu_si256((const __m256i *)(p + i + 64)), nl));
        const unsigned e = (unsigned)_mm256_movemask_epi8(_mm256_cmpeq_epi8(_mm256_loadu_si256((const __m256i *)(p + i + 96)), nl));
        c += (size_t)__builtin_popcountll((unsigned long long)a | ((unsigned long long)b << 32)) + (size_t)__builtin_popcountll((unsigned long long)d | ((unsigned long long)e << 32));
    }
    for (; i < n; ++i) c += p[i] == '\n';
    return c;
}
size_t count_nl(const char *p, size_t n) {
    static const bool avx2 = __builtin_cpu_supports("avx2");
    if (avx2) return count_nl_avx2(p, n);
    size_t c = 0;
    const __m128i nl = _mm_set1_epi8('\n');
    size_t i = 0;
    for (; i + 16 <= n; i += 16) c += (size_t)__builtin_popcount((unsigned)_mm_movemask_epi8(_mm_cmpeq_epi8(_mm_loadu_si128((const __m128i *)(p + i)), nl)));
    for (; i < n; ++i) c += p[i] == '\n';
    return c;
}
// offset just behind the k-th '\n' (k >= 1) of p[0, n); n when there are fewer
size_t after_kth_nl(const char *p, size_t n, size_t k) {
    const char *a = p, *e = p + n;
    while (k && a < e && (a = (const char *)memchr(a, '\n', (size_t)(e - a)))) { ++a; --k; }
    return (k || !a) ? n : (size_t)(a - p);
}
struct NlPiece { size_t begin, end, count; };           // offsets relative to the chunk's data
// newline counts of p[from, to) in pieces of 1 MB, appended to `pieces`; returns their sum
size_t count_pieces(const char *p, size_t from, size_t to, int workers, std::vector<NlPiece> &pieces) {
    const size_t step = (size_t)1 << 20, at = pieces.size();
    for (size_t a = from; a < to; a += step) pieces.push_back(NlPiece{a, std::min(to, a + step), 0});
    const int k = (int)(pieces.size() - at);
    parallel_for(workers, k, [&](int, int lo, int hi) {
        for (int w = lo; w < hi; ++w) { NlPiece &q = pieces[at + (size_t)w]; q.count = count_nl(p + q.begin, q.end - q.begin); }
    });
    size_t tot = 0;
    for (size_t w = at; w < pieces.size(); ++w) tot += pieces[w].count;
    return tot;
}
// offset behind the `want`-th newline, given the piece counts (their sum is >= want)
size_t locate_nl(const char *p, const std::vector<NlPiece> &pieces, size_t want) {
    size_t cum = 0;
    for (const NlPiece &q : pieces) {
        if (cum + q.count >= want) return q.begin + after_kth_nl(p + q.begin, q.end - q.begin, want - cum);
        cum += q.count;
    }
    return pieces.empty() ? 0 : pieces.back().end;
}
// line index of a count-mode chunk, built on the host when somebody needs it (the first batch: read lengths and the
// quality-system check)
void index_chunk(RawChunk *c, int space_num, int workers) {
    if (!c->ls.empty() || c->n == 0) return;
    std::vector<uint32_t> nlpos;
    parallel_newlines(c->base, c->nbytes, workers, nlpos);
    c->ls.reserve((size_t)c->n * 4);
    c->le.reserve((size_t)c->n * 4);
    size_t start = 0;
    for (uint32_t nl : nlpos) {
        const size_t e_incl = (size_t)nl + 1;
        c->ls.push_back((uint32_t)start);
        c->le.push_back((uint32_t)(e_incl > start + (size_t)space_num ? e_incl - (size_t)space_num : start));
        start = e_incl;
    }
    if (start < c->nbytes) { c->ls.push_back((uint32_t)start); c->le.push_back((uint32_t)c->nbytes); }    // last line without '\n'
    if (c->ls.size() != (size_t)c->n * 4) die("truncated fastq record");
}

// ---- .gz input decoded on the GPU (include/snk_gunzip.h, host/snk_dgunzip.h; SNK_DEVICE_INFLATE=1): the reader's source of bytes is
// either the host's parallel decoder or the device one -- same bytes, same errors
struct HipGunzipBackend : snk::DgBackend {
    snk_gunzip *g = nullptr;
    uint8_t *pinned[2] = {nullptr, nullptr};       // two text slots: one served to the parser while the next window is decoded
    size_t pinned_cap[2] = {0, 0};
    string err;
    int device_ = 0;
    HipGunzipBackend(int device, const snk::DeviceGunzip::Geometry &geo) : device_(device) {
        g = snk_gunzip_create(device, geo.window_bytes, geo.chunk_bytes, geo.syms_per_chunk, geo.ends_per_chunk);
        if (!g) err = snk_last_error();
    }
    ~HipGunzipBackend() override {
        for (int s = 0; s < 2; ++s) if (pinned[s]) (void)hipHostFree(pinned[s]);
        snk_gunzip_destroy(g);
    }
    bool decode(const uint8_t *comp, uint64_t nbytes, uint64_t first_bit, bool first_of_member, snk_gunzip_chunk *chunks, snk_gunzip_member *ends) override {
        if (!g) return false;
        if (snk_gunzip_decode(g, comp, nbytes, first_bit, first_of_member ? 1 : 0, chunks, ends) != SNK_OK) { err = snk_last_error(); return false; }
        return true;
    }
    bool resolve(const uint32_t *order, uint32_t k, const uint8_t *win_in, uint8_t *text, uint64_t text_bytes, uint8_t *win_out) override {
        if (snk_gunzip_resolve(g, order, k, win_in, text, text_bytes, win_out) != SNK_OK) { err = snk_last_error(); return false; }
        return true;
    }
    uint8_t *text_buffer(int slot, size_t bytes) override {
        if (bytes > pinned_cap[slot]) {
            if (pinned[slot]) (void)hipHostFree(pinned[slot]);
            pinned[slot] = nullptr;
            pinned_cap[slot] = 0;
            const size_t cap = bytes + bytes / 4;
            (void)hipSetDevice(device_);                 // (called on the decoder's producer thread)
            if (hipHostMalloc((void **)&pinned[slot], cap, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); err = "cannot pin the text buffer"; return nullptr; }
            pinned_cap[slot] = cap;
        }
        return pinned[slot];
    }
    std::string error() override { return err; }
};

struct GzSource {
    std::unique_ptr<snk::ParallelGunzip> host;
    std::unique_ptr<HipGunzipBackend> be;
    std::unique_ptr<snk::DeviceGunzip> dev;
    // a shard of a sharded run (g_shard.gz): the decoder starts at the shard's block, the first skip_b bytes and skip_nl lines
    // behind that belong to the shard in front, take_nl lines are this shard's
    bool limited = false, limit_done = false;
    uint64_t skip_b = 0, skip_nl = 0, take_nl = ~0ull;
    GzSource(const uint8_t *zin, size_t n, int workers, const string &path = string()) {
        uint64_t start_bit = ~0ull;
        const uint8_t *start_win = nullptr;
        auto it = g_shard.gz.find(path);
        if (it != g_shard.gz.end()) {
            limited = true;
            skip_b = it->second.skip_bytes;
            skip_nl = it->second.skip_records * 4;
            take_nl = it->second.nrec == ~0ull ? ~0ull : it->second.nrec * 4;
            if (it->second.bit != ~0ull) { start_bit = it->second.bit; start_win = it->second.win.data(); }
        }
        if (const char *e = getenv("SNK_DEVICE_INFLATE")) {
            if (atoi(e) != 0) {
                snk::DeviceGunzip::Geometry geo;
                geo.window_bytes = (uint64_t)(getenv("SNK_DGZ_WINDOW_MB") ? atol(getenv("SNK_DGZ_WINDOW_MB")) : 128) << 20;
                geo.chunk_bytes = (uint32_t)(getenv("SNK_DGZ_CHUNK_KB") ? atol(getenv("SNK_DGZ_CHUNK_KB")) : 128) << 10;
                geo.syms_per_chunk = geo.chunk_bytes * (uint32_t)(getenv("SNK_DGZ_RATIO") ? atol(getenv("SNK_DGZ_RATIO")) : 12);
                geo.ends_per_chunk = geo.chunk_bytes / 2048 + 4;
                if (geo.window_bytes > n + geo.chunk_bytes) geo.window_bytes = ((n + geo.chunk_bytes) / geo.chunk_bytes) * geo.chunk_bytes;
                int device = 0;
                (void)hipGetDevice(&device);
                be.reset(new HipGunzipBackend(device, geo));
                if (be->g) dev.reset(new snk::DeviceGunzip(zin, n, be.get(), geo, std::max(1, workers), start_bit, start_win));
                else cerr << "Warning:device inflate is not available (" << be->err << "), decoding on the host" << endl;
            }
        }
        if (!dev) {
            size_t pg_chunk = (size_t)2 << 20;
            if (const char *e = getenv("SNK_GZ_CHUNK")) { const long v = atol(e); if (v >= 65536) pg_chunk = (size_t)v; }
            host.reset(new snk::ParallelGunzip(zin, n, workers, pg_chunk, start_bit, start_win));
        }
    }
    size_t raw_run(uint8_t *out, size_t cap) { return dev ? dev->run(out, cap) : host->run(out, cap); }
    size_t run(uint8_t *out, size_t cap) {
        if (!limited) return raw_run(out, cap);
        while (!limit_done) {
            const size_t got = raw_run(out, cap);
            if (got == 0) return 0;                          // the end of the stream, or an error
            size_t from = 0;
            if (skip_b) { const size_t k = (size_t)std::min<uint64_t>(skip_b, got); from = k; skip_b -= k; }
            while (skip_nl && from < got) {
                const void *q = memchr(out + from, '\n', got - from);
                if (!q) { from = got; break; }
                from = (size_t)((const uint8_t *)q - out) + 1;
                --skip_nl;
            }
            if (from == got) continue;
            size_t end = got;
            if (take_nl != ~0ull)
                for (size_t q = from; q < got;) {
                    const void *nl = memchr(out + q, '\n', got - q);
                    if (!nl) break;
                    q = (size_t)((const uint8_t *)nl - out) + 1;
                    if (--take_nl == 0) { end = q; limit_done = true; break; }
                }
            if (from) memmove(out, out + from, end - from);
            if (end > from) return end - from;
        }
        return 0;
    }
    const char *error() const { return dev ? dev->error() : host->error(); }
    bool done() const { return limit_done || (dev ? dev->done() : host->done()); }
};

void reader_gz_count(const string path, int batch, int workers, Channel<RawChunk *> *out) {
    const int fd = open(path.c_str(), O_RDONLY);
    if (fd < 0) die("cannot open file," + path);
    struct stat st;
    fstat(fd, &st);
    const uint8_t *zin = (const uint8_t *)mmap(NULL, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
    if (zin == MAP_FAILED) die("cannot map file," + path);
    madvise((void *)zin, (size_t)st.st_size, MADV_SEQUENTIAL);
    // (owned by a pointer: a shard stops reading at its last record while the decoder's threads are still ahead of it in the
    // mapping -- they have to be gone before the mapping is)
    std::unique_ptr<GzSource> zp(new GzSource(zin, (size_t)st.st_size, workers, path));
    GzSource &z = *zp;
    const size_t H = snk::GzipInflate::HIST, block = (size_t)1 << 24;
    RawChunk *cur = RawChunk::get();
    const size_t cap0 = H + (size_t)batch * 400 + 2 * block;
    cur->reserve(cap0);
    memset(cur->own, 0, H);
    size_t fill = 0, counted = 0, nl_total = 0;
    const size_t want = (size_t)batch * 4;
    std::vector<NlPiece> pieces;
    bool eof = false;
    for (;;) {
        char *data = cur->own + H;
        if (counted < fill && nl_total < want) {
            Tick t_(1);
            nl_total += count_pieces(data, counted, fill, workers, pieces);
            counted = fill;
        }
        if (nl_total >= want) {                          // a full batch: hand it over, keep the unread tail + window
            const size_t end_off = locate_nl(data, pieces, want);
            RawChunk *next = RawChunk::get();
            next->reserve(std::max(cur->own_cap, cap0));
            const size_t left = fill - end_off;
            memcpy(next->own, data + end_off - H, H + left);
            cur->n = batch;
            cur->base = data;
            cur->nbytes = end_off;
            { Tick t_(2); out->push(cur); }
            cur = next;
            fill = left;
            counted = 0;
            nl_total = 0;
            pieces.clear();
            continue;
        }
        if (eof) {
            const size_t lines = nl_total + ((fill > 0 && data[fill - 1] != '\n') ? 1 : 0);
            if (lines % 4) die("truncated fastq record");
            cur->n = (int)(lines / 4);
            cur->base = data;
            cur->nbytes = fill;
            if (cur->n) out->push(cur); else RawChunk::put(cur);
            break;
        }
        if (H + fill + block > cur->own_cap) { cur->reserve(cur->own_cap * 2); data = cur->own + H; }
        size_t got;
        { Tick t_(0); got = z.run((uint8_t *)data + fill, block); }
        if (z.error()) die(string("read error in input fastq (") + z.error() + ")," + path);
        if (got == 0 && z.done()) eof = true;
        fill += got;
        if (fill > 0xF0000000ull) die("batch larger than 4 GB: lower the batch size");
    }
    zp.reset();
    munmap((void *)zin, (size_t)st.st_size);
    close(fd);
    out->close();
}

void reader_plain_count(const string path, int batch, int workers, Channel<RawChunk *> *out) {
    const int fd = open(path.c_str(), O_RDONLY);
    if (fd < 0) die("cannot open file," + path);
    struct stat st;
    fstat(fd, &st);
    size_t size = (size_t)st.st_size;
    const char *base = (const char *)mmap(NULL, size, PROT_READ, MAP_PRIVATE, fd, 0);
    if (base == MAP_FAILED) die("cannot map file," + path);
    madvise((void *)base, size, MADV_SEQUENTIAL);
    const size_t want = (size_t)batch * 4;
    size_t pos = 0;
    {   // a shard of a sharded run (main: sharded ingest): whole records [lo, hi) of this file
        auto it = g_shard.range.find(path);
        if (it != g_shard.range.end()) { pos = it->second.first; size = std::min(size, it->second.second); }
    }
    double per_line = 100.0;
    std::vector<NlPiece> pieces;
    while (pos < size) {
        pieces.clear();
        const char *p = base + pos;
        size_t win = 0, total = 0;                        // window [pos, pos + win) counted so far
        while (total < want && pos + win < size) {
            const size_t need = want - total;
            const size_t ext = std::min(size - pos, win + (size_t)(per_line * 1.02 * (double)need) + 65536);
            if (ext > 0xF0000000ull) die("batch larger than 4 GB: lower the batch size");
            { Tick t_(1); total += count_pieces(p, win, ext, workers, pieces); }
            win = ext;
            if (total) per_line = (double)win / (double)total;
        }
        RawChunk *c = RawChunk::get();
        c->base = p;
        if (total >= want) {
            c->nbytes = locate_nl(p, pieces, want);
            c->n = batch;
        } else {                                         // end of file (the last line may lack its '\n')
            const size_t lines = total + ((win > 0 && p[win - 1] != '\n') ? 1 : 0);
            if (lines % 4) die("truncated fastq record");
            c->nbytes = win;
            c->n = (int)(lines / 4);
        }
        pos += c->nbytes;
        if (c->n) { Tick t_(2); out->push(c); } else { RawChunk::put(c); break; }
    }
    out->close();                                        // the mapping stays until exit: chunks point into it
    close(fd);
}

void reader_gz(const string path, int batch, int space_num, int workers, Channel<RawChunk *> *out) {
    const int fd = open(path.c_str(), O_RDONLY);
    if (fd < 0) die("cannot open file," + path);
    struct stat st;
    fstat(fd, &st);
    const uint8_t *zin = (const uint8_t *)mmap(NULL, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
    if (zin == MAP_FAILED) die("cannot map file," + path);
    madvise((void *)zin, (size_t)st.st_size, MADV_SEQUENTIAL);
    // One gzip stream, decoded by `workers` threads (snk_pgunzip.h: block-start search, marker decoding, window
    // resolution; CRC-32 and ISIZE of every member are checked there).  The reference decodes with one zlib stream
    // per reading thread (src/peprocess.cpp:2089-2113).
    size_t pg_chunk = (size_t)2 << 20;
    if (const char *e = getenv("SNK_GZ_CHUNK")) { const long v = atol(e); if (v >= 65536) pg_chunk = (size_t)v; }
    (void)pg_chunk;
    std::unique_ptr<GzSource> zp(new GzSource(zin, (size_t)st.st_size, workers, path));     // (the host's parallel decoder, or the device one with SNK_DEVICE_INFLATE=1; a shard: its part)
    GzSource &z = *zp;
    auto crc_drain = [] {};
    const size_t H = snk::GzipInflate::HIST, block = (size_t)1 << 24;
    RawChunk *cur = RawChunk::get();
    const size_t cap0 = H + (size_t)batch * 400 + 2 * block;      // a whole batch of ~150 bp records without regrowth
    cur->reserve(cap0);
    memset(cur->own, 0, H);
    size_t fill = 0, scan = 0, line_start = 0;          // relative to the data area (own.data() + H)
    bool eof = false;
    const size_t want = (size_t)batch * 4;
    std::vector<uint32_t> nlpos;
    auto add_line = [&](size_t s, size_t e_incl_nl) {
        size_t e = e_incl_nl > s + (size_t)space_num ? e_incl_nl - (size_t)space_num : s;
        cur->ls.push_back((uint32_t)s);
        cur->le.push_back((uint32_t)e);
    };
    for (;;) {
        char *data = cur->own + H;
        if (scan < fill && cur->ls.size() < want) {     // index the new text (the inflate thread only waits for the scan)
            { Tick t_(1); parallel_newlines(data + scan, fill - scan, workers, nlpos); }
            const size_t base_off = scan;
            scan = fill;
            for (uint32_t rel : nlpos) {
                const size_t nl = base_off + rel;
                add_line(line_start, nl + 1);
                line_start = nl + 1;
                if (cur->ls.size() == want) break;      // the rest is indexed again as part of the next chunk
            }
        }
        if (cur->ls.size() == want) {                  // a full batch: hand it over, keep the unread tail + window
            RawChunk *next = RawChunk::get();
            next->reserve(std::max(cur->own_cap, cap0));
            const size_t left = fill - line_start;
            memcpy(next->own, data + line_start - H, H + left);   // (reaches into cur's own window area: valid)
            cur->n = batch;
            cur->base = data;
            crc_drain();
            { Tick t_(2); out->push(cur); }
            cur = next;
            fill = left;
            scan = 0;
            line_start = 0;
            continue;
        }
        if (eof) {
            if (line_start < fill) add_line(line_start, fill + (size_t)space_num);       // last line without '\n'
            if (cur->ls.size() % 4) die("truncated fastq record");
            cur->n = (int)(cur->ls.size() / 4);
            cur->base = data;
            crc_drain();
            if (cur->n) out->push(cur); else RawChunk::put(cur);
            break;
        }
        if (H + fill + block > cur->own_cap) { crc_drain(); cur->reserve(cur->own_cap * 2); data = cur->own + H; }
        size_t got;
        { Tick t_(0); got = z.run((uint8_t *)data + fill, block); }
        if (z.error()) die(string("read error in input fastq (") + z.error() + ")," + path);
        if (got == 0 && z.done()) eof = true;
        fill += got;
        if (fill > 0xF0000000ull) die("batch larger than 4 GB: lower the batch size");
    }
    zp.reset();
    munmap((void *)zin, (size_t)st.st_size);
    close(fd);
    out->close();
}

// plain input: the file is mapped, chunks point into the mapping (no copy), and the newline index of a
// window is built by `workers` threads in parallel
void reader_plain(const string path, int batch, int space_num, int workers, Channel<RawChunk *> *out) {
    const int fd = open(path.c_str(), O_RDONLY);
    if (fd < 0) die("cannot open file," + path);
    struct stat st;
    fstat(fd, &st);
    size_t size = (size_t)st.st_size;
    const char *base = (const char *)mmap(NULL, size, PROT_READ, MAP_PRIVATE, fd, 0);
    if (base == MAP_FAILED) die("cannot map file," + path);
    madvise((void *)base, size, MADV_SEQUENTIAL);
    const size_t want = (size_t)batch * 4;
    size_t pos = 0;
    {   // a shard of a sharded run: whole records [lo, hi) of this file (as reader_plain_count)
        auto it = g_shard.range.find(path);
        if (it != g_shard.range.end()) { pos = it->second.first; size = std::min(size, it->second.second); }
    }
    double per_line = 100.0;
    while (pos < size) {
        std::vector<std::vector<uint32_t>> found;
        size_t win_end = pos, total = 0;
        // grow the window until it holds `want` line ends (or the file ends); only the extension is scanned
        while (total < want && win_end < size) {
            const size_t need = want - total;
            size_t ext_end = std::min(size, win_end + (size_t)(per_line * 1.03 * (double)need) + 65536);
            if (ext_end - pos > 0xF0000000ull) die("batch larger than 4 GB: lower the batch size");
            const size_t a = win_end, len = ext_end - win_end;
            const int k = (int)std::max<size_t>(1, std::min<size_t>((size_t)workers, len / (1 << 20)));
            const size_t at = found.size();
            found.resize(at + (size_t)k);
            parallel_for(k, k, [&](int, int lo, int hi) {
                for (int w = lo; w < hi; ++w) {
                    const size_t s0 = a + len * (size_t)w / (size_t)k, s1 = a + len * (size_t)(w + 1) / (size_t)k;
                    std::vector<uint32_t> &v = found[at + (size_t)w];
                    v.reserve((size_t)((double)(s1 - s0) / per_line * 1.2) + 16);
                    const char *p = base + s0, *e = base + s1;
                    while (p < e && (p = (const char *)memchr(p, '\n', (size_t)(e - p)))) { v.push_back((uint32_t)(p - (base + pos))); ++p; }
                }
            });
            for (size_t w = at; w < found.size(); ++w) total += found[w].size();
            win_end = ext_end;
            if (total) per_line = (double)(win_end - pos) / (double)total;
        }
        RawChunk *c = RawChunk::get();
        c->base = base + pos;
        const size_t take = std::min(total, want);
        c->ls.reserve(take + 1);
        c->le.reserve(take + 1);
        uint32_t start = 0;
        size_t got = 0;
        for (const auto &v : found) {
            for (uint32_t nl : v) {
                if (got == take) break;
                const uint32_t e_incl = nl + 1;
                c->ls.push_back(start);
                c->le.push_back(e_incl > start + (uint32_t)space_num ? e_incl - (uint32_t)space_num : start);
                start = e_incl;
                ++got;
            }
            if (got == take) break;
        }
        size_t consumed = start;
        if (total < want && pos + start < size) {        // end of file, last line without '\n'
            c->ls.push_back(start);
            c->le.push_back((uint32_t)(size - pos));
            consumed = size - pos;
        }
        if (c->ls.size() % 4) die("truncated fastq record");
        c->n = (int)(c->ls.size() / 4);
        pos += consumed;
        if (c->n) out->push(c); else { RawChunk::put(c); break; }
    }
    out->close();                                        // the mapping stays until exit: chunks point into it
    close(fd);
}

bool is_gzip_file(const string &path) {
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) die("cannot open file," + path);
    unsigned char m[2] = {0, 0};
    const size_t n = fread(m, 1, 2, f);
    fclose(f);
    return n == 2 && m[0] == 0x1f && m[1] == 0x8b;
}

void reader_main(const string path, int batch, int space_num, int workers, bool index, Channel<RawChunk *> *out) {
    struct stat st;
    if (stat(path.c_str(), &st) != 0 || st.st_size == 0) die("cannot open file or empty file," + path);
    if (is_gzip_file(path)) { if (index) reader_gz(path, batch, space_num, workers, out); else reader_gz_count(path, batch, workers, out); }
    else { if (index) reader_plain(path, batch, space_num, workers, out); else reader_plain_count(path, batch, workers, out); }
}

int first_line_space_num(const string &path) {          // src/peprocess.cpp:2066-2077
    gzFile f = gzopen(path.c_str(), "rb");
    if (!f) die("cannot open file," + path);
    char buf[1000];                                      // READBUF
    int sp = 0;
    if (gzgets(f, buf, sizeof buf) != NULL) {
        int n = (int)strlen(buf);
        while (n > 0 && isspace((unsigned char)buf[n - 1])) { ++sp; --n; }
    }
    gzclose(f);
    return sp > 0 ? sp : 1;
}

#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) die(string(#x) + ": " + hipGetErrorString(e_)); } while (0)

// one gzip member of `in`, appended to `out`.  The reference writes level-2 zlib streams (src/peprocess.cpp:1809); the
// compressed bytes are not part of the contract, so the members come from this repo's own encoder (snk_deflate.h: the
// same ratio on FASTQ at about twice the speed), or from zlib level 2 with SNK_ZLIB_OUT=1.
void gzip_member(const char *in_p, size_t in_n, string &out) {
    static const bool use_zlib = getenv("SNK_ZLIB_OUT") != nullptr;
    if (!use_zlib) {
        thread_local snk::FastDeflate enc;
        enc.gzip_member(reinterpret_cast<const uint8_t *>(in_p), in_n, out, true);     // (every text this CLI compresses is FASTQ records)
        return;
    }
    const string in(in_p, in_n);
    z_stream z;
    memset(&z, 0, sizeof z);
    if (deflateInit2(&z, 2, Z_DEFLATED, 15 + 16, 8, Z_DEFAULT_STRATEGY) != Z_OK) die("deflateInit2 failed");
    const size_t bound = deflateBound(&z, (uLong)in.size()) + 64;
    const size_t at = out.size();
    out.resize(at + bound);
    z.next_in = (Bytef *)in.data();
    z.avail_in = (uInt)in.size();
    z.next_out = (Bytef *)&out[at];
    z.avail_out = (uInt)bound;
    if (deflate(&z, Z_FINISH) != Z_STREAM_END) die("deflate failed");
    out.resize(at + (bound - z.avail_out));
    deflateEnd(&z);
}
void gzip_member(const string &in, string &out) { gzip_member(in.data(), in.size(), out); }

struct OutFile {                                        // clean / dup output: bytes are produced by the workers
    int fd = -1;
    off_t pos = 0;
    bool gz = false;
    bool is_open() const { return fd >= 0; }
    // an output that is not a regular file (a named pipe into `md5sum`, /dev/stdout ...) cannot be written at offsets: its
    // pieces go out in order through write()
    static bool &is_stream(int fd) { static bool t[4096]; return t[fd >= 0 && fd < 4096 ? fd : 0]; }
    void open(const string &path, bool gzip) {
        fd = ::open(path.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644);
        if (fd < 0) die("cannot write to the file," + path);
        if (fd >= 4096) die("too many open files");
        struct stat st;
        is_stream(fd) = fstat(fd, &st) == 0 && !S_ISREG(st.st_mode);
        pos = 0;
        gz = gzip;
    }
    static void put_at(int fd, const char *p, size_t n, off_t at) {
        while (n) {
            const ssize_t w = is_stream(fd) ? ::write(fd, p, n) : pwrite(fd, p, n, at);
            if (w <= 0) die("write error (disk full?)");
            p += w; n -= (size_t)w; at += w;
        }
    }
    void write_bytes(const string &b) {
        if (b.empty()) return;
        put_at(fd, b.data(), b.size(), pos);
        pos += (off_t)b.size();
    }
    // the pieces of one batch, in order: every piece goes to its own offset, written by the pool's threads
    void write_parts(const std::vector<string> &parts) {
        std::vector<off_t> at(parts.size());
        off_t p = pos;
        size_t total = 0;
        for (size_t i = 0; i < parts.size(); ++i) { at[i] = p; p += (off_t)parts[i].size(); total += parts[i].size(); }
        if (total < ((size_t)8 << 20) || parts.size() < 2 || is_stream(fd)) {
            for (size_t i = 0; i < parts.size(); ++i) if (!parts[i].empty()) put_at(fd, parts[i].data(), parts[i].size(), at[i]);
        } else {
            const int fdc = fd;
            parallel_for(std::min<int>(8, (int)parts.size()), (int)parts.size(), [&](int, int lo, int hi) {     // (tmpfs page allocation does not scale past a few writers)
                for (int i = lo; i < hi; ++i) if (!parts[(size_t)i].empty()) put_at(fdc, parts[(size_t)i].data(), parts[(size_t)i].size(), at[(size_t)i]);
            });
        }
        pos = p;
    }
    void write_text(const string &text) {               // serial path (dup side files): compress here if needed
        if (text.empty()) return;
        if (!gz) { write_bytes(text); return; }
        string z;
        gzip_member(text, z);
        write_bytes(z);
    }
    void close() {
        if (fd < 0) return;
        if (gz && pos == 0) { string z; gzip_member(string(), z); write_bytes(z); }   // valid empty .gz
        ::close(fd);
        fd = -1;
    }
};

// totalReadsNum without "head" (src/peprocess.cpp:3198-3320, run_extract_random / sub_extract): after the run
// every k-th clean read (k = clean reads / wanted reads, integer) until the wanted number is reached goes to a new
// file that takes the clean file's name; the complete file is kept as total.<cleanFq>.
void extract_every_kth(const Options &o, int mates, uint64_t total_clean) {
    unsigned long long want = o.total_num;
    if (o.total_ratio > 0) {
        if (o.total_ratio >= 1) die("the ratio extract from clean fq file should not be more than 1");
        want = (unsigned long long)((float)total_clean * o.total_ratio);
    }
    if (total_clean < want) {
        cerr << "Warning:the reads number in clean fastq file(" << total_clean << ") is less than you assigned to output(" << want << ")" << endl;
        return;
    }
    if (want == 0) { cerr << "Error:assigned reads number should not be 0" << endl; return; }
    if ((float)total_clean / want < 1.1) return;
    const int mo = (int)(total_clean / want);
    const string names[2] = {o.clean1, o.clean2};
    for (int m = 0; m < mates; ++m) {
        const string in = o.out_dir + "/" + names[m];
        const string out = o.out_dir + (o.out_gz ? "/cleanRandomExtractReads.r" : "/cleanRandomExtractReads.r") + std::to_string(m + 1) + (o.out_gz ? ".fq.gz" : ".fq");
        char buf[1000];                                           // READBUF
        long line_num = 0, kept_lines = 0;
        if (o.out_gz) {
            gzFile fo = gzopen(out.c_str(), "wb"), fi = gzopen(in.c_str(), "rb");
            if (!fo || !fi) die("cannot open such file," + in);
            gzsetparams(fo, 2, Z_DEFAULT_STRATEGY);
            while (gzgets(fi, buf, sizeof buf)) {
                if (line_num % (4 * mo) <= 3) {
                    gzwrite(fo, buf, (unsigned)strlen(buf));
                    ++kept_lines;
                    if ((unsigned long long)(kept_lines / 4) >= want && kept_lines % 4 == 0) break;
                }
                ++line_num;
            }
            gzclose(fo);
            gzclose(fi);
        } else {
            FILE *fo = fopen(out.c_str(), "w"), *fi = fopen(in.c_str(), "r");
            if (!fo || !fi) die("cannot open such file," + in);
            while (fgets(buf, sizeof buf, fi)) {
                if (line_num % (4 * mo) <= 3) {
                    fputs(buf, fo);
                    ++kept_lines;
                    if ((unsigned long long)(kept_lines / 4) >= want && kept_lines % 4 == 0) break;
                }
                ++line_num;
            }
            fclose(fo);
            fclose(fi);
        }
        if (rename(in.c_str(), (o.out_dir + "/total." + names[m]).c_str()) != 0 || rename(out.c_str(), in.c_str()) != 0)
            die("cannot rename the extracted clean file," + in);
    }
}

struct Slot {                                           // one batch in flight
    uint8_t *h_seq[2] = {nullptr, nullptr}, *h_qual[2] = {nullptr, nullptr}, *d_seq[2] = {nullptr, nullptr}, *d_qual[2] = {nullptr, nullptr};
    uint16_t *h_len[2] = {nullptr, nullptr}, *d_len[2] = {nullptr, nullptr};
    snk_read_result *h_rec[2] = {nullptr, nullptr}, *d_rec[2] = {nullptr, nullptr};
    uint8_t *h_flags = nullptr, *d_flags = nullptr;      // per-pair host verdicts (tile / fov bits, + the duplicate bit)
    std::vector<uint64_t> snap_sum, snap_max;            // -j: the owning virtual thread's statistics right after this patch
    uint64_t *h_err = nullptr;                           // the context's error word after this batch (pinned)
    hipStream_t stream = nullptr;
    hipEvent_t done = nullptr;
    RawChunk *raw[2] = {nullptr, nullptr};
    int n = 0, dev = 0, lcap = 0;                        // records; index into the device list; capacity it was packed with
    uint64_t first = 0;
    // device-text mode (include/snk_fastq.h): the batch's raw text travels, the device parses and formats.  h_text doubles
    // as the landing buffer of the clean text (the upload is long done by then, and the clean text is never larger)
    uint8_t *h_text[2] = {nullptr, nullptr}, *d_text[2] = {nullptr, nullptr}, *d_out[2] = {nullptr, nullptr};
    uint32_t *d_line[2] = {nullptr, nullptr}, *d_outoff[2] = {nullptr, nullptr}, *h_outoff[2] = {nullptr, nullptr};
    uint32_t *d_status[2] = {nullptr, nullptr}, *h_status[2] = {nullptr, nullptr};
    void *d_tmp = nullptr;
    size_t tmp_bytes = 0, nbytes[2] = {0, 0};
    // .gz output compressed on the device (snk_fastq_deflate_device): members of the clean text, their total size
    uint8_t *d_gz[2] = {nullptr, nullptr};
    uint32_t *d_gzinfo[2] = {nullptr, nullptr}, *h_gzinfo[2] = {nullptr, nullptr};
    void *d_ztmp = nullptr;
    size_t ztmp_bytes = 0;
    hipEvent_t parsed = nullptr;
    // one-pass rmdup: the batch's hashes, and the raw text of its duplicate pairs for the dupReads side files
    uint64_t *d_hash = nullptr;
    // ... on a device other than the table's: hashes over to the table's device, flags back (8 + 1 bytes per pair)
    uint64_t *d_hash0 = nullptr;
    uint8_t *d_flags0 = nullptr;
    hipEvent_t hashed = nullptr, marked = nullptr;
    uint8_t *d_dupout[2] = {nullptr, nullptr};
    uint32_t *d_dupoff[2] = {nullptr, nullptr}, *h_dupoff[2] = {nullptr, nullptr};
};

// One-shot sanity check of the quality system on the first patch of fq1 (stat_pe_fqs / stat_se_fqs run it once,
// on the first patch that reaches them: src/peprocess.cpp:1207-1319, src/seprocess.cpp:745-852): the qualities are
// scored under qualSys and under the other system; a clearly better fit of the other one is an error, a
// slightly better one a warning.  Same messages, same exit code.
void phred_sanity(const Options &o, const RawChunk &fq1) {
    const int64_t patch = o.patch_size > 0 ? o.patch_size : (int64_t)o.threads * 20000 / 8;
    const int n = (int)std::min<int64_t>(fq1.n, patch);
    const int phred = o.p.quality_phred, other = phred == 64 ? 33 : 64, mbq = o.p.max_base_quality;
    int ex[2] = {0, 0}, normal[2] = {0, 0}, meanq[2] = {0, 0};      // (int accumulators, as there)
    int64_t bases = 0;
    for (int i = 0; i < n; ++i) {
        int ls, lq;
        fq1.line(4 * i + 1, ls);
        const char *ql = fq1.line(4 * i + 3, lq);
        bases += ls;
        for (int k = 0; k < lq; ++k)
            for (int w = 0; w < 2; ++w) {
                const int bq = (int)ql[k] - (w ? other : phred);
                meanq[w] += bq;
                if (bq >= 0 && bq <= mbq) ++normal[w];
                else if (bq < -10 || bq > mbq + 10) ++ex[w];
            }
    }
    if (bases == 0) die("no data");
    const float ratio[2] = {(float)normal[0] / bases, (float)normal[1] / bases}, mean[2] = {(float)meanq[0] / bases, (float)meanq[1] / bases};
    int score[2];
    for (int w = 0; w < 2; ++w) {
        score[w] = ex[w] ? 0 : 1;
        score[w] += (ratio[w] > ratio[1 - w] || ratio[w] == ratio[1 - w]) ? 3 : 0;
        score[w] += (mean[w] < 10 || mean[w] > mbq) ? 0 : 2;
    }
    if (score[0] - score[1] < -3) die("base quality seems abnormal,please check the quality system parameter or fastq file");
    if (score[0] - score[1] < 0) cerr << "Warning:base quality seems abnormal,please check the quality system parameter or fastq file" << endl;
}

// statistics block of capacity `ls` added into one of capacity `ld` >= ls (include/snk_filter.h layout:
// fs | 4 files x (gs | bs[L][5] | qs[L][nq] | ts)): the per-position tables grow, everything else is a plain sum
void widen_add(const uint64_t *src, int ls, uint64_t *dst, int ld, int nq) {
    if (ls == ld) { for (int64_t k = 0, n = snk_stats_u64(ls, nq); k < n; ++k) dst[k] += src[k]; return; }
    for (int k = 0; k < SNK_FS_N; ++k) dst[k] += src[k];
    for (int f = 0; f < 4; ++f) {
        const uint64_t *s = src + snk_file_off(ls, nq, f);
        uint64_t *d = dst + snk_file_off(ld, nq, f);
        for (int k = 0; k < SNK_GS_N; ++k) d[k] += s[k];
        for (int64_t k = 0; k < (int64_t)ls * 5; ++k) d[snk_bs_off(ld, nq) + k] += s[snk_bs_off(ls, nq) + k];
        for (int64_t k = 0; k < (int64_t)ls * nq; ++k) d[snk_qs_off(ld, nq) + k] += s[snk_qs_off(ls, nq) + k];
        const int64_t nts = snk_file_block_u64(ls, nq) - snk_ts_off(ls, nq);
        for (int64_t k = 0; k < nts; ++k) d[snk_ts_off(ld, nq) + k] += s[snk_ts_off(ls, nq) + k];
    }
}

// The path's one collective for a single host process driving several GPUs: one RCCL communicator per
// device (ncclCommInitAll), one host thread per device, every virtual thread's block sum/max all-reduced in one
// group.  ctx_of(g, t) binds block t on device g and returns that device's context.  False when RCCL cannot span the
// list (the same device twice -- a test configuration): the caller then adds the blocks up on the host.
bool rccl_allreduce_blocks(const std::vector<int> &devices, const std::function<snk_ctx *(int, int)> &ctx_of, int T) {
    const int G = (int)devices.size();
    for (int a = 0; a < G; ++a) for (int b = a + 1; b < G; ++b) if (devices[(size_t)a] == devices[(size_t)b]) return false;
    // any failure in here is a warning: the caller holds host copies of every block and adds them up itself
    auto give_up = [](const string &why) { cerr << "Warning:stats all-reduce over RCCL unavailable (" << why << "), merging on the host" << endl; return false; };
    void *h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) return give_up(string("cannot load RCCL: ") + dlerror());
    typedef int (*init_all_fn)(void **, int, const int *);
    typedef int (*comm_fn)(void *);
    typedef int (*void_fn)(void);
    init_all_fn init_all = (init_all_fn)dlsym(h, "ncclCommInitAll");
    comm_fn destroy = (comm_fn)dlsym(h, "ncclCommDestroy");
    void_fn gstart = (void_fn)dlsym(h, "ncclGroupStart"), gend = (void_fn)dlsym(h, "ncclGroupEnd");
    if (!init_all || !destroy || !gstart || !gend) return give_up("RCCL symbols missing");
    std::vector<void *> comms((size_t)G, nullptr);
    if (init_all(comms.data(), G, devices.data()) != 0) return give_up("ncclCommInitAll failed");
    std::vector<std::thread> th;
    std::atomic<int> failed(0);
    for (int g = 0; g < G; ++g)
        th.emplace_back([&, g] {
            if (hipSetDevice(devices[(size_t)g]) != hipSuccess) { failed = 1; return; }
            // blocks one after the other: every device walks t in the same order, so the collectives pair up
            for (int t = 0; t < T; ++t)
                if (snk_stats_allreduce(ctx_of(g, t), comms[(size_t)g], nullptr) != SNK_OK) { failed = 1; return; }
            if (hipDeviceSynchronize() != hipSuccess) failed = 1;
        });
    for (auto &t : th) t.join();
    for (void *c : comms) destroy(c);
    if (failed) return give_up(string("all-reduce failed: ") + snk_last_error());
    return true;
}

// bytes the rmdup pre-pass may keep of the inflated input for the main pass (SNK_RMDUP_CACHE_GB, else 40 % of what the host /
// the cgroup has available)
size_t rmdup_cache_budget() {
    if (const char *e = getenv("SNK_RMDUP_CACHE_GB")) return (size_t)(atof(e) * 1073741824.0);
    double avail = 0;
    if (FILE *f = fopen("/proc/meminfo", "r")) {
        char line[256];
        while (fgets(line, sizeof line, f)) { unsigned long long kb; if (sscanf(line, "MemAvailable: %llu kB", &kb) == 1) avail = (double)kb * 1024.0; }
        fclose(f);
    }
    if (FILE *f = fopen("/sys/fs/cgroup/memory.max", "r")) {
        char a[64] = "";
        if (fscanf(f, "%63s", a) == 1 && strcmp(a, "max") != 0 && atof(a) > 0) {
            double lim = atof(a), used = 0;
            if (FILE *g = fopen("/sys/fs/cgroup/memory.current", "r")) { char b[64]; if (fscanf(g, "%63s", b) == 1) used = atof(b); fclose(g); }
            if (avail == 0 || lim - used < avail) avail = lim - used;
        }
        fclose(f);
    }
    return avail > 0 ? (size_t)(avail * 0.4) : 0;
}

// ---------------------------------------------------------------- sharded ingest / egress (--devices a,b,...)
// Plain-text input and several devices: one child process per device takes the g-th contiguous range of RECORDS of the input
// (byte ranges cut at record boundaries: the parent counts the newlines of both files once, in parallel), runs the ordinary
// single-device pipeline on it with the global index of its first pair (the virtual reference threads of appendix C are a function
// of that index), writes its own ordered part files and dumps its statistics blocks; the parent concatenates the parts in rank
// order -- the trick of the reference's per-thread temporaries, src/peprocess.cpp:2386 -- adds the blocks up and writes the
// reports.  Readers, writers and PCIe streams scale with the devices instead of feeding all of them from one reader and one writer.
struct ShardStatsHeader { uint64_t magic; int32_t T, lcap, nq, pad; uint64_t clean_total, dup_written; };     // pad: 0 = a shard's own blocks, 1 / 2 = all-reduced over RCCL / the host wire
const uint64_t SHARD_MAGIC = 0x534E4B5348415244ull;      // "SNKSHARD"

// the newlines of a plain FASTQ file counted once (in parallel, 1 MB pieces), then any record's byte offset
struct RecordIndex {
    const char *base = nullptr;
    size_t size = 0;
    uint64_t n_records = 0;
    static constexpr size_t W = (size_t)1 << 30;          // windows of 1 GB: the piece offsets are per window
    std::vector<uint64_t> wcount;                          // newlines up to the end of window w
    std::vector<std::vector<NlPiece>> wpieces;
    void open_and_count(const string &path, int workers) {
        const int fd = open(path.c_str(), O_RDONLY);
        if (fd < 0) die("cannot open file," + path);
        struct stat st;
        fstat(fd, &st);
        size = (size_t)st.st_size;
        base = (const char *)mmap(NULL, size, PROT_READ, MAP_PRIVATE, fd, 0);
        if (base == MAP_FAILED) die("cannot map file," + path);
        close(fd);
        uint64_t lines = 0;
        for (size_t a = 0; a < size; a += W) {
            std::vector<NlPiece> pieces;
            lines += count_pieces(base + a, 0, std::min(size, a + W) - a, workers, pieces);
            wcount.push_back(lines);
            wpieces.push_back(std::move(pieces));
        }
        if (size && base[size - 1] != '\n') ++lines;
        if (lines % 4) die("truncated fastq record");
        n_records = lines / 4;
    }
    uint64_t offset_of(uint64_t rec) const {               // where record number rec starts
        const uint64_t want = rec * 4;                     // behind this many newlines
        if (want == 0) return 0;
        if (rec >= n_records) return size;
        size_t w = 0;
        while (wcount[w] < want) ++w;
        return (uint64_t)w * W + locate_nl(base + w * W, wpieces[w], (size_t)(want - (w ? wcount[w - 1] : 0)));
    }
    void done() { if (base) munmap((void *)base, size); base = nullptr; }
};

// ---- .gz input of a sharded run: the parent's scout pass.  A DEFLATE stream can only be entered at a block header with the 32 KiB
// of text in front of it known, and a shard has to know the global number of its first record (the virtual reference threads and
// the duplicate marking are functions of it) -- so the parent decodes every input once on the host's cores (snk_pgunzip.h: all
// member CRCs are verified here), counts its lines and notes, for every shard border, the first chunk of the parallel decoder's
// chain that starts at or behind g/G of the compressed bytes: bit offset of its block header, the window, the number of the
// first whole record behind it and the bytes in front of that record.  The two files of a pair are cut at the same RECORD: the
// shard of the file whose place lies earlier drops the records in between (about the difference of the two compression ratios).
struct GzCut { uint64_t bit = 0, text_off = 0, rec = 0, skip_bytes = 0, nl_left = 0; bool have_win = false, resolved = false; std::vector<uint8_t> win; };
struct GzScout { uint64_t n_records = 0; std::vector<GzCut> cuts; string why; };

// The scout's result kept between runs: SNK_GZ_INDEX_DIR=<dir> -- one small file per input, shard count and chunk size,
// named after the input's base name, size and modification time; a later run over the SAME file (size and mtime unchanged) with the same
// shard count reads the borders from it instead of decoding the input once more in front of the run.  What the index does not repeat is
// the scout's check of every member's CRC-32: the file was checked by the run that wrote the index (a shard that enters a member in its
// middle cannot check it), which is why this is opt-in.
struct GzIndexHeader { char magic[8]; uint64_t size, mtime_ns, n_records, chunk; uint32_t G, ncuts; };
string gz_index_path(const string &path, int G, uint64_t chunk, uint64_t &size, uint64_t &mtime_ns) {
    const char *dir = getenv("SNK_GZ_INDEX_DIR");
    struct stat st;
    if (!dir || !*dir || stat(path.c_str(), &st) != 0) return "";
    size = (uint64_t)st.st_size;
    mtime_ns = (uint64_t)st.st_mtim.tv_sec * 1000000000ull + (uint64_t)st.st_mtim.tv_nsec;
    const size_t k = path.find_last_of('/');
    return string(dir) + "/" + (k == string::npos ? path : path.substr(k + 1)) + "." + std::to_string(size) + "." + std::to_string(mtime_ns) + "." +
           std::to_string(G) + "." + std::to_string(chunk) + ".snkidx";
}
bool gz_index_load(const string &ip, uint64_t size, uint64_t mtime_ns, int G, uint64_t chunk, GzScout &out) {
    FILE *f = ip.empty() ? nullptr : fopen(ip.c_str(), "rb");
    if (!f) return false;
    GzIndexHeader h;
    bool ok = fread(&h, sizeof h, 1, f) == 1 && !memcmp(h.magic, "SNKGZIX1", 8) && h.size == size && h.mtime_ns == mtime_ns && h.G == (uint32_t)G &&
              h.chunk == chunk && h.ncuts == (uint32_t)(G - 1);
    std::vector<GzCut> cuts;
    for (uint32_t k = 0; ok && k < h.ncuts; ++k) {
        GzCut c;
        uint64_t v[4];
        c.win.resize(snk::GzipInflate::HIST);
        ok = fread(v, 8, 4, f) == 4 && fread(c.win.data(), 1, c.win.size(), f) == c.win.size();
        c.bit = v[0]; c.text_off = v[1]; c.rec = v[2]; c.skip_bytes = v[3];
        c.have_win = c.resolved = true;
        ok = ok && c.rec < h.n_records && (c.bit >> 3) < size;
        cuts.push_back(std::move(c));
    }
    fclose(f);
    if (!ok) return false;                         // (a stale, foreign or cut-off index: the scout runs and writes a new one)
    out.n_records = h.n_records;
    out.cuts = std::move(cuts);
    out.why.clear();
    return true;
}
void gz_index_store(const string &ip, uint64_t size, uint64_t mtime_ns, int G, uint64_t chunk, const GzScout &sc) {
    if (ip.empty()) return;
    const string tmp = ip + ".tmp" + std::to_string((long)getpid());
    FILE *f = fopen(tmp.c_str(), "wb");
    if (!f) return;                                // (an index is a convenience: a directory that cannot be written costs nothing but the next scout)
    GzIndexHeader h;
    memset(&h, 0, sizeof h);
    memcpy(h.magic, "SNKGZIX1", 8);
    h.size = size; h.mtime_ns = mtime_ns; h.n_records = sc.n_records; h.chunk = chunk; h.G = (uint32_t)G; h.ncuts = (uint32_t)sc.cuts.size();
    bool ok = fwrite(&h, sizeof h, 1, f) == 1;
    for (const GzCut &c : sc.cuts) {
        const uint64_t v[4] = {c.bit, c.text_off, c.rec, c.skip_bytes};
        ok = ok && fwrite(v, 8, 4, f) == 4 && fwrite(c.win.data(), 1, c.win.size(), f) == c.win.size();
    }
    ok = (fclose(f) == 0) && ok;
    if (!ok || rename(tmp.c_str(), ip.c_str()) != 0) unlink(tmp.c_str());
}

bool scout_gz(const string &path, int workers, int G, GzScout &out, bool *from_index = nullptr) {
    size_t pg_chunk_key = (size_t)2 << 20;
    if (const char *e = getenv("SNK_GZ_CHUNK")) { const long v = atol(e); if (v >= 65536) pg_chunk_key = (size_t)v; }
    uint64_t isize = 0, imtime = 0;
    const string ipath = gz_index_path(path, G, pg_chunk_key, isize, imtime);
    if (from_index) *from_index = false;
    if (gz_index_load(ipath, isize, imtime, G, pg_chunk_key, out)) { if (from_index) *from_index = true; return true; }
    const int fd = open(path.c_str(), O_RDONLY);
    if (fd < 0) die("cannot open file," + path);
    struct stat st;
    fstat(fd, &st);
    const size_t n = (size_t)st.st_size;
    const uint8_t *zin = (const uint8_t *)mmap(NULL, n, PROT_READ, MAP_PRIVATE, fd, 0);
    if (zin == MAP_FAILED) die("cannot map file," + path);
    madvise((void *)zin, n, MADV_SEQUENTIAL);
    const size_t H = snk::GzipInflate::HIST, block = (size_t)1 << 24;
    {
        size_t pg_chunk = (size_t)2 << 20;
        if (const char *e = getenv("SNK_GZ_CHUNK")) { const long v = atol(e); if (v >= 65536) pg_chunk = (size_t)v; }
        snk::ParallelGunzip z(zin, n, workers, pg_chunk);
        int next_t = 1;
        bool crowded = false;
        z.on_chunk = [&](uint64_t bit, uint64_t text_off) {
            if (bit == 0 || next_t >= G || (bit >> 3) < (uint64_t)n * (uint64_t)next_t / (uint64_t)G) return;
            ++next_t;
            if (next_t < G && (bit >> 3) >= (uint64_t)n * (uint64_t)next_t / (uint64_t)G) crowded = true;      // one chunk serves one border
            GzCut c;
            c.bit = bit;
            c.text_off = text_off;
            out.cuts.push_back(std::move(c));
        };
        std::vector<uint8_t> buf(H + block, 0);
        uint8_t *data = buf.data() + H;
        uint64_t total = 0, nl_total = 0;
        uint8_t last = '\n';
        std::vector<NlPiece> pieces;
        for (;;) {
            const size_t got = z.run(data, block);
            if (z.error()) die(string("read error in input fastq (") + z.error() + ")," + path);
            if (got == 0 && z.done()) break;
            for (GzCut &c : out.cuts) {
                if (c.resolved) continue;
                size_t from = 0;
                if (!c.have_win) {                            // the chunk began inside this block: the window, the lines in front of it
                    const size_t k = (size_t)(c.text_off - total);
                    c.win.assign(buf.data() + k, buf.data() + k + H);
                    c.have_win = true;
                    uint64_t nl = nl_total;
                    for (size_t q = 0; q < k;) { const void *e = memchr(data + q, '\n', k - q); if (!e) break; q = (size_t)((const uint8_t *)e - data) + 1; ++nl; }
                    const uint8_t prev = k ? data[k - 1] : buf[H - 1];
                    if (prev == '\n' && nl % 4 == 0) { c.rec = nl / 4; c.skip_bytes = 0; c.resolved = true; continue; }
                    c.rec = nl / 4 + 1;                       // the record that starts behind line 4 * rec
                    c.nl_left = 4 * c.rec - nl;
                    from = k;
                }
                for (size_t q = from; q < got && c.nl_left;) {
                    const void *e = memchr(data + q, '\n', got - q);
                    if (!e) break;
                    q = (size_t)((const uint8_t *)e - data) + 1;
                    if (--c.nl_left == 0) { c.skip_bytes = total + q - c.text_off; c.resolved = true; }
                }
            }
            pieces.clear();
            nl_total += count_pieces((const char *)data, 0, got, workers, pieces);
            last = data[got - 1];
            total += got;
            // the last 32 KiB stay in front of the next block
            if (got >= H) memcpy(buf.data(), data + got - H, H);
            else { memmove(buf.data(), buf.data() + got, H); }
        }
        const uint64_t lines = nl_total + ((total > 0 && last != '\n') ? 1 : 0);
        if (lines % 4) die("truncated fastq record");
        out.n_records = lines / 4;
        if (crowded) out.why = "the input is too small for that many shards";
    }
    munmap((void *)zin, n);
    close(fd);
    if (out.why.empty() && (int)out.cuts.size() != G - 1) out.why = "the stream offers too few places to start from (few deflate blocks, or a stream the parallel decoder walks sequentially)";
    for (const GzCut &c : out.cuts) if (out.why.empty() && (!c.resolved || c.rec >= out.n_records)) out.why = "a border fell behind the last record";
    if (out.why.empty()) gz_index_store(ipath, isize, imtime, G, pg_chunk_key, out);
    return out.why.empty();
}

void append_file(int out_fd, const string &part) {
    const int fd = open(part.c_str(), O_RDONLY);
    if (fd < 0) die("cannot open such file," + part);
    std::vector<char> buf((size_t)8 << 20);
    for (;;) {
        const ssize_t n = read(fd, buf.data(), buf.size());
        if (n < 0) die("read error," + part);
        if (n == 0) break;
        for (ssize_t w = 0; w < n;) {
            const ssize_t k = write(out_fd, buf.data() + w, (size_t)(n - w));
            if (k <= 0) die("write error (disk full?)");
            w += k;
        }
    }
    close(fd);
}

}  // namespace

int main(int argc, char **argv) {
    const long long t_main0 = StageClock::now();
    auto since_start = [&](const char *what) { if (g_clk.on) fprintf(stderr, "[timing] t+%.3f s  %s\n", (double)(StageClock::now() - t_main0) * 1e-9, what); };
    if (argc < 2) { usage(); return 1; }
#if defined(__AVX2__) && defined(__x86_64__)
    // (built for x86-64-v3, soapnuke_amd/build.py)
    if (!__builtin_cpu_supports("avx2") || !__builtin_cpu_supports("bmi2")) { cerr << "Error:this build of SOAPnuke needs a CPU with AVX2 and BMI2 (x86-64-v3)" << endl; return 1; }
#endif
    Options o;
    parse_args(argc, argv, o);
    const int mates = o.p.paired ? 2 : 1;
    mkdir(o.out_dir.c_str(), 0755);
    if (const char *e = getenv("SNK_SHARD")) o.log += string(".shard") + std::to_string(atoi(e));      // (a shard of a sharded run keeps its own log)
    std::ofstream log(o.log.c_str());
    if (!log) die("cannot open such file," + o.log);
    log << local_time() << "\tAnalysis start!" << endl;
    // host workers: not tied to -T (which only decides how the reference would have dealt the reads to its threads)
    int ht = std::max(1, std::min(64, usable_cpus()));
    if (const char *e = getenv("SNK_HOST_THREADS")) { const int v = atoi(e); if (v >= 1 && v <= 1024) ht = v; }
    g_host_threads = ht;
    g_pool.start(ht);
    g_clk.on = getenv("SNK_TIMING") != nullptr;
    const int B = o.batch_pairs, T = o.threads, WK = ht;
    const string inputs[2] = {o.fq1, o.fq2};
    { struct stat st; for (int m = 0; m < mates; ++m) if (stat(inputs[m].c_str(), &st) != 0 || st.st_size == 0) die("cannot open file or empty file," + inputs[m]); }
    // ---- sharded ingest / egress: am I a shard, or shall I start the shards?
    if (const char *e = getenv("SNK_SHARD")) {
        unsigned long long lo1 = 0, hi1 = 0, lo2 = 0, hi2 = 0, first = 0;
        if (sscanf(e, "%d/%d", &g_shard.g, &g_shard.G) != 2 || !getenv("SNK_SHARD_RANGE1") || sscanf(getenv("SNK_SHARD_RANGE1"), "%llu,%llu", &lo1, &hi1) != 2 ||
            (mates == 2 && (!getenv("SNK_SHARD_RANGE2") || sscanf(getenv("SNK_SHARD_RANGE2"), "%llu,%llu", &lo2, &hi2) != 2)) ||
            !getenv("SNK_SHARD_FIRST") || sscanf(getenv("SNK_SHARD_FIRST"), "%llu", &first) != 1 || !getenv("SNK_SHARD_STATS") ||
            g_shard.g < 0 || g_shard.g >= g_shard.G || (size_t)g_shard.G != o.devices.size())
            die("bad SNK_SHARD environment");
        g_shard.child = true;
        g_shard.first = first;
        if (hi1 > lo1) g_shard.range[o.fq1] = {(size_t)lo1, (size_t)hi1};
        if (mates == 2 && hi2 > lo2) g_shard.range[o.fq2] = {(size_t)lo2, (size_t)hi2};
        for (int m = 0; m < mates; ++m)                      // .gz input: "bit,skip_bytes,skip_records,records,window file" (bit -1: the stream's header)
            if (const char *v = getenv(m ? "SNK_SHARD_GZ2" : "SNK_SHARD_GZ1")) {
                ShardEnv::Gz z;
                long long bit = -1;
                unsigned long long sb = 0, sr = 0, nr = 0;
                // (the window file's path travels in a variable of its own: an output directory may hold blanks)
                const char *wp = getenv(m ? "SNK_SHARD_GZWIN2" : "SNK_SHARD_GZWIN1");
                if (sscanf(v, "%lld,%llu,%llu,%llu", &bit, &sb, &sr, &nr) != 4) die("bad SNK_SHARD environment");
                z.skip_bytes = sb; z.skip_records = sr; z.nrec = nr;
                if (bit >= 0) {
                    if (!wp) die("bad SNK_SHARD environment");
                    z.bit = (uint64_t)bit;
                    z.win.resize(snk::GzipInflate::HIST);
                    FILE *f = fopen(wp, "rb");
                    if (!f || fread(z.win.data(), 1, z.win.size(), f) != z.win.size()) die(string("cannot read such file,") + wp);
                    fclose(f);
                }
                g_shard.gz[m ? o.fq2 : o.fq1] = std::move(z);
            }
        g_shard.stats_path = getenv("SNK_SHARD_STATS");
        if (const char *v = getenv("SNK_SHARD_TOTAL")) g_shard.total = strtoull(v, nullptr, 10);
        g_shard.wire_kind = getenv("SNK_SHARD_WIRE_KIND") ? getenv("SNK_SHARD_WIRE_KIND") : "file";
        g_shard.wire_addr = getenv("SNK_SHARD_WIRE_ADDR") ? getenv("SNK_SHARD_WIRE_ADDR") : "";
        o.devices = std::vector<int>(1, o.devices[(size_t)g_shard.g]);
        o.clean1 += ".part" + std::to_string(g_shard.g);
        if (mates == 2) o.clean2 += ".part" + std::to_string(g_shard.g);
        for (int m = 0; m < mates; ++m)
            if (!o.trim_fq[m].empty()) { o.trim_fq_gz[m] = ends_with_gz(o.trim_fq[m]); o.trim_fq[m] += ".part" + std::to_string(g_shard.g); }
        if (getenv("SNK_SHARD_FAKE")) {
            // test hook (tests/test_shard_plumbing.py, no GPU): the shard copies its record range to its part files and reports empty
            // statistics -- what is left is exactly the parent's work: record boundaries, environment, concatenation, merging
            const string names[2] = {o.clean1, o.clean2};
            for (int m = 0; m < mates; ++m) {
                const auto r = g_shard.range[m ? o.fq2 : o.fq1];
                const int fd = open((m ? o.fq2 : o.fq1).c_str(), O_RDONLY);
                std::vector<char> buf(r.second - r.first);
                if (fd < 0 || pread(fd, buf.data(), buf.size(), (off_t)r.first) != (ssize_t)buf.size()) die("fake shard: read error");
                close(fd);
                FILE *f = fopen((o.out_dir + "/" + names[m]).c_str(), "wb");
                if (!f || fwrite(buf.data(), 1, buf.size(), f) != buf.size() || fclose(f) != 0) die("fake shard: write error");
            }
            const int lc = 150, nqf = o.p.max_base_quality + 1;
            ShardStatsHeader h{SHARD_MAGIC, o.threads, lc, nqf, 0, 0, 0};
            std::vector<uint64_t> z((size_t)o.threads * ((size_t)snk_stats_u64(lc, nqf) + SNK_MAX_N), 0);
            z[(size_t)SNK_FS_N + SNK_GS_READS] = (uint64_t)(g_shard.first + 1);         // (something to add up: raw1 reads of virtual thread 0)
            FILE *f = fopen(g_shard.stats_path.c_str(), "wb");
            if (!f || fwrite(&h, sizeof h, 1, f) != 1 || fwrite(z.data(), 8, z.size(), f) != z.size() || fclose(f) != 0) die("fake shard: stats");
            _exit(0);
        }
        // The wire comes up NOW, before a byte of input is read, and stays open: every shard has just been started by the
        // same parent, so the 600 s of HostWire::connect -- the RCCL bootstrap goes through the same socket -- bound the start-up only.
        // Made on first use (the rmdup exchange, or the all-reduce at the very end) the rendezvous was a race between shards that
        // finish minutes apart -- .gz borders are cut by compressed bytes, a GPU may be shared or slow --: the early one gave up, the
        // parent killed the rest, the finished work was lost.  Behind the rendezvous a shard waits for its peers as long as it takes
        // (blocking reads; a peer that DIES closes its socket, and the parent ends the run).
        if (g_shard.wire_kind == "rccl" || g_shard.wire_kind == "host") {
            HIPCHK(hipSetDevice(o.devices[0]));
            (void)shard_wire();
        }
    }
    // (opt-in until it has met the hardware: SNK_SHARDED=1; without it several devices are fed batch by batch from one reader)
    // What keeps a run from being sharded: outputs that count reads across the whole input (-j streaming, -w / cleanOutSplit,
    // totalReadsNum) -- everything else, the host formatter's variants included (trimFq1/2, fasta, index removal, tile / fov), is a
    // per-read matter and runs in the shards as it does in one process.
    const bool shardable = !g_shard.child && o.devices.size() > 1 && getenv("SNK_SHARDED") && !strcmp(getenv("SNK_SHARDED"), "1") &&
                           (mates == 1 || is_gzip_file(o.fq1) == is_gzip_file(o.fq2)) && !o.streaming && o.clean_out_split == 0 && !(o.total_reads > 0);
    if (shardable) {
        const int G = (int)o.devices.size();
        std::vector<uint64_t> rec((size_t)G + 1), off[2];
        uint64_t nrec[2] = {0, 0};
        // (an input of fewer records per shard than this is not worth the processes; SNK_SHARD_MIN_RECORDS: the tests' small inputs)
        const uint64_t min_rec = getenv("SNK_SHARD_MIN_RECORDS") ? std::max<uint64_t>(1, strtoull(getenv("SNK_SHARD_MIN_RECORDS"), nullptr, 10)) : 4096;
        const bool gz_in = is_gzip_file(o.fq1);
        GzScout sc[2];
        bool plan = false;
        const long long t_plan = StageClock::now();
        if (!gz_in) {
            RecordIndex ri[2];
            for (int m = 0; m < mates; ++m) ri[m].open_and_count(inputs[m], ht);
            if (mates == 2 && ri[0].n_records != ri[1].n_records) die("reads number in fq1 and fq2 are different");
            nrec[0] = ri[0].n_records; nrec[1] = ri[1].n_records;
            if (nrec[0] >= (uint64_t)G * min_rec) {
                plan = true;
                for (int g = 0; g <= G; ++g) rec[(size_t)g] = nrec[0] * (uint64_t)g / (uint64_t)G;
                for (int m = 0; m < mates; ++m) { off[m].resize((size_t)G + 1); for (int g = 0; g <= G; ++g) off[m][(size_t)g] = ri[m].offset_of(rec[(size_t)g]); }
            }
            for (int m = 0; m < mates; ++m) ri[m].done();
            log << local_time() << "\trecord count of the plain input: " << nrec[0] << (mates == 2 ? " pairs, " : " reads, ") << (double)(StageClock::now() - t_plan) * 1e-9 << " s" << endl;
        } else {
            // .gz: the scout pass over both files at once (scout_gz above); the border records are the later of the two files' places
            const long long t_sc = StageClock::now();
            bool ok[2] = {true, true}, idx[2] = {false, false};
            {
                std::vector<std::thread> th;
                for (int m = 0; m < mates; ++m) th.emplace_back([&, m] { ok[m] = scout_gz(inputs[m], std::max(1, ht / mates), G, sc[m], &idx[m]); });
                for (auto &t : th) t.join();
            }
            nrec[0] = sc[0].n_records; nrec[1] = sc[1].n_records;
            if (mates == 2 && nrec[0] != nrec[1]) die("reads number in fq1 and fq2 are different");
            plan = ok[0] && ok[1] && nrec[0] >= (uint64_t)G * min_rec;
            if (plan) {
                rec[0] = 0; rec[(size_t)G] = nrec[0];
                for (int g = 1; g < G; ++g) {
                    rec[(size_t)g] = sc[0].cuts[(size_t)g - 1].rec;
                    if (mates == 2) rec[(size_t)g] = std::max(rec[(size_t)g], sc[1].cuts[(size_t)g - 1].rec);
                }
                for (int g = 0; g < G; ++g) if (rec[(size_t)g] >= rec[(size_t)g + 1]) plan = false;
            }
            const bool all_idx = idx[0] && (mates == 1 || idx[1]);
            log << local_time() << (all_idx ? "\tborders of the .gz input from the index in SNK_GZ_INDEX_DIR (no decode in front of the run): "
                                            : "\tscout pass over the .gz input: ")
                << nrec[0] << (mates == 2 ? " pairs, " : " reads, ") << (double)(StageClock::now() - t_sc) * 1e-9 << " s"
                << (plan ? "" : "; not sharded (" + (!ok[0] ? sc[0].why : !ok[1] ? sc[1].why : string("the borders do not leave every shard a record")) + ")") << endl;
            for (int m = 0; m < mates; ++m) { off[m].assign((size_t)G + 1, 0); }
        }
        if (getenv("SNK_SHARD_PLAN_ONLY")) {               // (measuring the parent's pass in front of the run: the log has its time)
            for (int g = 0; plan && g <= G; ++g) log << "border " << g << ": record " << rec[(size_t)g] << endl;
            log.close();
            _exit(0);
        }
        if (plan) {
            log << local_time() << "\tsharded run: " << G << " shards of about " << nrec[0] / (uint64_t)G << (mates == 2 ? " pairs" : " reads") << endl;
            // the wire between the shards (host/snk_wire.h): RCCL when the device list is G different GPUs, else -- the same device
            // twice: a one-GPU test configuration -- a host wire; SNK_SHARD_WIRE=rccl|host|file overrides ("file": no wire, the
            // parent adds up the dumped blocks, as round 4 did; rmdup needs a wire)
            string wire_kind, wire_addr, wire_why;
            if (getenv("SNK_SHARD_FAKE")) wire_kind = "file";      // (tests/test_shard_plumbing.py: the shards are stand-ins)
            else if (const char *e = getenv("SNK_SHARD_WIRE")) wire_kind = e;
            else {
                bool distinct = true;
                for (int a = 0; a < G; ++a) for (int b = a + 1; b < G; ++b) if (o.devices[(size_t)a] == o.devices[(size_t)b]) distinct = false;
                wire_kind = distinct ? "rccl" : "host";
            }
            if (wire_kind != "rccl" && wire_kind != "host" && wire_kind != "file") die("SNK_SHARD_WIRE: rccl, host or file");
            if (wire_kind != "file") wire_addr = "snk-shard-" + std::to_string((long)getpid()) + "-" + std::to_string((long long)StageClock::now());
            if (wire_kind == "file" && o.p.rmdup) die("rmdup over several shards exchanges the hashes: SNK_SHARD_WIRE=file cannot");
            log << local_time() << "\tshards talk over: " << (wire_kind == "rccl" ? "RCCL (bootstrap and fallback: host wire)" : wire_kind == "host" ? "host wire" : "files") << endl;
            std::vector<pid_t> kids((size_t)G, 0);
            for (int g = 0; g < G; ++g) {
                std::vector<string> env;
                for (char **e = environ; *e; ++e) env.push_back(*e);
                env.push_back("SNK_SHARD=" + std::to_string(g) + "/" + std::to_string(G));
                env.push_back("SNK_SHARD_RANGE1=" + std::to_string(off[0][(size_t)g]) + "," + std::to_string(off[0][(size_t)g + 1]));
                if (mates == 2) env.push_back("SNK_SHARD_RANGE2=" + std::to_string(off[1][(size_t)g]) + "," + std::to_string(off[1][(size_t)g + 1]));
                if (gz_in)
                    for (int m = 0; m < mates; ++m) {
                        // shard g of file m: from the stream's header (g = 0) or from the scout's place for border g; the records between
                        // that place and the border record belong to the shard in front
                        string v = "-1,0,0," + std::to_string(rec[(size_t)g + 1] - rec[(size_t)g]);
                        if (g > 0) {
                            const GzCut &c = sc[m].cuts[(size_t)g - 1];
                            const string wp = o.out_dir + "/shard." + std::to_string(g) + "." + std::to_string(m) + ".win";
                            FILE *f = fopen(wp.c_str(), "wb");
                            if (!f || fwrite(c.win.data(), 1, c.win.size(), f) != c.win.size() || fclose(f) != 0) die("cannot write to the file," + wp);
                            v = std::to_string(c.bit) + "," + std::to_string(c.skip_bytes) + "," + std::to_string(rec[(size_t)g] - c.rec) + "," +
                                std::to_string(rec[(size_t)g + 1] - rec[(size_t)g]);
                            env.push_back(string(m ? "SNK_SHARD_GZWIN2=" : "SNK_SHARD_GZWIN1=") + wp);
                        }
                        env.push_back(string(m ? "SNK_SHARD_GZ2=" : "SNK_SHARD_GZ1=") + v);
                    }
                env.push_back("SNK_SHARD_FIRST=" + std::to_string(rec[(size_t)g]));
                env.push_back("SNK_SHARD_TOTAL=" + std::to_string(nrec[0]));
                env.push_back("SNK_SHARD_STATS=" + o.out_dir + "/shard." + std::to_string(g) + ".stats");
                env.push_back("SNK_SHARD_WIRE_KIND=" + wire_kind);
                env.push_back("SNK_SHARD_WIRE_ADDR=" + wire_addr);
                env.push_back("SNK_HOST_THREADS=" + std::to_string(std::max(2, ht / G)));
                std::vector<char *> envp;
                for (auto &x : env) envp.push_back(const_cast<char *>(x.c_str()));
                envp.push_back(nullptr);
                if (posix_spawn(&kids[(size_t)g], "/proc/self/exe", nullptr, nullptr, argv, envp.data()) != 0) die("cannot start a shard");
            }
            // a shard that fails takes the others with it (they may be waiting for it inside a collective)
            int worst = 0;
            for (int left = G; left > 0;) {
                int status = 0;
                const pid_t w = waitpid(-1, &status, 0);
                if (w < 0) { worst = std::max(worst, 1); break; }
                int g = -1;
                for (int k = 0; k < G; ++k) if (kids[(size_t)k] == w) g = k;
                if (g < 0) continue;
                kids[(size_t)g] = 0;
                --left;
                const int rc = WIFEXITED(status) ? WEXITSTATUS(status) : 1;
                if (rc) {
                    worst = std::max(worst, rc);
                    for (int k = 0; k < G; ++k) if (kids[(size_t)k] > 0) kill(kids[(size_t)k], SIGKILL);
                }
            }
            if (gz_in) for (int g = 1; g < G; ++g) for (int m = 0; m < mates; ++m) unlink((o.out_dir + "/shard." + std::to_string(g) + "." + std::to_string(m) + ".win").c_str());
            if (worst) _exit(worst);                          // (the shard has printed the reference's message)
            // the parts in rank order (gzip members concatenate legally, as the reference's per-thread temporaries do)
            auto join_parts = [&](const string &fin) {
                const int fd = open(fin.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644);
                if (fd < 0) die("cannot write to the file," + fin);
                for (int g = 0; g < G; ++g) {
                    const string part = fin + ".part" + std::to_string(g);
                    append_file(fd, part);
                    unlink(part.c_str());
                }
                close(fd);
            };
            const string names[2] = {o.clean1, o.clean2};
            for (int m = 0; m < mates; ++m) join_parts(o.out_dir + "/" + names[m]);
            if (!o.trim_fq[0].empty()) for (int m = 0; m < mates; ++m) join_parts(o.out_dir + "/" + o.trim_fq[m]);
            if (o.p.rmdup) {                                  // dupReads.<thread>.<mate>.gz: every shard wrote its share of every virtual thread's file
                for (int m = 0; m < mates; ++m)
                    for (int t = 0; t < o.threads; ++t) join_parts(o.out_dir + "/dupReads." + std::to_string(t) + "." + std::to_string(m + 1) + ".gz");
                cout << "totalReadsNum:\t" << nrec[0] << endl;
            }
            // the statistics blocks of the shards, virtual thread by virtual thread
            const int T_ = o.threads;
            int lfin = 0, nq_ = 0;
            uint64_t dup_written = 0;
            std::vector<ShardStatsHeader> hd((size_t)G);
            std::vector<std::vector<uint64_t>> blob((size_t)G);
            for (int g = 0; g < G; ++g) {
                const string sp_ = o.out_dir + "/shard." + std::to_string(g) + ".stats";
                FILE *f = fopen(sp_.c_str(), "rb");
                if (!f || fread(&hd[(size_t)g], sizeof(ShardStatsHeader), 1, f) != 1 || hd[(size_t)g].magic != SHARD_MAGIC || hd[(size_t)g].T != T_) die("cannot read such file," + sp_);
                const size_t words = (size_t)T_ * ((size_t)snk_stats_u64(hd[(size_t)g].lcap, hd[(size_t)g].nq) + SNK_MAX_N);
                blob[(size_t)g].resize(words);
                if (fread(blob[(size_t)g].data(), 8, words, f) != words) die("cannot read such file," + sp_);
                fclose(f);
                unlink(sp_.c_str());
                unlink((o.log + ".shard" + std::to_string(g)).c_str());
                lfin = std::max(lfin, (int)hd[(size_t)g].lcap);
                nq_ = hd[(size_t)g].nq;
                dup_written += hd[(size_t)g].dup_written;
            }
            if (o.p.rmdup) log << "dup number:\t" << dup_written << endl;
            std::vector<std::vector<uint64_t>> sums((size_t)T_, std::vector<uint64_t>((size_t)snk_stats_u64(lfin, nq_), 0)), maxs((size_t)T_, std::vector<uint64_t>(SNK_MAX_N, 0));
            std::vector<const uint64_t *> sp((size_t)T_), mp((size_t)T_);
            for (int g = 0; g < G; ++g) {
                const size_t ns = (size_t)snk_stats_u64(hd[(size_t)g].lcap, hd[(size_t)g].nq);
                for (int t = 0; t < T_; ++t) {
                    widen_add(blob[(size_t)g].data() + (size_t)t * ns, hd[(size_t)g].lcap, sums[(size_t)t].data(), lfin, nq_);
                    const uint64_t *mx = blob[(size_t)g].data() + (size_t)T_ * ns + (size_t)t * SNK_MAX_N;
                    for (int k = 0; k < SNK_MAX_N; ++k) maxs[(size_t)t][(size_t)k] = std::max(maxs[(size_t)t][(size_t)k], mx[k]);
                }
            }
            for (int t = 0; t < T_; ++t) { sp[(size_t)t] = sums[(size_t)t].data(); mp[(size_t)t] = maxs[(size_t)t].data(); }
            // rank 0 holds the blocks as the wire's all-reduce left them (the path's one collective), has written the
            // reports from them and left them here: the sum over the dumped blocks above is the check, and the fallback
            bool merged_ok = false;
            const string mpath = o.out_dir + "/shard.merged.stats";
            if (FILE *f = fopen(mpath.c_str(), "rb")) {
                ShardStatsHeader mh;
                std::vector<uint64_t> mb((size_t)T_ * ((size_t)snk_stats_u64(lfin, nq_) + SNK_MAX_N));
                if (fread(&mh, sizeof mh, 1, f) == 1 && mh.magic == SHARD_MAGIC && mh.T == T_ && mh.lcap == lfin && mh.nq == nq_ && fread(mb.data(), 8, mb.size(), f) == mb.size()) {
                    merged_ok = true;
                    const size_t ns = (size_t)snk_stats_u64(lfin, nq_);
                    for (int t = 0; t < T_ && merged_ok; ++t)
                        merged_ok = memcmp(mb.data() + (size_t)t * ns, sums[(size_t)t].data(), ns * 8) == 0 &&
                                    memcmp(mb.data() + (size_t)T_ * ns + (size_t)t * SNK_MAX_N, maxs[(size_t)t].data(), SNK_MAX_N * 8) == 0;
                    if (merged_ok) log << local_time() << "\tstatistics merged over " << (mh.pad == 1 ? "RCCL" : "the host wire") << " (" << G << " shards)" << endl;
                    else cerr << "Warning:the all-reduce of the statistics over the shards disagrees with the sum of their dumped blocks; using that sum" << endl;
                }
                fclose(f);
                unlink(mpath.c_str());
            } else if (wire_kind != "file") cerr << "Warning:the shards did not all-reduce their statistics; merging their dumped blocks on the host" << endl;
            char ebuf[512];
            o.p.max_read_len = lfin;
            if (!merged_ok && snk_write_reports(&o.p, T_, sp.data(), mp.data(), o.out_dir.c_str(), ebuf, sizeof ebuf) != 0) { cerr << ebuf << endl; _exit(1); }
            log << local_time() << "\tAnalysis accomplished!" << endl;
            log.close();
            _exit(0);
        }
    }
    const int space_num = first_line_space_num(o.fq1);
    char bc_from = 0, bc_to = 0;                          // baseConvert "TtoU" / "T2U" / "TU" (src/peprocess.cpp:1629-1646)
    if (!o.base_convert.empty()) {
        string b = o.base_convert;
        if (b.find("TO") != string::npos) b.replace(b.find("TO"), 2, "");
        if (b.find("2") != string::npos) b.replace(b.find("2"), 1, "");
        if (b.size() != 2) die("base_conver value format error");
        bc_from = (char)toupper(b[0]);
        bc_to = b[1];
    }

    // Device-text mode (include/snk_fastq.h): the raw text of a batch is uploaded as it is, the device builds the
    // line index, fills the SoA planes and, behind the filter kernels, gathers the clean text; the host only reads, counts
    // newlines, (de)compresses and writes.  The output variants that need per-record work on the names or several outputs
    // per read keep the host formatter (same bytes either way; SNK_HOST_TEXT=1 forces it).
    // rmdup of paired input in device-text mode is one pass (include/snk_rmdup.h, snk_rmdup_stream_*): every batch is hashed and
    // looked up in a table that stays in HBM (single end: with the reference's one-read shift of the flags inside full patches,
    // snk_rmdup_stream_mark_se_device -- the batches end on patch borders).  With several devices the table lives on the first:
    // the other devices send their batches' hashes there and get the flags back (9 bytes per pair against the ~700 of the
    // pair's text), marked in input order on a stream of the table's device.  SNK_RMDUP_TWO_PASS=1 forces the reference's two passes.
    const int64_t rmdup_patch = o.patch_size > 0 ? o.patch_size : (int64_t)o.threads * 20000 / 8;
    const bool proven_only = getenv("SNK_PROVEN_ONLY") && !strcmp(getenv("SNK_PROVEN_ONLY"), "1");      // (csrc/snk_tables.h: the hardware-green envelope only)
    // (a shard of a sharded run: "seen at an earlier index" spans the shards -- two passes with the hash exchange in between)
    bool rmdup_one_pass = o.p.rmdup && !g_shard.child && (mates == 2 || (!proven_only && rmdup_patch > 0 && o.batch_pairs % rmdup_patch == 0)) && !getenv("SNK_RMDUP_TWO_PASS");
    if (rmdup_one_pass) {
        // the one-pass table keeps 8 B per pair of hashes and up to 48 B per pair of table resident in HBM: when twice the
        // estimated number of pairs (file size / bytes of the first record) does not fit, the two passes run instead
        uint64_t guess = 0;
        struct stat stf;
        if (stat(inputs[0].c_str(), &stf) == 0) {
            if (gzFile g = gzopen(inputs[0].c_str(), "rb")) {          // (transparent for plain text)
                std::vector<char> line(1 << 16);
                size_t rec = 0;
                for (int k = 0; k < 4 && gzgets(g, line.data(), (int)line.size()); ++k) rec += strlen(line.data());
                gzclose(g);
                if (rec > 0) guess = (uint64_t)((double)stf.st_size * (is_gzip_file(inputs[0]) ? 6.0 : 1.05) / (double)rec) + 1024;
            }
        }
        size_t free_b = 0, total_b = 0;
        HIPCHK(hipSetDevice(o.devices[0]));
        HIPCHK(hipMemGetInfo(&free_b, &total_b));
        if (const char *e = getenv("SNK_RMDUP_FREE_MB_TEST")) free_b = (size_t)atol(e) << 20;
        const uint64_t want = snk_rmdup_stream_bytes(std::min<uint64_t>(2 * guess, 4294967295ull));
        if ((double)want > 0.85 * (double)free_b) {
            log << local_time() << "\trmdup: two passes (the one-pass table would need " << (want >> 20) << " MB of device memory, " << (free_b >> 20) << " MB are free)" << endl;
            rmdup_one_pass = false;
        }
    }
    const bool dev_text = !getenv("SNK_HOST_TEXT") && !o.streaming && (!o.p.rmdup || rmdup_one_pass) && o.trim_fq[0].empty() && o.clean_out_split == 0 &&
                          !(o.total_reads > 0) && o.out_file_type != "fasta" && !o.index_remove && o.tile.empty() && o.fov.empty();
    const bool rmdup_stream = dev_text && o.p.rmdup;
    // .gz output of device-text mode: the gzip members are made on the device too (SNK_HOST_DEFLATE=1: by the host's encoder)
    const bool dev_gz = dev_text && o.out_gz && !getenv("SNK_HOST_DEFLATE");
    const int GZ_RPM = 512;                                 // records per gzip member
    // ---- readers
    std::unique_ptr<Channel<RawChunk *>> chan[2];          // one per input file, re-made for every pass over the input
    std::vector<std::thread> readers;
    auto start_readers = [&] {
        for (int m = 0; m < mates; ++m) {
            chan[m].reset(new Channel<RawChunk *>(2));
            readers.emplace_back(reader_main, inputs[m], B, space_num, std::max(1, WK / mates), !dev_text, chan[m].get());
        }
    };
    auto join_readers = [&] { for (auto &t : readers) t.join(); readers.clear(); };
    auto next_chunks = [&](RawChunk *c[2]) -> bool {
        bool ok[2] = {true, true};
        c[0] = c[1] = nullptr;
        for (int m = 0; m < mates; ++m) ok[m] = chan[m]->pop(c[m]);
        if (mates == 2 && (ok[0] != ok[1] || (ok[0] && c[0]->n != c[1]->n))) die("reads number in fq1 and fq2 are different");
        return ok[0];
    };

    // ---- capacity (longest read) and pitch: sized from the first batch, regrown when a longer read appears
    start_readers();
    RawChunk *first[2];
    if (!next_chunks(first)) die("no data");
    since_start("first batch read");
    if (dev_text) for (int m = 0; m < mates; ++m) index_chunk(first[m], space_num, WK);      // read lengths + quality-system check of the first batch
    auto longest = [&](RawChunk *const c[2]) {
        int mx = 1;
        for (int m = 0; m < mates; ++m)
            for (int i = 0; i < c[m]->n; ++i) { int l; c[m]->line(4 * i + 1, l); mx = std::max(mx, l); }
        return mx;
    };
    if (!g_shard.child || g_shard.g == 0) phred_sanity(o, *first[0]);       // (the reference looks at the first patch of the file only)

    // one context + one set of batch slots per device; one accumulator per virtual reference thread
    //  and device.  A run is a sequence of epochs of constant capacity (normally one).
    const int G = (int)o.devices.size();
    // -j prints every patch's cumulative per-thread statistics from the accumulators of the device that ran it: with
    // several devices each holds only its share of a thread's patches
    if (o.streaming && G > 1) die("-j/--streaming runs on one device: give --devices a single id");
    const int NSLOT = dev_text ? 4 : 3;                   // (device-text mode: the main thread holds one staged batch back until its parse verdict is in)
    const int64_t vblock = snk_vthread_block(T, o.patch_size);
    struct Dev {
        int id = 0;
        snk_ctx *ctx = nullptr;
        std::vector<uint64_t *> d_sum, d_max;
        std::vector<Slot> slots;
        std::unique_ptr<Channel<Slot *>> free_slots;
    };
    std::vector<Dev> devs((size_t)G);
    for (int g = 0; g < G; ++g) devs[(size_t)g].id = o.devices[(size_t)g];
    int32_t lcap = 0, nq = 0;
    int64_t nsum = 0;
    int pitch = 0;
    size_t plane = 0;
    size_t text_cap = 0;                                  // device-text mode: bytes of one mate's text per batch the slots can take
    struct Epoch { int lcap; std::vector<std::vector<uint64_t>> sums, maxs; };
    std::vector<Epoch> epochs;

    std::vector<std::thread> slot_makers;
    auto setup = [&](int maxlen) {
        if (maxlen > SNK_READ_MAX_LEN) die("read longer than 1000 bases");
        o.p.max_read_len = maxlen;
        pitch = (maxlen + 15) / 16 * 16;
        plane = (size_t)B * (size_t)pitch;
        for (Dev &d : devs) {
            HIPCHK(hipSetDevice(d.id));
            d.ctx = snk_create(&o.p, d.id);
            if (!d.ctx) die(snk_last_error());
            if (snk_reserve(d.ctx, B, std::min(8, NSLOT)) != SNK_OK) die(snk_last_error());      // (no allocation on the launch path)
            snk_stats_geometry(d.ctx, &lcap, &nq, &nsum);
            d.d_sum.assign((size_t)T, nullptr);
            d.d_max.assign((size_t)T, nullptr);
            for (int t = 0; t < T; ++t) {
                HIPCHK(hipMalloc(&d.d_sum[t], nsum * sizeof(uint64_t)));
                HIPCHK(hipMalloc(&d.d_max[t], SNK_MAX_N * sizeof(uint64_t)));
                HIPCHK(hipMemset(d.d_sum[t], 0, nsum * sizeof(uint64_t)));
                HIPCHK(hipMemset(d.d_max[t], 0, SNK_MAX_N * sizeof(uint64_t)));
            }
            d.slots.assign(NSLOT, Slot());
            d.free_slots.reset(new Channel<Slot *>(NSLOT + 1));
            // the first slot now, the others behind the scenes while the first batches run (pinning a few hundred MB per slot
            // is most of the start-up time of a short run)
            auto alloc_slot = [&, dp = &d](Slot &sl) {
                Dev &d = *dp;
                HIPCHK(hipSetDevice(d.id));
                sl.dev = (int)(dp - devs.data());
                for (int m = 0; m < mates; ++m) {
                    HIPCHK(hipMalloc(&sl.d_seq[m], plane)); HIPCHK(hipMalloc(&sl.d_qual[m], plane));
                    HIPCHK(hipMalloc(&sl.d_len[m], (size_t)B * 2)); HIPCHK(hipMalloc(&sl.d_rec[m], (size_t)B * sizeof(snk_read_result)));
                    if (dev_text) {
                        HIPCHK(hipHostMalloc(&sl.h_text[m], text_cap + 64)); HIPCHK(hipMalloc(&sl.d_text[m], text_cap + 64)); HIPCHK(hipMalloc(&sl.d_out[m], text_cap + 64));
                        HIPCHK(hipMalloc(&sl.d_line[m], ((size_t)B * 4 + 1) * 4)); HIPCHK(hipMalloc(&sl.d_outoff[m], ((size_t)B + 1) * 4));
                        HIPCHK(hipHostMalloc(&sl.h_outoff[m], ((size_t)B + 1) * 4));
                        HIPCHK(hipMalloc(&sl.d_status[m], SNK_FQ_STATUS_N * 4)); HIPCHK(hipHostMalloc(&sl.h_status[m], SNK_FQ_STATUS_N * 4));
                        if (dev_gz) {
                            HIPCHK(hipMalloc(&sl.d_gz[m], text_cap + 64)); HIPCHK(hipMalloc(&sl.d_gzinfo[m], 16)); HIPCHK(hipHostMalloc(&sl.h_gzinfo[m], 16));
                        }
                        if (rmdup_stream) {
                            HIPCHK(hipMalloc(&sl.d_dupout[m], text_cap + 64)); HIPCHK(hipMalloc(&sl.d_dupoff[m], ((size_t)B + 1) * 4));
                            HIPCHK(hipHostMalloc(&sl.h_dupoff[m], ((size_t)B + 1) * 4));
                        }
                        continue;
                    }
                    HIPCHK(hipHostMalloc(&sl.h_seq[m], plane)); HIPCHK(hipHostMalloc(&sl.h_qual[m], plane));
                    HIPCHK(hipHostMalloc(&sl.h_len[m], (size_t)B * 2)); HIPCHK(hipHostMalloc(&sl.h_rec[m], (size_t)B * sizeof(snk_read_result)));
                    memset(sl.h_seq[m], 0, plane); memset(sl.h_qual[m], 0, plane);
                }
                if (dev_text) {
                    sl.tmp_bytes = snk_fastq_tmp_bytes(text_cap, B);
                    HIPCHK(hipMalloc(&sl.d_tmp, sl.tmp_bytes));
                    if (dev_gz) { sl.ztmp_bytes = snk_fastq_deflate_tmp_bytes(B, GZ_RPM); HIPCHK(hipMalloc(&sl.d_ztmp, sl.ztmp_bytes)); }
                    HIPCHK(hipEventCreateWithFlags(&sl.parsed, hipEventDisableTiming));
                    if (rmdup_stream) {
                        HIPCHK(hipMalloc(&sl.d_hash, (size_t)B * sizeof(uint64_t)));
                        if (d.id != devs[0].id) {
                            HIPCHK(hipEventCreateWithFlags(&sl.hashed, hipEventDisableTiming));
                            HIPCHK(hipSetDevice(devs[0].id));
                            HIPCHK(hipMalloc(&sl.d_hash0, (size_t)B * sizeof(uint64_t)));
                            HIPCHK(hipMalloc(&sl.d_flags0, (size_t)B));
                            HIPCHK(hipEventCreateWithFlags(&sl.marked, hipEventDisableTiming));
                            HIPCHK(hipSetDevice(d.id));
                        }
                    }
                }
                HIPCHK(hipHostMalloc(&sl.h_flags, (size_t)B)); HIPCHK(hipMalloc(&sl.d_flags, (size_t)B));
                HIPCHK(hipHostMalloc(&sl.h_err, sizeof(uint64_t)));
                HIPCHK(hipStreamCreate(&sl.stream));
                HIPCHK(hipEventCreate(&sl.done));
                d.free_slots->push(&sl);
            };
            alloc_slot(d.slots[0]);
            slot_makers.emplace_back([&, dp = &d, alloc_slot] { for (int k = 1; k < NSLOT; ++k) alloc_slot(dp->slots[(size_t)k]); });
        }
    };
    auto teardown = [&] {
        for (auto &t : slot_makers) t.join();
        slot_makers.clear();
        for (Dev &d : devs) {
            HIPCHK(hipSetDevice(d.id));
            HIPCHK(hipDeviceSynchronize());
            for (Slot &sl : d.slots) {
                for (int m = 0; m < mates; ++m) {
                    HIPCHK(hipFree(sl.d_seq[m])); HIPCHK(hipFree(sl.d_qual[m])); HIPCHK(hipFree(sl.d_len[m])); HIPCHK(hipFree(sl.d_rec[m]));
                    if (dev_text) {
                        HIPCHK(hipHostFree(sl.h_text[m])); HIPCHK(hipFree(sl.d_text[m])); HIPCHK(hipFree(sl.d_out[m])); HIPCHK(hipFree(sl.d_line[m]));
                        HIPCHK(hipFree(sl.d_outoff[m])); HIPCHK(hipHostFree(sl.h_outoff[m])); HIPCHK(hipFree(sl.d_status[m])); HIPCHK(hipHostFree(sl.h_status[m]));
                        if (dev_gz) { HIPCHK(hipFree(sl.d_gz[m])); HIPCHK(hipFree(sl.d_gzinfo[m])); HIPCHK(hipHostFree(sl.h_gzinfo[m])); }
                        if (rmdup_stream) { HIPCHK(hipFree(sl.d_dupout[m])); HIPCHK(hipFree(sl.d_dupoff[m])); HIPCHK(hipHostFree(sl.h_dupoff[m])); }
                        continue;
                    }
                    HIPCHK(hipHostFree(sl.h_seq[m])); HIPCHK(hipHostFree(sl.h_qual[m])); HIPCHK(hipHostFree(sl.h_len[m])); HIPCHK(hipHostFree(sl.h_rec[m]));
                }
                if (dev_text) { HIPCHK(hipFree(sl.d_tmp)); HIPCHK(hipEventDestroy(sl.parsed)); }
                if (rmdup_stream) {
                    HIPCHK(hipFree(sl.d_hash));
                    if (sl.d_hash0) {
                        HIPCHK(hipEventDestroy(sl.hashed));
                        HIPCHK(hipSetDevice(devs[0].id));
                        HIPCHK(hipFree(sl.d_hash0)); HIPCHK(hipFree(sl.d_flags0)); HIPCHK(hipEventDestroy(sl.marked));
                        HIPCHK(hipSetDevice(d.id));
                        sl.d_hash0 = nullptr; sl.d_flags0 = nullptr; sl.hashed = sl.marked = nullptr;
                    }
                }
                if (dev_gz) HIPCHK(hipFree(sl.d_ztmp));
                HIPCHK(hipHostFree(sl.h_flags)); HIPCHK(hipFree(sl.d_flags)); HIPCHK(hipHostFree(sl.h_err));
                HIPCHK(hipStreamDestroy(sl.stream));
                HIPCHK(hipEventDestroy(sl.done));
            }
            d.slots.clear();
            for (int t = 0; t < T; ++t) { HIPCHK(hipFree(d.d_sum[t])); HIPCHK(hipFree(d.d_max[t])); }
            snk_destroy(d.ctx);
            d.ctx = nullptr;
        }
    };
    auto report_device_error = [&](const snk_error &err) {
        if (err.code == SNK_E_BAD_BASE) die("unrecognized sequence, read " + std::to_string(err.index) + " of fq" + std::to_string(err.mate + 1));
        if (err.code == SNK_E_EMPTY_SEQ) die("empty sequence");
        if (err.code == SNK_E_QUAL_RANGE) die("quality is too high or too low,please check the quality system parameter or fastq file");
        if (err.code) die("device reported error " + std::to_string(err.code));
    };
    // end of an epoch: the per-device blocks of every virtual thread become one (the path's only collective,
    // a sum/max all-reduce over RCCL, issued by one host thread per GPU), then travel to the host
    auto collect_epoch = [&] {
        for (Dev &d : devs) { HIPCHK(hipSetDevice(d.id)); HIPCHK(hipDeviceSynchronize()); }
        Epoch e;
        e.lcap = lcap;
        e.sums.assign((size_t)T, std::vector<uint64_t>((size_t)nsum, 0));
        e.maxs.assign((size_t)T, std::vector<uint64_t>(SNK_MAX_N, 0));
        std::vector<uint64_t> ts((size_t)nsum), tm(SNK_MAX_N);
        auto fetch_block = [&](Dev &d, int t) {
            snk_error err;
            HIPCHK(hipSetDevice(d.id));
            if (snk_bind_stats(d.ctx, d.d_sum[(size_t)t], d.d_max[(size_t)t]) != SNK_OK) die(snk_last_error());
            if (snk_stats_fetch(d.ctx, ts.data(), tm.data(), &err, nullptr) != SNK_OK) die(snk_last_error());
            report_device_error(err);
        };
        // every device's blocks come to the host first and are added up here: the RCCL all-reduce below is then checked
        // against this sum, and a failure of the collective costs a warning, not the run
        for (int g = 0; g < G; ++g)
            for (int t = 0; t < T; ++t) {
                fetch_block(devs[(size_t)g], t);
                for (int64_t k = 0; k < nsum; ++k) e.sums[(size_t)t][(size_t)k] += ts[(size_t)k];
                for (int k = 0; k < SNK_MAX_N; ++k) e.maxs[(size_t)t][(size_t)k] = std::max(e.maxs[(size_t)t][(size_t)k], tm[(size_t)k]);
            }
        if (G > 1 && rccl_allreduce_blocks(o.devices, [&](int g, int t) {
                Dev &d = devs[(size_t)g];
                if (snk_bind_stats(d.ctx, d.d_sum[(size_t)t], d.d_max[(size_t)t]) != SNK_OK) die(snk_last_error());
                return d.ctx;
            }, T)) {
            // the path's one collective: after it every device holds the merged blocks
            bool same = true;
            for (int t = 0; t < T && same; ++t) {
                fetch_block(devs[0], t);
                same = memcmp(ts.data(), e.sums[(size_t)t].data(), (size_t)nsum * sizeof(uint64_t)) == 0 &&
                       memcmp(tm.data(), e.maxs[(size_t)t].data(), SNK_MAX_N * sizeof(uint64_t)) == 0;
            }
            if (!same) cerr << "Warning:the RCCL all-reduce of the statistics disagrees with the host-side sum; using the host-side sum" << endl;
            else log << local_time() << "\tstatistics merged over RCCL (" << G << " devices)" << endl;
        }
        epochs.push_back(std::move(e));
    };
    auto text_cap_for = [&](RawChunk *const c[2]) {      // room for a batch of this kind of records, with some slack for longer names
        size_t mx = 0;
        for (int m = 0; m < mates; ++m) mx = std::max(mx, c[m]->n ? c[m]->nbytes / (size_t)c[m]->n : 0);
        const int recs = c[0]->n < B ? c[0]->n : B;         // a short first batch is the whole input
        return (size_t)((double)(mx + 8) * 1.15 * (double)recs) + ((size_t)4 << 20);
    };
    if (dev_text) text_cap = text_cap_for(first);
    since_start("first batch indexed");
    setup(o.streaming ? std::max(longest(first), 256) : longest(first));   // -j: the first batch is one small patch, a poor sample of the read lengths
    since_start("contexts and slots ready");

    // records -> pinned planes (parallel over records); false: a read is longer than the capacity
    auto pack = [&](Slot &s, bool with_qual) -> bool {
        const int n = s.n;
        std::atomic<int> bad(0);
        parallel_for(WK, n, [&](int, int lo, int hi) {
            for (int m = 0; m < mates; ++m)
                for (int i = lo; i < hi; ++i) {
                    int ls, lq;
                    const char *sq = s.raw[m]->line(4 * i + 1, ls), *ql = s.raw[m]->line(4 * i + 3, lq);
                    if (ls > lcap) { bad |= 1; continue; }
                    if (lq != ls) { bad |= 2; continue; }
                    memcpy(s.h_seq[m] + (size_t)i * pitch, sq, ls);
                    if (with_qual) memcpy(s.h_qual[m] + (size_t)i * pitch, ql, lq);
                    s.h_len[m][i] = (uint16_t)ls;
                }
        });
        if (bad & 2) die("sequence and quality lengths differ");
        return !(bad & 1);
    };

    // ---- rmdup pre-pass (src/peprocess.cpp:3071-3152): hash every raw pair on the GPU, keep the hashes
    // resident, mark every later occurrence; the flags enter the cascade as snk_batch.dup below.
    uint8_t *d_dup_all = nullptr;
    std::vector<uint8_t> dup_host;                        // the same flags on the host (combined with tile/fov bits per batch)
    std::vector<OutFile> dupw[2];
    snk_rmdup_stream *dup_table = nullptr;
    hipStream_t mark_stream = nullptr;                      // (of the table's device: batches of the other devices are marked there)
    if (rmdup_stream) {
        HIPCHK(hipSetDevice(devs[0].id));
        if (G > 1) HIPCHK(hipStreamCreate(&mark_stream));
        uint64_t guess = 1u << 20;                          // pairs in the input, from the file sizes and the first batch (the table grows if it was short)
        struct stat st;
        if (stat(inputs[0].c_str(), &st) == 0 && first[0]->n > 0 && first[0]->nbytes > 0) {
            const double per = (double)first[0]->nbytes / (double)first[0]->n;
            guess = (uint64_t)((double)st.st_size * (is_gzip_file(inputs[0]) ? 6.0 : 1.05) / per) + 1024;
        }
        dup_table = snk_rmdup_stream_create(devs[0].ctx, std::min<uint64_t>(guess, 4294967295ull));
        if (!dup_table) die(snk_last_error());
        log << local_time() << "\trmdup: one pass (duplicate table resident on the device" << (mates == 1 ? "; batches of " + std::to_string(B) + " reads end on patch borders)" : ")") << endl;
        for (int m = 0; m < mates; ++m) {                   // dupReads.<thread>.<mate>.gz, src/peprocess.cpp:167-174
            dupw[m].resize(T);
            for (int t = 0; t < T; ++t) dupw[m][t].open(o.out_dir + "/dupReads." + std::to_string(t) + "." + std::to_string(m + 1) + ".gz", true);
        }
    }
    if (o.p.rmdup && !rmdup_stream) {
        if (const char *e = getenv("SNK_RMDUP_TWO_PASS")) if (!strcmp(e, "restarted")) log << local_time() << "\trmdup: two passes (restarted: sentinel hash in the input)" << endl;
        std::vector<uint64_t *> chunks;
        std::vector<int> chunk_n;
        uint64_t nall = 0;
        RawChunk *c[2] = {first[0], first[1]};
        bool have = true;
        std::vector<RawChunk *> cached[2];
        size_t cache_bytes = 0;
        const size_t cache_budget = rmdup_cache_budget();
        bool cache_on = cache_budget > 0 && is_gzip_file(inputs[0]);
        HIPCHK(hipSetDevice(devs[0].id));                   // the pre-pass runs on the first device (1 ms of kernel per 10 M pairs)
        snk_ctx *ctx = devs[0].ctx;
        while (have) {
            Slot *sp0 = &devs[0].slots[0];
            sp0->n = c[0]->n;
            sp0->raw[0] = c[0]; sp0->raw[1] = c[1];
            if (!pack(*sp0, false)) {                       // a longer read than any before: new capacity (nothing accumulated yet)
                teardown();
                setup(longest(c));
                HIPCHK(hipSetDevice(devs[0].id));
                ctx = devs[0].ctx;
                sp0 = &devs[0].slots[0];
                sp0->n = c[0]->n;
                sp0->raw[0] = c[0]; sp0->raw[1] = c[1];
                pack(*sp0, false);
            }
            Slot &s = *sp0;
            uint64_t *dh;
            HIPCHK(hipMalloc(&dh, (size_t)s.n * sizeof(uint64_t)));
            snk_batch b;
            memset(&b, 0, sizeof b);
            b.n = s.n;
            b.pitch = pitch;
            for (int m = 0; m < mates; ++m) {
                HIPCHK(hipMemcpyAsync(s.d_seq[m], s.h_seq[m], (size_t)s.n * pitch, hipMemcpyHostToDevice, s.stream));
                HIPCHK(hipMemcpyAsync(s.d_len[m], s.h_len[m], (size_t)s.n * 2, hipMemcpyHostToDevice, s.stream));
                b.seq[m] = s.d_seq[m]; b.qual[m] = s.d_qual[m]; b.len[m] = s.d_len[m];
            }
            if (snk_rmdup_hash_device(ctx, &b, dh, s.stream) != SNK_OK) die(snk_last_error());
            HIPCHK(hipStreamSynchronize(s.stream));        // the pinned planes are refilled next
            chunks.push_back(dh);
            chunk_n.push_back(s.n);
            nall += (uint64_t)s.n;
            // .gz input: the inflated batches are kept for the main pass as long as they fit the budget (the reference reads and
            // inflates its input twice with rmdup, src/peprocess.cpp:3071-3152; here the second inflate is the larger half of the run)
            if (cache_on) {
                for (int m = 0; m < mates; ++m) { cached[m].push_back(c[m]); cache_bytes += c[m]->own_cap + (c[m]->ls.capacity() + c[m]->le.capacity()) * 4; }
                if (cache_bytes > cache_budget) {
                    cache_on = false;
                    for (int m = 0; m < mates; ++m) { for (RawChunk *q : cached[m]) RawChunk::put(q); cached[m].clear(); }
                }
            } else {
                for (int m = 0; m < mates; ++m) RawChunk::put(c[m]);
            }
            have = next_chunks(c);
        }
        join_readers();
        const uint64_t nglobal = g_shard.child ? g_shard.total : nall, gfirst = g_shard.child ? g_shard.first : 0;
        if (nglobal > 4294967295ull) die("reads number is too large to do remove duplication," + std::to_string(nglobal));
        uint64_t *d_hash_all;
        HIPCHK(hipMalloc(&d_hash_all, std::max<size_t>((size_t)nall, 1) * sizeof(uint64_t)));
        HIPCHK(hipMalloc(&d_dup_all, std::max<size_t>((size_t)nall, 1)));
        uint64_t off = 0;
        for (size_t k = 0; k < chunks.size(); ++k) {
            HIPCHK(hipMemcpyAsync(d_hash_all + off, chunks[k], (size_t)chunk_n[k] * sizeof(uint64_t), hipMemcpyDeviceToDevice, 0));
            off += (uint64_t)chunk_n[k];
        }
        HIPCHK(hipStreamSynchronize(0));
        for (uint64_t *ch : chunks) HIPCHK(hipFree(ch));
        uint8_t prev_flag = 0;                              // SE: the true flag of the read in front of this shard
        if (!g_shard.child) {
            cout << "totalReadsNum:\t" << nall << endl;
            if (snk_rmdup_mark_device(ctx, d_hash_all, nullptr, (int64_t)nall, nall, -1, d_dup_all, nullptr) != SNK_OK) die(snk_last_error());
            HIPCHK(hipStreamSynchronize(0));
        } else {
            // The exchange of the sharded run (include/snk_rmdup.h "multi-GPU exchange helpers"): the population of the
            // sentinel's bucket summed over the shards, every hash to its owner (hash % G) with its global index, marked there
            // with explicit indices, the flags home again.  9 + 4 bytes per pair over the wire against the ~700 of its text.
            snk::ShardWire *w = shard_wire();
            if (!w) die("rmdup over several shards needs a wire between them");
            const int W = w->world;
            auto wire_ok = [&](bool ok) { if (!ok) die("rmdup exchange between the shards failed (" + w->err + ")"); };
            uint64_t *d_cnt = nullptr;
            HIPCHK(hipMalloc(&d_cnt, 8));
            if (snk_rmdup_bucket_count_device(ctx, d_hash_all, (int64_t)nall, nglobal, d_cnt, nullptr) != SNK_OK) die(snk_last_error());
            HIPCHK(hipStreamSynchronize(0));
            wire_ok(w->allreduce_u64(d_cnt, 1, snk::WIRE_SUM));
            uint64_t sentinel_total = 0;
            HIPCHK(hipMemcpy(&sentinel_total, d_cnt, 8, hipMemcpyDeviceToHost));
            HIPCHK(hipFree(d_cnt));
            uint64_t *d_sh = nullptr, *d_rh = nullptr;
            uint32_t *d_si = nullptr, *d_slot = nullptr, *d_ri = nullptr;
            uint8_t *d_rf = nullptr, *d_back = nullptr;
            const size_t n1 = std::max<size_t>((size_t)nall, 1);
            HIPCHK(hipMalloc(&d_sh, n1 * 8));
            HIPCHK(hipMalloc(&d_si, n1 * 4));
            HIPCHK(hipMalloc(&d_slot, n1 * 4));
            std::vector<uint64_t> sc((size_t)W, 0), rc((size_t)W, 0);
            if (snk_rmdup_partition_device(ctx, d_hash_all, (int64_t)nall, gfirst, W, d_sh, d_si, d_slot, sc.data(), nullptr) != SNK_OK) die(snk_last_error());
            wire_ok(w->exchange_counts(sc.data(), rc.data()));
            uint64_t nrecv = 0;
            for (uint64_t v : rc) nrecv += v;
            if (getenv("SNK_SHARD_DEBUG")) { fprintf(stderr, "shard %d: n %llu total %llu first %llu send", w->rank, (unsigned long long)nall, (unsigned long long)nglobal, (unsigned long long)gfirst); for (uint64_t v : sc) fprintf(stderr, " %llu", (unsigned long long)v); fprintf(stderr, " recv"); for (uint64_t v : rc) fprintf(stderr, " %llu", (unsigned long long)v); fprintf(stderr, " sentinel %llu\n", (unsigned long long)sentinel_total); }
            const size_t r1 = std::max<size_t>((size_t)nrecv, 1);
            HIPCHK(hipMalloc(&d_rh, r1 * 8));
            HIPCHK(hipMalloc(&d_ri, r1 * 4));
            HIPCHK(hipMalloc(&d_rf, r1));
            HIPCHK(hipMalloc(&d_back, n1));
            wire_ok(w->alltoallv(d_sh, sc.data(), d_rh, rc.data(), 8));
            wire_ok(w->alltoallv(d_si, sc.data(), d_ri, rc.data(), 4));
            HIPCHK(hipFree(d_sh));
            HIPCHK(hipFree(d_si));
            if (nrecv && snk_rmdup_mark_device(ctx, d_rh, d_ri, (int64_t)nrecv, nglobal, (int64_t)sentinel_total, d_rf, nullptr) != SNK_OK) die(snk_last_error());
            HIPCHK(hipStreamSynchronize(0));
            wire_ok(w->alltoallv(d_rf, rc.data(), d_back, sc.data(), 1));
            if (snk_rmdup_flags_home_device(ctx, d_back, d_slot, (int64_t)nall, d_dup_all, nullptr) != SNK_OK) die(snk_last_error());
            HIPCHK(hipStreamSynchronize(0));
            HIPCHK(hipFree(d_rh)); HIPCHK(hipFree(d_ri)); HIPCHK(hipFree(d_rf)); HIPCHK(hipFree(d_back)); HIPCHK(hipFree(d_slot));
            if (mates == 1) {
                // SE: read i of a full patch is filtered with the flag of read i - 1 (below): the first read of a shard needs the
                // last flag of the shard in front -- every shard puts its last true flag into its own word of a G-word block
                uint64_t *d_last = nullptr;
                HIPCHK(hipMalloc(&d_last, (size_t)W * 8));
                std::vector<uint64_t> last((size_t)W, 0);
                if (nall) { uint8_t f = 0; HIPCHK(hipMemcpy(&f, d_dup_all + nall - 1, 1, hipMemcpyDeviceToHost)); last[(size_t)w->rank] = f; }
                HIPCHK(hipMemcpy(d_last, last.data(), (size_t)W * 8, hipMemcpyHostToDevice));
                wire_ok(w->allreduce_u64(d_last, (size_t)W, snk::WIRE_SUM));
                HIPCHK(hipMemcpy(last.data(), d_last, (size_t)W * 8, hipMemcpyDeviceToHost));
                HIPCHK(hipFree(d_last));
                if (w->rank > 0) prev_flag = (uint8_t)last[(size_t)w->rank - 1];       // (no shard is empty: the parent only shards inputs of G x 4096 records and more)
            }
        }
        HIPCHK(hipFree(d_hash_all));
        std::vector<uint8_t> flags((size_t)nall);
        if (nall) HIPCHK(hipMemcpy(flags.data(), d_dup_all, (size_t)nall, hipMemcpyDeviceToHost));
        uint64_t ndup = 0;
        for (uint8_t f : flags) ndup += f;
        log << "duplicate reads number:\t" << ndup << endl;
        if (mates == 1) {
            // Reference quirk (SE only): seProcess records "reads so far" BEFORE counting the quality line of
            // the patch's last read (src/seprocess.cpp:1086,1159 vs src/peprocess.cpp:2147), so inside every
            // full patch read i is filtered with the flag of read i-1 (read 0: the byte in front of the
            // array, 0 in practice); only the partial patch at the end of the file (:1112) is aligned.
            // Reproduced here, on the host, so that outputs stay identical; the C ABI flags are the true ones.
            // (indices are global: a shard shifts its own stretch of the input and takes read first - 1's flag from its neighbour)
            const uint64_t ps = o.patch_size > 0 ? (uint64_t)o.patch_size : (uint64_t)T * 20000 / 8;
            const uint64_t full_end = nglobal / ps * ps;
            std::vector<uint8_t> eff(flags);
            for (uint64_t i = 0; i < nall && gfirst + i < full_end; ++i) eff[i] = i ? flags[i - 1] : (gfirst ? prev_flag : 0);
            if (nall) HIPCHK(hipMemcpy(d_dup_all, eff.data(), (size_t)nall, hipMemcpyHostToDevice));
            flags.swap(eff);
        }
        dup_host.swap(flags);
        for (int m = 0; m < mates; ++m) {                   // dupReads.<thread>.<mate>.gz, src/peprocess.cpp:167-174 (SE: .1.gz only)
            dupw[m].resize(T);
            for (int t = 0; t < T; ++t) dupw[m][t].open(o.out_dir + "/dupReads." + std::to_string(t) + "." + std::to_string(m + 1) + ".gz" + (g_shard.child ? ".part" + std::to_string(g_shard.g) : string()), true);
        }
        for (auto &t : slot_makers) t.join();
        slot_makers.clear();
        for (Dev &d : devs) for (Slot &sl : d.slots) for (int m = 0; m < mates; ++m) memset(sl.h_seq[m], 0, plane);
        for (Dev &d : devs) for (Slot &sl : d.slots) { sl.raw[0] = sl.raw[1] = nullptr; }
        // second pass over the input: from the kept batches, or through the readers again
        if (cache_on && !cached[0].empty()) {
            log << local_time() << "\trmdup: " << cached[0].size() << " inflated batches kept for the main pass (" << (cache_bytes >> 20) << " MB)" << endl;
            for (int m = 0; m < mates; ++m) {
                chan[m].reset(new Channel<RawChunk *>(2));
                readers.emplace_back([&, m, list = std::move(cached[m])] { for (RawChunk *q : list) chan[m]->push(q); chan[m]->close(); });
            }
        } else {
            start_readers();
        }
        if (!next_chunks(first)) die("no data");
    }

    // ---- main pass: pack+GPU stage (this thread) and write stage (its own thread), NSLOT batches in flight
    OutFile wr[2];
    // limited clean output: -w writes split.<k>.<cleanFq> files of clean_out_split reads (the next file is created
    // the moment one is full, src/peprocess.cpp:2474-2560,2772-2870); "<N>head" keeps the first N clean reads
    // (:2960-2985); both cut inside a worker's slice, so the slices are compressed after cutting
    const uint64_t split_n = o.clean_out_split, head_n = o.total_head ? o.total_num : 0;
    const bool cut_mode = split_n > 0 || head_n > 0;
    const string clean_name[2] = {o.clean1, o.clean2};
    uint64_t split_idx = 0, in_split = 0, clean_written = 0, clean_total = 0;
    auto open_clean = [&](uint64_t k) {
        for (int m = 0; m < mates; ++m) {
            wr[m].close();
            wr[m].open(split_n ? o.out_dir + "/split." + std::to_string(k) + "." + clean_name[m] : o.out_dir + "/" + clean_name[m], o.out_gz);
        }
    };
    if (!split_n) open_clean(0);
    // trimFq1/trimFq2: every read after trimming, before the discard cascade (src/peprocess.cpp:1460-1466,1938-1944)
    const bool trim_out = !o.trim_fq[0].empty();
    OutFile trimw[2];
    if (trim_out) {
        if (mates == 2 && o.trim_fq[1].empty()) die("trimFq2 is required with trimFq1");
        for (int m = 0; m < mates; ++m) trimw[m].open(o.out_dir + "/" + o.trim_fq[m], o.trim_fq_gz[m] >= 0 ? o.trim_fq_gz[m] != 0 : ends_with_gz(o.trim_fq[m]));
    }
    Channel<Slot *> to_write((size_t)NSLOT * (size_t)G + 1);
    const bool rmdup_on = !dup_host.empty();
    const int dq = o.p.output_quality_phred - o.p.quality_phred;
    const bool fasta = o.out_file_type == "fasta";
    uint64_t ndup_written = 0;
    // device-text mode: the clean text of a batch arrives whole (h_text, record offsets in h_outoff); plain output is written
    // as it is, .gz output is cut at record boundaries into one gzip member per worker
    uint8_t *dup_stage[2] = {nullptr, nullptr};            // (pinned landing buffers of the duplicates' text, grown on demand)
    size_t dup_stage_cap[2] = {0, 0};
    auto writer_dev = [&] {
        Slot *sp;
        std::vector<string> zbuf[2];
        for (;;) {
            { Tick t_(7); if (!to_write.pop(sp)) break; }
            Slot &s = *sp;
            { Tick t_(8); HIPCHK(hipEventSynchronize(s.done)); }
            if (*s.h_err != SNK_ERR_WORD_NONE) {
                snk_error err;
                snk_error_decode(*s.h_err, &err);
                report_device_error(err);
            }
            const int n = s.n;
            if (dev_gz) {
                // the members are in HBM: fetch exactly their bytes (h_text is free again: the upload is long done), write them
                Tick t_write_(10);
                HIPCHK(hipSetDevice(devs[(size_t)s.dev].id));
                bool overflow = false;
                for (int m = 0; m < mates; ++m) overflow = overflow || s.h_gzinfo[m][2] != 0;
                if (overflow) {                             // (the compressed text did not fit its buffer: incompressible input) host encoder
                    for (int m = 0; m < mates; ++m) {
                        HIPCHK(hipMemcpy(s.h_outoff[m], s.d_outoff[m], ((size_t)n + 1) * 4, hipMemcpyDeviceToHost));
                        HIPCHK(hipMemcpy(s.h_text[m], s.d_out[m], s.h_outoff[m][n], hipMemcpyDeviceToHost));
                        string z;
                        gzip_member((const char *)s.h_text[m], s.h_outoff[m][n], z);
                        wr[m].write_bytes(z);
                    }
                } else {
                    for (int m = 0; m < mates; ++m)
                        if (s.h_gzinfo[m][0]) HIPCHK(hipMemcpyAsync(s.h_text[m], s.d_gz[m], s.h_gzinfo[m][0], hipMemcpyDeviceToHost, s.stream));
                    HIPCHK(hipStreamSynchronize(s.stream));
                    struct Piece { int fd; const char *p; size_t n; off_t at; };
                    std::vector<Piece> pieces;
                    for (int m = 0; m < mates; ++m) {
                        const size_t tot = s.h_gzinfo[m][0];
                        const int np = tot < ((size_t)8 << 20) ? 1 : 4;
                        for (int k = 0; k < np; ++k) {
                            const size_t a = tot * (size_t)k / (size_t)np, b = tot * (size_t)(k + 1) / (size_t)np;
                            if (b > a) pieces.push_back(Piece{wr[m].fd, (const char *)s.h_text[m] + a, b - a, wr[m].pos + (off_t)a});
                        }
                        wr[m].pos += (off_t)tot;
                    }
                    const bool serial = OutFile::is_stream(wr[0].fd) || (mates == 2 && OutFile::is_stream(wr[1].fd));
                    parallel_for(serial ? 1 : (int)pieces.size(), (int)pieces.size(), [&](int, int lo, int hi) {
                        for (int k = lo; k < hi; ++k) OutFile::put_at(pieces[(size_t)k].fd, pieces[(size_t)k].p, pieces[(size_t)k].n, pieces[(size_t)k].at);
                    });
                }
            } else if (o.out_gz) {
                const long long t_fmt0 = g_clk.on ? StageClock::now() : 0;
                for (int m = 0; m < mates; ++m) { zbuf[m].resize(WK); for (int w = 0; w < WK; ++w) zbuf[m][w].clear(); }
                parallel_for(WK, mates * WK, [&](int, int lo, int hi) {
                    for (int k = lo; k < hi; ++k) {
                        const int m = k / WK, w = k % WK;
                        const uint32_t a = s.h_outoff[m][(size_t)((long)n * w / WK)], b = s.h_outoff[m][(size_t)((long)n * (w + 1) / WK)];
                        if (b > a) gzip_member((const char *)s.h_text[m] + a, (size_t)(b - a), zbuf[m][w]);
                    }
                });
                if (g_clk.on) g_clk.ns[9] += StageClock::now() - t_fmt0;
                Tick t_write_(10);
                for (int m = 0; m < mates; ++m) wr[m].write_parts(zbuf[m]);
            } else {
                Tick t_write_(10);
                // both files at once, a few writers each (page allocation of one file does not scale past a few threads)
                struct Piece { int fd; const char *p; size_t n; off_t at; };
                std::vector<Piece> pieces;
                for (int m = 0; m < mates; ++m) {
                    const size_t tot = s.h_outoff[m][n];
                    const int np = tot < ((size_t)8 << 20) ? 1 : 6;
                    for (int k = 0; k < np; ++k) {
                        const size_t a = tot * (size_t)k / (size_t)np, b = tot * (size_t)(k + 1) / (size_t)np;
                        if (b > a) pieces.push_back(Piece{wr[m].fd, (const char *)s.h_text[m] + a, b - a, wr[m].pos + (off_t)a});
                    }
                    wr[m].pos += (off_t)tot;
                }
                const bool serial = OutFile::is_stream(wr[0].fd) || (mates == 2 && OutFile::is_stream(wr[1].fd));
                parallel_for(serial ? 1 : (int)pieces.size(), (int)pieces.size(), [&](int, int lo, int hi) {
                    for (int k = lo; k < hi; ++k) OutFile::put_at(pieces[(size_t)k].fd, pieces[(size_t)k].p, pieces[(size_t)k].n, pieces[(size_t)k].at);
                });
            }
            if (rmdup_stream && s.h_dupoff[0][n]) {
                // the duplicates' raw text is in HBM, one run per mate; cut like the host formatter cuts it: a gzip member per
                // (worker slice, virtual thread) run, appended to that thread's side file in input order
                HIPCHK(hipSetDevice(devs[(size_t)s.dev].id));
                for (int m = 0; m < mates; ++m) {
                    const size_t tot = s.h_dupoff[m][n];
                    if (tot > dup_stage_cap[m]) {
                        if (dup_stage[m]) HIPCHK(hipHostFree(dup_stage[m]));
                        dup_stage_cap[m] = tot + tot / 4 + 4096;
                        HIPCHK(hipHostMalloc(&dup_stage[m], dup_stage_cap[m]));
                    }
                    HIPCHK(hipMemcpyAsync(dup_stage[m], s.d_dupout[m], tot, hipMemcpyDeviceToHost, s.stream));
                }
                HIPCHK(hipStreamSynchronize(s.stream));
                struct DupRun { int vt; uint32_t a[2], b[2]; string z[2]; };
                std::vector<std::vector<DupRun>> runs((size_t)WK);
                std::vector<uint64_t> cnt((size_t)WK, 0);
                parallel_for(WK, n, [&](int w, int lo, int hi) {
                    int cur = -1;
                    for (int i = lo; i < hi; ++i) {
                        if (s.h_dupoff[0][i + 1] == s.h_dupoff[0][i]) continue;
                        const int vt = (int)(((s.first + (uint64_t)i) / (uint64_t)vblock) % (uint64_t)T);
                        if (vt != cur) {
                            runs[(size_t)w].emplace_back();
                            runs[(size_t)w].back().vt = cur = vt;
                            for (int m = 0; m < mates; ++m) runs[(size_t)w].back().a[m] = s.h_dupoff[m][i];
                        }
                        for (int m = 0; m < mates; ++m) runs[(size_t)w].back().b[m] = s.h_dupoff[m][i + 1];
                        ++cnt[(size_t)w];
                    }
                    for (DupRun &r : runs[(size_t)w])
                        for (int m = 0; m < mates; ++m) gzip_member((const char *)dup_stage[m] + r.a[m], (size_t)(r.b[m] - r.a[m]), r.z[m]);
                });
                for (int w = 0; w < WK; ++w) {
                    for (const DupRun &r : runs[(size_t)w])
                        for (int m = 0; m < mates; ++m) dupw[m][(size_t)r.vt].write_bytes(r.z[m]);
                    ndup_written += cnt[(size_t)w];
                }
            }
            log << local_time() << " processed_reads:\t" << s.first + (uint64_t)n << endl;
            devs[(size_t)s.dev].free_slots->push(sp);
        }
    };
    std::thread writer = dev_text ? std::thread(writer_dev) : std::thread([&] {
        Slot *sp;
        std::vector<string> text[2], zbuf[2], ttext[2], tzbuf[2];
        std::vector<std::vector<uint32_t>> recoff[2];          // cut_mode: start of every kept record in text[m][w]
        std::vector<uint64_t> kcount;
        struct DupPiece { int vt; string z[2]; };            // one gzip member per (worker slice, virtual thread, mate)
        std::vector<std::vector<DupPiece>> dpieces;
        std::vector<uint64_t> dcount;
        for (;;) {
            { Tick t_(7); if (!to_write.pop(sp)) break; }
            Slot &s = *sp;
            { Tick t_(8); HIPCHK(hipEventSynchronize(s.done)); }
            if (*s.h_err != SNK_ERR_WORD_NONE) {                    // the reference exits at the offending read: nothing of this batch is written
                snk_error err;
                snk_error_decode(*s.h_err, &err);
                report_device_error(err);
            }
            const int n = s.n, lcap = s.lcap;                  // (the capacity this batch was packed with)
            for (int m = 0; m < mates; ++m) {               // (buffers keep their capacity from batch to batch: no page faults after the first)
                text[m].resize(WK); zbuf[m].resize(WK); ttext[m].resize(WK); tzbuf[m].resize(WK); recoff[m].resize(WK);
                for (int w = 0; w < WK; ++w) { text[m][w].clear(); zbuf[m][w].clear(); ttext[m][w].clear(); tzbuf[m][w].clear(); recoff[m][w].clear(); }
            }
            dpieces.assign(WK, std::vector<DupPiece>());
            dcount.assign(WK, 0);
            kcount.assign(WK, 0);
            // clean output, input order (src/peprocess.cpp:3383-3484): every worker formats (and deflates) a slice
            const long long t_fmt0 = g_clk.on ? StageClock::now() : 0;
            parallel_for(WK, n, [&](int w, int lo, int hi) {
                // one record of mate m in output form; pe_times: how often preOutput ran on the object
                // (twice for clean reads when the trim files are on)
                auto put = [&](string &out, int m, int i, int pe_times) {
                    const snk_read_result &x = s.h_rec[m][i];
                    int li, lsq, lql;
                    const char *id = s.raw[m]->line(4 * i, li), *sq = s.raw[m]->line(4 * i + 1, lsq), *ql = s.raw[m]->line(4 * i + 3, lql);
                    const bool stream_rec = o.streaming && !fasta;        // ">+\t<id without its first character>\t<mate>\t<seq>\t<qual>" (src/peprocess.cpp:3402-3411)
                    if (stream_rec) out += ">+\t";
                    const size_t id_at = out.size();
                    if (!o.index_remove) out.append(id, li);
                    else if (o.seq_type == "0") {                 // "@FC:4:1101:1799:2201#GAAGCACG/2": drop '#'..before '/' (src/read_filter.cpp:357-378)
                        bool cp = true;
                        for (int k = 0; k < li; ++k) {
                            if (id[k] == '#') cp = false;
                            if (cp) out += id[k];
                            else if (id[k] == '/') { cp = true; out += id[k]; }
                        }
                    } else {                                       // new style: cut at the last ':' (:379-381)
                        int cut = li;                                  // no ':' at all: substr(0, npos) keeps the whole id
                        for (int k = li - 1; k >= 0; --k) if (id[k] == ':') { cut = k; break; }
                        out.append(id, cut);
                    }
                    if (o.pe_info && mates == 2)                  // preOutput, src/peprocess.cpp:1617-1628 (seProcess::preOutput has no such step)
                        for (int t = 0; t < pe_times; ++t) out += (m == 0 ? "/1" : "/2");
                    if (fasta) {
                        const size_t at = out.find('@', id_at);
                        if (at != string::npos) out[at] = '>';
                    }
                    if (stream_rec) {
                        if (out.size() > id_at) out.erase(id_at, 1);
                        out += '\t';
                        out += (m == 0 ? '1' : '2');
                        out += '\t';
                    } else {
                        out += '\n';
                    }
                    const size_t sq_at = out.size();
                    const int cs = std::min<int>(x.clean_start, lsq), cl = std::min<int>(x.clean_len, lsq - cs);
                    out.append(sq + cs, cl);
                    if (bc_from)
                        for (size_t k = sq_at; k < out.size(); ++k) if (toupper((unsigned char)out[k]) == bc_from) out[k] = bc_to;
                    if (fasta) { out += '\n'; return; }
                    out += stream_rec ? "\t" : "\n+\n";
                    const size_t q_at = out.size();
                    out.append(ql + cs, cl);
                    if (dq) for (size_t k = q_at; k < out.size(); ++k) out[k] = (char)(out[k] + dq);
                    out += '\n';
                };
                for (int m = 0; m < mates; ++m) {
                    string &out = text[m][w];
                    out.reserve((size_t)(hi - lo) * (size_t)(2 * lcap + 64));
                    for (int i = lo; i < hi; ++i)
                        if (s.h_rec[0][i].reason == SNK_KEEP) {
                            if (m == 0) ++kcount[w];
                            if (cut_mode) recoff[m][w].push_back((uint32_t)out.size());
                            put(out, m, i, trim_out ? 2 : 1);
                        }
                    if (trim_out) {
                        string &tout = ttext[m][w];
                        tout.reserve((size_t)(hi - lo) * (size_t)(2 * lcap + 64));
                        for (int i = lo; i < hi; ++i) put(tout, m, i, 1);
                        if (trimw[m].gz && !tout.empty()) gzip_member(tout, tzbuf[m][w]);
                    }
                    if (o.out_gz && !cut_mode && !o.streaming && !out.empty()) gzip_member(out, zbuf[m][w]);
                }
                if (rmdup_on) {                                // C_fastq::toString of the raw records, src/peprocess.cpp:1541
                    string acc[2];
                    int cur_vt = -1;
                    auto flush_piece = [&] {
                        if (cur_vt < 0 || acc[0].empty()) return;
                        dpieces[w].emplace_back();
                        DupPiece &pc = dpieces[w].back();
                        pc.vt = cur_vt;
                        for (int m = 0; m < mates; ++m) { gzip_member(acc[m], pc.z[m]); acc[m].clear(); }
                    };
                    for (int i = lo; i < hi; ++i) {
                        if (s.h_rec[0][i].reason != SNK_R_DUP) continue;
                        const int vt = (int)(((s.first + (uint64_t)i) / (uint64_t)vblock) % (uint64_t)T);
                        if (vt != cur_vt) { flush_piece(); cur_vt = vt; }
                        for (int m = 0; m < mates; ++m) {
                            int li, lsq, lql;
                            const char *id = s.raw[m]->line(4 * i, li), *sq = s.raw[m]->line(4 * i + 1, lsq), *ql = s.raw[m]->line(4 * i + 3, lql);
                            string &out = acc[m];
                            out.append(id, li); out += '\n'; out.append(sq, lsq); out += "\n+\n"; out.append(ql, lql); out += '\n';
                        }
                        ++dcount[w];
                    }
                    flush_piece();
                }
            });
            if (g_clk.on) g_clk.ns[9] += StageClock::now() - t_fmt0;
            Tick t_write_(10);
            for (int w = 0; w < WK; ++w) clean_total += kcount[w];
            if (o.streaming) {
                // output_fastqs("1", ...), output_fastqs("2", ...), then the thread's statistics (src/peprocess.cpp:1952-1976);
                // with outFileType=fasta the reference prints no reads at all (they go to a string nobody writes)
                if (!fasta)
                    for (int m = 0; m < mates; ++m)
                        for (int w = 0; w < WK; ++w) fwrite(text[m][w].data(), 1, text[m][w].size(), stdout);
                string st;
                snk_streaming_stat_text(&o.p, s.snap_sum.data(), s.snap_max.data(), &st);
                fwrite(st.data(), 1, st.size(), stdout);
                fflush(stdout);
            } else if (!cut_mode) {
                for (int m = 0; m < mates; ++m) wr[m].write_parts(wr[m].gz ? zbuf[m] : text[m]);
            } else {
                for (int w = 0; w < WK; ++w) {
                    const size_t kept = recoff[0][w].size();
                    size_t pos = 0;
                    while (pos < kept) {
                        if (head_n && clean_written >= head_n) break;
                        if (split_n && !wr[0].is_open()) open_clean(split_idx);        // the first split file appears with the first clean read
                        const uint64_t room = split_n ? split_n - in_split : head_n - clean_written;
                        const size_t take = (size_t)std::min<uint64_t>(room, kept - pos);
                        for (int m = 0; m < mates; ++m) {
                            const size_t b = recoff[m][w][pos], e = pos + take < kept ? recoff[m][w][pos + take] : text[m][w].size();
                            wr[m].write_text(text[m][w].substr(b, e - b));
                        }
                        pos += take;
                        clean_written += take;
                        in_split += take;
                        if (split_n && in_split == split_n) { open_clean(++split_idx); in_split = 0; }
                    }
                }
            }
            if (trim_out)
                for (int m = 0; m < mates; ++m) trimw[m].write_parts(trimw[m].gz ? tzbuf[m] : ttext[m]);
            if (rmdup_on)
                for (int w = 0; w < WK; ++w) {                 // worker order = input order within every side file
                    for (const DupPiece &pc : dpieces[w])
                        for (int m = 0; m < mates; ++m) dupw[m][pc.vt].write_bytes(pc.z[m]);
                    ndup_written += dcount[w];
                }
            log << local_time() << " processed_reads:\t" << s.first + (uint64_t)n << endl;
            for (int m = 0; m < mates; ++m) { RawChunk::put(s.raw[m]); s.raw[m] = nullptr; }
            devs[(size_t)s.dev].free_slots->push(sp);
        }
    });

    uint64_t total = g_shard.first, batch_no = 0;         // (a shard: the global index of its first pair)
    RawChunk *c[2] = {first[0], first[1]};
    bool have = true;
    if (dev_text) {
        snk_fastq_format fmt[2];
        for (int m = 0; m < 2; ++m) {
            memset(&fmt[m], 0, sizeof fmt[m]);
            fmt[m].struct_size = (int32_t)sizeof(snk_fastq_format);
            fmt[m].space_num = space_num;
            fmt[m].qual_delta = dq;
            fmt[m].id_suffix_times = (o.pe_info && mates == 2) ? 1 : 0;
            fmt[m].id_suffix[0] = '/'; fmt[m].id_suffix[1] = m == 0 ? '1' : '2';
            fmt[m].base_from = (uint8_t)bc_from; fmt[m].base_to = (uint8_t)bc_to;
        }
        snk_fastq_format fmt_dup;                             // C_fastq::toString of the raw records, src/peprocess.cpp:1541
        memset(&fmt_dup, 0, sizeof fmt_dup);
        fmt_dup.struct_size = (int32_t)sizeof(snk_fastq_format);
        fmt_dup.space_num = space_num;
        fmt_dup.select_reason = SNK_R_DUP;
        fmt_dup.whole_read = 1;
        // raw text -> pinned -> device; line index + SoA planes by the device (asynchronous; s.parsed marks the end)
        auto stage = [&](Slot &s, RawChunk *const ch[2]) {
            s.n = ch[0]->n;
            s.lcap = lcap;
            {
                Tick t_(5);
                const size_t step = (size_t)4 << 20;
                struct Cp { uint8_t *d; const char *p; size_t n; };
                std::vector<Cp> cps;
                for (int m = 0; m < mates; ++m) {
                    s.nbytes[m] = ch[m]->nbytes;
                    for (size_t a = 0; a < ch[m]->nbytes; a += step) cps.push_back(Cp{s.h_text[m] + a, ch[m]->base + a, std::min(step, ch[m]->nbytes - a)});
                }
                parallel_for(WK, (int)cps.size(), [&](int, int lo, int hi) { for (int k = lo; k < hi; ++k) memcpy(cps[(size_t)k].d, cps[(size_t)k].p, cps[(size_t)k].n); });
            }
            for (int m = 0; m < mates; ++m) {
                HIPCHK(hipMemcpyAsync(s.d_text[m], s.h_text[m], s.nbytes[m], hipMemcpyHostToDevice, s.stream));
                if (snk_fastq_parse_device(s.d_text[m], s.nbytes[m], s.n, space_num, pitch, lcap, s.d_seq[m], s.d_qual[m], s.d_len[m], s.d_line[m],
                                           s.d_status[m], s.d_tmp, s.tmp_bytes, s.stream) != SNK_OK) die(snk_last_error());
                HIPCHK(hipMemcpyAsync(s.h_status[m], s.d_status[m], SNK_FQ_STATUS_N * 4, hipMemcpyDeviceToHost, s.stream));
            }
            HIPCHK(hipEventRecord(s.parsed, s.stream));
        };
        // parse verdicts, then filter kernels, clean text, copies back; 0 = submitted, else the length of a read longer than the capacity
        auto submit = [&](Slot &s) -> int {
            Dev &dv = devs[(size_t)s.dev];
            HIPCHK(hipSetDevice(dv.id));
            { Tick t_(6); HIPCHK(hipEventSynchronize(s.parsed)); }
            int too_long = 0;
            for (int m = 0; m < mates; ++m) {
                const uint32_t fl = s.h_status[m][SNK_FQ_ST_FLAGS];
                if (fl & SNK_FQ_F_TRUNCATED) die("truncated fastq record");
                if (fl & SNK_FQ_F_TOO_LONG) too_long = std::max<int>(too_long, (int)s.h_status[m][SNK_FQ_ST_MAXLEN]);
                else if (fl & SNK_FQ_F_LEN_MISMATCH) die("sequence and quality lengths differ");
            }
            if (too_long) return too_long;
            const int n = s.n;
            if (rmdup_stream) {                              // hash the raw pairs, look them up in (and add them to) the resident table
                snk_batch hb;
                memset(&hb, 0, sizeof hb);
                hb.n = n;
                hb.pitch = pitch;
                for (int m = 0; m < mates; ++m) { hb.seq[m] = s.d_seq[m]; hb.qual[m] = s.d_qual[m]; hb.len[m] = s.d_len[m]; }
                if (snk_rmdup_hash_device(dv.ctx, &hb, s.d_hash, s.stream) != SNK_OK) die(snk_last_error());
                // (single end: every batch but the file's last is whole patches -- a shorter one can only be the last)
                auto mark = [&](const uint64_t *dh, uint8_t *df, hipStream_t st) {
                    return mates == 2 ? snk_rmdup_stream_mark_device(dup_table, dh, s.first, n, df, st)
                                      : snk_rmdup_stream_mark_se_device(dup_table, dh, s.first, n, (int64_t)n / rmdup_patch * rmdup_patch, df, st);
                };
                int mrc;
                if (!s.d_hash0) mrc = mark(s.d_hash, s.d_flags, s.stream);
                else {                                       // the table is on the first device
                    HIPCHK(hipEventRecord(s.hashed, s.stream));
                    HIPCHK(hipSetDevice(devs[0].id));
                    HIPCHK(hipStreamWaitEvent(mark_stream, s.hashed, 0));
                    HIPCHK(hipMemcpyPeerAsync(s.d_hash0, devs[0].id, s.d_hash, dv.id, (size_t)n * sizeof(uint64_t), mark_stream));
                    mrc = mark(s.d_hash0, s.d_flags0, mark_stream);
                    if (mrc == SNK_OK) {
                        HIPCHK(hipMemcpyPeerAsync(s.d_flags, dv.id, s.d_flags0, devs[0].id, (size_t)n, mark_stream));
                        HIPCHK(hipEventRecord(s.marked, mark_stream));
                    }
                    HIPCHK(hipSetDevice(dv.id));
                    if (mrc == SNK_OK) HIPCHK(hipStreamWaitEvent(s.stream, s.marked, 0));
                }
                if (mrc == SNK_E_NOMEM)                      // (the estimate below was short by more than a factor of two)
                    die(string(snk_last_error()) + ": run again with SNK_RMDUP_TWO_PASS=1 in the environment (the memory-lean two passes)");
                if (mrc != SNK_OK) die(snk_last_error());
            }
            for (int lo = 0; lo < n;) {                      // split at virtual-thread block boundaries (appendix C)
                const uint64_t g = s.first + (uint64_t)lo;
                const int vt = (int)((g / (uint64_t)vblock) % (uint64_t)T);
                const uint64_t next = (g / (uint64_t)vblock + 1) * (uint64_t)vblock;
                const int hi = (int)std::min<uint64_t>((uint64_t)n, next - s.first);
                snk_batch b;
                memset(&b, 0, sizeof b);
                b.n = hi - lo;
                b.pitch = pitch;
                for (int m = 0; m < mates; ++m) {
                    b.seq[m] = s.d_seq[m] + (size_t)lo * pitch;
                    b.qual[m] = s.d_qual[m] + (size_t)lo * pitch;
                    b.len[m] = s.d_len[m] + lo;
                }
                b.first_index = g;
                if (rmdup_stream) b.dup = s.d_flags + lo;
                if (snk_bind_stats(dv.ctx, dv.d_sum[(size_t)vt], dv.d_max[(size_t)vt]) != SNK_OK) die(snk_last_error());
                if (snk_filter_batch_device(dv.ctx, &b, s.d_rec[0] + lo, mates == 2 ? s.d_rec[1] + lo : nullptr, s.stream, 0) != SNK_OK)
                    die(snk_last_error());
                lo = hi;
            }
            for (int m = 0; m < mates && rmdup_stream; ++m) {  // the duplicate pairs' raw records, for the side files
                if (snk_fastq_format_device(s.d_text[m], s.d_line[m], s.d_rec[0], s.d_rec[m], n, &fmt_dup, s.d_dupout[m], s.d_dupoff[m], s.d_tmp, s.tmp_bytes,
                                            s.stream) != SNK_OK) die(snk_last_error());
                HIPCHK(hipMemcpyAsync(s.h_dupoff[m], s.d_dupoff[m], ((size_t)n + 1) * 4, hipMemcpyDeviceToHost, s.stream));
            }
            for (int m = 0; m < mates; ++m) {
                if (snk_fastq_format_device(s.d_text[m], s.d_line[m], s.d_rec[0], s.d_rec[m], n, &fmt[m], s.d_out[m], s.d_outoff[m], s.d_tmp, s.tmp_bytes,
                                            s.stream) != SNK_OK) die(snk_last_error());
                if (dev_gz) {                               // gzip members on the device: only their total size travels now, the bytes when the writer knows it
                    static const bool tiny_cap = getenv("SNK_GZ_CAP_TEST") != nullptr;      // tests: the members never fit -> the host encoder takes every batch
                    if (snk_fastq_deflate_device(s.d_out[m], s.d_outoff[m], n, GZ_RPM, s.d_gz[m], tiny_cap ? 4096 : text_cap, s.d_gzinfo[m], s.d_ztmp, s.ztmp_bytes,
                                                 s.stream) != SNK_OK) die(snk_last_error());
                    HIPCHK(hipMemcpyAsync(s.h_gzinfo[m], s.d_gzinfo[m], 16, hipMemcpyDeviceToHost, s.stream));
                    continue;
                }
                HIPCHK(hipMemcpyAsync(s.h_outoff[m], s.d_outoff[m], ((size_t)n + 1) * 4, hipMemcpyDeviceToHost, s.stream));
                // the clean text is at most the input text (+ the pe_info suffixes): that much is copied, its real size is h_outoff[n]
                const size_t bound = std::min(text_cap + 64, s.nbytes[m] + (size_t)n * 2 * (size_t)fmt[m].id_suffix_times + 1);   // (+ 1: the newline a ragged last line did not have)
                HIPCHK(hipMemcpyAsync(s.h_text[m], s.d_out[m], bound, hipMemcpyDeviceToHost, s.stream));
            }
            if (snk_error_peek_async(dv.ctx, s.h_err, s.stream) != SNK_OK) die(snk_last_error());
            HIPCHK(hipEventRecord(s.done, s.stream));
            to_write.push(&s);
            return 0;
        };
        struct Held { Slot *sp = nullptr; RawChunk *raw[2] = {nullptr, nullptr}; uint64_t first = 0; };
        Held pend;                                            // staged, its parse verdict not looked at yet
        auto drain_all = [&] { for (Dev &d : devs) { Slot *t_[8]; for (int k = 0; k < NSLOT; ++k) d.free_slots->pop(t_[k]); for (int k = 0; k < NSLOT; ++k) d.free_slots->push(t_[k]); } };
        auto take_slot = [&](uint64_t no) -> Slot * {
            Dev &dv = devs[(size_t)(no % (uint64_t)G)];
            HIPCHK(hipSetDevice(dv.id));
            Slot *sp;
            { Tick t_(4); dv.free_slots->pop(sp); }
            return sp;
        };
        // a batch whose text does not fit the slots, or a read longer than the capacity: everything in flight is finished, the
        // epoch closed (capacity), buffers rebuilt, and the held batches run again
        auto rebuild = [&](int new_lcap, size_t new_text_cap, std::vector<Held> &again) {
            for (Held &h : again) if (h.sp) { devs[(size_t)h.sp->dev].free_slots->push(h.sp); h.sp = nullptr; }
            drain_all();
            collect_epoch();                                  // (teardown drops the accumulators: whatever was counted so far becomes an epoch)
            teardown();
            text_cap = std::max(text_cap, new_text_cap);
            setup(std::max<int>(lcap, new_lcap));
        };
        auto run_sync = [&](Held &h, uint64_t no) {           // stage + submit one held batch, growing the capacity as often as it takes
            for (;;) {
                Slot *sp = take_slot(no);
                sp->first = h.first;
                stage(*sp, h.raw);
                const int tl = submit(*sp);
                if (!tl) break;
                std::vector<Held> one(1);
                one[0].sp = sp;
                rebuild(tl, 0, one);
            }
            for (int m = 0; m < mates; ++m) RawChunk::put(h.raw[m]);
            h = Held();
        };
        auto finish_pending = [&](Held *staged_behind, uint64_t no_pend) {
            if (!pend.sp) return;
            const int tl = submit(*pend.sp);
            if (!tl) { for (int m = 0; m < mates; ++m) RawChunk::put(pend.raw[m]); pend = Held(); return; }
            std::vector<Held> again;
            again.push_back(pend);
            if (staged_behind && staged_behind->sp) again.push_back(*staged_behind);
            rebuild(tl, 0, again);
            pend.sp = nullptr;
            run_sync(pend, no_pend);
            if (staged_behind && staged_behind->sp) { staged_behind->sp = nullptr; run_sync(*staged_behind, no_pend + 1); }
        };
        while (have) {
            size_t need = 0;
            for (int m = 0; m < mates; ++m) need = std::max(need, c[m]->nbytes + (size_t)c[m]->n * 2 + 64);
            if (need > text_cap) {                            // longer names than the first batch had: larger text buffers
                finish_pending(nullptr, batch_no - 1);
                std::vector<Held> none;
                rebuild(0, need + need / 8, none);
            }
            const uint64_t no = batch_no++;
            Held cur;
            cur.sp = take_slot(no);
            cur.raw[0] = c[0]; cur.raw[1] = c[1];
            cur.first = total;
            cur.sp->first = total;
            stage(*cur.sp, cur.raw);
            total += (uint64_t)c[0]->n;
            if (pend.sp) {
                finish_pending(&cur, no - 1);
                if (!cur.sp && !cur.raw[0]) { { Tick t_(3); have = next_chunks(c); } continue; }     // it ran again behind the rebuilt capacity
            }
            pend = cur;
            { Tick t_(3); have = next_chunks(c); }
        }
        finish_pending(nullptr, batch_no - 1);
    }
    while (have && !dev_text) {
        Dev &dv = devs[(size_t)(batch_no++ % (uint64_t)G)];   // batches go round the devices; the writer keeps input order
        HIPCHK(hipSetDevice(dv.id));
        Slot *sp;
        { Tick t_(4); dv.free_slots->pop(sp); }
        sp->n = c[0]->n;
        sp->first = total;
        sp->raw[0] = c[0]; sp->raw[1] = c[1];
        bool packed;
        { Tick t_(5); packed = pack(*sp, true); }
        if (!packed) {
            // a read longer than every read before it (the reference takes any read up to 1000 nt at any
            // position): drain the pipeline, close the epoch, rebuild contexts and slots with the new capacity
            dv.free_slots->push(sp);
            for (Dev &d : devs) { Slot *t_[8]; for (int k = 0; k < NSLOT; ++k) d.free_slots->pop(t_[k]); }
            collect_epoch();
            teardown();
            setup(longest(c));
            HIPCHK(hipSetDevice(dv.id));
            dv.free_slots->pop(sp);
            sp->n = c[0]->n;
            sp->first = total;
            sp->raw[0] = c[0]; sp->raw[1] = c[1];
            pack(*sp, true);
        }
        Slot &s = *sp;
        s.lcap = lcap;
        const int n = s.n;
        const bool name_verdicts = !o.tile.empty() || !o.fov.empty();
        const bool host_flags = name_verdicts || rmdup_on;
        if (host_flags) {                                   // tile / fov of fq1's read name (src/read_filter.cpp:86-150, src/sequence.cpp:213-231) + the duplicate bit
            parallel_for(WK, n, [&](int, int lo, int hi) {
                for (int i = lo; i < hi; ++i) {
                    uint8_t f = rmdup_on ? (uint8_t)(dup_host[total - g_shard.first + (uint64_t)i] & 1) : 0;
                    if (name_verdicts) {
                        int li;
                        const char *id = s.raw[0]->line(4 * i, li);
                        if (!o.tile.empty() && check_tile_or_fov(read_tile(id, li, o.seq_type), o.tile)) f |= 2;
                        if (!o.fov.empty() && check_tile_or_fov(read_fov(id, li), o.fov)) f |= 4;
                    }
                    s.h_flags[i] = f;
                }
            });
            HIPCHK(hipMemcpyAsync(s.d_flags, s.h_flags, (size_t)n, hipMemcpyHostToDevice, s.stream));
        }
        for (int m = 0; m < mates; ++m) {
            HIPCHK(hipMemcpyAsync(s.d_seq[m], s.h_seq[m], (size_t)n * pitch, hipMemcpyHostToDevice, s.stream));
            HIPCHK(hipMemcpyAsync(s.d_qual[m], s.h_qual[m], (size_t)n * pitch, hipMemcpyHostToDevice, s.stream));
            HIPCHK(hipMemcpyAsync(s.d_len[m], s.h_len[m], (size_t)n * 2, hipMemcpyHostToDevice, s.stream));
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
                b.seq[m] = s.d_seq[m] + (size_t)lo * pitch;
                b.qual[m] = s.d_qual[m] + (size_t)lo * pitch;
                b.len[m] = s.d_len[m] + lo;
            }
            b.first_index = g;
            if (host_flags) b.dup = s.d_flags + lo;
            if (snk_bind_stats(dv.ctx, dv.d_sum[(size_t)vt], dv.d_max[(size_t)vt]) != SNK_OK) die(snk_last_error());
            if (snk_filter_batch_device(dv.ctx, &b, s.d_rec[0] + lo, mates == 2 ? s.d_rec[1] + lo : nullptr, s.stream, 0) != SNK_OK)
                die(snk_last_error());
            lo = hi;
        }
        for (int m = 0; m < mates; ++m)
            HIPCHK(hipMemcpyAsync(s.h_rec[m], s.d_rec[m], (size_t)n * sizeof(snk_read_result), hipMemcpyDeviceToHost, s.stream));
        if (snk_error_peek_async(dv.ctx, s.h_err, s.stream) != SNK_OK) die(snk_last_error());
        if (o.streaming) {                                  // the patch's thread, cumulative, before the next patch touches it
            HIPCHK(hipStreamSynchronize(s.stream));
            const int vt = (int)((total / (uint64_t)vblock) % (uint64_t)T);
            s.snap_sum.assign((size_t)nsum, 0);
            s.snap_max.assign(SNK_MAX_N, 0);
            snk_error err;
            if (snk_bind_stats(dv.ctx, dv.d_sum[(size_t)vt], dv.d_max[(size_t)vt]) != SNK_OK) die(snk_last_error());
            if (snk_stats_fetch(dv.ctx, s.snap_sum.data(), s.snap_max.data(), &err, nullptr) != SNK_OK) die(snk_last_error());
            // the thread's counts of earlier epochs (before a capacity regrowth the accumulators were closed and zeroed)
            for (const Epoch &e : epochs) {
                widen_add(e.sums[(size_t)vt].data(), e.lcap, s.snap_sum.data(), lcap, nq);
                for (int k = 0; k < SNK_MAX_N; ++k) s.snap_max[(size_t)k] = std::max(s.snap_max[(size_t)k], e.maxs[(size_t)vt][(size_t)k]);
            }
        }
        HIPCHK(hipEventRecord(s.done, s.stream));
        to_write.push(sp);
        total += (uint64_t)n;
        { Tick t_(3); have = next_chunks(c); }
    }
    since_start("last batch submitted");
    to_write.close();
    writer.join();
    since_start("writer done");
    join_readers();
    for (int m = 0; m < mates; ++m) wr[m].close();
    if (trim_out) for (int m = 0; m < mates; ++m) trimw[m].close();
    if (rmdup_stream) {
        uint64_t marked = 0;
        int32_t sentinel = 0;
        HIPCHK(hipSetDevice(devs[0].id));
        if (snk_rmdup_stream_stats(dup_table, &marked, &sentinel) != SNK_OK) die(snk_last_error());
        if (sentinel || getenv("SNK_RMDUP_SENTINEL_TEST")) {
            // a pair hashed to 2^64 - 1, the value the reference's duplicate search treats as "already removed": what it marks
            // then depends on the whole input's sort order (oracle: mark_duplicates), which only the two-pass path reproduces.
            // Run again that way; every output file is rewritten from the start.
            log << local_time() << "\trmdup: sentinel hash in the input, running again with two passes" << endl;
            log.close();
            cout.flush();
            for (int m = 0; m < mates; ++m) for (int t = 0; t < T; ++t) dupw[m][t].close();
            snk_rmdup_stream_destroy(dup_table);
            for (auto &t : slot_makers) t.join();
            slot_makers.clear();
            teardown();                                       // (the second run gets the device and the pinned memory to itself)
            setenv("SNK_RMDUP_TWO_PASS", "restarted", 1);
            unsetenv("SNK_RMDUP_SENTINEL_TEST");
            pid_t child = 0;
            int status = 0;
            if (posix_spawn(&child, "/proc/self/exe", nullptr, nullptr, argv, environ) != 0 || waitpid(child, &status, 0) != child) {
                cerr << "cannot start the two-pass rmdup run" << endl;
                _exit(1);
            }
            _exit(WIFEXITED(status) ? WEXITSTATUS(status) : 1);
        }
        cout << "totalReadsNum:\t" << total << endl;
        log << "duplicate reads number:\t" << marked << endl;
    }
    if (rmdup_on || rmdup_stream) {
        for (int m = 0; m < mates; ++m) for (int t = 0; t < T; ++t) dupw[m][t].close();
        log << "dup number:\t" << ndup_written << endl;
    }

    // ---- stats: reduce over the devices, fold the epochs into the widest geometry, write the reports
    collect_epoch();
    const int lfin = epochs.back().lcap;                      // capacities only grow
    std::vector<std::vector<uint64_t>> sums((size_t)T, std::vector<uint64_t>((size_t)snk_stats_u64(lfin, nq), 0)),
                                       maxs((size_t)T, std::vector<uint64_t>(SNK_MAX_N, 0));
    std::vector<const uint64_t *> sp(T), mp(T);
    for (const Epoch &e : epochs)
        for (int t = 0; t < T; ++t) {
            widen_add(e.sums[(size_t)t].data(), e.lcap, sums[(size_t)t].data(), lfin, nq);
            for (int k = 0; k < SNK_MAX_N; ++k) maxs[(size_t)t][(size_t)k] = std::max(maxs[(size_t)t][(size_t)k], e.maxs[(size_t)t][(size_t)k]);
        }
    for (int t = 0; t < T; ++t) {
        if (bc_from) {
            // preOutput converted the clean reads before stat_pe_fqs(..., "clean") counted them (src/peprocess.cpp:1601-1604,1960):
            // in the clean statistics the letter's counts belong to the letter it became (the switch there is case-insensitive)
            const string acgt = "ACGT";
            const size_t from = acgt.find(bc_from), to = acgt.find((char)toupper((unsigned char)bc_to));
            if (from != string::npos && to != string::npos && from != to)
                for (int k = 2; k < 2 + mates; ++k) {
                    uint64_t *f = sums[t].data() + snk_file_off(lfin, nq, k);
                    f[SNK_GS_A + to] += f[SNK_GS_A + from];
                    f[SNK_GS_A + from] = 0;
                    uint64_t *bs = f + snk_bs_off(lfin, nq);
                    for (int pos = 0; pos < lfin; ++pos) { bs[pos * 5 + to] += bs[pos * 5 + from]; bs[pos * 5 + from] = 0; }
                }
        }
        sp[t] = sums[t].data();
        mp[t] = maxs[t].data();
    }
    if (g_shard.child) {
        // a shard: its blocks are dumped for the parent (the check and the fallback), then all-reduced over the wire -- the path's
        // one collective (the reference: merge_stat once its threads have joined, src/peprocess.cpp:1994); rank 0
        // writes the reports from the reduced blocks and leaves them for the parent's comparison
        auto dump = [&](const string &path, int lc, int kind, const std::vector<const uint64_t *> &S, const std::vector<const uint64_t *> &M) {
            FILE *f = fopen(path.c_str(), "wb");
            ShardStatsHeader h{SHARD_MAGIC, T, lc, nq, kind, clean_total, ndup_written};
            bool ok = f && fwrite(&h, sizeof h, 1, f) == 1;
            const size_t ns = (size_t)snk_stats_u64(lc, nq);
            for (int t = 0; t < T && ok; ++t) ok = fwrite(S[(size_t)t], 8, ns, f) == ns;
            for (int t = 0; t < T && ok; ++t) ok = fwrite(M[(size_t)t], 8, (size_t)SNK_MAX_N, f) == (size_t)SNK_MAX_N;
            if (!ok || fclose(f) != 0) die("cannot write to the file," + path);
        };
        dump(g_shard.stats_path, lfin, 0, sp, mp);
        if (snk::ShardWire *w = shard_wire()) {
            for (auto &t : slot_makers) t.join();          // (a short shard can get here before its last slots exist)
            slot_makers.clear();
            HIPCHK(hipSetDevice(devs[0].id));
            // the shards agree on the geometry first (a longer read may have shown up in one of them only)
            uint64_t *d_geo = nullptr;
            HIPCHK(hipMalloc(&d_geo, 8));
            uint64_t geo = (uint64_t)lfin;
            HIPCHK(hipMemcpy(d_geo, &geo, 8, hipMemcpyHostToDevice));
            if (!w->allreduce_u64(d_geo, 1, snk::WIRE_MAX)) die("shard all-reduce failed (" + w->err + ")");
            HIPCHK(hipMemcpy(&geo, d_geo, 8, hipMemcpyDeviceToHost));
            HIPCHK(hipFree(d_geo));
            const int lall = (int)geo;
            const size_t ns = (size_t)snk_stats_u64(lall, nq);
            std::vector<uint64_t> wide((size_t)T * ns, 0), wmax((size_t)T * SNK_MAX_N, 0);
            for (int t = 0; t < T; ++t) {
                widen_add(sums[(size_t)t].data(), lfin, wide.data() + (size_t)t * ns, lall, nq);
                memcpy(wmax.data() + (size_t)t * SNK_MAX_N, maxs[(size_t)t].data(), SNK_MAX_N * 8);
            }
            uint64_t *d_s = nullptr, *d_m = nullptr;
            HIPCHK(hipMalloc(&d_s, wide.size() * 8));
            HIPCHK(hipMalloc(&d_m, wmax.size() * 8));
            HIPCHK(hipMemcpy(d_s, wide.data(), wide.size() * 8, hipMemcpyHostToDevice));
            HIPCHK(hipMemcpy(d_m, wmax.data(), wmax.size() * 8, hipMemcpyHostToDevice));
            if (void *comm = w->nccl_comm()) {
                // over RCCL: the C ABI's own collective, block by block (every shard walks t in the same order)
                snk_params pw = o.p;
                pw.max_read_len = lall;
                snk_ctx *cx = snk_create(&pw, devs[0].id);
                if (!cx) die(snk_last_error());
                for (int t = 0; t < T; ++t)
                    if (snk_bind_stats(cx, d_s + (size_t)t * ns, d_m + (size_t)t * SNK_MAX_N) != SNK_OK || snk_stats_allreduce(cx, comm, nullptr) != SNK_OK) die(snk_last_error());
                HIPCHK(hipDeviceSynchronize());
                snk_destroy(cx);
            } else if (!w->allreduce_u64(d_s, wide.size(), snk::WIRE_SUM) || !w->allreduce_u64(d_m, wmax.size(), snk::WIRE_MAX)) die("shard all-reduce failed (" + w->err + ")");
            if (g_shard.g == 0) {
                HIPCHK(hipMemcpy(wide.data(), d_s, wide.size() * 8, hipMemcpyDeviceToHost));
                HIPCHK(hipMemcpy(wmax.data(), d_m, wmax.size() * 8, hipMemcpyDeviceToHost));
                std::vector<const uint64_t *> S((size_t)T), M((size_t)T);
                for (int t = 0; t < T; ++t) { S[(size_t)t] = wide.data() + (size_t)t * ns; M[(size_t)t] = wmax.data() + (size_t)t * SNK_MAX_N; }
                char ebuf[512];
                o.p.max_read_len = lall;
                if (snk_write_reports(&o.p, T, S.data(), M.data(), o.out_dir.c_str(), ebuf, sizeof ebuf) != 0) { cerr << ebuf << endl; cout.flush(); fflush(stdout); _exit(1); }
                dump(o.out_dir + "/shard.merged.stats", lall, w->nccl_comm() ? 1 : 2, S, M);
            }
            g_shard.wire.reset();
        }
        for (auto &t : slot_makers) t.join();
        slot_makers.clear();
        log.close();
        cout.flush();
        fflush(stdout);
        _exit(0);
    }
    if (o.total_reads > 0 && !o.total_head) extract_every_kth(o, mates, clean_total);
    char ebuf[512];
    o.p.max_read_len = lfin;
    if (snk_write_reports(&o.p, T, sp.data(), mp.data(), o.out_dir.c_str(), ebuf, sizeof ebuf) != 0) { cerr << ebuf << endl; cout.flush(); fflush(stdout); _exit(1); }
    log << local_time() << "\tAnalysis accomplished!" << endl;
    since_start("reports written");
    g_clk.report();
    // everything is on disk: the process image goes away as a whole (unpinning and freeing a GB buffer by buffer, then the
    // runtime's own exit handlers, is a tenth of a second of a short run)
    for (auto &t : slot_makers) t.join();
    slot_makers.clear();
    if (getenv("SNK_CLEAN_EXIT")) { teardown(); return 0; }      // (profilers collect their data in exit handlers)
    log.flush();
    log.close();
    cout.flush();
    cerr.flush();
    fflush(stdout);
    fflush(stderr);
    _exit(0);
}
