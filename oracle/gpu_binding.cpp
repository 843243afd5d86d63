/*
 * gpu_binding.cpp -- TEST INFRASTRUCTURE: the reference-side binding of the drop-in boundary, compiled.
 *
 * A real `gpuPeProcess : public peProcess`, built by oracle/Makefile against the reference's own headers and
 * objects where they lie under $(REF)/src (nothing copied) and linked with soapnuke_amd/libsnk_filter.so.  It is
 * what a SOAPnuke maintainer would add to route the `filter` hot path through include/snk_filter.h:
 *
 *   - overrides exactly what the reference declares virtual on this seam (src/peprocess.h:61-62):
 *         virtual void filter_pe_fqs(PEcalOption *opt);
 *         virtual void filter_pe_fqs(PEcalOption *opt, int index);      // the rmdup variant
 *     Each call packs the patch into structure-of-arrays planes, calls snk_filter_batch() (GPU), and turns the
 *     16-byte records back into what the reference's caller expects: the cut fields on the raw records, the
 *     trim_result / clean_result vectors (trimmed private copies, preOutput applied as there) and the
 *     C_filter_stat counters of opt->local_fs (from the block's fs[] family counters).
 *   - does NOT replace stat_pe_fqs() / merge_stat(): they are NON-virtual (src/peprocess.h:60,70) and called
 *     statically from thread_process_reads (src/peprocess.cpp:1921,1957), so with the reference unmodified its
 *     statistics still run on the CPU, on the records this binding produced.  INTEGRATION.md states the one-word
 *     change (`virtual`) that lets the GPU statistics block replace them too.
 *   - seProcess has no virtual member at all (src/seprocess.h:38: `void filter_se_fqs(SEcalOption opt);`), so a
 *     subclass cannot hook the single-end path without that same one-word change; SE runs the reference unchanged.
 *
 * main() below is the reference's main (src/main.cpp:17-68) with `peProcess` replaced by `gpuPeProcess`.
 * tests/test_binding_gpu.py runs it and the unmodified binary on the same FASTQ files and compares every report
 * file and the clean FASTQ byte for byte.
 */
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <sstream>

#include "process_argv.h"
#include "global_parameter.h"
#include "peprocess.h"
#include "seprocess.h"
#include "../include/snk_filter.h"

namespace {

[[noreturn]] void die(const std::string &m) {           // the reference's convention
    cerr << "Error:" << m << endl;
    exit(1);
}

// C_global_parameter -> snk_params (the POD mirror of the hot-path fields)
struct ParamHolder {
    snk_params p;
    std::vector<std::string> keep;
    std::vector<const char *> ptrs[2];      // snk_params.adapter_list points here: owned by the holder, one per calling thread
};

void fill_params(ParamHolder &H, const C_global_parameter &gp, int max_len) {
    snk_params &P = H.p;
    snk_params_default(&P);
    P.paired = 1;
    P.quality_phred = gp.qualityPhred;
    P.output_quality_phred = gp.outputQualityPhred;
    P.max_base_quality = gp.maxBaseQuality;
    P.low_qual = gp.lowQual;
    P.low_qual_ratio = gp.lowQualityBaseRatio;
    P.n_ratio = gp.n_ratio;
    P.highA_ratio = gp.highA_ratio;
    P.polyG_tail = gp.polyG_tail;
    P.polyX_num = gp.polyX_num;
    P.mean_quality = gp.meanQuality;
    P.min_read_length = gp.min_read_length;
    P.max_read_length = gp.max_read_length;
    P.ada_trim = gp.adapter_discard_or_trim == "trim";
    P.contam_trim = gp.contam_discard_or_trim == "trim";
    if (!gp.trim.empty()) {
        int v[4] = {0, 0, 0, 0}, k = 0;
        std::stringstream ss(gp.trim);
        std::string e;
        while (std::getline(ss, e, ',') && k < 4) v[k++] = atoi(e.c_str());
        P.has_hard_trim = 1;
        for (int i = 0; i < 4; ++i) P.hard_trim[i] = v[i];
    }
    auto pair_of = [](const std::string &s, int32_t &a, int32_t &b) {
        const size_t c = s.find(',');
        if (c == std::string::npos) return;
        a = atoi(s.substr(0, c).c_str());
        b = atoi(s.substr(c + 1).c_str());
    };
    if (!gp.trimBadHead.empty() || !gp.trimBadTail.empty()) {
        P.has_lq_trim = 1;
        pair_of(gp.trimBadHead, P.lq_head_qual, P.lq_head_len);
        pair_of(gp.trimBadTail, P.lq_tail_qual, P.lq_tail_len);
    }
    P.ada_mis[0] = gp.adaMis;   P.ada_mr[0] = gp.adaMR;   P.ada_edge[0] = gp.adaEdge;
    P.ada_mis[1] = gp.adaMis2;  P.ada_mr[1] = gp.adaMR2;  P.ada_edge[1] = gp.adaEdge2;
    H.keep.clear();
    H.keep.reserve(gp.ada1s.size() + gp.ada2s.size() + 8);
    P.n_adapters[0] = (int)gp.ada1s.size();
    P.n_adapters[1] = (int)gp.ada2s.size();
    // lists of any length go through snk_params.adapter_list
    H.ptrs[0].clear(); H.ptrs[1].clear();
    for (size_t i = 0; i < gp.ada1s.size(); ++i) { H.keep.push_back(gp.ada1s[i]); H.ptrs[0].push_back(H.keep.back().c_str()); }
    for (size_t i = 0; i < gp.ada2s.size(); ++i) { H.keep.push_back(gp.ada2s[i]); H.ptrs[1].push_back(H.keep.back().c_str()); }
    for (int m = 0; m < 2; ++m) P.adapter_list[m] = H.ptrs[m].empty() ? nullptr : H.ptrs[m].data();
    auto str = [&](const std::string &s) -> const char * {
        if (s.empty()) return nullptr;
        H.keep.push_back(s);
        return H.keep.back().c_str();
    };
    P.contam[0] = str(gp.contam1_seq);
    P.contam[1] = str(gp.contam2_seq);
    P.ct_match_r = str(gp.ctMatchR);
    P.global_contams = str(gp.global_contams);
    P.g_mrs = str(gp.g_mrs);
    P.g_mms = str(gp.g_mms);
    P.rmdup = gp.rmdup ? 1 : 0;
    P.max_read_len = max_len;
}

// fs[] family counters of the block -> the C_filter_stat the reference's reports read
void add_fs(C_filter_stat &f, const uint64_t *fs) {
    f.dupReadsNum += fs[SNK_FS_DUP];
    f.tile_num += fs[SNK_FS_TILE];
    f.fov_num += fs[SNK_FS_FOV];
    f.over_lapped_num += fs[SNK_FS_OVERLAP];
#define FAM(B, a, b, c, d) f.a += fs[B]; f.b += fs[B + 1]; f.c += fs[B + 2]; f.d += fs[B + 3];
    FAM(SNK_FS_SHORT, short_len_num, short_len_num1, short_len_num2, short_len_num_overlap)
    FAM(SNK_FS_LONG, long_len_num, long_len_num1, long_len_num2, long_len_num_overlap)
    FAM(SNK_FS_GCONTAM, include_global_contam_seq_num, include_global_contam_seq_num1, include_global_contam_seq_num2, include_global_contam_seq_num_overlap)
    FAM(SNK_FS_CONTAM, include_contam_seq_num, include_contam_seq_num1, include_contam_seq_num2, include_contam_seq_num_overlap)
    FAM(SNK_FS_NRATE, n_ratio_num, n_ratio_num1, n_ratio_num2, n_ratio_num_overlap)
    FAM(SNK_FS_HIGHA, highA_num, highA_num1, highA_num2, highA_num_overlap)
    FAM(SNK_FS_POLYX, polyX_num, polyX_num1, polyX_num2, polyX_num_overlap)
    FAM(SNK_FS_LOWQUAL, low_qual_base_ratio_num, low_qual_base_ratio_num1, low_qual_base_ratio_num2, low_qual_base_ratio_num_overlap)
    FAM(SNK_FS_MEANQ, mean_quality_num, mean_quality_num1, mean_quality_num2, mean_quality_num_overlap)
    FAM(SNK_FS_ADAPTER, include_adapter_seq_num, include_adapter_seq_num1, include_adapter_seq_num2, include_adapter_seq_num_overlap)
#undef FAM
}

// one context per calling thread (the reference calls the seam concurrently from its T workers, each with
// disjoint patches and disjoint local_fs: src/peprocess.cpp:43-58)
struct ThreadCtx {
    snk_ctx *ctx = nullptr;
    ParamHolder params;
    int cap = 0;
    std::vector<uint8_t> seq[2], qual[2], dup;
    std::vector<uint16_t> len[2];
    std::vector<snk_read_result> rec[2];
    std::vector<uint64_t> sum;
    ~ThreadCtx() { if (ctx) snk_destroy(ctx); }
};
thread_local ThreadCtx tls;

}  // namespace

class gpuPeProcess : public peProcess {
public:
    explicit gpuPeProcess(C_global_parameter m_gp) : peProcess(m_gp) {
        if (gp.module_name != "filter") die("the GPU binding serves the filter module only");
        if (!gp.tile.empty() || !gp.fov.empty() || gp.index_remove)
            die("tile / fov / index removal need the read name: not routed through this binding");
    }
    void filter_pe_fqs(PEcalOption *opt) override { run(opt, nullptr); }                       // src/peprocess.h:61
    void filter_pe_fqs(PEcalOption *opt, int index) override {                                    // src/peprocess.h:62
        // the duplicate flags of this patch and the dupReads side files, as src/peprocess.cpp:1490-1548 does
        const size_t n = opt->fq1s->size();
        std::vector<uint8_t> flags(n, 0);
        if (gp.rmdup) {
            checkDup.lock();
            for (size_t k = 0; k < n && k < opt->fq2s->size(); ++k)
                if (dupFlag[threadCurReadReadsNumIdx[index] - n + k]) {
                    flags[k] = 1;
                    const std::string a = (*opt->fq1s)[k].toString(), b = (*opt->fq2s)[k].toString();
                    gzwrite(dupThreadOut1[index], a.c_str(), a.size());
                    gzwrite(dupThreadOut2[index], b.c_str(), b.size());
                }
            checkDup.unlock();
        }
        run(opt, flags.data());
    }

private:
    void run(PEcalOption *opt, const uint8_t *dup) {
        std::vector<C_fastq> &f1 = *opt->fq1s, &f2 = *opt->fq2s;
        const size_t n = std::min(f1.size(), f2.size());
        if (n == 0) return;
        int maxlen = 1;
        for (size_t i = 0; i < n; ++i) maxlen = std::max<int>(maxlen, (int)std::max(f1[i].sequence.size(), f2[i].sequence.size()));
        if (maxlen > SNK_READ_MAX_LEN) die("read longer than 1000 bases");
        ThreadCtx &T = tls;
        if (!T.ctx || maxlen > T.cap) {
            if (T.ctx) snk_destroy(T.ctx);
            T.cap = std::max(maxlen, T.cap);
            fill_params(T.params, gp, T.cap);
            T.ctx = snk_create(&T.params.p, 0);
            if (!T.ctx) die(snk_last_error());
            int32_t lcap, nq; int64_t nsum;
            snk_stats_geometry(T.ctx, &lcap, &nq, &nsum);
            T.sum.assign((size_t)nsum, 0);
        }
        const int pitch = (T.cap + 15) / 16 * 16;
        snk_batch b;
        memset(&b, 0, sizeof b);
        b.n = (int64_t)n;
        b.pitch = pitch;
        for (int m = 0; m < 2; ++m) {
            std::vector<C_fastq> &f = m ? f2 : f1;
            T.seq[m].assign(n * (size_t)pitch, 0);
            T.qual[m].assign(n * (size_t)pitch, 0);
            T.len[m].resize(n);
            T.rec[m].resize(n);
            for (size_t i = 0; i < n; ++i) {
                if (f[i].qual_seq.size() != f[i].sequence.size()) die("sequence and quality lengths differ");
                memcpy(&T.seq[m][i * pitch], f[i].sequence.data(), f[i].sequence.size());
                memcpy(&T.qual[m][i * pitch], f[i].qual_seq.data(), f[i].qual_seq.size());
                T.len[m][i] = (uint16_t)f[i].sequence.size();
            }
            b.seq[m] = T.seq[m].data();
            b.qual[m] = T.qual[m].data();
            b.len[m] = T.len[m].data();
        }
        b.dup = dup;
        if (snk_stats_clear(T.ctx, nullptr) != SNK_OK) die(snk_last_error());
        if (snk_filter_batch(T.ctx, &b, T.rec[0].data(), T.rec[1].data()) != SNK_OK) die(snk_last_error());
        snk_error err;
        if (snk_stats_fetch(T.ctx, T.sum.data(), nullptr, &err, nullptr) != SNK_OK) die(snk_last_error());
        if (err.code == SNK_E_BAD_BASE) die("unrecognized sequence," + (err.mate ? f2 : f1)[err.index].sequence);   // src/read_filter.cpp:283
        if (err.code == SNK_E_EMPTY_SEQ) die("empty sequence");                                                       // :251
        if (err.code) die("quality is too high or too low,please check the quality system parameter or fastq file");
        add_fs(*opt->local_fs, T.sum.data());

        const bool copy_back = gp.adapter_discard_or_trim == "trim" || gp.contam_discard_or_trim == "trim" || !gp.trim.empty() ||
                               !gp.trimBadHead.empty() || !gp.trimBadTail.empty();                                  // src/peprocess.cpp:1441
        for (size_t i = 0; i < n; ++i) {
            C_fastq c[2] = {f1[i], f2[i]};                 // the filter's private copies (src/sequence.cpp:182-196)
            for (int m = 0; m < 2; ++m) {
                const snk_read_result &r = T.rec[m][i];
                C_fastq &raw = m ? f2[i] : f1[i];
                c[m].raw_length = (int)raw.sequence.size();                                                          // src/read_filter.cpp:84
                c[m].head_hdcut = r.head_hdcut; c[m].head_lqcut = r.head_lqcut;
                c[m].tail_hdcut = r.tail_hdcut; c[m].tail_lqcut = r.tail_lqcut;
                c[m].adacut_pos = r.adacut_pos;
                c[m].sequence = raw.sequence.substr(r.clean_start, r.clean_len);
                c[m].qual_seq = raw.qual_seq.substr(r.clean_start, r.clean_len);
                if (copy_back) {
                    raw.head_hdcut = r.head_hdcut; raw.head_lqcut = r.head_lqcut;
                    raw.tail_hdcut = r.tail_hdcut; raw.tail_lqcut = r.tail_lqcut;
                    raw.adacut_pos = r.adacut_pos;
                }
            }
            if (!gp.trim_fq1.empty()) {                                                                              // :1460-1466
                preOutput(1, c[0]);
                preOutput(2, c[1]);
                opt->trim_result1->emplace_back(c[0]);
                opt->trim_result2->emplace_back(c[1]);
            }
            if (T.rec[0][i].reason == SNK_KEEP && !gp.clean_fq1.empty()) {                                           // :1467-1476
                preOutput(1, c[0]);
                preOutput(2, c[1]);
                opt->clean_result1->emplace_back(c[0]);
                opt->clean_result2->emplace_back(c[1]);
            }
        }
    }
};

int main(int argc, char *argv[]) {                        // src/main.cpp:17-68, PE `filter` through the binding
    C_global_parameter gp;
    check_module(argc, argv);
    global_parameter_initial(argc, argv, gp);
    check_parameter(argc, argv, gp);
    if (gp.module_name != "filter") die("this build carries the GPU binding of the filter module only");
    if (!gp.fq2_path.empty()) {
        gpuPeProcess new_task(gp);
        new_task.process();
    } else {
        seProcess new_task(gp);                            // no virtual seam on the single-end class: reference code
        new_task.process();
    }
    return 0;
}
