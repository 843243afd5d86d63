/*
 * ref_shim.cpp -- TEST INFRASTRUCTURE.  C-ABI wrapper around the *real*
 * reference objects, compiled by oracle/Makefile together with the reference's
 * own sources where they lie under $(REF)/src (nothing is copied into this
 * repo; the output goes to oracle/_ref/, which is git-ignored).
 *
 * It drives exactly the seam the product replaces:
 *   peProcess::filter_pe_fqs + stat_pe_fqs("raw"/"clean")   (src/peprocess.cpp:1424,1076)
 *   seProcess::filter_se_fqs + stat_se_fqs                   (src/seprocess.cpp:871,632)
 *   adapter_pos()                                            (src/read_filter.cpp:707)
 * and exports the results in the layout of include/snk_filter.h so that the
 * oracle restatement (and through it the HIP kernels) can be memcmp'd with the
 * reference.  The reference exit(1)s on malformed input -- callers must only
 * feed it valid reads.
 */
#include "gc.h"
#include "peprocess.h"
#include "rmdup.h"
#include "seprocess.h"
#include "read_filter.h"
#include "sequence.h"
#include "../include/snk_filter.h"
#include <cstdio>
#include <cstring>
#include <sstream>

static_assert(sizeof(C_reads_trim_stat) == SNK_TS_N * sizeof(uint64_t),
              "C_reads_trim_stat layout changed");

static C_global_parameter make_gp(const snk_params *P) {
    C_global_parameter gp;
    gp.module_name = "filter";
    gp.threads_num = 1;
    gp.qualityPhred = P->quality_phred;
    gp.outputQualityPhred = P->output_quality_phred;
    gp.maxBaseQuality = P->max_base_quality;
    gp.lowQual = P->low_qual;
    gp.lowQualityBaseRatio = P->low_qual_ratio;
    gp.n_ratio = P->n_ratio;
    gp.highA_ratio = P->highA_ratio;
    gp.polyG_tail = P->polyG_tail;
    gp.polyX_num = P->polyX_num;
    gp.meanQuality = P->mean_quality;
    gp.min_read_length = P->min_read_length;
    gp.max_read_length = P->max_read_length;
    gp.adapter_discard_or_trim = P->ada_trim ? "trim" : "discard";
    gp.contam_discard_or_trim = P->contam_trim ? "trim" : "discard";
    if (P->has_hard_trim) {
        std::ostringstream s;
        if (P->paired) s << P->hard_trim[0] << "," << P->hard_trim[1] << "," << P->hard_trim[2] << "," << P->hard_trim[3];
        else s << P->hard_trim[0] << "," << P->hard_trim[1];
        gp.trim = s.str();
    }
    if (P->has_lq_trim) {
        bool head = P->lq_head_qual || P->lq_head_len, tail = P->lq_tail_qual || P->lq_tail_len;
        if (!head && !tail) head = true;
        if (head) { std::ostringstream s; s << P->lq_head_qual << "," << P->lq_head_len; gp.trimBadHead = s.str(); }
        if (tail) { std::ostringstream s; s << P->lq_tail_qual << "," << P->lq_tail_len; gp.trimBadTail = s.str(); }
    }
    gp.adaMis = P->ada_mis[0];  gp.adaMR = P->ada_mr[0];  gp.adaEdge = P->ada_edge[0];
    gp.adaMis2 = P->ada_mis[1]; gp.adaMR2 = P->ada_mr[1]; gp.adaEdge2 = P->ada_edge[1];
    for (int i = 0; i < P->n_adapters[0]; i++) gp.ada1s.push_back(snk_adapter_at(P, 0, i));
    for (int i = 0; i < P->n_adapters[1]; i++) gp.ada2s.push_back(snk_adapter_at(P, 1, i));
    if (P->contam[0]) gp.contam1_seq = P->contam[0];
    if (P->contam[1]) gp.contam2_seq = P->contam[1];
    if (P->ct_match_r && *P->ct_match_r) gp.ctMatchR = P->ct_match_r;
    if (P->global_contams) gp.global_contams = P->global_contams;
    if (P->g_mrs) gp.g_mrs = P->g_mrs;
    if (P->g_mms) gp.g_mms = P->g_mms;
    gp.trim_fq1 = "t1"; gp.trim_fq2 = "t2";     /* keep every trimmed copy (trim_result) */
    gp.clean_fq1 = "c1"; gp.clean_fq2 = "c2";
    gp.rmdup = false;
    return gp;
}

static void export_file(const C_fastq_file_stat &st, const snk_params *P, uint64_t *f) {
    const int lcap = P->max_read_len, nq = P->max_base_quality + 1;
    f[SNK_GS_READS] += st.gs.reads_number;
    f[SNK_GS_BASES] += st.gs.base_number;
    f[SNK_GS_A] += st.gs.a_number; f[SNK_GS_C] += st.gs.c_number; f[SNK_GS_G] += st.gs.g_number;
    f[SNK_GS_T] += st.gs.t_number; f[SNK_GS_N_] += st.gs.n_number;
    f[SNK_GS_Q20] += st.gs.q20_num; f[SNK_GS_Q30] += st.gs.q30_num;
    uint64_t *bs = f + snk_bs_off(lcap, nq), *qs = f + snk_qs_off(lcap, nq), *ts = f + snk_ts_off(lcap, nq);
    for (int i = 0; i < lcap; i++) {
        for (int j = 0; j < 5; j++) bs[i * 5 + j] += st.bs.position_acgt_content[i][j];
        for (int j = 0; j < nq && j < P->max_base_quality; j++) qs[(int64_t)i * nq + j] += st.qs.position_qual[i][j];
    }
    const uint64_t *t = st.ts.hlq;   /* hlq,ht,ta,tlq,tt are contiguous (static_assert above) */
    for (int i = 0; i < SNK_TS_N; i++) ts[i] += t[i];
}

static void fill_rec(snk_read_result *o, const C_fastq &t, int reason, int v) {
    o->head_hdcut = (int16_t)t.head_hdcut; o->head_lqcut = (int16_t)t.head_lqcut;
    o->tail_hdcut = (int16_t)t.tail_hdcut; o->tail_lqcut = (int16_t)t.tail_lqcut;
    o->adacut_pos = (int16_t)t.adacut_pos;
    int clen = (int)t.sequence.size(), start = 0;
    if (clen > 0) { start = 0; if (t.head_hdcut > start) start = t.head_hdcut; if (t.head_lqcut > start) start = t.head_lqcut; }
    o->clean_start = (uint16_t)start; o->clean_len = (uint16_t)clen;
    o->reason = (uint8_t)reason; o->flags = (uint8_t)v;
}

/* which counter family did pe_discard/se_discard bump? */
static int reason_of(const C_filter_stat &f, int ret, int *v) {
    struct { uint64_t n, n1, n2, ov; int r; } fam[] = {
        {f.short_len_num, f.short_len_num1, f.short_len_num2, f.short_len_num_overlap, SNK_R_SHORT},
        {f.long_len_num, f.long_len_num1, f.long_len_num2, f.long_len_num_overlap, SNK_R_LONG},
        {f.include_global_contam_seq_num, f.include_global_contam_seq_num1, f.include_global_contam_seq_num2, f.include_global_contam_seq_num_overlap, SNK_R_GCONTAM},
        {f.include_contam_seq_num, f.include_contam_seq_num1, f.include_contam_seq_num2, f.include_contam_seq_num_overlap, SNK_R_CONTAM},
        {f.n_ratio_num, f.n_ratio_num1, f.n_ratio_num2, f.n_ratio_num_overlap, SNK_R_NRATE},
        {f.highA_num, f.highA_num1, f.highA_num2, f.highA_num_overlap, SNK_R_HIGHA},
        {f.polyX_num, f.polyX_num1, f.polyX_num2, f.polyX_num_overlap, SNK_R_POLYX},
        {f.low_qual_base_ratio_num, f.low_qual_base_ratio_num1, f.low_qual_base_ratio_num2, f.low_qual_base_ratio_num_overlap, SNK_R_LOWQUAL},
        {f.mean_quality_num, f.mean_quality_num1, f.mean_quality_num2, f.mean_quality_num_overlap, SNK_R_MEANQ},
        {f.include_adapter_seq_num, f.include_adapter_seq_num1, f.include_adapter_seq_num2, f.include_adapter_seq_num_overlap, SNK_R_ADAPTER},
    };
    *v = 0;
    if (f.dupReadsNum) return SNK_R_DUP;
    if (f.tile_num) return SNK_R_TILE;
    if (f.fov_num) return SNK_R_FOV;
    if (f.over_lapped_num) return SNK_R_OVERLAP;
    for (auto &x : fam)
        if (x.n) { *v = x.ov ? 3 : (x.n1 ? 1 : (x.n2 ? 2 : 0)); return x.r; }
    return ret == 1 ? SNK_R_EMPTY : SNK_KEEP;
}

static void export_fs(const C_filter_stat &f, uint64_t *fs) {
    fs[SNK_FS_DUP] += f.dupReadsNum; fs[SNK_FS_TILE] += f.tile_num; fs[SNK_FS_FOV] += f.fov_num;
    fs[SNK_FS_OVERLAP] += f.over_lapped_num;
#define FAM(B, a, b, c, d) fs[B] += f.a; fs[B + 1] += f.b; fs[B + 2] += f.c; fs[B + 3] += f.d;
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

extern "C" {

int snkref_adapter_pos(const char *read, int read_len, const char *adapter, int adapter_len,
                       int ada_mis, float ada_mr, int ada_edge) {
    C_global_parameter gp;
    gp.adaMis = ada_mis; gp.adaMR = ada_mr; gp.adaEdge = ada_edge;
    std::string r(read, read_len), a(adapter, adapter_len);
    return adapter_pos(r, a, gp);
}

int snkref_filter_batch(const snk_params *P, const snk_batch *B, snk_read_result *out1,
                        snk_read_result *out2, uint64_t *sum, uint64_t *maxb) {
    C_global_parameter gp = make_gp(P);
    const int lcap = P->max_read_len, nq = P->max_base_quality + 1;
    const int64_t n = B->n;
    uint64_t *file[4];
    for (int k = 0; k < 4; k++) file[k] = sum + snk_file_off(lcap, nq, k);

    if (P->paired) {
        peProcess proc(gp);
        proc.bq_check = 1;                   /* skip the one-shot phred heuristic, :1211 */
        proc.pair_check = 1;
        std::vector<C_fastq> fq1s, fq2s, t1, t2, c1, c2;
        for (int64_t i = 0; i < n; i++) {
            C_fastq a, b;
            proc.C_fastq_init(a, b);
            int l1 = B->len[0] ? B->len[0][i] : B->fixed_len[0], l2 = B->len[1] ? B->len[1][i] : B->fixed_len[1];
            a.seq_id = "@r/1"; b.seq_id = "@r/2";
            a.sequence.assign((const char *)B->seq[0] + i * B->pitch, l1);
            a.qual_seq.assign((const char *)B->qual[0] + i * B->pitch, l1);
            b.sequence.assign((const char *)B->seq[1] + i * B->pitch, l2);
            b.qual_seq.assign((const char *)B->qual[1] + i * B->pitch, l2);
            fq1s.push_back(a); fq2s.push_back(b);
        }
        std::vector<C_fastq> raw1 = fq1s, raw2 = fq2s;   /* pristine copies for the per-pair pass */
        C_filter_stat fs;
        PEcalOption opt;
        opt.local_fs = &fs; opt.fq1s = &fq1s; opt.fq2s = &fq2s;
        opt.trim_result1 = &t1; opt.trim_result2 = &t2; opt.clean_result1 = &c1; opt.clean_result2 = &c2;
        proc.filter_pe_fqs(&opt);
        export_fs(fs, sum);
        if ((int64_t)t1.size() != n) return -1;
        size_t kept = 0;
        for (int64_t i = 0; i < n; i++) {    /* per-pair reason: same calls, one pair at a time */
            C_pe_fastq_filter f(raw1[i], raw2[i], gp);
            f.pe_trim(gp);
            C_filter_stat one;
            int ret = f.pe_discard(&one, gp), v = 0;
            int reason = reason_of(one, ret, &v);
            fill_rec(&out1[i], t1[i], reason, v);
            fill_rec(&out2[i], t2[i], reason, v);
            uint64_t key = (B->first_index + (uint64_t)i + 1) << 16;
            if ((key | raw1[i].sequence.size()) > maxb[0]) maxb[0] = key | raw1[i].sequence.size();
            if ((key | raw2[i].sequence.size()) > maxb[1]) maxb[1] = key | raw2[i].sequence.size();
            if (reason == SNK_KEEP) {
                if ((key | t1[i].sequence.size()) > maxb[2]) maxb[2] = key | t1[i].sequence.size();
                if ((key | t2[i].sequence.size()) > maxb[3]) maxb[3] = key | t2[i].sequence.size();
                kept++;
            }
        }
        if (kept != c1.size()) return -2;
        C_fastq_file_stat s_raw1(gp), s_raw2(gp), s_c1(gp), s_c2(gp);
        PEstatOption o_raw; o_raw.fq1s = &fq1s; o_raw.fq2s = &fq2s; o_raw.stat1 = &s_raw1; o_raw.stat2 = &s_raw2;
        proc.stat_pe_fqs(o_raw, "raw");
        PEstatOption o_c; o_c.fq1s = &c1; o_c.fq2s = &c2; o_c.stat1 = &s_c1; o_c.stat2 = &s_c2;
        proc.stat_pe_fqs(o_c, "clean");
        export_file(s_raw1, P, file[0]); export_file(s_raw2, P, file[1]);
        export_file(s_c1, P, file[2]);   export_file(s_c2, P, file[3]);
    } else {
        seProcess proc(gp);
        proc.se_bq_check = 1;
        std::vector<C_fastq> fq1s, t1, c1;
        for (int64_t i = 0; i < n; i++) {
            C_fastq a;
            proc.C_fastq_init(a);
            int l1 = B->len[0] ? B->len[0][i] : B->fixed_len[0];
            a.seq_id = "@r";
            a.sequence.assign((const char *)B->seq[0] + i * B->pitch, l1);
            a.qual_seq.assign((const char *)B->qual[0] + i * B->pitch, l1);
            fq1s.push_back(a);
        }
        std::vector<C_fastq> raw1 = fq1s;
        C_filter_stat fs;
        SEcalOption opt;
        opt.se_local_fs = &fs; opt.fq1s = &fq1s; opt.trim_result1 = &t1; opt.clean_result1 = &c1;
        proc.filter_se_fqs(opt);
        export_fs(fs, sum);
        if ((int64_t)t1.size() != n) return -1;
        size_t kept = 0;
        for (int64_t i = 0; i < n; i++) {
            C_single_fastq_filter f(raw1[i], gp);
            f.se_trim(gp);
            C_filter_stat one;
            int ret = f.se_discard(&one, gp), v = 0;
            int reason = reason_of(one, ret, &v);
            fill_rec(&out1[i], t1[i], reason, 0);
            uint64_t key = (B->first_index + (uint64_t)i + 1) << 16;
            if ((key | raw1[i].sequence.size()) > maxb[0]) maxb[0] = key | raw1[i].sequence.size();
            if (reason == SNK_KEEP) {
                if ((key | t1[i].sequence.size()) > maxb[2]) maxb[2] = key | t1[i].sequence.size();
                kept++;
            }
        }
        if (kept != c1.size()) return -2;
        C_fastq_file_stat s_raw1(gp), s_c1(gp);
        SEstatOption o_raw; o_raw.fq1s = &fq1s; o_raw.stat1 = &s_raw1;
        proc.stat_se_fqs(o_raw, "raw");
        SEstatOption o_c; o_c.fq1s = &c1; o_c.stat1 = &s_c1;
        proc.stat_se_fqs(o_c, "clean");
        export_file(s_raw1, P, file[0]);
        export_file(s_c1, P, file[2]);
    }
    return 0;
}


/* std::hash<std::string> exactly as src/peprocess.cpp:3680 calls it */
// cal_quar_from_array(), src/gc.cpp:68-119 (data must hold len+1 counters)
void snkref_cal_quar(uint64_t *data, int len, float out[6]) {
    const quartile_result q = cal_quar_from_array(data, len);
    out[0] = q.mean; out[1] = q.median; out[2] = q.lower_quar; out[3] = q.upper_quar; out[4] = q.first10_quar; out[5] = q.last10_quar;
}

uint64_t snkref_hash(const char *p, uint64_t len) { return (uint64_t)std::hash<std::string>()(std::string(p, len)); }

/* rmdup::markDup of the reference (src/rmdup.cpp:14); the class frees `data` itself */
void snkref_markdup(const uint64_t *hash, uint64_t n, uint8_t *dup) {
    uint64_t *data = new uint64_t[n ? n : 1];
    memcpy(data, hash, n * sizeof(uint64_t));
    bool *flags = new bool[n ? n : 1];
    memset(flags, 0, n ? n : 1);
    rmdup *r = new rmdup(data, n);
    r->markDup(flags);
    delete r;
    for (uint64_t i = 0; i < n; ++i) dup[i] = flags[i] ? 1 : 0;
    delete[] flags;
}

/* hasContam(ref, contam, gp, mr) / global_contam_pos() of the reference, src/read_filter.cpp:507,961 */
int snkref_has_contam(const char *read, int read_len, const char *contam, int contam_len, float mr, int ada_mis, int ada_edge) {
    C_global_parameter gp;
    gp.adaMis = ada_mis; gp.adaEdge = ada_edge;
    std::string r(read, read_len), c(contam, contam_len);
    return hasContam(r, c, gp, mr);
}
int snkref_global_contam_pos(const char *read, int read_len, const char *contam, int contam_len, float mr, int mm) {
    std::string r(read, read_len), c(contam, contam_len);
    return global_contam_pos(r, c, mr, mm);
}
} /* extern "C" */
