/*
 * snk_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE (see snk_oracle.h).
 *
 * Plain-C scalar restatement of the reference `filter` hot path.  Every
 * function cites the reference lines it follows (paths relative to
 * /root/reference).  The code is deliberately sequential and literal: it is the
 * checker for the HIP kernels, not something to be fast.
 *
 * Pinned against the compiled reference (oracle/_ref/libsnkref.so) by
 * tests/test_oracle_vs_ref.py and the golden vectors in tests/golden/.
 */
#include "snk_oracle.h"
#include <ctype.h>
#include <limits.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* float -> int the way the reference binary does it on x86-64 (cvttss2si):
 * NaN / out-of-range give INT_MIN ("integer indefinite").  Matters for
 * adapter_pos() when (adptLen-adaEdge)/(adaMis+1) == 0, SURVEY H1.           */
static int f2i_x86(float f) {
    if (!(f == f)) return INT_MIN;
    if (f >= 2147483648.0f || f < -2147483648.0f) return INT_MIN;
    return (int)f;
}

void snk_oracle_params_default(snk_params *p) { /* src/global_parameter.h:20-83 */
    memset(p, 0, sizeof(*p));
    p->struct_size = (int32_t)sizeof(*p);
    p->paired = 1;
    p->quality_phred = 33;
    p->output_quality_phred = 33;
    p->max_base_quality = 42;
    p->low_qual = 5;
    p->low_qual_ratio = 0.5f;
    p->n_ratio = 0.05f;
    p->highA_ratio = -1;
    p->polyG_tail = -1;
    p->polyX_num = -1;
    p->mean_quality = -1;
    p->min_read_length = 30;
    p->max_read_length = -1;
    p->ada_mis[0] = p->ada_mis[1] = 2;
    p->ada_mr[0] = p->ada_mr[1] = 0.5f;
    p->ada_edge[0] = p->ada_edge[1] = 6;
    p->max_read_len = 150;
}

/* read character with the out-of-range convention used throughout this repo:
 * outside [0,len) reads as 0 (std::string gives '\0' at size(); beyond that the
 * reference is UB -- SURVEY Q6).                                              */
static inline int rd(const uint8_t *s, int len, int i) {
    return (i >= 0 && i < len) ? s[i] : 0;
}

/* src/read_filter.cpp:707-790 */
int snk_oracle_adapter_pos(const uint8_t *read, int readLen, const char *adapter,
                           int adptLen, int adaMis, float adaMR, int adaEdge) {
    if (adptLen == 0) return -1;                                   /* :709 */
    int minEdge5 = 5;
    float misGrad5 = (float)((adptLen - minEdge5) / (adaMis + 1)); /* :714 int division */
    float misGrad = (float)((adptLen - adaEdge) / (adaMis + 1));   /* :715 */
    int r1, mis, maxSegMatch;
    int segMatchThr = (int)ceilf((float)adptLen * adaMR);          /* :717 */
    int misMatchTemp;

    for (r1 = 1; r1 <= minEdge5; ++r1) {                           /* :720-742 */
        mis = 0;
        maxSegMatch = 0;
        misMatchTemp = f2i_x86((float)(adptLen - r1) / misGrad5);
        for (int c = 0; c < adptLen - r1; ++c) {
            if ((uint8_t)adapter[r1 + c] == rd(read, readLen, c)) {
                maxSegMatch++;
                if (maxSegMatch >= segMatchThr) return 0;
            } else {
                mis++;
                maxSegMatch = 0;
                if (mis > misMatchTemp) break;
            }
        }
        if (mis <= misMatchTemp) return 0;
    }
    for (r1 = 0; r1 <= readLen - adptLen; ++r1) {                  /* :743-764 */
        maxSegMatch = 0;
        mis = 0;
        for (int c = 0; c < adptLen; ++c) {
            if ((uint8_t)adapter[c] == rd(read, readLen, r1 + c)) {
                maxSegMatch++;
                if (maxSegMatch >= segMatchThr) return r1;
            } else {
                mis++;
                maxSegMatch = 0;
                if (mis > adaMis) break;
            }
        }
        if (mis <= adaMis) return r1;
    }
    for (r1 = 0; r1 < adptLen - adaEdge; ++r1) {                   /* :765-788 */
        mis = 0;
        maxSegMatch = 0;
        misMatchTemp = f2i_x86((float)r1 / misGrad);
        for (int c = 0; c < r1 + adaEdge; ++c) {
            if ((uint8_t)adapter[c] == rd(read, readLen, readLen - r1 - adaEdge + c)) {
                maxSegMatch++;
                if (maxSegMatch >= segMatchThr) return readLen - r1 - adaEdge;
            } else {
                mis++;
                maxSegMatch = 0;
                if (mis > misMatchTemp) break;
            }
        }
        if (mis <= misMatchTemp) return readLen - r1 - adaEdge;
    }
    return -1;
}

/* src/read_filter.cpp:472-482 */
int snk_oracle_polyG_number(const uint8_t *s, int len) {
    int n = 0;
    for (int i = len - 1; i >= 0; i--) {
        if (s[i] == 'G' || s[i] == 'g') n++;
        else break;
    }
    return n;
}

/* hasContam(), src/read_filter.cpp:507-603 / :604-700.  N in the READ never counts as a mismatch
 * (it does not count as a match either: the run is not reset, :536-542).                       */
int snk_oracle_has_contam(const uint8_t *ref, int readLen, const char *contam, int contamLen,
                          int segMatchThr, int adaMis, int adaEdge) {
    if (contamLen == 0) return -1;
    const float misGrad = (float)((contamLen - adaEdge) / (adaMis + 1));      /* :513 int division */
    float segGrad;
    if (segMatchThr - 7 + 1 == 0) segGrad = 0;                                /* :517-521 */
    else segGrad = (float)((contamLen - adaEdge) / (segMatchThr - 7 + 1));
    int r1, mis, maxSegMatch, misMatchTemp, segMatchTemp;
    for (r1 = 0; r1 < contamLen - adaEdge; ++r1) {                            /* :523-547 */
        mis = 0;
        maxSegMatch = 0;
        misMatchTemp = f2i_x86((float)r1 / misGrad);
        segMatchTemp = segGrad != 0 ? f2i_x86(7 + (float)r1 / segGrad) : 7;
        for (int c = 0; c < r1 + adaEdge; ++c) {
            const int rc = rd(ref, readLen, c);
            if ((uint8_t)contam[contamLen - r1 - adaEdge + c] == rc) {
                maxSegMatch++;
                if (maxSegMatch >= segMatchTemp) return 0;
            } else if (rc != 'N') {
                mis++;
                maxSegMatch = 0;
                if (mis > misMatchTemp) break;
            }
        }
        if (mis <= misMatchTemp) return 0;
    }
    for (r1 = 0; r1 <= readLen - contamLen; ++r1) {                           /* :549-573 */
        maxSegMatch = 0;
        mis = 0;
        for (int c = 0; c < contamLen; ++c) {
            const int rc = rd(ref, readLen, r1 + c);
            if ((uint8_t)contam[c] == rc) {
                maxSegMatch++;
                if (maxSegMatch >= segMatchThr) return r1;
            } else if (rc != 'N') {
                mis++;
                maxSegMatch = 0;
                if (mis > adaMis) break;
            }
        }
        if (mis <= adaMis) return r1;
    }
    for (r1 = 0; r1 < contamLen - adaEdge; ++r1) {                            /* :575-601: no segGrad guard here */
        mis = 0;
        maxSegMatch = 0;
        misMatchTemp = f2i_x86((float)r1 / misGrad);
        segMatchTemp = f2i_x86(7 + (float)r1 / segGrad);
        for (int c = 0; c < r1 + adaEdge; ++c) {
            const int rc = rd(ref, readLen, readLen - r1 - adaEdge + c);
            if ((uint8_t)contam[c] == rc) {
                maxSegMatch++;
                if (maxSegMatch >= segMatchTemp) return readLen - r1 - adaEdge;
            } else if (rc != 'N') {
                mis++;
                maxSegMatch = 0;
                if (mis > misMatchTemp) break;
            }
        }
        if (mis <= misMatchTemp) return readLen - r1 - adaEdge;
    }
    return -1;
}

/* global_contam_pos(), src/read_filter.cpp:961-1062.  total_score / overlap are carried from one
 * alignment to the next inside each of the three sections, exactly as there.                    */
int snk_oracle_global_contam_pos(const uint8_t *ref, int rl, const char *gc, int cl,
                                 float min_matchRatio, int mismatch_number) {
    const int mismatch_score = -200, match_score = 1;
    const int total_mismatch_score = mismatch_number * mismatch_score;
    const int min_match_len = (int)((float)cl * min_matchRatio);
    const int lower_score = (min_match_len - mismatch_number) + total_mismatch_score;
    int total_score = -1000, overlap = 0;
    for (int i = cl - min_match_len; i >= 0; i--) {                           /* front, :973-999 */
        const int j_max = cl - i > rl ? rl : cl - i;
        for (int j = 0; j != j_max; j++) {
            if (ref[j] == (uint8_t)gc[i + j]) {
                if (total_score > total_mismatch_score) { total_score += match_score; overlap++; }
                else {
                    if (j_max - j < min_match_len) break;
                    total_score = match_score;
                    overlap = 1;
                }
            } else {
                if (total_score > total_mismatch_score) { total_score += mismatch_score; overlap++; }
                else if (j_max - j < min_match_len) break;
            }
            if (total_score >= lower_score && overlap >= min_match_len) return 0;
        }
    }
    total_score = -1000;                                                      /* middle, :1001-1029 */
    overlap = 0;
    for (int i = 0; i <= rl - cl; i++) {
        for (int j = 0; j != cl; j++) {
            if (ref[i + j] == (uint8_t)gc[j]) {
                if (total_score > total_mismatch_score) { total_score += match_score; overlap++; }
                else {
                    if (cl - j < min_match_len) break;
                    total_score = match_score;
                    overlap = 1;
                }
            } else {
                if (total_score > total_mismatch_score) { total_score += mismatch_score; overlap++; }
                else if (cl - j < min_match_len) break;
            }
            if (total_score >= lower_score && overlap >= min_match_len) return i + j - overlap + 1;
        }
    }
    total_score = -1000;                                                      /* tail, :1031-1060 */
    overlap = 0;
    const int i_min = cl > rl ? cl - rl : 0;
    for (int i = i_min; i <= cl - min_match_len; i++) {
        for (int j = 0; j != cl - i; j++) {
            if (ref[rl - (cl - i) + j] == (uint8_t)gc[j]) {
                if (total_score > total_mismatch_score) { total_score += match_score; overlap++; }
                else {
                    total_score = match_score;
                    overlap = 1;
                    if (cl - i - j < min_match_len) break;
                }
            } else {
                if (total_score > total_mismatch_score) { total_score += mismatch_score; overlap++; }
                else if (cl - i - j < min_match_len) break;
            }
            if (total_score >= lower_score && overlap >= min_match_len) return rl - cl + i + j - overlap + 1;
        }
    }
    return -1;
}

/* the contaminant lists of one context, parsed once per batch */
#define ORC_MAX_CT 16
typedef struct {
    int n[2];                                   /* contam1 / contam2 entries */
    char seq[2][ORC_MAX_CT][260];
    int len[2][ORC_MAX_CT], thr[2][ORC_MAX_CT];
    int ng;                                     /* global contaminants */
    char gseq[ORC_MAX_CT][2][260];              /* forward, reverse complement */
    int glen[ORC_MAX_CT], gmm[ORC_MAX_CT];
    float gmr[ORC_MAX_CT];
    int bad;                                    /* list-size mismatch: the reference exits */
} contam_cfg;

static int split_commas(const char *s, char out[][260], int maxn) {
    int n = 0;
    if (!s) return 0;
    const char *p = s;
    for (;;) {
        const char *e = strchr(p, ',');
        size_t l = e ? (size_t)(e - p) : strlen(p);
        if (n < maxn) {
            if (l > 259) l = 259;
            memcpy(out[n], p, l);
            out[n][l] = 0;
            n++;
        }
        if (!e) break;
        p = e + 1;
    }
    return n;
}

static void parse_contams(const snk_params *P, contam_cfg *C) {
    memset(C, 0, sizeof(*C));
    static char mrs[ORC_MAX_CT][260], gm[ORC_MAX_CT][260], gfw[ORC_MAX_CT][260];
    for (int m = 0; m < 2; m++) {
        const char *cs = P->contam[m];
        if (!cs || !*cs) continue;
        const char *mr = P->ct_match_r ? P->ct_match_r : "0.2";
        if (!strchr(cs, ',')) {                                   /* src/read_filter.cpp:190-193 + :616 (double) */
            C->n[m] = 1;
            strncpy(C->seq[m][0], cs, 259);
            C->len[m][0] = (int)strlen(C->seq[m][0]);
            C->thr[m][0] = (int)ceil((double)C->len[m][0] * atof(mr));
        } else {                                                  /* hasContams, :483-506 (float) */
            C->n[m] = split_commas(cs, C->seq[m], ORC_MAX_CT);
            const int nm = strchr(mr, ',') ? split_commas(mr, mrs, ORC_MAX_CT) : -1;
            if (nm != C->n[m]) { C->bad = 1; return; }
            for (int i = 0; i < C->n[m]; i++) {
                C->len[m][i] = (int)strlen(C->seq[m][i]);
                const float tmp_mr = (float)atof(mrs[i]);
                C->thr[m][i] = (int)ceilf((float)C->len[m][i] * tmp_mr);
            }
        }
    }
    if (P->global_contams && *P->global_contams) {                /* hasGlobalContams, :927-960 */
        C->ng = split_commas(P->global_contams, gfw, ORC_MAX_CT);
        const int a = split_commas(P->g_mrs ? P->g_mrs : "", mrs, ORC_MAX_CT), b = split_commas(P->g_mms ? P->g_mms : "", gm, ORC_MAX_CT);
        if (a != C->ng || b != C->ng) { C->bad = 1; return; }
        for (int i = 0; i < C->ng; i++) {
            const int l = (int)strlen(gfw[i]);
            C->glen[i] = l;
            C->gmr[i] = (float)atof(mrs[i]);
            C->gmm[i] = atoi(gm[i]);
            memcpy(C->gseq[i][0], gfw[i], (size_t)l + 1);
            for (int k = 0; k < l; k++) {                         /* reversecomplementary(), :1064-1090 */
                const int ch = toupper((unsigned char)gfw[i][l - 1 - k]);
                char o;
                switch (ch) {
                case 'A': o = 'T'; break;
                case 'T': o = 'A'; break;
                case 'C': o = 'G'; break;
                case 'G': o = 'C'; break;
                case 'N': o = 'N'; break;
                default: C->bad = 1; return;                       /* "Error:unrecognized base" */
                }
                C->gseq[i][1][k] = o;
            }
            C->gseq[i][1][l] = 0;
        }
    }
}

/* include_contam / include_global_contam of one read, src/read_filter.cpp:189-248 */
static void contam_flags(const snk_params *P, const contam_cfg *C, int mate, const uint8_t *seq, int len,
                         int *inc_contam, int *inc_global) {
    *inc_contam = *inc_global = 0;
    /* mate 2 of a pair sees gp2 = gp with adaMis2 / adaEdge2 (src/sequence.cpp:182-189) */
    const int pm = (P->paired && mate == 1) ? 1 : 0;
    for (int i = 0; i < C->n[mate]; i++) {
        const int pos = snk_oracle_has_contam(seq, len, C->seq[mate][i], C->len[mate][i], C->thr[mate][i],
                                              P->ada_mis[pm], P->ada_edge[pm]);
        if (pos >= 0) { *inc_contam = 1; break; }                 /* later entries cannot change the verdict */
    }
    for (int i = 0; i < C->ng && !*inc_global; i++)
        for (int d = 0; d < 2 && !*inc_global; d++)
            if (snk_oracle_global_contam_pos(seq, len, C->gseq[i][d], C->glen[i], C->gmr[i], C->gmm[i]) >= 0) *inc_global = 1;
}

/* C_fastq_stat_result + the C_fastq cut fields of the filter's private copy */
typedef struct {
    int len;
    int a, c, g, t, n;
    int contig;
    int lowq, sumq;
    float n_ratio, a_ratio, lowq_ratio, mean_q;
    int include_adapter;                 /* include_adapter_seq: 1 / -1 */
    int inc_contam, inc_gcontam;         /* include_contam, include_global_contam */
    int hd_h, lq_h, hd_t, lq_t, adacut;  /* -1 == unset */
    int start, clen;                     /* trimmed view */
} rd_t;

/* stat_read(), src/read_filter.cpp:80-313 (the tile/fov strings of :86-150 depend on the read name only:
 * the host decides them and passes the verdict as bits 1-2 of snk_batch.dup)   */
static const contam_cfg *g_ct = NULL;          /* contaminant lists of the batch being processed */
static int stat_read(const snk_params *P, int mate, const uint8_t *seq,
                     const uint8_t *qual, int len, rd_t *r) {
    memset(r, 0, sizeof(*r));
    r->len = len;
    r->include_adapter = -1;                                       /* :83 */
    r->hd_h = r->lq_h = r->hd_t = r->lq_t = r->adacut = -1;        /* C_fastq_init */
    int ada_pos = -1;                                              /* :175-188 */
    for (int i = 0; i < P->n_adapters[mate]; i++) {
        const char *ad = snk_adapter_at(P, mate, i);
        ada_pos = snk_oracle_adapter_pos(seq, len, ad, (int)strlen(ad),
                                         P->ada_mis[mate], P->ada_mr[mate],
                                         P->ada_edge[mate]);
        if (ada_pos >= 0) break;
    }
    if (ada_pos >= 0) {
        r->include_adapter = 1;
        r->adacut = len - ada_pos;
    }
    if (g_ct) contam_flags(P, g_ct, mate, seq, len, &r->inc_contam, &r->inc_gcontam);   /* :189-248 */
    if (len == 0) return SNK_E_EMPTY_SEQ;                          /* :250-253 */
    int last_char = 'Q', contig_base = 0, max_contig = 1;          /* :255-257 */
    for (int ix = 0; ix < len; ix++) {                             /* :258-287 */
        if (P->polyX_num != -1) {
            if (seq[ix] == last_char) {
                contig_base++;
                if (max_contig < contig_base) max_contig = contig_base;
            } else {
                contig_base = 1;
            }
        }
        last_char = seq[ix];
        switch (seq[ix]) {
        case 'a': case 'A': r->a++; break;
        case 'c': case 'C': r->c++; break;
        case 'g': case 'G': r->g++; break;
        case 't': case 'T': r->t++; break;
        case 'n': case 'N': r->n++; break;
        default: return SNK_E_BAD_BASE;                            /* :282-285 */
        }
    }
    r->contig = max_contig;
    r->a_ratio = (float)r->a / (float)(size_t)len;                 /* :290 */
    r->n_ratio = (float)r->n / (float)(size_t)len;                 /* :294 */
    int total = 0;
    for (int ix = 0; ix < len; ix++) {                             /* :299-308 */
        int bq = (int)qual[ix] - P->quality_phred;
        total += bq;
        if (bq <= P->low_qual) r->lowq++;
    }
    r->sumq = total;
    r->lowq_ratio = (float)r->lowq / (float)(size_t)len;           /* :310 */
    r->mean_q = (float)total / (float)(size_t)len;                 /* :311 */
    r->start = 0;
    r->clen = len;
    return SNK_OK;
}

/* fastq_trim(), src/read_filter.cpp:338-471 (index removal :357-382 only edits
 * the ID, which never crosses the boundary)                                   */
static void fastq_trim(const snk_params *P, int mate, const uint8_t *seq,
                       const uint8_t *qual, rd_t *r) {
    int ht_flag = P->has_hard_trim, lqt_flag = P->has_lq_trim;
    int ada_trim_flag = P->ada_trim, contam_trim_flag = P->contam_trim;
    if (!ht_flag && !lqt_flag && !ada_trim_flag && !contam_trim_flag &&
        P->polyG_tail == -1)
        return;                                                    /* :354 */
    int len = r->len;
    int head_cut = 0, tail_cut = 0;
    if (ht_flag) {                                                 /* :384-389 */
        r->hd_h = P->hard_trim[P->paired ? 2 * mate : 0];
        r->hd_t = P->hard_trim[P->paired ? 2 * mate + 1 : 1];
        head_cut = r->hd_h;
        tail_cut = r->hd_t;
    }
    if (lqt_flag) {                                                /* :390-429 */
        int head_ix = 0, tail_ix = 0;
        /* Quirk Q11: neither loop checks the read length (:411-426).  With a limit above the
         * length and every base below the threshold the reference indexes qual_seq[size] ('\0', defined), then
         * past it / before qual_seq[0] (heap bytes: undefined).  Here everything outside [0,len) reads as 0,
         * i.e. the run goes on to the limit whenever 0 - phred < threshold.  Generators keep limits <= the
         * shortest read where the compiled reference is the judge. */
        for (int ix = 0; ix < P->lq_head_len; ix++) {
            int bq = rd(qual, len, ix) - P->quality_phred;
            if (bq < P->lq_head_qual) head_ix++;
            else break;
        }
        for (int ix = 0; ix < P->lq_tail_len; ix++) {
            int bq = rd(qual, len, len - ix - 1) - P->quality_phred;
            if (bq < P->lq_tail_qual) tail_ix++;
            else break;
        }
        r->lq_h = head_ix;
        r->lq_t = tail_ix;
        head_cut = head_cut >= head_ix ? head_cut : head_ix;
        tail_cut = tail_cut >= tail_ix ? tail_cut : tail_ix;
    }
    if (ada_trim_flag) {                                           /* :430-442 */
        if (r->adacut > 0) tail_cut = tail_cut >= r->adacut ? tail_cut : r->adacut;
    }
    if (P->polyG_tail != -1) {                                     /* :454-461 */
        int polyG_n = snk_oracle_polyG_number(seq, len);
        if ((float)polyG_n >= P->polyG_tail) {
            if (polyG_n > tail_cut) tail_cut = polyG_n;
        }
    }
    /* :462-468, int + int compared with size_t */
    if ((uint64_t)(int64_t)(head_cut + tail_cut) > (uint64_t)len) {
        r->start = 0;
        r->clen = 0;
    } else {
        r->clen = len - head_cut - tail_cut;
        r->start = r->clen ? head_cut : 0;   /* record convention: empty view starts at 0 */
    }
}

static int pe_dis(int a, int b) { return (a ? 1 : 0) + (b ? 2 : 0); }  /* src/sequence.cpp:392 */

static void fam(uint64_t *fs, int base, int v) {   /* the switch(v) blocks, e.g. src/sequence.cpp:235-241 */
    if (v == 1) fs[base + 1]++;
    else if (v == 2) fs[base + 2]++;
    else if (v == 3) { fs[base + 1]++; fs[base + 2]++; fs[base + 3]++; }
    fs[base]++;
}

/* pe_discard(), src/sequence.cpp:198-387 */
static int pe_discard(const snk_params *P, const rd_t *r1, const rd_t *r2, int dup,
                      uint64_t *fs, int *vout) {
    int v;
    *vout = 0;
    if (P->rmdup && (dup & 1)) { fs[SNK_FS_DUP]++; return SNK_R_DUP; }
    if (dup & 2) { fs[SNK_FS_TILE]++; return SNK_R_TILE; }         /* :213-231, verdict of fq1's read name */
    if (dup & 4) { fs[SNK_FS_FOV]++; return SNK_R_FOV; }
    if (P->min_read_length != -1) {                                /* :232-249 */
        v = pe_dis((uint64_t)r1->clen < (uint64_t)(int64_t)P->min_read_length,
                   (uint64_t)r2->clen < (uint64_t)(int64_t)P->min_read_length);
        if (v > 0) { fam(fs, SNK_FS_SHORT, v); *vout = v; return SNK_R_SHORT; }
    } else if (r1->clen == 0 || r2->clen == 0) {
        return SNK_R_EMPTY;
    }
    if (P->max_read_length != -1) {                                /* :250-263 */
        v = pe_dis((uint64_t)r1->clen > (uint64_t)(int64_t)P->max_read_length,
                   (uint64_t)r2->clen > (uint64_t)(int64_t)P->max_read_length);
        if (v > 0) { fam(fs, SNK_FS_LONG, v); *vout = v; return SNK_R_LONG; }
    }
    if (!P->contam_trim) {                                         /* :264-290 (global first in PE) */
        v = pe_dis(r1->inc_gcontam == 1, r2->inc_gcontam == 1);
        if (v > 0) { fam(fs, SNK_FS_GCONTAM, v); *vout = v; return SNK_R_GCONTAM; }
        v = pe_dis(r1->inc_contam == 1, r2->inc_contam == 1);
        if (v > 0) { fam(fs, SNK_FS_CONTAM, v); *vout = v; return SNK_R_CONTAM; }
    }
    if (P->n_ratio != -1) {                                        /* :291-303 */
        v = pe_dis(r1->n_ratio >= P->n_ratio, r2->n_ratio >= P->n_ratio);
        if (v > 0) { fam(fs, SNK_FS_NRATE, v); *vout = v; return SNK_R_NRATE; }
    }
    if (P->highA_ratio != -1) {                                    /* :305-317 */
        v = pe_dis(r1->a_ratio >= P->highA_ratio, r2->a_ratio >= P->highA_ratio);
        if (v > 0) { fam(fs, SNK_FS_HIGHA, v); *vout = v; return SNK_R_HIGHA; }
    }
    if (P->polyX_num != -1) {                                      /* :318-330 */
        v = pe_dis(r1->contig >= P->polyX_num, r2->contig >= P->polyX_num);
        if (v > 0) { fam(fs, SNK_FS_POLYX, v); *vout = v; return SNK_R_POLYX; }
    }
    if (P->low_qual_ratio != -1) {                                 /* :332-349 */
        v = pe_dis(r1->lowq_ratio >= P->low_qual_ratio, r2->lowq_ratio >= P->low_qual_ratio);
        if (v > 0) { fam(fs, SNK_FS_LOWQUAL, v); *vout = v; return SNK_R_LOWQUAL; }
    }
    if (P->mean_quality != -1) {                                   /* :351-363 */
        v = pe_dis(r1->mean_q < (float)P->mean_quality, r2->mean_q < (float)P->mean_quality);
        if (v > 0) { fam(fs, SNK_FS_MEANQ, v); *vout = v; return SNK_R_MEANQ; }
    }
    /* :364-371 over_lapped is forced false, src/sequence.cpp:195 */
    if (!P->ada_trim) {                                            /* :372-384 */
        v = pe_dis(r1->include_adapter == 1, r2->include_adapter == 1);
        if (v > 0) { fam(fs, SNK_FS_ADAPTER, v); *vout = v; return SNK_R_ADAPTER; }
    }
    return SNK_KEEP;
}

/* se_discard(), src/sequence.cpp:76-178 */
static int se_discard(const snk_params *P, const rd_t *r, int dup, uint64_t *fs) {
    if (P->rmdup && (dup & 1)) { fs[SNK_FS_DUP]++; return SNK_R_DUP; }
    if (dup & 2) { fs[SNK_FS_TILE]++; return SNK_R_TILE; }         /* :84-99 */
    if (dup & 4) { fs[SNK_FS_FOV]++; return SNK_R_FOV; }
    if (P->min_read_length != -1 &&
        (uint64_t)r->clen < (uint64_t)(int64_t)P->min_read_length) { fs[SNK_FS_SHORT]++; return SNK_R_SHORT; }
    if (P->max_read_length != -1 &&
        (uint64_t)r->clen > (uint64_t)(int64_t)P->max_read_length) { fs[SNK_FS_LONG]++; return SNK_R_LONG; }
    if (!P->contam_trim) {                                         /* :116-128 (contam first in SE) */
        if (r->inc_contam == 1) { fs[SNK_FS_CONTAM]++; return SNK_R_CONTAM; }
        if (r->inc_gcontam == 1) { fs[SNK_FS_GCONTAM]++; return SNK_R_GCONTAM; }
    }
    if (P->n_ratio != -1 && r->n_ratio >= P->n_ratio) { fs[SNK_FS_NRATE]++; return SNK_R_NRATE; }
    if (P->highA_ratio != -1 && r->a_ratio >= P->highA_ratio) { fs[SNK_FS_HIGHA]++; return SNK_R_HIGHA; }
    if (P->polyX_num != -1 && r->contig >= P->polyX_num) { fs[SNK_FS_POLYX]++; return SNK_R_POLYX; }
    if (P->low_qual_ratio != -1 && r->lowq_ratio >= P->low_qual_ratio) { fs[SNK_FS_LOWQUAL]++; return SNK_R_LOWQUAL; }
    if (P->mean_quality != -1 && r->mean_q < (float)P->mean_quality) { fs[SNK_FS_MEANQ]++; return SNK_R_MEANQ; }
    if (r->include_adapter == 1 && !P->ada_trim) { fs[SNK_FS_ADAPTER]++; return SNK_R_ADAPTER; }
    return SNK_KEEP;
}

static void ts_inc(uint64_t *ts, long idx) {
    if (idx >= 0 && idx < SNK_TS_N) ts[idx]++;   /* outside the struct the reference is UB */
}

/* trimming-position part of stat_pe_fqs / stat_se_fqs,
 * src/peprocess.cpp:1107-1143 (fq1), :1325-1360 (fq2), src/seprocess.cpp:647-682 */
static void ts_update(uint64_t *ts, int hd_h, int lq_h, int hd_t, int lq_t, int ada,
                      long base_len, int se) {
    if (hd_h > 0 || lq_h > 0) {
        if (hd_h >= lq_h) ts_inc(ts, SNK_TS_HT + hd_h);
        else ts_inc(ts, SNK_TS_HLQ + lq_h);
    }
    if (hd_t > 0 || lq_t > 0 || (se ? ada >= 0 : ada > 0)) {
        if (hd_t >= lq_t) {
            if (hd_t >= ada) ts_inc(ts, SNK_TS_TT + base_len - hd_t + 1);
            else ts_inc(ts, SNK_TS_TA + base_len - ada + 1);
        } else {
            if (lq_t >= ada) ts_inc(ts, SNK_TS_TLQ + base_len - lq_t + 1);
            else ts_inc(ts, SNK_TS_TA + base_len - ada + 1);
        }
    }
}

/* per-base part of stat_pe_fqs, src/peprocess.cpp:1144-1203 */
static int file_stat_read(const snk_params *P, uint64_t *file, int lcap, int nq,
                          const uint8_t *seq, const uint8_t *qual, int start, int n) {
    uint64_t *gs = file, *bs = file + snk_bs_off(lcap, nq), *qs = file + snk_qs_off(lcap, nq);
    int rc = SNK_OK;
    for (int i = 0; i < n; i++) {
        int b;
        switch (seq[start + i]) {
        case 'a': case 'A': b = 0; break;
        case 'c': case 'C': b = 1; break;
        case 'g': case 'G': b = 2; break;
        case 't': case 'T': b = 3; break;
        case 'n': case 'N': b = 4; break;
        default: return SNK_E_BAD_BASE;
        }
        bs[(int64_t)i * 5 + b]++;
        gs[SNK_GS_A + b]++;
        int bq = (int)qual[start + i] - P->quality_phred;
        if (bq < 0 || bq >= nq) { rc = SNK_E_QUAL_RANGE; continue; }
        qs[(int64_t)i * nq + bq]++;
        if (bq >= 20) gs[SNK_GS_Q20]++;
        if (bq >= 30) gs[SNK_GS_Q30]++;
    }
    gs[SNK_GS_READS]++;
    gs[SNK_GS_BASES] += (uint64_t)n;
    return rc;
}

static void set_rec(snk_read_result *o, const rd_t *r, int reason, int v) {
    o->head_hdcut = (int16_t)r->hd_h;
    o->head_lqcut = (int16_t)r->lq_h;
    o->tail_hdcut = (int16_t)r->hd_t;
    o->tail_lqcut = (int16_t)r->lq_t;
    o->adacut_pos = (int16_t)r->adacut;
    o->clean_start = (uint16_t)r->start;
    o->clean_len = (uint16_t)r->clen;
    o->reason = (uint8_t)reason;
    o->flags = (uint8_t)v;
}

int snk_oracle_filter_batch(const snk_params *P, const snk_batch *B,
                            snk_read_result *out1, snk_read_result *out2,
                            uint64_t *sum, uint64_t *maxb, snk_error *err) {
    const int lcap = P->max_read_len, nq = P->max_base_quality + 1;
    const int pe = P->paired ? 1 : 0;
    uint64_t *fs = sum;
    uint64_t *file[4];
    for (int k = 0; k < 4; k++) file[k] = sum + snk_file_off(lcap, nq, k);
    /* src/peprocess.cpp:1441 / src/seprocess.cpp:881 */
    const int copy_back = P->ada_trim || P->contam_trim || P->has_hard_trim || P->has_lq_trim;
    if (err) { err->code = SNK_OK; err->mate = 0; err->index = 0; }
    static contam_cfg ct;                                         /* (the oracle is single-threaded test code) */
    parse_contams(P, &ct);
    if (ct.bad) { if (err) err->code = SNK_E_PARAM; return SNK_E_PARAM; }
    g_ct = (ct.n[0] || ct.n[1] || ct.ng) ? &ct : NULL;

    for (int64_t i = 0; i < B->n; i++) {
        rd_t r[2];
        const uint8_t *s[2], *q[2];
        int rc = SNK_OK, emate = 0;
        for (int m = 0; m <= pe; m++) {
            int len = B->len[m] ? B->len[m][i] : B->fixed_len[m];
            s[m] = B->seq[m] + (int64_t)i * B->pitch;
            q[m] = B->qual[m] + (int64_t)i * B->pitch;
            if (len > lcap) { rc = SNK_E_TOO_LONG; emate = m; break; }
            rc = stat_read(P, m, s[m], q[m], len, &r[m]);           /* ctor: src/sequence.cpp:12-15,182-196 */
            if (rc) { emate = m; break; }
        }
        if (rc) {
            if (err) { err->code = rc; err->mate = emate; err->index = B->first_index + (uint64_t)i; }
            return rc;
        }
        for (int m = 0; m <= pe; m++) fastq_trim(P, m, s[m], q[m], &r[m]); /* pe_trim */

        int dup = B->dup ? B->dup[i] : 0, v = 0, reason;
        if (pe) reason = pe_discard(P, &r[0], &r[1], dup, fs, &v);
        else reason = se_discard(P, &r[0], dup, fs);
        set_rec(&out1[i], &r[0], reason, v);
        if (pe) set_rec(&out2[i], &r[1], reason, v);

        uint64_t key = (B->first_index + (uint64_t)i + 1) << 16;
        for (int m = 0; m <= pe; m++) {
            /* stat("raw"): raw record = cut fields copied back only when a trim
             * option is on; raw_length stays 0                    */
            int hh = -1, lh = -1, ht = -1, lt = -1, ad = -1;
            if (copy_back) { hh = r[m].hd_h; lh = r[m].lq_h; ht = r[m].hd_t; lt = r[m].lq_t; ad = r[m].adacut; }
            long base_len = (pe && m == 1) ? r[m].len : 0;
            ts_update(file[m] + snk_ts_off(lcap, nq), hh, lh, ht, lt, ad, base_len, !pe);
            rc = file_stat_read(P, file[m], lcap, nq, s[m], q[m], 0, r[m].len);
            if (rc) {
                if (err) { err->code = rc; err->mate = m; err->index = B->first_index + (uint64_t)i; }
                return rc;
            }
            if ((key | (uint64_t)r[m].len) > maxb[m]) maxb[m] = key | (uint64_t)r[m].len;
        }
        if (reason == SNK_KEEP) {
            for (int m = 0; m <= pe; m++) {
                /* stat("clean"): the filter's trimmed copy; fq1/SE use
                 * raw_length, fq2 uses the trimmed sequence.size()            */
                long base_len = (pe && m == 1) ? r[m].clen : r[m].len;
                ts_update(file[2 + m] + snk_ts_off(lcap, nq), r[m].hd_h, r[m].lq_h, r[m].hd_t,
                          r[m].lq_t, r[m].adacut, base_len, !pe);
                file_stat_read(P, file[2 + m], lcap, nq, s[m], q[m], r[m].start, r[m].clen);
                if ((key | (uint64_t)r[m].clen) > maxb[2 + m]) maxb[2 + m] = key | (uint64_t)r[m].clen;
            }
        }
    }
    return SNK_OK;
}

/* ------------------------------------------------------------------ rmdup pre-pass */

static uint64_t shift_mix(uint64_t v) { return v ^ (v >> 47); }

/* libstdc++ _Hash_bytes, size_t == 8 (hash_bytes.cc), seed as std::hash<std::string> passes it */
uint64_t snk_oracle_hash_bytes(const void *ptr, uint64_t len) {
    const uint64_t mul = (((uint64_t)0xc6a4a793UL) << 32) + (uint64_t)0x5bd1e995UL;
    const uint8_t *buf = (const uint8_t *)ptr;
    const uint64_t len_aligned = len & ~(uint64_t)7;
    uint64_t hash = (uint64_t)0xc70f6907UL ^ (len * mul);
    for (uint64_t o = 0; o < len_aligned; o += 8) {
        uint64_t w;
        memcpy(&w, buf + o, 8);                               /* unaligned_load, little endian */
        hash ^= shift_mix(w * mul) * mul;
        hash *= mul;
    }
    if (len & 7) {
        uint64_t w = 0;                                       /* load_bytes: little-endian partial word */
        for (int i = (int)(len & 7) - 1; i >= 0; --i) w = (w << 8) + buf[len_aligned + i];
        hash ^= w;
        hash *= mul;
    }
    hash = shift_mix(hash) * mul;
    hash = shift_mix(hash);
    return hash;
}

void snk_oracle_hash_batch(const snk_batch *B, int paired, uint64_t *out) {
    uint8_t *tmp = (uint8_t *)malloc(2 * (size_t)B->pitch + 8);
    for (int64_t i = 0; i < B->n; ++i) {
        const int l1 = B->len[0] ? B->len[0][i] : B->fixed_len[0];
        int l2 = 0;
        memcpy(tmp, B->seq[0] + i * (int64_t)B->pitch, l1);
        if (paired) {
            l2 = B->len[1] ? B->len[1][i] : B->fixed_len[1];
            memcpy(tmp + l1, B->seq[1] + i * (int64_t)B->pitch, l2);      /* fq1seq + fq2seq, :3665 */
        }
        out[i] = snk_oracle_hash_bytes(tmp, (uint64_t)(l1 + l2));
    }
    free(tmp);
}

uint32_t snk_oracle_rmdup_prime(uint64_t n) {
    const uint32_t real = n > 4294967295ull ? 4294967295u : (uint32_t)n;
    if (n > 0 && n < 10) return (uint32_t)n;
    uint32_t cur = real;
    while (cur--) {                                           /* first candidate: real - 1 */
        int is_prime = 1;
        for (uint32_t j = 2; (uint64_t)j * j <= cur; ++j)
            if (cur % j == 0) { is_prime = 0; break; }
        if (is_prime) return cur;
    }
    return 0;                                                 /* n == 0: the reference exits "code error" */
}

void snk_oracle_markdup(const uint64_t *hash, uint64_t n, uint8_t *dup) {
    memset(dup, 0, n);
    if (n == 0) return;
    /* earlier-occurrence test with an open-addressing set of the values seen so far */
    uint64_t cap = 16;
    while (cap < 2 * n) cap <<= 1;
    uint64_t *keys = (uint64_t *)malloc(cap * sizeof(uint64_t));
    uint8_t *used = (uint8_t *)calloc(cap, 1);
    const uint32_t prime = snk_oracle_rmdup_prime(n);
    const uint64_t minus1 = ~(uint64_t)0;
    uint64_t sentinel_bucket = 0;                             /* elements sharing the bucket of 2^64-1 */
    for (uint64_t i = 0; i < n; ++i)
        if (hash[i] % prime == minus1 % prime) sentinel_bucket++;
    for (uint64_t i = 0; i < n; ++i) {
        const uint64_t h = hash[i];
        uint64_t s = (h * 0x9E3779B97F4A7C15ull) >> 7 & (cap - 1);
        int seen = 0;
        while (used[s]) {
            if (keys[s] == h) { seen = 1; break; }
            s = (s + 1) & (cap - 1);
        }
        if (!seen) { used[s] = 1; keys[s] = h; }
        dup[i] = (uint8_t)(seen || (h == minus1 && sentinel_bucket > 1));
    }
    free(keys);
    free(used);
}
