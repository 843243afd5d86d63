// snk_tables.h -- host-side tables of one adapter: everything adapter_pos() (src/read_filter.cpp:707-790) derives from
// (adptLen, adaMis, adaMR, adaEdge) with the reference's own int -> float -> int arithmetic, and the compact descriptor the
// bit-sliced search reads (snk_adapter_bits.hip.h).  Plain C++ (no HIP): shared by the C ABI (snk_filter.cpp) and by the host
// emulation of the search that fuzzes it against the oracle without a GPU (tests/host_emul/).
#pragma once
#include <limits.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "snk_device.h"

namespace {

// SNK_PROVEN_ONLY=1 in the environment: dispatch only to paths that have passed the GPU parity suite on hardware -- the automatic
// choice of snk_filter_batch_device() becomes kernel 1 (generic decisions + LDS histograms, csrc/snk_filter.cpp), the adapter
// envelope of an explicitly requested tiled kernel is the hardware-green one (below), and the CLI reads the same switch for
// single-end rmdup in one pass (host/snk_main.cpp)
inline bool snk_proven_only() {
    static const bool v = [] { const char *e = getenv("SNK_PROVEN_ONLY"); return e && e[0] == '1' && !e[1]; }();
    return v;
}

// float -> int as the x86-64 reference build does it (cvttss2si): NaN/overflow -> INT_MIN
inline int f2i_x86(float f) {
    if (!(f == f)) return INT_MIN;
    if (f >= 2147483648.0f || f < -2147483648.0f) return INT_MIN;
    return (int)f;
}

inline void build_adapter(DevAdapter &A, const char *seq, int mis, float mr, int edge) {
    memset(&A, 0, sizeof(A));
    const int al = (int)strlen(seq);
    A.len = al;
    A.mis = mis;
    A.edge = edge;
    A.nC = al - edge;
    memcpy(A.seq, seq, al);
    // src/read_filter.cpp:714-717 : integer division, then float
    const float misGrad5 = (float)((al - 5) / (mis + 1));
    const float misGrad = (float)((al - edge) / (mis + 1));
    A.S = (int)ceilf((float)al * mr);
    int maxb = mis > 0 ? mis : 0;
    for (int r1 = 1; r1 <= 5; ++r1) {
        A.budgetA[r1] = f2i_x86((float)(al - r1) / misGrad5);
        if (A.budgetA[r1] > maxb) maxb = A.budgetA[r1];
    }
    for (int r1 = 0; r1 < A.nC && r1 < SNK_DEV_MAX_ADA_LEN; ++r1) {
        A.budgetC[r1] = f2i_x86((float)r1 / misGrad);
        if (A.budgetC[r1] > maxb) maxb = A.budgetC[r1];
    }
    A.maxBudget = maxb;
    A.negC = (misGrad == 0.0f) ? 1 : 0;
    for (int k = 1; k <= 4; ++k) {                        // rk[0] holds k = 4
        A.rk[k & 3] = A.nC > 0 ? A.nC : 0;
        for (int r1 = 0; r1 < A.nC && r1 < SNK_DEV_MAX_ADA_LEN; ++r1)
            if (A.budgetC[r1] >= k) { A.rk[k & 3] = r1; break; }
    }
    // bit-parallel view for the tiled kernel: code 0..3 = ACGT, 4 = matches no upper-case ACGT read base,
    // 5 = 'N'.  A lower-case adapter character (code 4) can only match a lower-case read character: reads that
    // hold anything but upper-case ACGT take the sequential matcher for such an adapter (has_lower).
    // Adapters of any length the ABI takes (1..255; round 4: the screen looks at the first 64 characters -- still a necessary
    // condition --, survivors of longer adapters are decided character by character; adaEdge may exceed the adapter: no
    // phase C then).  Any budget: beyond 3 the screen lets the offset through.  long_ok: the block-wise search of the long-read
    // kernel lets an adapter reach 64 positions past a block.
    A.tile_ok = (al >= 1 && al < SNK_DEV_MAX_ADA_LEN && edge >= 1 && mis >= 0) ? 1 : 0;
    // SNK_PROVEN_ONLY=1: the envelope the last hardware-green GPUTEST record covers -- 6..64 characters, adaEdge within
    // the adapter -- ; anything else takes the sequential matcher of the generic kernel as it did then
    if (snk_proven_only() && !(al >= 6 && al <= 64 && edge <= al)) A.tile_ok = 0;
    // (round 5: the blocks of the long-read kernel take what the tiled kernel takes -- a block is told how much of the read is left
    // and which offsets are its own, csrc/snk_adapter_bits.hip.h; SNK_PROVEN_ONLY=1 keeps the envelope of the last hardware-green run)
    A.long_ok = (A.tile_ok && (!snk_proven_only() || (al >= 6 && al <= 64 && edge <= al))) ? 1 : 0;
    for (int c = 0; c < al; ++c) {
        int k = 4;
        switch (seq[c]) { case 'A': k = 0; break; case 'C': k = 1; break; case 'G': k = 2; break; case 'T': k = 3; break; case 'N': k = 5; break; default: break; }
        A.code[c] = (uint8_t)k;
        if (k < 4 && c < 64) A.cmask[k] |= 1ull << c;
        if (k == 5 && c < 64) A.nmask |= 1ull << c;
        if (seq[c] == 'a' || seq[c] == 'c' || seq[c] == 'g' || seq[c] == 't' || seq[c] == 'n') A.has_lower = 1;
    }
}

// the compact descriptor of the wave-tiled / long-read kernels
inline void fill_tile_adapter(TileAdapter &T, const DevAdapter &A) {
    memset(&T, 0, sizeof(T));
    T.has_lower = A.has_lower;
    for (int k = 0; k < 4; ++k) T.cmask[k] = A.cmask[k];
    T.nmask = A.nmask;
    for (int ci = 0; ci < 64 && ci < A.len; ++ci) T.code4[ci >> 4] |= (uint64_t)(A.code[ci] & 15) << (4 * (ci & 15));
    T.len = A.len; T.S = A.S; T.mis = A.mis; T.edge = A.edge; T.negC = A.negC;
    T.maxb = A.mis > 0 ? A.mis : 0;
    for (int r1 = 0; r1 < A.nC && r1 < SNK_DEV_MAX_ADA_LEN; ++r1) if (A.budgetC[r1] > T.maxb) T.maxb = A.budgetC[r1];
    for (int k = 0; k < 6; ++k) T.budgetA[k] = A.budgetA[k];
    for (int k = 0; k < 4; ++k) T.rk[k] = A.rk[k];
}

}  // namespace
