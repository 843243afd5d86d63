// One lane of the bit-sliced adapter search (soapnuke_amd/csrc/snk_adapter_bits.hip.h) on the host: planes of a read built the way
// the kernels hand them over (exact-letter planes, ones beyond the read), adapter_tile<NW, FULL> called as snk_tiled.hip calls it,
// result = adapter_pos().  Exported for tests/test_host_emul.py, which fuzzes it against the oracle and the compiled reference.
#include <hip/hip_runtime.h>
#include "snk_tables.h"
#include "snk_adapter_bits.hip.h"

using namespace snk;

template <int NW, bool FULL>
static int run(const uint8_t *read, int len, const DevAdapter &AG, const TileAdapter &T) {
    u32 X[4][NW], XN[NW];
    bool bad = false;                                        // a character the planes do not hold exactly (lower case, others)
    for (int j = 0; j < NW; ++j) { XN[j] = 0; for (int k = 0; k < 4; ++k) X[k][j] = 0; }
    for (int p = 0; p < 32 * NW; ++p) {
        const u32 bit = 1u << (p & 31);
        if (p >= len) { for (int k = 0; k < 4; ++k) X[k][p >> 5] |= bit; XN[p >> 5] |= bit; continue; }   // beyond the read: matches anything
        const char *f = strchr("ACGT", read[p]);
        if (read[p] && f) X[f - "ACGT"][p >> 5] |= bit;
        else if (read[p] == 'N') XN[p >> 5] |= bit;
        else bad = true;
    }
    if (!FULL) for (int j = 0; j < NW; ++j) XN[j] = 0;       // (the FULL variant alone carries the N plane: need_n selects it)
    return adapter_tile<NW, FULL>(T, AG, X, XN, len, true, read, true, true, T.has_lower != 0 && bad);
}

extern "C" int snk_emul_adapter_pos(const char *read, int len, const char *adapter, int mis, float mr, int edge, int *tile_ok) {
    static DevAdapter AG;
    static TileAdapter T;
    build_adapter(AG, adapter, mis, mr, edge);
    fill_tile_adapter(T, AG);
    if (tile_ok) *tile_ok = AG.tile_ok;
    if (!AG.tile_ok) return -2;
    const bool full = AG.nmask != 0;                         // snk_filter.cpp: need_n
    const int nw = (len + 31) / 32 < 2 ? 2 : (len + 31) / 32;
    const uint8_t *r = (const uint8_t *)read;
#define GO(N) return full ? run<N, true>(r, len, AG, T) : run<N, false>(r, len, AG, T);
    if (nw <= 2) { GO(2) } else if (nw <= 4) { GO(4) } else if (nw <= 5) { GO(5) } else if (nw <= 6) { GO(6) } else if (nw <= 8) { GO(8) }
#undef GO
    return -3;
}
