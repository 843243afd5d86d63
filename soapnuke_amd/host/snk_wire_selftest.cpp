// snk_wire_selftest -- the sharded run's wires (host/snk_wire.h) on ONE device, world size 1: the only way a one-GPU box reaches the
// dlopen'ed RCCL ABI that `SOAPnuke filter --devices a,b,... ` with SNK_SHARDED=1 uses between its shards (ncclUniqueId by value into
// ncclCommInitRank, the data-type and reduction enum values, grouped ncclSend / ncclRecv).  Test infrastructure (tests/test_wire_gpu.py),
// built next to the CLI by soapnuke_amd/build.py; the reference has no counterpart (its threads share one address space,
// src/peprocess.cpp:1994-2005 merge_stat).
//   snk_wire_selftest [device]      exit 0: every check passed; 77: nothing to test here (no HIP device / no RCCL), reason on stdout; 1: a check failed
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <dlfcn.h>
#include <string>
#include <vector>
#include "snk_wire.h"

static int fail(const char *what, const std::string &why) { printf("FAIL %s: %s\n", what, why.c_str()); return 1; }
#define HIPOK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return fail(#x, hipGetErrorString(e_)); } while (0)

// one wire, world 1: the three reductions leave the words as they are, the exchange with oneself is a copy of exactly count * elem
// bytes (guard bytes behind the receive buffer catch a wrong element size: an `ncclUint8` that meant four bytes would run over)
static int check_wire(snk::ShardWire *w, const char *name) {
    const size_t n = 4097, elem = 13, guard = 256;           // (13: the rmdup exchange sends 8 + 4 + 1 bytes per pair)
    std::vector<uint64_t> h(n), back(n);
    for (size_t i = 0; i < n; ++i) h[i] = 0x9E3779B97F4A7C15ull * (i + 1);
    uint64_t *d = nullptr;
    HIPOK(hipMalloc((void **)&d, n * 8));
    for (int op = 0; op < 3; ++op) {
        HIPOK(hipMemcpy(d, h.data(), n * 8, hipMemcpyHostToDevice));
        if (!w->allreduce_u64(d, n, (snk::WireOp)op)) return fail(name, "allreduce_u64: " + w->err);
        HIPOK(hipMemcpy(back.data(), d, n * 8, hipMemcpyDeviceToHost));
        if (back != h) return fail(name, "allreduce_u64 at world 1 changed the words (op " + std::to_string(op) + ")");
    }
    HIPOK(hipFree(d));
    std::vector<unsigned char> src(n * elem), dst(n * elem + guard);
    for (size_t i = 0; i < src.size(); ++i) src[i] = (unsigned char)(i * 131 + 7);
    unsigned char *ds = nullptr, *dr = nullptr;
    HIPOK(hipMalloc((void **)&ds, src.size()));
    HIPOK(hipMalloc((void **)&dr, dst.size()));
    HIPOK(hipMemcpy(ds, src.data(), src.size(), hipMemcpyHostToDevice));
    HIPOK(hipMemset(dr, 0xA5, dst.size()));
    const uint64_t cnt[1] = {n};
    if (!w->alltoallv(ds, cnt, dr, cnt, elem)) return fail(name, "alltoallv: " + w->err);
    HIPOK(hipMemcpy(dst.data(), dr, dst.size(), hipMemcpyDeviceToHost));
    if (memcmp(dst.data(), src.data(), src.size()) != 0) return fail(name, "alltoallv with oneself is not a copy");
    for (size_t i = src.size(); i < dst.size(); ++i) if (dst[i] != 0xA5) return fail(name, "alltoallv wrote behind count * elem bytes");
    const uint64_t zero[1] = {0};
    if (!w->alltoallv(ds, zero, dr, zero, elem)) return fail(name, "alltoallv of nothing: " + w->err);
    uint64_t one[1] = {42}, got[1] = {0};
    if (!w->exchange_counts(one, got) || got[0] != 42) return fail(name, "exchange_counts: " + w->err);
    HIPOK(hipFree(ds));
    HIPOK(hipFree(dr));
    printf("%s wire, world 1: ok (allreduce sum / max / min of %zu words, alltoallv of %zu x %zu bytes, counts)\n", name, n, n, elem);
    return 0;
}

int main(int argc, char **argv) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { printf("SKIP: no HIP device\n"); return 77; }
    const int dev = argc > 1 ? atoi(argv[1]) : 0;
    if (dev < 0 || dev >= ndev) { printf("SKIP: device %d of %d\n", dev, ndev); return 77; }
    HIPOK(hipSetDevice(dev));
    std::string why;
    const std::string name = "snk-wire-selftest-" + std::to_string((long)getpid());
    snk::HostWire *hw = snk::HostWire::connect(0, 1, name, why, 5);
    if (!hw) return fail("host", why);
    int64_t v = 7;
    char blob[16] = "bootstrap";
    if (!hw->bcast_bytes(blob, sizeof blob) || !hw->min_of_all(v) || v != 7) return fail("host", "bootstrap helpers: " + hw->err);
    if (int rc = check_wire(hw, "host")) return rc;
    delete hw;
#ifdef SNK_SIMT_EMUL
    // (tests/simt: the emulated device is host memory -- RCCL has nothing to talk to; the host wire above ran for real)
    printf("SKIP: RCCL needs a HIP device, this is the emulated build (host wire: checked)\n");
    return 77;
#endif
    const std::string id = snk::RcclWire::make_id(why);
    if (id.empty()) { printf("SKIP: no RCCL communicator id (%s)\n", why.c_str()); return 77; }
    if (id.size() != 256) return fail("RCCL", "communicator id is not 256 hex digits");
    snk::RcclWire *rw = snk::RcclWire::connect(0, 1, id, why);
    if (!rw) return fail("RCCL", "ncclCommInitRank at world 1: " + why);
    if (!rw->nccl_comm()) return fail("RCCL", "no communicator handle");
    if (int rc = check_wire(rw, "RCCL")) return rc;
    // the same communicator through the typed API of the installed header's library, if the loader finds it: rank and size as RCCL sees them
    if (void *h = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD)) {
        typedef int (*q_fn)(void *, int *);
        q_fn count = (q_fn)dlsym(h, "ncclCommCount"), rank = (q_fn)dlsym(h, "ncclCommUserRank");
        int c = -1, r = -1;
        if (count && rank && (count(rw->nccl_comm(), &c) != 0 || rank(rw->nccl_comm(), &r) != 0 || c != 1 || r != 0)) return fail("RCCL", "ncclCommCount / ncclCommUserRank disagree");
        printf("RCCL communicator: %d rank(s), this is rank %d\n", c, r);
    }
    delete rw;
    printf("wire self-test passed\n");
    return 0;
}
