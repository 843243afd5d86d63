// snk_wire.h -- the collectives between the shard processes of a sharded `SOAPnuke filter` run (host/snk_main.cpp,
// SNK_SHARDED=1: one child process per device on a contiguous range of the input).
//
// The path has two exchange steps and no other: (1) at the end of the run the per-virtual-thread statistics blocks are summed
// (and the max block maxed) over the shards -- what the reference does in merge_stat once its threads have joined,
// src/peprocess.cpp:1994 -- and (2) with `rmdup` every pair's hash travels to its owner (hash % G) with its global index and
// the duplicate flag travels back (rmdup::markDup needs the global input order, src/rmdup.cpp:70-123).  Both run on DEVICE
// memory through this interface, with two engines behind it:
//
//   RcclWire   one RCCL communicator over the shards' GPUs (rank 0 calls ncclGetUniqueId and hands the id to the others over the
//              host wire, every shard calls ncclCommInitRank): ncclAllReduce for (1) -- issued through the C ABI's
//              snk_stats_allreduce() with this communicator -- and grouped ncclSend / ncclRecv pairs for (2), over xGMI.
//   HostWire   the same calls over an abstract Unix-domain socket through rank 0 (buffers staged through host memory): what a
//              run uses when RCCL cannot span the device list -- the same device twice, as the one-GPU test configurations
//              have it, or no RCCL at all as on the CPU emulator of tests/simt --, what SNK_SHARD_WIRE=host selects, the RCCL
//              wire's bootstrap, and what the shards agree to fall back to when one of them cannot join the communicator.
//
// RCCL is resolved with dlopen: the CLI has no link-time dependency on it.
#ifndef SNK_WIRE_H
#define SNK_WIRE_H
#include <dlfcn.h>
#include <errno.h>
#include <stddef.h>
#include <poll.h>
#include <stdint.h>
#include <string.h>
#include <sys/socket.h>
#include <sys/un.h>
#include <time.h>
#include <unistd.h>
#include <algorithm>
#include <string>
#include <thread>
#include <vector>

#include <hip/hip_runtime.h>

namespace snk {

enum WireOp { WIRE_SUM = 0, WIRE_MAX = 1, WIRE_MIN = 2 };

class ShardWire {
public:
    int rank = 0, world = 1;
    std::string err;
    virtual ~ShardWire() {}
    virtual const char *name() const = 0;
    virtual void *nccl_comm() { return nullptr; }          // RcclWire: the ncclComm_t for snk_stats_allreduce()
    // in place on n uint64 words of device memory
    virtual bool allreduce_u64(uint64_t *d_buf, size_t n, WireOp op) = 0;
    // d_send holds send_cnt[0] elements for rank 0, then send_cnt[1] for rank 1, ...; d_recv receives recv_cnt[p] elements
    // from rank p in rank order; `elem` bytes per element
    virtual bool alltoallv(const void *d_send, const uint64_t *send_cnt, void *d_recv, const uint64_t *recv_cnt, size_t elem) = 0;

    // every rank tells every other one number (recv[p] = what rank p had in send[me])
    bool exchange_counts(const uint64_t *send, uint64_t *recv) {
        uint64_t *d = nullptr;
        if (hipMalloc((void **)&d, (size_t)world * 2 * sizeof(uint64_t)) != hipSuccess) { err = "wire: out of device memory"; return false; }
        std::vector<uint64_t> ones((size_t)world, 1);
        bool ok = hipMemcpy(d, send, (size_t)world * 8, hipMemcpyHostToDevice) == hipSuccess &&
                  alltoallv(d, ones.data(), d + world, ones.data(), 8) &&
                  hipMemcpy(recv, d + world, (size_t)world * 8, hipMemcpyDeviceToHost) == hipSuccess;
        (void)hipFree(d);
        if (!ok && err.empty()) err = "wire: count exchange failed";
        return ok;
    }
};

// ------------------------------------------------------------------------------------------------ RCCL
struct NcclUniqueId { char internal[128]; };                 // rccl.h: ncclUniqueId (NCCL_UNIQUE_ID_BYTES)

class RcclWire : public ShardWire {
    typedef int (*get_id_fn)(NcclUniqueId *);
    typedef int (*init_rank_fn)(void **, int, NcclUniqueId, int);
    typedef int (*comm_fn)(void *);
    typedef int (*void_fn)(void);
    typedef int (*allreduce_fn)(const void *, void *, size_t, int, int, void *, hipStream_t);
    typedef int (*sendrecv_fn)(void *, size_t, int, int, void *, hipStream_t);
    struct Api {
        void *h = nullptr;
        get_id_fn get_id = nullptr;
        init_rank_fn init_rank = nullptr;
        comm_fn destroy = nullptr;
        void_fn gstart = nullptr, gend = nullptr;
        allreduce_fn allreduce = nullptr;
        sendrecv_fn send = nullptr, recv = nullptr;
        std::string why;
        bool load() {
            if (h) return true;
            h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
            if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
            if (!h) { why = std::string("cannot load RCCL: ") + dlerror(); return false; }
            get_id = (get_id_fn)dlsym(h, "ncclGetUniqueId");
            init_rank = (init_rank_fn)dlsym(h, "ncclCommInitRank");
            destroy = (comm_fn)dlsym(h, "ncclCommDestroy");
            gstart = (void_fn)dlsym(h, "ncclGroupStart");
            gend = (void_fn)dlsym(h, "ncclGroupEnd");
            allreduce = (allreduce_fn)dlsym(h, "ncclAllReduce");
            send = (sendrecv_fn)dlsym(h, "ncclSend");
            recv = (sendrecv_fn)dlsym(h, "ncclRecv");
            if (!get_id || !init_rank || !destroy || !gstart || !gend || !allreduce || !send || !recv) { why = "RCCL symbols missing"; h = nullptr; return false; }
            return true;
        }
    };
    static Api &api() { static Api a; return a; }
    void *comm_ = nullptr;

public:
    // rank 0: a fresh id as 256 hex digits ("" + why when RCCL is not there)
    static std::string make_id(std::string &why) {
        Api &a = api();
        if (!a.load()) { why = a.why; return ""; }
        NcclUniqueId id;
        memset(&id, 0, sizeof id);
        if (a.get_id(&id) != 0) { why = "ncclGetUniqueId failed"; return ""; }
        static const char *hx = "0123456789abcdef";
        std::string s;
        for (size_t i = 0; i < sizeof id.internal; ++i) { s += hx[(unsigned char)id.internal[i] >> 4]; s += hx[(unsigned char)id.internal[i] & 15]; }
        return s;
    }
    // a shard: joins the communicator (the device has been selected with hipSetDevice)
    static RcclWire *connect(int rank, int world, const std::string &id_hex, std::string &why) {
        Api &a = api();
        if (!a.load()) { why = a.why; return nullptr; }
        NcclUniqueId id;
        if (id_hex.size() != 2 * sizeof id.internal) { why = "bad communicator id"; return nullptr; }
        auto nib = [](char c) { return c <= '9' ? c - '0' : c - 'a' + 10; };
        for (size_t i = 0; i < sizeof id.internal; ++i) id.internal[i] = (char)((nib(id_hex[2 * i]) << 4) | nib(id_hex[2 * i + 1]));
        void *comm = nullptr;
        if (a.init_rank(&comm, world, id, rank) != 0 || !comm) { why = "ncclCommInitRank failed"; return nullptr; }
        RcclWire *w = new RcclWire();
        w->rank = rank; w->world = world; w->comm_ = comm;
        return w;
    }
    ~RcclWire() override { if (comm_) api().destroy(comm_); }
    const char *name() const override { return "RCCL"; }
    void *nccl_comm() override { return comm_; }
    bool allreduce_u64(uint64_t *d_buf, size_t n, WireOp op) override {
        const int ncclUint64 = 5, ops[3] = {0 /* ncclSum */, 2 /* ncclMax */, 3 /* ncclMin */};
        if (n == 0) return true;
        if (api().allreduce(d_buf, d_buf, n, ncclUint64, ops[op], comm_, nullptr) != 0 || hipStreamSynchronize(nullptr) != hipSuccess) { err = "ncclAllReduce failed"; return false; }
        return true;
    }
    bool alltoallv(const void *d_send, const uint64_t *send_cnt, void *d_recv, const uint64_t *recv_cnt, size_t elem) override {
        const int ncclUint8 = 1;
        Api &a = api();
        bool ok = a.gstart() == 0;
        uint64_t so = 0, ro = 0;
        for (int p = 0; p < world && ok; ++p) {
            if (send_cnt[p]) ok = a.send((void *)((const char *)d_send + so * elem), (size_t)(send_cnt[p] * elem), ncclUint8, p, comm_, nullptr) == 0;
            if (ok && recv_cnt[p]) ok = a.recv((char *)d_recv + ro * elem, (size_t)(recv_cnt[p] * elem), ncclUint8, p, comm_, nullptr) == 0;
            so += send_cnt[p];
            ro += recv_cnt[p];
        }
        ok = (a.gend() == 0) && ok;
        if (!ok || hipStreamSynchronize(nullptr) != hipSuccess) { err = "ncclSend / ncclRecv group failed"; return false; }
        return true;
    }
};

// ------------------------------------------------------------------------------------------------ host wire
// A star through rank 0 over an abstract Unix-domain socket (address = a name the parent made up; no file to clean up).  Every
// message is [uint64 bytes][payload].  Not a fast path: the fallback and the test tier.
class HostWire : public ShardWire {
    std::vector<int> fd_;            // rank 0: fd_[p] = connection of rank p; others: fd_[0] = connection to rank 0
    static bool put(int fd, const void *p, size_t n) {
        const char *c = (const char *)p;
        while (n) {
            const ssize_t k = ::send(fd, c, n, MSG_NOSIGNAL);
            if (k < 0 && errno == EINTR) continue;
            if (k <= 0) return false;
            c += k; n -= (size_t)k;
        }
        return true;
    }
    static bool get(int fd, void *p, size_t n) {
        char *c = (char *)p;
        while (n) {
            const ssize_t k = ::recv(fd, c, n, 0);
            if (k < 0 && errno == EINTR) continue;
            if (k <= 0) return false;
            c += k; n -= (size_t)k;
        }
        return true;
    }
    static bool put_msg(int fd, const void *p, uint64_t n) { return put(fd, &n, 8) && (n == 0 || put(fd, p, (size_t)n)); }
    static bool get_msg(int fd, std::vector<char> &v) {
        uint64_t n = 0;
        if (!get(fd, &n, 8)) return false;
        v.resize((size_t)n);
        return n == 0 || get(fd, v.data(), (size_t)n);
    }
    static socklen_t address(const std::string &name, sockaddr_un &a) {
        memset(&a, 0, sizeof a);
        a.sun_family = AF_UNIX;
        const size_t k = std::min(name.size(), sizeof a.sun_path - 2);
        memcpy(a.sun_path + 1, name.data(), k);              // sun_path[0] == 0: the abstract namespace
        return (socklen_t)(offsetof(sockaddr_un, sun_path) + 1 + k);
    }

public:
    static HostWire *connect(int rank, int world, const std::string &name, std::string &why, int timeout_s = 600) {
        HostWire *w = new HostWire();
        w->rank = rank; w->world = world;
        sockaddr_un a;
        const socklen_t alen = address(name, a);
        if (rank == 0) {
            w->fd_.assign((size_t)world, -1);
            const int ls = socket(AF_UNIX, SOCK_STREAM, 0);
            if (ls < 0 || bind(ls, (sockaddr *)&a, alen) != 0 || listen(ls, world) != 0) { why = std::string("host wire: cannot listen: ") + strerror(errno); if (ls >= 0) close(ls); delete w; return nullptr; }
            for (int k = 1; k < world; ++k) {
                pollfd pf{ls, POLLIN, 0};
                if (poll(&pf, 1, timeout_s * 1000) <= 0) { why = "host wire: a shard did not connect"; close(ls); delete w; return nullptr; }
                const int c = accept(ls, nullptr, nullptr);
                int32_t r = -1;
                if (c < 0 || !get(c, &r, 4) || r < 1 || r >= world || w->fd_[(size_t)r] != -1) { why = "host wire: bad hello"; if (c >= 0) close(c); close(ls); delete w; return nullptr; }
                w->fd_[(size_t)r] = c;
            }
            close(ls);
        } else {
            w->fd_.assign(1, -1);
            const time_t t0 = time(nullptr);
            for (;;) {
                const int c = socket(AF_UNIX, SOCK_STREAM, 0);
                if (c >= 0 && ::connect(c, (sockaddr *)&a, alen) == 0) { w->fd_[0] = c; break; }
                if (c >= 0) close(c);
                if (time(nullptr) - t0 > timeout_s) { why = "host wire: rank 0 is not listening"; delete w; return nullptr; }
                usleep(20000);
            }
            const int32_t r = rank;
            if (!put(w->fd_[0], &r, 4)) { why = "host wire: hello failed"; delete w; return nullptr; }
        }
        return w;
    }
    ~HostWire() override { for (int f : fd_) if (f >= 0) close(f); }
    const char *name() const override { return "host wire"; }

    // host memory: rank 0's n bytes to everybody / the smallest of everybody's numbers to everybody (the bootstrap of the RCCL wire)
    bool bcast_bytes(void *buf, size_t n) {
        if (world == 1) return true;
        if (rank == 0) { for (int p = 1; p < world; ++p) if (!put_msg(fd_[(size_t)p], buf, n)) { err = "host wire: a shard went away"; return false; } return true; }
        std::vector<char> in;
        if (!get_msg(fd_[0], in) || in.size() != n) { err = "host wire: rank 0 went away"; return false; }
        memcpy(buf, in.data(), n);
        return true;
    }
    bool min_of_all(int64_t &v) {
        if (world == 1) return true;
        if (rank == 0) {
            std::vector<char> in;
            for (int p = 1; p < world; ++p) {
                if (!get_msg(fd_[(size_t)p], in) || in.size() != 8) { err = "host wire: a shard went away"; return false; }
                int64_t x;
                memcpy(&x, in.data(), 8);
                v = std::min(v, x);
            }
        } else if (!put_msg(fd_[0], &v, 8)) { err = "host wire: rank 0 went away"; return false; }
        return bcast_bytes(&v, 8);
    }

    bool allreduce_u64(uint64_t *d_buf, size_t n, WireOp op) override {
        if (n == 0 || world == 1) return true;
        std::vector<uint64_t> mine(n);
        if (hipMemcpy(mine.data(), d_buf, n * 8, hipMemcpyDeviceToHost) != hipSuccess) { err = "host wire: copy from the device failed"; return false; }
        if (rank == 0) {
            std::vector<char> in;
            for (int p = 1; p < world; ++p) {
                if (!get_msg(fd_[(size_t)p], in) || in.size() != n * 8) { err = "host wire: a shard went away"; return false; }
                const uint64_t *x = (const uint64_t *)in.data();
                if (op == WIRE_SUM) for (size_t i = 0; i < n; ++i) mine[i] += x[i];
                else if (op == WIRE_MAX) for (size_t i = 0; i < n; ++i) mine[i] = std::max(mine[i], x[i]);
                else for (size_t i = 0; i < n; ++i) mine[i] = std::min(mine[i], x[i]);
            }
            for (int p = 1; p < world; ++p) if (!put_msg(fd_[(size_t)p], mine.data(), n * 8)) { err = "host wire: a shard went away"; return false; }
        } else {
            std::vector<char> in;
            if (!put_msg(fd_[0], mine.data(), n * 8) || !get_msg(fd_[0], in) || in.size() != n * 8) { err = "host wire: rank 0 went away"; return false; }
            memcpy(mine.data(), in.data(), n * 8);
        }
        if (hipMemcpy(d_buf, mine.data(), n * 8, hipMemcpyHostToDevice) != hipSuccess) { err = "host wire: copy to the device failed"; return false; }
        return true;
    }

    // Through rank 0, piece by piece (rank 0 used to hold every rank's whole payload plus one message per destination --
    // tens of GB for the duplicate exchange of a run at the reference's 2^32-pair limit).  A peer first sends its counts, then its
    // payload (already ordered by destination) from a THREAD of its own while its main thread receives; rank 0 walks the
    // destinations in order and, for each, the sources in order: the next cnt[src][dst] * elem bytes of src's stream go straight on to
    // dst in slabs of at most 4 MiB (its own pieces from / to its own buffers).  Rank 0 holds one slab, a peer its own two buffers.
    // Why the thread: rank 0 reads a peer's stream in destination-major order, so a peer is still sending its pieces for later
    // destinations while rank 0 already forwards other ranks' pieces TO it -- with a blocking send in its main thread both would wait.
    bool alltoallv(const void *d_send, const uint64_t *send_cnt, void *d_recv, const uint64_t *recv_cnt, size_t elem) override {
        uint64_t ns = 0, nr = 0;
        for (int p = 0; p < world; ++p) { ns += send_cnt[p]; nr += recv_cnt[p]; }
        std::vector<char> out((size_t)(ns * elem)), in((size_t)(nr * elem));
        if (ns && hipMemcpy(out.data(), d_send, out.size(), hipMemcpyDeviceToHost) != hipSuccess) { err = "host wire: copy from the device failed"; return false; }
        if (world == 1) in = out;
        else if (rank == 0) {
            std::vector<std::vector<uint64_t>> cnt((size_t)world, std::vector<uint64_t>((size_t)world, 0));
            cnt[0].assign(send_cnt, send_cnt + world);
            for (int p = 1; p < world; ++p) {
                std::vector<char> c;
                uint64_t total = 0, said = 0;
                if (!get_msg(fd_[(size_t)p], c) || c.size() != (size_t)world * 8 || !get(fd_[(size_t)p], &said, 8)) { err = "host wire: a shard went away"; return false; }
                memcpy(cnt[(size_t)p].data(), c.data(), c.size());
                for (int q = 0; q < world; ++q) total += cnt[(size_t)p][(size_t)q];
                if (said != total * elem) { err = "host wire: a shard's payload does not match its counts"; return false; }   // (the header of its payload message)
            }
            std::vector<char> slab((size_t)4 << 20);
            uint64_t own_off = 0;                            // rank 0's own pieces sit in `out` in destination order
            for (int dst = 0; dst < world; ++dst) {
                uint64_t total = 0;
                for (int src = 0; src < world; ++src) total += cnt[(size_t)src][(size_t)dst];
                if (dst == 0 ? total * elem != in.size() : false) { err = "host wire: counts do not match"; return false; }
                if (dst != 0) { const uint64_t bytes = total * elem; if (!put(fd_[(size_t)dst], &bytes, 8)) { err = "host wire: a shard went away"; return false; } }
                uint64_t in_off = 0;
                for (int src = 0; src < world; ++src) {
                    uint64_t left = cnt[(size_t)src][(size_t)dst] * elem;
                    if (src == 0) {
                        if (dst == 0) memcpy(in.data() + in_off, out.data() + own_off, (size_t)left);
                        else if (left && !put(fd_[(size_t)dst], out.data() + own_off, (size_t)left)) { err = "host wire: a shard went away"; return false; }
                        own_off += left;
                        in_off += left;
                        continue;
                    }
                    while (left) {
                        const size_t k = (size_t)std::min<uint64_t>(left, slab.size());
                        char *to = dst == 0 ? in.data() + in_off : slab.data();
                        if (!get(fd_[(size_t)src], to, k)) { err = "host wire: a shard went away"; return false; }
                        if (dst != 0 && !put(fd_[(size_t)dst], slab.data(), k)) { err = "host wire: a shard went away"; return false; }
                        left -= k;
                        in_off += k;
                    }
                }
            }
        } else {
            bool sent = false;
            std::thread sender([&] { sent = put_msg(fd_[0], send_cnt, (uint64_t)world * 8) && put_msg(fd_[0], out.data(), out.size()); });
            const bool got = get_msg(fd_[0], in);
            sender.join();
            if (!sent || !got || in.size() != (size_t)(nr * elem)) { err = "host wire: rank 0 went away or the counts do not match"; return false; }
        }
        if (nr && hipMemcpy(d_recv, in.data(), in.size(), hipMemcpyHostToDevice) != hipSuccess) { err = "host wire: copy to the device failed"; return false; }
        return true;
    }
};

}  // namespace snk
#endif
