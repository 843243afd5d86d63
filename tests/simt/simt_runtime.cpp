// SIMT emulator runtime (see hip/hip_runtime.h): fibers, the workgroup scheduler, the worker pool and the host API stand-ins.
// Test infrastructure only.
#include <unistd.h>
#include <hip/hip_runtime.h>

#include <dlfcn.h>
#include <sys/mman.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <map>
#include <string>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

// ---- context switch (x86-64 System V): callee-saved registers on the old stack, stack pointers exchanged
extern "C" void simt_switch(void **save_sp, void *load_sp);
asm(R"(
    .text
    .globl simt_switch
    .type simt_switch, @function
simt_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
    .size simt_switch, .-simt_switch
)");

// dynamic shared memory (HIP_DYNAMIC_SHARED): one 160 KB buffer per OS thread, the LDS of a gfx950 CU.  It is also what the
// absolute LDS addresses of simt_gfx950.h are relative to.
alignas(256) thread_local uint8_t simt_dyn_lds[160 * 1024];
uint8_t *simt_dyn_shared() { return simt_dyn_lds; }

// AddressSanitizer builds (tests/test_simt_sanitized.py): the runtime is told about every stack switch
#if defined(__has_feature)
#if __has_feature(address_sanitizer)
#define SIMT_ASAN 1
extern "C" void __sanitizer_start_switch_fiber(void **fake_stack_save, const void *bottom, size_t size);
extern "C" void __sanitizer_finish_switch_fiber(void *fake_stack_save, const void **bottom_old, size_t *size_old);
#endif
#endif

namespace simt {
thread_local Fiber *cur = nullptr;

namespace {
constexpr size_t STACK = 256 * 1024;
constexpr int MAXT = 1024;

struct Worker {
    char *stacks = nullptr;                    // MAXT stacks, touched on demand
    void *sched_sp = nullptr;
    Fiber fibers[MAXT];
    Wave waves[MAXT / 64];
    Block blk;
    const std::function<void()> *body = nullptr;
    const void *sched_bottom = nullptr;        // (AddressSanitizer: the scheduler's own stack)
    size_t sched_size = 0;
    void *sched_fake = nullptr;
    Worker() {
        stacks = (char *)mmap(nullptr, STACK * MAXT, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
        if (stacks == MAP_FAILED) { perror("simt: mmap of the fiber stacks"); abort(); }
    }
    ~Worker() { munmap(stacks, STACK * MAXT); }
};
thread_local std::unique_ptr<Worker> tls_worker;
thread_local Worker *W = nullptr;

void fiber_exit(Fiber *f) {
    f->done = true;
    Wave &w = *f->wave;
    Block &b = *f->blk;
    w.live_mask &= ~(1ull << f->lane);
    if (--w.live > 0 && w.arrived == w.live && w.uniform) wave_release(w, w.pend_mask);       // the others were waiting for this lane only
    if (--b.live > 0 && b.bar_arrived == b.live) b.release();
}

// nothing of the workgroup can run: complete, for one group of lanes per wave, the wave operation they wait in (see wave_exchange).
// The group: the lanes deepest in calls (largest stack depth), of those the ones at the lowest code address.
bool release_divergent(Worker &wk, int nwaves) {
    bool any = false;
    for (int wi = 0; wi < nwaves; ++wi) {
        Wave &w = wk.waves[wi];
        if (!w.pend_mask) continue;
        int best = -1;
        for (uint64_t t = w.pend_mask; t; t &= t - 1) {
            const int l = __builtin_ctzll(t);
            if (best < 0 || w.site[l][2] > w.site[best][2] || (w.site[l][2] == w.site[best][2] && w.site[l][0] < w.site[best][0])) best = l;
        }
        uint64_t group = 0;
        for (uint64_t t = w.pend_mask; t; t &= t - 1) {
            const int l = __builtin_ctzll(t);
            if (w.site[l][0] == w.site[best][0] && w.site[l][1] == w.site[best][1] && w.site[l][2] == w.site[best][2]) group |= 1ull << l;
        }
        if (getenv("SIMT_TRACE_DIVERGENCE"))
        {
            fprintf(stderr, "simt: wave %d: %s completes for lanes %016llx of %016llx", wi, w.kind[best], (unsigned long long)group, (unsigned long long)w.live_mask);
            Dl_info di;
            if (dladdr((void *)w.site[best][0], &di) && di.dli_fbase) {
                fprintf(stderr, "; waiting at (site+0x, caller+0x, depth):");
                uint64_t seen[8][3];
                int ns = 0;
                for (uint64_t t = w.pend_mask; t; t &= t - 1) {
                    const int l = __builtin_ctzll(t);
                    bool known = false;
                    for (int k = 0; k < ns; ++k) known = known || (seen[k][0] == w.site[l][0] && seen[k][1] == w.site[l][1] && seen[k][2] == w.site[l][2]);
                    if (known || ns == 8) continue;
                    seen[ns][0] = w.site[l][0]; seen[ns][1] = w.site[l][1]; seen[ns][2] = w.site[l][2];
                    ++ns;
                    fprintf(stderr, " (%llx, %llx, %llu)", (unsigned long long)(w.site[l][0] - (uint64_t)di.dli_fbase), (unsigned long long)(w.site[l][1] - (uint64_t)di.dli_fbase), (unsigned long long)w.site[l][2]);
                }
            }
            fprintf(stderr, "\n");
        }
        wave_release(w, group);
        any = true;
    }
    return any;
}

extern "C" void simt_fiber_main() {
    Fiber *f = cur;
#ifdef SIMT_ASAN
    __sanitizer_finish_switch_fiber(nullptr, &W->sched_bottom, &W->sched_size);
#endif
    (*W->body)();
    fiber_exit(f);
#ifdef SIMT_ASAN
    __sanitizer_start_switch_fiber(nullptr, W->sched_bottom, W->sched_size);      // (nullptr: this fiber's fake stack is given up)
#endif
    for (;;) simt_switch(&f->sp, W->sched_sp);       // never resumed
}

void run_block(Worker &wk, const std::function<void()> &body, dim3 grid, dim3 block, size_t shmem, uint64_t bi) {
    const int n = (int)(block.x * block.y * block.z);
    if (n > MAXT || n <= 0) { fprintf(stderr, "simt: workgroup of %d threads\n", n); abort(); }
    wk.body = &body;
    Block &b = wk.blk;
    b = Block();
    b.gdim = grid;
    b.bdim = block;
    b.bid = dim3((unsigned)(bi % grid.x), (unsigned)((bi / grid.x) % grid.y), (unsigned)(bi / ((uint64_t)grid.x * grid.y)));
    b.live = n;
    b.dyn_shared = shmem;
    const int nw = (n + 63) / 64;
    for (int w = 0; w < nw; ++w) {
        Wave &wv = wk.waves[w];
        wv = Wave();
        wv.live = std::min(64, n - 64 * w);
        wv.live_mask = wv.live == 64 ? ~0ull : ((1ull << wv.live) - 1);
        wv.first = &wk.fibers[64 * w];
    }
    for (int i = 0; i < n; ++i) {
        Fiber &f = wk.fibers[i];
        f = Fiber();
        f.index = i;
        f.lane = i & 63;
        f.wave = &wk.waves[i >> 6];
        f.blk = &b;
        f.tid = dim3((unsigned)(i % block.x), (unsigned)((i / block.x) % block.y), (unsigned)(i / (block.x * block.y)));
        uint64_t *top = (uint64_t *)(wk.stacks + STACK * (size_t)(i + 1));
        top[-1] = 0;                                   // (return address slot of the entry function: never used)
        top[-2] = (uint64_t)(uintptr_t)&simt_fiber_main;
        for (int k = 3; k <= 8; ++k) top[-k] = 0;      // rbp rbx r12..r15
        f.sp = top - 8;
        f.stack_top = (uint64_t)(uintptr_t)top;
    }
    int left = n;
    while (left > 0) {
        bool ran = false;
        for (int i = 0; i < n; ++i) {
            Fiber &f = wk.fibers[i];
            if (f.done) continue;
            if (f.wait_ptr) {
                if (*f.wait_ptr == f.wait_val) continue;
                f.wait_ptr = nullptr;
            }
            ran = true;
            cur = &f;
#ifdef SIMT_ASAN
            __sanitizer_start_switch_fiber(&wk.sched_fake, wk.stacks + STACK * (size_t)i, STACK);
#endif
            simt_switch(&wk.sched_sp, f.sp);
#ifdef SIMT_ASAN
            __sanitizer_finish_switch_fiber(wk.sched_fake, nullptr, nullptr);
#endif
            if (f.done) --left;
        }
        if (!ran && release_divergent(wk, nw)) continue;
        if (!ran) {
            fprintf(stderr, "simt: workgroup (%u,%u,%u) cannot make progress -- a wave-level operation or barrier inside divergent control flow?\n", b.bid.x, b.bid.y, b.bid.z);
            const char *seen[16];
            int nseen = 0;
            for (int i = 0; i < n; ++i) {
                Fiber &f = wk.fibers[i];
                if (f.done) continue;
                fprintf(stderr, "  thread %d (wave %d lane %d) waits in %s\n", i, i >> 6, f.lane, f.where ? f.where : "?");
                bool known = false;
                for (int k = 0; k < nseen; ++k) known = known || seen[k] == f.where;
                if (known || nseen == 16) continue;
                seen[nseen++] = f.where;
                // the fiber's call chain (frame pointers; `addr2line -f -C -i -e <library> <offset>` names the lines)
                const uint64_t *sp = (const uint64_t *)f.sp;
                const char *lo = wk.stacks + STACK * (size_t)i, *hi = lo + STACK;
                uint64_t rbp = sp[5], ret = sp[6];
                for (int depth = 0; depth < 12 && ret; ++depth) {
                    Dl_info di;
                    if (dladdr((void *)ret, &di) && di.dli_fname) fprintf(stderr, "      %s+0x%llx\n", di.dli_fname, (unsigned long long)(ret - (uint64_t)di.dli_fbase));
                    if ((const char *)rbp < lo || (const char *)rbp + 16 > hi) break;
                    ret = ((const uint64_t *)rbp)[1];
                    rbp = ((const uint64_t *)rbp)[0];
                }
            }
            abort();
        }
    }
    cur = nullptr;
}

struct Job {
    std::function<void()> body;
    dim3 grid, block;
    size_t shmem = 0;
    uint64_t total = 0;
    std::atomic<uint64_t> next{0}, done{0};
};

struct Pool {
    std::mutex m;
    std::condition_variable cv, cv_done;
    std::deque<std::shared_ptr<Job>> jobs;
    std::vector<std::thread> threads;
    bool quit = false;
    Pool() {
        int n = (int)std::thread::hardware_concurrency();
        if (const char *e = getenv("SIMT_THREADS")) n = atoi(e);
        n = std::max(1, std::min(n, 64)) - 1;          // the launching thread works too
        for (int i = 0; i < n; ++i) threads.emplace_back([this] { work(); });
    }
    ~Pool() {
        { std::lock_guard<std::mutex> l(m); quit = true; }
        cv.notify_all();
        for (auto &t : threads) t.join();
    }
    static void run_some(const std::shared_ptr<Job> &j) {
        if (!tls_worker) tls_worker.reset(new Worker());
        W = tls_worker.get();
        for (;;) {
            const uint64_t bi = j->next.fetch_add(1);
            if (bi >= j->total) break;
            run_block(*W, j->body, j->grid, j->block, j->shmem, bi);
            j->done.fetch_add(1);
        }
    }
    void work() {
        for (;;) {
            std::shared_ptr<Job> j;
            {
                std::unique_lock<std::mutex> l(m);
                cv.wait(l, [&] {
                    while (!jobs.empty() && jobs.front()->next.load() >= jobs.front()->total) jobs.pop_front();
                    return quit || !jobs.empty();
                });
                if (quit) return;
                j = jobs.front();
            }
            run_some(j);
            { std::lock_guard<std::mutex> l(m); }
            cv_done.notify_all();
        }
    }
    void launch(const std::shared_ptr<Job> &j) {
        if (j->total > 1 && !threads.empty()) {
            { std::lock_guard<std::mutex> l(m); jobs.push_back(j); }
            cv.notify_all();
        }
        run_some(j);
        std::unique_lock<std::mutex> l(m);
        cv_done.wait(l, [&] { return j->done.load() >= j->total; });
    }
};
Pool &pool() {
    static Pool *p = new Pool();       // (leaked on purpose: worker threads must not be joined from a static destructor of a dlopen'ed library)
    return *p;
}
}  // namespace

void wave_release(Wave &w, uint64_t group) {
    const int idx = (int)(w.seq++ & 1);
    const char *kind = nullptr;
    for (uint64_t t = group; t; t &= t - 1) {
        const int l = __builtin_ctzll(t);
        w.snap[idx][l] = w.val[l];
        Fiber &f = w.first[l];
        f.snap_idx = idx;
        ++f.wake;
        if (!kind) kind = w.kind[l];
        else if (kind != w.kind[l]) { fprintf(stderr, "simt: lanes at one place in different wave operations (%s, %s)\n", kind, w.kind[l]); abort(); }
    }
    w.snap_mask[idx] = group;
    w.pend_mask &= ~group;
    w.arrived -= __builtin_popcountll(group);
    // what is still waiting: uniform again?
    w.uniform = true;
    bool have = false;
    for (uint64_t t = w.pend_mask; t; t &= t - 1) {
        const int l = __builtin_ctzll(t);
        if (!have) { have = true; w.site0[0] = w.site[l][0]; w.site0[1] = w.site[l][1]; w.site0[2] = w.site[l][2]; }
        else if (w.site0[0] != w.site[l][0] || w.site0[1] != w.site[l][1] || w.site0[2] != w.site[l][2]) w.uniform = false;
    }
}

__attribute__((noinline, convergent)) const uint64_t *wave_exchange(uint64_t mine, uint64_t &mask, const char *where) {
    Fiber *f = cur;
    Wave &w = *f->wave;
    const int l = f->lane;
    const uint64_t site[3] = {(uint64_t)__builtin_return_address(0), (uint64_t)__builtin_return_address(1),
                              f->stack_top - (uint64_t)__builtin_frame_address(0)};
    w.val[l] = mine;
    w.kind[l] = where;
    w.site[l][0] = site[0]; w.site[l][1] = site[1]; w.site[l][2] = site[2];
    if (w.arrived == 0) {
        w.uniform = true;
        w.site0[0] = site[0]; w.site0[1] = site[1]; w.site0[2] = site[2];
    } else if (w.site0[0] != site[0] || w.site0[1] != site[1] || w.site0[2] != site[2]) {
        w.uniform = false;
    }
    w.pend_mask |= 1ull << l;
    const uint64_t woke = f->wake;
    if (++w.arrived == w.live && w.uniform) wave_release(w, w.pend_mask);
    if (f->wake == woke) {
        f->wait_ptr = &f->wake;
        f->wait_val = woke;
        f->where = where;
        yield_to_scheduler();
    }
    mask = w.snap_mask[f->snap_idx];
    return w.snap[f->snap_idx];
}

void yield_to_scheduler() {
    Fiber *f = cur;
#ifdef SIMT_ASAN
    void *fake = nullptr;
    __sanitizer_start_switch_fiber(&fake, W->sched_bottom, W->sched_size);
#endif
    simt_switch(&f->sp, W->sched_sp);
#ifdef SIMT_ASAN
    __sanitizer_finish_switch_fiber(fake, &W->sched_bottom, &W->sched_size);
#endif
}

void launch(dim3 grid, dim3 block, size_t shmem, const std::function<void()> &body) {
    if (cur) { fprintf(stderr, "simt: kernel launch from inside a kernel\n"); abort(); }
    auto j = std::make_shared<Job>();
    j->body = body;
    j->grid = grid;
    j->block = block;
    j->shmem = shmem;
    j->total = (uint64_t)grid.x * grid.y * grid.z;
    if (j->total == 0) return;
    if (shmem > 160 * 1024) { fprintf(stderr, "simt: %zu bytes of dynamic LDS\n", shmem); abort(); }
    pool().launch(j);
}
}  // namespace simt

// ---------------------------------------------------------------------------------------------------------------- host API
namespace {
std::atomic<size_t> g_allocated{0};
size_t total_mem() {
    if (const char *e = getenv("SIMT_DEVICE_MB")) return (size_t)atol(e) << 20;
    return (size_t)32 << 30;
}
struct Hdr { size_t n; size_t magic; };
constexpr size_t HDR = 256, MAGIC = 0x53494D54414C4C4Full;
// every live allocation a kernel may touch (device and pinned host memory): what SIMT_DUMP_DIR snapshots
std::mutex g_reg_mu;
std::map<uintptr_t, size_t> g_reg;
void reg_add(void *p, size_t n) { std::lock_guard<std::mutex> l(g_reg_mu); g_reg[(uintptr_t)p] = n; }
void reg_del(void *p) { std::lock_guard<std::mutex> l(g_reg_mu); g_reg.erase((uintptr_t)p); }
}  // namespace

namespace simt {
bool dump_wanted() {
    static const bool on = getenv("SIMT_DUMP_DIR") != nullptr;
    return on;
}
std::mutex &dump_serial() { static std::mutex m; return m; }
namespace {
std::atomic<int> g_dump_seq{0};
std::vector<std::pair<uintptr_t, size_t>> dump_allocs() {
    const size_t cap = getenv("SIMT_DUMP_MAX_ALLOC") ? (size_t)atol(getenv("SIMT_DUMP_MAX_ALLOC")) : ((size_t)256 << 20);
    std::lock_guard<std::mutex> l(g_reg_mu);
    std::vector<std::pair<uintptr_t, size_t>> v;
    for (auto &kv : g_reg) if (kv.second && kv.second <= cap) v.push_back(kv);
    return v;
}
// (under g_reg_mu: another host thread cannot free an allocation while it is being written; one that went away between the two
// snapshots of a launch -- or came back with another size -- is written as zeros and named in L<k>.gone: the replay skips it)
void dump_mem(const std::string &path, const std::vector<std::pair<uintptr_t, size_t>> &al, std::vector<size_t> *gone = nullptr) {
    // the file: "SNKDUMP1", u64 total bytes, then {u64 offset into the concatenated allocations, u64 n, n bytes} for every 4 KiB piece
    // that is not all 0xEE (what hipMalloc leaves behind: most of a run's buffers most of the time; n with bit 63 set: one byte
    // follows, the piece holds it n times); the reader starts from 0xEE
    FILE *f = fopen(path.c_str(), "wb");
    if (!f) { perror(path.c_str()); abort(); }
    std::lock_guard<std::mutex> l(g_reg_mu);
    static const std::vector<unsigned char> fill(4096, 0xEE);
    uint64_t total = 0, off = 0;
    for (auto &a : al) total += a.second;
    fwrite("SNKDUMP1", 1, 8, f);
    fwrite(&total, 8, 1, f);
    for (size_t i = 0; i < al.size(); off += al[i].second, ++i) {
        auto it = g_reg.find(al[i].first);
        if (it == g_reg.end() || it->second != al[i].second) {
            if (gone) gone->push_back(i);
            continue;
        }
        const unsigned char *p = (const unsigned char *)al[i].first;
        for (size_t o = 0; o < al[i].second; o += 4096) {
            const uint64_t n = std::min<size_t>(4096, al[i].second - o), at = off + o;
            if (memcmp(p + o, fill.data(), n) == 0) continue;
            fwrite(&at, 8, 1, f);
            if (n > 1 && memcmp(p + o, p + o + 1, n - 1) == 0) {          // one byte value all over (cleared tables): {offset, n | 1 << 63, the byte}
                const uint64_t tagged = n | (1ull << 63);
                fwrite(&tagged, 8, 1, f);
                fwrite(p + o, 1, 1, f);
                continue;
            }
            fwrite(&n, 8, 1, f);
            fwrite(p + o, 1, n, f);
        }
    }
    fclose(f);
}
std::mutex g_dump_mu;
std::map<int, std::vector<std::pair<uintptr_t, size_t>>> g_dump_al;
}  // namespace
int dump_pre(const void *kernel, dim3 grid, dim3 block, size_t shmem, const void *kernarg, size_t kernarg_bytes) {
    Dl_info di;
    if (!dladdr(kernel, &di) || !di.dli_fbase) return -1;
    const uintptr_t off = (uintptr_t)kernel - (uintptr_t)di.dli_fbase;
    if (const char *e = getenv("SIMT_DUMP_OFFSETS")) {          // only these kernels (offsets into the library, hex, comma-separated)
        bool hit = false;
        for (const char *q = e; *q;) {
            char *end = nullptr;
            const uintptr_t v = (uintptr_t)strtoull(q, &end, 16);
            if (end == q) break;
            hit |= v == off;
            q = *end ? end + 1 : end;
        }
        if (!hit) return -1;
    }
    if (const char *e = getenv("SIMT_DUMP_PER_KERNEL")) {      // at most this many launches of one kernel (a whole CLI run launches hundreds)
        static std::mutex mu;
        static std::map<uintptr_t, int> seen;
        std::lock_guard<std::mutex> l(mu);
        if (seen[off]++ >= atoi(e)) return -1;
    }
    // SIMT_DUMP_BY_PID=1: a run that spawns processes (the shards of a sharded run) numbers each process's launches in a range of its own
    static const long pid_base = getenv("SIMT_DUMP_BY_PID") ? (long)(getpid() % 20000) * 10000 : 0;
    const int id = (int)(pid_base + g_dump_seq.fetch_add(1));
    const std::string base = std::string(getenv("SIMT_DUMP_DIR")) + "/L" + std::to_string(id);
    auto al = dump_allocs();
    dump_mem(base + ".pre", al);
    FILE *f = fopen((base + ".json").c_str(), "w");
    if (!f) { perror(base.c_str()); abort(); }
    fprintf(f, "{\"lib\": \"%s\", \"base\": %llu, \"offset\": %llu, \"grid\": [%u, %u, %u], \"block\": [%u, %u, %u], \"shmem\": %zu, \"kernarg\": \"",
            di.dli_fname ? di.dli_fname : "", (unsigned long long)(uintptr_t)di.dli_fbase, (unsigned long long)off, grid.x, grid.y, grid.z, block.x, block.y, block.z, shmem);
    for (size_t i = 0; i < kernarg_bytes; ++i) fprintf(f, "%02x", ((const unsigned char *)kernarg)[i]);
    fprintf(f, "\", \"allocs\": [");
    for (size_t i = 0; i < al.size(); ++i) fprintf(f, "%s[%llu, %zu]", i ? ", " : "", (unsigned long long)al[i].first, al[i].second);
    fprintf(f, "]}\n");
    fclose(f);
    std::lock_guard<std::mutex> l(g_dump_mu);
    g_dump_al[id] = std::move(al);
    return id;
}
}  // namespace simt
// memory the caller owns (numpy arrays handed in as "device" pointers) joins / leaves the snapshots
extern "C" void simt_dump_register(void *p, size_t n) { if (simt::dump_wanted()) reg_add(p, n); }
extern "C" void simt_dump_unregister(void *p) { if (simt::dump_wanted()) reg_del(p); }
namespace simt {
void dump_post(int id) {
    std::vector<std::pair<uintptr_t, size_t>> al;
    {
        std::lock_guard<std::mutex> l(g_dump_mu);
        al = std::move(g_dump_al[id]);
        g_dump_al.erase(id);
    }
    std::vector<size_t> gone;
    const std::string base = std::string(getenv("SIMT_DUMP_DIR")) + "/L" + std::to_string(id);
    dump_mem(base + ".post", al, &gone);
    if (!gone.empty()) {
        FILE *f = fopen((base + ".gone").c_str(), "w");
        if (f) { for (size_t i : gone) fprintf(f, "%zu\n", i); fclose(f); }
    }
}
}  // namespace simt

struct simt_stream { int id; };
struct simt_event { std::chrono::steady_clock::time_point t; bool recorded = false; };

hipError_t hipMalloc(void **p, size_t n) {
    if (!p) return hipErrorInvalidValue;
    if (g_allocated.load() + n > total_mem()) { *p = nullptr; return hipErrorOutOfMemory; }
    char *raw = nullptr;
    if (posix_memalign((void **)&raw, 256, HDR + n + 64) != 0) { *p = nullptr; return hipErrorOutOfMemory; }
    ((Hdr *)raw)->n = n;
    ((Hdr *)raw)->magic = MAGIC;
    memset(raw + HDR, 0xEE, n + 64);
    g_allocated.fetch_add(n);
    *p = raw + HDR;
    if (simt::dump_wanted()) reg_add(*p, n + 64);
    return hipSuccess;
}
hipError_t hipFree(void *p) {
    if (!p) return hipSuccess;
    char *raw = (char *)p - HDR;
    if (((Hdr *)raw)->magic != MAGIC) { fprintf(stderr, "simt: hipFree of a pointer hipMalloc did not return\n"); abort(); }
    ((Hdr *)raw)->magic = 0;
    if (simt::dump_wanted()) reg_del(p);
    g_allocated.fetch_sub(((Hdr *)raw)->n);
    free(raw);
    return hipSuccess;
}
hipError_t hipMallocAsync(void **p, size_t n, hipStream_t) { return hipMalloc(p, n); }
hipError_t hipFreeAsync(void *p, hipStream_t) { return hipFree(p); }
hipError_t hipHostMalloc(void **p, size_t n, unsigned) {
    void *q = nullptr;
    if (posix_memalign(&q, 256, n ? n : 1) != 0) { *p = nullptr; return hipErrorOutOfMemory; }
    *p = q;
    if (simt::dump_wanted()) reg_add(q, n);
    return hipSuccess;
}
hipError_t hipHostFree(void *p) { if (simt::dump_wanted()) reg_del(p); free(p); return hipSuccess; }
hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind) {
    std::unique_lock<std::mutex> serial;
    if (simt::dump_wanted()) serial = std::unique_lock<std::mutex>(simt::dump_serial());
    if (n) memmove(d, s, n);
    return hipSuccess;
}
hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind k, hipStream_t) { return hipMemcpy(d, s, n, k); }
hipError_t hipMemcpyPeerAsync(void *d, int, const void *s, int, size_t n, hipStream_t) { return hipMemcpy(d, s, n, hipMemcpyDeviceToDevice); }
hipError_t hipMemset(void *d, int v, size_t n) {
    std::unique_lock<std::mutex> serial;
    if (simt::dump_wanted()) serial = std::unique_lock<std::mutex>(simt::dump_serial());
    if (n) memset(d, v, n);
    return hipSuccess;
}
hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t) { return hipMemset(d, v, n); }
hipError_t hipMemGetInfo(size_t *free_b, size_t *total_b) {
    const size_t t = total_mem(), a = g_allocated.load();
    if (total_b) *total_b = t;
    if (free_b) *free_b = a < t ? t - a : 0;
    return hipSuccess;
}
hipError_t hipSetDevice(int d) { return d >= 0 && d < 8 ? hipSuccess : hipErrorInvalidValue; }
hipError_t hipGetDevice(int *d) { *d = 0; return hipSuccess; }
hipError_t hipGetDeviceCount(int *n) { *n = getenv("SIMT_DEVICES") ? atoi(getenv("SIMT_DEVICES")) : 1; return hipSuccess; }
hipError_t hipGetDeviceProperties(hipDeviceProp_t *p, int) {
    memset(p, 0, sizeof *p);
    snprintf(p->name, sizeof p->name, "SIMT emulator (CPU)");
    snprintf(p->gcnArchName, sizeof p->gcnArchName, "gfx950:emulated");
    p->totalGlobalMem = total_mem();
    p->multiProcessorCount = getenv("SIMT_CUS") ? atoi(getenv("SIMT_CUS")) : 8;
    p->warpSize = 64;
    p->maxThreadsPerBlock = 1024;
    p->clockRate = 2400000;
    p->memoryClockRate = 2000000;
    p->memoryBusWidth = 8192;
    p->sharedMemPerBlock = 160 * 1024;
    p->maxSharedMemoryPerMultiProcessor = 160 * 1024;
    p->major = 9;
    p->minor = 5;
    return hipSuccess;
}
hipError_t hipDeviceSynchronize() { return hipSuccess; }
hipError_t hipGetLastError() { return hipSuccess; }
hipError_t hipPeekAtLastError() { return hipSuccess; }
const char *hipGetErrorString(hipError_t e) { return e == hipSuccess ? "no error" : e == hipErrorOutOfMemory ? "out of memory" : "error (SIMT emulator)"; }
hipError_t hipStreamCreate(hipStream_t *s) { *s = new simt_stream{1}; return hipSuccess; }
hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { return hipStreamCreate(s); }
hipError_t hipStreamDestroy(hipStream_t s) { delete s; return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
hipError_t hipEventCreate(hipEvent_t *e) { *e = new simt_event(); return hipSuccess; }
hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { return hipEventCreate(e); }
hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t e, hipStream_t) { e->t = std::chrono::steady_clock::now(); e->recorded = true; return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventQuery(hipEvent_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b) {
    *ms = std::chrono::duration<float, std::milli>(b->t - a->t).count();
    return hipSuccess;
}
