// snk_dgunzip.h -- one gzip stream decoded by the GPU (include/snk_gunzip.h), window by window, for the FASTQ readers.
//
// Same streaming interface as ParallelGunzip (snk_pgunzip.h: run(out, cap) / done() / error()), other engine: the host only cuts
// the compressed file into windows, checks that the chunks of a window CHAIN (every chunk ends exactly where the next one starts;
// block starts the chain steps over are false positives), verifies CRC-32 and ISIZE of every member on the resolved text, and
// falls back to the sequential host decoder (snk_inflate.h) from the last good block on for whatever the device path refuses --
// a chunk that overflowed its symbol slots, invalid data, a window without a single complete chunk -- for ONE SPELL: the host
// decoder stops in front of the first block header a chunk's length further on and the device windows resume there (a spell
// that the device refuses again right away is twice as long; ADVICE r4: one poly-N region used to put the rest of a 100 GB file
// on one core).  The bytes are always zlib's bytes or an error.  The device calls sit behind DgBackend so that tests/host_emul/ can run this very class on the CPU
// with the same chunk decoder (csrc/snk_inflate_core.hip.h) -- tests/test_inflate_emul.py.
// Reference: the gzgets() reading loop, src/peprocess.cpp:2063-2113.
#ifndef SNK_DGUNZIP_H
#define SNK_DGUNZIP_H
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>
#include <zlib.h>

#include "../../include/snk_gunzip.h"
#include "snk_inflate.h"

namespace snk {

struct DgBackend {
    virtual ~DgBackend() {}
    virtual bool decode(const uint8_t *comp, uint64_t nbytes, uint64_t first_bit, bool first_of_member, snk_gunzip_chunk *chunks, snk_gunzip_member *ends) = 0;
    virtual bool resolve(const uint32_t *order, uint32_t k, const uint8_t *win_in, uint8_t *text, uint64_t text_bytes, uint8_t *win_out) = 0;
    virtual uint8_t *text_buffer(int slot, size_t bytes) = 0;   // slot 0 / 1: at least `bytes` (pinned memory on the device backend, grown on demand); owned by the backend
    virtual std::string error() = 0;
};

class DeviceGunzip {
public:
    enum { HIST = 32768 };
    struct Geometry { uint64_t window_bytes; uint32_t chunk_bytes, syms_per_chunk, ends_per_chunk; };
    // start_bit / start_win: from the deflate block header at that bit offset on, the 32 KiB of text in front of it given (a shard
    // of a sharded run; the member the start lies in is not checked again -- the parent's scout pass has, host/snk_main.cpp)
    DeviceGunzip(const uint8_t *in, size_t n, DgBackend *be, const Geometry &g, int crc_threads, uint64_t start_bit = ~0ull, const uint8_t *start_win = nullptr)
        : in_(in), n_(n), be_(be), g_(g), crc_threads_(crc_threads < 1 ? 1 : crc_threads) {
        win_.assign(HIST, 0);
        const uint32_t maxc = (uint32_t)((g_.window_bytes + g_.chunk_bytes - 1) / g_.chunk_bytes);
        chunks_.resize(maxc);
        ends_.resize((size_t)maxc * g_.ends_per_chunk);
        if (start_bit != ~0ull && start_win && (start_bit >> 3) < n_) {
            pos_bit_ = start_bit;
            memcpy(win_.data(), start_win, HIST);
            first_of_member_ = false;
            member_checkable_ = false;
            mid_start_ = true;
            producer_ = std::thread([this] { produce(); });
            return;
        }
        // the first member's header (the later ones are followed on the device)
        if (!member_header_at(0, pos_bit_)) { seq_from_start(); return; }
        first_of_member_ = true;
        producer_ = std::thread([this] { produce(); });
    }
    ~DeviceGunzip() {
        { std::lock_guard<std::mutex> l(m_); quit_ = true; }
        cv_.notify_all();
        if (producer_.joinable()) producer_.join();
    }
    DeviceGunzip(const DeviceGunzip &) = delete;
    DeviceGunzip &operator=(const DeviceGunzip &) = delete;
    const char *error() const { return err_.empty() ? nullptr : err_.c_str(); }
    bool done() const { return done_; }
    uint64_t windows() const { return windows_; }
    uint64_t fallback_bit() const { return fallback_bit_; }            // ~0: the device decoded everything; else where the first host spell began
    uint64_t host_spells() const { return spells_; }                   // how often the host decoder took over ...
    uint64_t resumes() const { return resumes_; }                      // ... and how often the device windows resumed behind it

    // The windows are made by a producer thread one ahead of the reader (two text slots): the device decodes window k + 1 while
    // window k is consumed.
    size_t run(uint8_t *out, size_t cap) {
        size_t got = 0;
        while (got < cap && !done_ && err_.empty()) {
            if (seq_) { got += seq_run(out + got, cap - got); continue; }
            if (!cur_) {                                     // the next full slot, or the end of what the producer makes
                std::unique_lock<std::mutex> l(m_);
                cv_.wait(l, [&] { return slot_[cons_].full || finished_; });
                if (slot_[cons_].full) { cur_ = &slot_[cons_]; t_off_ = 0; }
                else {                                       // the producer is through: an error, the end, or the hand-over to the host decoder
                    l.unlock();
                    if (producer_.joinable()) producer_.join();
                    if (!perr_.empty()) { err_ = perr_; break; }
                    if (want_seq_) { start_sequential(); continue; }
                    done_ = true;
                    break;
                }
            }
            const size_t k = std::min(cap - got, (size_t)(cur_->have - t_off_));
            memcpy(out + got, cur_->text + t_off_, k);
            got += k;
            t_off_ += k;
            if (t_off_ == cur_->have) {
                { std::lock_guard<std::mutex> l(m_); cur_->full = false; }
                cv_.notify_all();
                cur_ = nullptr;
                cons_ ^= 1;
            }
        }
        return got;
    }

private:
    const uint8_t *in_;
    size_t n_;
    DgBackend *be_;
    Geometry g_;
    int crc_threads_;
    std::vector<uint8_t> win_;
    std::vector<snk_gunzip_chunk> chunks_;
    std::vector<snk_gunzip_member> ends_;
    std::vector<uint32_t> order_;
    struct Slot { uint8_t *text = nullptr; uint64_t have = 0; bool full = false; };
    Slot slot_[2];
    Slot *cur_ = nullptr;                      // (consumer) the slot being read
    int cons_ = 0;
    uint64_t t_off_ = 0;
    uint8_t *text_ = nullptr;                  // (producer) the slot being filled
    std::thread producer_;
    std::mutex m_;
    std::condition_variable cv_;
    bool quit_ = false, finished_ = false, want_seq_ = false;
    std::string perr_;                         // the producer's error, handed over with finished_
    uint64_t pos_bit_ = 0;                     // the next block header of the stream
    bool first_of_member_ = false, stream_done_ = false;
    bool done_ = false, seq_ = false;
    std::string err_;
    uint32_t mcrc_ = 0;                        // CRC-32 / length of the current member so far
    uint64_t mlen_ = 0;
    bool member_checkable_ = true, mid_start_ = false;
    uint64_t windows_ = 0, fallback_bit_ = ~0ull;
    uint64_t spells_ = 0, resumes_ = 0, spell_bits_ = 0, windows_at_spell_ = ~0ull;
    // sequential fallback
    GzipInflate sq_;
    std::vector<uint8_t> sbuf_;
    size_t s_have_ = 0, s_off_ = 0;

    void fail(const std::string &m) { if (err_.empty()) err_ = m; }

    // gzip member header at byte p (RFC 1952): the bit offset of its first deflate block
    bool member_header_at(uint64_t p, uint64_t &first_bit) const {
        if (p + 18 > n_ || in_[p] != 0x1F || in_[p + 1] != 0x8B || in_[p + 2] != 8) return false;
        const unsigned flg = in_[p + 3];
        if (flg & 0xE0) return false;
        p += 10;
        if (flg & 4) { if (p + 2 > n_) return false; p += 2 + ((unsigned)in_[p] | ((unsigned)in_[p + 1] << 8)); }
        if (flg & 8) { while (p < n_ && in_[p]) ++p; ++p; }
        if (flg & 16) { while (p < n_ && in_[p]) ++p; ++p; }
        if (flg & 2) p += 2;
        if (p >= n_) return false;
        first_bit = p * 8;
        return true;
    }

    // CRC-32 of text_[lo, hi) on crc_threads_ threads (pieces combined in order)
    uint32_t crc_range(uint64_t lo, uint64_t hi) const {
        const uint64_t len = hi - lo;
        const int T = (int)std::min<uint64_t>((uint64_t)crc_threads_, len / (4u << 20) + 1);
        if (T <= 1) return snk::crc32_fast(0, text_ + lo, (size_t)len);
        std::vector<uint32_t> part((size_t)T);
        std::vector<std::thread> th;
        for (int t = 0; t < T; ++t)
            th.emplace_back([&, t] {
                const uint64_t a = lo + len * (uint64_t)t / (uint64_t)T, b = lo + len * (uint64_t)(t + 1) / (uint64_t)T;
                part[(size_t)t] = snk::crc32_fast(0, text_ + a, (size_t)(b - a));
            });
        for (auto &x : th) x.join();
        uint32_t c = 0;
        for (int t = 0; t < T; ++t) {
            const uint64_t a = lo + len * (uint64_t)t / (uint64_t)T, b = lo + len * (uint64_t)(t + 1) / (uint64_t)T;
            c = t == 0 ? part[0] : (uint32_t)crc32_combine(c, part[(size_t)t], (z_off_t)(b - a));
        }
        return c;
    }
    void add_text(uint64_t lo, uint64_t hi) {
        if (hi <= lo) return;
        if (member_checkable_) mcrc_ = (uint32_t)crc32_combine(mcrc_, crc_range(lo, hi), (z_off_t)(hi - lo));
        mlen_ += hi - lo;
    }
    bool end_member(uint32_t want_crc, uint32_t want_isize) {
        if (member_checkable_ && (mcrc_ != want_crc || (uint32_t)mlen_ != want_isize)) {
            if (seq_) fail("invalid gzip data (CRC-32 / length of a member)"); else pfail("invalid gzip data (CRC-32 / length of a member)");
            return false;
        }
        mcrc_ = 0; mlen_ = 0; member_checkable_ = true;
        return true;
    }

    // producer thread: windows into the two slots until the stream ends, an error, or the hand-over to the sequential decoder
    void produce() {
        int p = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> l(m_);
                cv_.wait(l, [&] { return !slot_[p].full || quit_; });
                if (quit_) break;
            }
            uint64_t have = 0;
            const bool ok = !stream_done_ && next_window(p, have);
            std::lock_guard<std::mutex> l(m_);
            if (ok) { slot_[p].text = text_; slot_[p].have = have; slot_[p].full = true; }
            else finished_ = true;
            cv_.notify_all();
            if (!ok) break;
            p ^= 1;
        }
        std::lock_guard<std::mutex> l(m_);
        finished_ = true;
        cv_.notify_all();
    }
    void pfail(const std::string &m) { if (perr_.empty()) perr_ = m; }

    bool next_window(int slot, uint64_t &have) {
        const uint64_t wb = (pos_bit_ >> 3) & ~3ull;
        const uint64_t nbytes = std::min<uint64_t>(g_.window_bytes, n_ - wb);
        const uint64_t first = pos_bit_ - wb * 8;
        const uint32_t nc = (uint32_t)((nbytes + g_.chunk_bytes - 1) / g_.chunk_bytes);
        if (!be_->decode(in_ + wb, nbytes, first, first_of_member_, chunks_.data(), ends_.data())) { want_seq_ = true; return false; }
        ++windows_;
        // the chain: chunks whose start is where the one before stopped
        order_.clear();
        uint64_t expect = first, text_bytes = 0;
        bool at_file_end = false, at_member_seam = false;
        for (uint32_t c = 0; c < nc; ++c) {
            const snk_gunzip_chunk &ck = chunks_[c];
            if (ck.start_bit == ~0ull || ck.start_bit < expect) continue;     // no start found / a start inside a block already decoded
            if (ck.start_bit != expect) break;                                 // the chain does not meet: what follows is not trusted
            if (ck.status != SNK_GZ_OK) break;
            order_.push_back(c);
            text_bytes += ck.n_syms;
            expect = ck.end_bit;
            // stream_end says "the trailer ended where the input given to the kernel ends": the end of the FILE only when this
            // window is the file's tail -- otherwise a member (BGZF, concatenated .gz) ended exactly on the window's edge and the
            // next window starts at a member header (that case used to end the stream silently)
            if (ck.stream_end) { if (wb + nbytes == n_) at_file_end = true; else at_member_seam = true; break; }
        }
        // a chunk that ran into the end of the WINDOW (not of the file) reports invalid data and is decoded again by the next
        // window; a window that yields nothing at all cannot make progress
        if (order_.empty()) { want_seq_ = true; return false; }
        text_ = be_->text_buffer(slot, (size_t)text_bytes + 64);      // (grown on demand: pinning gigabytes up front costs a second)
        if (!text_) { want_seq_ = true; return false; }
        std::vector<uint8_t> wout(HIST);
        if (!be_->resolve(order_.data(), (uint32_t)order_.size(), first_of_member_ ? nullptr : win_.data(), text_, text_bytes, wout.data())) { want_seq_ = true; return false; }
        // members that ended in this window: CRC-32 and ISIZE over the text
        uint64_t at = 0, from = 0;
        for (uint32_t c : order_) {
            const snk_gunzip_chunk &ck = chunks_[c];
            for (uint32_t e = 0; e < ck.n_ends; ++e) {
                const snk_gunzip_member &m = ends_[(size_t)ck.ends_off + e];
                add_text(from, at + m.sym_index);
                from = at + m.sym_index;
                if (!end_member(m.crc, m.isize)) return false;
            }
            at += ck.n_syms;
        }
        add_text(from, at);
        win_.swap(wout);
        have = text_bytes;
        pos_bit_ = wb * 8 + expect;
        // where the next window starts: inside a member, unless the last chunk ended exactly behind a member header
        const snk_gunzip_chunk &last = chunks_[order_.back()];
        first_of_member_ = last.known_from != 0xFFFFFFFFu && last.known_from == last.n_syms;
        if (at_member_seam) {                  // (all of the member's text has been checked above: mlen_ is 0 here)
            const uint64_t p = pos_bit_ >> 3;
            uint64_t fb = 0;
            // as GzipInflate::member_header(): the magic first -- whatever is not 1F 8B behind a member is trailing garbage (zlib's
            // gzread stops quietly), a member that begins within the last 17 bytes is truncated
            if (n_ - p < 2 || in_[p] != 0x1F || in_[p + 1] != 0x8B) stream_done_ = true;
            else if (n_ - p < 18) { pfail("truncated gzip member"); return false; }
            else if (member_header_at(p, fb)) { pos_bit_ = fb; first_of_member_ = true; }
            else { pfail("invalid gzip data (member header)"); return false; }
        }
        if (at_file_end) { stream_done_ = true; if (mlen_ != 0) { pfail("device inflate: text behind the last member"); return false; } }
        return true;
    }

    // ---- sequential host decoder from the block header at pos_bit_ on (the rest of the stream)
    void start_sequential() {                  // (the producer has ended: its state -- position, window, CRC so far -- is the reader's now)
        if (fallback_bit_ == ~0ull) fallback_bit_ = pos_bit_;
        // one spell: up to the first block header a chunk further on; twice as far when the device made no window since the last spell
        spell_bits_ = (spells_ && windows_ == windows_at_spell_) ? std::min<uint64_t>(spell_bits_ * 2, 1ull << 40) : (uint64_t)g_.chunk_bytes * 8;
        windows_at_spell_ = windows_;
        ++spells_;
        if (getenv("SNK_PGZ_DEBUG")) fprintf(stderr, "device inflate: sequential from bit %llu (%s)\n", (unsigned long long)pos_bit_, be_->error().c_str());
        seq_ = true;
        sq_.init(in_, n_);
        sq_.set_verify_crc(false);
        sq_.start_at_block(pos_bit_);
        if (!getenv("SNK_DGZ_NO_RESUME")) sq_.set_stop_bit(pos_bit_ + spell_bits_);
        if (first_of_member_) { mcrc_ = 0; mlen_ = 0; }
        sbuf_.assign(HIST + ((size_t)1 << 22), 0);
        memcpy(sbuf_.data(), win_.data(), HIST);
        s_have_ = s_off_ = 0;
    }
    void seq_from_start() {
        seq_ = true;
        fallback_bit_ = 0;
        sq_.init(in_, n_);
        sq_.set_verify_crc(false);
        sbuf_.assign(HIST + ((size_t)1 << 22), 0);
        s_have_ = s_off_ = 0;
    }
    // the host spell is over (sq_ stopped in front of a block header, everything it made has been handed out): the device windows
    // go on from there -- position, the 32 KiB behind it and the member's CRC / length so far are the producer's again
    void resume_device() {
        pos_bit_ = sq_.bitpos();
        memcpy(win_.data(), sbuf_.data() + s_have_, HIST);
        first_of_member_ = sq_.at_member_start();
        s_have_ = s_off_ = 0;
        seq_ = false;
        ++resumes_;
        if (getenv("SNK_PGZ_DEBUG")) fprintf(stderr, "device inflate: windows resume at bit %llu\n", (unsigned long long)pos_bit_);
        {
            std::lock_guard<std::mutex> l(m_);
            finished_ = false; want_seq_ = false;
            slot_[0].full = slot_[1].full = false;
        }
        cur_ = nullptr; cons_ = 0;
        producer_ = std::thread([this] { produce(); });
    }
    size_t seq_run(uint8_t *out, size_t cap) {
        if (s_off_ == s_have_) {
            if (sq_.stopped() && fallback_bit_ != 0) { resume_device(); return 0; }
            if (s_have_) memmove(sbuf_.data(), sbuf_.data() + s_have_, HIST);      // keep the window in front
            uint8_t *p = sbuf_.data() + HIST;
            s_have_ = sq_.run(p, sbuf_.size() - HIST);
            s_off_ = 0;
            if (sq_.error()) { fail(sq_.error()); return 0; }
            size_t from = 0;
            auto piece = [&](size_t a, size_t b) {
                if (b > a) { mcrc_ = (uint32_t)crc32_combine(mcrc_, snk::crc32_fast(0, p + a, b - a), (z_off_t)(b - a)); mlen_ += b - a; }
            };
            for (const auto &e : sq_.member_ends()) {
                piece(from, e.out_off);
                if (!end_member(e.crc, e.isize)) return 0;
                from = e.out_off;
            }
            piece(from, s_have_);
            if (s_have_ == 0 && sq_.done()) { done_ = true; return 0; }
        }
        const size_t n = std::min(cap, s_have_ - s_off_);
        memcpy(out, sbuf_.data() + HIST + s_off_, n);
        s_off_ += n;
        return n;
    }
};

}  // namespace snk
#endif
