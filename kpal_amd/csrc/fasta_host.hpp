// fasta_host.hpp -- the host side of the FASTA ingest, free of HIP: where the text comes from (a byte range of a file read by
// the pool's threads, or host memory, optionally preceded by a short prefix) and how it is cut into the chunks the flattening
// kernels of fasta_kernels.hpp take.  kpal_count.hip drives a FaChunker with its pinned staging buffers and queues the
// copies and kernels per chunk; tests/native/fasta_host_check.cpp drives the same class with malloc'ed buffers under
// AddressSanitizer / ThreadSanitizer and compares the chunks, flattened by a restatement of the kernels' rules, with the
// text flattened in one piece.
//
// A chunk may be cut ANYWHERE.  What the kernels cannot see from inside a chunk travels with it:
//   * state         what the chunk's first byte continues (0 line start, 1 inside a header line, 2 inside a sequence line);
//   * tail_trailing the chunk ends in a run of blanks: whether those trail their line (str.rstrip() drops them) or stand
//                   inside it (a tab stays and separates k-mer windows) is decided by the first byte that is not a blank
//                   AFTER the chunk -- the chunker looks ahead in the source for it (a few bytes; rare).
// Text before the first header line is skipped (kpal/klib.py:111: Bio.SeqIO starts at the first '>').
#pragma once
#include <cerrno>
#include <cstdint>
#include <cstring>
#include <functional>
#include <vector>

#include <unistd.h>

#include "host_pool.hpp"

namespace kpal {

static inline bool fa_host_is_eol(uint8_t c) { return c == '\n' || c == '\r'; }
// blanks the flattening drops at the end of a line only (str.rstrip() of a latin-1 text handle), and the space it drops everywhere
static inline bool fa_host_is_blank(uint8_t c) { return c == ' ' || c == 9 || c == 11 || c == 12 || (c >= 28 && c <= 31) || c == 0x85 || c == 0xA0; }

// First byte of the first header line ('>' at a line start) of buf[0, n), or n.  at_line_start: buf[0] begins a line.
static inline size_t fasta_first_header(const uint8_t *buf, size_t n, bool at_line_start)
{
    size_t next_cr = 0;        // position of the next '\r' at or after the scan position (n: none); found lazily, once per '\r'
    bool cr_known = false;
    auto next_eol = [&](size_t from) -> size_t {
        if (!cr_known || next_cr < from) {
            const void *cr = from < n ? memchr(buf + from, '\r', n - from) : nullptr;
            next_cr = cr ? (size_t)((const uint8_t *)cr - buf) : n;
            cr_known = true;
        }
        const size_t stop = next_cr;   // a '\n' beyond the next '\r' does not matter
        const void *nl = from < stop ? memchr(buf + from, '\n', stop - from) : nullptr;
        return nl ? (size_t)((const uint8_t *)nl - buf) : stop;
    };
    size_t i = 0;
    if (!at_line_start) {
        i = next_eol(0);
        if (i >= n) return n;
        ++i;
    }
    while (i < n) {
        if (buf[i] == '>') return i;
        i = next_eol(i);
        if (i >= n) return n;
        ++i;
    }
    return n;
}

struct FaSource {
    int fd = -1;                       // a byte range [pos, end) of a file ...
    const uint8_t *mem = nullptr;      // ... or of host memory (mem[pos .. end))
    uint64_t pos = 0, end = 0;
    const uint8_t *prefix = nullptr;   // text that logically precedes the range (a record's header and the bases before a cut)
    size_t prefix_left = 0;
    bool more() const { return prefix_left > 0 || pos < end; }
};

// (returns 0, or the errno of the failed read; EIO when the file turned out shorter than its size said)
static inline int pread_all(int fd, uint8_t *dst, size_t n, uint64_t off)
{
    while (n) {
        const ssize_t r = pread(fd, dst, n, (off_t)off);
        if (r < 0 && errno == EINTR) continue;
        if (r < 0) return errno ? errno : EIO;
        if (r == 0) return EIO;
        dst += r;
        off += (uint64_t)r;
        n -= (size_t)r;
    }
    return 0;
}

// n bytes of the source's range from `pos` into dst, by the pool: the page cache hands ONE reader ~9 GB/s (a copy_to_user per
// page), the link takes 56.  start: the pool's workers copy while the caller does something else; ok[] says afterwards
// (HostPool::wait) whether every part arrived (0, or the errno of its reader: errno itself is per thread).  Pieces below
// `split` bytes (4 MiB) are not split.
static inline void fa_copy_start(const FaSource &s, uint8_t *dst, uint64_t pos, size_t n, std::vector<int> &ok, size_t split = (size_t)4 << 20)
{
    HostPool &pool = HostPool::instance();
    const int parts = std::max(1, (int)std::min<size_t>((size_t)pool.size(), n / std::max<size_t>(split, 1)));
    const size_t part = (((n + (size_t)parts - 1) / (size_t)parts) + 4095) & ~(size_t)4095;
    ok.assign((size_t)parts, 0);
    int *flags = ok.data();
    const int fd = s.fd;
    const uint8_t *mem = s.mem;
    pool.start(parts, [=](int i) {
        const size_t off = (size_t)i * part;
        if (off >= n) return;
        const size_t len = std::min(part, n - off);
        if (mem) memcpy(dst + off, mem + pos + off, len);
        else flags[i] = pread_all(fd, dst + off, len, pos + off);
    });
}

struct FaChunk {
    const uint8_t *data = nullptr;     // inside the staging buffer of `slot`
    size_t n = 0;
    int state = 0;                     // what data[0] continues
    bool tail_trailing = true;         // blanks at the chunk's end trail their line
    int slot = 0;
};

class FaChunker {
public:
    // buf[0], buf[1]: staging buffers of `stage` bytes each (pinned memory in the library).  wait_slot(slot) is called before
    // a buffer is written again: it returns (0) once whatever the caller queued on the buffer's previous contents -- the DMA
    // out of it -- is done, or an error code that ends the chunking.  split: see fa_copy_start (tests make it tiny).
    FaChunker(FaSource &src, uint8_t *buf0, uint8_t *buf1, size_t stage, std::function<int(int)> wait_slot, size_t split = (size_t)4 << 20)
        : src_(src), stage_(stage), wait_slot_(std::move(wait_slot)), split_(split)
    {
        buf_[0] = buf0;
        buf_[1] = buf1;
    }
    ~FaChunker()
    {
        if (ra_active_) HostPool::instance().wait();   // (an error return must not leave the pool writing into a staging buffer)
    }
    FaChunker(const FaChunker &) = delete;
    FaChunker &operator=(const FaChunker &) = delete;

    int io_errno() const { return io_errno_; }   // after next() returned -1
    int user_error() const { return user_error_; }   // after next() returned -2: what wait_slot returned

    // 1: `out` is the next chunk (valid until the call after the next one: the other buffer is filled first); 0: the end of
    // the text; -1: a read failed (io_errno()); -2: wait_slot failed (user_error()).
    int next(FaChunk &out)
    {
        for (;;) {
            uint8_t *hp = buf_[slot_];
            size_t n = 0;
            if (ra_active_) {
                HostPool::instance().wait();
                ra_active_ = false;
                for (int e : ra_ok_)
                    if (e) {
                        io_errno_ = e;
                        return -1;
                    }
                n = ra_n_;                               // (read into buf_[slot_] from src_.pos while the chunk before was handled)
                src_.pos += ra_n_;
            } else {
                if (!src_.more()) return 0;
                if (int rc = wait_slot_(slot_)) {
                    user_error_ = rc;
                    return -2;
                }
                const long got = fill(hp, stage_);
                if (got < 0) return -1;
                if (got == 0) return 0;
                n = (size_t)got;
            }
            // the next chunk: the pool reads it while this one is scanned, copied and its kernels are issued
            if (src_.prefix_left == 0 && src_.pos < src_.end) {
                const int other = slot_ ^ 1;
                if (int rc = wait_slot_(other)) {
                    user_error_ = rc;
                    return -2;
                }
                ra_n_ = (size_t)std::min<uint64_t>(stage_, src_.end - src_.pos);
                fa_copy_start(src_, buf_[other], src_.pos, ra_n_, ra_ok_, split_);
                ra_active_ = true;
            }
            const int slot = slot_;
            slot_ ^= 1;
            size_t first = 0;
            if (skipping_) {
                first = fasta_first_header(hp, n, at_line_start_);
                if (first >= n) {
                    at_line_start_ = fa_host_is_eol(hp[n - 1]);
                    continue;
                }
                skipping_ = false;
                state_ = 0;
            }
            out.data = hp + first;
            out.n = n - first;
            out.state = state_;
            out.slot = slot;
            out.tail_trailing = true;
            if (fa_host_is_blank(hp[n - 1]) && src_.more()) {
                const int t = peek_trailing();
                if (t < 0) return -1;
                out.tail_trailing = t != 0;
            }
            // what the chunk after this one continues: the chunk's last line
            {
                const uint8_t *chunk = out.data;
                const size_t m = out.n;
                size_t e = m;
                while (e > 0 && !fa_host_is_eol(chunk[e - 1])) --e;      // e = one past the last end of line (0: none)
                if (e == 0) state_ = state_ == 0 ? (chunk[0] == '>' ? 1 : 2) : state_;
                else if (e == m) state_ = 0;
                else state_ = chunk[e] == '>' ? 1 : 2;
            }
            return 1;
        }
    }

private:
    // The next bytes of the source (at most `want`) into dst, now; returns how many (0: the end), -1 on a read error.
    long fill(uint8_t *dst, size_t want)
    {
        size_t got = 0;
        if (src_.prefix_left) {
            const size_t n = std::min(want, src_.prefix_left);
            memcpy(dst, src_.prefix, n);
            src_.prefix += n;
            src_.prefix_left -= n;
            got = n;
        }
        const size_t n = (size_t)std::min<uint64_t>(want - got, src_.end - src_.pos);
        if (n) {
            std::vector<int> ok;
            fa_copy_start(src_, dst + got, src_.pos, n, ok, split_);
            HostPool::instance().wait();
            for (int e : ok)
                if (e) {
                    io_errno_ = e;
                    return -1;
                }
            src_.pos += n;
            got += n;
        }
        return (long)got;
    }

    // The first byte of the source, from its current position on, that is not a blank: 1 if it ends a line or the text ends
    // first (the blanks before it trail their line), 0 otherwise; -1 on a read error.  The source is not advanced.
    int peek_trailing()
    {
        for (size_t i = 0; i < src_.prefix_left; ++i)
            if (!fa_host_is_blank(src_.prefix[i])) return fa_host_is_eol(src_.prefix[i]) ? 1 : 0;
        uint64_t at = src_.pos;
        uint8_t tmp[4096];
        while (at < src_.end) {
            const size_t n = (size_t)std::min<uint64_t>(sizeof(tmp), src_.end - at);
            const uint8_t *p = tmp;
            if (src_.mem) p = src_.mem + at;
            else if (int e = pread_all(src_.fd, tmp, n, at)) {
                io_errno_ = e;
                return -1;
            }
            for (size_t i = 0; i < n; ++i)
                if (!fa_host_is_blank(p[i])) return fa_host_is_eol(p[i]) ? 1 : 0;
            at += n;
        }
        return 1;
    }

    FaSource &src_;
    uint8_t *buf_[2];
    size_t stage_;
    std::function<int(int)> wait_slot_;
    size_t split_;
    int slot_ = 0;
    bool ra_active_ = false;
    size_t ra_n_ = 0;
    std::vector<int> ra_ok_;
    int state_ = 0;               // what the next chunk's first byte continues
    bool skipping_ = true;        // only text before the first header so far
    bool at_line_start_ = true;
    int io_errno_ = 0, user_error_ = 0;
};

}  // namespace kpal
