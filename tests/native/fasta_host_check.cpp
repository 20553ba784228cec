// fasta_host_check.cpp -- AddressSanitizer / ThreadSanitizer harness of kpal_amd/csrc/fasta_host.hpp: the host side of the FASTA
// ingest (source = memory or a byte range of a file read with pread by the pool's threads, optional prefix; read-ahead into the
// other staging buffer; text before the first header skipped; line state and the look-ahead behind a run of blanks carried with
// every chunk).  The chunks are flattened by a sequential restatement of the rules of fasta_kernels.hpp (which the GPU tests
// check against the tokeniser of tests/test_gpu_fasta.py) and their concatenation must equal the text flattened in ONE piece --
// for chunk sizes from 16 bytes (a seam at every position, runs of blanks many chunks long) to the whole text, staging buffers
// of exactly the chunk size (a byte written past one is an ASan finding), and a pool whose jobs really split (KPAL_READ_THREADS
// from the test, split = 1 KiB).  Test infrastructure; run by tests/test_native_sanitized.py.
#include <cstdio>
#include <cstdlib>
#include <string>

#include <fcntl.h>

#include "../../kpal_amd/csrc/fasta_host.hpp"

using namespace kpal;

static uint64_t rng_state = 0x2545F4914F6CDD1Dull;
static uint64_t rnd()
{
    rng_state ^= rng_state << 13;
    rng_state ^= rng_state >> 7;
    rng_state ^= rng_state << 17;
    return rng_state;
}

static bool soft(uint8_t c) { return c == 9 || c == 11 || c == 12 || (c >= 28 && c <= 31) || c == 0x85 || c == 0xA0; }

// the rules of fasta_kernels.hpp on one chunk, byte by byte
static void flatten_chunk(const uint8_t *d, size_t n, int state, bool tail, std::string &out)
{
    // follow[i]: only blanks between byte i (inclusive) and the end of its line -- or of the chunk: then `tail` decides (one backward pass)
    std::vector<char> follow(n + 1);
    follow[n] = tail ? 1 : 0;
    for (size_t i = n; i-- > 0;) {
        const uint8_t c = d[i];
        follow[i] = fa_host_is_eol(c) ? 1 : ((c == ' ' || soft(c)) ? follow[i + 1] : 0);
    }
    int cur = state;
    for (size_t i = 0; i < n; ++i) {
        const uint8_t c = d[i];
        if (cur == 0) {
            if (c == '>') {
                out.push_back('\n');
                cur = 1;
                continue;
            }
            cur = 2;
        }
        if (cur == 1) {
            if (fa_host_is_eol(c)) cur = 0;
            continue;
        }
        if (fa_host_is_eol(c)) {
            cur = 0;
            continue;
        }
        if (c == ' ') continue;
        if (soft(c) && follow[i + 1]) continue;
        out.push_back((char)c);
    }
}

static std::string random_text(size_t lines, size_t long_blanks)
{
    std::string t;
    const char *eols[] = {"\n", "\r\n", "\r"};
    const char blanks[] = {' ', '\t', '\v', '\f', (char)0x1c, (char)0x85, (char)0xA0};
    if (rnd() & 1) t += "text before the first header\n;comment\n";
    if ((rnd() & 3) == 0) t += "no newline before the header";      // (then the header below is not at a line start unless an EOL follows)
    if (rnd() & 1) t += eols[rnd() % 3];
    for (size_t l = 0; l < lines; ++l) {
        const uint64_t kind = rnd() % 10;
        if (kind == 0 || l == 0) {
            t += ">rec";
            t += std::to_string(l);
            if (rnd() & 1) t += " title with > inside \t";
        } else if (kind == 1) {
            // an empty line, or a line of blanks only
            for (uint64_t i = rnd() % 4; i > 0; --i) t.push_back(blanks[rnd() % sizeof(blanks)]);
        } else {
            const size_t len = (size_t)(rnd() % 80);
            for (size_t i = 0; i < len; ++i) {
                const uint64_t r = rnd() % 100;
                t.push_back(r < 90 ? "ACGTacgtNn"[rnd() % 10] : (r < 95 ? blanks[rnd() % sizeof(blanks)] : (r < 97 ? '>' : '-')));
            }
            // runs of blanks at the end of the line / in its middle, some longer than any small chunk
            if ((rnd() & 3) == 0) {
                const size_t run = (rnd() % 8 == 0 && long_blanks) ? long_blanks + (size_t)(rnd() % 97) : (size_t)(rnd() % 12);
                for (size_t i = 0; i < run; ++i) t.push_back(blanks[rnd() % sizeof(blanks)]);
                if (rnd() & 1) t += "ACGT";                          // ... then the run was interior
            }
        }
        if (l + 1 < lines || (rnd() & 1)) t += eols[rnd() % 3];
    }
    return t;
}

static int failures = 0;

static std::string chunked(FaSource src, size_t stage, size_t *nchunks)
{
    uint8_t *b0 = (uint8_t *)malloc(stage), *b1 = (uint8_t *)malloc(stage);   // exactly `stage` bytes each
    std::string out;
    int waits = 0;
    {
        FaChunker ck(src, b0, b1, stage, [&](int) { ++waits; return 0; }, 1024);
        FaChunk c;
        int rc;
        size_t n = 0;
        while ((rc = ck.next(c)) == 1) {
            if (c.n == 0 || c.n > stage || c.data < (c.slot ? b1 : b0) || c.data + c.n > (c.slot ? b1 : b0) + stage) ++failures;
            flatten_chunk(c.data, c.n, c.state, c.tail_trailing, out);
            ++n;
        }
        if (rc != 0) ++failures;
        if (nchunks) *nchunks = n;
    }
    free(b0);
    free(b1);
    return out;
}

int main()
{
    char path[] = "/tmp/kpal_fasta_host_check_XXXXXX";
    const int fd = mkstemp(path);
    if (fd < 0) {
        perror("mkstemp");
        return 2;
    }
    size_t cases = 0;
    const size_t stages[] = {16, 17, 31, 64, 257, 4096, 70000, (size_t)1 << 22};
    for (int round = 0; round < 30; ++round) {
        const std::string text = random_text((size_t)(rnd() % (round < 14 ? 40 : 1500)) + 1, round % 3 == 0 ? 300 : (round % 3 == 1 ? 5000 : 0));
        // the whole text in one piece (text before the first header skipped)
        std::string want;
        {
            const size_t first = fasta_first_header((const uint8_t *)text.data(), text.size(), true);
            flatten_chunk((const uint8_t *)text.data() + first, text.size() - first, 0, true, want);
        }
        if (pwrite(fd, text.data(), text.size(), 0) != (ssize_t)text.size() || ftruncate(fd, (off_t)text.size()) != 0) {
            perror("pwrite");
            return 2;
        }
        for (size_t stage : stages) {
            if (stage < 64 && text.size() > 20000) continue;        // (a seam at every position: small texts only -- the look-ahead is per chunk)
            FaSource mem;
            mem.mem = (const uint8_t *)text.data();
            mem.end = text.size();
            size_t nchunks = 0;
            if (chunked(mem, stage, &nchunks) != want) {
                std::printf("MISMATCH memory source, round %d, stage %zu (%zu chunks)\n", round, stage, nchunks);
                ++failures;
            }
            FaSource file;
            file.fd = fd;
            file.end = text.size();
            if (chunked(file, stage, nullptr) != want) {
                std::printf("MISMATCH file source, round %d, stage %zu\n", round, stage);
                ++failures;
            }
            // prefix + range: the text cut at a header-less place is NOT what the product does (its prefix is a header line + bases);
            // here: the first p bytes from the first header on as the prefix, the rest as the range of the file
            const size_t first = fasta_first_header((const uint8_t *)text.data(), text.size(), true);
            if (first < text.size()) {
                const size_t p = (size_t)(rnd() % std::min<size_t>(text.size() - first, 3 * stage + 50)) + 1;
                FaSource pre;
                pre.fd = fd;
                pre.pos = first + p;
                pre.end = text.size();
                pre.prefix = (const uint8_t *)text.data() + first;
                pre.prefix_left = p;
                if (chunked(pre, stage, nullptr) != want) {
                    std::printf("MISMATCH prefix + file range, round %d, stage %zu, prefix %zu\n", round, stage, p);
                    ++failures;
                }
            }
            cases += 3;
        }
    }
    // a read error surfaces as -1 with the errno: a range beyond the end of the file
    {
        FaSource bad;
        bad.fd = fd;
        bad.pos = 0;
        bad.end = (uint64_t)1 << 40;
        uint8_t *b0 = (uint8_t *)malloc(4096), *b1 = (uint8_t *)malloc(4096);
        {
            FaChunker ck(bad, b0, b1, 4096, [](int) { return 0; }, 1024);
            FaChunk c;
            int rc;
            while ((rc = ck.next(c)) == 1) {
            }
            if (rc != -1 || ck.io_errno() == 0) ++failures;
        }
        free(b0);
        free(b1);
    }
    close(fd);
    unlink(path);
    if (failures) {
        std::printf("fasta_host_check: %d failure(s) in %zu cases\n", failures, cases);
        return 1;
    }
    std::printf("fasta_host_check: %zu cases, pool of %d\nSANITIZE_OK\n", cases, HostPool::instance().size());
    return 0;
}
