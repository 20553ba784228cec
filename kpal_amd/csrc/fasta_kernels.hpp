// fasta_kernels.hpp -- FASTA text -> flat sequence stream on the device (gfx950).
//
// Replaces the tokenising the reference delegates to Bio.SeqIO.parse (kpal/klib.py:111): header
// lines are dropped, the lines of a record are concatenated (so k-mers span line breaks) and
// records are separated (so k-mers never span records).  Output is the flat byte stream the
// counting kernels consume: every record contributes its sequence bytes, whitespace removed,
// preceded by one '\n'.
//
// Rules (same as kpal_amd.klib._fasta_records, the host-side tokeniser):
//   * a line is a header iff its first byte is '>'; lines end at '\n' or '\r';
//   * the header's '>' becomes the record separator '\n', the rest of the header line is dropped;
//   * in sequence lines (Biopython's SimpleFastaParser: line.rstrip(), then ' ' and '\r' removed from the
//     joined record) spaces and line ends are dropped everywhere, other whitespace (\t, \v, \f, 0x1c..0x1f, 0x85, 0xa0)
//     only where it trails its line; an interior tab stays and separates k-mer windows like any byte
//     outside the alphabet (kpal/klib.py:152-156);
//   * the text of one feed is handed to these kernels in chunks cut ANYWHERE (64 MiB pieces of a file); `start_state` says
//     what the chunk's first byte continues: 0 = it is the first byte of a line (header iff '>'), 1 = the middle of a header
//     line, 2 = the middle of a sequence line.  The first chunk of a feed starts at a header (the host skips anything before
//     the first header, like Biopython does); when a chunk ends inside a run of blanks, `tail_trailing` says what the host found
//     behind it (the run trails its line, or a base follows), so "only whitespace follows on this line" is decided with it.
// Whether byte i is inside a header depends only on the first byte of its line, i.e. on the last
// end-of-line before i: a prefix-max over the buffer.  Five small passes: per-block last EOL,
// carry scan, per-block kept-byte count, offset scan, scatter (staged through LDS so the output
// is written with contiguous stores).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace kpal {

constexpr int kFaThreads = 256;
constexpr int kFaPerThread = 16;
constexpr int kFaBlockBytes = kFaThreads * kFaPerThread;  // 4 KiB per workgroup

__device__ __forceinline__ bool fa_is_eol(uint8_t c) { return c == '\n' || c == '\r'; }
// whitespace that str.rstrip() removes at the end of a line but that stays inside it
__device__ __forceinline__ bool fa_is_soft_space(uint8_t c) { return c == 9 || c == 11 || c == 12 || (c >= 28 && c <= 31) || c == 0x85 || c == 0xA0; }   // str.isspace() of a latin-1 text handle
// true iff only whitespace follows byte i on its line (rare bytes: a forward walk per occurrence); a walk that reaches the end of
// the chunk returns `tail_trailing` -- what the host found behind the chunk (fasta_host.hpp; true at the end of the text)
__device__ __forceinline__ bool fa_trailing(const uint8_t *__restrict__ in, uint64_t n, uint64_t i, bool tail_trailing)
{
    for (uint64_t j = i + 1; j < n; ++j) {
        const uint8_t d = in[j];
        if (fa_is_eol(d)) return true;
        if (d != ' ' && !fa_is_soft_space(d)) return false;
    }
    return tail_trailing;
}

// inclusive prefix-max of one int64 per thread over a 256-thread workgroup; returns the exclusive
// value for this thread (max over lower threads, or `seed`)
__device__ __forceinline__ long long fa_block_exclusive_max(long long v, long long seed, long long *sh /* [4] */)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    long long incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const long long o = __shfl_up(incl, d);
        if (lane >= d) incl = max(incl, o);
    }
    if (lane == 63) sh[wave] = incl;
    __syncthreads();
    long long before = seed;
#pragma unroll
    for (int w = 0; w < 4; ++w)
        if (w < wave) before = max(before, sh[w]);
    const long long up = __shfl_up(incl, 1);
    if (lane > 0) before = max(before, up);
    __syncthreads();
    return before;
}

// K1: index of the last end-of-line byte in each block (-1 if none).
__global__ __launch_bounds__(kFaThreads) void fa_last_eol_kernel(const uint8_t *__restrict__ in, uint64_t n,
                                                                 long long *__restrict__ last_eol)
{
    __shared__ long long sh[4];
    const uint64_t i0 = (uint64_t)blockIdx.x * kFaBlockBytes + (uint64_t)threadIdx.x * kFaPerThread;
    long long last = -1;
    for (int j = 0; j < kFaPerThread; ++j)
        if (i0 + j < n && fa_is_eol(in[i0 + j])) last = (long long)(i0 + j);
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) last = max(last, __shfl_down(last, d));
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = last;
    __syncthreads();
    if (threadIdx.x == 0) last_eol[blockIdx.x] = max(max(sh[0], sh[1]), max(sh[2], sh[3]));
}

// K2: carry[b] = last EOL index in blocks < b (-1 if none).  One workgroup, serial over chunks.
__global__ __launch_bounds__(256) void fa_carry_kernel(const long long *__restrict__ last_eol, uint32_t nblocks,
                                                       long long *__restrict__ carry)
{
    __shared__ long long sh[4];
    long long running = -1;
    for (uint32_t base = 0; base < nblocks; base += 256) {
        const uint32_t b = base + threadIdx.x;
        const long long v = b < nblocks ? last_eol[b] : -1;
        const long long before = fa_block_exclusive_max(v, running, sh);
        if (b < nblocks) carry[b] = before;
        // new running max = max over this chunk
        long long m = max(before, v);
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) m = max(m, __shfl_down(m, d));
        if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = m;
        __syncthreads();
        running = max(max(sh[0], sh[1]), max(sh[2], sh[3]));
        __syncthreads();
    }
}

// Classify the thread's 16 bytes: bit j of `keep` set iff byte j is emitted; out[j] its value.
__device__ __forceinline__ void fa_classify(const uint8_t *__restrict__ in, uint64_t n, uint64_t i0, long long prev_eol, int start_state,
                                            bool tail_trailing, uint32_t &keep, uint8_t (&out)[kFaPerThread])
{
    keep = 0;
    long long last = prev_eol;   // last EOL strictly before the current byte
    bool header = false, known = false;
    for (int j = 0; j < kFaPerThread; ++j) {
        const uint64_t i = i0 + j;
        if (i >= n) break;
        const uint8_t c = in[i];
        if (!known) {            // first byte of this thread, or first byte after an EOL
            // (no end of line yet in this chunk: the line began in the previous one, which knows what it is)
            header = last < 0 && start_state != 0 ? start_state == 1 : in[last + 1] == '>';
            known = true;
        }
        if (header) {
            if ((long long)i == last + 1 && !(last < 0 && start_state == 1)) {   // the '>' itself: record separator
                keep |= 1u << j;
                out[j] = '\n';
            }
        } else if (c != ' ' && !fa_is_eol(c) && !(fa_is_soft_space(c) && fa_trailing(in, n, i, tail_trailing))) {
            keep |= 1u << j;
            out[j] = c;
        }
        if (fa_is_eol(c)) {
            last = (long long)i;
            known = false;
        }
    }
}

// K3: kept bytes per block.
__global__ __launch_bounds__(kFaThreads) void fa_count_kernel(const uint8_t *__restrict__ in, uint64_t n,
                                                              const long long *__restrict__ carry, int start_state, int tail_trailing,
                                                              uint32_t *__restrict__ kept)
{
    __shared__ long long sh[4];
    __shared__ uint32_t shc[4];
    const uint64_t i0 = (uint64_t)blockIdx.x * kFaBlockBytes + (uint64_t)threadIdx.x * kFaPerThread;
    long long mine = -1;
    for (int j = 0; j < kFaPerThread; ++j)
        if (i0 + j < n && fa_is_eol(in[i0 + j])) mine = (long long)(i0 + j);
    const long long prev = fa_block_exclusive_max(mine, carry[blockIdx.x], sh);
    uint32_t keep;
    uint8_t out[kFaPerThread];
    fa_classify(in, n, i0, prev, start_state, tail_trailing != 0, keep, out);
    uint32_t c = __popc(keep);
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) c += __shfl_down(c, d);
    if ((threadIdx.x & 63) == 0) shc[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) kept[blockIdx.x] = shc[0] + shc[1] + shc[2] + shc[3];
}

// K4: exclusive scan of the per-block counts -> 64-bit offsets; offs[nblocks] = total.
__global__ __launch_bounds__(256) void fa_offset_kernel(const uint32_t *__restrict__ kept, uint32_t nblocks,
                                                        uint64_t *__restrict__ offs)
{
    __shared__ uint64_t sh[4];
    uint64_t running = 0;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (uint32_t base = 0; base < nblocks; base += 256) {
        const uint32_t b = base + threadIdx.x;
        const uint64_t v = b < nblocks ? kept[b] : 0;
        uint64_t incl = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint64_t o = __shfl_up(incl, d);
            if (lane >= d) incl += o;
        }
        if (lane == 63) sh[wave] = incl;
        __syncthreads();
        uint64_t before = running;
#pragma unroll
        for (int w = 0; w < 4; ++w)
            if (w < wave) before += sh[w];
        if (b < nblocks) offs[b] = before + incl - v;
        const uint64_t total = running + sh[0] + sh[1] + sh[2] + sh[3];
        __syncthreads();
        running = total;
    }
    if (threadIdx.x == 0) offs[nblocks] = running;
}

// K5: write the kept bytes of each block contiguously at offs[block].
__global__ __launch_bounds__(kFaThreads) void fa_scatter_kernel(const uint8_t *__restrict__ in, uint64_t n,
                                                                const long long *__restrict__ carry, int start_state, int tail_trailing,
                                                                const uint64_t *__restrict__ offs,
                                                                uint8_t *__restrict__ flat)
{
    __shared__ long long sh[4];
    __shared__ uint32_t wsum[4];
    __shared__ uint8_t stage[kFaBlockBytes];
    const uint64_t i0 = (uint64_t)blockIdx.x * kFaBlockBytes + (uint64_t)threadIdx.x * kFaPerThread;
    long long mine = -1;
    for (int j = 0; j < kFaPerThread; ++j)
        if (i0 + j < n && fa_is_eol(in[i0 + j])) mine = (long long)(i0 + j);
    const long long prev = fa_block_exclusive_max(mine, carry[blockIdx.x], sh);
    uint32_t keep;
    uint8_t out[kFaPerThread];
    fa_classify(in, n, i0, prev, start_state, tail_trailing != 0, keep, out);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t c = __popc(keep);
    uint32_t incl = c;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = __shfl_up(incl, d);
        if (lane >= d) incl += o;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    uint32_t at = incl - c;
#pragma unroll
    for (int w = 0; w < 4; ++w)
        if (w < wave) at += wsum[w];
    const uint32_t total = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    for (int j = 0; j < kFaPerThread; ++j)
        if (keep & (1u << j)) stage[at++] = out[j];
    __syncthreads();
    uint8_t *dst = flat + offs[blockIdx.x];
    for (uint32_t t = threadIdx.x; t < total; t += kFaThreads) dst[t] = stage[t];
}

// ---- record index (Profile.from_fasta_by_record, kpal/klib.py:114-133: one profile per record, named by the record) ----
// Which bytes start a record: MODE 0, in the FLATTENED stream: every '\n' (the only '\n' bytes of that stream are the separators the
// headers became); MODE 1, in the RAW text: a '>' at a line start (byte 0 of the text begins a line).  The r-th mark of either kind
// belongs to record r.  Two passes around fa_offset_kernel: marks per 4 KiB block, then their positions written in order.
template <int MODE>
__device__ __forceinline__ bool fa_is_mark(const uint8_t *__restrict__ in, uint64_t i)
{
    if (MODE == 0) return in[i] == '\n';
    return in[i] == '>' && (i == 0 || fa_is_eol(in[i - 1]));
}

template <int MODE>
__global__ __launch_bounds__(kFaThreads) void fa_mark_count_kernel(const uint8_t *__restrict__ in, uint64_t n, uint32_t *__restrict__ marks)
{
    __shared__ uint32_t shc[4];
    const uint64_t i0 = (uint64_t)blockIdx.x * kFaBlockBytes + (uint64_t)threadIdx.x * kFaPerThread;
    uint32_t c = 0;
    for (int j = 0; j < kFaPerThread; ++j)
        if (i0 + j < n && fa_is_mark<MODE>(in, i0 + j)) ++c;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) c += __shfl_down(c, d);
    if ((threadIdx.x & 63) == 0) shc[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) marks[blockIdx.x] = shc[0] + shc[1] + shc[2] + shc[3];
}

template <int MODE>
__global__ __launch_bounds__(kFaThreads) void fa_mark_scatter_kernel(const uint8_t *__restrict__ in, uint64_t n, const uint64_t *__restrict__ offs,
                                                                     uint64_t *__restrict__ positions)
{
    __shared__ uint32_t wsum[4];
    const uint64_t i0 = (uint64_t)blockIdx.x * kFaBlockBytes + (uint64_t)threadIdx.x * kFaPerThread;
    uint32_t mask = 0;
    for (int j = 0; j < kFaPerThread; ++j)
        if (i0 + j < n && fa_is_mark<MODE>(in, i0 + j)) mask |= 1u << j;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t c = __popc(mask);
    uint32_t incl = c;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = __shfl_up(incl, d);
        if (lane >= d) incl += o;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    uint64_t at = offs[blockIdx.x] + (incl - c);
#pragma unroll
    for (int w = 0; w < 4; ++w)
        if (w < wave) at += wsum[w];
    for (int j = 0; j < kFaPerThread; ++j)
        if (mask & (1u << j)) positions[at++] = i0 + (uint64_t)j;
}

// starts of a batch of records relative to the batch's first byte (count_records_kernel takes them so)
__global__ __launch_bounds__(256) void fa_rebase_kernel(const uint64_t *__restrict__ starts, uint64_t n, uint64_t base, uint64_t *__restrict__ out)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = starts[i] - base;
}

}  // namespace kpal
