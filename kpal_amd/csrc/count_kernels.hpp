// count_kernels.hpp -- k-mer counting kernels for gfx950 (MI355X).
//
// Replaces the per-character Python loop of Profile.from_sequences (kpal/klib.py:154-168).
// Three strategies, all bit-exact (integer adds commute):
//   * global atomic   : one 64-bit global_atomic_add_x2 per k-mer straight into the 4^k table
//                       (any k; the k >= 13 path, where the table is 0.5 - 32 GiB).
//   * LDS direct      : k <= 7, the whole table is privatised per workgroup in LDS as u32
//                       (bank-replicated for tiny k), merged with one atomic per non-zero bin.
//   * partition       : 8 <= k <= 16: partition_kernels.hpp.
#pragma once
#include "kpal_device.hpp"

namespace kpal {

// ------------------------------------------------------------------------------------------
// Strategy 1: global atomics.  Each wave walks `steps_per_wave` consecutive 1-KiB steps.
// ------------------------------------------------------------------------------------------
template <int K>
__global__ __launch_bounds__(256) void count_global_atomic_kernel(Span s, uint64_t steps_per_wave,
                                                                  unsigned long long *__restrict__ table)
{
    const int lane = threadIdx.x & 63;
    const uint64_t wave = (uint64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const uint64_t step0 = wave * steps_per_wave;
    const uint64_t total_steps = (s.nchunks + 63) / 64;
    if (step0 >= total_steps) return;
    const uint64_t step1 = min(step0 + steps_per_wave, total_steps);
    Chunk carry = load_chunk(s, (int64_t)(step0 * 64) - 1);
    for (uint64_t st = step0; st < step1; ++st) {
        uint64_t window;
        uint32_t mask;
        if (interior_range(s, st * 64, st * 64 + 64)) wave_step<K, false>(s, (int64_t)(st * 64 + lane), carry, window, mask);
        else wave_step<K, true>(s, (int64_t)(st * 64 + lane), carry, window, mask);
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            if (mask & (1u << (15 - j))) atomicAdd(&table[kmer_at<K>(window, j)], 1ULL);
        }
    }
}

// ------------------------------------------------------------------------------------------
// Batched per-record counting (Profile.from_fasta_by_record, kpal/klib.py:114-133): the flat
// stream holds many records, starts[r] = position of record r's first byte (ascending,
// starts[R] = stream length; a separator byte lies between records).  Every k-mer is added to the
// table of the record its LAST byte lies in -- windows never span a separator, so that is the
// record that contains it.  Records are short (that is what the batch entry point is for): one
// global atomic per k-mer into out[r * 4^k + kmer].
// ------------------------------------------------------------------------------------------
template <int K>
__global__ __launch_bounds__(256) void count_records_kernel(Span s, uint64_t steps_per_wave,
                                                            const uint64_t *__restrict__ starts, uint32_t n_records,
                                                            unsigned long long *__restrict__ out)
{
    const int lane = threadIdx.x & 63;
    const uint64_t wave = (uint64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const uint64_t step0 = wave * steps_per_wave;
    const uint64_t total_steps = (s.nchunks + 63) / 64;
    if (step0 >= total_steps) return;
    const uint64_t step1 = min(step0 + steps_per_wave, total_steps);
    Chunk carry = load_chunk(s, (int64_t)(step0 * 64) - 1);
    for (uint64_t st = step0; st < step1; ++st) {
        uint64_t window;
        uint32_t mask;
        if (interior_range(s, st * 64, st * 64 + 64)) wave_step<K, false>(s, (int64_t)(st * 64 + lane), carry, window, mask);
        else wave_step<K, true>(s, (int64_t)(st * 64 + lane), carry, window, mask);
        if (mask == 0) continue;
        // record of the lane's first byte: largest r with starts[r] <= position (positions are relative
        // to the first fed byte, s.lo)
        // (the stream may begin inside the lane's chunk -- a batch of records that starts at any byte of the flattened text,
        // kpal_fasta_records_count -- so every byte's position is taken relative to s.lo on its own)
        const uint64_t p0 = (st * 64 + lane) * 16;
        const uint64_t rel0 = p0 >= s.lo ? p0 - s.lo : 0;
        uint32_t lo = 0, hi = n_records;   // invariant: starts[lo] <= rel0 < starts[hi]
        while (hi - lo > 1) {
            const uint32_t mid = (lo + hi) >> 1;
            if (starts[mid] <= rel0) lo = mid;
            else hi = mid;
        }
        uint32_t r = lo;
        uint64_t next_start = starts[r + 1];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const uint64_t rel = p0 + j >= s.lo ? p0 + j - s.lo : 0;
            while (rel >= next_start && r + 1 < n_records) {   // records shorter than a chunk: may advance more than once
                ++r;
                next_start = starts[r + 1];
            }
            if (mask & (1u << (15 - j))) atomicAdd(&out[((uint64_t)r << (2 * K)) + kmer_at<K>(window, j)], 1ULL);
        }
    }
}

// ------------------------------------------------------------------------------------------
// Strategy 2: LDS-direct histogram for k <= 7 (4^k <= 16384 bins).  R bank-interleaved
// replicas (R*4^k*4 B = 64 KiB at most) so that for tiny k the 64 lanes of a wave do not
// serialise on a handful of addresses: replica r = lane % R lives at dword bin*R + r.
// A workgroup never sees 2^32 k-mers in one launch (the host caps a launch at 2^31 bytes).
// ------------------------------------------------------------------------------------------
template <int K>
struct LdsDirectCfg {
    static constexpr int kBins = 1 << (2 * K);
    static constexpr int kRep = (16384 / kBins) > 32 ? 32 : (16384 / kBins);
};

template <int K>
__global__ __launch_bounds__(512) void count_lds_direct_kernel(Span s, uint64_t steps_per_wave,
                                                               unsigned long long *__restrict__ table)
{
    constexpr int BINS = LdsDirectCfg<K>::kBins;
    constexpr int R = LdsDirectCfg<K>::kRep;
    __shared__ uint32_t h[BINS * R];
    for (int i = threadIdx.x; i < BINS * R; i += blockDim.x) h[i] = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int rep = lane % R;
    const uint64_t wave = (uint64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const uint64_t step0 = wave * steps_per_wave;
    const uint64_t total_steps = (s.nchunks + 63) / 64;
    if (step0 < total_steps) {
        const uint64_t step1 = min(step0 + steps_per_wave, total_steps);
        Chunk carry = load_chunk(s, (int64_t)(step0 * 64) - 1);
        for (uint64_t st = step0; st < step1; ++st) {
            uint64_t window;
            uint32_t mask;
            if (interior_range(s, st * 64, st * 64 + 64)) wave_step<K, false>(s, (int64_t)(st * 64 + lane), carry, window, mask);
            else wave_step<K, true>(s, (int64_t)(st * 64 + lane), carry, window, mask);
            // branch-free: a k-mer that must not be counted adds 0 to whatever bin its bits name
#pragma unroll
            for (int j = 0; j < 16; ++j) atomicAdd(&h[kmer_at<K>(window, j) * R + rep], (mask >> (15 - j)) & 1u);
        }
    }
    __syncthreads();
    for (int b = threadIdx.x; b < BINS; b += blockDim.x) {
        unsigned long long v = 0;
#pragma unroll
        for (int r = 0; r < R; ++r) v += h[b * R + r];
        if (v) atomicAdd(&table[b], v);
    }
}

// ------------------------------------------------------------------------------------------
// Synthetic reads.  One thread per 16 output bytes.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void synth_reads_kernel(uint64_t seed, uint64_t first_read, uint64_t n_reads,
                                                          uint32_t read_len, int noisy, uint8_t *__restrict__ out)
{
    const uint64_t total = n_reads * (uint64_t)(read_len + 1);
    const uint64_t nvec = (total + 15) / 16;
    for (uint64_t v = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; v < nvec;
         v += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t o0 = v * 16;
        uint64_t r = o0 / (read_len + 1);
        uint32_t pos = (uint32_t)(o0 - r * (read_len + 1));
        uint64_t cached_word = ~0ULL, w = 0;
        uint8_t bytes[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            uint8_t c;
            if (pos == read_len) {
                c = '\n';
            } else {
                const uint64_t g = (first_read + r) * (uint64_t)read_len + pos;
                if ((g >> 5) != cached_word) {
                    cached_word = g >> 5;
                    w = mix64(seed * 0xD1342543DE82EF95ULL + cached_word);
                }
                const uint32_t code = (uint32_t)(w >> (2 * (g & 31))) & 3u;
                c = (uint8_t)((0x54474341u >> (8 * code)) & 0xFFu);  // "ACGT"
                if (noisy) {
                    const uint64_t hsh = mix64(~seed + g);
                    if (hsh % 1000 == 0) c = 'N';
                    else if (hsh % 100 == 1) c |= 0x20;
                }
            }
            bytes[q] = c;
            if (++pos > read_len) {
                pos = 0;
                ++r;
            }
        }
        if (o0 + 16 <= total) {
            uint4 pk;
            pk.x = bytes[0] | (bytes[1] << 8) | (bytes[2] << 16) | ((uint32_t)bytes[3] << 24);
            pk.y = bytes[4] | (bytes[5] << 8) | (bytes[6] << 16) | ((uint32_t)bytes[7] << 24);
            pk.z = bytes[8] | (bytes[9] << 8) | (bytes[10] << 16) | ((uint32_t)bytes[11] << 24);
            pk.w = bytes[12] | (bytes[13] << 8) | (bytes[14] << 16) | ((uint32_t)bytes[15] << 24);
            *reinterpret_cast<uint4 *>(out + o0) = pk;
        } else {
            for (int q = 0; q < 16 && o0 + q < total; ++q) out[o0 + q] = bytes[q];
        }
    }
}

}  // namespace kpal
