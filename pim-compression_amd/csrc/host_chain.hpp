// host_chain.hpp -- the host pre-scan of a framed stream's u32 size chain (snappy_decompress.c:306-341), in parallel.
// Plain C++, no HIP: tests/test_host_chain.py compiles it on the CPU.
//
// The drop-in pair splits a file's blocks over devices before anything is copied, so it needs every block's offset first,
// and the chain is a serial pointer chase through host memory: ~7 ms per GiB of compressed stream (32,768 hops, a cache miss
// each) -- nothing for one device, whose pipeline walks the chain chunk by chunk underneath the copies, but the whole of
// `pre` for N devices, and it does not shrink with N.  Same remedy as on the device (csrc/snappy_kernels.hpp,
// chain_*_kernel): T threads each find a block boundary at the start of their share of the stream -- a plausible size field
// (1 .. 32 + BS + BS/6) followed by a literal tag, whose chain keeps landing on such positions and which some such position
// within one maximal block before it points at exactly -- and walk from there to the next thread's start.  Exactness does
// not rest on the recognition: share 0 starts at the first block, and a walk that starts on a boundary and ends EXACTLY on
// the next share's start proves that start to be on the chain; the shares are accepted iff every one ends on the next
// one's start, the last on the stream's end, and the hops number num_blocks.  Otherwise the caller walks serially (which
// also produces the precise error for a damaged stream).
#pragma once
#include <stdint.h>
#include <string.h>
#include <atomic>
#include <thread>
#include <vector>

namespace host_chain {

inline uint32_t le32(const uint8_t* p)
{
    uint32_t v;
    memcpy(&v, p, 4);          // (little-endian hosts only, like the rest of the library)
    return v;
}
inline uint32_t max_block(uint32_t block_size) { return 32u + block_size + block_size / 6u; }

// could a block start at offset o of buf[0, len)?  `first` = offset of the first block
inline bool plausible(const uint8_t* buf, uint64_t len, uint64_t first, uint32_t maxc, uint64_t o, uint64_t* next)
{
    if (o < first || o + 5 > len) return false;
    const uint32_t size = le32(buf + o);
    *next = o + 4 + (uint64_t)size;
    return size - 1u < maxc && *next <= len && (buf[o + 4] & 3u) == 0u;
}

constexpr uint64_t kNone = ~0ull;

// the first recognised boundary in [lo, hi), kNone if there is none
inline uint64_t find_anchor(const uint8_t* buf, uint64_t len, uint64_t first, uint32_t maxc, uint64_t lo, uint64_t hi)
{
    unsigned tries = 0;                                           // bounded work whatever the bytes are
    for (uint64_t a = lo; a < hi && tries < 64; ++a) {
        uint64_t at = a, nx = 0;
        if (!plausible(buf, len, first, maxc, a, &nx)) continue;
        ++tries;
        bool good = true;
        at = nx;
        for (int hop = 0; hop < 4 && good && at != len; ++hop) {
            good = plausible(buf, len, first, maxc, at, &nx);
            at = nx;
        }
        if (!good) continue;
        const uint64_t reach = (uint64_t)maxc + 4;
        const uint64_t from = a > reach + first ? a - reach : first;
        bool pointed = false;
        for (uint64_t c = from; c + 5 <= a && !pointed; ++c) pointed = plausible(buf, len, first, maxc, c, &nx) && nx == a;
        if (pointed) return a;
    }
    return kNone;
}

// off[0 .. num_blocks] (the last entry = len) if the shares fit together; false: walk serially.
// threads <= 1, or a stream too short to share out, returns false at once.
// min_share: bytes of stream a thread must have to be worth starting (~50 us; the serial walk does ~8 ms per GiB)
inline bool parallel_walk(const uint8_t* buf, uint64_t len, uint64_t first, uint64_t num_blocks, uint32_t block_size, unsigned threads,
                          std::vector<uint64_t>& off, uint64_t min_share = 16u << 20)
{
    if (len <= first || num_blocks == 0 || min_share == 0) return false;
    const uint64_t body = len - first;
    uint64_t shares = body / min_share;
    if (shares > threads) shares = threads;
    if (shares < 2) return false;
    const uint32_t maxc = max_block(block_size);
    // one thread per share: find the share's starting point, publish it, then walk to the next published starting point
    // (a share waits only for the shares behind it to have LOOKED, not walked)
    std::vector<uint64_t> anchor(shares, kNone);
    std::vector<std::atomic<int>> looked(shares);
    for (auto& l : looked) l.store(0, std::memory_order_relaxed);
    std::vector<std::vector<uint64_t>> hops(shares);
    std::vector<char> ok(shares, 1);
    {
        std::vector<std::thread> th;
        for (uint64_t k = 0; k < shares; ++k)
            th.emplace_back([&, k] {
                anchor[k] = k == 0 ? first : find_anchor(buf, len, first, maxc, first + body * k / shares, first + body * (k + 1) / shares);
                looked[k].store(1, std::memory_order_release);
                if (anchor[k] == kNone) return;                   // no walker: the one before it walks through this share
                uint64_t next = len;
                for (uint64_t j = k + 1; j < shares; ++j) {
                    while (!looked[j].load(std::memory_order_acquire)) std::this_thread::yield();
                    if (anchor[j] != kNone) {
                        next = anchor[j];
                        break;
                    }
                }
                std::vector<uint64_t>& h = hops[k];
                h.reserve((size_t)(num_blocks / shares + 64));
                uint64_t at = anchor[k];
                while (at < next) {
                    if (at + 4 > len || h.size() > num_blocks) {
                        ok[k] = 0;
                        return;
                    }
                    h.push_back(at);
                    at += 4 + (uint64_t)le32(buf + at);
                }
                if (at != next) ok[k] = 0;
            });
        for (auto& t : th) t.join();
    }
    uint64_t total = 0;
    for (uint64_t k = 0; k < shares; ++k) {
        if (!ok[k]) return false;
        total += hops[k].size();
    }
    if (total != num_blocks) return false;
    off.resize(num_blocks + 1);
    uint64_t i = 0;
    for (uint64_t k = 0; k < shares; ++k) {
        if (!hops[k].empty()) memcpy(&off[i], hops[k].data(), hops[k].size() * sizeof(uint64_t));
        i += hops[k].size();
    }
    off[num_blocks] = len;
    return true;
}

}  // namespace host_chain
