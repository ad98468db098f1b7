// Host side of the K1 ceiling experiment (csrc/ablation/k1_oracle_table.hpp; VERDICT r03 item 1, gate (b)).
// Runs the reference parse (snappy_compress.c:284-413, restated as in oracle/snappy_oracle.c) over a container and writes one
// u32 per input position:
//   bits  0-15  the content of the position's hash-table slot at the moment the parse first probes inside the position's
//               64-aligned window (= what the GPU kernel's gather of that window reads); 0 for windows never probed
//   bits 16-21  nearest earlier position of the same aligned window with the same hash (lane number), bit 22: there is one,
//   bit  23     its 4 bytes equal this position's, bits 24-27: how many of the 8 bytes behind the key match (0..8),
//   bit  28     that partner has a partner itself                      (bits 16-28 do not depend on the parse)
//   gcc -O2 -fopenmp -shared -fPIC -o /tmp/libgate_b_records.so tools/gate_b_records.c
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
static inline uint32_t le32(const uint8_t* p) { uint32_t v; memcpy(&v, p, 4); return v; }

// `in` must be readable up to len + 16 (zero padded); rec has len + 64 entries, zeroed by the caller
static void one_block(const uint8_t* blk, uint32_t n, uint32_t* rec, uint16_t* table)
{
    uint32_t ts = 256; while (ts < 16384 && ts < n) ts <<= 1;
    int lg = 0; while ((1u << (lg + 1)) <= ts) lg++;
    const int shift = 32 - lg;
#define HASH(pos) ((le32(blk + (pos)) * 0x1e35a7bdu) >> shift)
    // parse-independent part: partners inside each aligned window
    for (uint32_t base = 0; base < n; base += 64) {
        uint32_t h[64]; int has[64];
        const uint32_t cnt = (n - base < 64) ? n - base : 64;
        for (uint32_t l = 0; l < cnt; l++) {
            h[l] = HASH(base + l);
            has[l] = 0;
            for (int j = (int)l - 1; j >= 0; j--)
                if (h[j] == h[l]) {
                    const uint8_t *a = blk + base + l, *b = blk + base + j;
                    uint32_t ext = 0; while (ext < 8 && a[4 + ext] == b[4 + ext]) ext++;
                    rec[base + l] |= ((uint32_t)j << 16) | (1u << 22) | ((le32(a) == le32(b)) ? 1u << 23 : 0u) | (ext << 24) | (has[j] ? 1u << 28 : 0u);
                    has[l] = 1;
                    break;
                }
        }
    }
    if (n < 15) return;
    memset(table, 0, 2 * ts);
    const uint32_t limit = n - 15;
    uint32_t ip = 1, next_hash = HASH(ip), cur_win = 0xffffffffu;
#define ENTER(pos) do { const uint32_t wb_ = (pos) & ~63u; if (wb_ != cur_win) { cur_win = wb_; \
        for (uint32_t q_ = wb_; q_ < wb_ + 64 && q_ + 4 <= n; q_++) rec[q_] = (rec[q_] & 0xffff0000u) | table[HASH(q_)]; } } while (0)
    for (;;) {
        uint32_t skip = 32, next_ip = ip, cand;
        do {
            ip = next_ip; const uint32_t h = next_hash; next_ip = ip + (skip++ >> 5); if (next_ip > limit) return; next_hash = HASH(next_ip);
            ENTER(ip);
            cand = table[h]; table[h] = (uint16_t)ip;
        } while (le32(blk + ip) != le32(blk + cand));
        uint32_t cb;
        do {
            uint32_t a = cand + 4, b = ip + 4, m = 4; while (b < n && blk[a] == blk[b]) { a++; b++; m++; }
            ip += m; if (ip >= limit) return;
            table[HASH(ip - 1)] = (uint16_t)(ip - 1);
            ENTER(ip);
            const uint32_t h = HASH(ip); cand = table[h]; cb = le32(blk + cand); table[h] = (uint16_t)ip;
        } while (le32(blk + ip) == cb);
        next_hash = HASH(ip + 1); ip++;
    }
}

// prevw[p] = nearest earlier position of p's block with the same hash that lies BEFORE p's 64-aligned window, 0 if none
// (what a parse-independent pass 1 would deliver; position 0 is never inserted, so 0 can stand for "none")
static void one_block_prevw(const uint8_t* blk, uint32_t n, uint16_t* prevw, uint16_t* last)
{
    uint32_t ts = 256; while (ts < 16384 && ts < n) ts <<= 1;
    int lg = 0; while ((1u << (lg + 1)) <= ts) lg++;
    const int shift = 32 - lg;
    memset(last, 0, 2 * ts);
    for (uint32_t base = 0; base < n; base += 64) {
        const uint32_t end = (base + 64 < n) ? base + 64 : n;
        for (uint32_t p = base; p < end; p++) prevw[p] = (p + 4 <= n) ? last[HASH(p)] : 0;
        for (uint32_t p = base ? base : 1; p < end; p++) if (p + 4 <= n) last[HASH(p)] = (uint16_t)p;
    }
}

void gate_b_prevw(const uint8_t* in, uint64_t len, uint32_t block_size, uint16_t* prevw)
{
    const int64_t nb = (int64_t)((len + block_size - 1) / block_size);
#pragma omp parallel
    {
        uint16_t* last = (uint16_t*)malloc(2 * 16384);
#pragma omp for schedule(dynamic, 16)
        for (int64_t b = 0; b < nb; b++) {
            const uint64_t start = (uint64_t)b * block_size;
            const uint32_t n = (uint32_t)((len - start < block_size) ? len - start : block_size);
            one_block_prevw(in + start, n, prevw + start, last);
        }
        free(last);
    }
}

void gate_b_records(const uint8_t* in, uint64_t len, uint32_t block_size, uint32_t* rec)
{
    const int64_t nb = (int64_t)((len + block_size - 1) / block_size);
#pragma omp parallel
    {
        uint16_t* table = (uint16_t*)malloc(2 * 16384);
#pragma omp for schedule(dynamic, 16)
        for (int64_t b = 0; b < nb; b++) {
            const uint64_t start = (uint64_t)b * block_size;
            const uint32_t n = (uint32_t)((len - start < block_size) ? len - start : block_size);
            one_block(in + start, n, rec + start, table);
        }
        free(table);
    }
}
