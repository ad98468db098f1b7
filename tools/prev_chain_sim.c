// Analysis tool (CPU), gate (a) of the two-pass K1: can the hash table of the reference parse
// (snappy_compress.c:284-413, restated as in oracle/snappy_oracle.c) be replaced by
//   prev[p]  = nearest earlier position q >= 1 with hash(q) == hash(p)   (parse-INDEPENDENT, 0 = none)
//   inserted = one bit per position, set where the parse stores into the table (:346-347 probes, :391-392 ip-1, :397 ip)
// so that table[hash(p)] == first inserted position along p -> prev[p] -> prev[prev[p]] ..., else 0 (:145 empty slot)?
// The tool runs the reference parse with its real table, answers every table read a second time by walking the chain,
// ASSERTS both answers agree, and reports how long the walks are:
//   * per probe of the reference parse (what a lazy walk would pay);
//   * per lane of a 64-position window that starts where the parse stands when it leaves the previous window (what a
//     speculative 64-lane gather would pay: every lane resolves its candidate against the positions before the window;
//     chain elements inside the window are the lanes' own business and are counted separately).
//   gcc -O2 -o /tmp/prev_chain_sim tools/prev_chain_sim.c && /tmp/prev_chain_sim <file> [block size]
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
static inline uint32_t le32(const uint8_t* p) { uint32_t v; memcpy(&v, p, 4); return v; }
#define MAXB 65536
#define HMAX 64
static uint16_t table[16384], last[16384], prev[MAXB];
static uint8_t ins[MAXB];
static uint64_t probes, probe_hops[HMAX + 1], probe_zero, probe_hop_sum, probe_max;
static uint64_t windows, lanes, lane_hops[HMAX + 1], lane_zero, lane_hop_sum, lane_inwin[HMAX + 1], win_max[HMAX + 1], lane_first_ok;
static uint64_t hop_dist_le[6];  // hop targets within 1,2,4,8,16,32 KiB behind the window start
static uint64_t hop_loads;       // global prev[] loads a lane issues (every out-of-window hop after the first)
static uint64_t mism;
static uint64_t inserted_total, positions_total;

static uint32_t walk(uint32_t p, uint32_t* hops) {
  uint32_t c = prev[p], h = 1;
  while (c != 0 && !ins[c]) { c = prev[c]; h++; }
  *hops = h;
  return c;
}
static void probe_stat(uint32_t ip, uint32_t cand) {
  uint32_t h; uint32_t c = walk(ip, &h);
  if (c != cand) mism++;
  probes++; probe_hop_sum += h; if (h > probe_max) probe_max = h; probe_hops[h > HMAX ? HMAX : h]++; if (c == 0) probe_zero++;
}
// the speculative view: window [ws, ws+64) resolved against positions < ws, bitmap as it stands now
static void window_stat(uint32_t ws, uint32_t n) {
  uint32_t wmax = 0; windows++;
  for (uint32_t l = 0; l < 64 && ws + l + 4 <= n; l++) {
    uint32_t p = ws + l, c = prev[p], inwin = 0, h = 0;
    while (c >= ws && c != 0) { c = prev[c]; inwin++; }          // lanes of this window: resolved among the lanes
    // c < ws now: the first out-of-window element (comes with the window's own prev[] read / bpermutes)
    h = 1;
    if (c != 0 && ins[c]) lane_first_ok++;
    while (c != 0 && !ins[c]) {
      hop_loads++;                                               // needs prev[c]: a 2-byte load from the block's prev array
      uint32_t d = ws - c; for (int k = 0; k < 6; k++) if (d <= (1024u << k)) hop_dist_le[k]++;
      c = prev[c]; h++;
    }
    lanes++; lane_hop_sum += h; lane_hops[h > HMAX ? HMAX : h]++; lane_inwin[inwin > HMAX ? HMAX : inwin]++; if (c == 0) lane_zero++;
    if (h > wmax) wmax = h;
  }
  win_max[wmax > HMAX ? HMAX : wmax]++;
}
static void block(const uint8_t* blk, uint32_t n) {
  uint32_t ts = 256; while (ts < 16384 && ts < n) ts <<= 1; int lg = 0; while ((1u << (lg + 1)) <= ts) lg++; const int shift = 32 - lg;
  memset(table, 0, sizeof table); memset(last, 0, sizeof last); memset(ins, 0, n);
#define HASH(pos) ((le32(blk + (pos)) * 0x1e35a7bdu) >> shift)
  if (n < 15) return;
  for (uint32_t p = 1; p + 4 <= n; p++) { uint32_t h = HASH(p); prev[p] = last[h]; last[h] = (uint16_t)p; }
  prev[0] = 0;
  const uint32_t limit = n - 15; uint32_t ip = 1, next_hash = HASH(ip), ws_next = 0;
#define WINDOW_CHECK() do { if (ip >= ws_next) { window_stat(ip, n); ws_next = ip + 64; } } while (0)
  for (;;) { uint32_t skip = 32, next_ip = ip, cand;
    do { ip = next_ip; uint32_t h = next_hash; next_ip = ip + (skip++ >> 5); if (next_ip > limit) goto done; next_hash = HASH(next_ip);
         WINDOW_CHECK();
         cand = table[h]; probe_stat(ip, cand); table[h] = (uint16_t)ip; ins[ip] = 1; } while (le32(blk + ip) != le32(blk + cand));
    uint32_t cb;
    do { uint32_t a = cand + 4, b = ip + 4, m = 4; while (b < n && blk[a] == blk[b]) { a++; b++; m++; } ip += m; if (ip >= limit) goto done;
         uint32_t h1 = HASH(ip - 1); table[h1] = (uint16_t)(ip - 1); ins[ip - 1] = 1;
         WINDOW_CHECK();
         uint32_t h = HASH(ip); cand = table[h]; probe_stat(ip, cand); cb = le32(blk + cand); table[h] = (uint16_t)ip; ins[ip] = 1; } while (le32(blk + ip) == cb);
    next_hash = HASH(ip + 1); ip++; }
done:
  for (uint32_t p = 0; p < n; p++) inserted_total += ins[p];
  positions_total += n;
}
static void hist(const char* name, const uint64_t* h, uint64_t total) {
  printf("%s:", name); uint64_t acc = 0; int p50 = -1, p90 = -1, p99 = -1, p999 = -1;
  for (int i = 0; i <= HMAX; i++) { acc += h[i];
    if (p50 < 0 && acc * 2 >= total) p50 = i; if (p90 < 0 && acc * 10 >= total * 9) p90 = i;
    if (p99 < 0 && acc * 100 >= total * 99) p99 = i; if (p999 < 0 && acc * 1000 >= total * 999) p999 = i; }
  printf(" p50 %d p90 %d p99 %d p99.9 %d |", p50, p90, p99, p999);
  for (int i = 0; i <= 12; i++) printf(" %d:%.4f", i, (double)h[i] / total);
  uint64_t rest = 0; for (int i = 13; i <= HMAX; i++) rest += h[i]; printf(" 13+:%.4f\n", (double)rest / total);
}
int main(int argc, char** argv) {
  FILE* f = fopen(argv[1], "rb"); if (!f) return 2; fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
  uint8_t* d = malloc(n + 64); if (fread(d, 1, n, f) != (size_t)n) return 1; memset(d + n, 0, 64);
  uint32_t bs = argc > 2 ? atoi(argv[2]) : 32768;
  for (long o = 0; o < n; o += bs) block(d + o, (uint32_t)((n - o < bs) ? n - o : bs));
  printf("file %s  %ld bytes  block size %u\n", argv[1], n, bs);
  printf("chain walk == table read on every probe: %s (%lu mismatches of %lu probes)\n", mism ? "NO" : "yes", mism, probes);
  printf("positions inserted into the table: %.3f of all positions\n", (double)inserted_total / positions_total);
  printf("reference probes: %.2f per 64 input bytes; hops mean %.3f max %lu; walks that end at 'no position' (empty slot -> 0): %.4f\n",
         (double)probes / (positions_total / 64.0), (double)probe_hop_sum / probes, probe_max, (double)probe_zero / probes);
  hist("  hops per reference probe", probe_hops, probes);
  printf("64-lane windows: %lu (%.3f per 64 input bytes), lanes %lu\n", windows, (double)windows / (positions_total / 64.0), lanes);
  printf("  lanes whose first out-of-window chain element is an inserted position: %.4f; lanes that end at 'none': %.4f\n",
         (double)lane_first_ok / lanes, (double)lane_zero / lanes);
  printf("  out-of-window hops per lane: mean %.3f; prev[] hop loads per window %.2f\n", (double)lane_hop_sum / lanes, (double)hop_loads / windows);
  hist("  out-of-window hops per lane", lane_hops, lanes);
  hist("  in-window chain elements per lane", lane_inwin, lanes);
  hist("  max hops over the lanes of a window (dependent rounds per window)", win_max, windows);
  printf("  hop loads whose target lies within 1/2/4/8/16/32 KiB behind the window:");
  for (int k = 0; k < 6; k++) printf(" %.3f", hop_loads ? (double)hop_dist_le[k] / hop_loads : 0.0); printf("\n");
  return mism ? 3 : 0;
}
