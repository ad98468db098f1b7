// Analysis tool (CPU): the hash-table accesses of the reference parse (snappy_compress.c:284-413, restated as in
// oracle/snappy_oracle.c) per 64-byte window of a data file, and what write-back caches of 256..8192 slots in front of
// the table (direct-mapped / 2-way) would leave of them.  Used for profiles/r03_k1_global_table_bound.txt:
//   gcc -O2 -o /tmp/slot_cache_sim tools/slot_cache_sim.c && /tmp/slot_cache_sim <file> [block size]
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
static inline uint32_t le32(const uint8_t*p){uint32_t v;memcpy(&v,p,4);return v;}
#define NC 6
static const uint32_t csize[NC]={256,512,1024,2048,4096,8192};
static uint64_t g_reads, g_filtered_reads, g_writes, windows;
static uint64_t c_rd_miss[NC], c_wr_evict[NC], c2_rd_miss[NC], c2_wr_evict[NC];
static uint16_t table[16384]; static uint8_t written[16384];
// direct mapped: tag = slot index (or 0xffff empty), dirty
static uint16_t ctag[NC][8192]; static uint8_t cdirty[NC][8192];
// 2-way LRU
static uint16_t c2tag[NC][8192]; static uint8_t c2dirty[NC][8192]; static uint8_t c2lru[NC][4096];
static void cache_access(uint32_t h, int is_write, int was_written){
  for(int c=0;c<NC;c++){
    uint32_t i=h&(csize[c]-1);
    if(ctag[c][i]!=h){
      if(ctag[c][i]!=0xffff && cdirty[c][i]) c_wr_evict[c]++;
      if(!is_write){ if(was_written) c_rd_miss[c]++; ctag[c][i]=h; cdirty[c][i]=0; }
      else { ctag[c][i]=h; cdirty[c][i]=1; }
    } else if(is_write) cdirty[c][i]=1;
    // 2-way
    uint32_t sets=csize[c]/2, s=h&(sets-1); uint16_t*t=&c2tag[c][2*s]; uint8_t*d=&c2dirty[c][2*s];
    int w=-1; if(t[0]==h)w=0; else if(t[1]==h)w=1;
    if(w<0){ w=c2lru[c][s]; if(t[w]!=0xffff&&d[w]) c2_wr_evict[c]++; if(!is_write&&was_written) c2_rd_miss[c]++; t[w]=h; d[w]=is_write; }
    else if(is_write) d[w]=1;
    c2lru[c][s]=!w;
  }
}
static void rd(uint32_t h){ g_reads++; if(written[h]) g_filtered_reads++; cache_access(h,0,written[h]); }
static void wr(uint32_t h){ g_writes++; written[h]=1; cache_access(h,1,1); }
static void block(const uint8_t*blk,uint32_t n){
  uint32_t ts=256; while(ts<16384&&ts<n)ts<<=1; int lg=0; while((1u<<(lg+1))<=ts)lg++; const int shift=32-lg;
  memset(table,0,sizeof table); memset(written,0,sizeof written);
  memset(ctag,0xff,sizeof ctag); memset(c2tag,0xff,sizeof c2tag); memset(cdirty,0,sizeof cdirty); memset(c2dirty,0,sizeof c2dirty);
  windows+=n/64;
#define HASH(pos) ((le32(blk+(pos))*0x1e35a7bdu)>>shift)
  if(n<15)return; const uint32_t limit=n-15; uint32_t ip=1,next_hash=HASH(ip);
  for(;;){ uint32_t skip=32,next_ip=ip,cand;
    do{ ip=next_ip; uint32_t h=next_hash; next_ip=ip+(skip++>>5); if(next_ip>limit)return; next_hash=HASH(next_ip);
        rd(h); cand=table[h]; wr(h); table[h]=(uint16_t)ip; }while(le32(blk+ip)!=le32(blk+cand));
    uint32_t cb;
    do{ uint32_t a=cand+4,b=ip+4,m=4; while(b<n&&blk[a]==blk[b]){a++;b++;m++;} ip+=m; if(ip>=limit)return;
        uint32_t h1=HASH(ip-1); wr(h1); table[h1]=(uint16_t)(ip-1);
        uint32_t h=HASH(ip); rd(h); cand=table[h]; cb=le32(blk+cand); wr(h); table[h]=(uint16_t)ip; }while(le32(blk+ip)==cb);
    next_hash=HASH(ip+1); ip++; }
}
int main(int argc,char**argv){
  FILE*f=fopen(argv[1],"rb"); fseek(f,0,SEEK_END); long n=ftell(f); fseek(f,0,SEEK_SET); uint8_t*d=malloc(n+64); if(fread(d,1,n,f)!=(size_t)n)return 1; memset(d+n,0,64);
  uint32_t bs=argc>2?atoi(argv[2]):32768;
  for(long o=0;o<n;o+=bs) block(d+o,(uint32_t)((n-o<bs)?n-o:bs));
  printf("windows %lu  per window: probes(reads) %.2f  reads of written slots %.2f  stores %.2f\n",windows,(double)g_reads/windows,(double)g_filtered_reads/windows,(double)g_writes/windows);
  printf("global requests per window without a cache: %.2f\n",(double)(g_filtered_reads+g_writes)/windows);
  for(int c=0;c<NC;c++) printf("cache %5u entries: direct-mapped read misses %.2f + evictions %.2f = %.2f | 2-way %.2f + %.2f = %.2f\n",csize[c],
     (double)c_rd_miss[c]/windows,(double)c_wr_evict[c]/windows,(double)(c_rd_miss[c]+c_wr_evict[c])/windows,
     (double)c2_rd_miss[c]/windows,(double)c2_wr_evict[c]/windows,(double)(c2_rd_miss[c]+c2_wr_evict[c])/windows);
}
