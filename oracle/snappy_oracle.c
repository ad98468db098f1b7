/*
 * snappy_oracle.c -- CPU restatement of the reference's HOST Snappy block codec.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under pim-compression_amd/ (the product)
 * may include, link or call this file.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg use it, and there only as the checker /
 * reported CPU baseline -- never as the thing shipped or measured as "ours".
 *
 * Parity pinning: this restatement is pinned, in BOTH directions, by the
 * reference's own committed test vectors (reference test/NAME.txt + test/NAME.snappy,
 * used by its `make test_host`, snappy/Makefile:54-56), copied as data fixtures
 * to tests/golden/.  See tests/test_oracle_golden.py.  The reference sources
 * themselves are not buildable in this image without writing stand-in headers
 * (snappy_compress.c:1-3 needs the UPMEM SDK <dpu.h>, dpu_snappy.h:4 needs the
 * un-vendored PIM-common "common.h"), so there is no oracle/_ref build.
 *
 * Each function cites the reference file:line whose behaviour it restates.
 * The code is written position-indexed (offsets from block start) instead of
 * the reference's cursor-in-struct style; it is a restatement, not a copy.
 */
#include <stdint.h>
#include <stddef.h>
#include <string.h>
#include <stdlib.h>
#include <pthread.h>

#define ORACLE_OK 0
#define ORACLE_INVALID_INPUT 1     /* dpu_snappy.h:21-25 */
#define ORACLE_BUFFER_TOO_SMALL 2

/* ---- small helpers ---------------------------------------------------- */

/* little-endian 32-bit load, alignment-free (snappy_compress.c:120-129) */
static inline uint32_t le32(const uint8_t *p)
{
	return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
}

/* little-endian 32-bit store (snappy_compress.c:106-112) */
static inline void put_le32(uint8_t *p, uint32_t v)
{
	p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); p[2] = (uint8_t)(v >> 16); p[3] = (uint8_t)(v >> 24);
}

/* varint32 writer: 7 bits per byte, MSB = continue (snappy_compress.c:69-98) */
static size_t put_varint32(uint8_t *dst, uint32_t v)
{
	size_t k = 0;
	while (v >= 0x80) {
		dst[k++] = (uint8_t)(v | 0x80);
		v >>= 7;
	}
	dst[k++] = (uint8_t)v;
	return k;
}

/* varint32 reader, at most 5 bytes (snappy_decompress.c:23-37).
 * Returns bytes consumed, or 0 if the 5th byte still has the continue bit. */
static size_t get_varint32(const uint8_t *src, size_t avail, uint32_t *out)
{
	uint32_t v = 0;
	for (size_t k = 0; k < 5 && k < avail; k++) {
		uint8_t c = src[k];
		v |= (uint32_t)(c & 0x7f) << (7 * k);
		if (!(c & 0x80)) {
			*out = v;
			return k + 1;
		}
	}
	return 0;
}

/* snappy_compress.c:55-60 */
uint64_t oracle_max_compressed_length(uint64_t n)
{
	return n ? 32 + n + n / 6 : 0;
}

/* hash-table sizing rule (snappy_compress.c:139-146): 256 doubling up to 16384
 * until it covers the block length. */
uint32_t oracle_table_size(uint32_t n)
{
	uint32_t ts = 256;
	while (ts < 16384 && ts < n)
		ts <<= 1;
	return ts;
}

/* ---- element emitters -------------------------------------------------- */

/* snappy_compress.c:202-225 */
static uint8_t *emit_literal(uint8_t *op, const uint8_t *src, uint32_t len)
{
	uint32_t n = len - 1;
	if (n < 60) {
		*op++ = (uint8_t)(n << 2);
	} else {
		uint8_t *tag = op++;
		uint32_t cnt = 0;
		while (n > 0) {
			*op++ = (uint8_t)(n & 0xff);
			n >>= 8;
			cnt++;
		}
		*tag = (uint8_t)((59 + cnt) << 2);
	}
	memcpy(op, src, len);
	return op + len;
}

/* snappy_compress.c:234-245: one copy element, 4 <= len <= 64 */
static uint8_t *emit_copy_piece(uint8_t *op, uint32_t off, uint32_t len)
{
	if (len < 12 && off < 2048) {
		*op++ = (uint8_t)(1 + ((len - 4) << 2) + ((off >> 8) << 5));
		*op++ = (uint8_t)(off & 0xff);
	} else {
		*op++ = (uint8_t)(2 + ((len - 1) << 2));
		*op++ = (uint8_t)(off & 0xff);
		*op++ = (uint8_t)((off >> 8) & 0xff);
	}
	return op;
}

/* snappy_compress.c:254-272: split rule >=68 -> 64, >64 -> 60, rest */
static uint8_t *emit_copy(uint8_t *op, uint32_t off, uint32_t len)
{
	while (len >= 68) {
		op = emit_copy_piece(op, off, 64);
		len -= 64;
	}
	if (len > 64) {
		op = emit_copy_piece(op, off, 60);
		len -= 60;
	}
	return emit_copy_piece(op, off, len);
}

/* snappy_compress.c:176-193: common prefix of blk[a..] and blk[b..], b bounded by n */
static uint32_t match_extend(const uint8_t *blk, uint32_t a, uint32_t b, uint32_t n)
{
	uint32_t m = 0;
	while (b + 4 <= n && le32(blk + b) == le32(blk + a + m)) {
		b += 4;
		m += 4;
	}
	while (b < n && blk[a + m] == blk[b]) {
		b++;
		m++;
	}
	return m;
}

/*
 * Compress ONE block of n bytes (snappy_compress.c:284-413 compress_block +
 * :139-146 get_hash_table).  Writes u32le(size) followed by the elements at
 * dst; returns total bytes written (4 + size).  `table` must hold 16384 u16.
 */
uint32_t oracle_compress_block(const uint8_t *blk, uint32_t n, uint8_t *dst, uint16_t *table)
{
	const uint32_t ts = oracle_table_size(n);
	int lg = 0;
	while ((1u << (lg + 1)) <= ts)
		lg++;
	const int shift = 32 - lg;                                  /* :288 */
	memset(table, 0, ts * sizeof(uint16_t));                    /* :145 */
#define HASH(pos) ((le32(blk + (pos)) * 0x1e35a7bdu) >> shift) /* :161-166 */

	uint8_t *op = dst + 4;                                      /* :291 */
	uint32_t next_emit = 0;

	if (n >= 15) {                                              /* :299-301 */
		const uint32_t limit = n - 15;
		uint32_t ip = 1;                                        /* :305 */
		uint32_t next_hash = HASH(ip);
		for (;;) {
			uint32_t skip = 32;                                 /* :333 */
			uint32_t next_ip = ip;
			uint32_t cand;
			do {                                                /* :336-348 */
				ip = next_ip;
				uint32_t h = next_hash;
				next_ip = ip + (skip++ >> 5);
				if (next_ip > limit)
					goto remainder;
				next_hash = HASH(next_ip);
				cand = table[h];
				table[h] = (uint16_t)ip;
			} while (le32(blk + ip) != le32(blk + cand));

			op = emit_literal(op, blk + next_emit, ip - next_emit);   /* :355 */

			uint32_t cand_bytes;
			do {                                                /* :370-398 */
				uint32_t base = ip;
				uint32_t m = 4 + match_extend(blk, cand + 4, ip + 4, n);
				ip += m;
				op = emit_copy(op, base - cand, m);
				next_emit = ip;
				if (ip >= limit)
					goto remainder;
				table[HASH(ip - 1)] = (uint16_t)(ip - 1);
				uint32_t h = HASH(ip);
				cand = table[h];
				cand_bytes = le32(blk + cand);
				table[h] = (uint16_t)ip;
			} while (le32(blk + ip) == cand_bytes);

			next_hash = HASH(ip + 1);                           /* :400-401 */
			ip++;
		}
	}
remainder:
	if (next_emit < n)                                          /* :405-410 */
		op = emit_literal(op, blk + next_emit, n - next_emit);
#undef HASH
	put_le32(dst, (uint32_t)(op - dst - 4));                    /* :412 */
	return (uint32_t)(op - dst);
}

/* header = varint(U) varint(BS)  (snappy_compress.c:461-465) */
uint32_t oracle_write_header(uint8_t *dst, uint32_t total_len, uint32_t block_size)
{
	size_t k = put_varint32(dst, total_len);
	k += put_varint32(dst + k, block_size);
	return (uint32_t)k;
}

/*
 * Whole-buffer compress (snappy_compress.c:455-485 snappy_compress_host).
 * Returns compressed length, or 0 if dst_cap is too small for the worst case.
 */
uint64_t oracle_compress(const uint8_t *src, uint64_t n, uint32_t block_size, uint8_t *dst, uint64_t dst_cap)
{
	if (block_size == 0 || n > 0xffffffffull)
		return 0;
	uint64_t nblocks = (n + block_size - 1) / block_size;
	uint64_t worst = 10 + n + nblocks * (4 + 3) + n / 60 + 64;
	if (dst_cap < worst)
		return 0;
	uint16_t *table = (uint16_t *)malloc(16384 * sizeof(uint16_t));     /* :458 */
	uint8_t *op = dst + oracle_write_header(dst, (uint32_t)n, block_size);
	uint64_t done = 0;
	while (done < n) {                                                   /* :467-479 */
		uint32_t todo = (uint32_t)((n - done < block_size) ? (n - done) : block_size);
		op += oracle_compress_block(src + done, todo, op, table);
		done += todo;
	}
	free(table);
	return (uint64_t)(op - dst);
}

/* upper bound callers can use to size dst for oracle_compress */
uint64_t oracle_compress_bound(uint64_t n, uint32_t block_size)
{
	uint64_t nblocks = block_size ? (n + block_size - 1) / block_size : 0;
	return 10 + n + nblocks * 7 + n / 60 + 64;
}

/* ---- decompression ------------------------------------------------------ */

/* Parse the two header varints (snappy_decompress.c:193-198, :220-225).
 * Returns header length or 0 on malformed header. */
uint32_t oracle_read_header(const uint8_t *src, uint64_t n, uint32_t *total_len, uint32_t *block_size)
{
	size_t a = get_varint32(src, (size_t)n, total_len);
	if (!a)
		return 0;
	size_t b = get_varint32(src + a, (size_t)(n - a), block_size);
	if (!b)
		return 0;
	return (uint32_t)(a + b);
}

/*
 * Whole-buffer decompress following snappy_decompress.c:218-289
 * (snappy_decompress_host) including its leniencies: literal and copy writes
 * are clipped at the end of input / output (:145-147, :174-175); a copy whose
 * source lies before the start of the whole output buffer is the only hard
 * error (:169-173).  `out_len` must be the uncompressed length from the header
 * (setup_decompression, :193-209).
 */
int oracle_decompress(const uint8_t *src, uint64_t n, uint8_t *out, uint64_t out_len)
{
	uint32_t total, bs;
	uint32_t hdr = oracle_read_header(src, n, &total, &bs);
	if (!hdr)
		return ORACLE_INVALID_INPUT;
	if ((uint64_t)total > out_len)
		return ORACLE_BUFFER_TOO_SMALL;
	out_len = total;
	uint64_t ip = hdr, op = 0;
	while (ip < n) {                                            /* :227 */
		if (ip + 4 > n)
			return ORACLE_INVALID_INPUT;                        /* reference would read past the end */
		uint64_t blk_end = ip + 4 + le32(src + ip);            /* :229-230 */
		ip += 4;
		while (ip != blk_end) {                                 /* :232 */
			if (ip >= n)
				return ORACLE_INVALID_INPUT;                    /* reference would run off the buffer */
			uint8_t tag = src[ip++];
			uint32_t len, off;
			switch (tag & 3) {
			case 0:                                             /* :244-256 */
				len = (tag >> 2) + 1;
				if (len > 60) {
					uint32_t nb = len - 60;                     /* :64-74 */
					if (ip + nb >= n) {
						len = 1;
					} else {
						uint32_t v = 0;
						for (uint32_t k = 0; k < nb; k++)
							v |= (uint32_t)src[ip++] << (8 * k);
						len = (uint16_t)(v + 1);                /* uint16_t length, :233 */
					}
				}
				while (len && ip < n && op < out_len) {        /* :142-154 */
					out[op++] = src[ip++];
					len--;
				}
				continue;
			case 1:                                             /* :264-269, :83-88 */
				len = ((tag >> 2) & 7) + 4;
				off = (ip >= n) ? 0 : ((uint32_t)src[ip++] | (((tag >> 5) & 7u) << 8));
				break;
			case 2:                                             /* :271-276, :97-109 */
				len = (tag >> 2) + 1;
				if (ip + 2 > n) {
					off = 0;
				} else {
					off = (uint32_t)src[ip] | ((uint32_t)src[ip + 1] << 8);
					ip += 2;
				}
				break;
			default:                                            /* :278-283, :118-133 */
				len = (tag >> 2) + 1;
				if (ip + 4 > n) {
					off = 0;
				} else {
					off = le32(src + ip);
					ip += 4;
				}
				break;
			}
			if ((uint64_t)off > op)                             /* :167-173 "bad offset!" */
				return ORACLE_INVALID_INPUT;
			if (off == 0)                                       /* reference copies unwritten bytes; refuse */
				return ORACLE_INVALID_INPUT;
			while (len && op < out_len) {                       /* :174-181 */
				out[op] = out[op - off];
				op++;
				len--;
			}
		}
	}
	return ORACLE_OK;
}

/*
 * Walk the u32 size chain and record where each block's size prefix starts
 * (snappy_decompress.c:317-340, the host pre-scan of the DPU path).
 * Returns number of blocks found, or -1 on a malformed chain.
 */
int64_t oracle_index_blocks(const uint8_t *src, uint64_t n, uint64_t *offsets, uint64_t max_blocks)
{
	uint32_t total, bs;
	uint32_t hdr = oracle_read_header(src, n, &total, &bs);
	if (!hdr || bs == 0)
		return -1;
	uint64_t nblocks = ((uint64_t)total + bs - 1) / bs;          /* :306 */
	uint64_t ip = hdr;
	for (uint64_t i = 0; i < nblocks; i++) {
		if (ip + 4 > n)
			return -1;
		if (i < max_blocks)
			offsets[i] = ip;
		ip += 4 + (uint64_t)le32(src + ip);
	}
	return (ip == n) ? (int64_t)nblocks : -1;
}

/* ---- multi-threaded drivers (CPU baseline on "all cores") ---- */

struct mt_job {
	const uint8_t *src;
	uint64_t n;
	uint32_t block_size;
	uint64_t first_block, last_block;      /* [first, last) */
	uint8_t *slots;                        /* compress: per-block worst-case slots */
	uint64_t slot_stride;
	uint32_t *csize;
	/* decompress */
	const uint64_t *offsets;
	uint8_t *out;
	uint64_t out_len;
	int status;
};

static void *mt_compress_worker(void *arg)
{
	struct mt_job *j = (struct mt_job *)arg;
	uint16_t *table = (uint16_t *)malloc(16384 * sizeof(uint16_t));
	for (uint64_t b = j->first_block; b < j->last_block; b++) {
		uint64_t start = b * j->block_size;
		uint32_t todo = (uint32_t)((j->n - start < j->block_size) ? (j->n - start) : j->block_size);
		j->csize[b] = oracle_compress_block(j->src + start, todo, j->slots + b * j->slot_stride, table);
	}
	free(table);
	return NULL;
}

/*
 * Same bytes as oracle_compress, produced by `nthreads` pthreads over disjoint
 * contiguous block ranges (the partitioning of snappy_compress.c:494-520),
 * then a serial concat.  Returns compressed length (0 on failure).
 */
uint64_t oracle_compress_mt(const uint8_t *src, uint64_t n, uint32_t block_size, uint8_t *dst, uint64_t dst_cap, int nthreads)
{
	if (block_size == 0 || n > 0xffffffffull || nthreads < 1)
		return 0;
	if (dst_cap < oracle_compress_bound(n, block_size))
		return 0;
	uint64_t nblocks = (n + block_size - 1) / block_size;
	uint64_t stride = 4 + oracle_max_compressed_length(block_size);
	uint8_t *slots = (uint8_t *)malloc(nblocks ? nblocks * stride : 1);
	uint32_t *csize = (uint32_t *)malloc((nblocks ? nblocks : 1) * sizeof(uint32_t));
	if (!slots || !csize) {
		free(slots);
		free(csize);
		return 0;
	}
	if (nthreads > 256)
		nthreads = 256;
	pthread_t tid[256];
	unsigned char created[256];
	struct mt_job jobs[256];
	uint64_t per = (nblocks + nthreads - 1) / nthreads;
	int started = 0;
	for (int t = 0; t < nthreads; t++) {
		uint64_t f = (uint64_t)t * per, l = f + per;
		if (f >= nblocks)
			break;
		if (l > nblocks)
			l = nblocks;
		jobs[t] = (struct mt_job){ .src = src, .n = n, .block_size = block_size, .first_block = f, .last_block = l,
			.slots = slots, .slot_stride = stride, .csize = csize };
		/* a thread that cannot be created (thread / pid limits of the container): its range runs here */
		created[t] = pthread_create(&tid[t], NULL, mt_compress_worker, &jobs[t]) == 0;
		if (!created[t])
			mt_compress_worker(&jobs[t]);
		started++;
	}
	for (int t = 0; t < started; t++)
		if (created[t])
			pthread_join(tid[t], NULL);
	uint8_t *op = dst + oracle_write_header(dst, (uint32_t)n, block_size);
	for (uint64_t b = 0; b < nblocks; b++) {
		memcpy(op, slots + b * stride, csize[b]);
		op += csize[b];
	}
	free(slots);
	free(csize);
	return (uint64_t)(op - dst);
}

/* Decode one block strictly inside its own [0, out_len) window; semantics of
 * snappy_decompress.c:232-285 for well-formed streams. */
static int decode_block(const uint8_t *src, uint64_t ip, uint64_t blk_end, uint64_t n, uint8_t *out, uint64_t out_len)
{
	uint64_t op = 0;
	if (blk_end > n)
		return ORACLE_INVALID_INPUT;
	while (ip < blk_end) {
		uint8_t tag = src[ip++];
		uint32_t len, off;
		switch (tag & 3) {
		case 0:
			len = (tag >> 2) + 1;
			if (len > 60) {
				uint32_t nb = len - 60, v = 0;
				if (ip + nb > blk_end)
					return ORACLE_INVALID_INPUT;
				for (uint32_t k = 0; k < nb; k++)
					v |= (uint32_t)src[ip++] << (8 * k);
				len = v + 1;
			}
			if (ip + len > blk_end || op + len > out_len)
				return ORACLE_INVALID_INPUT;
			memcpy(out + op, src + ip, len);
			ip += len;
			op += len;
			continue;
		case 1:
			if (ip + 1 > blk_end)
				return ORACLE_INVALID_INPUT;
			len = ((tag >> 2) & 7) + 4;
			off = (uint32_t)src[ip++] | (((tag >> 5) & 7u) << 8);
			break;
		case 2:
			if (ip + 2 > blk_end)
				return ORACLE_INVALID_INPUT;
			len = (tag >> 2) + 1;
			off = (uint32_t)src[ip] | ((uint32_t)src[ip + 1] << 8);
			ip += 2;
			break;
		default:
			if (ip + 4 > blk_end)
				return ORACLE_INVALID_INPUT;
			len = (tag >> 2) + 1;
			off = le32(src + ip);
			ip += 4;
			break;
		}
		if (off == 0 || off > op || op + len > out_len)
			return ORACLE_INVALID_INPUT;
		for (uint32_t k = 0; k < len; k++, op++)
			out[op] = out[op - off];
	}
	return (op == out_len) ? ORACLE_OK : ORACLE_INVALID_INPUT;
}

static void *mt_decompress_worker(void *arg)
{
	struct mt_job *j = (struct mt_job *)arg;
	j->status = ORACLE_OK;
	for (uint64_t b = j->first_block; b < j->last_block; b++) {
		uint64_t at = j->offsets[b];
		uint64_t ostart = b * j->block_size;
		uint64_t olen = (j->out_len - ostart < j->block_size) ? (j->out_len - ostart) : j->block_size;
		int st = decode_block(j->src, at + 4, at + 4 + le32(j->src + at), j->n, j->out + ostart, olen);
		if (st != ORACLE_OK)
			j->status = st;
	}
	return NULL;
}

/*
 * Block-parallel decompress: host pre-scan of the size chain
 * (snappy_decompress.c:317-340) then `nthreads` pthreads, each decoding a
 * contiguous block range into out + i*block_size (snappy_decompress.c:330).
 * Strict per block; identical output to oracle_decompress on valid streams.
 */
int oracle_decompress_mt(const uint8_t *src, uint64_t n, uint8_t *out, uint64_t out_cap, int nthreads)
{
	uint32_t total, bs;
	uint32_t hdr = oracle_read_header(src, n, &total, &bs);
	if (!hdr || bs == 0 || nthreads < 1)
		return ORACLE_INVALID_INPUT;
	if ((uint64_t)total > out_cap)
		return ORACLE_BUFFER_TOO_SMALL;
	uint64_t nblocks = ((uint64_t)total + bs - 1) / bs;
	uint64_t *offsets = (uint64_t *)malloc((nblocks ? nblocks : 1) * sizeof(uint64_t));
	if (oracle_index_blocks(src, n, offsets, nblocks) != (int64_t)nblocks) {
		free(offsets);
		return ORACLE_INVALID_INPUT;
	}
	if (nthreads > 256)
		nthreads = 256;
	pthread_t tid[256];
	unsigned char created[256];
	struct mt_job jobs[256];
	uint64_t per = (nblocks + nthreads - 1) / nthreads;
	int started = 0;
	for (int t = 0; t < nthreads; t++) {
		uint64_t f = (uint64_t)t * per, l = f + per;
		if (f >= nblocks)
			break;
		if (l > nblocks)
			l = nblocks;
		jobs[t] = (struct mt_job){ .src = src, .n = n, .block_size = bs, .first_block = f, .last_block = l,
			.offsets = offsets, .out = out, .out_len = total };
		created[t] = pthread_create(&tid[t], NULL, mt_decompress_worker, &jobs[t]) == 0;
		if (!created[t])
			mt_decompress_worker(&jobs[t]);
		started++;
	}
	int status = ORACLE_OK;
	for (int t = 0; t < started; t++) {
		if (created[t])
			pthread_join(tid[t], NULL);
		if (jobs[t].status != ORACLE_OK)
			status = jobs[t].status;
	}
	free(offsets);
	return status;
}

/* ---- all-cores baseline with a persistent, pre-faulted workspace (SURVEY 8d "CPU baseline") ----
 *
 * oracle_compress_mt / oracle_decompress_mt above allocate their slots per call and concatenate serially, which is
 * fine for a checker but times page faults and one core's memcpy when used as "the all-cores CPU number".  The
 * context below is what bench.py's cpu_baseline leg times: every buffer is allocated and touched once in
 * oracle_mt_create, a call is one pthread launch in which each thread (1) runs the per-block routine of
 * snappy_compress.c:284-413 over its contiguous block range (the partitioning of snappy_compress.c:494-520) into its own
 * region, (2) meets the others at a barrier where the region offsets are summed, (3) copies its own region to its place
 * in the stream -- a parallel concat.  Same bytes as oracle_compress.
 */
struct oracle_mt_ctx {
	uint64_t max_n;
	uint32_t block_size;
	int nthreads;
	uint64_t nblocks_max, per;            /* blocks per thread */
	uint64_t region_stride;               /* bytes reserved per thread */
	uint8_t *regions;                     /* nthreads * region_stride */
	uint64_t *region_len;                 /* bytes each thread produced */
	uint64_t *region_at;                  /* where each thread's bytes go in the stream */
	uint64_t *offsets;                    /* decompress: block offsets */
};

struct mt2_job {
	struct oracle_mt_ctx *c;
	int t;
	const uint8_t *src;
	uint64_t n;
	uint8_t *dst;
	uint64_t hdr_len, nblocks;
	uint8_t *out;
	uint64_t out_len;
	int status;
};

void *oracle_mt_create(uint64_t max_n, uint32_t block_size, int nthreads)
{
	if (block_size == 0 || nthreads < 1 || max_n > 0xffffffffull)
		return NULL;
	if (nthreads > 256)
		nthreads = 256;
	struct oracle_mt_ctx *c = (struct oracle_mt_ctx *)calloc(1, sizeof(*c));
	if (!c)
		return NULL;
	c->max_n = max_n;
	c->block_size = block_size;
	c->nthreads = nthreads;
	c->nblocks_max = (max_n + block_size - 1) / block_size;
	c->per = (c->nblocks_max + nthreads - 1) / nthreads;
	c->region_stride = c->per * (4 + oracle_max_compressed_length(block_size)) + 64;
	c->regions = (uint8_t *)malloc((uint64_t)nthreads * c->region_stride);
	c->region_len = (uint64_t *)calloc(nthreads, sizeof(uint64_t));
	c->region_at = (uint64_t *)calloc(nthreads, sizeof(uint64_t));
	c->offsets = (uint64_t *)malloc((c->nblocks_max + 1) * sizeof(uint64_t));
	if (!c->regions || !c->region_len || !c->region_at || !c->offsets) {
		free(c->regions);
		free(c->region_len);
		free(c->region_at);
		free(c->offsets);
		free(c);
		return NULL;
	}
	memset(c->regions, 0x5a, (uint64_t)nthreads * c->region_stride);     /* pre-fault */
	memset(c->offsets, 0, (c->nblocks_max + 1) * sizeof(uint64_t));
	return c;
}

void oracle_mt_destroy(void *ctx)
{
	struct oracle_mt_ctx *c = (struct oracle_mt_ctx *)ctx;
	if (!c)
		return;
	free(c->regions);
	free(c->region_len);
	free(c->region_at);
	free(c->offsets);
	free(c);
}

/* Runs fn over jobs[0..count): one pthread each; a job whose thread cannot be created (thread / pid limits of the
 * container) runs on the calling thread instead, so a call never hangs or loses work.  No barriers: phases are separate
 * launches. */
static void mt2_run(struct mt2_job *jobs, int count, void *(*fn)(void *))
{
	pthread_t tid[256];
	unsigned char started[256];
	for (int t = 0; t < count; t++)
		started[t] = pthread_create(&tid[t], NULL, fn, &jobs[t]) == 0;
	for (int t = 0; t < count; t++)
		if (!started[t])
			fn(&jobs[t]);
	for (int t = 0; t < count; t++)
		if (started[t])
			pthread_join(tid[t], NULL);
}

/* phase 1: the thread's contiguous block range into its own region */
static void *mt2_compress_worker(void *arg)
{
	struct mt2_job *j = (struct mt2_job *)arg;
	struct oracle_mt_ctx *c = j->c;
	uint16_t table[16384];
	uint8_t *region = c->regions + (uint64_t)j->t * c->region_stride;
	uint64_t f = (uint64_t)j->t * c->per, l = f + c->per;
	if (l > j->nblocks)
		l = j->nblocks;
	uint64_t put = 0;
	for (uint64_t b = f; b < l; b++) {
		uint64_t start = b * c->block_size;
		uint32_t todo = (uint32_t)((j->n - start < c->block_size) ? (j->n - start) : c->block_size);
		put += oracle_compress_block(j->src + start, todo, region + put, table);
	}
	c->region_len[j->t] = put;
	return NULL;
}

/* phase 2: the parallel concat (region_at is the prefix sum of the region lengths) */
static void *mt2_concat_worker(void *arg)
{
	struct mt2_job *j = (struct mt2_job *)arg;
	struct oracle_mt_ctx *c = j->c;
	memcpy(j->dst + c->region_at[j->t], c->regions + (uint64_t)j->t * c->region_stride, c->region_len[j->t]);
	return NULL;
}

/* returns the stream length, 0 on failure; dst_cap >= oracle_compress_bound(n, block_size) */
uint64_t oracle_mt_compress(void *ctx, const uint8_t *src, uint64_t n, uint8_t *dst, uint64_t dst_cap)
{
	struct oracle_mt_ctx *c = (struct oracle_mt_ctx *)ctx;
	if (!c || n > c->max_n || dst_cap < oracle_compress_bound(n, c->block_size))
		return 0;
	uint64_t nblocks = (n + c->block_size - 1) / c->block_size;
	uint64_t hdr = oracle_write_header(dst, (uint32_t)n, c->block_size);
	struct mt2_job jobs[256];
	for (int t = 0; t < c->nthreads; t++)
		jobs[t] = (struct mt2_job){ .c = c, .t = t, .src = src, .n = n, .dst = dst, .hdr_len = hdr, .nblocks = nblocks };
	mt2_run(jobs, c->nthreads, mt2_compress_worker);
	uint64_t at = hdr;
	for (int t = 0; t < c->nthreads; t++) {
		c->region_at[t] = at;
		at += c->region_len[t];
	}
	mt2_run(jobs, c->nthreads, mt2_concat_worker);
	return at;
}

static void *mt2_decompress_worker(void *arg)
{
	struct mt2_job *j = (struct mt2_job *)arg;
	struct oracle_mt_ctx *c = j->c;
	uint64_t f = (uint64_t)j->t * c->per, l = f + c->per;
	if (l > j->nblocks)
		l = j->nblocks;
	j->status = ORACLE_OK;
	for (uint64_t b = f; b < l; b++) {
		uint64_t at = c->offsets[b];
		uint64_t ostart = b * c->block_size;
		uint64_t olen = (j->out_len - ostart < c->block_size) ? (j->out_len - ostart) : c->block_size;
		int st = decode_block(j->src, at + 4, at + 4 + le32(j->src + at), j->n, j->out + ostart, olen);
		if (st != ORACLE_OK)
			j->status = st;
	}
	return NULL;
}

/* the host pre-scan of the size chain (snappy_decompress.c:317-340) stays serial and inside the call, as in the reference */
int oracle_mt_decompress(void *ctx, const uint8_t *src, uint64_t n, uint8_t *out, uint64_t out_cap)
{
	struct oracle_mt_ctx *c = (struct oracle_mt_ctx *)ctx;
	uint32_t total, bs;
	uint32_t hdr = c ? oracle_read_header(src, n, &total, &bs) : 0;
	if (!hdr || bs != c->block_size || (uint64_t)total > c->max_n)
		return ORACLE_INVALID_INPUT;
	if ((uint64_t)total > out_cap)
		return ORACLE_BUFFER_TOO_SMALL;
	uint64_t nblocks = ((uint64_t)total + bs - 1) / bs;
	if (oracle_index_blocks(src, n, c->offsets, nblocks) != (int64_t)nblocks)
		return ORACLE_INVALID_INPUT;
	struct mt2_job jobs[256];
	for (int t = 0; t < c->nthreads; t++)
		jobs[t] = (struct mt2_job){ .c = c, .t = t, .src = src, .n = n, .nblocks = nblocks, .out = out, .out_len = total };
	mt2_run(jobs, c->nthreads, mt2_decompress_worker);
	int status = ORACLE_OK;
	for (int t = 0; t < c->nthreads; t++)
		if (jobs[t].status != ORACLE_OK)
			status = jobs[t].status;
	return status;
}
