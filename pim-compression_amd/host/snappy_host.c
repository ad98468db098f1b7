/*
 * snappy_host.c -- CPU mode of the dpu_snappy CLI (the tool's default when -d is absent, as in
 * the reference: snappy/dpu_snappy.c:173-183, :193-203).  Produces / consumes exactly the
 * reference's block-framed format:  varint(U) varint(BS) { u32le(size) elements }*
 * (snappy/README.md:19-33).  Behaviour follows snappy/snappy_compress.c:284-485 and
 * snappy/snappy_decompress.c:187-289; the decoder is stricter on malformed input.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/time.h>

#include "snappy_host.h"

#define TABLE_MAX 16384u          /* snappy_compress.c:16-17 */
#define HASH_MULT 0x1e35a7bdu     /* snappy_compress.c:163 */
#define TAIL_MARGIN 15u           /* snappy_compress.c:299 */

double get_runtime(struct timeval *start, struct timeval *end)   /* dpu_snappy.c:87-91 */
{
	return ((double)end->tv_sec - (double)start->tv_sec) + ((double)end->tv_usec - (double)start->tv_usec) / 1e6;
}

static uint32_t load32(const uint8_t *p)
{
	uint32_t v;
	memcpy(&v, p, 4);          /* little-endian hosts only, like the GPU */
	return v;
}

static uint8_t *varint_put(uint8_t *p, uint32_t v)
{
	for (; v > 0x7f; v >>= 7)
		*p++ = (uint8_t)(v | 0x80);
	*p++ = (uint8_t)v;
	return p;
}

static const uint8_t *varint_get(const uint8_t *p, const uint8_t *end, uint32_t *v)
{
	uint32_t acc = 0;
	for (int k = 0; k < 5 && p < end; k++) {
		uint8_t c = *p++;
		acc |= (uint32_t)(c & 0x7f) << (7 * k);
		if (c < 0x80) {
			*v = acc;
			return p;
		}
	}
	return NULL;
}

/* ---- compress ----------------------------------------------------------- */

struct sink {
	uint8_t *p;
};

static void put_literal(struct sink *s, const uint8_t *src, uint32_t len)   /* snappy_compress.c:202-225 */
{
	uint32_t n = len - 1;
	if (n < 60) {
		*s->p++ = (uint8_t)(n << 2);
	} else {
		uint8_t *tag = s->p++;
		uint32_t extra = 0;
		for (; n; n >>= 8, extra++)
			*s->p++ = (uint8_t)n;
		*tag = (uint8_t)((59 + extra) << 2);
	}
	memcpy(s->p, src, len);
	s->p += len;
}

static void put_copy(struct sink *s, uint32_t off, uint32_t len)           /* snappy_compress.c:234-272 */
{
	for (;;) {
		uint32_t piece = len;
		if (len >= 68)
			piece = 64;
		else if (len > 64)
			piece = 60;
		if (piece < 12 && off < 2048) {
			*s->p++ = (uint8_t)(1u | ((piece - 4) << 2) | ((off >> 8) << 5));
			*s->p++ = (uint8_t)off;
		} else {
			*s->p++ = (uint8_t)(2u | ((piece - 1) << 2));
			*s->p++ = (uint8_t)off;
			*s->p++ = (uint8_t)(off >> 8);
		}
		len -= piece;
		if (!len)
			return;
	}
}

static void host_compress_block(const uint8_t *b, uint32_t n, struct sink *s, uint16_t *tab)
{
	uint32_t entries = 256;                                /* snappy_compress.c:139-146 */
	while (entries < TABLE_MAX && entries < n)
		entries <<= 1;
	memset(tab, 0, entries * sizeof(*tab));
	const int shift = __builtin_clz(entries) + 1;          /* :288 */
	uint8_t *size_at = s->p;
	s->p += 4;                                             /* :291 */
	uint32_t lit = 0;                                      /* start of pending literal */

	if (n >= TAIL_MARGIN) {
		const uint32_t last = n - TAIL_MARGIN;
		uint32_t pos = 1, hcur = (load32(b + 1) * HASH_MULT) >> shift;
		for (;;) {
			uint32_t tries = 32, probe = pos, cand;
			for (;;) {                                     /* :336-348 */
				pos = probe;
				uint32_t h = hcur;
				probe = pos + (tries++ >> 5);
				if (probe > last)
					goto tail;
				hcur = (load32(b + probe) * HASH_MULT) >> shift;
				cand = tab[h];
				tab[h] = (uint16_t)pos;
				if (load32(b + pos) == load32(b + cand))
					break;
			}
			put_literal(s, b + lit, pos - lit);            /* :355 */
			for (;;) {                                     /* :370-398 */
				uint32_t from = pos, a = cand + 4, m = 4;
				pos += 4;
				while (pos + 4 <= n && load32(b + pos) == load32(b + a)) {
					pos += 4; a += 4; m += 4;
				}
				while (pos < n && b[pos] == b[a]) {
					pos++; a++; m++;
				}
				put_copy(s, from - cand, m);
				lit = pos;
				if (pos >= last)
					goto tail;
				tab[(load32(b + pos - 1) * HASH_MULT) >> shift] = (uint16_t)(pos - 1);
				uint32_t h = (load32(b + pos) * HASH_MULT) >> shift;
				cand = tab[h];
				tab[h] = (uint16_t)pos;
				if (load32(b + pos) != load32(b + cand))
					break;
			}
			pos++;                                         /* :400-401 */
			hcur = (load32(b + pos) * HASH_MULT) >> shift;
		}
	}
tail:
	if (lit < n)
		put_literal(s, b + lit, n - lit);                  /* :405-410 */
	uint32_t sz = (uint32_t)(s->p - size_at - 4);
	size_at[0] = (uint8_t)sz; size_at[1] = (uint8_t)(sz >> 8); size_at[2] = (uint8_t)(sz >> 16); size_at[3] = (uint8_t)(sz >> 24);
}

void setup_compression(struct host_buffer_context *input, struct host_buffer_context *output, struct program_runtime *runtime)
{
	struct timeval t0, t1;
	gettimeofday(&t0, NULL);
	/* The reference reserves 32 + n + n/6 (snappy_compress.c:446-447), which cannot hold the 4-byte
	 * prefixes of very small blocks; reserve for the smallest block size the CLI accepts (64) too. */
	unsigned long n = input->length;
	unsigned long cap = 64 + n + n / 6 + (n / 64 + 1) * 8;
	output->buffer = malloc(cap);
	output->curr = output->buffer;
	output->length = 0;
	gettimeofday(&t1, NULL);
	runtime->pre = get_runtime(&t0, &t1);
}

snappy_status snappy_compress_host(struct host_buffer_context *input, struct host_buffer_context *output, uint32_t block_size)
{
	if (block_size < 64 || block_size > 65535 || input->length > 0xffffffffUL)
		return SNAPPY_INVALID_INPUT;
	uint16_t *tab = malloc(TABLE_MAX * sizeof(*tab));
	struct sink s = { output->buffer };
	s.p = varint_put(s.p, (uint32_t)input->length);       /* :461-465 */
	s.p = varint_put(s.p, block_size);
	const uint8_t *in = input->buffer;
	unsigned long left = input->length;
	while (left) {                                        /* :467-479 */
		uint32_t n = left < block_size ? (uint32_t)left : block_size;
		host_compress_block(in, n, &s, tab);
		in += n;
		left -= n;
	}
	free(tab);
	input->curr = input->buffer + input->length;
	output->curr = s.p;
	output->length = (unsigned long)(s.p - output->buffer);
	return SNAPPY_OK;
}

/* ---- decompress ---------------------------------------------------------- */

snappy_status setup_decompression(struct host_buffer_context *input, struct host_buffer_context *output, struct program_runtime *runtime)
{
	struct timeval t0, t1;
	gettimeofday(&t0, NULL);
	uint32_t total;
	const uint8_t *p = varint_get(input->curr, input->buffer + input->length, &total);   /* :193-198 */
	if (!p) {
		fprintf(stderr, "Failed to read decompressed length\n");
		return SNAPPY_INVALID_INPUT;
	}
	input->curr = (uint8_t *)p;
	if (total > output->max) {                            /* :200-204 */
		fprintf(stderr, "Output length is to big: max=%ld len=%d\n", output->max, total);
		return SNAPPY_BUFFER_TOO_SMALL;
	}
	output->buffer = malloc((((unsigned long)total + 7) & ~7UL) | 2047);   /* :207 */
	output->curr = output->buffer;
	output->length = total;
	gettimeofday(&t1, NULL);
	runtime->pre = get_runtime(&t0, &t1);
	return SNAPPY_OK;
}

snappy_status snappy_decompress_host(struct host_buffer_context *input, struct host_buffer_context *output)
{
	const uint8_t *end = input->buffer + input->length;
	uint32_t bs;
	const uint8_t *ip = varint_get(input->curr, end, &bs);   /* :220-225 */
	if (!ip) {
		fprintf(stderr, "Failed to read decompressed block size\n");
		return SNAPPY_INVALID_INPUT;
	}
	uint8_t *const out0 = output->buffer;
	uint8_t *const out_end = out0 + output->length;
	uint8_t *op = out0;
	while (ip < end) {                                    /* :227-231 */
		if (end - ip < 4)
			return SNAPPY_INVALID_INPUT;
		uint32_t csz = load32(ip);
		ip += 4;
		if ((unsigned long)(end - ip) < csz)
			return SNAPPY_INVALID_INPUT;
		const uint8_t *bend = ip + csz;
		while (ip < bend) {                               /* :232-285 */
			uint32_t tag = *ip++, len, off;
			if ((tag & 3) == 0) {
				len = (tag >> 2) + 1;
				if (len > 60) {
					uint32_t nb = len - 60;
					if ((uint32_t)(bend - ip) < nb)
						return SNAPPY_INVALID_INPUT;
					len = 0;
					for (uint32_t k = 0; k < nb; k++)
						len |= (uint32_t)ip[k] << (8 * k);
					len += 1;
					ip += nb;
				}
				if (len == 0 || (unsigned long)(bend - ip) < len || (unsigned long)(out_end - op) < len)
					return SNAPPY_INVALID_INPUT;
				memcpy(op, ip, len);
				ip += len;
				op += len;
				continue;
			}
			uint32_t need = (tag & 3) == 1 ? 1 : ((tag & 3) == 2 ? 2 : 4);
			if ((uint32_t)(bend - ip) < need)
				return SNAPPY_INVALID_INPUT;
			if ((tag & 3) == 1) {
				len = ((tag >> 2) & 7) + 4;
				off = ((tag >> 5) << 8) | ip[0];
			} else if ((tag & 3) == 2) {
				len = (tag >> 2) + 1;
				off = ip[0] | ((uint32_t)ip[1] << 8);
			} else {
				len = (tag >> 2) + 1;
				off = load32(ip);
			}
			ip += need;
			if (off == 0 || (unsigned long)(op - out0) < off) {
				printf("bad offset!\n");                    /* :171 */
				return SNAPPY_INVALID_INPUT;
			}
			if ((unsigned long)(out_end - op) < len)
				return SNAPPY_INVALID_INPUT;
			for (const uint8_t *from = op - off; len; len--)
				*op++ = *from++;
		}
	}
	input->curr = (uint8_t *)ip;
	output->curr = op;
	return (op == out_end) ? SNAPPY_OK : SNAPPY_INVALID_INPUT;
}
