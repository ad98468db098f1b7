/*
 * dpu_snappy -- the reference's command line tool (snappy/dpu_snappy.c) re-authored for MI355X:
 * identical flags and stdout lines, `-d` now means "offload to the GPU(s)" through libsnappy_hip.so
 * (snappy_compress_gpu / snappy_decompress_gpu) where the reference called snappy_compress_dpu /
 * snappy_decompress_dpu (dpu_snappy.c:169-172, :189-192).  Without -d the host CPU codec runs,
 * as in the reference.  -d never falls back to the CPU.
 *
 *   dpu_snappy [-d] [-c] [-b <block_size>] [-g <gpus>] -i <input_file> [-o <output_file>]
 */
#include <getopt.h>
#include <limits.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/time.h>

#include "snappy_host.h"

static void usage(const char *exe)                          /* dpu_snappy.c:70-82 */
{
#ifdef DEBUG
	fprintf(stderr, "**DEBUG BUILD**\n");
#endif
	fprintf(stderr, "Compress or decompress a file with Snappy\nCan use either the host CPU or MI355X GPUs\n");
	fprintf(stderr, "usage: %s [-d] [-c] [-b <block_size>] [-g <gpus>] -i <input_file> [-o <output_file>]\n", exe);
	fprintf(stderr, "d: use the GPU(s), by default host is used\n");
	fprintf(stderr, "c: perform compression, by default performs decompression\n");
	fprintf(stderr, "b: block size used for compression, default is 32KB, ignored for decompression\n");
	fprintf(stderr, "g: number of GPUs to shard blocks over with -d, default all visible\n");
	fprintf(stderr, "i: input file\n");
	fprintf(stderr, "o: output file\n");
}

/* In -d mode the file buffers are page-locked (snappy_hip_host_alloc) so that the copies inside
 * snappy_*_gpu run at PCIe rate instead of through the driver's pageable bounce buffers. */
static int g_pinned = 0;

static uint8_t *buffer_alloc(unsigned long bytes)
{
	if (g_pinned) {
		uint8_t *p = snappy_hip_host_alloc(bytes);
		if (p)
			return p;
		g_pinned = 0;                /* no device: -d will fail loudly later; keep going with malloc */
	}
	return malloc(bytes ? bytes : 8);
}

static int slurp(const char *path, struct host_buffer_context *in)    /* dpu_snappy.c:23-50 */
{
	FILE *f = fopen(path, "rb");
	if (!f) {
		fprintf(stderr, "Invalid input file: %s\n", path);
		return 1;
	}
	fseek(f, 0, SEEK_END);
	long sz = ftell(f);
	rewind(f);
	if (sz < 0 || (unsigned long)sz > in->max) {
		fprintf(stderr, "input_size is too big (%ld > %ld)\n", sz, in->max);
		fclose(f);
		return 1;
	}
	in->length = (unsigned long)sz;
	in->buffer = buffer_alloc(((unsigned long)sz + 7) & ~7UL);
	in->curr = in->buffer;
	size_t got = fread(in->buffer, 1, in->length, f);
	fclose(f);
#ifdef DEBUG
	printf("%s: read %ld bytes from %s (%lu)\n", __func__, in->length, path, got);
#endif
	return got != in->length;
}

static int spill(const char *path, const struct host_buffer_context *out)   /* dpu_snappy.c:58-63 */
{
	FILE *f = fopen(path, "wb");
	if (!f) {
		fprintf(stderr, "Cannot open output file: %s\n", path);
		return 1;
	}
	size_t put = fwrite(out->buffer, 1, out->length, f);
	fclose(f);
	return put != out->length;
}

int main(int argc, char **argv)
{
	int use_gpu = 0, compress = 0, opt;
	int block_size = 32 * 1024;                              /* dpu_snappy.c:100 */
	const char *in_path = NULL, *out_path = NULL;
	struct host_buffer_context input = { 0 }, output = { 0 };
	input.max = ULONG_MAX;
	output.max = ULONG_MAX;

	while ((opt = getopt(argc, argv, "dcb:g:i:o:")) != -1) {
		switch (opt) {
		case 'd': use_gpu = 1; break;
		case 'c': compress = 1; break;
		case 'b': block_size = atoi(optarg); break;
		case 'g': setenv("SNAPPY_HIP_NUM_GPUS", optarg, 1); break;
		case 'i': in_path = optarg; break;
		case 'o': out_path = optarg; break;
		default:
			usage(argv[0]);
			return -2;
		}
	}
	if (!in_path) {
		usage(argv[0]);
		return -1;
	}
	if (use_gpu) {
		/* the overlapped copy-in / kernel / copy-out pipeline of the library keeps six HIP streams busy; HIP maps
		 * streams onto GPU_MAX_HW_QUEUES hardware queues (default 4).  The host program owns its environment, so it
		 * is set here (before the first HIP call), not by the library. */
		setenv("GPU_MAX_HW_QUEUES", "8", 0);
		/* the format's length field is a uint32 (snappy_compress.c:461); HBM is not the limit */
		input.max = 0xffffffffUL;
		output.max = 0xffffffffUL;
	}
	input.file_name = in_path;
	printf("Using input file %s\n", in_path);
	if (!out_path)
		out_path = "output.txt";                             /* dpu_snappy.c:155-157 */
	output.file_name = out_path;
	printf("Using output file %s\n", out_path);

	g_pinned = use_gpu;
	if (slurp(in_path, &input))
		return -1;

	struct program_runtime rt;
	memset(&rt, 0, sizeof(rt));                              /* the reference leaves this uninitialised */
	snappy_status st;
	struct timeval t0, t1;
	if (compress) {
		if (use_gpu && g_pinned && block_size >= 1 && block_size <= 65535) {
			/* page-locked output of the stream's upper bound; max = capacity tells the library not to realloc */
			struct timeval a, b;
			gettimeofday(&a, NULL);
			output.max = snappy_hip_stream_bound(input.length, (uint32_t)block_size);
			output.buffer = buffer_alloc(output.max);
			output.curr = output.buffer;
			gettimeofday(&b, NULL);
			rt.pre = get_runtime(&a, &b);
		} else if (use_gpu) {
			/* no page-locked memory (or an out-of-range -b, which the library rejects): the library allocates the
			 * output itself; setup_compression's 32+n+n/6 is too small for tiny block sizes (snappy_compress.c:446-449) */
			output.buffer = NULL;
			output.curr = NULL;
			output.max = ULONG_MAX;
		} else {
			setup_compression(&input, &output, &rt);
		}
		if (use_gpu) {
			st = snappy_compress_gpu(&input, &output, (uint32_t)block_size, &rt);
		} else {
			gettimeofday(&t0, NULL);
			st = snappy_compress_host(&input, &output, (uint32_t)block_size);
			gettimeofday(&t1, NULL);
			rt.run = get_runtime(&t0, &t1);
		}
	} else {
		if (setup_decompression(&input, &output, &rt))
			return -1;
		if (use_gpu && g_pinned) {           /* swap the malloc'd plaintext buffer for a page-locked one */
			free(output.buffer);
			output.buffer = buffer_alloc((output.length + 7) & ~7UL);
			output.curr = output.buffer;
		}
		if (use_gpu) {
			st = snappy_decompress_gpu(&input, &output, &rt);
		} else {
			gettimeofday(&t0, NULL);
			st = snappy_decompress_host(&input, &output);
			gettimeofday(&t1, NULL);
			rt.run = get_runtime(&t0, &t1);
		}
	}

	if (st != SNAPPY_OK) {
		fprintf(stderr, "Encountered Snappy error %u\n", st);   /* dpu_snappy.c:229-233 */
		return -1;
	}
	/* unlike snappy_compress_dpu (snappy_compress.c:626-627) the GPU path hands the stream back in
	 * output.buffer, so main writes the file in every mode */
	if (spill(out_path, &output))
		return -1;

	if (compress) {
		printf("Compressed %ld bytes to: %s\n", output.length, out_path);
		printf("Compression ratio: %f\n", 1 - (double)output.length / (double)input.length);
	} else {
		printf("Decompressed %ld bytes to: %s\n", output.length, out_path);
		printf("Compression ratio: %f\n", 1 - (double)input.length / (double)output.length);
	}
	printf("Pre-processing time: %f\n", rt.pre);
	printf("Alloc time: %f\n", rt.d_alloc);
	printf("Load time: %f\n", rt.load);
	printf("Copy in time: %f\n", rt.copy_in);
	printf("Host time: %f\n", rt.run);
	printf("Copy out time: %f\n", rt.copy_out);
	printf("Free time: %f\n", rt.d_free);
	return 0;
}
