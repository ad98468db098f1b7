/*
 * snappy_hip.h -- C ABI of libsnappy_hip.so, the MI355X (gfx950) drop-in for the
 * UPMEM-DPU offload path of UBC-ECE-Sasha/PIM-compression's `dpu_snappy`.
 *
 * Two layers are exported:
 *
 *  1. The drop-in pair, with the exact shape of the reference's L2 entry points
 *       snappy_compress_dpu    (reference snappy/snappy_compress.h:37,  snappy_compress.c:487)
 *       snappy_decompress_dpu  (reference snappy/snappy_decompress.h:34, snappy_decompress.c:292)
 *     They take the reference's own `struct host_buffer_context` /
 *     `struct program_runtime` (reference snappy/dpu_snappy.h:37-55) and return its
 *     `snappy_status` (dpu_snappy.h:21-25).  `main` in dpu_snappy.c:169-172 / :189-192
 *     calls them where it called the *_dpu functions.
 *
 *  2. A resident API over device pointers (what the drop-in pair is built from, and what
 *     bench.py / the tests drive): per-block compress into worst-case slots, scan+compact
 *     into the framed stream, size-chain indexing, per-block decompress.  It replaces the
 *     dpu_alloc / dpu_push_xfer / dpu_launch plumbing (snappy_compress.c:535-618,
 *     snappy_decompress.c:351-439) with hipMalloc / hipMemcpy / kernel launches.
 *
 * Plain C: pointers and sizes only.  No CPU fallback exists behind any entry point: if no
 * HIP device / code object is usable they fail with SNAPPY_HIP_ERR_* (resident API) or
 * SNAPPY_INVALID_INPUT (drop-in pair, as the reference maps a failed dpu_launch,
 * snappy_compress.c:618-623) and say why on stderr / via snappy_hip_last_error().
 */
#ifndef SNAPPY_HIP_H_
#define SNAPPY_HIP_H_

#include <stdint.h>
#include <stddef.h>

/* The library is built with -fvisibility=hidden: the functions declared in this header are its whole dynamic symbol table. */
#if defined(__GNUC__)
#define SNAPPY_HIP_API __attribute__((visibility("default")))
#else
#define SNAPPY_HIP_API
#endif

#ifdef __cplusplus
extern "C" {
#endif

/* ---- types shared with the reference (layout-identical; skipped when the
 *      reference's own dpu_snappy.h was included first) ------------------- */
#ifndef _DPU_SNAPPY_H_
typedef enum {
	SNAPPY_OK = 0,
	SNAPPY_INVALID_INPUT,
	SNAPPY_BUFFER_TOO_SMALL
} snappy_status;                        /* dpu_snappy.h:21-25 */

typedef struct host_buffer_context {
	const char *file_name;
	uint8_t *buffer;
	uint8_t *curr;
	unsigned long length;
	unsigned long max;
} host_buffer_context;                  /* dpu_snappy.h:37-44 */

struct program_runtime {
	double pre;
	double d_alloc;
	double load;
	double copy_in;
	double run;
	double copy_out;
	double d_free;
};                                      /* dpu_snappy.h:47-55 */
#endif

/* ---- 1. drop-in pair ----------------------------------------------------- */

/*
 * Replaces snappy_compress_dpu (snappy_compress.c:487-714).
 * Entry: input->buffer/curr at file start, input->length = n.  output->buffer may be NULL
 * or any malloc'd block; it is realloc'd to the needed size (the reference's
 * 32+n+n/6, snappy_compress.c:446-447, is too small for tiny block sizes).  If the caller
 * sets output->max to a finite capacity (anything but ULONG_MAX, the reference's default,
 * dpu_snappy.c:112), output->buffer is used as is -- e.g. a page-locked buffer -- and
 * SNAPPY_BUFFER_TOO_SMALL is returned if the stream does not fit.
 * Exit: framed stream in output->buffer[0..output->length); caller writes the file.
 * Fills every field of *runtime (pre is accumulated with +=, as snappy_compress.c:528).
 * Uses SNAPPY_HIP_NUM_GPUS devices (env, default: all visible), contiguous block ranges
 * per device (snappy_compress.c:494-520), host-side concat of per-device outputs.
 */
SNAPPY_HIP_API snappy_status snappy_compress_gpu(struct host_buffer_context *input, struct host_buffer_context *output,
                                  uint32_t block_size, struct program_runtime *runtime);

/*
 * Replaces snappy_decompress_dpu (snappy_decompress.c:292-493).
 * Entry (as left by setup_decompression, snappy_decompress.c:187-215): input->curr just
 * past the first varint; output->buffer malloc'd, output->length = uncompressed length.
 * Reads the block-size varint itself (snappy_decompress.c:300), walks the u32 size chain on
 * the host (:317-340), decodes block i into output->buffer + i*block_size (:330).
 * Stricter than the host decoder: a block that overruns its compressed size, its output
 * window, or references bytes before its own start yields SNAPPY_INVALID_INPUT.
 */
SNAPPY_HIP_API snappy_status snappy_decompress_gpu(struct host_buffer_context *input, struct host_buffer_context *output,
                                    struct program_runtime *runtime);

/* ---- 2. resident API ----------------------------------------------------- */

#define SNAPPY_HIP_OK            0
#define SNAPPY_HIP_ERR_NO_DEVICE 1   /* no usable HIP device / runtime */
#define SNAPPY_HIP_ERR_ARG       2   /* bad argument (alignment, sizes, null) */
#define SNAPPY_HIP_ERR_RUNTIME   3   /* a HIP call failed; see snappy_hip_last_error() */

/* per-block status written by the decompress kernel */
#define SNAPPY_HIP_BLOCK_OK        0u
#define SNAPPY_HIP_BLOCK_INVALID   1u

#define SNAPPY_HIP_MIN_BLOCK_SIZE  1u
#define SNAPPY_HIP_MAX_BLOCK_SIZE  65535u   /* u16 hash table entries, snappy_compress.c:347 */

/* Description of one framed stream for snappy_hip_index_streams. */
typedef struct snappy_hip_stream_desc {
	const uint8_t *stream;      /* device: start of the framed stream (its header)      */
	uint64_t stream_len;        /* bytes                                                  */
	uint64_t *block_offsets;    /* device out: offset of each block's u32 size prefix    */
	uint32_t *result;           /* device out: [0]=status (SNAPPY_HIP_BLOCK_*), [1]=blocks walked */
	uint32_t total_len;         /* uncompressed length from the header                    */
	uint32_t block_size;        /* from the header                                        */
	uint32_t header_len;        /* bytes of the two varints                               */
	uint32_t num_blocks;        /* ceil(total_len / block_size)                           */
} snappy_hip_stream_desc;

/* Page-locked host memory for callers that want PCIe-rate copies through the drop-in pair (the CLI reads its
 * input file straight into such a buffer).  NULL on failure. */
SNAPPY_HIP_API void *snappy_hip_host_alloc(size_t bytes);
SNAPPY_HIP_API void snappy_hip_host_free(void *p);

SNAPPY_HIP_API int snappy_hip_device_count(void);
SNAPPY_HIP_API int snappy_hip_set_device(int device);
SNAPPY_HIP_API const char *snappy_hip_last_error(void);
/* name of the code-object architecture this library was built for ("gfx950") */
SNAPPY_HIP_API const char *snappy_hip_arch(void);

/* Bytes reserved per block in the slot buffer: 16-byte multiple >= 4 + 32 + bs + bs/6
 * (u32 prefix + snappy_max_compressed_length, snappy_compress.c:55-60). */
SNAPPY_HIP_API uint32_t snappy_hip_slot_stride(uint32_t block_size);
SNAPPY_HIP_API uint64_t snappy_hip_num_blocks(uint64_t input_len, uint32_t block_size);
/* Upper bound of the framed stream for input_len bytes (header + all slots' payload). */
SNAPPY_HIP_API uint64_t snappy_hip_stream_bound(uint64_t input_len, uint32_t block_size);
/* Writes varint(total_len) varint(block_size) (snappy_compress.c:461-465) to a HOST buffer
 * of >= 10 bytes; returns header length. */
SNAPPY_HIP_API uint32_t snappy_hip_write_header(uint8_t *dst, uint32_t total_len, uint32_t block_size);
/* Parses the two header varints from a HOST buffer (snappy_decompress.c:193-198, :220-225);
 * returns header length, 0 if malformed. */
SNAPPY_HIP_API uint32_t snappy_hip_parse_header(const uint8_t *src, uint64_t avail, uint32_t *total_len, uint32_t *block_size);

/*
 * K1: compress every block of d_in independently (semantics of compress_block,
 * snappy_compress.c:284-413).  Block b's u32 size prefix + elements go to
 * d_slots + b*slot_stride; d_block_bytes[b] = 4 + compressed size.
 * d_in must be 16-byte aligned.  `stream` is a hipStream_t (NULL = default stream).
 *
 * d_scratch: 256-byte aligned device workspace of snappy_hip_compress_scratch_bytes() bytes (one 64 KiB
 * hash table per wavefront slot of the CURRENT device + a work counter: 512 MiB on a whole MI355X, 64 MiB on a
 * 32-CU partition -- the size is taken from hipGetDeviceProperties, so ask with the device selected that will
 * run the launch; contents need not be initialised, the buffer must not be shared by launches that run
 * concurrently).  If NULL or too small the LDS-table kernel is used instead (lower occupancy, same bytes).
 * After EVERY launch that was given a scratch, the u32 at byte 16 of the scratch holds the number of blocks that
 * were compressed by LDS-table wavefronts (statistics only): all of them when a small input went to the
 * LDS-table kernel alone, none with SNAPPY_HIP_LDS_WAVES=0.
 */
SNAPPY_HIP_API uint64_t snappy_hip_compress_scratch_bytes(void);
/* Wavefronts per CU whose hash table lives in LDS in a default K1 launch at this block size (the table is sized by the
 * block size, so small blocks get more of them: reference dpu_compress.c:16, :472-476 sizes its table to the tasklet's
 * memory the same way).  For the block-size sweep's occupancy column. */
SNAPPY_HIP_API uint32_t snappy_hip_k1_lds_waves_per_cu(uint32_t block_size);
SNAPPY_HIP_API int snappy_hip_compress_blocks(const uint8_t *d_in, uint64_t input_len, uint32_t block_size,
                               uint8_t *d_slots, uint32_t slot_stride, uint32_t *d_block_bytes,
                               void *d_scratch, uint64_t scratch_bytes, void *stream);

/*
 * K1 over a batch of containers in ONE launch: the same per-block semantics as snappy_hip_compress_blocks for every
 * item (its own input, slot array and size array; all with the same block_size and slot_stride), the persistent
 * wavefronts drawing blocks of all containers from one counter, so the batch has one tail instead of one per container.
 * This is the device-side form of the reference compressing many independent files, one `dpu_snappy -c` run each
 * (snappy/dpu_snappy.c:160-172); items is a HOST array, empty containers are skipped, lists longer than 8 non-empty
 * containers are issued as several launches on `stream`.
 */
struct snappy_hip_compress_item {
    const void *d_input;        /* 16-byte aligned device pointer */
    uint64_t input_len;         /* < 4 GiB */
    void *d_slots;              /* num_blocks(input_len) * slot_stride bytes, 16-byte aligned */
    void *d_block_bytes;        /* num_blocks(input_len) u32 */
};
SNAPPY_HIP_API int snappy_hip_compress_blocks_batch(const struct snappy_hip_compress_item *items, uint32_t count, uint32_t block_size,
                                     uint32_t slot_stride, void *d_scratch, uint64_t scratch_bytes, void *stream);

/*
 * Exclusive scan of d_block_bytes + gather of the slots into the contiguous framed stream
 * (header written too).  d_offsets: scratch/out, num_blocks+1 u64 (offset of each block in
 * d_stream; [num_blocks] = stream length, also stored to *d_stream_len if non-NULL).
 * This is the device-side form of the per-tasklet fwrite concat, snappy_compress.c:697-704.
 */
SNAPPY_HIP_API int snappy_hip_compact(const uint8_t *d_slots, uint32_t slot_stride, const uint32_t *d_block_bytes,
                       uint64_t input_len, uint32_t block_size,
                       uint8_t *d_stream, uint64_t *d_offsets, uint64_t *d_stream_len, void *stream);

/*
 * Find the u32 size chains of `count` streams from the streams' bytes alone, the device form of the
 * host pre-scan snappy_decompress.c:317-340.  d_descs: device array of `count` descriptors; for every stream
 * block_offsets[0 .. num_blocks) and result[0] (SNAPPY_HIP_BLOCK_OK / _INVALID), result[1] (blocks found) are written.
 * The chain is first sought in parallel: 256 walkers per stream start at recognised block boundaries and walk their
 * share; the shares are laid end to end iff each one ends exactly on the next one's starting point and the hops number
 * num_blocks -- which makes them the chain, whatever the recognition did.  A stream this leaves unresolved (blocks of a
 * few bytes, a damaged stream, a stream of 4 GiB or more) is walked serially, one wavefront per stream, with the same
 * result.  SNAPPY_HIP_INDEX_PARALLEL=0: the serial walk only.
 * Uses a library-owned device workspace (2.1 MB per stream of the call), allocated on first use and when a call brings
 * more streams than any before it on this device: that is the only case in which this function calls the allocator (and
 * waits for the previous call on that device); otherwise it only enqueues.
 */
SNAPPY_HIP_API int snappy_hip_index_streams(const snappy_hip_stream_desc *d_descs, uint32_t count, void *stream);

/*
 * Check candidate indexes against the size chains of `count` streams, every link in parallel: a caller that already
 * holds the block offsets -- the d_offsets array snappy_hip_compact just produced for the same stream, or an index kept
 * beside the file -- need not repeat the serial walk, but the stream stays the authority: d_descs[i].block_offsets must
 * hold num_blocks + 1 entries with [0] = header_len, [num_blocks] = stream_len, and the u32 stored at [b] leading
 * exactly to [b + 1] for every b, which is the result of the walk of snappy_decompress.c:317-340 by induction.
 * result[0] = SNAPPY_HIP_BLOCK_OK when every link holds, SNAPPY_HIP_BLOCK_INVALID otherwise (then use
 * snappy_hip_index_streams, which needs no candidate); result[1] = number of links that hold.
 */
SNAPPY_HIP_API int snappy_hip_verify_index(const snappy_hip_stream_desc *d_descs, uint32_t count, void *stream);

/*
 * K2: decode every block (semantics of snappy_decompress.c:232-285 on well-formed streams,
 * strict otherwise).  Block i is read at d_stream + d_block_offsets[i] and decoded to
 * d_out + i*block_size; d_status[i] = SNAPPY_HIP_BLOCK_*.
 */
SNAPPY_HIP_API int snappy_hip_decompress_blocks(const uint8_t *d_stream, uint64_t stream_len, const uint64_t *d_block_offsets,
                                 uint64_t total_len, uint32_t block_size,
                                 uint8_t *d_out, uint32_t *d_status, void *stream);

/*
 * K2 over a batch of streams in ONE launch: the same per-block semantics as snappy_hip_decompress_blocks for every item
 * (its own stream, block offsets, output and status arrays; all with the same block_size), the persistent wavefronts
 * drawing blocks of all streams from one counter, so the batch has one tail instead of one per stream -- the device-side
 * form of the reference decoding many independent files, one `dpu_snappy` run each (snappy/dpu_snappy.c:186-192).  items is
 * a HOST array, empty streams are skipped, lists longer than 8 non-empty streams are issued as several launches.
 */
struct snappy_hip_decompress_item {
    const void *d_stream;          /* device: the framed stream (its header)                */
    uint64_t stream_len;           /* bytes; ignored when d_stream_len is given             */
    const void *d_stream_len;      /* device u64 or NULL: the length as left by snappy_hip_compact (*d_stream_len), so that a
                                      compress -> decompress chain needs no host round trip for it */
    const void *d_block_offsets;   /* device: num_blocks(total_len) u64                     */
    uint64_t total_len;            /* uncompressed length from the header                   */
    void *d_out;                   /* device: total_len bytes                               */
    void *d_status;                /* device: num_blocks(total_len) u32                     */
};
SNAPPY_HIP_API int snappy_hip_decompress_blocks_batch(const struct snappy_hip_decompress_item *items, uint32_t count, uint32_t block_size,
                                       void *stream);

#ifdef __cplusplus
}
#endif
#endif /* SNAPPY_HIP_H_ */
