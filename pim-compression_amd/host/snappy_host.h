/*
 * snappy_host.h -- host (CPU) mode of the dpu_snappy CLI: the same four entry points the
 * reference exposes for its host path (snappy/snappy_compress.h:16-26, snappy/snappy_decompress.h:15-24).
 * This is the explicit "no -d" mode of the tool, selected by the user; it is never used as a
 * fallback for the GPU path (-d fails loudly instead).
 */
#ifndef SNAPPY_HOST_H_
#define SNAPPY_HOST_H_

#include "../../include/snappy_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

void setup_compression(struct host_buffer_context *input, struct host_buffer_context *output, struct program_runtime *runtime);
snappy_status snappy_compress_host(struct host_buffer_context *input, struct host_buffer_context *output, uint32_t block_size);
snappy_status setup_decompression(struct host_buffer_context *input, struct host_buffer_context *output, struct program_runtime *runtime);
snappy_status snappy_decompress_host(struct host_buffer_context *input, struct host_buffer_context *output);
double get_runtime(struct timeval *start, struct timeval *end);

#ifdef __cplusplus
}
#endif
#endif
