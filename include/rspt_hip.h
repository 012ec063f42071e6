/*
 * rspt_hip.h -- C ABI of the MI355X (gfx950) signal_packer hot path.
 *
 * This is the drop-in boundary: plain pointers and sizes, no C++ or torch
 * types.  The reference has no FFI of its own (it is a single C++ library);
 * the entry points below are what its i_signal_packer factories and
 * compress()/decompress() virtuals (lib_rspt/signal_packer.h:29-73) bind to
 * when the path runs on the GPU -- include/signal_packer.h holds the
 * API-identical C++ classes that call them, INTEGRATION.md shows the
 * reference-side binding.
 *
 * Conventions: every function returns RSPT_HIP_OK (0) or a negative
 * rspt_hip_status; nothing throws across this boundary.  A handle owns its
 * device workspace and one HIP stream and is not thread-safe (one handle per
 * host thread), exactly like a reference packer instance
 * (signal_packer_base.h:20-21 scratch tensors).  There is no CPU fallback: if
 * no gfx950 device is usable, create() fails.
 */
#ifndef RSPT_HIP_H_
#define RSPT_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rspt_hip_packer rspt_hip_packer;

/* Factory selector -- NOT the stream's method byte (which is 0,0,1,2).
 * Replaces i_signal_packer::new_hzr / new_xdelta_hzr / new_dct / new_hadamard
 * (lib_rspt/signal_packer.h:59-69). */
enum rspt_hip_kind {
    RSPT_HIP_KIND_HZR = 0,        /* lib_signalpacker/signal_packer_hzr.cpp:34-68        */
    RSPT_HIP_KIND_XDELTA_HZR = 1, /* lib_signalpacker/signal_packer_xdelta_hzr.cpp:34-88 */
    RSPT_HIP_KIND_DCT = 2,        /* lib_signalpacker/signal_packer_dct.cpp:36-156       */
    RSPT_HIP_KIND_HADAMARD = 3    /* lib_signalpacker/signal_packer_hadamard.cpp:35-107  */
};

/* Test hook, OR-ed into `kind` at create: a dct packer takes the fp64 FFT path also where the bit-exact dense-table path
 * would run (ns = 2^k <= 8192), so that the two can be compared with each other (tests/test_gpu_dct_fft.py). */
#define RSPT_HIP_DCT_FORCE_FFT 0x100

enum rspt_hip_status {
    RSPT_HIP_OK = 0,
    RSPT_HIP_ERR_ARG = -1,         /* bad kind / sizes (bps not 1..4, nb not 1..4, ns not 2^k for hadamard ...) */
    RSPT_HIP_ERR_NO_DEVICE = -2,   /* no usable gfx950 device; there is no CPU path */
    RSPT_HIP_ERR_ALLOC = -3,       /* device or host allocation failed */
    RSPT_HIP_ERR_LAUNCH = -4,      /* a HIP call or kernel launch failed (see rspt_hip_last_hip_error) */
    RSPT_HIP_ERR_DST_TOO_SMALL = -5, /* compressed block does not fit dst_max_len / dst_stride */
    RSPT_HIP_ERR_CORRUPT = -6,     /* decompress: malformed stream */
    RSPT_HIP_ERR_UNSUPPORTED = -7, /* shape outside what the kernels handle (documented in DESIGN.md) */
    RSPT_HIP_ERR_BUSY = -8         /* rspt_hip_feed_push: every slot of the feed is in flight (poll, then push again) */
};

const char* rspt_hip_status_string(int status);
/* hipError_t of the last failing HIP call on this handle (0 if none). */
int rspt_hip_last_hip_error(const rspt_hip_packer* p);

/* Number of gfx950 devices visible to this process (0 if none). */
int rspt_hip_device_count(void);

/* Constructor.  Replaces the packer constructors
 * (signal_packer_xdelta_hzr.cpp:42-50, _hzr.cpp:42-49, _hadamard.cpp:47-55,
 * _dct.cpp:49-58): same meaning of (bytes_per_sample, nr_channels,
 * nr_samples_per_channel, nr_bytes_to_encode); `nb` is read by
 * RSPT_HIP_KIND_XDELTA_HZR only.  `device` is the HIP device ordinal. */
int rspt_hip_packer_create(rspt_hip_packer** out, int kind, size_t bps, size_t nch, size_t ns, size_t nb, int device);

/* Replaces i_signal_packer::delete_* (signal_packer.h:60-72). */
void rspt_hip_packer_destroy(rspt_hip_packer* p);

/* i_signal_packer::compress (signal_packer.h:44), host buffers.
 * src = bps*nch*ns bytes, interleaved sample-major little-endian.
 * Fails with RSPT_HIP_ERR_DST_TOO_SMALL instead of the reference's undefined
 * result when the stream does not fit dst_max_len. */
int rspt_hip_compress(rspt_hip_packer* p, const void* src_host, void* dst_host, size_t dst_max_len, size_t* dst_len);

/* i_signal_packer::decompress (signal_packer.h:57), host buffers.
 * *src_len is an OUTPUT (bytes consumed), as in the reference. */
int rspt_hip_decompress(rspt_hip_packer* p, const void* src_host, size_t* src_len, void* dst_host);

/* The same for a caller that knows how many bytes are readable at src_host (no counterpart in the reference, whose decompress
 * trusts the stream's own length fields -- as rspt_hip_decompress does, up to rspt_hip_max_compressed_size: a damaged length field
 * of a stream from an untrusted source can send it past the end of a shorter buffer).  Nothing beyond src_host + src_cap is
 * read; a stream whose framing says otherwise is RSPT_HIP_ERR_CORRUPT. */
int rspt_hip_decompress_bounded(rspt_hip_packer* p, const void* src_host, size_t src_cap, size_t* src_len, void* dst_host);

/* Worst-case stream size of ONE block for this packer, allowing for nb
 * escalation up to 4 (1 + header + nb*(4 + hzr_max_compressed_size(nch*ns)),
 * hzr_encode.c:489-497, signal_packer_base.cpp:83-95). */
size_t rspt_hip_max_compressed_size(const rspt_hip_packer* p);

/* Input bytes per block: bps*nch*ns. */
size_t rspt_hip_block_bytes(const rspt_hip_packer* p);

/* Current nr_bytes_to_compress_ (signal_packer_xdelta_hzr.cpp:39,66): the
 * state that mutates on escalation and must match between compressor and
 * decompressor because it is not in the stream.  Synchronises the stream. */
unsigned rspt_hip_current_nb(rspt_hip_packer* p);
/* Set it (a decoder fed streams from another instance needs this). */
int rspt_hip_set_nb(rspt_hip_packer* p, unsigned nb);

/* Decompress normally trusts the stream like the reference does (hzr_decode.c:343 skips the block CRCs).  With
 * verify on, every Huffman / PlainCopy block's CRC-32C is recomputed on the device and compared with its header
 * (what hzr_verify does, hzr_decode.c:569-624); a mismatch makes the call return RSPT_HIP_ERR_CORRUPT (batch
 * form: bit 63 of d_consumed[b]).  Off by default. */
int rspt_hip_set_verify(rspt_hip_packer* p, int on);

/* Byte order of the samples (default 0 = little-endian, what every reference packer passes: signal_packer_xdelta_hzr.cpp:54).
 * With big_endian != 0 compress reads samples whose bytes are reversed -- convert_native_to_i32(..., reverse_byte_order =
 * true), lib_signalpacker/utils.cpp:127-137,145-154,162-170 -- and decompress writes them that way (convert_i32_to_native,
 * utils.cpp:57-64,77-85,97-104): a 24-bit ADC feed in network byte order needs no host pass.  The streams are those of the
 * byte-reversed (little-endian) block. */
int rspt_hip_set_byte_order(rspt_hip_packer* p, int big_endian);

/* Page-locked host memory for the host-pointer entry points: with src / dst in such buffers (16-byte aligned source)
 * rspt_hip_compress has no copy phases at all -- the front end reads the samples across the link as it transforms them and the
 * encoders write the stream into the caller's buffer -- and rspt_hip_decompress writes the samples there from its last kernel;
 * the many-block forms move their data by DMA at link rate instead of through the runtime's pageable staging copies
 * (the acquisition front end of the reference, lib_ring_buffer/ring_buffers.h:150-203 io_buffer, would allocate its
 * slots here).  NULL on failure. */
void* rspt_hip_host_alloc(size_t bytes);
void rspt_hip_host_free(void* p);

/* A sequence of blocks from host memory: what a caller of the reference does one compress() call per block
 * (the call itself: lib_rspt_test/rspt_test.cpp:66-72; a packer is made for ONE block shape, signal_packer.h:60-72, so a
 * recording longer than that is a loop of such calls on the same instance).  nblocks consecutive blocks at src_host
 * (rspt_hip_block_bytes apart); stream i goes to dst_host + i * dst_stride, its length to dst_len[i]; the nb
 * state moves from block to block as in repeated compress() calls.  Chunks of blocks pass through
 * upload | compress | download on three HIP streams, so a long sequence runs at the rate of the slowest
 * stage (the upload over PCIe when src_host is page-locked) instead of at the sum of the three.  Blocking.
 * A stream that does not fit dst_stride: dst_len[i] = the size it needs, nothing copied for it, and the call
 * returns RSPT_HIP_ERR_DST_TOO_SMALL after finishing the others. */
int rspt_hip_compress_many(rspt_hip_packer* p, const void* src_host, size_t nblocks, void* dst_host, size_t dst_stride, size_t* dst_len);

/* The inverse: nblocks streams at src_host + i * src_stride (src_stride <= rspt_hip_max_compressed_size rounded up to 256)
 * -> blocks at dst_host + i * rspt_hip_block_bytes, bytes consumed to consumed[i].  As in decompress() a stream's length is
 * not needed; where the caller knows it, src_len[i] (may be NULL) bounds what is uploaded of stream i -- with 16 MiB blocks
 * and a stride sized for the worst case that is a fifth of the bytes.  All streams are decoded with the packer's current nb
 * (rspt_hip_set_nb), like successive decompress() calls.  Same three-stage pipeline; here the download bounds it.  A stream
 * that does not decode: consumed[i] = 0 and the call returns RSPT_HIP_ERR_CORRUPT after finishing the others. */
int rspt_hip_decompress_many(rspt_hip_packer* p, const void* src_host, size_t src_stride, const size_t* src_len, size_t nblocks, void* dst_host,
                             size_t* consumed);

/* ---- a feed of blocks that arrive over time (the acquisition side: lib_ring_buffer/ring_buffers.h:150-203 io_buffer hands
 * a consumer one filled buffer after the other; the consumer's loop is rspt_test.cpp:66-72, one compress() per block) ----------
 * Non-blocking: push a block when it is there, poll for finished streams when convenient.  Behind it the same three-stage
 * pipeline as rspt_hip_compress_many (upload | compress | download on three HIP streams), over a ring of `slots` groups of
 * `blocks_per_launch` blocks; the streams are byte for byte those of successive compress() calls in push order (the nb state
 * moves from block to block).
 *   rspt_hip_feed_begin   allocate the ring (slots >= 2; blocks_per_launch >= 1: 1 = lowest latency, more = higher rate)
 *   rspt_hip_feed_push    queue one block: src_host (rspt_hip_block_bytes() bytes; page-locked memory from rspt_hip_host_alloc
 *                         uploads by DMA) -> its stream will be written to dst_host (dst_cap bytes).  Both must stay valid
 *                         until the block is reported by rspt_hip_feed_poll.  Returns RSPT_HIP_OK, or RSPT_HIP_ERR_BUSY when
 *                         every slot is in flight (poll, then push again) -- it never waits.  A group is launched when it is
 *                         full; rspt_hip_feed_submit launches a partly filled one (when no more blocks are expected soon).
 *   rspt_hip_feed_poll    report ONE finished block, in push order: returns 1 and sets *seq (0, 1, 2, ... in push order),
 *                         *dst_len and *status (RSPT_HIP_OK, or RSPT_HIP_ERR_DST_TOO_SMALL with *dst_len = the size needed
 *                         and nothing copied; or the status of a launch that failed -- rspt_hip_feed_push / _submit returned it
 *                         when it happened -- for every block of that group, with *dst_len = 0); returns 0 when none is
 *                         ready yet; never waits.
 *   rspt_hip_feed_flush   submit what is queued and wait until everything pushed so far can be polled
 *   rspt_hip_feed_end     flush, drop unpolled results, free the ring.
 * One feed per handle; while a feed is open the batch and many-block entry points of the handle return RSPT_HIP_ERR_ARG. */
int rspt_hip_feed_begin(rspt_hip_packer* p, size_t blocks_per_launch, size_t slots);
int rspt_hip_feed_push(rspt_hip_packer* p, const void* src_host, void* dst_host, size_t dst_cap);
int rspt_hip_feed_submit(rspt_hip_packer* p);
int rspt_hip_feed_poll(rspt_hip_packer* p, size_t* seq, size_t* dst_len, int* status);
int rspt_hip_feed_flush(rspt_hip_packer* p);
int rspt_hip_feed_end(rspt_hip_packer* p);

/* ---- device-resident, batched forms (bench, multi-GPU shards) ------------ */

/* Grow the workspace so that up to max_blocks blocks can go through one
 * launch sequence.  Called implicitly by the batch entry points; call it
 * yourself to keep allocation out of a timed region. */
int rspt_hip_reserve(rspt_hip_packer* p, size_t max_blocks);

/* Compress nblocks independent blocks that are resident in device memory.
 * Semantics = nblocks successive compress() calls on one packer instance
 * (including the persistent nb escalation, in block order).
 *   d_src      nblocks * rspt_hip_block_bytes() bytes, blocks back to back
 *   d_dst      block b's stream is written at d_dst + b*dst_stride
 *   d_sizes    nblocks uint64: stream length of block b; if it would not fit
 *              dst_stride nothing is written for that block and bit 63 is set
 *   stream     hipStream_t (as void*) to enqueue on; NULL is the HIP null
 *              stream, as everywhere in HIP.  rspt_hip_stream() returns the
 *              handle's own stream for callers that want that one.
 * Asynchronous: returns after enqueueing.
 * Ordering contract of a handle: it is ONE packer instance (one workspace, one nb state), so successive batch calls on it
 * must be stream-ordered -- the same `stream` for all of them, or each call ordered behind the previous one by an event.
 * (Inside, the per-call scratch that must start from zero exists twice and the kernels of call i zero the copy that call
 * i+1 works in; the small-block encoder runs on a side stream of the handle, forked from and joined to `stream` inside the
 * call.)  Two calls racing on different streams would share planes, counters and that zeroing.  One host word is read
 * without synchronisation, by design: the small-block count a RECENT batch left in page-locked memory only decides whether
 * the side stream is used at all (a batch shape without small blocks skips the fork/join); a stale value costs a few
 * microseconds, never correctness. */
int rspt_hip_compress_batch_dev(rspt_hip_packer* p, const void* d_src, size_t nblocks, void* d_dst, size_t dst_stride,
                                uint64_t* d_sizes, void* stream);

/* Decompress nblocks streams resident in device memory (stream b at
 * d_src + b*src_stride) into d_dst (nblocks * rspt_hip_block_bytes() bytes).
 * d_consumed[b] = bytes of stream b used; bit 63 set = malformed stream. */
int rspt_hip_decompress_batch_dev(rspt_hip_packer* p, const void* d_src, size_t src_stride, size_t nblocks, void* d_dst,
                                  uint64_t* d_consumed, void* stream);

/* ---- container: many streams, back to back (what a multi-GPU gather ships) ---
 * layout (little-endian):
 *   u64 magic 'RSPTPACK' | u64 nblocks | u64 payload_bytes | u64 nb of the LAST stream (low 32 bits), flagged streams (high 32)
 *   nblocks x { u64 offset, u64 length | nb << 56 | invalid << 63 }   offsets relative to the payload, 16-byte aligned
 *   payload                                    stream b at payload + offset[b], zero padded to 16 bytes
 * rspt_hip_pack_bound() bytes always suffice for d_packed.  d_total (device u64)
 * receives the container length.  The reference keeps nb out of the stream
 * (signal_packer_xdelta_hzr.cpp:39,66) and nb escalates inside a batch (and on every rank
 * independently), so each index entry carries the plane count of ITS stream: a container decodes
 * in one call whatever mix of nb it holds.  A stream that did not fit dst_stride at compress time
 * (bit 63 of its d_sizes entry) becomes an empty entry with the invalid bit set and is counted in
 * the header; nothing is copied for it.  d_sizes / nblocks must describe the handle's LAST
 * compress_batch call (that call's per-block nb is what the index records). */
size_t rspt_hip_pack_bound(const rspt_hip_packer* p, size_t nblocks);
int rspt_hip_pack_batch_dev(rspt_hip_packer* p, const void* d_dst, size_t dst_stride, const uint64_t* d_sizes, size_t nblocks, void* d_packed,
                            uint64_t* d_total, void* stream);

/* Decompress the nblocks streams of a container resident in device memory (16-byte aligned, packed_len bytes), as
 * laid out above: the consumer side of the gather.  Every stream is decoded with the nb of its own index entry; the
 * handle's nb state is neither used nor changed.  Header and index are checked against packed_len on the device: a
 * truncated or corrupt container flags its streams (bit 63 of d_consumed[b]) instead of reading out of bounds. */
int rspt_hip_decompress_packed_dev(rspt_hip_packer* p, const void* d_packed, size_t packed_len, size_t nblocks, void* d_dst, uint64_t* d_consumed,
                                   void* stream);

/* ---- multi-GPU: gather the containers of all ranks to one rank over RCCL -----------------------------
 * One process (or thread) per GPU, each with its own handle and its own ncclComm_t of one communicator.  The ranks hold
 * contiguous shards of independent blocks (no data-path collective); what travels is the result: the sizes by
 * ncclAllGather, the payload as ONE group of ncclSend / ncclRecv straight from every peer to the root (a gatherv over the
 * direct xGMI links; no ring).  `comm` is the caller's ncclComm_t, passed as void* so that this header needs no RCCL
 * header; the library binds to RCCL at the first call (see below which copy) and fails with RSPT_HIP_ERR_UNSUPPORTED when
 * there is none.
 *
 *   rspt_hip_gather_sizes     d_total (device u64: this rank's container length, as rspt_hip_pack_batch_dev wrote it) ->
 *                             d_totals[world] on every rank, and -- if h_totals is not NULL -- a copy in the caller's
 *                             page-locked host array h_totals[world] (asynchronous: valid once `stream` has got there)
 *   rspt_hip_gather_payload   with the sizes known on the host: root receives rank r's container at
 *                             d_recv + r * recv_stride (its own is copied there too), the others send theirs.
 *                             recv_stride >= the largest container (rspt_hip_pack_bound() always suffices), a multiple of
 *                             16 (rspt_hip_decompress_packed_dev wants 16-byte aligned containers; else RSPT_HIP_ERR_ARG),
 *                             the SAME value on every rank: a container that does not fit makes every rank return
 *                             RSPT_HIP_ERR_DST_TOO_SMALL before anything is posted
 *   rspt_hip_gather_containers  both, with one stream synchronisation in between (the sizes must reach the host before
 *                             the transfers can be posted); h_totals[world] receives the sizes on every rank.
 *                             A caller that gathers every step posts the payload of step i after the sizes of step
 *                             i+1 instead (no synchronisation: see rspt_amd/shard.py LaggedGather for the pattern).
 *   rspt_hip_gather_post_sizes / _post_payload / _wait   the same gather for a caller that gathers EVERY step, without a host
 *                             synchronisation in the step (two slots, 0 and 1, alternate):
 *                               step i:  ... compress + rspt_hip_pack_batch_dev on `stream` ...
 *                                        rspt_hip_gather_post_payload(slot of step i-1)   -- its sizes arrived a step ago
 *                                        rspt_hip_gather_post_sizes(slot of step i, stream)
 *                               at the end: rspt_hip_gather_post_payload(last slot).
 *                             Both run on a gather stream of the handle: post_sizes orders it behind `stream` (the pack),
 *                             the payload of step i-1 is posted in front of that and overlaps the kernels of step i.
 *                             post_payload reads the sizes on the host (waits for them only if they have not arrived),
 *                             copies them to h_totals[world] if that is not NULL, and posts the transfers;
 *                             rspt_hip_gather_wait(slot, stream) makes `stream` wait for the slot's payload (before the
 *                             container buffer of that slot is written again, before d_recv is read).
 * The library binds to the copy of RCCL the process has already mapped (where the caller's ncclComm_t came from); with none
 * mapped it loads librccl.so.1; with two different copies mapped (e.g. a framework's bundled one next to /opt/rocm's) every
 * gather call returns RSPT_HIP_ERR_UNSUPPORTED unless the environment variable RSPT_RCCL_LIB names the one to use.
 * All ranks must make the same calls in the same order.  Asynchronous on `stream` except where said. */
int rspt_hip_gather_sizes(rspt_hip_packer* p, void* comm, int world, const uint64_t* d_total, uint64_t* d_totals, uint64_t* h_totals, void* stream);
int rspt_hip_gather_payload(rspt_hip_packer* p, void* comm, int rank, int world, int root, const void* d_packed, const uint64_t* h_totals,
                            void* d_recv, size_t recv_stride, void* stream);
int rspt_hip_gather_containers(rspt_hip_packer* p, void* comm, int rank, int world, int root, const void* d_packed, const uint64_t* d_total,
                               void* d_recv, size_t recv_stride, uint64_t* h_totals, void* stream);
int rspt_hip_gather_post_sizes(rspt_hip_packer* p, void* comm, int world, const uint64_t* d_total, int slot, void* stream);
int rspt_hip_gather_post_payload(rspt_hip_packer* p, void* comm, int rank, int world, int root, const void* d_packed, int slot, void* d_recv,
                                 size_t recv_stride, uint64_t* h_totals);
int rspt_hip_gather_wait(rspt_hip_packer* p, int slot, void* stream);

/* ---- optional stage in front of compress: the reference's IIR pre-filter ---------------------------------------
 * Replaces the filter step of the reference's own pipeline (lib_rspt_test/rspt_test.cpp:116-136): i_filter::new_iir
 * (n, d, nr_coefficients), init_history_values(first sample of the channel, init_nr_samples), filter_opt on every sample
 * (lib_rspt/lib_filter/iir_filter.cpp:46-116), result truncated to int32 and stored back in the native sample width --
 * nblocks device-resident blocks (interleaved native layout, as for compress), IN PLACE, bit-identical with the reference.
 *   n, d            host arrays of nr_coefficients doubles (2..5): feedback (n[0] unused) and feed-forward coefficients
 *   per_channel     0: one filter object for all channels of a block, its state running on from channel to channel, as
 *                      in the reference's harness (the channels of a block are then a serial chain: one thread per block);
 *                   1: a fresh filter per channel (one i_filter per channel): one thread per channel
 * Asynchronous on `stream`. */
int rspt_hip_iir_prefilter_batch_dev(rspt_hip_packer* p, void* d_buf, size_t nblocks, const double* n, const double* d, size_t nr_coefficients,
                                     int init_nr_samples, int per_channel, void* stream);

/* The handle's own (non-blocking) stream, as a hipStream_t. */
void* rspt_hip_stream(rspt_hip_packer* p);

/* Wait for the handle's own stream. */
int rspt_hip_synchronize(rspt_hip_packer* p);

/* ---- measurement hooks ---------------------------------------------------- */

/* When enabled, every batch call brackets each kernel of the sequence with
 * HIP events on the launch stream.  rspt_hip_stage_times() then synchronises
 * and returns, for the LAST batch call, the per-stage elapsed milliseconds.
 * Stage names are returned by rspt_hip_stage_name(i). */
int rspt_hip_set_profiling(rspt_hip_packer* p, int on);
int rspt_hip_stage_count(const rspt_hip_packer* p);
const char* rspt_hip_stage_name(const rspt_hip_packer* p, int i);
int rspt_hip_stage_times(rspt_hip_packer* p, float* ms, int n);

/* Test hook: copy a workspace buffer of the LAST batch call to the host
 * (synchronises).  which: 0 planes [blocks][4][plane_stride] u8, 1 planar
 * int32 [blocks][N], 2 second int32 buffer (dct), 3 token histograms
 * [blocks*4*nblk][264] u32, 4 block records [..][4] u32 (mode, payload_len,
 * tree_bits, fill), 5 nb per block [blocks] u32, 6 means header bytes.
 * Returns the number of bytes copied (<= cap) or a negative status. */
long long rspt_hip_debug_read(rspt_hip_packer* p, int which, void* host_buf, size_t cap);

#ifdef __cplusplus
}
#endif
#endif /* RSPT_HIP_H_ */
