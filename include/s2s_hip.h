/* s2s_hip.h -- C ABI of the MI355X-native seq2squiggle predict path.
 *
 * The reference (ZKI-PH-ImageAnalysis/seq2squiggle v0.3.4) is pure Python and has no FFI of
 * its own; this header is the boundary a maintainer binds with ctypes (INTEGRATION.md shows
 * the stub).  Each entry point names the reference code it replaces (paths relative to the
 * reference's src/seq2squiggle/).
 *
 * Conventions
 *   - plain pointers and sizes only; every buffer is owned by the caller;
 *   - pointers marked "device" are HIP device pointers on the handle's device;
 *   - all launches are stream-ordered on `stream` (a hipStream_t passed as void*, NULL = default
 *     stream) and asynchronous w.r.t. the host;
 *   - return value 0 = success, negative = error; s2s_last_error() gives the message;
 *   - a handle may be used from one host thread and on one stream at a time (it owns the per-workgroup hand-off slots that
 *     consecutive launches reuse in stream order); every call makes the handle's device current for its duration
 *     and restores the caller's; there is no global state.
 */
#ifndef S2S_HIP_H
#define S2S_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define S2S_OK 0
#define S2S_ERR_ARG (-1)     /* bad argument / unsupported configuration */
#define S2S_ERR_HIP (-2)     /* a HIP runtime call failed                 */
#define S2S_ERR_BLOB (-3)    /* weight blob has the wrong size            */
#define S2S_ERR_CODEC (-4)   /* a host compression codec is unavailable (libzstd.so.1 not loadable) or failed */

/* Decoder arithmetic.  F32 and F16X3 stay inside the 1e-4 pA MAE parity bound.
 *   S2S_MODE_F32    every product on the f32-input MFMA (v_mfma_f32_16x16x4_f32), exact fp32;
 *   S2S_MODE_F16X3  operands split into two f16 halves (22 bits), three f16 MFMA products with
 *                   fp32 accumulation per original product (v_mfma_f32_16x16x32_f16). */
#define S2S_MODE_F32 0
#define S2S_MODE_F16X3 1
/* (value 2 is retired: a 32x32x16-tiled variant of F16X3 that never beat it) */
#define S2S_MODE_F16 3      /* REDUCED PRECISION, outside the 1e-4 pA bound: decoder operands rounded to f16 once (the
                             * precision class of the reference's own fp16-autocast GPU path, inference.py:404), one MFMA
                             * product per product; the frontend stays F16X3, so dwell indices remain bit-exact */

#define S2S_T_ENC 16         /* config.yaml:18 max_dna_len    */
#define S2S_T_DEC 250        /* config.yaml:19 max_signal_len */
#define S2S_DMODEL 64        /* config.yaml:25 dmodel         */
#define S2S_DFF 256          /* config.yaml:26 dff            */
#define S2S_HEADS 8          /* config.yaml:28,30             */

/* Model hyper-parameters that change the predict arithmetic (config.yaml:17-31; the keys
 * Encoder/Decoder.__init__ read, modules.py:22-63, 97-131).  Only the shipped architecture
 * family is supported: dmodel 64, dff 256, 8 heads, 16 k-mers in, 250 samples out; the layer
 * counts and the k-mer size are free. */
typedef struct s2s_config {
    int32_t seq_kmer;          /* 9 (dna-r10*, rna-004*) or 6 (dna-r9*), utils.py:257-260 */
    int32_t max_dna_len;       /* must be 16  */
    int32_t max_signal_len;    /* must be 250 */
    int32_t dmodel;            /* must be 64  */
    int32_t dff;               /* must be 256 */
    int32_t n_heads;           /* must be 8 (encoder_heads == decoder_heads) */
    int32_t encoder_layers;    /* 1..4 */
    int32_t decoder_layers;    /* 1..4 */
    int32_t pre_layers;        /* 0..4 */
    float scaling_max_value;   /* 165.0, model.py:221 */
    int32_t compute_mode;      /* arithmetic of the decoder FFT blocks: one of the S2S_MODE_* values */
} s2s_config;

/* Per-call scalars: the attributes predict_step reads from the LightningModule
 * (model.py:55-63, 207-240). */
typedef struct s2s_params {
    float dwell_mean;          /* used when duration_sampling == 0 (modules.py:420-432) */
    float dwell_std;           /* > 0: dwell ~ N(dwell_mean, dwell_std), clamp(min_duration) */
    float noise_std;           /* <= 0: no noise (model.py:224) */
    float min_noise;           /* clamp of the predicted sigma (model.py:228) */
    float min_duration;        /* clamp of the dwell (modules.py:414-416, 430-432) */
    int32_t noise_sampling;    /* 1: per-sample sigma from the NoiseSampler (model.py:227-234) */
    int32_t duration_sampling; /* 1: Gamma(conc, rate) dwell from the DurationSampler (modules.py:410-416) */
    uint64_t seed;             /* key of the counter-based Philox4x32-10 generator */
} s2s_params;

/* Optional stage outputs for parity tests (all nullable, device). */
typedef struct s2s_debug {
    float* emb_out;            /* [B][16][64]  Encoder.forward 2nd result (modules.py:72-77) */
    float* enc_out;            /* [B][16][64]  Encoder.forward 1st result (modules.py:80-89) */
    float* sigma;              /* [B][16]      NoiseSampler.forward (modules.py:275-278) */
    float* conc;               /* [B][16]      DurationSampler conc, clamped (modules.py:215-216) */
    float* rate;               /* [B][16]      DurationSampler rate, clamped (modules.py:217-218) */
    float* g;                  /* [B][16]      the dwell value before round() (modules.py:414-416/420-432) */
    float* y_scaled;           /* [B][250]     Decoder.forward output (modules.py:140-141) */
    float* z01;                /* [B][250]     the standard normals used for the noise term */
    /* stage INPUTS (nullable): with these the launch serves as a stand-alone sub-module operator, the reference's
     * NoiseSampler(x) / DurationSampler(x) on any emb_out and Decoder(x) on any [B,250,64] tensor (modules.py:275-278, 197-225, 133-142) */
    const float* emb_in;       /* [B][16][64]  replaces the pre-net output: heads, dwell and encoder blocks run on it */
    const float* dec_in;       /* [B][250][64] replaces the length-regulated + position-encoded decoder input */
} s2s_debug;

typedef struct s2s_handle s2s_handle;

/* Number of fp32 values the weight blob must hold for `cfg`, and the order:
 *   encoders.position_enc[16*64]; src_emb.weight[64][5k], .bias[64];
 *   pre_net_stack.i.weight[64][64], .bias[64]                               (i < pre_layers)
 *   per encoder layer: LAYER (below)
 *   noise_sampler.stdv_layer: 0.weight[64][64], 0.bias[64], 3.weight[64], 3.bias[1]
 *   duration_sampler.conc_layer: same four; duration_sampler.rate_layer: same four
 *   decoders.position_enc[250*64]; per decoder layer: LAYER;
 *   decoders.out_linear.weight[64], .bias[1]
 * LAYER = slf_attn.{w_qs,w_ks,w_vs}.{weight[64][64],bias[64]}, slf_attn.fc.{weight,bias},
 *         slf_attn.layer_norm.{weight,bias}[64], pos_ffn.w_1.{weight[256][64],bias[256]},
 *         pos_ffn.w_2.{weight[64][256],bias[64]}, pos_ffn.layer_norm.{weight,bias}[64]
 * i.e. the reference state_dict (SURVEY.md section 8 a-W) in its native [out][in] layouts. */
size_t s2s_blob_floats(const s2s_config* cfg);

/* Replaces seq2squiggle.load_from_checkpoint + .to(device) (inference.py:386-399): takes the
 * host fp32 blob, re-packs it into MFMA fragment order and uploads it to `device`. */
int s2s_create(const s2s_config* cfg, const void* weights_blob, size_t blob_bytes, int device,
               s2s_handle** out);
void s2s_destroy(s2s_handle* h);

/* Last error of this handle (or of the last failed s2s_create when h == NULL, thread-local). */
const char* s2s_last_error(const s2s_handle* h);

/* Replaces seq2squiggle.predict_step up to the clamp (model.py:195-240) for B chunks.
 *
 *  bases    device [B][16+k-1] ASCII: chunk c of a read is read[16c : 16c+15+k]
 *           (split_sequence, utils.py:350-356); bytes past the read's end are ignored;
 *  n_valid  device [B], 1..16: k-mers j >= n_valid are the all-"_" pad k-mer
 *           (add_remainder, utils.py:342-347), not a shifted window;
 *  first_global_chunk  index of chunk 0 in the whole job: the RNG counter is
 *           (first_global_chunk + b, position, draw kind), so results do not depend on batch
 *           size or on how chunks are sharded over GPUs;
 *  inject_g   nullable device [B][16]: value of Gamma.sample() (modules.py:221-222) to use
 *             instead of the built-in sampler (duration_sampling only);
 *  inject_zdw nullable device [B][16]: standard normals for the dwell_std > 0 mode;
 *  inject_z01 nullable device [B][250]: standard normals for the noise term;
 *  out_signal device [B][250] fp32 pA (rows of `prediction`, model.py:240);
 *  out_dur    device [B][16] int32 rounded dwell (modules.py:437-438);
 *  dbg        nullable.
 */
int s2s_predict_chunks(s2s_handle* h, void* stream, const uint8_t* bases, const uint8_t* n_valid,
                       int64_t first_global_chunk, int32_t B, const s2s_params* params,
                       const float* inject_g, const float* inject_zdw, const float* inject_z01,
                       float* out_signal, int32_t* out_dur, const s2s_debug* dbg);

/* Same as s2s_predict_chunks without materialising the chunk windows: `read_bytes` (device) holds whole reads back
 * to back, each padded with '_' to 16*C + k - 1 bytes (C = its chunk count); chunk b is the 16+k-1 bytes at
 * read_bytes + chunk_start[b] (device int64 [B]).  This is what process_read/split_sequence (dataloader.py:358-398,
 * utils.py:350-356) produce, minus the one-hot: adjacent chunks of a read overlap by k-1 bytes in place. */
int s2s_predict_packed(s2s_handle* h, void* stream, const uint8_t* read_bytes, const int64_t* chunk_start,
                       const uint8_t* n_valid, int64_t first_global_chunk, int32_t B, const s2s_params* params,
                       float* out_signal, int32_t* out_dur);

/* Replaces the per-read cat + zero-strip of export_and_clear_results (model.py:284-286) and the
 * pA -> int16 conversion of BLOW5Writer/POD5Writer.save (signal_io.py:134-141, 246-253).
 *
 *  signal       device [B][250] (output of s2s_predict_chunks), chunks of a read contiguous;
 *  read_first   device [R+1]: read r owns chunks read_first[r] .. read_first[r+1]-1;
 *  out_offsets  device [R+1] int64: on return out_offsets[r] is the start of read r in the
 *               packed output, out_offsets[R] the total sample count;
 *  out_pa       nullable device fp32 [capacity]: packed non-zero samples (the tensors handed to
 *               writer.signals, model.py:290);
 *  out_dac      nullable device int16 [capacity]: round_half_even(pa*digitisation/range - offset)
 *               wrapped to int16; reversed per read when rna != 0 (signal_io.py:140-141);
 *  capacity     size of out_pa/out_dac in samples (B*250 always suffices).
 */
int s2s_export_reads(s2s_handle* h, void* stream, const float* signal, int32_t B,
                     const int32_t* read_first, int32_t R, int64_t* out_offsets, float* out_pa,
                     int16_t* out_dac, int64_t capacity, float digitisation, float range,
                     float offset_mean, int32_t rna);

/* Replaces the signal compression that pyslow5.write_record_batch (svb-zd) and pod5.Writer.add_reads (the svb16 stage of
 * VBZ) run on the host (reference signal_io.py:167-171, 268-282): StreamVByte encoding of the zig-zag deltas of the packed
 * int16 samples, one output blob per row, so that only ~1.1 bytes per sample cross PCIe.
 *
 *  samples       device int16: out_dac of s2s_export_reads;
 *  read_offsets  device [R+1]: out_offsets of s2s_export_reads;
 *  row_read, row_index  device [N]: row i covers samples [row_index[i]*row_samples, (row_index[i]+1)*row_samples) of read
 *               row_read[i], clipped to the read; a row that starts past the end of its read yields an empty blob (the
 *               caller enumerates candidate rows from the chunk counts without knowing the stripped lengths);
 *  row_samples   samples per row: 102400 for POD5 signal-table rows, any value >= the longest read for whole reads (SLOW5);
 *  variant       32: slow5 svb-zd blob = u32 n, (n+3)/4 control bytes (2 bits per value), data (zig-zag of 32-bit deltas);
 *               16: pod5 svb16 stream = (n+7)/8 control bytes (1 bit per value), data (zig-zag of 16-bit deltas): the
 *               input of the row's zstd frame;
 *  out           device bytes [capacity]; a row of n samples takes at most 4 + ceil(n/4) + 3n bytes (variant 32: a zig-zag
 *               delta of two int16 samples can need 3 bytes) or ceil(n/8) + 2n (variant 16), so capacity >=
 *               4N + 3 total + (total + 3N)/4 (variant 32) or 2 total + total/8 + N (variant 16) always suffices;
 *  out_offsets   device [N+1] int64: blob i = out[out_offsets[i] : out_offsets[i+1]].  If the rows need more than
 *               `capacity` bytes, the rows that do not fit are NOT written and out_offsets[N] comes back NEGATIVE
 *               (minus the bytes needed): the caller must check it before framing the blobs.
 */
int s2s_svb_encode(s2s_handle* h, void* stream, const int16_t* samples, const int64_t* read_offsets,
                   const int32_t* row_read, const int32_t* row_index, int32_t N, int64_t row_samples, int32_t variant,
                   uint8_t* out, int64_t capacity, int64_t* out_offsets);

/* Test hook: raw Philox4x32-10 words, out[i*4..i*4+3] = philox(counter = {c0+i, c1, c2, c3}, key = seed). */
int s2s_philox_u32(s2s_handle* h, void* stream, uint64_t seed, uint32_t c0, uint32_t c1, uint32_t c2,
                   uint32_t c3, int32_t n, uint32_t* out /* device [n][4] */);

/* Device time (ms) of the predict kernel (s2s_fused_kernel: frontend + decoder in one launch) over the launches since the last
 * call, measured with HIP events on the launch stream when profiling is enabled. */
int s2s_set_profiling(s2s_handle* h, int32_t enabled);
int s2s_get_kernel_ms(s2s_handle* h, double* kernel_ms_total, int64_t* launches, int64_t* chunks);

/* ---- host-side helper (no GPU work, no handle): frames and compresses one batch of BLOW5 records on `threads` worker
 * threads -- what pyslow5's write_record_batch(threads = cpu_count) does for the reference (signal_io.py:167-171).
 * Record i's body is prefix[prefix_offs[i] : prefix_offs[i+1]] + signal[signal_offs[i] : signal_offs[i+1]] +
 * suffix[suffix_offs[i] : suffix_offs[i+1]] (the fields before and after raw_signal, built by the caller; the signal bytes
 * are little-endian int16 samples or an svb-zd blob); `out` receives, in record order, [u64 compressed size][body compressed
 * with `method`: 0 none, 1 zlib container (RFC 1950; written by libdeflate when that library loads, else zlib), 2 zstd, 3 zlib
 * container written by the library's own Huffman-only deflate encoder (dynamic blocks of literals, no match search; a record is
 * coded in independent pieces of <= 128 KiB on several threads and is still ONE ordinary zlib stream; `level` is ignored)] at
 * `level`.  `capacity` must be at least s2s_blow5_pack_bound(total body bytes, n).  Returns the bytes written, or < 0. */
int64_t s2s_blow5_pack_bound(int64_t body_bytes_total, int32_t n_records);
int64_t s2s_blow5_pack(const uint8_t* prefix, const int64_t* prefix_offs, const uint8_t* suffix, const int64_t* suffix_offs,
                       const uint8_t* signal, const int64_t* signal_offs, int32_t n_records, int32_t method, int32_t level,
                       int32_t threads, uint8_t* out, int64_t capacity);

/* The same worker threads on plain byte rows: row i = in[in_offs[i] : in_offs[i+1]] is compressed on its own (method 1: zlib
 * container, 2: one zstd frame -- the second stage of POD5's VBZ after s2s_svb_encode's svb16 stream, what pod5.Writer.add_reads
 * runs per signal row, reference signal_io.py:268-282) and the results are laid back to back in `out` with their bounds in
 * out_offs [n+1].  capacity >= s2s_blow5_pack_bound(total bytes, n) always suffices.  Returns the bytes written, or < 0. */
int64_t s2s_compress_rows(const uint8_t* in, const int64_t* in_offs, int32_t n_rows, int32_t method, int32_t level,
                          int32_t threads, uint8_t* out, int64_t capacity, int64_t* out_offs);

/* ---- host-side helper (no GPU work, no handle): replays the DRAWS of the reference's read sampler (utils.py:415-479
 * `sampling`, with the read-length law of utils.py:325-331 `draw_expon_dis`) without building a read, so that a rank of a sharded
 * run finds the generator state of its first read in microseconds per thousand reads instead of replaying them in the
 * interpreter.  Per attempt the reference draws a start position (random.randint on the genome-wide coordinate), a strand
 * (random.choice("+-"), DNA profiles only) and, for an accepted read, one random.choice("ACGT") per N in it; the length of
 * attempt (read_i, retry) is scipy's expon law seeded with seed + read_i * (max_retries + 1) + retry, truncated and clipped to
 * [1, total_len]; an attempt is rejected when a DNA read is cut short by its contig's end, is shorter than min_read_len or
 * holds more than 10 % N (utils.py:381-400).
 *
 *  mt_state       in/out [625]: Python's random.getstate()[1] (624 Mersenne-Twister words + the index); on return the state
 *                 in front of the first draw of read *out_next_read_i;
 *  contig_ends    [n_contigs] running sum of the contig lengths (genome < 2^31 bases);
 *  n_pos, n_pos_count   nullable [n_contigs]: per contig the sorted offsets of its N bases (NULL / 0 for a contig without N);
 *  num_seqs, first_read_i   reads first_read_i .. num_seqs-1 are attempted in order;
 *  stop_after     stop once this many reads have been accepted (< 0: never);
 *  out_lengths    nullable [>= number of accepted reads]: their lengths.
 * Returns the number of accepted reads, or < 0 (S2S_ERR_ARG: also when seed + num_seqs * (max_retries + 1) >= 2^32, where
 * the reference's scipy seed would leave the range this fast path mirrors -- the caller then replays in Python). */
int64_t s2s_sampler_replay(uint32_t* mt_state, const int64_t* contig_ends, int32_t n_contigs,
                           const int64_t* const* n_pos, const int64_t* n_pos_count, int64_t num_seqs, int64_t first_read_i, int64_t r,
                           uint64_t seed, int64_t total_len, int32_t is_dna, int32_t min_read_len, int32_t max_retries,
                           int64_t stop_after, int32_t* out_lengths, int64_t* out_next_read_i);

/* The same replay for any of the reference's three read-length laws (utils.py:311-331; --distr): law 0 = expon (as above), 1 = gamma
 * (scipy gamma.rvs(6.3693711, loc = 0.53834893) * r / 4.39), 2 = beta (beta.rvs(1.778, 7.892, loc = 316.758, scale = 34191.257) * r /
 * 6615): numpy's legacy samplers -- Marsaglia-Tsang on the polar-method normal, whose cached second variate carries over between
 * the two gammas of a beta -- mirrored draw for draw on a generator seeded per (read, retry).  s2s_length_law: one such length
 * (test hook). */
int64_t s2s_sampler_replay_law(uint32_t* mt_state, const int64_t* contig_ends, int32_t n_contigs,
                               const int64_t* const* n_pos, const int64_t* n_pos_count, int64_t num_seqs, int64_t first_read_i, int64_t r,
                               uint64_t seed, int64_t total_len, int32_t is_dna, int32_t min_read_len, int32_t max_retries,
                               int64_t stop_after, int32_t law, int32_t* out_lengths, int64_t* out_next_read_i);
int64_t s2s_length_law(int32_t law, uint32_t seed, double r, int64_t total_len);

/* Plain FASTA text -> cleaned sequences on the host (no GPU): what utils.read_fasta (pysam.FastxFile, utils.py:290-308) and
 * process_genome (upper-case, non-ACGT -> N: utils.py:594-597) do line by line in the interpreter; every rank of a sharded
 * run parses the whole reference before its first kernel.  s2s_fasta_count: number of records ('>' first on a line), -2 when
 * the first non-blank line starts with '@' (FASTQ: s2s_fastq_clean), -3 when a line holds a lone carriage return (a line break
 * of its own for the reference's reader: the caller's line loop decides).  s2s_fasta_clean: out (>= n bytes) receives the sequences back to
 * back, line ends removed and lines stripped of blanks; map_acgtn = 1 applies process_genome's mapping; seq_offs [records+1]
 * delimits them; name_span [2*records]: begin / end of each record's name (first token of its header) inside data.
 * Returns the number of records, -1 when there are more than max_records. */
int64_t s2s_fasta_count(const uint8_t* data, int64_t n);
/* Four-line FASTQ records the same way (--read-input files; pysam.FastxFile reads either format): `out` receives the sequence
 * lines as they stand minus their line ends (map_acgtn as above); -2 for anything the caller's line loop must judge itself (no '@'
 * where a header should be, a missing '+' line, a truncated record, a lone carriage return inside a line); out == NULL with
 * max_records == 0 only counts the records. */
int64_t s2s_fastq_clean(const uint8_t* data, int64_t n, int32_t map_acgtn, uint8_t* out, int64_t* seq_offs,
                        int64_t* name_span, int64_t max_records);
int64_t s2s_fasta_clean(const uint8_t* data, int64_t n, int32_t map_acgtn, uint8_t* out, int64_t* seq_offs,
                        int64_t* name_span, int64_t max_records);

/* ---- host-side helpers of the rank-shard merge (no GPU work, no handle; `predict --gpus N` / `merge-shards`): the reference
 * leaves ONE output file (inference.py:65-79, signal_io.py:167-171, 268-282), a sharded run one per rank.
 * s2s_copy_ranges copies n byte ranges (src_fd[i], src_off[i], len[i]) -> (dst_fd[i], dst_off[i]); ranges must not overlap inside
 * one file.  engine 0: copy_file_range on the descriptors (in the kernel, no user-space buffer; pread / pwrite through a bounce
 * buffer where the file system refuses it), ONE writer whatever `threads` says -- buffered writes of one file take its inode lock and
 * more writers are slower (profiles/r05/fs_write_probe_shm.txt: 6.5 GB/s with one, 3.2-4.1 GB/s with 2-8).  engine 1: where the
 * destination is on tmpfs, its ranges are allocated first (posix_fallocate: 18.6 GB/s on that box) and then filled on `threads`
 * threads, each pread()ing the source into a populated shared mapping of the destination (pages that exist: no lock, no
 * allocation; 8.0 against 5.7 GB/s for engine 0); any other file system, a refused allocation or mapping: engine 0.  engine 2: the
 * same on any file system (A/B: it loses on a disk's page cache).  Returns the bytes copied, or < 0 (S2S_ERR_ARG, or -errno of the
 * failing call; -EIO when a source is shorter than its range).
 * s2s_blow5_scan walks the [u64 size][body] records of a BLOW5 file between byte offsets begin and end (the end of the
 * header and the start of the end-of-file marker) reading the size prefixes only: the record count, or -2 when the chain of
 * sizes does not end exactly at `end` (a truncated shard). */
int64_t s2s_copy_ranges(int32_t n, const int32_t* src_fd, const int64_t* src_off, const int32_t* dst_fd, const int64_t* dst_off,
                        const int64_t* len, int32_t threads, int32_t engine);
int64_t s2s_blow5_scan(int32_t fd, int64_t begin, int64_t end);
/* ... and over a file that is still growing (the live join, seq2squiggle_amd/merge.py: LiveJoin): the complete records from `begin`
 * that end at or before `limit` (the file's size now), at most max_records; *out_end = the offset behind the last of them.  An
 * incomplete record (or the 5-byte end marker) ends the walk without an error.  Returns the number of records, or S2S_ERR_ARG. */
int64_t s2s_blow5_scan_upto(int32_t fd, int64_t begin, int64_t limit, int64_t max_records, int64_t* out_end);

/* Which softmax path the split-f16 decoder attention (S2S_MODE_F16X3 / S2S_MODE_F16) runs (layers.py:20-40 is one unmasked
 * softmax over 250 keys; both paths compute it within the parity bound, with the same error against an fp64 evaluation):
 *   0  the FAST path: shift = the row's maximum over the keys of PASS 0 + 2 log2 units, no maximum in later passes, straight-line
 *      code.  Pass 0 is a SAMPLE of the whole row -- the fast instance stores the K / V^T images with the blocks of four consecutive
 *      keys dealt out over the four 64-key passes, so pass 0 holds the key blocks b = 0 (mod 4).  A head whose later keys beat that
 *      shift by more than the f16 range shows in its row sum and is redone by an out-of-line online softmax.  Best for diffuse
 *      attention (the committed synthetic checkpoints redo nothing; their decoder w_qs / w_ks x 2: 6.4 % of the heads);
 *   1  the EXACT path at once: the online softmax (running row maximum raised and sums rescaled in every 64-key pass, no branch;
 *      the shift rides in the score MFMA's k-slots, a pass's row maxima come from its first score MFMA alone, natural key
 *      order) as the only path of its own kernel instance: 201.6 k shader cycles per chunk and CU against the fast path's 190.2 k
 *      on diffuse attention (6.0 % more), the same for ANY weights -- "fast, then redo" costs 310-328 k once most heads overflow
 *      (sharply peaked attention).
 * s2s_create chooses by a calibration launch on a fixed pseudo-random batch: exact when more than S2S_ATTENTION_REDO_THRESHOLD of
 * its heads had to be redone (where the two cost the same).  The environment variable S2S_ATTENTION_PATH = fast | exact (any
 * case) skips the launch, auto or empty means calibrate, anything else makes s2s_create fail.  `calibration_redo_rate` returns
 * the launch's share (-1 when no calibration ran).  Results are deterministic per chunk for a given path. */
#define S2S_ATTENTION_REDO_THRESHOLD 0.055
double s2s_attention_redo_threshold(void);
int s2s_set_attention_path(s2s_handle* h, int32_t path);
int s2s_get_attention_path(const s2s_handle* h, int32_t* path, double* calibration_redo_rate);

/* Counters of the predict kernel since the last call (every build; synchronises the device, then resets them).  The fast
 * softmax of the split-f16 decoder is data dependent -- a head whose later keys beat the maximum over its pass-0 key sample (the
 * key blocks b = 0 mod 4, see s2s_set_attention_path) by more than the f16 range is redone on a safe path -- and the chip's clock
 * under this kernel depends on the operands, so a throughput
 * figure is a statement about one set of weights: these counters say how a run behaved.  out10 =
 *   [0] chunks launched, [1] (wave, head, layer) softmax runs (chunks x 8 waves x 8 heads x decoder layers),
 *   [2] ... of them redone on the safe path (0 in S2S_MODE_F32, which has no fast path),
 *   [3] shader-clock cycles (s_memtime) and [4] 100 MHz ticks (s_memrealtime) of one thread per workgroup over the whole
 *       kernel, summed over [5] workgroups: [3] / [4] / 10 is the clock in GHz the SIMDs really ran at,
 *   [6] chunks launched on the exact attention path (s2s_set_attention_path), [7..9] reserved (0). */
int s2s_stats_read(s2s_handle* h, uint64_t* out10);

/* Diagnostic builds (-DS2S_DIAG, never the shipped library): per wave of a workgroup (8 rows) 48 per-phase shader-cycle sums
 * since the last call, summed over the workgroups (slots 0-15 decoder phases, 16-18 whole-kernel s_memtime / s_memrealtime /
 * wave count, 32-47 the frontend's own phases; tools/diag_phases.py names them); out384 = [8][48].  S2S_ERR_ARG in a normal build. */
int s2s_diag_read(s2s_handle* h, uint64_t* out384);

#ifdef __cplusplus
}
#endif
#endif /* S2S_HIP_H */
