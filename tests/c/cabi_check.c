/* Plain-C consumer of include/s2s_hip.h: proves the boundary is a C ABI (no C++ or torch types leak).
 * Built and run by tests/test_host_cpu.py with gcc; needs no GPU (only argument-checking paths are called). */
#include <dlfcn.h>
#include <stdio.h>
#include <string.h>
#include "s2s_hip.h"

typedef size_t (*blob_floats_fn)(const s2s_config*);
typedef int (*create_fn)(const s2s_config*, const void*, size_t, int, s2s_handle**);
typedef const char* (*last_error_fn)(const s2s_handle*);
typedef int64_t (*pack_bound_fn)(int64_t, int32_t);
typedef int64_t (*pack_fn)(const uint8_t*, const int64_t*, const uint8_t*, const int64_t*, const uint8_t*, const int64_t*, int32_t,
                           int32_t, int32_t, int32_t, uint8_t*, int64_t);

int main(int argc, char** argv) {
    if (argc < 2) return 2;
    void* lib = dlopen(argv[1], RTLD_NOW | RTLD_LOCAL);
    if (!lib) { fprintf(stderr, "dlopen: %s\n", dlerror()); return 3; }
    const char* names[] = {"s2s_blob_floats", "s2s_create", "s2s_destroy", "s2s_last_error", "s2s_predict_chunks", "s2s_predict_packed",
                           "s2s_export_reads", "s2s_svb_encode", "s2s_philox_u32", "s2s_set_profiling", "s2s_get_kernel_ms", "s2s_stats_read", "s2s_set_attention_path", "s2s_get_attention_path", "s2s_diag_read",
                           "s2s_blow5_pack_bound", "s2s_blow5_pack", "s2s_compress_rows", "s2s_sampler_replay", "s2s_sampler_replay_law", "s2s_length_law",
                           "s2s_fasta_count", "s2s_fasta_clean", "s2s_fastq_clean"};
    for (unsigned i = 0; i < sizeof names / sizeof *names; ++i)
        if (!dlsym(lib, names[i])) { fprintf(stderr, "missing symbol %s\n", names[i]); return 4; }
    blob_floats_fn blob_floats = (blob_floats_fn)dlsym(lib, "s2s_blob_floats");
    create_fn create = (create_fn)dlsym(lib, "s2s_create");
    last_error_fn last_error = (last_error_fn)dlsym(lib, "s2s_last_error");

    s2s_config cfg = {9, S2S_T_ENC, S2S_T_DEC, S2S_DMODEL, S2S_DFF, S2S_HEADS, 2, 2, 1, 165.0f, S2S_MODE_F16X3};
    const size_t n = blob_floats(&cfg);
    if (n != 236804) { fprintf(stderr, "blob floats %zu\n", n); return 5; }    /* the reference's parameter count */
    cfg.seq_kmer = 6;
    if (blob_floats(&cfg) != 236804 - 64 * 15) return 6;
    cfg.n_heads = 4;
    if (blob_floats(&cfg) != 0) return 7;
    s2s_handle* h = (s2s_handle*)1;
    if (create(&cfg, 0, 0, 0, &h) != S2S_ERR_ARG || h != 0) return 8;
    if (!strstr(last_error(0), "n_heads")) return 9;
    cfg.n_heads = 8;
    float dummy[4] = {0};
    if (create(&cfg, dummy, sizeof dummy, 0, &h) != S2S_ERR_BLOB) return 10;
    /* the host-side record packer needs no GPU: two records, stored uncompressed, from plain C */
    pack_bound_fn pack_bound = (pack_bound_fn)dlsym(lib, "s2s_blow5_pack_bound");
    pack_fn pack = (pack_fn)dlsym(lib, "s2s_blow5_pack");
    const uint8_t head[] = "AB", tail[] = "xyz", sig[] = {1, 2, 3, 4, 5, 6};
    const int64_t head_offs[] = {0, 1, 2}, tail_offs[] = {0, 1, 3}, sig_offs[] = {0, 2, 6};
    uint8_t out[8192];
    if (pack_bound(12, 2) > (int64_t)sizeof out) return 11;
    const int64_t got = pack(head, head_offs, tail, tail_offs, sig, sig_offs, 2, 0, 1, 3, out, sizeof out);
    const uint8_t want[] = {4, 0, 0, 0, 0, 0, 0, 0, 'A', 1, 2, 'x', 7, 0, 0, 0, 0, 0, 0, 0, 'B', 3, 4, 5, 6, 'y', 'z'};
    if (got != (int64_t)sizeof want || memcmp(out, want, sizeof want)) return 12;
    if (pack(head, head_offs, tail, tail_offs, sig, sig_offs, 2, 0, 1, 3, out, 8) >= 0) return 13;   /* too small a buffer */
    /* the read-sampler replay from plain C: a generator state of init_genrand(5489) (the Mersenne Twister's reference seed, index
     * 624), one 1000-base contig without N, five reads of mean length 100: all accepted on some retry, state advanced */
    typedef int64_t (*replay_fn)(uint32_t*, const int64_t*, int32_t, const int64_t* const*, const int64_t*, int64_t, int64_t, int64_t,
                                 uint64_t, int64_t, int32_t, int32_t, int32_t, int64_t, int32_t*, int64_t*);
    replay_fn replay = (replay_fn)dlsym(lib, "s2s_sampler_replay");
    uint32_t mt[625];
    mt[0] = 5489u;
    for (int i = 1; i < 624; ++i) mt[i] = 1812433253u * (mt[i - 1] ^ (mt[i - 1] >> 30)) + (uint32_t)i;
    mt[624] = 624;
    const int64_t ends[] = {1000};
    int32_t lens[5];
    int64_t next_i = -1;
    if (replay(mt, ends, 1, 0, 0, 5, 0, 100, 42, 1000, 1, 30, 20, -1, lens, &next_i) != 5 || next_i != 5 || mt[624] >= 624) return 14;
    for (int i = 0; i < 5; ++i) if (lens[i] < 30 || lens[i] > 1000) return 15;
    if (replay(mt, ends, 1, 0, 0, 5, 0, 0, 42, 1000, 1, 30, 20, -1, lens, &next_i) != S2S_ERR_ARG) return 16;   /* r must be > 0 */
    /* the FASTA parser from plain C: two records, CRLF line ends, lower case and an ambiguity code mapped by process_genome's rule */
    typedef int64_t (*fcount_fn)(const uint8_t*, int64_t);
    typedef int64_t (*fclean_fn)(const uint8_t*, int64_t, int32_t, uint8_t*, int64_t*, int64_t*, int64_t);
    fcount_fn fcount = (fcount_fn)dlsym(lib, "s2s_fasta_count");
    fclean_fn fclean = (fclean_fn)dlsym(lib, "s2s_fasta_clean");
    const char fa[] = ">chr1 first\r\nACgt\r\nnR\r\n>c2\nTT\n";
    const int64_t fa_n = (int64_t)sizeof fa - 1;
    uint8_t seqs[64];
    int64_t so[3], ns[4];
    if (fcount((const uint8_t*)fa, fa_n) != 2) return 17;
    if (fclean((const uint8_t*)fa, fa_n, 1, seqs, so, ns, 2) != 2 || so[0] != 0 || so[1] != 6 || so[2] != 8) return 18;
    if (memcmp(seqs, "ACGTNNTT", 8) != 0 || ns[1] - ns[0] != 4 || memcmp(fa + ns[0], "chr1", 4) != 0 || memcmp(fa + ns[2], "c2", 2) != 0) return 19;
    if (fclean((const uint8_t*)fa, fa_n, 0, seqs, so, ns, 1) != -1) return 20;                        /* more records than room */
    if (fcount((const uint8_t*)"@q\nAC\n+\nII\n", 11) != -2) return 21;                              /* FASTQ: not for this parser */
    printf("CABI_OK %zu\n", n);
    return 0;
}
