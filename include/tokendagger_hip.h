/*
 * tokendagger_hip.h — C ABI of the MI355X-native tokenizer hot path (libtokendagger_hip.so).
 *
 * This is the drop-in boundary for the ONE path this repository accelerates: raw UTF-8 text ->
 * regex pre-tokenization -> byte-pair merge -> token ids (reference: tiktoken::CoreBPE,
 * /root/reference/src/tiktoken/tiktoken.hpp:38-88).  Plain pointers and sizes only; no torch,
 * pybind or C++ types cross it.  Every entry point names the reference interface it replaces.
 * INTEGRATION.md shows the reference-side binding (pybind11 / ctypes) a maintainer would add.
 *
 * Conventions
 *   - every function returns TD_OK (0) or a TD_E_* code; td_last_error() gives the message
 *     (the C ABI equivalent of the reference's `TiktokenError` exception, tiktoken.hpp:32-35);
 *   - text is UTF-8, documents are concatenated: document d = text[doc_offsets[d], doc_offsets[d+1]),
 *     doc_offsets[0] == 0, doc_offsets[n_docs] == total bytes, offsets are int64;
 *   - token ids are int32 and equal the `rank` values given at construction;
 *   - there is no CPU fallback: without a usable HIP device td_create fails.
 */
#ifndef TOKENDAGGER_HIP_H
#define TOKENDAGGER_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TD_OK 0
#define TD_E_INVALID 1      /* bad argument */
#define TD_E_PATTERN 2      /* pat_str uses regex syntax outside what the device pre-tokenizer implements (td_last_error says which) */
#define TD_E_VOCAB 3        /* vocabulary cannot be represented */
#define TD_E_UNKNOWN_BYTE 4 /* input contains a byte / part that is not in the vocabulary
                               (reference: TiktokenError "No value found for pair", tiktoken.cpp:364) */
#define TD_E_CAPACITY 5     /* output buffer too small */
#define TD_E_SCRATCH 6      /* long-piece scratch exhausted; raise it with td_set_option */
#define TD_E_HIP 7          /* HIP runtime failure */
#define TD_E_BAD_TOKEN 8    /* decode: id not in the vocabulary (reference: "Invalid token for decoding", tiktoken.cpp:249) */
#define TD_E_SPECIAL 9      /* allowed special token not in the special vocabulary (tiktoken.cpp:178-180) */

/* encode modes */
#define TD_MODE_ENCODE 0    /* CoreBPE::encode(text, {}): whole-piece lookup, then merge  (tiktoken.cpp:169-234) */
#define TD_MODE_ORDINARY 1  /* CoreBPE::encode_ordinary(text): merge only               (tiktoken.cpp:156-167) */

typedef struct td_tokenizer td_tokenizer;

/*
 * Replaces CoreBPE::CoreBPE(pattern, vocab, special_vocab)  (tiktoken.hpp:48-67, py_binding.cpp:22-24).
 * The vocabulary is passed as concatenated token bytes + n+1 offsets + n ranks (what a list of
 * VocabItem{rank, token_bytes} holds, tiktoken.hpp:12-16); specials likewise (token_string, rank).
 * Builds the device tables and uploads them to HIP device `device` (-1: current device).
 * pat_str (init_regex, tiktoken.cpp:47-68): the known tokenizer patterns (o200k / Llama-4, tekken, cl100k_base / Llama-3, Qwen2, GPT-2)
 * have kernels of their own; any other pattern within the backtracking subset listed in tokendagger_amd/csrc/td_regex.h is
 * compiled and matched with PCRE2's semantics (text it skips gets no tokens, tiktoken.cpp:86-122), in parallel inside a
 * document too (speculative chunks of 64 B .. 1 KiB, checked against their predecessors);
 * anything else is TD_E_PATTERN.  There is no CPU regex fallback.
 */
int td_create(const char* pat_str, int64_t n_vocab, const uint8_t* token_bytes, const int64_t* token_offsets,
              const int32_t* ranks, int64_t n_special, const uint8_t* special_bytes,
              const int64_t* special_offsets, const int32_t* special_ids, int device, td_tokenizer** out);

/* A second handle on the SAME tables: its own lock, workspace, control block and streams, but the device tables td_create
 * uploaded (and their host copies) are shared, not copied — what a host thread per HIP stream needs to run encodes
 * concurrently (the reference shares one CoreBPE between the threads of its pool, tokendagger/wrapper.py:212-235, each with
 * its own match data, tiktoken.cpp:13-45).  Options set on `src` so far are inherited.  The tables are freed with the last
 * handle that shares them; every handle is destroyed with td_destroy, in any order. */
int td_clone(td_tokenizer* src, td_tokenizer** out);

/* Replaces CoreBPE::~CoreBPE (tiktoken.hpp:69-73). */
void td_destroy(td_tokenizer* t);

/* Message of the calling thread's last failing call on this handle ("" if none; t == NULL: last td_create failure of
 * this thread).  The pointer stays valid until the same thread's next failing call.
 *
 * Threads and streams: a handle may be shared by host threads; calls serialise on one internal lock.  All calls of a
 * handle share ONE device workspace: work of a handle is ordered across streams by the library (a call on another
 * stream than the previous call's waits, on the device, for that call's kernels), so asynchronous calls on different
 * streams do not overlap each other — use one handle per stream for concurrency (td_clone: without a second copy of the tables).  Every entry point leaves the
 * caller's current HIP device as it found it.
 *
 * The legacy (null) stream: the library never uses it on its own — copies it waits for, table uploads and the host-buffer
 * entry points run on a private non-blocking stream of the handle, results come back through pinned memory — so it neither
 * synchronises with the application's blocking streams nor breaks a stream capture another thread has open.  Kernels are
 * launched on the null stream only where the CALLER passes hip_stream == NULL to a *_device entry point.  (Workspace growth
 * calls hipMalloc / hipFree: td_reserve up front if the application captures in global mode.) */
const char* td_last_error(const td_tokenizer* t);

/*
 * Bulk encode from HOST buffers: the batched form of CoreBPE::encode / encode_ordinary over n_docs
 * independent documents (the reference reaches the same through a thread pool over single calls,
 * tokendagger/wrapper.py:212-235).  out_tokens (capacity out_capacity ids) receives all ids,
 * out_offsets[n_docs+1] the per-document token offsets, *n_tokens the total.  If the capacity is
 * too small the call fails with TD_E_CAPACITY and *n_tokens holds the required size.
 * Synchronous.  Three paths by size: inputs of at most 4 KiB (and 1024 documents) take ONE kernel launch that reads and writes pinned
 * host memory directly (TD_OPT_SMALL_PATH); up to 4 MiB (and 262 144 documents) text and offsets travel through one pinned buffer and one
 * asynchronous copy, the step's last kernels write ids and offsets straight into pinned host memory, and the host waits on a sequence
 * number instead of stream synchronisations (TD_MID_PATH=0 in the environment at td_create time: the copy-and-synchronise path that
 * inputs between 4 and 32 MiB still take); from half of TD_OPT_PIPE_CHUNK_BYTES on (default: 32 MiB) a four-slot pipeline of pinned bounce
 * buffers (host copy || H2D || kernels || ids out by a kernel || host copy, TD_OPT_PIPE_*: about 4 x (64 + 256) MiB of device memory
 * plus the pinned buffers while such calls are made).
 */
int td_encode_batch(td_tokenizer* t, const uint8_t* text, const int64_t* doc_offsets, int64_t n_docs, int mode,
                    int32_t* out_tokens, int64_t out_capacity, int64_t* out_offsets, int64_t* n_tokens);

/*
 * The same on DEVICE-resident buffers, asynchronously on `hip_stream` (a hipStream_t; NULL = the
 * default stream): d_text[n_bytes] (uint8), d_doc_offsets[n_docs+1] (int64), d_out_tokens[out_capacity]
 * (int32), d_out_offsets[n_docs+1] (int64; element n_docs = total token count).  Nothing is
 * synchronised; call td_device_status() after the stream has drained to learn about device-side
 * errors.  Workspace is (re)allocated on demand — call td_reserve() first to keep hipMalloc out of
 * a timed or captured region.
 */
int td_encode_device(td_tokenizer* t, const void* d_text, int64_t n_bytes, const void* d_doc_offsets,
                     int64_t n_docs, int mode, void* d_out_tokens, int64_t out_capacity, void* d_out_offsets,
                     void* hip_stream);

/* Pre-allocates workspace for inputs up to max_bytes / max_docs. */
int td_reserve(td_tokenizer* t, int64_t max_bytes, int64_t max_docs);

/* Synchronises `hip_stream` and returns the device error raised by calls made since the last status
 * check (TD_OK if none); *err_pos (optional) receives the byte offset it refers to. */
int td_device_status(td_tokenizer* t, void* hip_stream, int64_t* err_pos);

/*
 * Replaces CoreBPE::decode_bytes(tokens) (tiktoken.cpp:236-255, py_binding.cpp:40-44) for host
 * buffers: ids -> concatenated bytes.  *n_bytes receives the size (also on TD_E_CAPACITY).
 */
int td_decode_bytes(td_tokenizer* t, const int32_t* tokens, int64_t n_tokens, uint8_t* out, int64_t out_capacity,
                    int64_t* n_bytes);

/*
 * Replaces CoreBPE::encode(text, allowed_special) with a NON-empty allowed set (tiktoken.cpp:169-234,
 * find_next_special_token :130-154) with tiktoken semantics: the text is cut at the earliest
 * occurrences of allowed special strings, ordinary segments go through the kernels, each special
 * contributes its id.  allowed_ids lists the special ids that are allowed.
 * *last_piece_token_len (optional) is the second element of the reference's return pair.
 */
int td_encode_with_special(td_tokenizer* t, const uint8_t* text, int64_t n_bytes, const int32_t* allowed_ids,
                           int64_t n_allowed, int32_t* out_tokens, int64_t out_capacity, int64_t* n_tokens,
                           int32_t* last_piece_token_len);

/* The same over a batch of documents: every document is cut at its allowed special tokens (one pass, all literals at
 * once), all ordinary segments of all documents go to the GPU as ONE batch, ids and per-document offsets come back
 * stitched.  *n_tokens = ids needed (also on TD_E_CAPACITY). */
int td_encode_batch_with_special(td_tokenizer* t, const uint8_t* text, const int64_t* doc_offsets, int64_t n_docs,
                                 const int32_t* allowed_ids, int64_t n_allowed, int32_t* out_tokens, int64_t out_capacity,
                                 int64_t* out_offsets, int64_t* n_tokens);

/* The same two calls with the allowed set given as the special-token STRINGS themselves (concatenated UTF-8 bytes +
 * n_allowed+1 offsets), which is how tiktoken's `allowed_special` names them: exactly the listed literals are cut
 * out.  (With ids, every special string that carries a listed id is allowed — two strings may share one id.)
 * A string that is not a special token fails with TD_E_SPECIAL, like tiktoken.cpp:178-180. */
int td_encode_with_special_strs(td_tokenizer* t, const uint8_t* text, int64_t n_bytes, const uint8_t* allowed_bytes,
                                const int64_t* allowed_offsets, int64_t n_allowed, int32_t* out_tokens, int64_t out_capacity,
                                int64_t* n_tokens, int32_t* last_piece_token_len);
int td_encode_batch_with_special_strs(td_tokenizer* t, const uint8_t* text, const int64_t* doc_offsets, int64_t n_docs,
                                      const uint8_t* allowed_bytes, const int64_t* allowed_offsets, int64_t n_allowed,
                                      int32_t* out_tokens, int64_t out_capacity, int64_t* out_offsets, int64_t* n_tokens);

/* Introspection (tests, benchmarks). */
#define TD_INFO_N_PAIRS 1        /* entries of the (id,id)->rank pair table */
#define TD_INFO_MERGE_CLOSED 2   /* 1 if encode == encode_ordinary for every input with this vocab */
#define TD_INFO_MAX_ID 3
#define TD_INFO_TILE_BYTES 4
#define TD_INFO_WORKSPACE_BYTES 5
#define TD_INFO_N_SPECIAL 6
#define TD_INFO_LONG_PIECES 7    /* long pieces seen by the last td_encode_batch call */
#define TD_INFO_FAR_PIECES 8     /* pieces whose end the pre-tokenizer's window could not see (td_split_far_pieces), last call */
#define TD_INFO_DEFERRED_TILES 9 /* token tiles (4 KiB) the fused tile loop left to td_probe_tiles, last call (as of the last td_device_status) */
#define TD_INFO_FLAGGED_TILES 10 /* token tiles whose missed pieces went to td_merge_pieces, last call */
#define TD_INFO_LB_TIMEOUTS 12   /* ... and tiles it staged because their output base was not known in time (bounded look-back), last call */
#define TD_INFO_REPEATS 13       /* missed pieces whose ids were taken from another piece with the same bytes (TD_OPT_DEDUPE), last call */
#define TD_INFO_LISTED_PIECES 14 /* ... and missed pieces of the same tiles that were merged themselves, last call */
#define TD_INFO_CHAR_SEEDS 15    /* characters of 2..3 bytes this vocabulary allows to enter the merge of a long piece as one part (td_common.h) */
#define TD_INFO_SPARSE 16        /* 1: the last td_encode_device call of the handle took the sparse launch sequence (TD_OPT_SPARSE), 0: the dense one */
#define TD_INFO_DIRECT_TILES 11  /* pre-tokenizer tiles (8 KiB) whose ids the fused tile loop wrote straight to the output (TD_OPT_DIRECT), last call */
int64_t td_info(const td_tokenizer* t, int what);

/*
 * CoreBPE::encode(text, allowed_special) (tiktoken.cpp:169-234) on DEVICE-RESIDENT text, asynchronous like td_encode_device: the
 * allowed special tokens (by id) are searched for and cut out on the device — scanning forward, the longest allowed literal
 * that starts at a position is replaced by its id, the text between two cuts is tokenized as a subject of its own
 * (td_special.hip; the same results as td_encode_batch_with_special, whose search runs on host threads).  One allowed set is
 * kept on the device per handle; a call with a different set first waits for the previous call's kernels.  Patterns of the
 * scanner family only (generic patterns: TD_E_PATTERN).  Limit of the device search: an allowed literal of more than 48 bytes
 * fails with TD_E_INVALID and a message that says so (TD_E_SPECIAL stays "no such special token"); the host-buffer entry
 * points (td_encode_batch_with_special*) have no such limit — they search on host threads then.
 */
int td_encode_device_with_special(td_tokenizer* t, const void* d_text, int64_t n_bytes, const void* d_doc_offsets, int64_t n_docs,
                                  const int32_t* allowed_ids, int64_t n_allowed, void* d_out_tokens, int64_t out_capacity,
                                  void* d_out_offsets, void* hip_stream);

/* Options. */
#define TD_OPT_LONG_POOL_BYTES 1 /* scratch for pieces longer than 64 bytes (default max(64 MiB, 2 x input)) */
#define TD_OPT_PROFILE 2         /* 1: bracket the kernels of every td_encode_device call with HIP events on the
                                    call's stream (td_profile_read_ex) */
#define TD_OPT_PIPE_CHUNK_BYTES 3 /* td_encode_batch cuts inputs of at least HALF this many bytes (default 64 MiB) into chunks of whole
                                    documents of about this size — a sixth of the input, down to an eighth of the size, when the input
                                    is less than six chunks — and overlaps host copies, PCIe transfers (both directions) and kernels */
#define TD_OPT_SMALL_PATH 5       /* 0: never take the one-launch path for inputs of at most 4 KiB (default 1: on) */
#define TD_OPT_FUSED 6            /* 0: pre-tokenizer and lookup as two kernels, two passes over the text (default 1: one fused pass;
                                    TD_FUSED=0 in the environment at td_create time also turns it off).  Same results either way. */
#define TD_OPT_GRAPH 7            /* 1: replay a repeated td_encode_device call as a hipGraph (default 0: off; TD_GRAPH=1 in the environment
                                    at td_create time also turns it on): the second identical call in a row captures the step's launches on
                                    a PRIVATE non-blocking stream of the handle (the caller's stream is never put into capture), the
                                    following ones are one hipGraphLaunch on the caller's stream.  A capture that fails is ended and the
                                    step is launched kernel by kernel; same results either way. */
#define TD_OPT_DEVICE_SPECIALS 8  /* 0: td_encode_batch_with_special* always search for the allowed specials on host threads (default 1:
                                    batches of a MiB and more search on the device, td_special.hip; same results) */
#define TD_OPT_DIRECT 9            /* 0 (default): every tile's ids are staged and packed (rounds 1-3).  1 (measured slower): the fused tile loop
                                    writes a tile's ids straight to the output when the tile's base is known in time (decoupled look-back
                                    over per-tile id counts; tiles that are not — and everything behind a tile whose count cannot be settled
                                    inside the loop — are staged as before).  Same results either way; TD_DIRECT=1 in the environment at
                                    td_create time also turns it on. */
#define TD_OPT_PACK_SPLIT 10       /* 1 (default): the ids are placed by a pair of kernels — td_pack_plain, a wavefront per tile that has no long
                                    piece and at most a few merged ones (nearly a copy: 5 TB/s), and td_pack_rest for the others; 0: one
                                    kernel for all tiles (td_pack_tokens, rounds 2-4).  Same results either way; TD_PACK_SPLIT=0 in the
                                    environment at td_create time also turns it off. */
#define TD_OPT_DEDUPE 11           /* 1 (default): a piece that is not a token is merged ONCE per call however often its bytes occur in it — the
                                    pieces of the tiles with many missed pieces are looked up in a table of the call's distinct ones (bytes
                                    compared, not hashes) and the repeats copy the ids; 0: every piece is merged (rounds 1-4).  Same results
                                    either way (bpe_merge reads nothing but the piece, tiktoken.cpp:298-368); TD_DEDUPE=0 in the environment at
                                    td_create time also turns it off. */
#define TD_OPT_OVERLAP 12          /* 1 (default): once the handle has seen long pieces (>= 2048 in the last call whose counters were read:
                                    td_device_status and every host-buffer entry point read them), the kernels of the pieces above 64 bytes run on
                                    a second stream of the handle beside the kernels of the shorter ones (fork behind the lookups, join in front of
                                    the scan; parallel branches when the step is captured into a graph); 0: always one kernel after the other.
                                    Same results either way; TD_OVERLAP=0 in the environment at td_create time also turns it off. */
#define TD_OPT_GIANT_COOP_MIN 13   /* bytes (default 16384, at least 1024): a single piece above this length is merged by ALL workgroups of
                                    td_giant_pieces together (grid barriers between the sweeps; a megabyte of random letters: 0.55 s on one
                                    workgroup); shorter pieces above 1 KiB get a workgroup each as before.  Same results either way;
                                    TD_GP_COOP_MIN in the environment at td_create time sets it too. */
#define TD_OPT_SPARSE 14           /* The launch sequence of a step.  -1 (default): chosen by the counters of the last call that were read
                                    (td_device_status and every host-buffer entry point read them): text that leaves the kernels for far
                                    pieces, deferred and flagged tiles and long pieces (nearly) idle — plain prose — takes the SPARSE
                                    sequence, six launches (td_prepare_mark, td_split_tiles, td_tail, td_giant_scan, td_pack_plain,
                                    td_pack_rest); other text, and a handle whose counters nobody has read yet, the DENSE one, where those
                                    kernels are launches of their own at their own occupancies.  1: always sparse, 0: always dense.  Same
                                    results either way — td_tail walks every phase the dense sequence has kernels for; TD_SPARSE in the
                                    environment at td_create time sets it too. */
#define TD_OPT_PIPE_THREADS 4     /* host threads that fill / drain the pinned bounce buffers of that pipeline (default 16) */
int td_set_option(td_tokenizer* t, int what, int64_t value);

/* Sums (ms) of the pre-tokenizer kernel and token kernel (probe + merge) durations and the number of calls recorded
 * since the last read (TD_OPT_PROFILE); synchronises the recorded events. */
int td_profile_read(td_tokenizer* t, double* split_ms_sum, double* encode_ms_sum, int64_t* launches);
/* The same per kernel segment: ms_sums[i] = summed duration of segment i (td_profile_segment_name(i); "" past the last
 * one) over the calls recorded since the last read. */
int td_profile_read_ex(td_tokenizer* t, double* ms_sums, int n_segments, int64_t* launches);
const char* td_profile_segment_name(int i);

/* Special-token table access: replaces CoreBPE::special_tokens() (tiktoken.cpp:258-265). */
int64_t td_special_count(const td_tokenizer* t);
int td_special_get(const td_tokenizer* t, int64_t i, const char** str, int64_t* len, int32_t* id);

/* Batch decode (replaces the thread pool of Tokenizer.decode_batch, tokendagger/wrapper.py:237-256): the ids of all
 * documents concatenated + tok_offsets[n_docs+1] -> the bytes of all documents concatenated + out_offsets[n_docs+1],
 * one device pass.  *n_bytes = total bytes (also on TD_E_CAPACITY). */
int td_decode_batch(td_tokenizer* t, const int32_t* tokens, const int64_t* tok_offsets, int64_t n_docs, uint8_t* out,
                    int64_t out_capacity, int64_t* out_offsets, int64_t* n_bytes);

/* Device-resident decode: d_tokens int32[n_tokens] -> d_out bytes (capacity out_capacity), total byte count to
 * *d_n_bytes (device int64, may be NULL).  Asynchronous on hip_stream; an id outside the vocabulary (TD_E_BAD_TOKEN,
 * position = its index) or a too small d_out (TD_E_CAPACITY, position = bytes needed) surface through
 * td_device_status.  Same semantics as td_decode_bytes / CoreBPE::decode_bytes (tiktoken.cpp:236-255). */
int td_decode_device(td_tokenizer* t, const void* d_tokens, int64_t n_tokens, void* d_out, int64_t out_capacity,
                     void* d_n_bytes, void* hip_stream);

/* Vocabulary accessors (host tables, no launch): the bytes of one token id (tiktoken decode_single_token_bytes;
 * TD_E_BAD_TOKEN if the id is not in the vocabulary) and the id of one whole token (tiktoken encode_single_token:
 * regular tokens first, then special tokens; TD_E_UNKNOWN_BYTE if the bytes are not a token). */
int td_token_bytes(const td_tokenizer* t, int32_t id, const uint8_t** bytes, int64_t* len);
int td_single_token(const td_tokenizer* t, const uint8_t* bytes, int64_t len, int32_t* id);

/* ---- vocabulary files (host only; no GPU needed) ------------------------------------------------------------
 * Replaces the reference's loaders, which live in its demo and in Python: LoadBPEFile / LoadTokenizer
 * (src/main.cpp:70-137), load_bpe_vocab / load_special_tokens / load_mistral_config (tests/throughput_test.py:
 * 106-180) and Tokenizer._load_vocab_file / _load_special_tokens_file (tokendagger/wrapper.py:116-134).
 * A td_vocab accumulates regular tokens, special tokens and (tekken only) the split pattern; every loader appends.
 * All return TD_OK, TD_E_INVALID (bad argument) or TD_E_VOCAB (unreadable / malformed file: td_vocab_error). */
typedef struct td_vocab td_vocab;
int td_vocab_create(td_vocab** out);
void td_vocab_destroy(td_vocab* v);
const char* td_vocab_error(const td_vocab* v);
/* tiktoken ".model": lines of "<base64 token bytes> <rank>" */
int td_vocab_load_tiktoken(td_vocab* v, const char* path);
/* Hugging Face tokenizer_config.json: added_tokens_decoder {"<id>": {"content": "..."}} -> special tokens;
 * also_mergeable != 0 additionally enters them as regular tokens, as the reference's tests do
 * (tests/throughput_test.py:211-213). */
int td_vocab_load_hf_special(td_vocab* v, const char* path, int also_mergeable);
/* Mistral tekken.json: config.pattern + the first default_vocab_size - default_num_special_tokens entries of
 * "vocab", id = index + default_num_special_tokens */
int td_vocab_load_tekken(td_vocab* v, const char* path);
/* the reference wrapper's JSON files; either path may be NULL */
int td_vocab_load_json(td_vocab* v, const char* vocab_json_path, const char* special_json_path);
int td_vocab_set_pattern(td_vocab* v, const char* pat_str);
const char* td_vocab_pattern(const td_vocab* v); /* "" if none was set / loaded */
/* flat views (valid until the next load / destroy): which = 0 regular, 1 special */
int td_vocab_arrays(const td_vocab* v, int which, const uint8_t** bytes, const int64_t** offsets, const int32_t** ranks,
                    int64_t* n);
/* td_create over a loaded vocabulary (pattern = td_vocab_pattern) */
int td_create_from_vocab(const td_vocab* v, int device, td_tokenizer** out);

/*
 * Multi-GPU epilogue (one process per GPU, documents sharded across ranks).  The reference parallelises over
 * independent texts on a thread pool (tokendagger/wrapper.py:231-235) and needs no exchange; across GPUs the only one is
 * the gather of every rank's {tokens, documents} -> global token / document bases, and optionally of the ids to one rank.
 * RCCL (over xGMI on an MI355X node) is opened at run time; the tokenizer library links against HIP only.
 *   td_comm_unique_id      rank 0 makes the id (ncclGetUniqueId); the caller carries its 128 bytes to the other ranks
 *   td_comm_create         ncclCommInitRank on `device` (-1: the current device); collective over all ranks
 *   td_comm_gather_counts  ncclAllGather of d_counts[2] = {tokens, documents} (device memory: e.g. elements n_docs and
 *                          n_docs + 1 of the offsets buffer td_encode_device wrote, with the document count stored behind the
 *                          total) into d_table[2 * world] on every rank; asynchronous on `stream`
 *   td_comm_bases          host: exclusive prefix sums of a gathered table -> this rank's token / document base and the totals
 *   td_comm_gather_tokens  grouped ncclSend / ncclRecv: every rank's ids (table[2 r] of them) end up contiguous, in rank order,
 *                          in d_root_tokens on `root`; `table` is the gathered table on the HOST; asynchronous on `stream`.
 *                          A root buffer that is too small fails on the ROOT only (TD_E_CAPACITY), after the root has taken
 *                          the other ranks' ids into a scratch buffer: no rank is left with a pending send.
 */
#define TD_COMM_ID_BYTES 128
typedef struct td_comm td_comm;
int td_comm_unique_id(uint8_t id[TD_COMM_ID_BYTES]);
int td_comm_create(const uint8_t id[TD_COMM_ID_BYTES], int world, int rank, int device, td_comm** out);
void td_comm_destroy(td_comm* c);
int td_comm_gather_counts(td_comm* c, const int64_t* d_counts, int64_t* d_table, void* stream);
int td_comm_bases(const int64_t* table, int world, int rank, int64_t* token_base, int64_t* doc_base, int64_t* token_total,
                  int64_t* doc_total);
int td_comm_gather_tokens(td_comm* c, const int32_t* d_tokens, const int64_t* table, int root, int32_t* d_root_tokens,
                          int64_t root_capacity, void* stream);
const char* td_comm_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* TOKENDAGGER_HIP_H */
