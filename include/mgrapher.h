/* mgrapher.h — C ABI of libmgrapher_hip.so
 *
 * MI355X-native (gfx950, hand-written HIP) implementation of ONE path of DS4SD/MarkushGrapher: the
 * MarkushGrapher-2 VTL (UDOP) encoder + autoregressive CXSMILES decoder forward pass, i.e. what the reference
 * reaches through its transformers fork (`transformers.models.markushgrapher`, imported at
 * /root/reference/markushgrapher/core/common/begin.py:7-13):
 *
 *   mg_generate         replaces  model.generate(**encoding, num_beams, max_length)
 *                                 /root/reference/markushgrapher/utils/ocsr/utils_evaluation.py:269-285
 *   mg_encode + mg_decoder_forward
 *                       replace   model(**sample).logits (teacher-forced forward)
 *                                 /root/reference/markushgrapher/core/trainers/curriculumTrainer.py:654-656
 *   mg_create / mg_load_tensor / mg_finalize
 *                       replace   MarkushgrapherForConditionalGeneration.from_pretrained(...).to(device)
 *                                 /root/reference/markushgrapher/core/common/begin.py:128-133
 *   mg_beam_reorder     the KV-cache reorder of beam search (stock transformers cache_utils.py:100-104),
 *                       exposed for the beam-reorder micro-benchmark (BASELINE.json configs[2])
 *
 * Conventions: plain pointers and sizes, no framework types.  Every `void* stream` is a hipStream_t.  Every data
 * pointer is a DEVICE pointer unless the name ends in `_host`.  The caller owns all buffers (weights arena,
 * workspace, inputs, outputs); the library allocates no device memory.  Calls only enqueue work on `stream`
 * unless documented as synchronising.  Return value: 0 (MG_OK) or a negative MG_E_* code, message via
 * mg_last_error() (thread-local).  One mg_model may be used from one thread at a time.
 */
#ifndef MGRAPHER_H
#define MGRAPHER_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

enum { MG_OK = 0, MG_E_SHAPE = -1, MG_E_ARG = -2, MG_E_STATE = -3, MG_E_HIP = -4, MG_E_UNSUPPORTED = -5,
       MG_E_WORKSPACE = -6, MG_E_KEY = -7, MG_E_INPUT = -8 };
enum { MG_KEY_IGNORED = 1 };   /* mg_load_tensor: key recognised as not on this path (e.g. encoder.molscribe_*) */
enum { MG_F32 = 0, MG_BF16 = 1 };

/* UdopConfig fields the path reads (stock transformers models/udop/configuration_udop.py:43-71). */
typedef struct mg_config {
    int vocab_size, d_model, d_kv, d_ff, num_layers, num_decoder_layers, num_heads;
    int relative_attention_num_buckets, relative_attention_max_distance;
    int max_2d_position_embeddings, image_size, patch_size, num_channels;
    int pad_token_id, eos_token_id, decoder_start_token_id;
    float layer_norm_epsilon;
    int max_decode_len;          /* capacity of the decoder self-attention cache / bias table (>= max_length) */
    int tie_word_embeddings;     /* 1 (UDOP / MarkushGrapher default): lm_head = shared.weight, logits scaled by d_model^-0.5
                                    (stock modeling_udop.py:1405-1413,1554-1557) and a checkpoint's lm_head.weight is ignored,
                                    as HF's tie_weights() does; 0: lm_head.weight is a required tensor, no scale */
} mg_config;

typedef struct mg_model mg_model;

int mg_create(const mg_config* cfg, mg_model** out);
void mg_destroy(mg_model* m);
const char* mg_last_error(void);

/* Weights: caller allocates mg_weights_bytes() of device memory and binds it; mg_load_tensor converts one HF
 * state-dict tensor (key = HF name, e.g. "encoder.block.0.layer.0.SelfAttention.q.weight"; device pointer;
 * MG_F32 or MG_BF16) into the library's bf16 fragment-tile layout inside the arena. mg_finalize builds the derived
 * bias tables, ties lm_head to shared.weight when lm_head.weight was not loaded, and checks completeness. */
size_t mg_weights_bytes(const mg_model* m);
int mg_bind_weights(mg_model* m, void* arena);
int mg_load_tensor(mg_model* m, void* stream, const char* hf_key, const void* src, int dtype, const int64_t* shape,
                   int ndim);
int mg_finalize(mg_model* m, void* stream);

/* A further execution context on the weights of a finalized model: the clone reads the same arena (no copy; the arena must
 * outlive it) and owns everything a call mutates - captured decode graph, events, run-ahead stream, profiling state, last-error
 * text is per host thread - so calls on different contexts may run at the same time from different host threads, each on its
 * own stream with its own workspace.  That is how several batches are kept in flight on one GPU: five of the six launches of
 * a decoder layer are latency-sized, a second and third batch's launches fill the machine they leave idle, and every batch's
 * ids are exactly what it gets alone (rows never meet).  Calls on ONE context stay serial.  Tensors loaded later through any
 * of the contexts land in the shared arena; run mg_finalize on the context that loaded them before the next call of any. */
int mg_clone(const mg_model* src, mg_model** out);

/* Workspace for a batch of B images with L text tokens, num_beams beams, decoder length max_length, (for
 * mg_decoder_forward) T teacher-forced positions (0 if unused) and M_e1 OCSR-branch tokens per image (0 if unused). */
int mg_workspace_bytes(const mg_model* m, int B, int L, int num_beams, int max_length, int T, int M_e1, size_t* out_bytes);

/* Encoder (stock modeling_udop.py:1102-1246 for the encoder stack).  attention_mask may be NULL (= everything
 * attended, incl. the zero-padded visual slots: stock:1183-1186).  Leaves the encoder state (final hidden states,
 * mask, compaction map) in the workspace for mg_decoder_forward.
 * enc_out [B][L+P][d_model] fp32 and enc_mask [B][L+P] u8 are optional outputs (the VTL states e2 only).  Rows of enc_out at
 * positions with enc_mask == 0 (padded text slots, the slots of dropped patches) are UNSPECIFIED: nothing downstream reads them
 * (the decoder's cross-attention masks them), and the encoder GEMMs skip whole 32-row tiles of such positions.  Attended rows are
 * bit-identical with the skip off (environment MG_ENC_ROW_TILES=0, which also makes the unattended rows follow the reference).
 * e1 (nullable, with M_e1 = 0): [B][M_e1][d_model] fp32, the projected embeddings of MarkushGrapher-2's OCSR vision branch
 * (`encoder.molscribe_encoder` Swin-B + `encoder.molscribe_projector`, /root/reference/markushgrapher/utils/model/
 * utils_model_loading.py:23,36), computed by the caller: that branch lives only in the reference's un-vendored fork and is
 * NOT part of this library.  They are fused as the reference describes (README.md:212-215: "e1 is
 * concatenated with the VTL embedding e2 and fed to the text decoder"): the decoder cross-attends over [e1 | e2], every e1
 * token attended.  Parity of this fusion is UNPINNED (no fork source): order and normalisation are inferred. */
int mg_encode(mg_model* m, void* stream, void* ws, size_t ws_bytes, const int64_t* input_ids, const float* bbox,
              const uint8_t* attention_mask, const float* pixel_values, const float* e1, int M_e1, int B, int L,
              float* enc_out, uint8_t* enc_mask);

/* Teacher-forced decoder + lm_head over T positions (stock:1448-1574): logits [B][T][vocab] fp32.
 * decoder_attention_mask may be NULL.  Requires a preceding mg_encode on the same workspace. */
int mg_decoder_forward(mg_model* m, void* stream, void* ws, size_t ws_bytes, const int64_t* decoder_input_ids,
                       const uint8_t* decoder_attention_mask, int B, int T, float* logits);

/* generate(): encoder once, then KV-cached decode.  num_beams == 1: greedy (stock generation/utils.py:2783-2975);
 * num_beams > 1: beam search (utils.py:3208-3525, beams_to_keep = 2*num_beams, length_penalty, early_stopping 0/1).
 * min_length suppresses EOS while the sequence is shorter (MinLengthLogitsProcessor).
 * out_ids [B][max_length] i64 (row = [start, tok..., eos, pad...]); *out_cols_host = number of valid columns
 * (the length HF's generate would return).  out_scores [B] (beam: sequences_scores; nullable).
 * step_top2 [max_length][B*num_beams][2] fp32 top-1/top-2 logits per step (greedy only, nullable; parity tests).
 * SYNCHRONISES the stream before returning. */
int mg_generate(mg_model* m, void* stream, void* ws, size_t ws_bytes, const int64_t* input_ids, const float* bbox,
                const uint8_t* attention_mask, const float* pixel_values, const float* e1 /* nullable, see mg_encode */,
                int M_e1, int B, int L, int num_beams, int max_length,
                int min_length, float length_penalty, int early_stopping, int64_t* out_ids, int* out_cols_host,
                float* out_scores, float* step_top2);

/* Limits: B * num_beams <= 256 live sequences per call (MG_E_UNSUPPORTED beyond; split the batch), num_beams <= 8. */

/* Continuous greedy decoding of N images - what the reference's evaluation loop does one image at a time with
 * model.generate(**encoding, num_beams=1, max_length=512) until EOS (/root/reference/markushgrapher/utils/ocsr/utils_evaluation.py:140,
 * 269-285), for a whole queue of images: `slots` decode rows work through the queue, a row that ends (EOS / max_length) frees its
 * slot for the next image (no step is spent on finished rows, the batch does not wait for its longest member), and the encoder +
 * cross-K/V projection of the next `chunk` images run ahead on a second stream underneath the launch-bound decode steps.
 *   inputs   input_ids [N][L] i64, bbox [N][L][4] f32, attention_mask [N][L] u8 (nullable), pixel_values [N][3][I][I] f32, all resident
 *   outputs  out_ids [N][max_length] i64 (row = [start, tok..., eos, pad...]), out_len [N] i32 = valid columns per image (device)
 *   chunk    images per encoder pass (32); slots = live decode rows (<= 256); pool_chunks >= 2: the cross-K/V pool holds
 *            pool_chunks * chunk images (the encoder runs at most that far ahead of the oldest unfinished image)
 * Every image's ids equal mg_generate's for that image.  *steps_host (nullable) = decode steps run.  SYNCHRONISES before returning. */
int mg_stream_workspace_bytes(const mg_model* m, int chunk, int L, int slots, int pool_chunks, size_t* out_bytes);
int mg_generate_stream(mg_model* m, void* stream, void* ws, size_t ws_bytes, const int64_t* input_ids, const float* bbox,
                       const uint8_t* attention_mask, const float* pixel_values, int N, int L, int chunk, int slots, int pool_chunks,
                       int max_length, int min_length, int64_t* out_ids, int32_t* out_len, long* steps_host);
/* The same queue with BEAM SEARCH - the reference's shipped decode mode (/root/reference/config/predict.yaml:12-13 beam_search: True;
 * utils_evaluation.py:269-285 generate(num_beams=5, max_length=512)): `slots` IMAGE slots of num_beams rows each (slots * num_beams <= 256)
 * work through the queue; an image whose own stopping condition holds (stock 5.15 generation/utils.py:3055-3075 for that image alone - in a
 * batch call such an image is frozen until the whole batch stops) is written out and its slot handed to the next image.
 *   outputs  out_ids [N][max_length] i64 = best hypothesis (unfinished tails filled as stock does), out_len [N] = its columns,
 *            out_scores [N] f32 (nullable) = its length-penalised score
 * Every image's hypothesis and score equal mg_generate(num_beams)'s for that image.  SYNCHRONISES before returning. */
int mg_stream_beam_workspace_bytes(const mg_model* m, int chunk, int L, int slots, int pool_chunks, int num_beams, int max_length, size_t* out_bytes);
int mg_generate_stream_beam(mg_model* m, void* stream, void* ws, size_t ws_bytes, const int64_t* input_ids, const float* bbox,
                            const uint8_t* attention_mask, const float* pixel_values, int N, int L, int chunk, int slots, int pool_chunks,
                            int num_beams, int max_length, int min_length, float length_penalty, int early_stopping, int64_t* out_ids,
                            int32_t* out_len, float* out_scores, long* steps_host);
/* Where the encoder of mg_generate_stream runs: 0 = on the caller's stream (serial), 1 = own stream at the lowest priority
 * (default), 2 = own stream restricted to the compute units of cu_mask (nwords x 32 bits, hipExtStreamCreateWithCUMask). */
int mg_stream_encoder_mode(mg_model* m, int mode, const uint32_t* cu_mask, int nwords);

/* Parity-test instrumentation of mg_generate (tests only; while set, the decode steps are launched eagerly with by-value
 * arguments instead of replaying the captured graph - the same kernels).  logits_capture [capture_steps][B*num_beams][vocab]
 * fp32 (device) receives the pre-argmax logits of decode steps 0 .. capture_steps-1.  forced_ids [B][max_length] i64
 * (device, greedy only): the token fed into step t+1 is forced_ids[b][t+1] instead of step t's argmax, i.e. teacher
 * forcing THROUGH the KV-cached decode path; out_ids / step_top2 still record every step's own selection.  NULL clears. */
int mg_debug_decode_capture(mg_model* m, float* logits_capture, int capture_steps, const int64_t* forced_ids);

/* Beam-search KV-cache reorder, physical form: for every decoder layer, dst K/V rows [r] = src rows [beam_idx[r]]
 * (cache_utils.py:100-104 index_select).  kv_src/kv_dst: [layers][2][rows][H][t_cap][64] bf16; only the first
 * `t_used` positions of every row are copied. */
int mg_beam_reorder(void* stream, const void* kv_src, void* kv_dst, const int32_t* beam_idx, int layers, int rows,
                    int H, int t_cap, int t_used);

/* Live timing of the dominant decode kernel (single-query cross-attention over the per-image K/V stream): when
 * every > 0, each `every`-th decode step of mg_generate brackets every decoder layer's launch with HIP events on the
 * caller's stream (at most max_samples launches per call). mg_profile_read returns the number of launches timed, their
 * summed duration and the summed number of (image, key) rows each of them streamed (bytes = rows * H * 64 * 2 * 2). */
int mg_profile_cross_attention(mg_model* m, int every, int max_samples);
int mg_profile_read(mg_model* m, long* launches_host, double* total_ms_host, double* total_keys_host);
/* Calibration of the bracket: after every timed launch a third event is recorded right behind the second; this returns
 * the summed duration of those EMPTY brackets (what two hipEventRecord cost on the stream with nothing in between). */
int mg_profile_read_overhead(mg_model* m, double* empty_ms_host);
/* Phase timing of mg_generate: three HIP events per call bracket [encoder + cross-K/V precompute] and [decode loop];
 * mg_profile_phases_read returns the number of calls and the summed durations since it was enabled. */
int mg_profile_phases(mg_model* m, int enable);
int mg_profile_phases_read(mg_model* m, long* calls_host, double* enc_ms_host, double* dec_ms_host);
int mg_debug_bucket_table(const mg_model* m, int which, int* out_host, int n);

/* Decode-step replay. The ~200 launches of one decode step (the body of the reference's generation loop,
 * transformers generation/utils.py `_sample` / `_beam_search` while-loop) are captured once as a HIP graph whose
 * kernels read the step index from device memory; later steps - and later calls with the same buffers and
 * generation arguments - replay it with one hipGraphLaunch. enable = 0 launches every kernel eagerly (same kernels,
 * same results); enable = 2 launches the graph's device-counter form of the kernels eagerly (test hook). Default: 1.
 * Returns the previous setting. */
int mg_set_decode_graph(mg_model* m, int enable);
/* Trailing text padding of a batch (attention_mask with zeros at the end of a row).  0 (default): the padded slots are part of the
 * sequence, exactly as stock HF computes a padded batch - UDOP's 1-D relative position bias then counts them between the text and
 * the patches, so an image's result depends (slightly) on how far its batch was padded.  1: per-image semantics - every image's
 * patches follow its own last attended text token and its result is what the reference's batch-size-1 loop computes for it
 * (/root/reference/markushgrapher/utils/ocsr/utils_evaluation.py:140), whatever its batch mates are.  Returns the previous setting. */
int mg_set_padding_semantics(mg_model* m, int per_image);
/* Several execution contexts share the GPU (mg_clone; markushgrapher_amd/inflight.py sets it on the contexts it runs).  shared = 1: the one
 * long-lived, chip-filling launch of a decode step - the cross-attention K/V stream - keeps ONE workgroup per CU resident instead of four, so
 * that the latency-sized launches of the other contexts find wave slots (4 contexts x 128 rows: 143 -> 148 images/s; a call alone: 107 -> 103,
 * hence off by default).  Same kernels, same results.  Takes effect from the context's next call (its captured decode step is dropped).
 * Returns the previous setting.  No counterpart in the reference (one batch at a time, utils_evaluation.py:269-285). */
int mg_set_shared_gpu(mg_model* m, int shared);
/* Cross-attention of the greedy decode step (num_beams = 1, batch and queue forms).  Weight-absorbed form: with K_l = enc·Wk_l^T,
 * V_l = enc·Wv_l^T (stock transformers models/udop/modeling_udop.py:524-550, reached from
 * /root/reference/markushgrapher/utils/ocsr/utils_evaluation.py:278-281) softmax(q_h K_h^T) V_h = [softmax((q_h·Wk_h) enc^T) enc]·Wv_h^T, so
 * every layer streams the attended encoder states (2·d_model bytes per position) instead of its own K and V (4·d_model): half the
 * dominant HBM stream of decoding, no cross-K/V projections after the encoder, no per-layer K/V buffers (the workspace shrinks; sizes are
 * computed for the current setting and call shape).  The stream runs one workgroup per decode row: it pays from ~100 rows per call on
 * (160 rows, 4 contexts in flight: 148 -> 197 images/s) and loses below (32 rows alone: 82 -> 50 images/s, latency-bound).
 *   absorb = 2 (default wherever the geometry has the form: d_model a supported multiple of 64, at most 16 heads): by the call's decode
 *              rows - absorbed from 96 rows on, K / V form below;
 *   absorb = 1: absorbed for every greedy call;   absorb = 0: the K / V form always (what beam search always uses);
 *   absorb < 0: query only.
 * key_splits in 1..4: workgroups per decode row of the stream (0 keeps the setting).  The two forms round at different points (q' = q·Wk_h
 * and the normalised context are rounded to bf16 instead of K and V): logits agree within the stated tolerance, NOT bitwise - with
 * absorb = 2 a row decoded in a 160-row call and the same row in a 32-row call go through different forms; pin 0 or 1 where ids must be
 * reproducible across call sizes (markushgrapher_amd/inflight.py does, for the calls it packs).  Takes effect from the context's next
 * call.  Returns the previous `absorb`. */
int mg_set_cross_absorb(mg_model* m, int absorb, int key_splits);
/* 1 if the last mg_generate replayed a captured graph, 0 if it launched eagerly (mode 0/2, capture unavailable). */
int mg_decode_graph_active(const mg_model* m);

/* The multi-GPU path's one collective behind the C ABI (image shards are independent, the ranks exchange decoded ids):
 * an RCCL all-gather of equal-sized byte blocks on a stream the caller names.  The library resolves librccl at run time (the copy the
 * process already holds, else the system's) and adds no link dependency.  Rendezvous stays with the host: rank 0 calls
 * mg_dist_unique_id and ships the 128 bytes to every rank by whatever it has (markushgrapher_amd/dist.py: torch.distributed's
 * store / broadcast), then EVERY rank calls mg_dist_create (collective).  mg_dist_allgather enqueues ncclAllGather(send, recv) of
 * bytes_per_rank bytes per rank (recv holds world x bytes_per_rank, rank order) and returns; the buffers must stay valid until the
 * stream has passed it.  No counterpart in the reference (one device, /root/reference/markushgrapher/utils/ocsr/utils_evaluation.py:140). */
typedef struct mg_dist mg_dist;
int mg_dist_unique_id(void* out, int bytes);
int mg_dist_create(const void* unique_id, int bytes, int rank, int world, mg_dist** out);
int mg_dist_allgather(mg_dist* d, void* stream, const void* send, void* recv, size_t bytes_per_rank);
void mg_dist_destroy(mg_dist* d);

/* Page preprocessing on the device ("next" row f-3): replaces page_image.resize((512,512), Image.LANCZOS)
 * (/root/reference/markushgrapher/core/datasets/mdu_dataset.py:118) + MarkushgrapherImageProcessor's rescale 1/255 and
 * mean = std = 0.5 normalisation (/root/reference/markushgrapher/core/common/begin.py:105-109), bit-exactly (Pillow's
 * 8-bit fixed-point resampler).  pages_u8 [B][Hs][Ws][3] (RGB, HWC) -> pixel_values [B][3][out][out] f32.
 * scratch: mg_preprocess_scratch_bytes() of device memory. */
size_t mg_preprocess_scratch_bytes(int B, int Hs, int Ws, int out_size);
int mg_preprocess_pages(void* stream, const uint8_t* pages_u8, int B, int Hs, int Ws, int out_size, float* pixel_values,
                        void* scratch, size_t scratch_bytes);

/* Device self-test of the hardware assumptions the kernels rely on (MFMA fragment layout, global_load_lds
 * destination rule, cross-half shuffle). Synchronises. msg_host receives a short report. */
int mg_selftest(void* stream, void* scratch_256k, char* msg_host, int msg_len);

/* ---- kernel-level operator entry points (used by the parity tests) ---- */
int mgk_pack_weight(void* stream, const void* src, int src_is_bf16, int N, int K, void* dst_pk, int Npad);
int mgk_rmsnorm_pack(void* stream, const float* h, const float* gain, void* x_pk, float* out_f32, int M, int d,
                     float eps, float scale);
int mgk_im2col_pack(void* stream, const float* pix, void* x_pk, int B, int C, int I, int ps);
/* TEST / A-B SWITCHES (mgk_set_rows_split, mgk_set_resid_f16, mgk_gemm_set_variant, mgk_set_rows_mt, mgk_set_attention_qt, mgk_set_pp_parts): PROCESS-WIDE, for the parity tests and tools/.
 * Each selects among kernels whose results are bit-identical (that is what the tests that flip them check), so a call that races
 * with a flip still returns the right bits; they are nevertheless meant to be set while no call is running, and the product
 * (markushgrapher_amd/) never touches them.
 * row-tile split policy of the decode-step projections with more than 32 live rows: -1 default (by weight size), 0 never,
 * 1 always one row tile per workgroup */
int mgk_set_rows_split(int mode);
/* residual projections of the decode step with several row tiles: 1 (default) 16 features per workgroup, 0: 8 */
int mgk_set_resid_f16(int on);
/* Half-tile decode projections with three or more 32-row tiles of live rows: 1 = a workgroup takes both 16-feature halves of a weight tile
 * (half the activation re-reads through L2), 0 = one half per workgroup, -1 (default) = as the caller asks (the engine asks for it on
 * contexts that share the GPU, mg_set_shared_gpu).  Same bits either way. */
int mgk_set_rows_ft2(int mode);
/* projections of the decode step with several row tiles: 1 the K-slab form (K chunks as workgroups, partial sums merged by the last
 * arrival in the one-workgroup forms' order) where the caller provides its scratch, 2 the same in two launches (partial sums, then a
 * chip-wide merge launch), 0 (default: measured faster) the one-workgroup forms */
int mgk_set_rows_mt(int on);
/* encoder attention: 1 (default) one 32-query tile per wave / 8 waves per workgroup, 2 two tiles per wave / 4 waves (same bits, slower) */
int mgk_set_attention_qt(int qt);
/* persistent ping-pong GEMM: the row tiles of one problem over several launches - 0 (default) never (problems with more tiles per workgroup
 * than the kernel's table holds go to the two-stage kernel: measured equally fast), 1 when needed, 2 .. 8 always that many (same bits) */
int mgk_set_pp_parts(int mode);
/* mgk_gemm_resid with the scratch of the K-slab form: kpart [16][M padded to 32][N] fp32, ticket [N / 32] i32 zero-initialised;
 * wide_tiles as ResidArgs (8 in the decode steps) */
int mgk_gemm_resid_mt(void* stream, const void* X_pk, const void* W_pk, float* h, const float* gain, float gscale, void* x_pk,
                      float* part, int M, int N, int K, const float* rs_part, int rs_nparts, float rs_inv_d, float rs_eps, int wide_tiles,
                      float* kpart, int* ticket);
int mgk_gemm(void* stream, int mode, int epi, const void* X_pk, const void* W_pk, int M, int N, int K, float* out_f32,
             int ldo, const float* bias, void* out_pk);
/* Deferred-RMSNorm pair of the encoder (tiled large-M kernels): epi 5 (EPI_RESID_NORM): h_tiled (fp32, tiles of
 * [32 rows][4 features]: index ((m/32)*(N/4) + n/4)*128 + (m%32)*4 + n%4) += X W^T, out_pk = pack(bf16(h * gain)) un-normalised,
 * part[M][part_ld] = per-64-column partial sums of h^2; epi 3 / 2 (packed / packed relu): rows scaled by
 * rsqrt(sum(rs_part[m][0..rs_nparts)) * rs_inv_d + rs_eps) (rs_part NULL: no scale). */
int mgk_gemm_norm(void* stream, int epi, const void* X_pk, const void* W_pk, int M, int N, int K, float* h_tiled, const float* gain,
                  void* out_pk, float* part, int part_ld, const float* rs_part, int rs_nparts, float rs_inv_d, float rs_eps);
/* mgk_gemm / mgk_gemm_norm restricted to the 32-row tiles that hold a non-zero entry of row_mask [M] (the encoder's row-tile list:
 * whole tiles of padded text slots / dropped patch slots are neither read nor written).  list_scratch: M/32 + 1 ints. */
int mgk_gemm_row_tiles(void* stream, int epi, const void* X_pk, const void* W_pk, int M, int N, int K, float* out_f32, const float* gain,
                       void* out_pk, float* part, int part_ld, const float* rs_part, int rs_nparts, float rs_inv_d, float rs_eps,
                       const uint8_t* row_mask, int* list_scratch);
int mgk_gemm_heads(void* stream, int mode, const void* X_pk, const void* W_pk, int M, int N, int K, void* p0, void* p1,
                   void* p2, int f0, int f1, int f2, int H, int S_in, int S_cap, const int* row_map, int pos);
/* mode 0 (encoder): tab1/tabh/tabv are the RAW bucket tables [32][H]; bk1[257] / bkhv[201] the bucket ids of integer
 * distances -128..128 / -100..100; bidx_scratch [B][Sk_cap/32][Sk_cap][32] u16 receives the per-pair bucket indices.
 * mode 1 (decoder self): tab1 = [tab1_len][H] by distance; mode 2 (cross): no bias. */
int mgk_attention(void* stream, int mode, const void* Q, const void* K, const void* Vt, void* ctx_pk, int B, int H,
                  int Sq, int Sk, int Sq_cap, int Sk_cap, const uint8_t* kmask, const float* tab1, int tab1_len,
                  const float* tabh, const float* tabv, const double* cx, const double* cy, const int* bk1, const int* bkhv,
                  void* bidx_scratch);
/* mgk_attention(mode 0) with the skip lists mg_encode uses: 64-key stages / 128-query blocks without an attended position
 * are not visited (kst_scratch: B*(1+S_cap/64) ints, qbv_scratch: B*ceil(S_cap/128) bytes). */
int mgk_attention_enc_skip(void* stream, const void* Q, const void* K, const void* Vt, void* ctx_pk, int B, int H, int S, int S_cap,
                           const uint8_t* kmask, const float* tab1, const float* tabh, const float* tabv, const double* cx,
                           const double* cy, const int* bk1, const int* bkhv, void* bidx_scratch, int* kst_scratch,
                           uint8_t* qbv_scratch);
int mgk_attention_step(void* stream, const void* q, const void* Kc, const void* Vc, void* ctx_pk, int rows, int H,
                       int group, int cap, const int* len, int n_keys, const float* bias, const int* anc, int t);
/* Weight-absorbed cross-attention of one decoder layer and step (mg_set_cross_absorb; kernels of markushgrapher_amd/csrc/k_xattn.hip):
 * ctx[row][h] = softmax(q_h (enc Wk_h^T)^T) (enc Wv_h^T) (stock modeling_udop.py:524-575 without a position bias) evaluated as
 * [softmax((q_h Wk_h) enc^T) enc] Wv_h^T on the states themselves.  q [rows][H][64] bf16; wkv fp32 [2*H*64][d], K rows first;
 * enc [owners][cap][d] bf16; len [owners] keys per owner; kv_owner [rows] (null: row r reads owner r); scratch wk, wv (H*d*64 bf16
 * each), qx (rows*H*d bf16), part (rows*nsplit*H*d bf16), ml (rows*nsplit*H*2 fp32); ctx_pk packed [rows padded to 32][H*64]. */
int mgk_xattn(void* stream, const void* q, const float* wkv, const void* enc, const int* len, const int* kv_owner, int rows, int H, int d,
              int cap, int nsplit, int nstg, void* wk, void* wv, void* qx, void* part, float* ml, void* ctx_pk);
int mgk_enc_rows(void* stream, const void* src_pk, const int* row_map, void* dst, int B, int rows_per_image, int cap, int d);
int mgk_embed_assemble(void* stream, void* meta_ws, const int64_t* input_ids, const float* bbox,
                       const uint8_t* attention_mask, const float* patch_emb, const void* tok_emb, const void* x_emb,
                       const void* y_emb, int B, int L, int P, int d, int n_side, int M2, int V, int S_cap,
                       float* hidden, double* cx, double* cy, uint8_t* mask, int* xrow, int* xlen, int* err);
size_t mgk_embed_meta_bytes(int B, int S_cap);
int mgk_greedy_select(void* stream, const float* logits, int rows, int V, int ldl, int eos, int pad, int min_len,
                      int64_t* next_ids, int64_t* out_ids, int max_len, int pos, int* unfinished, int* n_unfinished,
                      float* top2);

/* decode-step split-K projection: P[ks][m*ldp + n] partial sums (ks < KS); consumers sum the slabs */
int mgk_gemm_splitk(void* stream, const void* X_pk, const void* W_pk, float* P, int M, int N, int K, int ldp,
                    size_t slab_stride, int KS);
int mgk_splitk_factor(int N, int K);
/* A/B switch of the large-M GEMM kernel (0: 128x128 two-stage, 1: 256x128 three-stage, default) */
int mgk_gemm_set_variant(int v);
/* decode-step residual projection with the next RMSNorm folded in: h += X W^T (optionally scaled per row by the deferred
 * statistic rs_*), x_pk = bf16(h*gain*gscale) un-normalised, part[m][N/8] = per-block sums of h^2 */
int mgk_gemm_resid(void* stream, const void* X_pk, const void* W_pk, float* h, const float* gain, float gscale, void* x_pk,
                   float* part, int M, int N, int K, const float* rs_part, int rs_nparts, float rs_inv_d, float rs_eps);
/* Pair projection of the decode step with its product weight (test entry for gemm_rows_pair / the mg_finalize helpers):
 *   W2_pk  <- pack([Wn*diag(gain) | Wn*diag(gain)*Wr])   built on the device from Wn_pk [N2][d], Wr_pk [d][inner]
 *             (scratch_f32: at least N2*(2*d+inner) + d*inner floats);
 *   one launch: h[M][d] += ctx*Wr^T, hb_out_pk = pack(bf16(h_new)), part = per-row partial sums of h_new^2   |
 *               out2_pk = pack(relu?(W2 * [hb ; ctx]))     with xwin_pk = packed [rows][d+inner] = [bf16(h_old) | ctx]. */
int mgk_gemm_pair(void* stream, const void* Wn_pk, const void* Wr_pk, const float* gain, int N2, int d, int inner, void* W2_pk,
                  float* scratch_f32, const void* xwin_pk, float* h, void* hb_out_pk, float* part, void* out2_pk, int M, int relu);
int mgk_add_norm_pack(void* stream, float* h, const float* P, int KS, int ldp, size_t slab_stride, const float* gain,
                      void* x_pk, int M, int d, float eps, float scale);
int mgk_relu_pack(void* stream, const float* P, int KS, int ldp, size_t slab_stride, void* y_pk, int M, int N);

/* ------------------------------------------------------------------------------------------------------------------------------
 * ChemicalOCR stage ("next" row f-1).  Replaces, for the OCR pass that produces the cells the main model
 * reads, what /root/reference/markushgrapher/ocr/chemical_ocr.py does on one 512-px page at a time:
 *     processor, model = AutoProcessor / AutoModelForVision2Seq.from_pretrained(model_path)            chemical_ocr.py:76-84
 *     inputs = processor(text=prompt, images=[image], return_tensors="pt", size={"longest_edge": 512})  chemical_ocr.py:368-373
 *     generated_ids = model.generate(**inputs, max_new_tokens=4096, do_sample=False)                    chemical_ocr.py:375-380
 * i.e. stock Idefics3ForConditionalGeneration (SigLIP-style vision tower, pixel-shuffle connector, Llama-style text model),
 * greedy.  Tokenisation / image resizing stay on the host (the stock processor); this library takes the processor's tensors.
 * Padded frames (non-square pages: pixel_attention_mask not all ones) are given as the patch grid the stock model derives from the
 * mask: patch_mask [N][P] u8 (a patch is valid if any of its pixels is) and patch_pos [N][P] i32 (the bucketed fractional coordinates
 * of Idefics3VisionEmbeddings, modeling_idefics3.py:128-172, computed by the host wrapper with the same torch ops); both NULL = full
 * frames.  v1 limits: head dim 64 everywhere; every sequence of a call has the same prompt length (no padding), the same number of
 * frames and exactly n_img * image_seq_len <image> tokens.  Same conventions as above (device pointers, caller-owned buffers).
 * ------------------------------------------------------------------------------------------------------------------------------ */
typedef struct mg_ocr_model mg_ocr_model;
typedef struct mg_ocr_config {
    /* Idefics3VisionConfig */
    int v_hidden, v_inter, v_layers, v_heads, image_size, patch_size;
    /* LlamaConfig of the text model */
    int t_hidden, t_inter, t_layers, t_heads, t_kv_heads, vocab;
    /* Idefics3Config */
    int scale_factor, image_token_id, eos_token_id, pad_token_id, tie_word_embeddings;
    float v_eps, rms_eps, rope_theta;
    /* further stop tokens: generation_config.json's eos_token_id may be a list (e.g. <|im_end|> and <end_of_utterance>); a row
     * finishes on eos_token_id or any of eos_extra[0 .. n_eos_extra) (generation/utils.py:2927-2937 with a list of EOS ids) */
    int n_eos_extra, eos_extra[3];
} mg_ocr_config;

int mg_ocr_create(const mg_ocr_config* cfg, mg_ocr_model** out);
void mg_ocr_destroy(mg_ocr_model* m);
/* a further execution context on a finalized model's weights (see mg_clone): own captured graphs, calls may overlap in time with the
 * source's when made from another host thread on another stream with another workspace */
int mg_ocr_clone(const mg_ocr_model* src, mg_ocr_model** out);
size_t mg_ocr_weights_bytes(const mg_ocr_model* m);
int mg_ocr_bind_weights(mg_ocr_model* m, void* arena);
/* hf_key: a state-dict key of stock Idefics3ForConditionalGeneration ("model.vision_model....", "model.connector....",
 * "model.text_model....", "lm_head.weight"); src: device tensor, fp32 (src_is_bf16 = 0) or bf16 bits. */
int mg_ocr_load_tensor(mg_ocr_model* m, void* stream, const char* hf_key, const void* src, int src_is_bf16, const int64_t* shape, int ndim);
int mg_ocr_finalize(mg_ocr_model* m, void* stream);
int mg_ocr_workspace_bytes(const mg_ocr_model* m, int B, int n_img, int L, int max_new_tokens, int full_logits, size_t* out_bytes);
/* get_image_features (modeling_idefics3.py:563-622): pixel_values [N][3][I][I] fp32 -> out [N][image_seq_len][t_hidden] fp32
 * (workspace: mg_ocr_workspace_bytes(m, N, 1, 1, 0, 0)). */
int mg_ocr_image_features(mg_ocr_model* m, void* stream, void* ws, size_t ws_bytes, const float* pixel_values, const int32_t* patch_pos,
                          const uint8_t* patch_mask, int N, float* out);
/* Teacher-forced forward (modeling_idefics3.py:750-840): input_ids [B][L] i64, pixel_values [B][n_img][3][I][I] fp32 (NULL with
 * n_img = 0: text only) -> logits [B][L][vocab] fp32.  Workspace with full_logits = 1.  SYNCHRONISES (reports bad inputs). */
int mg_ocr_forward(mg_ocr_model* m, void* stream, void* ws, size_t ws_bytes, const int64_t* input_ids, const float* pixel_values,
                   const int32_t* patch_pos, const uint8_t* patch_mask, int B, int n_img, int L, float* logits);
/* generate(max_new_tokens, do_sample=False) (generation/utils.py:2783-2975): out_ids [B][max_new_tokens] i64 = the NEW tokens
 * (finished rows padded with pad_token_id), *out_cols_host = columns HF would have produced (steps until every row had emitted
 * EOS).  step_logits (nullable, tests): [capture_steps][B][vocab] fp32 pre-argmax logits of the first steps.  SYNCHRONISES. */
int mg_ocr_generate(mg_ocr_model* m, void* stream, void* ws, size_t ws_bytes, const int64_t* input_ids, const float* pixel_values,
                    const int32_t* patch_pos, const uint8_t* patch_mask, int B, int n_img, int L, int max_new_tokens, int64_t* out_ids,
                    int* out_cols_host, float* step_logits, int capture_steps);

/* Queue form of mg_ocr_generate (continuous decoding): N pages, `slots` decode rows (<= 256).  Vision tower + prompt prefill of all
 * pages first (chunk <= 256 pages per pass) into per-page KV caches, each prefill selecting its page's first token; then a row whose
 * page emits a stop token / reaches max_new_tokens takes the next page (a pointer change: the cache is already there), so the
 * call runs sum(lengths) / slots steps instead of walking every row to the longest page.  Per-page ids equal mg_ocr_generate's.
 * out_ids [N][max_new_tokens] i64 (pad after the stop token), out_len [N] i32 (device), *steps_host decode steps (nullable).
 * Same input contract as mg_ocr_generate (equal prompt length L and n_img frames for every page; prompts of different lengths:
 * mg_ocr_generate_stream_ragged below).  SYNCHRONISES. */
int mg_ocr_stream_workspace_bytes(const mg_ocr_model* m, int N, int n_img, int L, int max_new_tokens, int slots, int chunk, size_t* out_bytes);
int mg_ocr_generate_stream(mg_ocr_model* m, void* stream, void* ws, size_t ws_bytes, const int64_t* input_ids, const float* pixel_values,
                           const int32_t* patch_pos, const uint8_t* patch_mask, int N, int n_img, int L, int max_new_tokens, int slots, int chunk,
                           int64_t* out_ids, int32_t* out_len, long* steps_host);
/* The same with prompts of DIFFERENT lengths: input_ids [N][L] holds every page's prompt_len[n] <= L tokens LEFT-ALIGNED (anything behind them is
 * never attended), prompt_len [N] i32 on the device (null: mg_ocr_generate_stream).  What stock transformers computes for the left-padded batch the
 * Idefics3 processor makes of such prompts (position = cumsum(attention_mask) - 1, pad keys masked: every row as if alone; modeling_llama.py
 * position_ids, generation/utils.py prepare_inputs_for_generation) - the reference itself never makes one: its transformers backend calls
 * generate() one page at a time with one prompt (markushgrapher/ocr/chemical_ocr.py:366-392).  Every page still carries n_img frames. */
int mg_ocr_generate_stream_ragged(mg_ocr_model* m, void* stream, void* ws, size_t ws_bytes, const int64_t* input_ids, const int32_t* prompt_len,
                                  const float* pixel_values, const int32_t* patch_pos, const uint8_t* patch_mask, int N, int n_img, int L,
                                  int max_new_tokens, int slots, int chunk, int64_t* out_ids, int32_t* out_len, long* steps_host);

/* ------------------------------------------------------------------------------------------------------------------------------
 * OCSR vision branch "e1" (SURVEY.md section 8 rows a7 / f-2).  The reference's model evaluates, inside forward() / generate() of its
 * transformers fork,
 *     model.encoder.molscribe_encoder    MolScribe's Swin-B (timm swin_base_patch4_window12_384; weights swin_base_char_aux_1m680k.pth)
 *     model.encoder.molscribe_projector  an MLP projector into d_model
 * and concatenates the result in front of the decoder with the VTL encoder's states ("late fusion", architecture_variant
 * me-lf-stack-1): /root/reference/markushgrapher/core/common/begin.py:119-120,137-151, utils/model/utils_model_loading.py:20-36,
 * config/predict.yaml:12,17-18, README.md:212-215.  mg_e1_* computes that branch; mg_attach_e1 makes mg_encode / mg_generate /
 * mg_generate_stream* evaluate it themselves whenever no precomputed e1 is passed.
 * The Swin arithmetic is stock transformers models/swin/modeling_swin.py (patch embedding + LayerNorm, (shifted-)window attention with
 * the relative-position bias table and the -100 region mask, exact-erf GELU MLP, patch merging, final LayerNorm) and is pinned on
 * stock SwinModel; what the fork does around it is INFERRED and therefore configuration: the branch's input = bilinear resize
 * (torch.nn.functional.interpolate, align_corners = false) of the model's pixel_values from src_image_size to image_size followed by
 * x * pix_scale[c] + pix_shift[c]; the projector = n_proj Linear layers (with bias) with proj_act between them.
 * Keys of mg_e1_load_tensor: "swin." + the state-dict names of stock SwinModel(add_pooling_layer=False), and "proj.{j}.weight|bias"
 * (markushgrapher_amd/e1_shapes.py maps timm / transformers-4.x checkpoints onto them).  v1 limits (errors, never silent): head dim
 * 32, stage widths 64 .. 1024 (powers of two), window 4 / 8 / 12, every stage's map a multiple of the window (the reference geometry:
 * 96 / 48 / 24 / 12 with window 12).
 * ------------------------------------------------------------------------------------------------------------------------------ */
typedef struct mg_e1_model mg_e1_model;
typedef struct mg_e1_config {
    /* SwinConfig */
    int image_size, patch_size, num_channels, embed_dim, n_stages;
    int depths[4], num_heads[4];
    int window_size, mlp_ratio;
    float layer_norm_eps;
    /* projector: Linear(C_last -> proj_dims[0]) [act] ... Linear(-> proj_dims[n_proj - 1] = d_model); proj_act 0 none, 1 GELU (erf) */
    int n_proj, proj_dims[4], proj_act;
    /* input derivation */
    int src_image_size;
    float pix_scale[3], pix_shift[3];
} mg_e1_config;

int mg_e1_create(const mg_e1_config* cfg, mg_e1_model** out);
void mg_e1_destroy(mg_e1_model* m);
size_t mg_e1_weights_bytes(const mg_e1_model* m);
int mg_e1_bind_weights(mg_e1_model* m, void* arena);
int mg_e1_load_tensor(mg_e1_model* m, void* stream, const char* key, const void* src, int src_is_bf16, const int64_t* shape, int ndim);
int mg_e1_finalize(mg_e1_model* m, void* stream);
int mg_e1_out_tokens(const mg_e1_model* m);          /* M: tokens per image (144 for Swin-B at 384 px) */
int mg_e1_workspace_bytes(const mg_e1_model* m, int B, size_t* out_bytes);
/* pixel_values [B][num_channels][src_image_size][src_image_size] fp32 (the VTL model's input) ->
 *   e1_out       [B][M][d_model] fp32 (nullable)        what mg_encode / mg_generate take as `e1`
 *   features_out [B][M][C_last] fp32 (nullable)         SwinModel.last_hidden_state (the pinned part)
 * Reads the model, writes only caller buffers: one mg_e1_model may be used by several threads / execution contexts at once (each
 * with its own workspace and stream).  Enqueues only. */
int mg_e1_encode(const mg_e1_model* m, void* stream, void* ws, size_t ws_bytes, const float* pixel_values, int B, float* e1_out,
                 float* features_out);
/* Attach the branch to a model (and to the clones made from it afterwards; NULL detaches): mg_encode / mg_generate calls that pass
 * e1 = NULL, and every mg_generate_stream / mg_generate_stream_beam call, then evaluate mg_e1_encode on their pixel_values inside the
 * call (in the call's own workspace: mg_workspace_bytes / mg_stream_*_workspace_bytes account for it once attached) and decode over
 * [e1 | e2] - what the reference's model does with architecture_variant me-lf-stack-1.  A precomputed e1 passed by the caller still
 * takes precedence.  The branch must outlive the model; d_model, src_image_size and num_channels must match the model's config. */
int mg_attach_e1(mg_model* m, const mg_e1_model* e1);

#ifdef __cplusplus
}
#endif
#endif
