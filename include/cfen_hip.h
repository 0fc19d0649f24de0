/* cfen_hip.h -- C ABI of the MI355X-native CFEN-ViT dehazing inference path (libcfen_hip.so).
 *
 * Drop-in boundary for the generator forward of phoenixtreesky7/CFEN-ViT-Dehazing
 *   models/networks_iid_hlgvit_crs_gd4_cfs_v3.py:392-1020   dec_ipt.forward
 * called by the reference at models/model_iid_dehazing.py:143 (`self.netG(self.real_B)`), plus the
 * standalone deformable-convolution operator of dcn/src/deform_conv_cuda.cpp.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer on the current HIP device unless stated otherwise;
 *   - `dtype` selects storage + MFMA input type: CFEN_F32 (exact fp32 MFMA) or CFEN_F16 (fp16 storage,
 *     fp32 accumulate); biases / norm parameters / affine tables are always fp32;
 *   - feature maps are NHWC with an explicit channel stride `cs`; token matrices are row-major;
 *   - all functions are asynchronous on `stream` (a hipStream_t passed as void*), allocate nothing,
 *     return 0 on success or a negative CFEN_ERR_* code; cfen_last_error() gives the message.
 *   - no exceptions cross this boundary.
 */
#ifndef CFEN_HIP_H
#define CFEN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CFEN_F32 0
#define CFEN_F16 1

#define CFEN_OK 0
#define CFEN_ERR_ARG (-1)   /* invalid argument / unsupported shape      */
#define CFEN_ERR_HIP (-2)   /* HIP runtime reported an error             */
#define CFEN_ERR_STATE (-3) /* missing parameter, workspace too small... */

int cfen_abi_version(void);
const char* cfen_last_error(void);
/* Process-wide tuning knobs for benchmarking kernel variants (tools/bench_gemm.py); the defaults are what ships.
 *   "gemm.kernel": -1 choose by shape (default), 0 register-staged k_gemm_nt, 1 k_gemm_skinny, 2..5 LDS-DMA k_gemm_dma with a
 *                  96 x 128 / 96 / 64 / 32 (features x tokens) tile (1..5 need K * element size to be a multiple of 128 bytes)
 *   "embed.lds": bit 2 (default 6): the LDS-DMA-ring front half k_embed_qkv2; else k_embed_qkv stages its weights through LDS for D = 96 (bit 0) / D = 192 (bit 1)
 *   "gemm.splitk": 0 (default, round 4) / 1: K-heavy few-token GEMMs inside the net run split-K with the in-launch reduction (scratch from the workspace).
 *                  (off by default because the unsplit launches are faster with several forwards in flight; bitwise right there too)
 *   "mlp.small_tiles": fused-MLP kernel variant: 0..4 the register-staged k_mlp tilings, >= 10 (default 10) the LDS-DMA k_mlp2 (tens digit:
 *                  D = 96 variant, ones digit: D = 192 variant, see k_mlp.hip);  "net.fused_front_max_dim": largest LViT embedding dim that uses k_embed_qkv
 *   "net.skip_classes": bit mask of kernel classes the net does NOT launch (what-if timing only, outputs invalid)
 *   "net.attn_head_major": 1 (default) LViT levels with a fused front half pass qkv to attention per (window, head); 0 row-major [M][3D]
 *   "net.ln_fold": 1 (default) LN1 / LN2 of GViT and LViT level 3 are folded into the qkv / ffn1 GEMMs (cfen_gemm_ln), 0 separate LayerNorm launches
 *   "net.embed_gather": 1 (default) the LViT embedding GEMM gathers its patch tokens from the map, 0 separate patchify launch
 *                  (read when a forward is enqueued or a graph is built)
 *   "gemm.large" / "gemm.small": the k_gemm_dma tile (2..5) the shape rule uses for problems with >= / < 1024 tiles of 96 x 64
 *   "gemm.mid": the k_gemm_dma tile for launches of more than 512 tiles of 96 x 32 and fewer than 1024 of 96 x 64 (the grouped GViT-2 decoder GEMMs): 2 (default,
 *                  round 5) 96 x 128, 5 / 15 the 96 x 32 tiles of rounds 2-4 on 2 / 3 stages
 *   "net.tail_fused": the decoders' output tails (us_conv_d01* ConvTranspose, 3x3, reflect-pad 7x7 + tanh): 2 (default, round 5) ONE launch (k_tail_fused, both
 *                  intermediate maps in LDS), 1 ConvTranspose + 3x3 fused (k_up_conv3_fused) + the 7x7, 0 three launches -- bitwise equal;  "tail.segments" (default 1):
 *                  vertical segments a 64-column strip of k_tail_fused is cut into;  "tail.debug": timing experiments (results invalid)
 *   what-if probes (timing only, results invalid): "net.skip_from" / "net.skip_to" leave out the launches of that number range of a forward; "net.extra_launches" adds
 *                  that many one-workgroup launches in front of every ViT block; "net.gvit_dummy_wgs" / "_us" / "_levels" replace GViT blocks by a launch that only holds CUs
 *   "net.gvit_stream": GViT level 1 (embedding dim 384) on the LViT-3 stream kernels: 0 never, 1 in the single-lane plan only, 2 (default, round 5) on every
 *                  plan -- which kernels produce the outputs does not depend on the launch plan or on profiling
 *   "gvit.max_concurrent": 1 (default) .. 8: forwards of the persistent-chain plan ("net.gvit_chain") that may be in flight at once; the teams of all of them
 *                  must be resident together (grid barriers), so the host caps a team at 256 / (groups x this) CUs
 *   "mlp3.pair" (default 1, round 6): the D = 384 block (projection + LayerNorm + both stages) on k_mlp3p -- wave pairs, two waves per SIMD; 0 = k_mlp3<24, ...> (one wave per SIMD;
 *       results equal to 1-2 fp16 ulp: the LayerNorm sums associate over the pair); 2 = k_mlp3p's stamped timing build (prints to stderr, synchronises)
 *   "net.stream_mlp192" (default 1) / "mlp3.tm192" (default 22): LViT level 2's proj + MLP block on k_mlp3: 22 = two 4-wave 78 KB workgroups a CU on a three-slot ring
 *                  (256 registers); 24 four slots; 28 / 29 one 8-wave workgroup a CU; 2 / 3 / 4 token tiles a wave on one 150 KB workgroup a CU; net.stream_mlp192 = 0: k_mlp2
 *   "embed.stages": 2 .. 5 ring stages of k_embed_qkv2 at D = 192 (default 4)
 *   "lvit.shape": k_lvit_window's workgroup shape / schedule: 2 (default) 16 waves x 1 token tile, 0 8 x 2, 1 4 x 4, 3 denominator by MFMA, 4 / 5 the 16 x 1 / 8 x 2
 *                  shapes with hand-issued K / V fragment reads, 6 64-row front chunks (round 5: bitwise equal, not faster); 8 / 9 timing experiments without the softmax (results invalid) */
int cfen_tune(const char* key, int value);

/* ---- whole generator: replaces define_G (v3:93-100) + dec_ipt.forward (v3:392-1020) ------------- */
typedef struct cfen_net cfen_net;

typedef struct cfen_net_config {
  int32_t batch;            /* images per forward on this GPU                                   */
  int32_t n_feats;          /* options/base_options.py:110                                     */
  int32_t hidden_dim_ratio; /* options/base_options.py:104                                     */
  int32_t patch_size;       /* LViT window edge in feature-map pixels, base_options.py:96      */
  int32_t load_size;        /* edge of the half-resolution map xf, base_options.py:16          */
  int32_t num_heads;        /* base_options.py:192                                             */
  int32_t dtype;            /* CFEN_F32 | CFEN_F16                                             */
  int32_t reserved;         /* bit 0: single-lane launch plan; bits 8..15: generator variant (models/model_iid_dehazing.py:84-92):
                               0 networks_iid_hlgvit_crs_gd4_cfs_v3 (image edge 2*load_size), 1 networks_iid_hlgvit_crs_gd4_cfs
                               (full-resolution head, no ds_conv_e01 / us_conv_d01*, image edge = load_size)                    */
} cfen_net_config;

int cfen_net_create(cfen_net** out, const cfen_net_config* cfg);
void cfen_net_destroy(cfen_net* net);
/* bytes of device scratch the caller must pass to cfen_net_forward (256-byte aligned).  The workspace belongs to ONE net: the first forward on a
 * (workspace address, size) pair zeroes its synchronisation words with a launch on the caller's stream, graph capture zeroes them synchronously; the
 * caller must not write into the workspace between forwards, and hands a new net a workspace of its own (stage maps of the last forward live there:
 * cfen_net_stage). */
size_t cfen_net_workspace_bytes(const cfen_net* net);
/* register one packed parameter (layout: cfen_vit_dehazing_amd/packing.py); the pointer must stay
 * valid for the life of the net.  `nbytes` is checked against what the layer needs.               */
int cfen_net_set_param(cfen_net* net, const char* name, const void* dev_ptr, size_t nbytes);
/* ActNorm2d first-call initialisation (models/actnorm.py:25-37).  Mark `layer` (a conv / conv-transpose layer followed by
 * ActNorm2d, packing.py names) as uninitialised: the next EAGER cfen_net_forward runs it raw (`ones` [Cout_pad] of 1.0f as scale,
 * `conv_bias` [Cout_pad] as shift), takes per-channel statistics of that output over the whole batch, overwrites the layer's
 * registered "<layer>.scale" / "<layer>.shift" tables in place with the folded ActNorm epilogue and stores the raw parameters
 * an_out[0][c] = weight = -0.5 log(max(var_unbiased, 0.2)), an_out[1][c] = bias = -mean  (an_out is [2][Cout_pad] fp32), then runs the
 * layer normally -- the outputs of that forward are those of the reference's first forward.  Capture / profile refuse while
 * layers are pending.                                                                                                         */
int cfen_net_actnorm_pending(cfen_net* net, const char* layer, const float* ones, const float* conv_bias, float* an_out);
int cfen_net_actnorm_pending_count(const cfen_net* net);
/* Input format of cfen_net_forward / _graph_capture / _profile: 0 (default) x is fp32 NCHW in [-1,1]; 1: x is uint8 HWC
 * (B,H,W,3) as decoded from the image file, normalised (v/255 - 0.5)/0.5 on the device by the plan's first launch
 * (ToTensor + Normalize(0.5, 0.5), data/base_dataset.py:44-46).  Pass the uint8 pointer through the `x` argument.            */
int cfen_net_set_input_u8(cfen_net* net, int enabled);
/* Output format of cfen_net_forward / _graph_capture / _profile: 0 (default) xr / xs / xd are fp32 NCHW; 1: each of the three is uint8 HWC
 * (B,H,W,3) -- util.tensor2im of the fp32 result ((x + 1) / 2 * 255 in fp32, truncating cast, no clamp, the 1-channel xs tiled to 3;
 * util/util.py:12-24), written by the tails' last launch instead of a cfen_tensor2im_u8 pass over fp32 planes.  Pass the uint8 pointers through
 * the xr / xs / xd arguments (16-byte aligned).  CFEN_ERR_ARG when the tails do not run on the Toeplitz 7x7 kernel (fp32 nets, image edges that are
 * not multiples of 64): take fp32 outputs and cfen_tensor2im_u8 there.                                                                          */
int cfen_net_set_output_u8(cfen_net* net, int enabled);
/* enabled: the forward writes xr / xs / xd as fp16 NCHW (same shapes and [xr | xs | xd] order as the fp32 outputs, 2 bytes per element) from the fused tail launch --
 * the wire type of the sharded run's output all-gather (reference analogue: nn.DataParallel's gather, networks_iid_hlgvit_crs_gd4_cfs_v3.py:77-83), without a
 * conversion pass.  Values = the fp32 outputs rounded to nearest even.  CFEN_ERR_ARG where the tails do not run on the fused kernel (fp32 nets, image edges that
 * are not multiples of 64); cfen_net_forward then fails if the plan falls back to the separate tail launches ("net.tail_fused" != 2, "net.keep_stages").        */
int cfen_net_set_output_f16(cfen_net* net, int enabled);
/* names still missing, written as a ';'-separated list into buf; returns the count                */
int cfen_net_missing_params(const cfen_net* net, char* buf, size_t buflen);
/* x: (B,3,H,W) fp32 NCHW in [-1,1];  xr: (B,3,H,W), xs: (B,1,H,W), xd: (B,3,H,W) fp32 NCHW        */
int cfen_net_forward(cfen_net* net, const float* x, float* xr, float* xs, float* xd, void* workspace, size_t workspace_bytes,
                     void* stream);
/* The same launch plan as an explicit hipGraph: kernel nodes with dependency edges (the branches that eager mode
 * runs on internal side streams -- GViT beside LViT, S decoder beside R decoder -- become parallel graph branches).
 * All pointers are baked in; replay with cfen_net_graph_launch on any stream.  Returns an id in *graph_id.     */
int cfen_net_graph_capture(cfen_net* net, const float* x, float* xr, float* xs, float* xd, void* workspace, size_t workspace_bytes,
                           int32_t* graph_id);
int cfen_net_graph_launch(cfen_net* net, int32_t graph_id, void* stream);
/* one forward with a HIP event pair around every kernel launch on `stream` (synchronises the stream at
 * the end -- not graph-capturable).  Classes: 0 token GEMMs, 1 attention, 2 LayerNorm, 3 patchify /
 * unpatchify / upsample / layout, 4 convolutions, 5 InstanceNorm / CFSM2G, 6 fused token MLP.  Per class: summed kernel
 * milliseconds, algorithmic FLOPs (2*MAC) of those launches, number of launches.                  */
#define CFEN_NUM_KERNEL_CLASSES 7
int cfen_net_profile(cfen_net* net, const float* x, float* xr, float* xs, float* xd, void* workspace, size_t workspace_bytes,
                     void* stream, double* ms_per_class, double* flops_per_class, int32_t* launches_per_class, int nclass);
/* per-launch detail of the latest cfen_net_profile: entry `index` (launch order) -> layer / block-step label (owned by
 * the net, valid until the next profile), kernel class, algorithmic FLOPs, milliseconds.  Returns CFEN_ERR_STATE past
 * the last entry.                                                                                   */
int cfen_net_profile_entry(const cfen_net* net, int index, const char** label, int32_t* kernel_class, double* flops, double* ms);
/* ... and the DEVICE KERNEL(S) that launch ran (the kernel template as instantiated, e.g. "k_gemm_dma<T, 1, 3>"; several joined by " + ") with its
 * algorithmic bytes where the net prices it against HBM (token GEMMs: weights + tokens in + tokens out; 0 otherwise) */
int cfen_net_profile_entry_kernel(const cfen_net* net, int index, const char** kernel, double* bytes);
/* device pointer + geometry of a named top-level stage output (SURVEY Appendix D names) inside the
 * workspace of the LAST forward; NHWC, element type = net dtype.                                  */
int cfen_net_stage(const cfen_net* net, const char* name, const void** ptr, int32_t* C, int32_t* cs, int32_t* H, int32_t* W);
/* Device addresses of the 3 error words of the persistent GViT chains (csrc/k_gvit.hip): a word becomes 1 if a grid-barrier wait ever gave up
 * (never expected: the teams fit the chip); outputs of that forward are then invalid.  Valid after the first forward. */
int cfen_net_chain_error_words(const cfen_net* net, const void** words, int n);
/* 2*MAC count of one forward per image (SURVEY 8d closed form), for roofline reporting             */
double cfen_net_flops_per_image(const cfen_net* net);

/* ---- individual operators (unit-tested against the oracle; also what cfen_net_forward launches) -- */

/* Y[m][n] = act(sum_k X[m][k] W[n][k] + bias[n]) + R[m][n] + P[m % period][n]      (nn.Linear family) */
int cfen_gemm_nt(int dtype, const void* X, int ldx, const void* W, int ldw, const float* bias, const void* R, int ldr, const void* P,
                 int period, void* Y, int ldy, int M, int N, int K, int relu, void* stream);
/* Linear after a LayerNorm with the LayerNorm folded into the GEMM (TransformerEncoderLayer norm1 -> in_proj, norm2 -> linear1: v3:1383-1389):
 *   Y[m][n] = act(rstd_m (sum_k X[m][k] Wl[n][k] - mean_m s[n]) + bias[n]),  mean_m / rstd_m = LayerNorm statistics of row m of X over K,
 * with Wl = W * gamma (columns), s[n] = sum_k Wl[n][k] of the ROUNDED Wl, bias = W beta + b -- what cfen_net_forward runs for the blocks
 * that have no fused kernel (GViT, LViT level 3).  K * element size must be a multiple of 128 bytes; eps must be 1e-5. */
int cfen_gemm_ln(int dtype, const void* X, int ldx, const void* Wl, int ldw, const float* s, const float* bias, void* Y, int ldy, int M, int N,
                 int K, int relu, float eps, void* stream);
/* The same GEMMs with K cut into `nsplit` slices that run as separate workgroups (few tokens against a big matrix: the launch is bound by the
 * bytes one workgroup streams).  Every slice parks its fp32 partial tile in `scratch`; the workgroup that arrives last on the tile's counter adds
 * the slices in index order and applies the epilogue -- one launch, deterministic, no float atomics.  lnf_s != NULL: W is the LayerNorm-folded Wl of
 * cfen_gemm_ln.  scratch: 4096 bytes of arrival counters, which must be ZERO before the first call and are left zero by every call, followed by
 * ceil(N/96) * ceil(M/32) * nsplit * 14336 bytes of partial slabs.  K * element size / 128 must be divisible by nsplit.   (v3:1364, 1388-1389, 1173) */
int cfen_gemm_splitk(int dtype, const void* X, int ldx, const void* W, int ldw, const float* lnf_s, const float* bias, const void* R, int ldr, void* Y,
                      int ldy, int M, int N, int K, int relu, int nsplit, void* scratch, size_t scratch_bytes, void* stream);
/* head.0.0 read straight from the network input (csrc/k_head5.hip): out = act(conv5x5(in, pad 2) * scale + shift) as a 16-channel-stride NHWC
 * CFEN_F16 map; in = (B,3,H,W) fp32 NCHW (in_u8 = 0: what model.set_input hands over, models/model_iid_dehazing.py:143) or (B,H,W,3) uint8
 * (in_u8 = 1: ToTensor + Normalize(0.5, 0.5) folded in, data/base_dataset.py:44-46).  w5: [16][5][8][4] halfs (packing.pack_head5).  H % 8 == 0,
 * W % 64 == 0.  Replaces cfen_nchw_to_nhwc / cfen_u8hwc_to_nhwc + cfen_conv2d for that layer (v3:123-127, common.py:11-14). */
int cfen_head_conv5(int dtype, int in_u8, const void* in, const void* w5, const float* scale, const float* shift, void* out, int B, int H, int W,
                    int cs_out, int act, void* stream);
/* Persistent GEMM chain (csrc/k_gvit.hip): up to 5 DEPENDENT token GEMMs  y_p = act(x_p W_p^T + bias) + residual + pos[m % period]  (or the
 * LayerNorm-folded form of cfen_gemm_ln when lnf_s != NULL) run by ONE launch of `team` workgroups (one per CU) that keep their CUs and meet at a
 * grid barrier between phases -- the nn.Linear chains of a GViT instance: linear_encoding -> in_proj, and out_proj -> linear1 -> linear2 ->
 * mlp_head.0 -> mlp_head.3 (+ fold) (v3:1143,1166,1364,1386-1389,1173,1186).  CFEN_F16 only.  w_stream: the [N][K] weight as MFMA fragment stream
 * [N/16][K/32][64 lanes][8 halfs] (packing.pack_stream_tiles); N % 128 == 0, K % 64 == 0; nsplit > 1 cuts K into slices reduced inside the launch
 * (not with lnf_s).  fold = 1: y is the NHWC map (fold_H x fold_W pixels, channel stride fold_cs) the M tokens tile with fold_p x fold_p patches of
 * fold_C channels, feature n = (i, j, c).  A phase may read what an earlier phase of the same call wrote.  sync_ws: 8192 bytes of synchronisation
 * words (the call zeroes the ones it needs; word 1 is set to 1 if a wait ever gave up -- results are then invalid) followed by the split-K slabs:
 * ceil(M/128) * max(N/128 * nsplit) * 65536 bytes.  team * 1 <= 256: every workgroup must be resident for the grid barrier. */
typedef struct cfen_chain_phase {
  const void* x; const void* w_stream; const float* bias; const float* lnf_s; const void* residual; const void* pos; void* y;
  int32_t ldx, ldr, ldy, period, N, K, relu, nsplit, fold;
} cfen_chain_phase;
typedef struct cfen_chain_args {
  cfen_chain_phase phase[5];
  int32_t nphases, M;
  int32_t fold_H, fold_W, fold_cs, fold_C, fold_p;
  void* sync_ws; size_t sync_ws_bytes;
} cfen_chain_args;
int cfen_gemm_chain(int dtype, const cfen_chain_args* a, int team, void* stream);
/* LViT token embedding without a token buffer: tok = patchify(fmap) (window partition + unfold, as cfen_patchify with pool 1)
 * is gathered by the GEMM's loader;  Y[m][n] = sum_k tok[m][k] W[n][k] + bias[n] + tok[m][n] + pos[m % period][n],
 * D = p*p*C, W is [D][D] (ldw), Y is [M][D] (ldy).                                        (v3:1140-1143, 1166) */
int cfen_embed_gather(int dtype, const void* fmap, int B, int H, int W, int C, int cs, int ws, int p, const void* weight, int ldw,
                      const float* bias, const void* pos, int period, void* Y, int ldy, void* stream);
/* LViT front half in one launch (D = p*p*C in {96, 192}): tok = patchify(fmap);  y = We tok + be + tok + pos[m % S] -> x1 [M][D];
 * qkv = Wqkv LayerNorm(y) -> qkv [M][3D].  We [D][D] and Wqkv [3D][D] with the k axis in packing.kperm32 order for CFEN_F16.
 *                                                                                      (v3:1140-1143, 1166, 1364-1371) */
typedef struct cfen_embed_qkv_args {
  const void* fmap; int32_t B, H, W, C, cs, ws, p;
  const void* we; const float* be; const void* pos;
  const float* ln_gamma; const float* ln_beta;
  const void* wqkv;
  void* x1; void* qkv;
  float eps;
  int32_t head_major_heads;   /* 0: qkv is [M][3D] row-major; > 0 (= number of heads): qkv is written per (window, head),
                                 [(window * heads + head) * 3 + {q, k, v}][S][D / heads] -- the input of cfen_attention_head_major */
} cfen_embed_qkv_args;
int cfen_embed_qkv(int dtype, const cfen_embed_qkv_args* a, void* stream);
/* The same front half for D = 384 (LViT level 3; CFEN_F16, token count a multiple of 128) on the fragment-stream ring of cfen_mlp_stream_block:
 * `we` / `wqkv` are ROW-TILE fragment streams (packing.pack_stream_rows of the kperm32-slotted matrices): [rows / 32][2 row tiles][D / 32 k-chunks][1 KiB]. */
int cfen_embed_qkv_stream(int dtype, const cfen_embed_qkv_args* a, void* stream);
/* LayerNorm over the last dim (eps as given), gamma/beta fp32                       (v3:1370-1371) */
int cfen_layernorm(int dtype, const void* X, void* Y, const float* gamma, const float* beta, int M, int D, float eps, void* stream);
/* softmax(QK^T/sqrt(dh))V per (sequence, head); QKV is [nseq*S][3*heads*dh], out [nseq*S][heads*dh]  (v3:1364) */
int cfen_attention(int dtype, const void* qkv, void* out, int nseq, int S, int heads, int dh, void* stream);
/* the same attention reading the head-major qkv layout of cfen_embed_qkv (head_major_heads = heads): the q, k and v of one
 * (sequence, head) are contiguous [S][dh] blocks.  CFEN_F16, dh = 24, S in {64, 256} (every LViT window of the 512x512 configs) or 1024 (the
 * 1024x1024 configuration: K and V of a head stay in LDS, keys in blocks of 256 with the running-maximum rescale);
 * out stays [nseq*S][heads*dh] row-major.                                                                       (v3:1364) */
int cfen_attention_head_major(int dtype, const void* qkv, void* out, int nseq, int S, int heads, int dh, void* stream);
/* Fused token MLP block (D in {96,192}; H a multiple of 64 (fp16) / 32 (fp32)):
 *   y1 = x + W2a relu(W1a LN(x) + b1a) + b2a  (LN skipped when ln_gamma is NULL);  y2 = y1 + W2b relu(W1b y1 + b1b) + b2b
 *   (second stage skipped when W1b is NULL).  With att / w_proj ([M][D] / [D][D], natural k order) x is first replaced by
 *   x + w_proj att, the attention block's output projection and residual.  Output: token-major `y` and/or folded into the NHWC map `fmap`
 *   (F.fold + Join2x2: map H x W, channels C with stride cs, window ws, patch p).  Weights [H][D] / [D][H] with
 *   the k axis in packing.kperm32 order for CFEN_F16 (natural order for CFEN_F32).   (v3:1387-1389, 1173, 1186) */
typedef struct cfen_mlp_args {
  const void* x; void* y; void* fmap;
  const void* att; const void* w_proj;   /* optional prologue x <- x + w_proj att (out_proj + residual, v3:1386); both NULL = none */
  const float* ln_gamma; const float* ln_beta;
  const void* w1a; const float* b1a; const void* w2a; const float* b2a;
  const void* w1b; const float* b1b; const void* w2b; const float* b2b;
  int64_t M;
  int32_t D, H;
  float eps;
  int32_t mapH, mapW, C, cs, ws, p;
} cfen_mlp_args;
int cfen_mlp_block(int dtype, const cfen_mlp_args* a, void* stream);
/* The same block for the embedding dims the register-resident kernel cannot hold at two waves per SIMD (CFEN_F16; D = 384 with H <= 1536: LViT
 * level 3 / GViT level 1, and D = 192 with H <= 768), one wave per SIMD on the whole register file, weights as FRAGMENT STREAMS
 * (packing.pack_stream_pair / pack_stream_sq): wa_stream / wb_stream = [H/32][W1 slice, W2 slice][D/16 fragments][1 KiB], a fragment = the
 * 16 x 32 MFMA A operand in lane order (lane l: 16 bytes of row l & 15, k quarter l >> 4), k axes in packing.kperm32 order;
 * wp_stream = [D/32][D/16][1 KiB], natural k order.  Same arithmetic as cfen_mlp_block.                (v3:1386-1389, 1173, 1186) */
typedef struct cfen_mlp_stream_args {
  const void* x; void* y; void* fmap;
  const void* att; const void* wp_stream;
  const float* ln_gamma; const float* ln_beta;
  const void* wa_stream; const float* b1a; const float* b2a;
  const void* wb_stream; const float* b1b; const float* b2b;
  int64_t M;
  int32_t D, H;
  float eps;
  int32_t mapH, mapW, C, cs, ws, p;
} cfen_mlp_stream_args;
int cfen_mlp_stream_block(int dtype, const cfen_mlp_stream_args* a, void* stream);
/* One whole LViT block per workgroup-window (CFEN_F16, C = 24, p = 2, 32-pixel windows: 256 tokens of dim 96, 4 heads of 24):
 * map in -> Crop2x2 + unfold + linear_encoding + pos -> pre-LN MHA -> FFN -> mlp_head -> fold + Join2x2 -> map out, with q / k / v, the
 * attention output and both hidden activations never leaving the chip (K and V of the window live in LDS).  `w_stream`: every weight
 * matrix of the block as one stream of 1 KiB MFMA fragments in consumption order, packing.pack_lvit_window.                                                              (v3:1025-1056, 1136-1189, 1382-1390) */
typedef struct cfen_lvit_args {
  const void* fmap; void* out;
  int32_t B, H, W, C, cs_in, cs_out, ws, p;
  const void* w_stream; const float* be; const void* pos;
  const float* ln1_gamma; const float* ln1_beta;
  const float* ln2_gamma; const float* ln2_beta;
  const float* b1a; const float* b2a;
  const float* b1b; const float* b2b;
  int32_t hidden;
  float eps;
} cfen_lvit_args;
int cfen_lvit_window(int dtype, const cfen_lvit_args* a, void* stream);
/* window partition + unfold (+ optional 4x4 mean pool) / fold + window join          (v3:1025-1056,1140,1186,1274) */
int cfen_patchify(int dtype, const void* fmap, void* tokens, int B, int H, int W, int C, int cs, int ws, int p, int pool, void* stream);
int cfen_unpatchify(int dtype, const void* tokens, void* fmap, int B, int H, int W, int C, int cs, int ws, int p, void* stream);
/* two successive bilinear x2 upsamples, align_corners=False                           (v3:1323) */
int cfen_upsample4(int dtype, const void* small, void* out, int B, int h, int w, int C, int cs_in, int cs_out, void* stream);
int cfen_nchw_to_nhwc(int dtype, const float* in, void* out, int B, int C, int H, int W, int cs, void* stream);
/* pre / post-processing on the device:
 * uint8 (B,H,W,3) image -> NHWC T, channels zero-padded to cs, v -> ((v/255) - 0.5) / 0.5   (ToTensor + Normalize, data/base_dataset.py:44-46)
 * (1|3,H,W) fp32 in [-1,1] -> (H,W,3) uint8, (x+1)/2*255 truncated, 1 channel tiled to 3    (tensor2im, util/util.py:12-24)              */
int cfen_u8hwc_to_nhwc(int dtype, const unsigned char* in, void* out, int B, int H, int W, int cs, void* stream);
int cfen_tensor2im_u8(const float* in, unsigned char* out, int C, int H, int W, void* stream);

/* Conv2d / ConvTranspose2d(4,2,1) as implicit GEMM with fused affine + activation + residuals.
 * kind 0: Conv2d(k, stride, pad) over nsrc (1..3) channel-concatenated inputs (src0 | src1 | src2, the concat is never
 * materialised: v3:488 torch.cat((local, global), 1); crs_gd4:854 cat of three); kind 1: ConvTranspose2d k4 s2 p1.
 * weight: packed [nphase][Cout_pad][Kpad] (packing.py); scale/shift: [Cout_pad] fp32.
 * wlayout 0: k = tap*Cin + c (any geometry).  wlayout 1 ("rows", LDS-tiled kernel): stride 1, pad k/2, one source,
 * Cout <= 16, H % 8 == 0, W % 64 == 0, pixel stride 16/32/64 bytes; each kernel row is padded with zero taps to a
 * multiple of 64 bytes: k = (ky*KSP + kx)*cs_in + c, KSP = ceil(k*cs_in*esz/64)*64/(cs_in*esz), Kpad = k*KSP*cs_in.
 * wlayout 1 with kind 1 (fp16, Hin % 4 == 0, Win % 32 == 0, (cs_in, Cout_pad) in {(24,16),(48,32),(96,48)}): every tap's
 * channels zero-padded to a multiple of 64 bytes: [4 phases][Cout_pad][4 taps][CP], CP = ceil(cs_in*esz/64)*64/esz.        */
typedef struct cfen_conv_args {
  int32_t kind, k, stride, pad, reflect, nsrc;
  int32_t B, Hin, Win, Cin, cs_in;
  int32_t Cout, Cout_pad, Kpad, cs_out;
  int32_t act;            /* 0 none, 1 ReLU, 2 tanh */
  int32_t out_nchw_f32;   /* write (B,Cout,H,W) fp32 instead of NHWC */
  int32_t cs_res;
  int32_t wlayout;        /* 0 tap-major, 1 rows (see above), 2 Toeplitz 7x7 (fp16, cs_in 16, Cout <= 4, fp32 NCHW output, H % 16 == 0,
                             W % 64 == 0): [16 rows = co*4 + dxo][7][10 taps][16], w[co][ky][kx' - dxo][ci], Kpad 1120 */
  const void* src0;
  const void* src1;
  const void* weight;
  const float* scale;
  const float* shift;
  const void* res0;
  const void* res1;
  void* out;
  const void* src2;       /* third concatenated input (nsrc == 3), else NULL */
} cfen_conv_args;
int cfen_conv2d(int dtype, const cfen_conv_args* a, void* stream);

size_t cfen_stats_workspace(int B, int C);
/* InstanceNorm2d(affine=False) + ReLU in place                                        (v3:292-302) */
int cfen_instnorm_relu(int dtype, void* x, float* stats_ws, int B, int HW, int C, int cs, float eps, void* stream);
/* CFSM2G: out = x0 + x1*g1 + x2*g2, gates from global avg/max pools of x0+x1+x2          (v3:1481-1517)
 * w: fp32 [avg_cf1, avg_cf2, max_cf1, max_cf2] x {W0 (C/4 x C), W2 (C x C/4)}                      */
int cfen_cfsm2g(int dtype, const void* x0, const void* x1, const void* x2, void* out, const float* w, float* stats_ws, int B, int HW,
                int C, int cs, void* stream);

/* ---- deformable convolution (replaces the pybind module of dcn/src/deform_conv_cuda.cpp:681-695) ---- */
/* deform_conv_forward_cuda (dcn/src/deform_conv_cuda.cpp:151-156): NCHW tensors; note the reference passes W before H for
 * kernel/stride/pad/dilation here.  `columns` is the reference's scratch tensor of that name: device memory of at least
 * cfen_deform_conv_columns_bytes() bytes, 16-byte aligned (the kernel keeps an NHWC copy of the input and tap-major weights
 * there -- the column matrix itself is never materialised; the reference's `ones` buffer has no counterpart).  With columns ==
 * NULL (or too small, or C/group not a multiple of the 16-byte channel vector) a slower kernel gathers from the NCHW planes. */
size_t cfen_deform_conv_columns_bytes(int dtype, int B, int Cin, int H, int W, int Cout, int kH, int kW, int group);
int cfen_deform_conv_forward(int dtype, const void* input, const void* weight, const void* offset, void* output, int B, int Cin, int H,
                             int W, int Cout, int kW, int kH, int dW, int dH, int padW, int padH, int dilationW, int dilationH, int group,
                             int deformable_group, int im2col_step, void* columns, size_t columns_bytes, void* stream);
/* (extension, round 6) the same with `input` laid out [B][H][W][Cin] -- the memory of a torch channels_last tensor: the NHWC copy the kernel samples from is the input
 * itself, the layout pre-pass only re-orders the weights (12-14 us of a 107 us call at (8, 24, 256, 256)).  Needs `columns` (for the weights) and Cin / group a multiple
 * of the 16-byte channel vector: CFEN_ERR_ARG otherwise.  offset / mask / output stay NCHW as in the reference.                                                    */
int cfen_deform_conv_forward_nhwc(int dtype, const void* input, const void* weight, const void* offset, void* output, int B, int Cin, int H,
                                  int W, int Cout, int kW, int kH, int dW, int dH, int padW, int padH, int dilationW, int dilationH, int group,
                                  int deformable_group, int im2col_step, void* columns, size_t columns_bytes, void* stream);
int cfen_modulated_deform_conv_forward_nhwc(int dtype, const void* input, const void* weight, const void* bias, const void* offset,
                                            const void* mask, void* output, int B, int Cin, int H, int W, int Cout, int kernel_h, int kernel_w,
                                            int stride_h, int stride_w, int pad_h, int pad_w, int dilation_h, int dilation_w, int group,
                                            int deformable_group, int with_bias, void* columns, size_t columns_bytes, void* stream);
/* modulated_deform_conv_cuda_forward (dcn/src/deform_conv_cuda.cpp:486-492): h before w; bias may be NULL; columns as above */
int cfen_modulated_deform_conv_forward(int dtype, const void* input, const void* weight, const void* bias, const void* offset,
                                       const void* mask, void* output, int B, int Cin, int H, int W, int Cout, int kernel_h, int kernel_w,
                                       int stride_h, int stride_w, int pad_h, int pad_w, int dilation_h, int dilation_w, int group,
                                       int deformable_group, int with_bias, void* columns, size_t columns_bytes, void* stream);

/* ---- deformable convolution, backward direction ----
 * NCHW tensors of `dtype`, arithmetic in fp32.  `columns`: device scratch of at least cfen_deform_conv_backward_bytes() bytes, 16-byte
 * aligned (required: NHWC / transposed operand copies, the column-gradient matrix, fp32 accumulators).  gradInput / gradOffset / grad_mask
 * are OVERWRITTEN; gradWeight / grad_bias are ACCUMULATED INTO (the reference's addmm_ with beta = 1: its Python side passes zeros).
 * grad_input is summed with fp32 atomics like the reference's col2im (dcn/src/deform_conv_cuda_kernel.cu:322): reproducible up to
 * summation order. */
size_t cfen_deform_conv_backward_bytes(int B, int Cin, int H, int W, int Cout, int kH, int kW, int Hout, int Wout, int group);
/* col2im variant switch (process-wide, for A/B runs and tests): 1 (default) accumulates grad_input per 16 x 16 output-pixel tile in LDS, parks
 * each tile's footprint in `columns` and sums the overlapping footprints per input pixel in a fixed order; 2 flushes the tiles with global
 * atomics instead; 0 sends every add to global memory as the reference does; returns the previous setting */
int cfen_deform_conv_backward_set_lds(int enabled);
/* deform_conv_backward_input_cuda (dcn/src/deform_conv_cuda.cpp:260-265): W before H, as in the forward */
int cfen_deform_conv_backward_input(int dtype, const void* input, const void* offset, const void* gradOutput, void* gradInput, void* gradOffset,
                                    const void* weight, int B, int Cin, int H, int W, int Cout, int kW, int kH, int dW, int dH, int padW,
                                    int padH, int dilationW, int dilationH, int group, int deformable_group, int im2col_step, void* columns,
                                    size_t columns_bytes, void* stream);
/* deform_conv_backward_parameters_cuda (dcn/src/deform_conv_cuda.cpp:370-376): gradWeight += scale * d loss / d weight */
int cfen_deform_conv_backward_parameters(int dtype, const void* input, const void* offset, const void* gradOutput, void* gradWeight, int B,
                                         int Cin, int H, int W, int Cout, int kW, int kH, int dW, int dH, int padW, int padH, int dilationW,
                                         int dilationH, int group, int deformable_group, float scale, int im2col_step, void* columns,
                                         size_t columns_bytes, void* stream);
/* modulated_deform_conv_cuda_backward (dcn/src/deform_conv_cuda.cpp:566-574): h before w; `bias` is unused (shape only in the reference) */
int cfen_modulated_deform_conv_backward(int dtype, const void* input, const void* weight, const void* bias, const void* offset, const void* mask,
                                        void* grad_input, void* grad_weight, void* grad_bias, void* grad_offset, void* grad_mask,
                                        const void* grad_output, int B, int Cin, int H, int W, int Cout, int kernel_h, int kernel_w, int stride_h,
                                        int stride_w, int pad_h, int pad_w, int dilation_h, int dilation_w, int group, int deformable_group,
                                        int with_bias, void* columns, size_t columns_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CFEN_HIP_H */
